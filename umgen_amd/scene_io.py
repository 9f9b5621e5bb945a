"""Scene pickles <-> token dicts: the data format on the INPUT side of the rollout.

Mirrors, for the evaluation configuration only, what the reference does between a ``tokenized_origin_scenes`` pickle and
``UMGen.inference(input_cond_tokens=...)``:

  NuPlanTokenDataset (projects/plugin/data/datasets/UMGen_nuplan_dataset.py)
      get_frame_indices            :145-175   which frames of the clip are used (inference: start_index 10, stride sampling_gap)
      get_format_sceneior_data     :231-417   ego motion between sampled frames, per-frame boxes / categories / track ids,
                                               category + 64 m range filter, map and image token slices
  transforms_val (projects/configs/UMGen_config_evaluation.py:247-257)
      SplitAttriute / Normalize / MergeAttribute   common.py:117-157, normalize.py:79-137   min-max per attribute
      Normalize_Standard                           normalize.py:7-63                        ego: (v - 0) * float32(1/std)
      BBox3DTokenizer.__call__ + bbox_slotting     tokenizer.py:515-600, 809-952            bins + category token + 60 track slots
      DigitalBinsTokenizer.encode                  tokenizer.py:316-330                     np.digitize on linspace bins, clipped

and the artefact on the output side (projects/tools/model_pl.py:350-355): ``<name>_tokens.pkl``.

The (de)tokenisers and normalisers are native entry points of libumgen_hip.so (umgen_tokenize_ego / _boxes, umgen_detokenize_*;
csrc/tokenizers.hip), dtype for dtype what the reference's numpy does (float32 boxes, float64 ego motion, float64 bin edges), so
token ids are identical; the numpy statements of the same arithmetic below (`*_numpy`) are what the tests compare them with.
Frame selection, ego motion and the track slotting are host Python.  `tests/test_scene_io.py` pins everything on seeded synthetic
scenes against vectors recorded from the reference's own dataset class (`tests/golden/make_scene_golden.py`).  Control scenes (`data/controlled_scenes`) are
already token dicts and are passed through (UMGen_nuplan_dataset.py:196-200).
"""
from __future__ import annotations

import os
import pickle
from typing import Dict, List, Optional, Sequence

import numpy as np

import ctypes as C

from . import _lib
from .config import BBOX_PAD, BBOX_RANGE, EGO_STD, N_SLOTS, SLOT_LEN

CATEGORIES = ("vehicle", "bicycle", "pedestrian")          # projects/configs/category.txt
_BOX_BINS = np.linspace(0.0, 1.0, 1024)                     # agent_bins, config.py:147
_EGO_BINS = np.linspace(-1.0, 1.0, 1024)                    # bins_ego, config.py:121


def frame_indices(seq_len: int, block_size: int, sampling_gap: int = 4, start_index: int = 10) -> List[int]:
    """Frames of a clip used at inference (UMGen_nuplan_dataset.py:145-175 with inference_flag=True)."""
    max_start = seq_len - block_size * sampling_gap - sampling_gap
    if max_start < sampling_gap:     # clip shorter than the block: as many frames as fit
        max_start = sampling_gap
        block_size = (seq_len - sampling_gap - 1) // sampling_gap
    start = min(start_index, max_start)
    return [start + i * sampling_gap for i in range(block_size)]


def encode_bins(values: np.ndarray, bins: np.ndarray) -> np.ndarray:
    """DigitalBinsTokenizer.encode (tokenizer.py:316-330) without special tokens: digitize, clip to the vocabulary."""
    return np.clip(np.digitize(values, bins), 0, bins.shape[0] - 1)


def ego_motion(meta_info: Sequence[dict], ego_pose_all: np.ndarray, indices: Sequence[int], sampling_gap: int) -> np.ndarray:
    """(dx, dy, dheading) from the frame before each sampled frame to that frame, in the earlier frame's lidar coordinates
    (UMGen_nuplan_dataset.py:252-280)."""
    out = []
    for i, fi in enumerate(indices):
        index = fi - sampling_gap if i == 0 else indices[i - 1]
        if index < 0:
            raise ValueError("first sampled frame has no predecessor at distance sampling_gap")
        tr = np.linalg.inv(meta_info[index]["T_lidar2global"]) @ (
            meta_info[index + sampling_gap]["T_lidar2global"] @ np.array([0, 0, 0, 1.0]).T)
        dh = ego_pose_all[index + sampling_gap, 6] - ego_pose_all[index, 6]
        if dh >= np.pi:
            dh -= 2 * np.pi
        if dh < -np.pi:
            dh += 2 * np.pi
        out.append([tr[0], tr[1], dh])
    return np.asarray(out)


def _ptr(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def _native(rc: int, what: str):
    if rc != 0:
        raise ValueError(f"{what}: libumgen_hip returned {rc}")


def encode_ego(pose_diff: np.ndarray) -> np.ndarray:
    """Normalize_Standard(mean 0, std [10, 4, 1]) then the 1024-bin tokenizer over [-1, 1]  ->  [T, 3] tokens (native)."""
    pd = np.ascontiguousarray(pose_diff, dtype=np.float64).reshape(-1, 3)
    out = np.empty(pd.shape, dtype=np.int64)
    _native(_lib.load_host_library().umgen_tokenize_ego(_ptr(pd, C.c_double), pd.shape[0], _ptr(out, C.c_int64)), "umgen_tokenize_ego")
    return out.reshape(np.shape(pose_diff))


def decode_ego(pose_tokens: np.ndarray) -> np.ndarray:
    """Tokens [..., 3] -> (dx, dy, dheading) float32 -- what UMGen.decode_pose returns (native)."""
    t = np.ascontiguousarray(pose_tokens, dtype=np.int64).reshape(-1, 3)
    out = np.empty(t.shape, dtype=np.float32)
    _native(_lib.load_host_library().umgen_detokenize_ego(_ptr(t, C.c_int64), t.shape[0], _ptr(out, C.c_float)), "umgen_detokenize_ego")
    return out.reshape(np.shape(pose_tokens))


def encode_ego_numpy(pose_diff: np.ndarray) -> np.ndarray:
    """numpy statement of encode_ego (reference arithmetic; used by the tests)."""
    inv_std = 1.0 / np.array(EGO_STD, dtype=np.float32)           # float32 reciprocal, as normalize.py:26
    mean = np.array([0, 0, 0], dtype=np.float32)
    return encode_bins((pose_diff - mean) * inv_std, _EGO_BINS).astype(np.int64)


def decode_ego_numpy(pose_tokens: np.ndarray) -> np.ndarray:
    """Tokens [..., 3] -> (dx, dy, dheading) float32: bin mid-points (DigitalBinsTokenizer.decode, tokenizer.py:332-354) divided by
    the float32 reciprocal std (Normalize_Standard.unnormalize_ego, normalize.py:65-76) -- what UMGen.decode_pose returns."""
    t = np.asarray(pose_tokens, dtype=np.int64)
    v = (_EGO_BINS[np.clip(t - 1, 0, 1023)] + _EGO_BINS[np.clip(t, 0, 1023)]) / 2
    inv_std = 1.0 / np.array(EGO_STD, dtype=np.float32)
    return (v / inv_std + np.zeros(3, dtype=np.float32)).astype(np.float32)


def decode_boxes(bbox3d_tokens: np.ndarray):
    """Frame tokens [660] -> (boxes float64 [n, 10], categories list[str], slot indices), native attribute decode."""
    t = np.asarray(bbox3d_tokens, dtype=np.int64).reshape(N_SLOTS, SLOT_LEN)
    keep = np.nonzero(~np.any(t == BBOX_PAD, axis=1))[0]
    sl = np.ascontiguousarray(t[keep])
    out = np.empty((sl.shape[0], 10), dtype=np.float64)
    _native(_lib.load_host_library().umgen_detokenize_boxes(_ptr(sl, C.c_int64), sl.shape[0], _ptr(out, C.c_double)), "umgen_detokenize_boxes")
    cats = [CATEGORIES[c - 1024] if 1024 <= c < 1024 + len(CATEGORIES) else "none" for c in t[keep, 10].tolist()]
    return out, cats, keep


def decode_boxes_numpy(bbox3d_tokens: np.ndarray):
    """Frame tokens [660] -> (boxes float64 [n, 10], categories list[str], slot indices): slots holding any pad token are dropped
    (BBox3DTokenizer.decode, tokenizer.py:689-806), attributes are bin mid-points mapped back through their min-max ranges
    (Normalize.unnormalize_bbox3d, normalize.py:189-229)."""
    t = np.asarray(bbox3d_tokens, dtype=np.int64).reshape(N_SLOTS, SLOT_LEN)
    keep = np.nonzero(~np.any(t == BBOX_PAD, axis=1))[0]
    a = t[keep, :10]
    v = (_BOX_BINS[np.clip(a - 1, 0, 1023)] + _BOX_BINS[np.clip(a, 0, 1023)]) / 2
    lo = np.array([r[0] for r in BBOX_RANGE], dtype=np.float64)
    hi = np.array([r[1] for r in BBOX_RANGE], dtype=np.float64)
    cats = [CATEGORIES[c - 1024] if 1024 <= c < 1024 + len(CATEGORIES) else "none" for c in t[keep, 10].tolist()]
    return v * (hi - lo) + lo, cats, keep


def filter_boxes(boxes: Sequence, cats: Sequence[Sequence[str]], track_ids: Sequence, vocab: Sequence[str] = CATEGORIES):
    """Keep boxes whose category is in the vocabulary and whose centre is within 64 m in x and y
    (categories_fliter + range filter, UMGen_nuplan_dataset.py:317-346)."""
    fb, fc, ft = [], [], []
    for b, c, t in zip(boxes, cats, track_ids):
        b = np.array(b).astype(np.float32)
        keep = [j for j in range(len(c)) if c[j] in vocab and not (abs(b[j][0]) > 64 or abs(b[j][1]) > 64)]
        fb.append(b[keep])
        fc.append(np.array(c)[keep].tolist())
        ft.append(np.array(t)[keep])
    return fb, fc, ft


def box_tokens(b: np.ndarray, cat_index: Sequence[int]) -> np.ndarray:
    """Boxes [n, >=10] float32 + vocabulary indices -> [n, 11] tokens (min-max normalise, bin, category token): native."""
    b = np.ascontiguousarray(np.atleast_2d(b), dtype=np.float32)
    ci = np.ascontiguousarray(cat_index, dtype=np.int32)
    out = np.empty((b.shape[0], SLOT_LEN), dtype=np.int64)
    _native(_lib.load_host_library().umgen_tokenize_boxes(_ptr(b, C.c_float), b.shape[0], b.shape[1], _ptr(ci, C.c_int32), _ptr(out, C.c_int64)),
            "umgen_tokenize_boxes")
    return out


def box_tokens_numpy(b: np.ndarray, cat_index: Sequence[int]) -> np.ndarray:
    """numpy statement of box_tokens (Normalize.normalize_* + DigitalBinsTokenizer.encode per attribute; used by the tests)."""
    b = np.atleast_2d(b)
    cols = [encode_bins((b[:, a] - lo) / (hi - lo), _BOX_BINS) for a, (lo, hi) in enumerate(BBOX_RANGE)]
    return np.concatenate([np.stack(cols, axis=-1), (np.asarray(cat_index) + 1024)[:, None]], axis=-1).astype(np.int64)


def encode_boxes(boxes: Sequence[np.ndarray], cats: Sequence[Sequence[str]], track_ids: Sequence[np.ndarray],
                 vocab: Sequence[str] = CATEGORIES) -> np.ndarray:
    """Per-frame boxes [n_t, >=10] (float32), categories and track ids -> [T, 660] tokens: each of the first 10 attributes is
    min-max normalised with its range (config.py:126-137), binned over linspace(0, 1, 1024); the category token is
    1024 + vocabulary index; a clip-wide slot is given to each track id in order of first appearance (60 at most, later ids are
    dropped); empty slots are 11 x pad (1027)."""
    per_frame = []
    for b, c in zip(boxes, cats):
        if len(c) == 0:
            per_frame.append(np.zeros((0, SLOT_LEN), dtype=np.int64))
            continue
        per_frame.append(box_tokens(b, [vocab.index(x) for x in c]))
    # bbox_slotting (tokenizer.py:809-952): np.any() decides whether a frame "has" boxes, so a frame whose ids are all 0 counts
    # as empty -- kept as is
    ids = np.concatenate([np.asarray(t)[:] if np.any(t) else np.array([]) for t in track_ids]) if len(track_ids) else np.array([])
    if np.any(ids):
        _, first = np.unique(ids, return_index=True)
        ids = ids[np.sort(first)]
    if ids.size > N_SLOTS:
        ids = ids[:N_SLOTS]
    slot_of = {tid: i for i, tid in enumerate(ids)}
    out = np.full((len(per_frame), N_SLOTS, SLOT_LEN), BBOX_PAD, dtype=np.int64)
    for f, (tok, tids) in enumerate(zip(per_frame, track_ids)):
        tids = np.asarray(tids)
        if not np.any(tids):
            continue
        keep = [i for i, t in enumerate(tids) if t in slot_of]
        if not np.any(tids[keep]):
            continue
        out[f, [slot_of[t] for t in tids[keep]]] = tok[keep]
    return out.reshape(len(per_frame), N_SLOTS * SLOT_LEN)


def scene_tokens(frame_data: dict, block_size: int, sampling_gap: int = 4, start_index: int = 10, view: str = "CAM_F0",
                 vocab: Sequence[str] = CATEGORIES) -> Dict[str, np.ndarray]:
    """One raw scene pickle (already loaded) -> {"pose" [T,3], "map" [T,1024], "bbox3d" [T,660], "image" [T,512]} int64."""
    image = np.stack(frame_data["tokens"][view]["tokens"], axis=0)
    idx = frame_indices(image.shape[0], block_size, sampling_gap, start_index)
    meta = frame_data["meta_info"]
    pose = encode_ego(ego_motion(meta, np.asarray(frame_data["ego_pose_all"]), idx, sampling_gap))
    boxes, cats, tids = filter_boxes([meta[i]["bboxes_3d"] for i in idx], [meta[i]["categories"] for i in idx],
                                     [meta[i]["track_ids"] for i in idx], vocab)
    box = encode_boxes(boxes, cats, tids, vocab)
    mp = np.asarray(frame_data["raster_tokens"])[idx]
    return {"pose": pose, "map": mp.reshape(mp.shape[0], -1).astype(np.int64), "bbox3d": box,
            "image": image[idx].reshape(len(idx), -1).astype(np.int64)}


class SceneReader:
    """Counterpart of ``NuPlanTokenDataset(..., inference_flag=True, return_scene_name=True)`` as evaluate.py configures it
    (infer_fun.py:160-207): sorted ``*.pkl`` under the roots; item = token dict (+ "file_name")."""

    def __init__(self, data_root, block_size: int, sampling_gap: int = 4, start_index: int = 10, views: Sequence[str] = ("CAM_F0",),
                 categories: Sequence[str] = CATEGORIES, control_test: bool = False):
        roots = [data_root] if isinstance(data_root, str) else list(data_root)
        self.files: List[str] = []
        for path in roots:
            if os.path.isfile(path) and path.endswith(".pkl"):
                self.files.append(path)
            elif os.path.isdir(path):
                self.files += [os.path.join(path, f) for f in os.listdir(path) if f.endswith(".pkl")]
        self.files.sort()
        self.block_size, self.sampling_gap, self.start_index = block_size, sampling_gap, start_index
        self.view, self.categories, self.control_test = views[0], tuple(categories), control_test

    def __len__(self):
        return len(self.files)

    def __getitem__(self, i: int) -> dict:
        with open(self.files[i], "rb") as f:
            raw = pickle.load(f)
        if self.control_test:                       # already a token dict (UMGen_nuplan_dataset.py:196-200)
            return raw
        data = scene_tokens(raw, self.block_size, self.sampling_gap, self.start_index, self.view, self.categories)
        data["file_name"] = f"{i}_{self.files[i]}"  # UMGen_nuplan_dataset.py:415
        return data


def is_raw_scene(obj) -> bool:
    return isinstance(obj, dict) and "meta_info" in obj and "tokens" in obj and "raster_tokens" in obj


def save_tokens(out_tokens: Dict[str, np.ndarray], output_path: str, name: str) -> Optional[str]:
    """model_pl.save_tokens (model_pl.py:350-355): ``<output_path>/saved_token/<name>_tokens.pkl``; existing files are kept
    (model_pl.py:215-216).  Returns the path written, or None when skipped."""
    d = os.path.join(output_path, "saved_token")
    os.makedirs(d, exist_ok=True)
    p = os.path.join(d, f"{name}_tokens.pkl")
    if os.path.exists(p):
        return None
    with open(p, "wb") as f:
        pickle.dump({k: np.asarray(v) for k, v in out_tokens.items()}, f)
    return p


# ---------------------------------------------------------------------------------------------------------------------
# seeded synthetic RAW scenes (the real nuPlan-derived pickles are not available offline): same schema, small, and built to hit
# the reader's edge cases -- agents beyond 64 m, categories outside the vocabulary, more than 60 tracks, empty frames, track id 0
# ---------------------------------------------------------------------------------------------------------------------
def synthetic_raw_scene(seed: int, n_frames: int = 120, n_tracks: int = 70) -> dict:
    rng = np.random.Generator(np.random.PCG64(9000 + seed))
    heading = np.cumsum(rng.normal(0, 0.02, n_frames)) + rng.uniform(-np.pi, np.pi)
    heading = (heading + np.pi) % (2 * np.pi) - np.pi                   # wraps across +-pi inside the clip
    speed = np.abs(rng.normal(1.0, 0.5, n_frames))
    xy = np.cumsum(np.stack([np.cos(heading), np.sin(heading)], 1) * speed[:, None], axis=0)
    ego = np.zeros((n_frames, 16))
    ego[:, 0:2] = xy
    ego[:, 6] = heading
    cats_all = ["vehicle", "bicycle", "pedestrian", "barrier", "czone_sign"]
    track_cat = [cats_all[int(c)] for c in rng.integers(0, len(cats_all), n_tracks)]
    track_pos = rng.uniform(-80, 80, (n_tracks, 2))
    track_vel = rng.normal(0, 0.3, (n_tracks, 2))
    born = rng.integers(0, n_frames // 2, n_tracks)
    meta, lidar = [], {"bboxes_3d": [], "categories": [], "track_ids": []}
    for t in range(n_frames):
        T = np.eye(4)
        c, s = np.cos(heading[t]), np.sin(heading[t])
        T[:2, :2] = [[c, -s], [s, c]]
        T[:2, 3] = xy[t]
        alive = [k for k in range(n_tracks) if born[k] <= t] if t % 17 != 5 else []      # some frames have no agents at all
        boxes = []
        for k in alive:
            p = track_pos[k] + track_vel[k] * (t - born[k])
            boxes.append([p[0], p[1], rng.uniform(-2, 2), rng.uniform(0.5, 12), rng.uniform(0.3, 3), rng.uniform(0.5, 4),
                          rng.uniform(-3.2, 3.2), rng.normal(0, 6), rng.normal(0, 4), rng.normal(0, 0.1), 0.0, 0.0])
        meta.append({"T_lidar2global": T, "bboxes_3d": np.array(boxes, dtype=np.float64).reshape(len(alive), 12),
                     "track_ids": np.array(alive, dtype=np.int64), "categories": [track_cat[k] for k in alive]})
        lidar["bboxes_3d"].append(np.zeros((0, 4)))
        lidar["categories"].append([])
        lidar["track_ids"].append(np.zeros((0,), dtype=np.int64))
    return {
        "tokens": {"CAM_F0": {"tokens": [rng.integers(0, 8192, (16, 32), dtype=np.int64) for _ in range(n_frames)],
                              "file_list": [f"{t:06d}.jpg" for t in range(n_frames)]}},
        "ego_pose_all": ego, "meta_info": meta, "raster_tokens": rng.integers(0, 8192, (n_frames, 32, 32), dtype=np.int64),
        "lidar_bboxes": {"CAM_F0": lidar},
    }
