"""Scene pickle reader / token writer against vectors recorded from the reference's own
dataset class and transforms (tests/golden/make_scene_golden.py), plus the pieces in isolation."""
import os
import pickle

import numpy as np
import pytest

from umgen_amd import scene_io
from umgen_amd.config import BBOX_PAD, MOD_ORDER

GOLD = os.path.join(os.path.dirname(__file__), "golden", "scene_reader.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_reader_reproduces_the_reference_dataset_tokens(gold, tmp_path):
    for seed, n_frames, block, n_tracks in gold["cases"].tolist():
        path = tmp_path / f"scene_{seed}_synthetic_clip_000.pkl"
        with open(path, "wb") as f:
            pickle.dump(scene_io.synthetic_raw_scene(seed, n_frames, n_tracks), f)
        item = scene_io.SceneReader(str(path), block_size=block)[0]
        for m in MOD_ORDER:
            np.testing.assert_array_equal(item[m], gold[f"s{seed}_{m}"].astype(np.int64), err_msg=f"seed {seed} {m}")
            assert item[m].dtype == np.int64
        assert item["file_name"].endswith(str(path))


def test_goldens_cover_the_edge_cases(gold):
    """The recorded scenes exercise: a clip shorter than the block, >60 tracks (60 slots full, later tracks dropped), frames
    without any agent, agents filtered for range / category."""
    cases = {c[0]: c for c in gold["cases"].tolist()}
    assert gold["s2_pose"].shape[0] == (60 - 4 - 1) // 4 < cases[2][2]
    slots3 = gold["s3_bbox3d"].reshape(-1, 60, 11)
    assert (slots3[..., 10] != BBOX_PAD).any(0).all()                      # all 60 slots taken
    assert any((gold[f"s{s}_bbox3d"] == BBOX_PAD).all(1).any() for s in cases)   # a frame with no agents
    for s in cases:
        b = gold[f"s{s}_bbox3d"].reshape(-1, 60, 11)
        full = b[..., 10] != BBOX_PAD
        assert ((b[full][:, :10] >= 0) & (b[full][:, :10] <= 1023)).all() and ((b[full][:, 10] >= 1024) & (b[full][:, 10] <= 1026)).all()
        assert (b[~full] == BBOX_PAD).all()


def test_frame_indices_inference_rule():
    # long clip: start at 10, stride 4 (UMGen_nuplan_dataset.py:145-175)
    assert scene_io.frame_indices(200, 22, 4, 10) == [10 + 4 * i for i in range(22)]
    # the start moves back when the block would run past the clip
    assert scene_io.frame_indices(100, 22, 4, 10)[0] == 100 - 22 * 4 - 4
    # clip shorter than the block: start = sampling_gap, as many frames as fit
    assert scene_io.frame_indices(60, 22, 4, 10) == [4 + 4 * i for i in range(13)]


def test_bin_encoding_edges():
    bins = np.linspace(0.0, 1.0, 1024)
    v = np.array([-5.0, 0.0, 1e-9, 0.5, 1.0, 7.0])
    t = scene_io.encode_bins(v, bins)
    assert t.tolist() == [0, 1, 1, 512, 1023, 1023]      # below range -> 0, above -> clipped to the last bin
    # ego: (v - 0) * float32(1 / std) over linspace(-1, 1): zero motion sits in the centre bin
    assert scene_io.encode_ego(np.zeros((1, 3))).tolist() == [[512, 512, 512]]
    assert scene_io.encode_ego(np.array([[100.0, -100.0, 0.999]])).tolist() == [[1023, 0, 1023]]


def test_slotting_by_first_appearance_and_track_id_zero_quirk():
    box = lambda x: np.array([[x, 0, 0, 4, 2, 1.5, 0, 0, 0, 0]], dtype=np.float32)   # noqa: E731
    boxes = [box(1.0), np.concatenate([box(2.0), box(3.0)]), box(4.0), np.zeros((0, 10), np.float32)]
    cats = [["vehicle"], ["pedestrian", "vehicle"], ["bicycle"], []]
    tids = [np.array([7]), np.array([9, 7]), np.array([0]), np.array([], dtype=np.int64)]
    tok = scene_io.encode_boxes(boxes, cats, tids).reshape(4, 60, 11)
    assert tok[0, 0, 10] == 1024 and (tok[0, 1:] == BBOX_PAD).all()          # track 7 -> slot 0
    assert tok[1, 0, 10] == 1024 and tok[1, 1, 10] == 1026                    # track 9 appears second -> slot 1
    assert (tok[2] == BBOX_PAD).all()      # a frame whose only track id is 0 counts as empty (np.any, tokenizer.py:880-886)
    assert (tok[3] == BBOX_PAD).all()


def test_token_pickle_writer_skips_existing(tmp_path):
    out = {m: np.zeros((1, 3, 4), np.int64) for m in MOD_ORDER}
    p = scene_io.save_tokens(out, str(tmp_path), "clip_a")
    assert p == str(tmp_path / "saved_token" / "clip_a_tokens.pkl")
    with open(p, "rb") as f:
        back = pickle.load(f)
    assert set(back) == set(MOD_ORDER) and back["pose"].dtype == np.int64
    assert scene_io.save_tokens(out, str(tmp_path), "clip_a") is None          # model_pl.py:215-216


@pytest.mark.skipif(not os.path.isdir("/root/reference/projects"), reason="reference checkout not present")
def test_live_against_reference_dataset_on_a_fresh_seed(tmp_path):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_scene_golden
    ref = make_scene_golden.reference_tokens(11, 150, 30, 120)
    path = tmp_path / "scene_11_synthetic_clip_000.pkl"
    with open(path, "wb") as f:
        pickle.dump(scene_io.synthetic_raw_scene(11, 150, 120), f)
    item = scene_io.SceneReader(str(path), block_size=30)[0]
    for m in MOD_ORDER:
        np.testing.assert_array_equal(item[m], ref[m], err_msg=m)


def test_evaluate_loads_raw_clips_through_the_reader(tmp_path, gold):
    """`python -m umgen_amd.evaluate --data_test_root <dir of raw clips>`: load_scene tokenises a raw clip on the fly with the
    block size evaluate.py derives (set_num_new_frames + 20)."""
    from umgen_amd.evaluate import load_scene
    seed, n_frames, block, n_tracks = gold["cases"].tolist()[0]
    path = tmp_path / "clip.pkl"
    with open(path, "wb") as f:
        pickle.dump(scene_io.synthetic_raw_scene(seed, n_frames, n_tracks), f)
    sc, ctl = load_scene(str(path), block_size=block)
    assert ctl is None
    for m in MOD_ORDER:
        np.testing.assert_array_equal(sc[m][0], gold[f"s{seed}_{m}"].astype(np.int64))


def test_decoders_match_the_oracle_helpers_and_round_trip(gold):
    """decode side of f-1: same numbers as the oracle's helpers (which are pinned on the reference's tokenizer / normaliser in
    tests/test_oracle_vs_reference.py), and encode(decode(tokens)) == tokens for every recorded in-vocabulary token."""
    from oracle.umgen_oracle import decode_box_values, decode_pose_values
    pose = gold["s0_pose"].astype(np.int64)
    np.testing.assert_array_equal(scene_io.decode_ego(pose), decode_pose_values(pose))
    frame = gold["s3_bbox3d"][5].astype(np.int64)
    boxes, cats, slots = scene_io.decode_boxes(frame)
    assert len(slots) > 0 and set(cats) <= set(scene_io.CATEGORIES)
    for b, s in zip(boxes, slots):
        np.testing.assert_array_equal(b, decode_box_values(frame.reshape(60, 11)[s]))
    # round trip through the encoder (bin mid-points fall back into their own bin; tokens 0 / 1023 are the open-ended bins)
    re = scene_io.encode_boxes([boxes.astype(np.float32)], [cats], [np.arange(1, len(cats) + 1)]).reshape(60, 11)[:len(cats)]
    orig = frame.reshape(60, 11)[slots]
    inner = (orig[:, :10] > 0) & (orig[:, :10] < 1023)
    np.testing.assert_array_equal(re[:, :10][inner], orig[:, :10][inner])
    np.testing.assert_array_equal(re[:, 10], orig[:, 10])


def test_native_tokenizers_equal_the_numpy_statement_of_the_reference_arithmetic():
    """f-1: umgen_tokenize_* / umgen_detokenize_* (csrc/tokenizers.hip) against the numpy restatement of DigitalBinsTokenizer /
    Normalize / Normalize_Standard, on random values incl. exact bin edges, out-of-range values and every token id."""
    rng = np.random.default_rng(0)
    edges = np.linspace(-1.0, 1.0, 1024)
    pd = np.concatenate([rng.normal(0, 6, (4000, 3)), (edges[:1023, None] * np.array([10.0, 4.0, 1.0]))[:1000],
                         np.array([[1e9, -1e9, 0.0], [10.0, 4.0, 1.0], [-10.0, -4.0, -1.0]])])
    np.testing.assert_array_equal(scene_io.encode_ego(pd), scene_io.encode_ego_numpy(pd))
    toks = np.stack([np.arange(1024)] * 3, axis=1)
    np.testing.assert_array_equal(scene_io.decode_ego(toks), scene_io.decode_ego_numpy(toks))
    lo = np.array([r[0] for r in scene_io.BBOX_RANGE]); hi = np.array([r[1] for r in scene_io.BBOX_RANGE])
    b = (rng.uniform(-0.1, 1.1, (5000, 12)) * np.concatenate([hi - lo, [1, 1]]) + np.concatenate([lo, [0, 0]])).astype(np.float32)
    b[:1024, :10] = (np.linspace(0.0, 1.0, 1024)[:, None] * (hi - lo) + lo).astype(np.float32)          # exact edges
    ci = rng.integers(0, 3, 5000)
    np.testing.assert_array_equal(scene_io.box_tokens(b, ci), scene_io.box_tokens_numpy(b, ci))
    frame = np.full((60, 11), BBOX_PAD, dtype=np.int64)
    frame[:40, :10] = rng.integers(0, 1024, (40, 10)); frame[:40, 10] = rng.integers(1024, 1027, 40)
    frame[:4, :10] = np.array([0, 1, 1022, 1023])[:, None]
    got, want = scene_io.decode_boxes(frame.reshape(-1)), scene_io.decode_boxes_numpy(frame.reshape(-1))
    np.testing.assert_array_equal(got[0], want[0]); assert got[1] == want[1]; np.testing.assert_array_equal(got[2], want[2])
