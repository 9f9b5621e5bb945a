"""Records what the REFERENCE's own dataset class + transforms_val produce for seeded synthetic raw scenes
(build container only: imports /root/reference through tests/golden/refimport.py).

    python tests/golden/make_scene_golden.py      ->  tests/golden/scene_reader.npz

The raw scenes are regenerated from their seeds by `umgen_amd.scene_io.synthetic_raw_scene`, so only the expected tokens are
stored.  tests/test_scene_io.py compares `umgen_amd.scene_io.SceneReader` with these vectors.
"""
import os
import pickle
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

import refimport  # noqa: E402
from umgen_amd.scene_io import synthetic_raw_scene  # noqa: E402

CASES = [  # (seed, n_frames of the clip, block_size, tracks in the clip)
    (0, 120, 22, 70), (2, 60, 22, 70),            # the second clip is shorter than the block: fewer frames come back
    (3, 200, 40, 400),                            # more than 60 in-range tracks of known categories: later ones are dropped
]


def reference_tokens(seed, n_frames, block_size, n_tracks):
    refimport.install_stubs()
    with refimport.reference_cwd():
        import projects.configs.UMGen_config_evaluation as cfg
        from projects.plugin.data.datasets.UMGen_nuplan_dataset import NuPlanTokenDataset
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, f"scene_{seed}_synthetic_clip_000.pkl")
            with open(path, "wb") as f:
                pickle.dump(synthetic_raw_scene(seed, n_frames, n_tracks), f)
            ds = NuPlanTokenDataset(data_root=[d], training=False, block_size=block_size, categories_file="projects/configs/category.txt",
                                    views=["CAM_F0"], sampling_gap=cfg.sampling_gap, transforms=cfg.transforms_val, inference_flag=True,
                                    start_index=10, sample_img=True, return_scene_name=True, control_test=False)
            item = ds[0]
    return {m: np.asarray(item[m]).astype(np.int64) for m in ("pose", "map", "bbox3d", "image")}


def main():
    out = {"cases": np.array(CASES, dtype=np.int64)}
    for seed, n_frames, block, n_tracks in CASES:
        tok = reference_tokens(seed, n_frames, block, n_tracks)
        for m, a in tok.items():
            out[f"s{seed}_{m}"] = a.astype(np.int16 if a.max() < 32768 else np.int32)
        print(seed, {m: a.shape for m, a in tok.items()}, "slots used:", int((tok["bbox3d"].reshape(-1, 60, 11)[..., 10] != 1027).any(0).sum()))
    np.savez_compressed(os.path.join(HERE, "scene_reader.npz"), **out)


if __name__ == "__main__":
    main()
