"""Golden vectors for the engine's ONE-PASS given-token prefix in the 16-bit modes (engine.hip run_prefix_prefill): the rounding-aware oracle with
``prefix_contract="stack"`` -- positions 0 .. P - 2 of a frame whose map is GIVEN go through the BlockOAR layers with the TAR stacks' rounding points, position
P - 1 as a decode step -- at production width (E = 768, H = 16, tiny depth), one greedy frame per 16-bit type.

    python tests/golden/make_prefix_golden.py        # ~5 CPU minutes

Stored per type: the frame's tokens (the engine is teacher-forced with them), the bbox3d / image logit rows at LOGIT_POS, and -- for the record -- the same rows
of the oracle with ``prefix_contract="decode"`` (the step-by-step replay's arithmetic): their distance is what the one-pass form changes.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle.umgen_oracle import OracleUMGen  # noqa: E402
from tests.golden.make_full_width_golden import LOGIT_POS, config  # noqa: E402
from umgen_amd.config import MOD_ORDER  # noqa: E402
from umgen_amd.synth import synthetic_given_map, synthetic_scene  # noqa: E402
from umgen_amd.weights import synthetic_state_dict  # noqa: E402

WEIGHT_SEED, SCENE_ID = 23, 37


def main():
    torch.set_num_threads(int(os.environ.get("UMGEN_GOLDEN_THREADS", "8")))
    cfg = config("full_width")
    sd = synthetic_state_dict(cfg, seed=WEIGHT_SEED)
    scene = synthetic_scene(SCENE_ID, n_frames=2)
    given = {"map": synthetic_given_map(SCENE_ID, n_frames=1)["map"]}
    for prec in ("bf16", "fp16"):
        o = OracleUMGen(cfg, sd, weight_dtype=f"{prec}_engine", prefix_contract="stack")
        out = o.inference(1, 3, scene, input_cond_frames=2, init_tokens=given, trace=True, seed=0)
        forced = {m: out[m][:, 2] for m in MOD_ORDER}
        blob = {"meta": np.array([WEIGHT_SEED, SCENE_ID])}
        for m in MOD_ORDER:
            blob[f"tok_{m}"] = out[m][0, 2].astype(np.int16)
        for m in ("bbox3d", "image"):
            blob[f"logits_{m}"] = o.trace["logits"][0][m][LOGIT_POS[m]].astype(np.float32)
        od = OracleUMGen(cfg, sd, weight_dtype=f"{prec}_engine", prefix_contract="decode")
        od.inference(1, 3, scene, input_cond_frames=2, init_tokens=given, trace=True, seed=0, forced=forced)
        for m in ("bbox3d", "image"):
            blob[f"decode_contract_logits_{m}"] = od.trace["logits"][0][m][LOGIT_POS[m]].astype(np.float32)
            print(prec, m, "stack vs decode contract: max |dlogit|", float(np.abs(blob[f"logits_{m}"] - blob[f"decode_contract_logits_{m}"]).max()))
        path = os.path.join(HERE, f"full_width_mapgiven_{prec}_engine.npz")
        np.savez_compressed(path, **blob)
        print("wrote", path)


if __name__ == "__main__":
    main()
