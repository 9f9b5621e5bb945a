"""Golden vectors at the production WIDTH (E=768, H=16, all vocabularies at production size; one block per stack and two BlockOAR
layers, or -- "deep" -- two blocks per stack and ten BlockOAR layers) from the CPU
oracle, so that the -m gpu tests do not spend minutes of GPU-box time re-running the oracle:

    python tests/golden/make_full_width_golden.py [full_width] [wide2x] [deep] [--mode=fp32]   ->  tests/golden/{full_width,wide2x,deep}_{fp32,bf16_engine}.npz

fp32: the oracle in the reference's semantics (itself pinned on the reference goldens, tests/test_oracle.py);
bf16_engine: the rounding-aware restatement of the engine's production mode (oracle/umgen_oracle.py header).
Both are teacher-forced with the fp32 oracle's greedy tokens.  Stored: the forced tokens, conditioning rows / ego logits /
logit rows at fixed positions, and for EVERY sampled position the arg-max and the top-2 logit gap (to classify arg-max flips
of the engine as near-ties)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.umgen_oracle import OracleUMGen  # noqa: E402
from umgen_amd.config import MOD_ORDER, tiny_config  # noqa: E402
from umgen_amd.synth import synthetic_scene  # noqa: E402
from umgen_amd.weights import synthetic_state_dict  # noqa: E402

WEIGHT_SEED, SCENE_ID = 21, 31
COND_ROWS = [0, 1, 4, 5, 6, 7, 100, 500, 1029, 1030, 1031, 1032, 1042, 1043, 1400, 1692, 1693, 1694, 2000, 2205, 2206]
LOGIT_POS = {"map": [0, 1, 2, 511, 512, 1022, 1023], "bbox3d": [0, 9, 10, 11, 330, 658, 659], "image": [0, 1, 255, 510, 511]}


# "deep": production width with SEVERAL layers per stack -- 2 blocks in every TAR stack and in the ego decoder, 10 BlockOAR layers: the
# decode engine's layer -> XCD-group rotation wraps (layers 8, 9 run on groups 0, 1 again), the x vector crosses the fabric nine times,
# and the TAR stacks chain a second block behind the first (the production decode kernel met the oracle
# only at the depth of the other goldens)
DEEP = dict(n_ego_tar_layer=2, n_ego_ca_layer=2, n_map_tar_layer=2, n_box_tar_layer=2, n_tar_layer=2, n_oar_layer=10)


def config(width: str = "full_width", **over):
    """full_width: UMGen_Large's E=768 / H=16; wide2x: BASELINE.json config #5's E=1536 / H=32 -- tiny_config's depth either way (one
    block per stack, two BlockOAR layers); deep: E=768 / H=16 with DEEP's layer counts."""
    E, H = (1536, 32) if width == "wide2x" else (768, 16)
    if width == "deep":
        over = {**DEEP, **over}
    return tiny_config(n_embd=E, n_head=H, rule_constrain=False, **over).greedy()


def main(width: str = "full_width", modes=("fp32", "bf16_engine")):
    cfg = config(width)
    sd = synthetic_state_dict(cfg, seed=WEIGHT_SEED)
    scene = synthetic_scene(SCENE_ID, n_frames=2)
    forced = None
    if "fp32" not in modes:      # teacher forcing always uses the committed fp32 oracle's greedy tokens
        g = np.load(os.path.join(ROOT, "tests", "golden", f"{width}_fp32.npz"))
        forced = {m: g[f"tok_{m}"].astype(np.int64)[None] for m in MOD_ORDER}
    for mode in modes:
        o = OracleUMGen(cfg, sd, weight_dtype=mode)
        ref = o.inference(1, 3, scene, input_cond_frames=2, trace=True, seed=0, forced=forced)
        if forced is None:
            forced = {m: ref[m][:, 2] for m in MOD_ORDER}
        out = {"meta": np.array([WEIGHT_SEED, SCENE_ID]), "cond_rows": o.trace["cond"][0][COND_ROWS].astype(np.float32),
               "ego_logits": o.trace["ego_logits"][0].astype(np.float32)}
        for m in MOD_ORDER:
            out[f"tok_{m}"] = forced[m][0].astype(np.int16)
        for m, pos in LOGIT_POS.items():
            lg = o.trace["logits"][0][m]
            out[f"logits_{m}"] = lg[pos].astype(np.float32)
            srt = np.sort(lg, axis=-1)
            out[f"argmax_{m}"] = lg.argmax(-1).astype(np.int16)
            out[f"gap_{m}"] = (srt[:, -1] - srt[:, -2]).astype(np.float32)
            out[f"rms_{m}"] = np.sqrt((lg.astype(np.float64) ** 2).mean(-1)).astype(np.float32)
        path = os.path.join(ROOT, "tests", "golden", f"{width}_{mode}.npz")
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    only = [a[7:] for a in sys.argv[1:] if a.startswith("--mode=")]
    import torch
    torch.set_num_threads(int(os.environ.get("UMGEN_GOLDEN_THREADS", "8")))
    for w in (args or ["full_width", "wide2x"]):
        main(w, tuple(only) if only else ("fp32", "bf16_engine"))
