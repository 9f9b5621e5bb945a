"""Accumulation-order ensemble of the rounding-aware oracle: the noise floor the 16-bit engine modes are bounded by.

    python tests/golden/make_ensemble.py [tiny] [full_width] [deep] [--modes bf16_engine fp16_engine] [--members 8]
        ->  tests/golden/ensemble_<width>_<mode>.npz

The rounding-aware oracle (oracle/umgen_oracle.py, weight_dtype="bf16_engine" / "fp16_engine") rounds to 16 bits wherever the
engine stores 16 bits, but it cannot follow the ORDER of the engine's fp32 additions (MFMA tiles, split softmax, DPP sums).
Every storage point turns that order noise into 1-ulp flips of the stored value, so two mathematically identical evaluations
differ by far more than fp32 epsilon.  Instead of measuring the engine and then loosening the test bars until it passes, the
noise floor is measured on the ORACLE ITSELF: the same teacher-forced frame is evaluated with the natural summation order and
with `members` seeded random orders (OracleUMGen(perm_seed=..., mfma_noise=True): every F.linear sums K in a random order and carries
the MEASURED fp32 accumulation noise of the matrix cores -- relative rms 1.56e-7 sqrt(K / 768), profiles/r03_mfma_error.txt --,
every attention sums its keys in a random order).  Recorded per quantity (conditioning rows, ego logits, OAR logit rows):
    center  = mean over the members                      (the reference values of the -m gpu tests)
    spread  = max over members and elements of |member - center|
The -m gpu tests assert  max |engine - center| <= 2 x spread  (the engine is one more summation order; the factor 2 covers what an
ensemble of 9 under-samples), and that every arg-max flip of the engine is a position where the members themselves are within
2 x spread of a tie.  Teacher forcing uses the fp32 oracle's greedy tokens (the committed goldens), so all members see the same
inputs at every step.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.umgen_oracle import OracleUMGen  # noqa: E402
from umgen_amd.config import MOD_ORDER, tiny_config  # noqa: E402
from umgen_amd.synth import synthetic_scene  # noqa: E402
from umgen_amd.weights import synthetic_state_dict  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def case(width):
    """(cfg, state dict, scene, cond_frames, input_cond_frames, forced tokens, cond rows, logit positions)"""
    if width == "tiny":
        from tests.golden.make_oracle_cases import COND_ROWS, LOGIT_POS
        g = np.load(os.path.join(GOLD, "tiny_video_greedy.npz"))
        ws, sid, cf, icf, nf, ctl = [int(x) for x in g["meta"]]
        cfg = tiny_config().greedy()
        forced = {m: g[f"out_{m}"][0, icf].astype(np.int64)[None] for m in MOD_ORDER}
        return cfg, synthetic_state_dict(cfg, seed=ws), synthetic_scene(sid, n_frames=icf), cf, icf, forced, COND_ROWS, LOGIT_POS
    from tests.golden.make_full_width_golden import COND_ROWS, LOGIT_POS, SCENE_ID, WEIGHT_SEED, config
    g = np.load(os.path.join(GOLD, f"{width}_fp32.npz"))
    cfg = config(width)
    forced = {m: g[f"tok_{m}"].astype(np.int64)[None] for m in MOD_ORDER}
    return cfg, synthetic_state_dict(cfg, seed=WEIGHT_SEED), synthetic_scene(SCENE_ID, n_frames=2), 3, 2, forced, COND_ROWS, LOGIT_POS


def main(width, mode, members):
    cfg, sd, scene, cf, icf, forced, cond_rows, logit_pos = case(width)
    runs = []
    for i in range(members + 1):
        t0 = time.time()
        o = OracleUMGen(cfg, sd, weight_dtype=mode, perm_seed=None if i == 0 else 1000 + i, mfma_noise=True)
        o.inference(1, cf, scene, input_cond_frames=icf, trace=True, seed=0, forced=forced)
        tr = o.trace
        runs.append({"cond": tr["cond"][0].astype(np.float64), "ego": tr["ego_logits"][0].astype(np.float64),
                     **{m: tr["logits"][0][m].astype(np.float64) for m in logit_pos}})
        print(f"{width} {mode} member {i}: {time.time() - t0:.0f} s", flush=True)
    out = {"members": np.int32(members + 1)}
    for m in MOD_ORDER:
        out[f"tok_{m}"] = forced[m][0].astype(np.int16)
    for q in ["cond", "ego"] + list(logit_pos):
        st = np.stack([r[q] for r in runs])
        center = st.mean(0)
        dev = np.abs(st - center).max(0)          # per element: the largest member deviation
        out[f"{q}_spread"] = np.float32(dev.max())
        out[f"{q}_rms"] = np.float32(np.sqrt((center ** 2).mean()))
        if q == "cond":
            out["cond_center"] = center[cond_rows].astype(np.float32)
        elif q == "ego":
            out["ego_center"] = center.astype(np.float32)
        else:
            out[f"{q}_center"] = center[logit_pos[q]].astype(np.float32)
            srt = np.sort(center, axis=-1)
            out[f"{q}_argmax"] = center.argmax(-1).astype(np.int16)
            out[f"{q}_gap"] = (srt[:, -1] - srt[:, -2]).astype(np.float32)          # top-2 gap of the ensemble centre, every position
    path = os.path.join(GOLD, f"ensemble_{width}_{mode}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), {k: float(v) for k, v in out.items() if k.endswith("_spread")})


if __name__ == "__main__":
    import torch
    torch.set_num_threads(int(os.environ.get("UMGEN_GOLDEN_THREADS", "8")))
    ap = argparse.ArgumentParser()
    ap.add_argument("widths", nargs="*", default=["tiny", "full_width"])
    ap.add_argument("--modes", nargs="*", default=["bf16_engine", "fp16_engine"])
    ap.add_argument("--members", type=int, default=8)
    a = ap.parse_args()
    for w in a.widths:
        for md in a.modes:
            main(w, md, a.members)
