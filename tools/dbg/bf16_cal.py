"""Calibration of the bf16 tolerance: engine (bf16 mode) vs the rounding-aware oracle (bf16_engine), teacher forced."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.umgen_oracle import OracleUMGen
from umgen_amd.config import MOD_ORDER, tiny_config
from umgen_amd.engine import Engine
from umgen_amd.synth import synthetic_scene
from umgen_amd.weights import synthetic_state_dict

for name, cfg, seed in (("tiny", tiny_config(rule_constrain=False).greedy(), 7),):
    sd = synthetic_state_dict(cfg, seed=seed)
    scene = synthetic_scene(31, n_frames=2)
    for mode in ("bf16", "bf16_engine"):
        t0 = time.time()
        o = OracleUMGen(cfg, sd, weight_dtype=mode)
        ref = o.inference(1, 3, scene, input_cond_frames=2, trace=True, seed=0)
        forced = {m: ref[m][0, 2] for m in MOD_ORDER}
        e = Engine(cfg, precision="bf16", max_cond_frames=4)
        e.load_state_dict(sd); e.finalize()
        toks, tr = e.frame({m: scene[m][0] for m in MOD_ORDER}, frame_idx=0, trace=True, forced=forced)
        e.close()
        rel = lambda a, b: float(np.sqrt(((a - b) ** 2).mean()) / np.sqrt((b ** 2).mean()))
        print("   relative rms: cond", rel(tr["cond"], o.trace["cond"][0]), "ego", rel(tr["ego_logits"], o.trace["ego_logits"][0]), [rel(tr[f"logits_{m}"], o.trace["logits"][0][m]) for m in ("map", "bbox3d", "image")])
        d = {"cond": float(np.abs(tr["cond"] - o.trace["cond"][0]).max()), "ego": float(np.abs(tr["ego_logits"] - o.trace["ego_logits"][0]).max())}
        for m in ("map", "bbox3d", "image"):
            a, b = tr[f"logits_{m}"], o.trace["logits"][0][m]
            d[m] = float(np.abs(a - b).max())
            d[m + "_argmax_agree"] = float((a.argmax(-1) == b.argmax(-1)).mean())
        print(name, mode, {k: round(v, 5) for k, v in d.items()}, "sampled!=forced", tr["counters"]["sampled_ne_forced"], f"({time.time()-t0:.0f}s)", flush=True)
