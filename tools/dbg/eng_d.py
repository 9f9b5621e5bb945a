import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from umgen_amd.config import MOD_ORDER, tiny_config
from umgen_amd.engine import Engine
from umgen_amd.synth import synthetic_scene
from umgen_amd.weights import synthetic_state_dict

cfg = tiny_config(n_embd=768, n_head=16, n_oar_layer=5, rule_constrain=False)
sd = synthetic_state_dict(cfg, seed=21)
scene = synthetic_scene(46, n_frames=2)
window = {m: scene[m][0] for m in MOD_ORDER}
res = {}
for D in (8, 4, 2, 1):
    os.environ["UMGEN_DEBUG_ENGINE_D"] = str(D)
    e = Engine(cfg, precision="bf16", max_batch=1, max_cond_frames=4, use_graphs=False)
    e.load_state_dict(sd); e.finalize()
    toks, tr = e.frame(window, frame_idx=0, seed=106, trace=True)
    res[D] = (toks, tr)
    e.close()
t8, r8 = res[8]
for D in (4, 2, 1):
    t, r = res[D]
    for m in ("map", "bbox3d", "image"):
        d = np.abs(r[f"logits_{m}"] - r8[f"logits_{m}"]).max(axis=1)
        bad = np.argwhere(d > 0).ravel()
        print("D", D, m, "rows with any logit difference:", len(bad), "first:", bad[:5].tolist(), "max:", float(d.max()),
              "tok mismatches:", int((t[m] != t8[m]).sum()))
