import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from umgen_amd.config import tiny_config
from umgen_amd.engine import Engine
from umgen_amd.weights import synthetic_state_dict
cfg = tiny_config(n_embd=768, n_head=16, n_oar_layer=5, rule_constrain=False)
sd = synthetic_state_dict(cfg, seed=21)
N = 40
rng = np.random.default_rng(0)
X = rng.standard_normal((N, 1, 768)).astype(np.float32)
e = Engine(cfg, precision="bf16", max_batch=1, max_cond_frames=4)
e.load_state_dict(sd); e.finalize()
a = [e.dbg_oar_step(X[L], L, True, unmasked=True) for L in range(N)]
b = [e.dbg_oar_step(X[L], L, True, unmasked=False) for L in range(N)]
c = [e.dbg_oar_step(X[L], L, False, unmasked=True) for L in range(N)]
d = [float(np.abs(a[L] - b[L]).max()) for L in range(N)]
print("NG=8 vs NG=6:", [f"{v:.1e}" for v in d[:8]], "n elems differing at step 0:", int((a[0] != b[0]).sum()))
d = [float(np.abs(a[L] - c[L]).max()) for L in range(N)]
print("NG=8 vs launches:", [f"{v:.1e}" for v in d[:8]], "n elems differing at step 0:", int((a[0] != c[0]).sum()))
d = [float(np.abs(b[L] - c[L]).max()) for L in range(N)]
print("NG=6 vs launches:", [f"{v:.1e}" for v in d[:8]], "n elems differing at step 0:", int((b[0] != c[0]).sum()))
e.close()
