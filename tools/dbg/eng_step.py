import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from umgen_amd.config import tiny_config
from umgen_amd.engine import Engine
from umgen_amd.weights import synthetic_state_dict

cfg = tiny_config(n_embd=768, n_head=16, n_oar_layer=5, rule_constrain=False)
sd = synthetic_state_dict(cfg, seed=21)
B, N = 2, 300
rng = np.random.default_rng(0)
X = rng.standard_normal((N, B, 768)).astype(np.float32)
e = Engine(cfg, precision="bf16", max_batch=B, max_cond_frames=4)
e.load_state_dict(sd); e.finalize()
both = [e.dbg_oar_step(X[L], L, True) for L in range(N)]
one = [[e.dbg_oar_step(X[L, b:b + 1], L, True) for L in range(N)] for b in range(B)]
lau = [e.dbg_oar_step(X[L], L, False) for L in range(N)]
for b in range(B):
    d = [float(np.abs(both[L][b] - one[b][L][0]).max()) for L in range(N)]
    bad = [L for L in range(N) if d[L] > 0]
    print("scene", b, "B=2 vs B=1 engine: steps differing:", len(bad), bad[:10], "max", max(d))
d = [float(np.abs(both[L] - lau[L]).max()) for L in range(N)]
print("engine vs launches: max abs diff over steps:", max(d), "at", int(np.argmax(d)), " typical", float(np.median(d)))
e.close()
