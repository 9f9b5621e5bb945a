"""One decode step through the layers (umgen_dbg_oar_step) with decode_chain_kernel against the five launches per layer, same inputs, growing KV length."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from umgen_amd.config import tiny_config
from umgen_amd.engine import Engine
from umgen_amd.weights import synthetic_state_dict

B = int(os.environ.get("B", "5")); NL = int(os.environ.get("LAYERS", "1")); N = int(os.environ.get("STEPS", "6"))
cfg = tiny_config(n_embd=768, n_head=16, n_oar_layer=NL, rule_constrain=False)
sd = synthetic_state_dict(cfg, seed=21)
rng = np.random.default_rng(0)
X = rng.standard_normal((N, B, 768)).astype(np.float32)
eng = {}
for chain in ("0", "1"):
    os.environ.update(UMGEN_DECODE_CHAIN=chain, UMGEN_DECODE_BATCHED="1", UMGEN_DECODE_MS="0")
    e = Engine(cfg, precision="bf16", max_batch=max(B, 2), max_cond_frames=4)
    e.load_state_dict(sd); e.finalize()
    eng[chain] = e
for L in range(N):
    a = eng["0"].dbg_oar_step(X[L], L, 0)
    b = eng["1"].dbg_oar_step(X[L], L, 0)
    d = np.abs(a - b)
    print(f"L={L}: max |diff| {d.max():.3e}, elements differing {(a != b).sum()} of {a.size}, rows differing {np.unique(np.nonzero(a != b)[0]).tolist()[:8]}, nan {np.isnan(b).sum()}")
for e in eng.values():
    e.close()
