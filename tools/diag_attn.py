import numpy as np, sys
sys.path.insert(0,'.')
from tests.gpu_util import *
from tests.test_gpu_kernels import ref_attention
for (F,S,H) in [(1,64,1),(1,128,1),(1,70,1),(1,2207,2)]:
    E=H*48
    rng=np.random.default_rng(0)
    q=bf16_round(rng.standard_normal((F,S,E),dtype=np.float32)*1.5); k=bf16_round(rng.standard_normal((F,S,E),dtype=np.float32)*1.5); v=bf16_round(rng.standard_normal((F,S,E),dtype=np.float32))
    qk=np.ascontiguousarray(np.concatenate([q,k],-1)); y=np.zeros((F,S,E),np.uint16)
    check(lib().umgen_dbg_attn_spatial(1, vp(bf16_bits(qk)), vp(bf16_bits(v)), F,S,H, vp(y)))
    got=from_bits(y); ref=ref_attention(q,k,v,H,False)
    err=np.abs(got-ref)
    print(F,S,H,"max err",err.max(), "bad frac", (err>2e-2).mean())
    bad_q = np.where(err.max(axis=(0,2))>2e-2)[0]
    print(" bad queries:", bad_q[:20], "...", bad_q[-5:] if len(bad_q) else "", len(bad_q))
    bad_d = np.where(err.max(axis=(0,1))>2e-2)[0]
    print(" bad d:", bad_d)
