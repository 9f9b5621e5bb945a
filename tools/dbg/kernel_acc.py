import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.gpu_util import bf16_bits, bf16_round, check, fp, from_bits, lib, vp

rng = np.random.default_rng(0)
# linear, bf16 store
R, N, K = 2207, 768, 768
act = bf16_round(rng.standard_normal((R, K), dtype=np.float32)); W = bf16_round((rng.standard_normal((N, K), dtype=np.float32) / np.sqrt(K)).astype(np.float32))
bias = rng.standard_normal((N,), dtype=np.float32) * 0.1
out = np.zeros((R, N), np.uint16)
check(lib().umgen_dbg_linear(1, vp(bf16_bits(act)), vp(bf16_bits(W)), fp(bias), R, N, K, 0, 0, vp(out)))
got = from_bits(out)
ref64 = act.astype(np.float64) @ W.astype(np.float64).T + bias
ref = bf16_round(ref64.astype(np.float32))
print("linear store: exact-match fraction vs bf16(fp64 ref):", float((got == ref).mean()), "max |err|/|ref|max", float(np.abs(got - ref64).max() / np.abs(ref64).max()))
# resid fp32
x0 = rng.standard_normal((R, N), dtype=np.float32); o2 = x0.copy()
check(lib().umgen_dbg_linear(1, vp(bf16_bits(act)), vp(bf16_bits(W)), fp(bias), R, N, K, 0, 1, vp(o2)))
print("linear resid fp32: max abs err", float(np.abs(o2 - (ref64 + x0)).max()))
# spatial attention
F, S, H = 2, 2207, 16
E = H * 48
for scale in (0.5, 1.5):
    q = bf16_round(rng.standard_normal((F, S, E), dtype=np.float32) * scale); k = bf16_round(rng.standard_normal((F, S, E), dtype=np.float32) * scale)
    v = bf16_round(rng.standard_normal((F, S, E), dtype=np.float32))
    qk = np.ascontiguousarray(np.concatenate([q, k], axis=-1))
    y = np.zeros((F, S, E), np.uint16)
    check(lib().umgen_dbg_attn_spatial(1, vp(bf16_bits(qk)), vp(bf16_bits(v)), F, S, H, vp(y)))
    got = from_bits(y)
    D = 48
    qh = torch.from_numpy(q).double().view(F, S, H, D).permute(0, 2, 1, 3); kh = torch.from_numpy(k).double().view(F, S, H, D).permute(0, 2, 1, 3)
    vh = torch.from_numpy(v).double().view(F, S, H, D).permute(0, 2, 1, 3)
    att = (qh @ kh.transpose(-1, -2)) * float(np.float32(1.0 / np.sqrt(D)))
    e = torch.exp(att - att.amax(-1, keepdim=True))
    exact = ((e @ vh) / e.sum(-1, keepdim=True)).permute(0, 2, 1, 3).reshape(F, S, E).numpy()
    er = e.float().bfloat16().double()
    rnd = ((er @ vh) / e.sum(-1, keepdim=True)).permute(0, 2, 1, 3).reshape(F, S, E).numpy()
    for nm, r in (("exact softmax", exact), ("bf16-P model", rnd)):
        print(f"attn scale {scale} vs {nm}: max abs {np.abs(got - r).max():.4e} mean abs {np.abs(got - r).mean():.4e} (|ref| mean {np.abs(r).mean():.3f}); vs bf16(ref): exact-match {float((got == bf16_round(r.astype(np.float32))).mean()):.3f}")
