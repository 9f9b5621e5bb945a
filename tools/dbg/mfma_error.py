"""How far is the fp32 result of the bf16 MFMA GEMM from the exact (fp64) product of the same bf16 operands, compared with a CPU
fp32 matmul of the same operands?  (residual mode: the kernel's fp32 accumulators are written without any 16-bit rounding.)
Signed mean != 0 would mean a biased (truncating) accumulation inside the matrix core."""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, ".")
from tests.gpu_util import bits16, round16, vp, fp, lib
import torch
for (R, N, K) in [(2048, 768, 768), (2048, 768, 3072)]:
    rng = np.random.default_rng(K)
    act = round16(rng.standard_normal((R, K), dtype=np.float32), 1)
    W = round16((rng.standard_normal((N, K), dtype=np.float32) / np.sqrt(K)).astype(np.float32), 1)
    ref = act.astype(np.float64) @ W.astype(np.float64).T
    cpu = (torch.from_numpy(act) @ torch.from_numpy(W).T).numpy().astype(np.float64)
    for tag, flag in (("128-tile", 32), ("256-tile", 16)):
        out = np.zeros((R, N), np.float32)
        rc = lib().umgen_dbg_linear(1 | flag, vp(bits16(act, 1)), vp(bits16(W, 1)), None, R, N, K, 0, 1, vp(out))
        e = out.astype(np.float64) - ref
        print(f"K={K} {tag}: rc={rc} mfma err mean {e.mean():+.3e} rms {np.sqrt((e**2).mean()):.3e} max {np.abs(e).max():.3e} | rms(ref) {np.sqrt((ref**2).mean()):.3f}")
    e = cpu - ref
    print(f"K={K} cpu fp32 matmul: err mean {e.mean():+.3e} rms {np.sqrt((e**2).mean()):.3e} max {np.abs(e).max():.3e}")
