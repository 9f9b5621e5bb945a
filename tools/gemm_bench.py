"""Isolated timing of the 16-bit MFMA GEMMs at the shapes of the TAR stacks (and a square reference shape): the 256 x 256 x 64
deep-pipelined kernel (gemm256.hip, mode bit 5) against the 128 x 128 kernels of round 2 (gemm.hip, mode bit 6), random operands.
R = 44140 tokens is one scene's 20-frame window; 353120 = 8 scenes per GPU."""
import ctypes as C, sys
sys.path.insert(0, ".")
from umgen_amd import _lib
lib = _lib.load_library()
ms = C.c_float()
rows = [int(x) for x in sys.argv[1:]] or [44140, 353120]
for R in rows:
    for (N, K, mode, name) in [(1536, 768, 0, "qk"), (2304, 768, 0, "qkv temporal"), (3072, 768, 0, "fc (no gelu)"), (3072, 768, 16, "fc + gelu"),
                               (768, 768, 1, "proj resid"), (768, 3072, 1, "proj2 resid")]:
        line = f"{name:14s} R={R:6d} N={N:4d} K={K:4d}:"
        for tag, bit in (("256-tile", 32), ("128-tile", 64)):
            rc = lib.umgen_dbg_gemm_bench(R, N, K, mode | bit, 10, C.byref(ms))
            line += f"  {tag} {ms.value*1e3:8.1f} us {2.0*R*N*K/ms.value/1e9:7.1f} TFLOP/s" if rc == 0 else f"  {tag} rc={rc}"
        print(line, flush=True)
for F in (20, 160):      # the spatial attention's V transposed per frame (GEMM_VT): one / eight scenes' frames of 2207 tokens
    line = f"v transposed   F={F:3d} S=2207 N= 768 K= 768:"
    for tag, bit in (("256-tile", 32), ("128-tile", 64)):
        rc = lib.umgen_dbg_gemm_vt_bench(F, 2207, 768, 768, bit, 10, C.byref(ms))
        line += f"  {tag} {ms.value*1e3:8.1f} us {2.0*F*2207*768*768/ms.value/1e9:7.1f} TFLOP/s" if rc == 0 else f"  {tag} rc={rc}"
    print(line, flush=True)
for tag, bit in (("256-tile", 32), ("128-tile", 64)):
    rc = lib.umgen_dbg_gemm_bench(4096, 4096, 4096, bit, 20, C.byref(ms))
    print(f"square 4096^3 store {tag}: rc={rc} {ms.value*1e3:8.1f} us  {2.0*4096**3/ms.value/1e9:7.1f} TFLOP/s", flush=True)
