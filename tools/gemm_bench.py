"""Isolated timing of the bf16 MFMA GEMM at the shapes of the TAR stacks (and a square reference shape)."""
import ctypes as C, sys
sys.path.insert(0, ".")
from umgen_amd import _lib
lib = _lib.load_library()
ms = C.c_float()
for (R, N, K, mode, name) in [(4096, 4096, 4096, 0, "square 4096^3 store"), (44140, 1536, 768, 0, "qk"), (44140, 3072, 768, 0, "fc (no gelu)"), (44140, 3072, 768, 16, "fc + gelu"),
                              (44140, 768, 768, 1, "proj resid"), (44140, 768, 3072, 1, "proj2 resid"), (44140, 2304, 768, 0, "qkv temporal")]:
    rc = lib.umgen_dbg_gemm_bench(R, N, K, mode, 20, C.byref(ms))
    print(f"{name:22s} R={R} N={N} K={K}: rc={rc} {ms.value*1e3:8.1f} us  {2.0*R*N*K/ms.value/1e9:7.1f} TFLOP/s", flush=True)
