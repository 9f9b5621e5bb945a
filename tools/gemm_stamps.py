"""Where a 256 x 256 GEMM tile's time goes (measurement build: tools/build_variant.sh stamps -DUMGEN_G256_STAMPS, UMGEN_LIB_PATH=...):
per shape the microseconds workgroup 9 spends per tile in the k-loop, in the first k-tile (which waits for the previous tile's
stores: CDNA4's vmcnt counts them) and in the epilogue, for the first (wave 0) and second (wave 4) wave group."""
import ctypes as C, sys
sys.path.insert(0, ".")
from umgen_amd import _lib
lib = _lib.load_library()
ms = C.c_float()
st = (C.c_ulonglong * 16)()
rows = [int(x) for x in sys.argv[1:]] or [44140, 353120]
for R in rows:
    for (N, K, mode, name) in [(2304, 768, 0, "qkv temporal"), (3072, 768, 0, "fc (no gelu)"), (3072, 768, 16, "fc + gelu"),
                               (768, 768, 1, "proj resid"), (768, 3072, 1, "proj2 resid"), (4096, 4096, 0, "K=4096 store")]:
        rc = lib.umgen_dbg_gemm_stamps(st)          # reset
        rc = lib.umgen_dbg_gemm_bench(R, N, K, mode | 32, 10, C.byref(ms))
        rs = lib.umgen_dbg_gemm_stamps(st)
        line = f"{name:14s} R={R:6d} N={N:4d} K={K:4d}: {ms.value*1e3:8.1f} us {2.0*R*N*K/ms.value/1e9:7.1f} TFLOP/s"
        if rs == 0:
            for g in range(2):
                n = max(1, st[g * 8 + 3])
                line += f" | group {g}: tiles {n/11:.0f}/launch, per tile k-loop {st[g*8]/n/100:.2f} us (first k-tile {st[g*8+1]/n/100:.2f}, other k-tiles {(st[g*8]-st[g*8+1])/n/100/(K/64-1):.3f} each) epilogue {st[g*8+2]/n/100:.2f} us, shader clock in the k-loops {st[g*8+4]/max(1,st[g*8])*100:.0f} MHz"
        print(line, flush=True)
