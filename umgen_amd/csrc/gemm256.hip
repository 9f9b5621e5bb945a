// 256 x 256 x 64 MFMA GEMM of the TAR / ego prefill stacks: the whole-chip launch of gemm256_body.h (the kernel's body, shared with the decode
// engine's background workers), shape support and the launcher.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "bg_queue.h"
#include "gemm256_body.h"

namespace umgen {

namespace {

template <int MODE, typename TT>
__global__ __launch_bounds__(512) void gemm16_256_kernel(GemmArgs a, int nI, int nJ, int splitI, int tpf) {
    gemm16_256_body<MODE, TT, const GemmArgs>(a, nI, nJ, splitI, tpf, blockIdx.x & 7, 8, blockIdx.x >> 3, gridDim.x >> 3, 0, 1 << 24);
}

}  // namespace

size_t gemm256_lds_bytes() { return (size_t)kLds256; }

// measurement builds (UMGEN_G256_STAMPS): ticks of workgroup 9, wave 0 / wave 4: [k-loops, first k-tiles, epilogues, tiles, k-loops in shader clocks] x 2; reset
int gemm256_read_stamps(unsigned long long* out16) {
#ifdef UMGEN_G256_STAMPS
    unsigned long long z[16] = {};
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g256_stamps), sizeof(z)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g256_stamps), z, sizeof(z)) != hipSuccess) return -1;
    return 0;
#else
    (void)out16;
    return -2;
#endif
}

// true when the 256-tile kernel can run this GEMM (otherwise the caller keeps the 128-tile kernels of gemm.hip)
bool gemm256_supported(const GemmArgs& a) {
    // (GEMM_VT arrives here in THIS kernel's operand roles -- P = the weight rows, Q = the frames' token rows, Nj tokens per frame,
    //  batch = frames, strideQ = Nj * ldq, ldo = the padded row length -- see launch_gemm_mfma)
    if (a.mode == GEMM_VT) {
        if (a.batch < 1 || a.strideQ != (long)a.Nj * a.ldq || a.strideP != 0 || a.ldo % 8 != 0 || a.ldo < a.Nj) return false;
    } else if (a.batch != 1 || (a.mode != GEMM_STORE && a.mode != GEMM_RESID && a.mode != GEMM_STORE_F32)) return false;
    if (a.Mi % TM != 0 || a.K % (2 * HK) != 0 || a.K < 2 * HK) return false;
    if ((long)a.Mi * a.ldp >= (1L << 31) || (long)TM * a.ldq >= (1L << 31)) return false;   // 32-bit element offsets (token rows: chunked by the launcher)
    return true;
}

// Per DEVICE, before the first launch there (umgen_create / umgen_vq_create call it behind hipSetDevice): the 160 KB dynamic-LDS
// attribute applies to the current device only, and the persistent grid is one workgroup per CU of THAT device.
static int g_ncu256[64] = {};
hipError_t gemm256_prepare() {
    int dev = 0;
    hipDeviceProp_t prop;
    hipError_t rc0 = hipGetDevice(&dev);
    if (rc0 != hipSuccess) return rc0;
    rc0 = hipGetDeviceProperties(&prop, dev);
    if (rc0 != hipSuccess) return rc0;
    if (dev >= 0 && dev < 64) g_ncu256[dev] = (prop.multiProcessorCount / 8) * 8;
    const void* fns[] = {
        reinterpret_cast<const void*>(gemm16_256_kernel<GEMM_STORE, bf16_t>), reinterpret_cast<const void*>(gemm16_256_kernel<GEMM_RESID, bf16_t>),
        reinterpret_cast<const void*>(gemm16_256_kernel<GEMM_STORE_F32, bf16_t>), reinterpret_cast<const void*>(gemm16_256_kernel<GEMM_STORE, f16_t>),
        reinterpret_cast<const void*>(gemm16_256_kernel<GEMM_RESID, f16_t>), reinterpret_cast<const void*>(gemm16_256_kernel<GEMM_STORE_F32, f16_t>),
        reinterpret_cast<const void*>(gemm16_256_kernel<GEMM_VT, bf16_t>), reinterpret_cast<const void*>(gemm16_256_kernel<GEMM_VT, f16_t>)};
    for (const void* f : fns) {
        hipError_t rc = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kLds256);
        if (rc != hipSuccess) return rc;
    }
    return hipSuccess;
}

// the op of the decode engine's background workers for this GEMM (bg_queue.h): the tile geometry launch_gemm256 would launch -- one launch only (the
// row chunking of very large batches does not occur at one scene)
void record_gemm256(const GemmArgs& a) {
    BgRecorder* rec = g_bg_rec;
    if (!rec) return;
    if (!gemm256_supported(a) || (a.mode != GEMM_STORE && a.mode != GEMM_RESID && a.mode != GEMM_VT)) { rec->failed = "GEMM outside the 256-tile kernel"; return; }
    int nI = a.Mi / TM, nJ, splitI = 1, tpf = 0;
    if (a.mode == GEMM_VT) {
        tpf = (a.Nj + TM - 1) / TM;
        if ((long)a.batch * a.Nj * a.ldq >= (1L << 31)) { rec->failed = "V^T GEMM needs row chunks"; return; }
        nJ = a.batch * tpf;
    } else {
        if ((long)a.Nj > ((((1L << 31) - 1) / a.ldq) / TM) * TM) { rec->failed = "GEMM needs row chunks"; return; }
        nJ = (a.Nj + TM - 1) / TM;
        splitI = (nI % 2 == 0 && (size_t)a.Mi * a.K * 2 > (size_t)(3u << 20)) ? 2 : 1;
    }
    const double tile_flops = 2.0 * TM * TM * (double)a.K;
    BgOp& o = rec->add(BG_GEMM, a.mode, (long)nI * nJ, 64, (unsigned)(tile_flops / 2.8e12 * 1e8) + 300u);
    o.h.i0 = nI; o.h.i1 = nJ; o.h.i2 = splitI; o.h.i3 = tpf;
    o.g = a;
}

template <typename TT>
void launch_gemm256(hipStream_t s, const GemmArgs& a) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    if (!g_ncu256[dev]) (void)gemm256_prepare();   // a caller that skipped the per-device prepare (debug hooks): do it here, for this device
    const int n_cu = g_ncu256[dev] ? g_ncu256[dev] : 256;
    // The kernel addresses its operands with 32-bit element offsets: token rows beyond 2^31 / ldq elements (16 scenes' K = 3072
    // activations, 64 scenes' K = 768 ones) go out as further launches on whole-tile row chunks -- every output element is computed
    // exactly as in one launch (its accumulation order does not depend on the tile's position).
    if (a.mode == GEMM_VT) {      // frames as the chunk unit: (frames x tokens per frame) x ldq elements stay below 2^31
        const int tpf = (a.Nj + TM - 1) / TM;
        int max_frames = (int)std::max<long>(1, ((1L << 31) - 1) / ((long)a.Nj * a.ldq));
        if (const char* dbg = getenv("UMGEN_DEBUG_GEMM256_MAX_ROWS")) max_frames = std::max(1, std::min(max_frames, (int)(atol(dbg) / a.Nj)));   // test hook
        for (int z0 = 0; z0 < a.batch; z0 += max_frames) {
            GemmArgs c = a;
            c.batch = std::min(max_frames, a.batch - z0);
            c.Q = reinterpret_cast<const TT*>(a.Q) + (long)z0 * a.strideQ;
            c.out = reinterpret_cast<TT*>(a.out) + (long)z0 * a.Mi * a.ldo;
            const int nI = c.Mi / TM, nJ = c.batch * tpf;
            hipLaunchKernelGGL((gemm16_256_kernel<GEMM_VT, TT>), dim3(n_cu), dim3(512), kLds256, s, c, nI, nJ, 1, tpf);
        }
        return;
    }
    long max_rows = ((((1L << 31) - 1) / a.ldq) / TM) * TM;
    if (const char* dbg = getenv("UMGEN_DEBUG_GEMM256_MAX_ROWS")) max_rows = std::max<long>(TM, std::min<long>(max_rows, (atol(dbg) / TM) * TM));   // test hook
    if (a.Nj > max_rows) {
        const size_t osz = a.mode == GEMM_STORE ? sizeof(TT) : sizeof(float);
        for (long r0 = 0; r0 < a.Nj; r0 += max_rows) {
            GemmArgs c = a;
            c.Nj = (int)std::min<long>(max_rows, a.Nj - r0);
            c.Q = reinterpret_cast<const TT*>(a.Q) + r0 * a.ldq;
            c.out = reinterpret_cast<unsigned char*>(a.out) + (size_t)r0 * a.ldo * osz;
            launch_gemm256<TT>(s, c);
        }
        return;
    }
    const int nI = a.Mi / TM, nJ = (a.Nj + TM - 1) / TM;
    // feature split over the XCDs only when the weight matrix would not stay in one 4 MB L2 and the feature tiles divide evenly
    const int splitI = (nI % 2 == 0 && (size_t)a.Mi * a.K * 2 > (size_t)(3u << 20)) ? 2 : 1;
    const dim3 grid(n_cu), block(512);
    switch (a.mode) {
        case GEMM_STORE: hipLaunchKernelGGL((gemm16_256_kernel<GEMM_STORE, TT>), grid, block, kLds256, s, a, nI, nJ, splitI, 0); break;
        case GEMM_RESID: hipLaunchKernelGGL((gemm16_256_kernel<GEMM_RESID, TT>), grid, block, kLds256, s, a, nI, nJ, splitI, 0); break;
        default: hipLaunchKernelGGL((gemm16_256_kernel<GEMM_STORE_F32, TT>), grid, block, kLds256, s, a, nI, nJ, splitI, 0); break;
    }
}
template void launch_gemm256<bf16_t>(hipStream_t, const GemmArgs&);
template void launch_gemm256<f16_t>(hipStream_t, const GemmArgs&);

}  // namespace umgen
