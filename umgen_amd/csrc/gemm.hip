// GEMM kernels for the TAR / ego prefill stacks (replace F.linear at module.py:184-190, 236-242 on [B*T*S, E] rows).
//   * gemm_bf16_mfma: bf16 operands, fp32 accumulate on the CDNA4 matrix cores (v_mfma_f32_16x16x32_bf16),
//     128x128x64 tiles, 4 waves (2x2) per workgroup, XOR-swizzled LDS read with ds_read_b128, register-prefetched
//     global->LDS staging.  Epilogues fuse bias, exact GELU, the fp32 residual add and the transposed V store.
//   * gemm_valu: exact fp32 FMA chain (parity mode and the one-off GMLP(codebook) tables).
#include <cstdlib>

#include "bg_queue.h"
#include "kernels.h"

namespace umgen {

template <int MODE, typename TO>
__device__ inline void epilogue4(const GemmArgs& a, int z, int i0, int j, const float (&v)[4]) {
    // v[r] = C[i0 + r][j]
    if (j >= a.Nj || i0 >= a.Mi) return;
    if (MODE == GEMM_VT) {
        const float b = a.bias ? a.bias[j] : 0.f;
        TO* dst = reinterpret_cast<TO*>(a.out) + (((long)z * a.H + j / kHeadDim) * kHeadDim + j % kHeadDim) * a.ldo + i0;
        if (i0 + 3 < a.Mi) {
            float o[4] = {v[0] + b, v[1] + b, v[2] + b, v[3] + b};
            store4(dst, o);
        } else {
            for (int r = 0; r < 4 && i0 + r < a.Mi; ++r) dst[r] = Cvt<TO>::from_f(v[r] + b);
        }
        return;
    }
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float t = v[r] + (a.bias ? a.bias[i0 + r] : 0.f);   // Mi (features) is a multiple of 4
        if (MODE == GEMM_STORE && a.gelu) t = gelu_for<TO>(t);
        o[r] = t;
    }
    if (MODE == GEMM_RESID) {
        float* x = reinterpret_cast<float*>(a.out) + (long)z * a.strideO + (long)j * a.ldo + i0;
        float c[4];
        load4(x, c);
#pragma unroll
        for (int r = 0; r < 4; ++r) c[r] += o[r];
        store4(x, c);
    } else if (MODE == GEMM_STORE_F32) {
        store4(reinterpret_cast<float*>(a.out) + (long)z * a.strideO + (long)j * a.ldo + i0, o);
    } else {
        store4(reinterpret_cast<TO*>(a.out) + (long)z * a.strideO + (long)j * a.ldo + i0, o);
    }
}

// ---------------------------------------------------------------------------------------------------------
// MFMA bf16
// ---------------------------------------------------------------------------------------------------------
constexpr int BM = 128, BN = 128, BK = 64;

__device__ inline int swz(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }  // byte offset

template <int MODE, typename TT>
__global__ __launch_bounds__(256) void gemm_bf16_mfma_kernel(GemmArgs a, int nI, int nJ) {
    typedef typename Mma16<TT>::vec vec8;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2 * BM * BK * 2];   // double buffered: [buf][P tile | Q tile]
    const int z = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    // XCD-aware tile map (block b runs on XCD b % 8): the nI feature tiles that share one activation (Q) tile are issued
    // back-to-back on the same XCD, so every activation tile is fetched into exactly one L2.
#ifndef UMGEN_GEMM_XCD
#define UMGEN_GEMM_XCD 1
#endif
    const int b = blockIdx.x;
#if UMGEN_GEMM_XCD
    const int grp = b / (8 * nI), rem = b % (8 * nI);
    const int tj = grp * 8 + (rem & 7), ti = rem >> 3;
#else
    const int tj = b / nI, ti = b % nI;
#endif
    if (tj >= nJ) return;
    const int i_base = ti * BM, j_base = tj * BN;
    const TT* P = reinterpret_cast<const TT*>(a.P) + (long)z * a.strideP;
    const TT* Q = reinterpret_cast<const TT*>(a.Q) + (long)z * a.strideQ;

    // staging assignment: 4 x 16-byte chunks per operand per thread
    const int srow = tid >> 3, schunk = tid & 7;
    const TT* pp[4];
    const TT* qq[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        int r = srow + 32 * it;
        int gi = min(i_base + r, a.Mi - 1), gj = min(j_base + r, a.Nj - 1);
        pp[it] = P + (long)gi * a.ldp + schunk * 8;
        qq[it] = Q + (long)gj * a.ldq + schunk * 8;
    }
    uint4 rp[4], rq[4];
    auto gload = [&](int k0) {
        const bool ok = (k0 + schunk * 8) < a.K;   // K is a multiple of 8; chunks past K contribute zeros
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            rp[it] = ok ? *reinterpret_cast<const uint4*>(pp[it] + k0) : make_uint4(0, 0, 0, 0);
            rq[it] = ok ? *reinterpret_cast<const uint4*>(qq[it] + k0) : make_uint4(0, 0, 0, 0);
        }
    };
    auto lstore = [&](int buf) {
        unsigned char* ldsP = lds[buf];
        unsigned char* ldsQ = lds[buf] + BM * BK * 2;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            int r = srow + 32 * it;
            *reinterpret_cast<uint4*>(ldsP + swz(r, schunk)) = rp[it];
            *reinterpret_cast<uint4*>(ldsQ + swz(r, schunk)) = rq[it];
        }
    };
    f32x4_t acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, g = lane >> 4;
    const int nkt = (a.K + BK - 1) / BK;
    gload(0);
    lstore(0);
    if (nkt > 1) gload(BK);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const unsigned char* ldsP = lds[kt & 1];
        const unsigned char* ldsQ = lds[kt & 1] + BM * BK * 2;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            vec8 af[4], bfr[4];
            const int c = kk * 4 + g;
#pragma unroll
            for (int m = 0; m < 4; ++m) af[m] = *reinterpret_cast<const vec8*>(ldsP + swz(wi * 64 + m * 16 + frow, c));
#pragma unroll
            for (int n = 0; n < 4; ++n) bfr[n] = *reinterpret_cast<const vec8*>(ldsQ + swz(wj * 64 + n * 16 + frow, c));
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = Mma16<TT>::mfma(af[m], bfr[n], acc[m][n]);
        }
        // tile kt+1 (already in registers) goes to the other buffer (last read in iteration kt-1), then tile kt+2 is requested
        if (kt + 1 < nkt) {
            lstore((kt + 1) & 1);
            if (kt + 2 < nkt) gload((kt + 2) * BK);
        }
        __syncthreads();
    }
    // C/D layout of the 16x16 MFMA: col = lane&15 (j), row = 4*(lane>>4) + reg (i)
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int i0 = i_base + wi * 64 + m * 16 + 4 * g;
            const int j = j_base + wj * 64 + n * 16 + frow;
            float v[4] = {acc[m][n][0], acc[m][n][1], acc[m][n][2], acc[m][n][3]};
            epilogue4<MODE, TT>(a, z, i0, j, v);
        }
}

// bf16 epilogue of the 8-wave (2 x 4, 64 x 32 per wave) kernels through LDS: every wave stores whole 256-byte token rows (16 B per
// lane, 16 lanes per row).  Writing from the MFMA layout directly is 8 B per lane in 32-byte pieces: measured 143 of the fc GEMM's
// 374 us.  `stg` is a 32 KB slab buffer nobody reads any more ([128 tokens][128 features] bf16); 8-byte granule p of token t sits at
// p ^ (t & 15): conflict-free for the MFMA-layout writes and for the row reads.  The caller guarantees a barrier before the buffer
// is reused.
template <typename TT>
__device__ __forceinline__ void store_tile_rows_bf16(const GemmArgs& a, int z, int ti, int tj, const f32x4_t (&acc)[4][2], unsigned char* stg,
                                                     int tid, int wi, int wj, int frow, int g) {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // every wave is done reading that buffer
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int i0 = ti * BM + wi * 64 + m * 16 + 4 * g;
        float bb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bb[r] = a.bias ? a.bias[i0 + r] : 0.f;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t = acc[m][n][r] + bb[r];
                o[r] = a.gelu ? gelu_fast(t) : t;
            }
            const int tl = wj * 32 + n * 16 + frow;
            const int pg = (wi * 16 + m * 4 + g) ^ frow;
            store4(reinterpret_cast<TT*>(stg + tl * 256 + pg * 8), o);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    TT* out = reinterpret_cast<TT*>(a.out) + (long)z * a.strideO + (long)ti * BM;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int idx = tid + 512 * it;
        const int tl = idx >> 4, q = idx & 15;
        const int pos = ((2 * q) ^ (tl & 15)) & ~1;
        uint4 v = *reinterpret_cast<const uint4*>(stg + tl * 256 + pos * 8);
        if (tl & 1) v = make_uint4(v.z, v.w, v.x, v.y);
        const int token = tj * BN + tl;
        if (token < a.Nj) {   // written once, read by a later launch: keep the lines out of the way of the L2-resident weights
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4*>(out + (long)token * a.ldo + 8 * q));
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // staging reads done before a later barrier lets the buffer be refilled
}

// fp32 residual epilogue (x += acc + bias) of the same kernels through LDS, two 64-token halves of 32 KB: the read-modify-write
// then touches whole 512-byte token rows (16 B per lane, 32 lanes per row) instead of 64-byte pieces.  16-byte granule p of token
// t sits at p ^ (t & 7).  Same arithmetic as epilogue4<GEMM_RESID>: x + (acc + bias).
__device__ __forceinline__ void rmw_tile_rows_f32(const GemmArgs& a, int z, int ti, int tj, const f32x4_t (&acc)[4][2], unsigned char* stg,
                                                  int tid, int wi, int wj, int frow, int g) {
    float* xbase = reinterpret_cast<float*>(a.out) + (long)z * a.strideO + (long)ti * BM;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();             // slab reads (hf = 0) / the other half's row reads (hf = 1) are done
        if ((wj >> 1) == hf) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int i0 = ti * BM + wi * 64 + m * 16 + 4 * g;
                float bb[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) bb[r] = a.bias ? a.bias[i0 + r] : 0.f;
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int tl = (wj & 1) * 32 + n * 16 + frow;               // token inside the half
                    const int pg = (wi * 16 + m * 4 + g) ^ (tl & 7);
                    *reinterpret_cast<float4*>(stg + tl * 512 + pg * 16) =
                        make_float4(acc[m][n][0] + bb[0], acc[m][n][1] + bb[1], acc[m][n][2] + bb[2], acc[m][n][3] + bb[3]);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = tid + 512 * it;
            const int tl = idx >> 5, q = idx & 31;
            const float4 v = *reinterpret_cast<const float4*>(stg + tl * 512 + ((q ^ (tl & 7)) << 4));
            const int token = tj * BN + hf * 64 + tl;
            if (token < a.Nj) {
                float* x = xbase + (long)token * a.ldo + 4 * q;
                float c[4];
                load4(x, c);
                c[0] += v.x; c[1] += v.y; c[2] += v.z; c[3] += v.w;
                store4(x, c);
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// Same tile and MFMA schedule, but the operand tiles go HBM -> LDS directly (global_load_lds_dwordx4, 1 KB per wave
// instruction, no staging VGPRs, no ds_write pass).  The LDS image is lane-linear per wave instruction, so the XOR swizzle is
// applied to the per-lane SOURCE address (chunk c' of the image holds global chunk c' ^ (row & 7)).  Two LDS buffers: the
// loads of tile kt+1 are in flight while tile kt is multiplied; one barrier per k-tile.  Requires K % 64 == 0.
template <int MODE, typename TT>
__global__ __launch_bounds__(512) void gemm_bf16_glds_kernel(GemmArgs a, int nI, int nJ) {
    typedef typename Mma16<TT>::vec vec8;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2 * BM * BK * 2];
    const int z = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 8 waves: 2 (i) x 4 (j); each wave owns a 64 x 32 sub-tile (4 x 2 MFMA tiles).  Twice the waves of the 4-wave form at the
    // same LDS footprint: 4 waves per SIMD hide the short-K (12 k-tiles) prologue / epilogue bubbles of the TAR shapes.
    const int wi = wave >> 2, wj = wave & 3;
    // XCD-aware 2-D tile map (block b runs on XCD b % 8).  Measured with FETCH_SIZE: when every XCD walks all feature
    // tiles, a 4.7 MB weight matrix does not stay in the 4 MB L2 and is re-streamed for every token tile (7x the
    // algorithmic traffic).  So the 8 XCDs form a 2 (feature halves) x 4 (token quarters) grid: each L2 keeps half of the
    // weight matrix resident, every activation tile is fetched by 2 XCDs, and inside an XCD the feature tiles that share an
    // activation tile are issued back to back.
    const int b = blockIdx.x;
    const int xcd = b & 7, lb = b >> 3;
    const int hI = (nI + 1) >> 1, qJ = (nJ + 3) >> 2;
    const int ti = (xcd & 1) * hI + lb % hI, tj = (xcd >> 1) * qJ + lb / hI;
    if (ti >= nI || tj >= nJ || ti >= ((xcd & 1) + 1) * hI || tj >= ((xcd >> 1) + 1) * qJ) return;
    const int i_base = ti * BM, j_base = tj * BN;
    const TT* P = reinterpret_cast<const TT*>(a.P) + (long)z * a.strideP;
    const TT* Q = reinterpret_cast<const TT*>(a.Q) + (long)z * a.strideQ;
    // wave w fills the 1 KB segments w, w+8 (8 rows each) of both operand images
    const TT* pp[2];
    const TT* qq[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int r = (wave + 8 * it) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (r & 7);
        pp[it] = P + (long)min(i_base + r, a.Mi - 1) * a.ldp + c * 8;
        qq[it] = Q + (long)min(j_base + r, a.Nj - 1) * a.ldq + c * 8;
    }
    auto issue = [&](int buf, int k0) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            unsigned char* dp = lds[buf] + (wave + 8 * it) * 1024;
            __builtin_amdgcn_global_load_lds((const void*)(pp[it] + k0), (__attribute__((address_space(3))) void*)dp, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void*)(qq[it] + k0), (__attribute__((address_space(3))) void*)(dp + BM * BK * 2), 16, 0, 0);
        }
    };
    f32x4_t acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, g = lane >> 4;
    const int nkt = a.K / BK;
    issue(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) issue((kt + 1) & 1, (kt + 1) * BK);
        const unsigned char* ldsP = lds[kt & 1];
        const unsigned char* ldsQ = lds[kt & 1] + BM * BK * 2;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            vec8 af[4], bfr[2];
            const int c = kk * 4 + g;
#pragma unroll
            for (int m = 0; m < 4; ++m) af[m] = *reinterpret_cast<const vec8*>(ldsP + swz(wi * 64 + m * 16 + frow, c));
#pragma unroll
            for (int n = 0; n < 2; ++n) bfr[n] = *reinterpret_cast<const vec8*>(ldsQ + swz(wj * 32 + n * 16 + frow, c));
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = Mma16<TT>::mfma(af[m], bfr[n], acc[m][n]);
        }
        __syncthreads();
    }
    if (MODE == GEMM_STORE && a.Mi % BM == 0) {
        store_tile_rows_bf16<TT>(a, z, ti, tj, acc, lds[0], tid, wi, wj, frow, g);   // (the loop's last barrier has retired every slab read)
        return;
    }
    if (MODE == GEMM_RESID && a.Mi % BM == 0) {
        rmw_tile_rows_f32(a, z, ti, tj, acc, lds[0], tid, wi, wj, frow, g);
        return;
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int i0 = i_base + wi * 64 + m * 16 + 4 * g;
            const int j = j_base + wj * 32 + n * 16 + frow;
            float v[4] = {acc[m][n][0], acc[m][n][1], acc[m][n][2], acc[m][n][3]};
            epilogue4<MODE, TT>(a, z, i0, j, v);
        }
}

// Persistent form of gemm_bf16_glds_kernel for the large token counts of the TAR / ego stacks: 2 workgroups per CU walk the tile
// list (same XCD-aware order: ids b, b + G, b + 2G, ... stay on XCD b % 8), and the first k-slab of a workgroup's NEXT tile is
// requested during the last k-step of the current one, so the epilogue (bias / GELU / residual read-modify-write) overlaps that
// load instead of being followed by a cold HBM round trip.  At K = 768 a tile is only 12 k-steps, and the cold start was as long
// as the whole MFMA loop.
template <int MODE, typename TT>
__global__ __launch_bounds__(512) void gemm_bf16_pers_kernel(GemmArgs a, int nI, int nJ) {
    typedef typename Mma16<TT>::vec vec8;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2 * BM * BK * 2];
    const int z = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 2, wj = wave & 3;
    const int hI = (nI + 1) >> 1, qJ = (nJ + 3) >> 2;
    const int total = 8 * hI * qJ;
    const TT* P = reinterpret_cast<const TT*>(a.P) + (long)z * a.strideP;
    const TT* Q = reinterpret_cast<const TT*>(a.Q) + (long)z * a.strideQ;
    auto decode = [&](int b, int& ti, int& tj) {   // false: id b names no tile (padding of the 2 x 4 XCD partition)
        const int xcd = b & 7, lb = b >> 3;
        ti = (xcd & 1) * hI + lb % hI;
        tj = (xcd >> 1) * qJ + lb / hI;
        return !(ti >= nI || tj >= nJ || ti >= ((xcd & 1) + 1) * hI || tj >= ((xcd >> 1) + 1) * qJ);
    };
    auto next_valid = [&](int b) {
        int ti, tj;
        while (b < total && !decode(b, ti, tj)) b += gridDim.x;
        return b;
    };
    struct Src { const TT* pp[2]; const TT* qq[2]; };
    auto sources = [&](int b) {
        int ti, tj;
        decode(b, ti, tj);
        Src s;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int r = (wave + 8 * it) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ (r & 7);
            s.pp[it] = P + (long)min(ti * BM + r, a.Mi - 1) * a.ldp + c * 8;
            s.qq[it] = Q + (long)min(tj * BN + r, a.Nj - 1) * a.ldq + c * 8;
        }
        return s;
    };
    auto issue = [&](int buf, const Src& s, int k0) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            unsigned char* dp = lds[buf] + (wave + 8 * it) * 1024;
            __builtin_amdgcn_global_load_lds((const void*)(s.pp[it] + k0), (__attribute__((address_space(3))) void*)dp, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void*)(s.qq[it] + k0), (__attribute__((address_space(3))) void*)(dp + BM * BK * 2), 16, 0, 0);
        }
    };
    const int frow = lane & 15, g = lane >> 4;
    const int nkt = a.K / BK;
    int tile = next_valid(blockIdx.x);
    if (tile >= total) return;
    Src cur = sources(tile);
    int gs = 0;                       // global k-slab counter of this workgroup: slab s lives in buffer s & 1
    issue(0, cur, 0);
    while (tile < total) {
        const int nxt = next_valid(tile + gridDim.x);
        const bool has_next = nxt < total;
        Src nx = cur;
        if (has_next) nx = sources(nxt);
        f32x4_t acc[4][2];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        __syncthreads();              // slab gs has landed; every wave is done with the previous tile's LDS reads
        for (int kt = 0; kt < nkt; ++kt) {
            if (kt + 1 < nkt) issue((gs + 1) & 1, cur, (kt + 1) * BK);
            else if (has_next) issue((gs + 1) & 1, nx, 0);
            const unsigned char* ldsP = lds[gs & 1];
            const unsigned char* ldsQ = lds[gs & 1] + BM * BK * 2;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                vec8 af[4], bfr[2];
                const int c = kk * 4 + g;
#pragma unroll
                for (int m = 0; m < 4; ++m) af[m] = *reinterpret_cast<const vec8*>(ldsP + swz(wi * 64 + m * 16 + frow, c));
#pragma unroll
                for (int n = 0; n < 2; ++n) bfr[n] = *reinterpret_cast<const vec8*>(ldsQ + swz(wj * 32 + n * 16 + frow, c));
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = Mma16<TT>::mfma(af[m], bfr[n], acc[m][n]);
            }
            ++gs;
            if (kt + 1 < nkt) __syncthreads();   // (the last k-step's barrier is the one at the top of the next tile)
        }
        int ti, tj;
        decode(tile, ti, tj);
        if (MODE == GEMM_STORE && a.Mi % BM == 0) {
            // the staging tile is the slab buffer the last k-step just finished with (the other one already receives the next tile's
            // first slab)
            store_tile_rows_bf16<TT>(a, z, ti, tj, acc, lds[(gs - 1) & 1], tid, wi, wj, frow, g);
        } else if (MODE == GEMM_RESID && a.Mi % BM == 0) {
            rmw_tile_rows_f32(a, z, ti, tj, acc, lds[(gs - 1) & 1], tid, wi, wj, frow, g);
        } else {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int i0 = ti * BM + wi * 64 + m * 16 + 4 * g;
                    const int j = tj * BN + wj * 32 + n * 16 + frow;
                    float v[4] = {acc[m][n][0], acc[m][n][1], acc[m][n][2], acc[m][n][3]};
                    epilogue4<MODE, TT>(a, z, i0, j, v);
                }
        }
        tile = nxt;
        cur = nx;
    }
}

template <typename TT>
void launch_gemm_mfma(hipStream_t s, const GemmArgs& a) {
    // large launches: the 256 x 256 deep-pipelined kernel (gemm256.hip); it needs >= 1.5 tiles per CU to fill the chip
    // (measured, profiles/r03_gemm_bench_256tile_stagger.txt: at one scene -- 44 140 tokens -- it wins on qkv / fc / the K = 3072
    //  projection (+4 / +10 / +16 %) and loses on the two launches with ~1000 tiles, where 256 workgroups x 1 tile quantise badly;
    //  from two scenes on it wins everywhere, +12-20 %).  That was one launch at a time: with the three TAR stacks on three streams
    //  (engine.hip, UMGEN_CONCURRENT_STACKS) the other stacks' workgroups fill a launch's last partial round, and the threshold that
    //  is best for a whole frame drops to <= 300 tiles (one scene: TAR 233.6 / 230.0 / 229.5 / 227.8 ms per frame at 1100 / 700 / 500 /
    //  300, no further change below; profiles/r03_engine_experiments.txt)
    if (g_bg_rec) {      // the decode engine's background workers run every GEMM of the pass on the 256-tile body (same products, same k order)
        GemmArgs v = a;
        if (a.mode == GEMM_VT) {      // (operand roles as below)
            v.P = a.Q; v.Mi = a.Nj; v.ldp = a.ldq; v.strideP = a.strideQ;
            v.Q = a.P; v.Nj = a.Mi; v.ldq = a.ldp; v.strideQ = a.strideP;
        }
        record_gemm256(v);
        return;
    }
    static const int min_tiles256 = getenv("UMGEN_GEMM256_MIN_TILES") ? atoi(getenv("UMGEN_GEMM256_MIN_TILES")) : 300;
    const long tiles256 = (long)(a.Mi / 256) * ((a.Nj + 255) / 256);
    if (a.tile256 >= 0 && gemm256_supported(a) && (a.tile256 > 0 || tiles256 >= min_tiles256 || (a.K >= 2048 && tiles256 >= min_tiles256 / 3))) {
        launch_gemm256<TT>(s, a);
        return;
    }
    if (a.mode == GEMM_VT && a.tile256 >= 0) {
        // V transposed: the 256-tile kernel takes it with the operand roles the other way round (its P rows are the weight rows, its Q
        // rows the tokens -- the lanes of one accumulator register then hold 16 consecutive tokens of a feature, what a V^T row wants);
        // token tiles per frame, frames as the batch.  Products and k order are those of this file's kernels: bit-identical rows.
        GemmArgs v = a;
        v.P = a.Q; v.Mi = a.Nj; v.ldp = a.ldq; v.strideP = a.strideQ;
        v.Q = a.P; v.Nj = a.Mi; v.ldq = a.ldp; v.strideQ = a.strideP;
        const long tiles = (long)(v.Mi / 256) * ((v.Nj + 255) / 256) * v.batch;
        if (gemm256_supported(v) && (a.tile256 > 0 || tiles >= min_tiles256)) {
            launch_gemm256<TT>(s, v);
            return;
        }
    }
    const int nI = (a.Mi + BM - 1) / BM, nJ = (a.Nj + BN - 1) / BN;
    dim3 grid(((nJ + 7) / 8) * 8 * nI, 1, a.batch), block(256);
#ifndef UMGEN_NO_PERSISTENT_GEMM
    if (a.K % BK == 0 && a.batch == 1 && a.mode != GEMM_VT && (long)nI * nJ >= 1024) {
        static int n_cu = 0;
        if (!n_cu) {
            int dev = 0;
            hipDeviceProp_t prop;
            n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
        }
        const dim3 pg(2 * n_cu, 1, 1), block8(512);   // 2 workgroups (64 KB of LDS each) per CU; 2 * n_cu is a multiple of 8
        switch (a.mode) {
            case GEMM_STORE: hipLaunchKernelGGL((gemm_bf16_pers_kernel<GEMM_STORE, TT>), pg, block8, 0, s, a, nI, nJ); break;
            case GEMM_RESID: hipLaunchKernelGGL((gemm_bf16_pers_kernel<GEMM_RESID, TT>), pg, block8, 0, s, a, nI, nJ); break;
            default: hipLaunchKernelGGL((gemm_bf16_pers_kernel<GEMM_STORE_F32, TT>), pg, block8, 0, s, a, nI, nJ); break;
        }
        return;
    }
#endif
    if (a.K % BK == 0) {
        grid = dim3(8 * ((nI + 1) / 2) * ((nJ + 3) / 4), 1, a.batch);
        const dim3 block8(512);
        switch (a.mode) {
            case GEMM_STORE: hipLaunchKernelGGL((gemm_bf16_glds_kernel<GEMM_STORE, TT>), grid, block8, 0, s, a, nI, nJ); break;
            case GEMM_RESID: hipLaunchKernelGGL((gemm_bf16_glds_kernel<GEMM_RESID, TT>), grid, block8, 0, s, a, nI, nJ); break;
            case GEMM_STORE_F32: hipLaunchKernelGGL((gemm_bf16_glds_kernel<GEMM_STORE_F32, TT>), grid, block8, 0, s, a, nI, nJ); break;
            default: hipLaunchKernelGGL((gemm_bf16_glds_kernel<GEMM_VT, TT>), grid, block8, 0, s, a, nI, nJ); break;
        }
        return;
    }
    switch (a.mode) {
        case GEMM_STORE: hipLaunchKernelGGL((gemm_bf16_mfma_kernel<GEMM_STORE, TT>), grid, block, 0, s, a, nI, nJ); break;
        case GEMM_RESID: hipLaunchKernelGGL((gemm_bf16_mfma_kernel<GEMM_RESID, TT>), grid, block, 0, s, a, nI, nJ); break;
        case GEMM_STORE_F32: hipLaunchKernelGGL((gemm_bf16_mfma_kernel<GEMM_STORE_F32, TT>), grid, block, 0, s, a, nI, nJ); break;
        default: hipLaunchKernelGGL((gemm_bf16_mfma_kernel<GEMM_VT, TT>), grid, block, 0, s, a, nI, nJ); break;
    }
}
template void launch_gemm_mfma<bf16_t>(hipStream_t, const GemmArgs&);
template void launch_gemm_mfma<f16_t>(hipStream_t, const GemmArgs&);

// ---------------------------------------------------------------------------------------------------------
// VALU fp32 (exact): 64x64 tile, 16-deep k-slab, each thread a 4(i) x 4(j) micro-tile; k ascending fmaf chain
// ---------------------------------------------------------------------------------------------------------
template <int MODE, typename TP, typename TQ, typename TO>
__global__ __launch_bounds__(256) void gemm_valu_kernel(GemmArgs a) {
    __shared__ float sP[16][64 + 4];
    __shared__ float sQ[16][64 + 4];
    const int z = blockIdx.z;
    const int tid = threadIdx.x;
    const int i_base = blockIdx.x * 64, j_base = blockIdx.y * 64;
    const TP* P = reinterpret_cast<const TP*>(a.P) + (long)z * a.strideP;
    const TQ* Q = reinterpret_cast<const TQ*>(a.Q) + (long)z * a.strideQ;
    const int ti = tid & 15, tj = tid >> 4;   // thread owns i = i_base + 4*ti + r, j = j_base + 4*tj + c
    float acc[4][4] = {};
    const int lr = tid >> 2, lk = (tid & 3) * 4;   // staging: row lr (0..63), k offset lk (0..12)
    for (int k0 = 0; k0 < a.K; k0 += 16) {
        {
            const int gi = min(i_base + lr, a.Mi - 1), gj = min(j_base + lr, a.Nj - 1);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = k0 + lk + e;
                sP[lk + e][lr] = (k < a.K) ? Cvt<TP>::to_f(P[(long)gi * a.ldp + k]) : 0.f;
                sQ[lk + e][lr] = (k < a.K) ? Cvt<TQ>::to_f(Q[(long)gj * a.ldq + k]) : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float p[4], q[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) p[r] = sP[k][4 * ti + r];
#pragma unroll
            for (int c = 0; c < 4; ++c) q[c] = sQ[k][4 * tj + c];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[r][c] = fmaf(p[r], q[c], acc[r][c]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float v[4] = {acc[0][c], acc[1][c], acc[2][c], acc[3][c]};
        epilogue4<MODE, TO>(a, z, i_base + 4 * ti, j_base + 4 * tj + c, v);
    }
}

// ---------------------------------------------------------------------------------------------------------
// fp32 on the matrix cores (parity mode): v_mfma_f32_32x32x2_f32 is bit-for-bit the k-ascending fmaf chain of gemm_valu_kernel
// (one rounding per product, no wider accumulator: MI355X guide, "FP32-input MFMA"), at the fp32 vector peak from one wave per
// SIMD -- the VALU kernel reaches a seventh of that (21 TFLOP/s in the TAR stacks).  128 x 128 x 32 tiles, 4 waves (2 x 2), each
// wave 2 x 2 MFMA tiles of 32 x 32 (64 accumulator VGPRs); operands staged k-major in LDS ([k][row], so that the A / B fragments --
// lane l: row l % 32, k = l / 32 -- are conflict-free 4-byte reads), next slab prefetched into registers, two LDS buffers.
// Same k order and same epilogue as gemm_valu_kernel => identical bits (tests/test_gpu_kernels.py::test_fp32_mfma_gemm_is_the_fma_chain).
// ---------------------------------------------------------------------------------------------------------
constexpr int FM = 128, FK = 32, FLD = FM + 4;
template <int MODE, typename TO>
__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(GemmArgs a) {
    __shared__ __attribute__((aligned(16))) float sP[2][FK][FLD];
    __shared__ __attribute__((aligned(16))) float sQ[2][FK][FLD];
    const int z = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int i_base = blockIdx.x * FM, j_base = blockIdx.y * FM;
    const float* P = reinterpret_cast<const float*>(a.P) + (long)z * a.strideP;
    const float* Q = reinterpret_cast<const float*>(a.Q) + (long)z * a.strideQ;
    const int lr = tid >> 3, lk = (tid & 7) * 4;     // staging: rows lr + 32 it, k offset lk .. lk + 3
    // 16-byte loads: rows, batch strides and both bases must keep every float4 aligned (the VALU kernel this one replaced had no such requirement)
    const bool k4 = (a.K % 4 == 0) && (a.ldp % 4 == 0) && (a.ldq % 4 == 0) && (a.strideP % 4 == 0) && (a.strideQ % 4 == 0) &&
                    (((uintptr_t)a.P | (uintptr_t)a.Q) % 16 == 0);
    float4 rp[4], rq[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int gi = min(i_base + lr + 32 * it, a.Mi - 1), gj = min(j_base + lr + 32 * it, a.Nj - 1);
            const float* pp = P + (long)gi * a.ldp + k0 + lk;
            const float* qq = Q + (long)gj * a.ldq + k0 + lk;
            if (k4 && k0 + lk + 3 < a.K) {
                rp[it] = *reinterpret_cast<const float4*>(pp);
                rq[it] = *reinterpret_cast<const float4*>(qq);
            } else {
                float tp[4], tq[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool in = k0 + lk + e < a.K;
                    tp[e] = in ? pp[e] : 0.f;
                    tq[e] = in ? qq[e] : 0.f;
                }
                rp[it] = make_float4(tp[0], tp[1], tp[2], tp[3]);
                rq[it] = make_float4(tq[0], tq[1], tq[2], tq[3]);
            }
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int r = lr + 32 * it;
            sP[buf][lk + 0][r] = rp[it].x; sP[buf][lk + 1][r] = rp[it].y; sP[buf][lk + 2][r] = rp[it].z; sP[buf][lk + 3][r] = rp[it].w;
            sQ[buf][lk + 0][r] = rq[it].x; sQ[buf][lk + 1][r] = rq[it].y; sQ[buf][lk + 2][r] = rq[it].z; sQ[buf][lk + 3][r] = rq[it].w;
        }
    };
    f32x16_t acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    fetch(0);
    stash(0);
    __syncthreads();
    const int l32 = lane & 31, lk2 = lane >> 5;
    int buf = 0;
    for (int k0 = 0; k0 < a.K; k0 += FK) {
        const bool more = k0 + FK < a.K;
        if (more) fetch(k0 + FK);
        const int kmax = min(FK, a.K - k0);          // k beyond K is never multiplied (an odd tail's partner is a zero product)
        for (int kk = 0; kk < kmax; kk += 2) {
            const float a0 = sP[buf][kk + lk2][wi * 64 + l32], a1 = sP[buf][kk + lk2][wi * 64 + 32 + l32];
            const float b0 = sQ[buf][kk + lk2][wj * 64 + l32], b1 = sQ[buf][kk + lk2][wj * 64 + 32 + l32];
#if defined(__HIP_DEVICE_COMPILE__)
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
#endif
        }
        if (more) {
            stash(buf ^ 1);       // the other buffer: its last readers passed the barrier at the end of the previous slab
            __syncthreads();
            buf ^= 1;
        }
    }
    // C layout of the 32 x 32 tile: column (j) = lane % 32, row (i) = 8 (reg / 4) + 4 (lane / 32) + reg % 4
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i0 = i_base + wi * 64 + m * 32 + 8 * q + 4 * lk2;
                const int j = j_base + wj * 64 + n * 32 + l32;
                float v[4] = {acc[m][n][4 * q], acc[m][n][4 * q + 1], acc[m][n][4 * q + 2], acc[m][n][4 * q + 3]};
                epilogue4<MODE, TO>(a, z, i0, j, v);
            }
}

static bool use_f32_mfma() {
    static const bool on = !(getenv("UMGEN_FP32_MFMA") && getenv("UMGEN_FP32_MFMA")[0] == '0');   // 0: the VALU FMA-chain kernel (A/B, tests)
    return on;
}

template <>
void launch_gemm_valu<float, float>(hipStream_t s, const GemmArgs& a) {
    if (use_f32_mfma() && a.tile256 >= 0 && a.Mi % 4 == 0) {
        dim3 grid((a.Mi + FM - 1) / FM, (a.Nj + FM - 1) / FM, a.batch), block(256);
        switch (a.mode) {
            case GEMM_STORE: hipLaunchKernelGGL((gemm_f32_mfma_kernel<GEMM_STORE, float>), grid, block, 0, s, a); break;
            case GEMM_RESID: hipLaunchKernelGGL((gemm_f32_mfma_kernel<GEMM_RESID, float>), grid, block, 0, s, a); break;
            case GEMM_STORE_F32: hipLaunchKernelGGL((gemm_f32_mfma_kernel<GEMM_STORE_F32, float>), grid, block, 0, s, a); break;
            default: hipLaunchKernelGGL((gemm_f32_mfma_kernel<GEMM_VT, float>), grid, block, 0, s, a); break;
        }
        return;
    }
    dim3 grid((a.Mi + 63) / 64, (a.Nj + 63) / 64, a.batch), block(256);
    switch (a.mode) {
        case GEMM_STORE: hipLaunchKernelGGL((gemm_valu_kernel<GEMM_STORE, float, float, float>), grid, block, 0, s, a); break;
        case GEMM_RESID: hipLaunchKernelGGL((gemm_valu_kernel<GEMM_RESID, float, float, float>), grid, block, 0, s, a); break;
        case GEMM_STORE_F32: hipLaunchKernelGGL((gemm_valu_kernel<GEMM_STORE_F32, float, float, float>), grid, block, 0, s, a); break;
        default: hipLaunchKernelGGL((gemm_valu_kernel<GEMM_VT, float, float, float>), grid, block, 0, s, a); break;
    }
}

template <typename TP, typename TQ>
void launch_gemm_valu(hipStream_t s, const GemmArgs& a) {
    dim3 grid((a.Mi + 63) / 64, (a.Nj + 63) / 64, a.batch), block(256);
    // output element type of STORE / VT follows the activation operand (Q for STORE, P for VT)
    switch (a.mode) {
        case GEMM_STORE: hipLaunchKernelGGL((gemm_valu_kernel<GEMM_STORE, TP, TQ, TQ>), grid, block, 0, s, a); break;
        case GEMM_RESID: hipLaunchKernelGGL((gemm_valu_kernel<GEMM_RESID, TP, TQ, float>), grid, block, 0, s, a); break;
        case GEMM_STORE_F32: hipLaunchKernelGGL((gemm_valu_kernel<GEMM_STORE_F32, TP, TQ, float>), grid, block, 0, s, a); break;
        default: hipLaunchKernelGGL((gemm_valu_kernel<GEMM_VT, TP, TQ, TP>), grid, block, 0, s, a); break;
    }
}
template void launch_gemm_valu<bf16_t, float>(hipStream_t, const GemmArgs&);   // bf16 weights x fp32 activations (tables)
template void launch_gemm_valu<bf16_t, bf16_t>(hipStream_t, const GemmArgs&);
template void launch_gemm_valu<f16_t, float>(hipStream_t, const GemmArgs&);
template void launch_gemm_valu<f16_t, f16_t>(hipStream_t, const GemmArgs&);

}  // namespace umgen
