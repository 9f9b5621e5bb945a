// Bodies of the spatial (S x S, matrix cores) and temporal (causal over the history slots) attention kernels of the TAR / ego stacks as device
// functions: attn.hip launches them as kernels, the decode engine's background workers (bg_worker.h) run the same code on their share of the blocks.
#pragma once
#include <type_traits>

#include "kernels.h"

#ifndef UMGEN_ATTN_QT
#define UMGEN_ATTN_QT 2   // measured: 2 query tiles per wave (2 waves/SIMD) beats 4 (1 wave/SIMD)
#endif

namespace umgen {

constexpr float kScale = 0.14433756729740643f;          // float32(1/sqrt(48)), module.py:196-198
constexpr float kLog2e = 1.4426950408889634f;

// ---------------------------------------------------------------------------------------------------------
// spatial (non-causal, S x S per frame and head) -- bf16 MFMA, "swapped" products so softmax rows are lane-local:
//   S^T[key][query] = K[key][d] . Q^T[d][query]      (v_mfma_f32_16x16x32_bf16 for d 0..31 + 16x16x16 for d 32..47)
//   O^T[d][query]  += Vt[d][key] . P^T[key][query]   (16x16x32; the k-slot -> key map is chosen so that each lane's own
//                                                     P registers are exactly its B-operand elements: no cross-lane traffic)
// Q, K are row-major [token][2E]; V is stored transposed per (frame, head) by the QKV GEMM epilogue (GEMM_VT).
// A workgroup = 4 waves x (QT*16) queries of one (frame, head).  The 64-key K tile [64][48] and Vt tile [48][64] are staged
// once per workgroup into double-buffered LDS (16-byte global loads prefetched into registers one tile ahead; images laid out
// for conflict-free ds_read_b128 / ds_read_b64 fragment reads, see below) and shared by the 4 waves.
// Blocks of one (frame, head) are mapped to the same XCD (block b runs on XCD b % 8) so its K/V stay in one L2.
// ---------------------------------------------------------------------------------------------------------
// LDS images (SQ_LDS_BANK_CONFLICT was 44 % of the LDS cycles with plain 112 B / 144 B rows and 8-byte fragment reads):
//   K row (128 B) = eight 16 B chunks: d 0..31 in chunks 0..3, d 32..47 in chunks 4, 5, zeros (written once) in chunks 6, 7 -- the
//     half-filled MFMA takes d 32..47 in k-slot groups 0, 1 and zeros in groups 2, 3; both A operands are one ds_read_b128.  Chunk c of
//     row r is stored at c ^ 2*((r >> 1) & 3): a ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...
//     (MI355X_MICROARCH.md, LDS), and this XOR puts each group's 16 lanes on 16 distinct 16-byte slots of the 256-byte bank row
//     (round 1's 144 B rows were 2-way: 27 M of 86 M LDS cycles per launch, profiles/r02_attn_variants.txt).
//   Vt row (128 B): sixteen 8 B key groups, group s stored at s ^ (row & 15): the 16 rows of a b64 phase hit 16 distinct bank pairs.
constexpr int kKStride = 128;
constexpr int kVStride = 128;
constexpr int kTileBytes = 64 * kKStride + 48 * kVStride;   // 14336

// CAUSAL (the OAR prefix pass, engine.hip run_prefix_prefill; module.py:402-416 with flash-attn's causal mask at q_len == k_len): query i sees
// keys 0 .. i.  The key loop stops behind the workgroup's last query; inside it keys past a query are masked (the first tile always holds
// key 0, so a query's running maximum is finite before the first fully masked tile: exp2(-inf - m) = 0, alpha = 1).
// The kernel's body as a device function (the decode engine's background workers, bg_worker.h, run two 256-thread blocks of it per 512-thread
// workgroup: WORKER = true -- block `b` of the virtual grid, thread `tid` of 256, the half's 28 KB at `lds_off` of the dynamic LDS; both halves
// pass the same barriers, a half without a block (`valid` false) computes a clamped block and stores nothing).
template <int QT, typename TT, bool CAUSAL, bool WORKER>
__device__ __forceinline__ void attn_spatial_mfma_body(const TT* __restrict__ qk, const TT* __restrict__ vt, TT* __restrict__ y, int S, int S_pad, int H,
                                                       int nq, int npairs, int b, int tid, int lds_off, bool valid) {
    typedef typename Mma16<TT>::vec vec8;
    typedef typename Mma16<TT>::elem elem_t;
    unsigned char* lds;
    if constexpr (WORKER) {
        extern __shared__ __attribute__((aligned(1024))) unsigned char dyn_lds[];
        lds = dyn_lds + lds_off;
    } else {
        __shared__ __attribute__((aligned(16))) unsigned char own_lds[2 * kTileBytes];
        lds = own_lds;
    }
    const int E = H * kHeadDim;
    // XCD-aware decode of the flat block id: 8 consecutive (frame, head) pairs form a group; inside it b = q*8 + j
    const int group = b / (8 * nq), rem = b % (8 * nq);
    const int pair_raw = group * 8 + (rem & 7), qb = rem >> 3;
    if (!WORKER && pair_raw >= npairs) return;
    const bool live = valid && pair_raw < npairs;      // (a worker half without a block of its own runs pair 0 and stores nothing)
    const int pair = live ? pair_raw : 0;
    const int f = pair / H, h = pair % H;
    const int lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, g = lane >> 4;
    const int q0 = (qb * 4 + wave) * (QT * 16);
    const long ld = 2L * E;
    const TT* qbase = qk + (long)f * S * ld + h * kHeadDim;
    const TT* kbase = qbase + E;
    const TT* vbase = vt + ((long)f * H + h) * kHeadDim * S_pad;

    // staging assignment: 768 16-byte chunks per tile (K: 64 rows x 6, Vt: 48 rows x 8), 3 per thread.  Chunks 0..383 are K,
    // 384..767 are Vt, so thread tid's chunks tid / tid+256 / tid+512 are K / (K if tid < 128 else Vt) / Vt: the three staging
    // registers are named scalars (an indexed array ended up in scratch memory).
    const bool mid_is_k = tid < 128;
    const int r0 = tid / 6, p0 = tid % 6;                                   // chunk tid: K row r0, piece p0
    const int c1 = tid + 256;
    const int r1 = mid_is_k ? c1 / 6 : (c1 - 384) >> 3, p1 = mid_is_k ? c1 % 6 : (c1 - 384) & 7;
    const int c2 = tid + 512 - 384, r2 = c2 >> 3, p2 = c2 & 7;             // chunk tid+512: Vt row r2, piece p2
    const TT* g0 = kbase + p0 * 8;
    const TT* g1 = mid_is_k ? kbase + p1 * 8 : vbase + (long)r1 * S_pad + p1 * 8;
    const TT* g2 = vbase + (long)r2 * S_pad + p2 * 8;
    // LDS byte offsets: K piece p -> chunk p ^ 2*((row >> 1) & 3) of the row;
    // Vt piece p (key groups 2p, 2p+1) -> the aligned 16 B pair (2p ^ row) & ~1, halves swapped when the row is odd
    auto koff = [&](int r, int pp) { return r * kKStride + ((pp ^ (2 * ((r >> 1) & 3))) << 4); };
    auto voff = [&](int r, int pp) { return 64 * kKStride + r * kVStride + (((2 * pp) ^ (r & 15)) & ~1) * 8; };
    const int l0 = koff(r0, p0);
    const int l1 = mid_is_k ? koff(r1, p1) : voff(r1, p1);
    const int l2 = voff(r2, p2);
    const bool swap1 = !mid_is_k && (r1 & 1), swap2 = r2 & 1;
    uint4 sg0, sg1, sg2;
    auto gload = [&](int k0) {
        sg0 = *reinterpret_cast<const uint4*>(g0 + (long)min(k0 + r0, S - 1) * ld);
        sg1 = *reinterpret_cast<const uint4*>(mid_is_k ? g1 + (long)min(k0 + r1, S - 1) * ld : g1 + k0);
        sg2 = *reinterpret_cast<const uint4*>(g2 + k0);
    };
    auto vstore = [](unsigned char* dst, uint4 v, bool sw) {
        *reinterpret_cast<uint4*>(dst) = sw ? make_uint4(v.z, v.w, v.x, v.y) : v;
    };
    auto lstore = [&](int buf) {
        unsigned char* base = lds + buf * kTileBytes;
        *reinterpret_cast<uint4*>(base + l0) = sg0;
        vstore(base + l1, sg1, swap1);   // (swap1 is false for the K chunks)
        vstore(base + l2, sg2, swap2);
    };
    {   // the zero chunks 6, 7 of every K row (both buffers) are written once
        const int buf = tid >> 7, r = (tid >> 1) & 63, sl = 6 + (tid & 1);
        *reinterpret_cast<uint4*>(lds + buf * kTileBytes + r * kKStride + ((sl ^ (2 * ((r >> 1) & 3))) << 4)) = make_uint4(0u, 0u, 0u, 0u);
    }

    // head_dim 48 = one full K=32 MFMA (d 0..31) + one half-filled K=32 MFMA (d 32..47 in k-slots 0..15, zeros in 16..31).
    // (The legacy v_mfma_f32_16x16x16_bf16 for the 16-wide remainder gave tile-dependent wrong results under some register
    //  allocations on ROCm 7.2 -- chained behind the 8-pass 16x16x32 through SrcC -- so it is not used.)
    vec8 qlo[QT], qhi[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int qr = min(q0 + t * 16 + c16, S - 1);
        qlo[t] = *reinterpret_cast<const vec8*>(qbase + qr * ld + 8 * g);
        union { vec8 v; uint4 w; } qh;
        qh.w = *reinterpret_cast<const uint4*>(qbase + qr * ld + 32 + 8 * (g & 1));
        if (g >= 2) qh.w = make_uint4(0u, 0u, 0u, 0u);
        qhi[t] = qh.v;
    }
    f32x4_t o[QT][3];
    float m[QT], l[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        m[t] = -INFINITY;
        l[t] = 0.f;
#pragma unroll
        for (int d = 0; d < 3; ++d) o[t][d] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    const float c = kScale * kLog2e;
    const int ntile = CAUSAL ? min((S + 63) / 64, ((qb * 4 + 4) * (QT * 16) + 63) / 64) : (S + 63) / 64;
    gload(0);
    __syncthreads();   // zero halves in place before the first real halves land next to them
    lstore(0);
    if (ntile > 1) gload(64);
    __syncthreads();
    for (int it = 0; it < ntile; ++it) {
        const int k0 = it * 64;
        const unsigned char* kt_l = lds + (it & 1) * kTileBytes;
        const unsigned char* vt_l = kt_l + 64 * kKStride;
        f32x4_t st[QT][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const unsigned char* kr = kt_l + (kt * 16 + c16) * kKStride;
            const int ksw = 2 * ((c16 >> 1) & 3);
            const vec8 klo = *reinterpret_cast<const vec8*>(kr + ((g ^ ksw) << 4));
            const vec8 khi = *reinterpret_cast<const vec8*>(kr + (((4 + g) ^ ksw) << 4));   // d 32+8g .. 39+8g (g < 2) | zeros
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                f32x4_t a = Mma16<TT>::mfma(klo, qlo[t], f32x4_t{0.f, 0.f, 0.f, 0.f});
                st[t][kt] = Mma16<TT>::mfma(khi, qhi[t], a);
            }
        }
        if (k0 + 64 > S) {   // key tail: rows past S are masked out
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (k0 + kt * 16 + 4 * g + r >= S) {
#pragma unroll
                        for (int t = 0; t < QT; ++t) st[t][kt][r] = -INFINITY;
                    }
        }
        if (CAUSAL && k0 + 63 > q0) {   // (wave-uniform: tiles that reach past this wave's first query)
#pragma unroll
            for (int t = 0; t < QT; ++t)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (k0 + kt * 16 + 4 * g + r > q0 + t * 16 + c16) st[t][kt][r] = -INFINITY;
        }
        vec8 pb[QT][2];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float mx = st[t][0][0];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[t][kt][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mn = fmaxf(m[t], mx);
            const float alpha = __builtin_amdgcn_exp2f((m[t] - mn) * c);   // raw v_exp_f32: arguments are <= 0, denormal results may flush
            m[t] = mn;
            const float mc = mn * c;
            float ps = 0.f;
            float p[4][4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[kt][r] = __builtin_amdgcn_exp2f(st[t][kt][r] * c - mc);
                    ps += p[kt][r];
                }
            l[t] = l[t] * alpha + ps;
            if (!__all(alpha == 1.0f)) {   // the running maxima of a tile's 16 queries stop moving after a few tiles: skip the no-op rescale
#pragma unroll
                for (int d = 0; d < 3; ++d)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[t][d][r] *= alpha;
            }
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int j = 0; j < 8; ++j) pb[t][hh][j] = (elem_t)p[2 * hh + (j >> 2)][j & 3];
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const unsigned char* vr = vt_l + (d * 16 + c16) * kVStride;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                union { vec8 v; uint2 u[2]; } a;
                a.u[0] = *reinterpret_cast<const uint2*>(vr + (((8 * hh + g) ^ c16) << 3));       // keys k0 + 32hh + 4g .. +3
                a.u[1] = *reinterpret_cast<const uint2*>(vr + (((8 * hh + g + 4) ^ c16) << 3));   // keys k0 + 32hh + 16 + 4g .. +3
#pragma unroll
                for (int t = 0; t < QT; ++t) o[t][d] = Mma16<TT>::mfma(a.v, pb[t][hh], o[t][d]);
            }
        }
        // publish tile it+1 (already in registers) into the other buffer, then fetch tile it+2
        if (it + 1 < ntile) {
            lstore((it + 1) & 1);
            if (it + 2 < ntile) gload(k0 + 128);
        }
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        float lt = l[t];
        lt += __shfl_xor(lt, 16);
        lt += __shfl_xor(lt, 32);
        const float inv = 1.0f / lt;
        const int qr = q0 + t * 16 + c16;
        if (qr < S && live) {
            TT* yr = y + ((long)f * S + qr) * E + h * kHeadDim + 4 * g;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                float v[4] = {o[t][d][0] * inv, o[t][d][1] * inv, o[t][d][2] * inv, o[t][d][3] * inv};
                store4(yr + d * 16, v);
            }
        }
    }
}

template <int QT, typename TT, bool CAUSAL = false>
__global__ __launch_bounds__(256) void attn_spatial_mfma_kernel(const TT* __restrict__ qk, const TT* __restrict__ vt,
                                                                TT* __restrict__ y, int S, int S_pad, int H, int nq, int npairs) {
    attn_spatial_mfma_body<QT, TT, CAUSAL, false>(qk, vt, y, S, S_pad, H, nq, npairs, blockIdx.x, threadIdx.x, 0, true);
}

constexpr int kTmax = 32;
// Slot ranges (TemporalRange, kernels.h): the qkv rows hold the Tn "new" history slots [t0, t0 + Tn) of every (scene, position);
// the k | v rows of the t0 earlier slots come from the layer's persistent cache [B][Tcap][S][2E] (written by an earlier launch with
// write = 1).  Every query runs the same sequential online softmax over keys 0 .. tq whichever launch its keys came from, so a
// split pass (slots 0..P-1 ahead of time, slot P later) is bit-identical to one pass over all P + 1 slots.
// (body as a device function: the decode engine's background workers, bg_worker.h, call it with the threads of their 512-thread workgroup --
//  threads >= HG * TMAX * 4 only take the barrier; `blk` = block of the virtual grid)
template <typename T, int HG, int TMAX>
__device__ __forceinline__ void attn_temporal_body(const T* __restrict__ qkv, T* __restrict__ y, int Tn, int S, int H, TemporalRange tr, long blk, int tid) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char dyn_lds[];
    float* const sm = reinterpret_cast<float*>(dyn_lds);   // [T][3][HG*48]
    constexpr int W = HG * kHeadDim;
    constexpr int NT = HG * TMAX * 4;
    const int hg = (int)(blk % (H / HG));
    const long bs = blk / (H / HG);          // b*S + s
    const int s = (int)(bs % S);
    const int b = (int)(bs / S);
    const int E = H * kHeadDim;
    const long ld = 3L * E;
    const int t0 = tr.t0, T_ = tr.t0 + Tn;
    T* cache = reinterpret_cast<T*>(tr.cache);
    constexpr int chunks_per_seg = W / 8;
    const int n_chunks = T_ * 3 * chunks_per_seg;
    constexpr int kIter = (TMAX * 3 * chunks_per_seg + NT - 1) / NT;
#pragma unroll
    for (int it = 0; it < kIter; ++it) {   // all 16-byte loads of the thread are in flight together
        const int c = tid + NT * it;
        if (c < n_chunks && tid < NT) {
            const int cc = c % chunks_per_seg, seg = (c / chunks_per_seg) % 3, t = c / (3 * chunks_per_seg);
            if ((t < t0 || t < tr.q0) && seg == 0) continue;        // queries of cached slots / of slots nobody consumes are not needed (their q rows may not exist)
            float v8[8];
            T* cp = cache ? cache + (((long)b * tr.Tcap + t) * S + s) * 2L * E + (long)(seg - 1) * E + hg * W + cc * 8 : nullptr;
            if (t < t0) load8(cp, v8);
            else load8(qkv + (((long)b * Tn + (t - t0)) * S + s) * ld + (long)seg * E + hg * W + cc * 8, v8);
            float* d = sm + ((t * 3 + seg) * W + cc * 8);
            *reinterpret_cast<float4*>(d) = make_float4(v8[0], v8[1], v8[2], v8[3]);
            *reinterpret_cast<float4*>(d + 4) = make_float4(v8[4], v8[5], v8[6], v8[7]);
            if (tr.write && t >= t0 && seg > 0) {
                const float lo4[4] = {v8[0], v8[1], v8[2], v8[3]}, hi4[4] = {v8[4], v8[5], v8[6], v8[7]};
                store4(cp, lo4);
                store4(cp + 4, hi4);
            }
        }
    }
    __syncthreads();
    if (tid >= NT) return;
    // 4 lanes per (head, query frame): lane part p owns head-dim slice [12p, 12p+12)
    const int part = tid & 3, tq = (tid >> 2) % TMAX, hl = tid / (4 * TMAX);
    const bool active = tq >= t0 && tq >= tr.q0 && tq < T_;
    float q[12], o[12];
    const float* qp = sm + ((active ? tq : t0) * 3 + 0) * W + hl * kHeadDim + part * 12;
#pragma unroll
    for (int d = 0; d < 12; d += 4) {
        const float4 q4 = *reinterpret_cast<const float4*>(qp + d);
        q[d] = q4.x; q[d + 1] = q4.y; q[d + 2] = q4.z; q[d + 3] = q4.w;
        o[d] = 0.f; o[d + 1] = 0.f; o[d + 2] = 0.f; o[d + 3] = 0.f;
    }
    float m = -INFINITY, l = 0.f;
    const int tmax = active ? tq : -1;
    for (int tk = 0; tk < T_; ++tk) {      // uniform trip count (shuffles below need the 4 partner lanes); masked past tq
        const float* kp = sm + (tk * 3 + 1) * W + hl * kHeadDim + part * 12;
        const float* vp = kp + W;
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < 12; d += 4) {
            const float4 k4 = *reinterpret_cast<const float4*>(kp + d);
            a = fmaf(q[d], k4.x, a); a = fmaf(q[d + 1], k4.y, a); a = fmaf(q[d + 2], k4.z, a); a = fmaf(q[d + 3], k4.w, a);
        }
        // the 4 lanes of a query are one DPP quad: quad_perm [1,0,3,2] / [2,3,0,1] (same sums as __shfl_xor 1 / 2, without the two
        // dependent ds_bpermute round trips per key)
        a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0xB1, 0xf, 0xf, true));
        a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x4E, 0xf, 0xf, true));
        if (tk <= tmax) {
            a *= kScale;
            const float mn = fmaxf(m, a);
            // bf16 mode: hardware exp (as in the spatial kernel); fp32 parity mode keeps expf
            const float alpha = sizeof(T) == 2 ? __expf(m - mn) : expf(m - mn);
            const float p = sizeof(T) == 2 ? __expf(a - mn) : expf(a - mn);
            m = mn;
            l = l * alpha + p;
#pragma unroll
            for (int d = 0; d < 12; d += 4) {
                const float4 v4 = *reinterpret_cast<const float4*>(vp + d);
                o[d] = fmaf(p, v4.x, o[d] * alpha); o[d + 1] = fmaf(p, v4.y, o[d + 1] * alpha);
                o[d + 2] = fmaf(p, v4.z, o[d + 2] * alpha); o[d + 3] = fmaf(p, v4.w, o[d + 3] * alpha);
            }
        }
    }
    if (!active) return;
    const float inv = 1.0f / l;
    T* yp = y + (((long)b * Tn + (tq - t0)) * S + s) * (long)E + (hg * HG + hl) * kHeadDim + part * 12;
#pragma unroll
    for (int d = 0; d < 12; d += 4) {
        float t4[4] = {o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv};
        store4(yp + d, t4);
    }
}

template <typename T, int HG, int TMAX = kTmax>
__global__ __launch_bounds__(HG * TMAX * 4) void attn_temporal_kernel(const T* __restrict__ qkv, T* __restrict__ y, int Tn, int S, int H,
                                                                      TemporalRange tr) {
    attn_temporal_body<T, HG, TMAX>(qkv, y, Tn, S, H, tr, (long)blockIdx.x, (int)threadIdx.x);
}

}  // namespace umgen
