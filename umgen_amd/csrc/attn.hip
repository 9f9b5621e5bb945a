// Attention kernels (replace flash_attn_func at module.py:218-225 / 497-504; semantics: exact softmax attention,
// scale 1/sqrt(48) applied to q.k in fp32, bottom-right aligned causal mask where causal).  head_dim is 48 everywhere.
#include <cstdlib>
#include <type_traits>

#include "bg_queue.h"
#include "attn_body.h"

namespace umgen {

#ifdef UMGEN_ATTN_VARIANTS   // only tools/micro/attn_bench.hip builds these (measured round 2, profiles/r02_attn_variants.txt)
// ---------------------------------------------------------------------------------------------------------
// Second form of the same kernel (same tiles, LDS images and MFMA operand maps), with the levers of the round-2 VALU diet as
// template bits so each one is measured on its own (tools/micro/attn_bench.hip):
//   1  cross-row maxima through v_permlane16_swap / v_permlane32_swap (VALU) instead of two ds_bpermute round trips
//   2  softmax denominators on the matrix pipe: a fourth "d block" whose A operand is all ones sums the SAME bf16 p values the
//      P.V product uses (l = sum of rounded p; 4 MFMAs per key tile instead of 32 v_add_f32 + the final cross-lane sum)
//   4  O is rescaled on every tile (no wave-uniform branch in the middle of the tile body)
//   8  P.V per query tile right behind its softmax (V fragments of the whole key tile held in registers), so the second query
//      tile's softmax runs beside the first one's MFMAs
//   16 s_setprio 1 around the MFMA groups
// The key tail (S % 64) is a peeled last tile: the body of full tiles carries no masking code.
// ---------------------------------------------------------------------------------------------------------
// (inline-asm maxima reading MFMA results gave NaNs: the hazard recogniser places no wait states for asm operands)
__device__ __forceinline__ float vmax2(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ float vmax3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float xmax16(float x) {   // max(x[lane], x[lane ^ 16])
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xmax32(float x) {   // max(x[lane], x[lane ^ 32])
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

constexpr int kTB = 15360;    // tile buffer of the variants (fits round 1's 144-byte K rows too)
int g_attn_variant = 0;   // set by the bench

template <int QT, int OPT>
__global__ __launch_bounds__((OPT & 8192) ? 512 : 256) void attn_spatial_mfma2_kernel(const bf16_t* __restrict__ qk, const bf16_t* __restrict__ vt,
                                                                 bf16_t* __restrict__ y, int S, int S_pad, int H, int nq, int npairs) {
    constexpr bool PERM = OPT & 1, ONES = OPT & 2, ALWAYS = OPT & 4, PER_T = OPT & 8, PRIO = OPT & 16;
    constexpr bool X_NOLOAD = OPT & 64, X_NOEXP = OPT & 128, X_NOBAR = OPT & 256, X_UNIF = OPT & 512, X_UNIFV = OPT & 1024;
    // 2048: K image with 128-byte rows, 16-byte chunk c of row r at c ^ 2*((r >> 1) & 3) (conflict-free for the ds_read_b128 lane groups
    // {0-3,12-15,20-27} ...; the 144-byte rows are 2-way), d 32..47 as two whole chunks + two zero chunks (k-slot groups 2, 3 are zero)
    // 4096: K / Vt tiles straight from global memory into the LDS images (global_load_lds_dwordx4, swizzles applied on the source
    // side, 16-byte chunk XOR for Vt as well): no staging registers, no ds_write pass
    constexpr bool GLDS = OPT & 4096;
    // 8192: 8 waves (256 queries) per workgroup share the K / Vt tiles: half the staging traffic per query (GLDS staging only)
    constexpr int NWV = (OPT & 8192) ? 8 : 4;
    static_assert(NWV == 4 || GLDS, "the 8-wave form stages with global_load_lds");
    constexpr bool NEWK = (OPT & 2048) || GLDS;
    constexpr int kKStride = 144;   // (round 1's K rows, kept here for the A/B; shadows the namespace constant)
    constexpr int kKB = NEWK ? 64 * 128 : 64 * kKStride;   // timing experiments only (wrong results)
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kTB];
    const int E = H * kHeadDim;
    const int b = blockIdx.x;
    const int group = b / (8 * nq), rem = b % (8 * nq);
    const int pair = group * 8 + (rem & 7), qb = rem >> 3;
    if (pair >= npairs) return;
    const int f = pair / H, h = pair % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, g = lane >> 4;
    const int q0 = (qb * NWV + wave) * (QT * 16);
    const long ld = 2L * E;
    const bf16_t* qbase = qk + (long)f * S * ld + h * kHeadDim;
    const bf16_t* kbase = qbase + E;
    const bf16_t* vbase = vt + ((long)f * H + h) * kHeadDim * S_pad;

    // staging: identical to attn_spatial_mfma_kernel
    const bool mid_is_k = tid < 128;
    const int r0 = tid / 6, p0 = tid % 6;
    const int c1 = tid + 256;
    const int r1 = mid_is_k ? c1 / 6 : (c1 - 384) >> 3, p1 = mid_is_k ? c1 % 6 : (c1 - 384) & 7;
    const int c2 = tid + 512 - 384, r2 = c2 >> 3, p2 = c2 & 7;
    const bf16_t* g0 = kbase + p0 * 8;
    const bf16_t* g1 = mid_is_k ? kbase + p1 * 8 : vbase + (long)r1 * S_pad + p1 * 8;
    const bf16_t* g2 = vbase + (long)r2 * S_pad + p2 * 8;
    auto koff = [&](int r, int pp) {
        return NEWK ? r * 128 + ((pp ^ (2 * ((r >> 1) & 3))) << 4) : r * kKStride + (pp < 4 ? pp * 16 : 64 + (pp - 4) * 32);
    };
    auto voff = [&](int r, int pp) { return kKB + r * kVStride + (((2 * pp) ^ (r & 15)) & ~1) * 8; };
    const int l0 = koff(r0, p0);
    const int l1 = mid_is_k ? koff(r1, p1) : voff(r1, p1);
    const int l2 = voff(r2, p2);
    const bool hi0 = !NEWK && p0 >= 4, hi1 = !NEWK && mid_is_k && p1 >= 4, swap1 = !mid_is_k && (r1 & 1), swap2 = r2 & 1;
    uint4 sg0, sg1, sg2;
    auto gload = [&](int k0) {
        sg0 = *reinterpret_cast<const uint4*>(g0 + (long)min(k0 + r0, S - 1) * ld);
        sg1 = *reinterpret_cast<const uint4*>(mid_is_k ? g1 + (long)min(k0 + r1, S - 1) * ld : g1 + k0);
        sg2 = *reinterpret_cast<const uint4*>(g2 + k0);
    };
    auto kstore = [](unsigned char* dst, uint4 v, bool hi) {
        if (hi) {
            *reinterpret_cast<uint2*>(dst) = make_uint2(v.x, v.y);
            *reinterpret_cast<uint2*>(dst + 16) = make_uint2(v.z, v.w);
        } else {
            *reinterpret_cast<uint4*>(dst) = v;
        }
    };
    auto vstore = [](unsigned char* dst, uint4 v, bool sw) {
        *reinterpret_cast<uint4*>(dst) = sw ? make_uint4(v.z, v.w, v.x, v.y) : v;
    };
    auto lstore = [&](int buf) {
        unsigned char* base = lds + buf * kTB;
        kstore(base + l0, sg0, hi0);
        if (mid_is_k) kstore(base + l1, sg1, hi1); else vstore(base + l1, sg1, swap1);
        vstore(base + l2, sg2, swap2);
    };
    // GLDS: 14 one-KB segments per tile (8 K rows or 8 Vt rows each); wave w issues segments w, w + 4, w + 8, w + 12
    auto issue = [&](int buf, int k0) {
#pragma unroll
        for (int i = 0; i < (14 + NWV - 1) / NWV; ++i) {
            const int seg = wave + NWV * i;
            if (seg < 14) {
                unsigned char* dst = lds + buf * kTB + seg * 1024;
                const int row = (seg < 8 ? 8 * seg : 8 * (seg - 8)) + (lane >> 3), pc = lane & 7;
                if (seg < 8) {
                    const int lc = pc ^ (2 * ((row >> 1) & 3));
                    if (lc < 6)   // (chunks 6, 7 hold the zeros written once: their lanes stay off)
                        __builtin_amdgcn_global_load_lds((const void*)(kbase + (long)min(k0 + row, S - 1) * ld + lc * 8),
                                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                } else {
                    const int lc = pc ^ ((row >> 1) & 7);
                    __builtin_amdgcn_global_load_lds((const void*)(vbase + (long)row * S_pad + k0 + lc * 8),
                                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                }
            }
        }
    };
    if (NEWK && tid < 256) {
        const int buf = tid >> 7, r = (tid >> 1) & 63, sl = 6 + (tid & 1);   // chunks 6, 7 of every K row, both buffers
        *reinterpret_cast<uint4*>(lds + buf * kTB + r * 128 + ((sl ^ (2 * ((r >> 1) & 3))) << 4)) = make_uint4(0u, 0u, 0u, 0u);
    } else {
        for (int i = tid; i < 2 * 64 * 4; i += 256) {
            const int buf = i >> 8, r = (i >> 2) & 63, sl = i & 3;
            *reinterpret_cast<uint2*>(lds + buf * kTB + r * kKStride + 64 + sl * 16 + 8) = make_uint2(0u, 0u);
        }
    }

    bf16x8_t qlo[QT], qhi[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int qr = min(q0 + t * 16 + c16, S - 1);
        qlo[t] = *reinterpret_cast<const bf16x8_t*>(qbase + qr * ld + 8 * g);
        union { bf16x8_t v; uint2 u[2]; uint4 w; } qh;
        if (NEWK) {
            qh.w = *reinterpret_cast<const uint4*>(qbase + qr * ld + 32 + 8 * (g & 1));
            if (g >= 2) qh.w = make_uint4(0u, 0u, 0u, 0u);
        } else {
            qh.u[0] = *reinterpret_cast<const uint2*>(qbase + qr * ld + 32 + 4 * g);
            qh.u[1] = make_uint2(0u, 0u);
        }
        qhi[t] = qh.v;
    }
    f32x4_t o[QT][3], ls[QT];
    float m[QT], l[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        m[t] = -INFINITY;
        l[t] = 0.f;
        ls[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < 3; ++d) o[t][d] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    bf16x8_t ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
    const float c = kScale * kLog2e;
    const int ntile = (S + 63) / 64;

    auto tile = [&](auto masked, int it) {
        constexpr bool MASK = decltype(masked)::value;
        // byte offset of 8-byte key group gr inside this lane's Vt row
        auto vgran = [&](int gr) { return GLDS ? (((gr >> 1) ^ ((c16 >> 1) & 7)) << 4) + (gr & 1) * 8 : (gr ^ c16) << 3; };
        const int k0 = it * 64;
        const unsigned char* kt_l = lds + (it & 1) * kTB;
        const unsigned char* vt_l = kt_l + kKB;
        if (GLDS && !X_NOLOAD && it + 1 < ntile) issue((it + 1) & 1, k0 + 64);   // (that buffer's readers passed the barrier)
        f32x4_t st[QT][4];
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const unsigned char* kr = kt_l + (kt * 16 + (X_UNIF ? 0 : c16)) * (NEWK ? 128 : kKStride);   // X_UNIF: one address per wave (broadcast reads)
            const int ksw = 2 * ((c16 >> 1) & 3);
            const bf16x8_t klo = *reinterpret_cast<const bf16x8_t*>(kr + (X_UNIF ? 0 : NEWK ? (g ^ ksw) << 4 : 16 * g));
            const bf16x8_t khi = *reinterpret_cast<const bf16x8_t*>(kr + (X_UNIF ? 64 : NEWK ? ((4 + g) ^ ksw) << 4 : 64 + 16 * g));
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                f32x4_t a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(klo, qlo[t], f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                st[t][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(khi, qhi[t], a, 0, 0, 0);
            }
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        bf16x8_t va[3][2];
        if (PER_T) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const unsigned char* vr = vt_l + (d * 16 + c16) * kVStride;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    union { bf16x8_t v; uint2 u[2]; } a;
                    a.u[0] = *reinterpret_cast<const uint2*>(vr + vgran(8 * hh + g));
                    a.u[1] = *reinterpret_cast<const uint2*>(vr + vgran(8 * hh + g + 4));
                    va[d][hh] = a.v;
                }
            }
        }
        if (MASK) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (k0 + kt * 16 + 4 * g + r >= S) {
#pragma unroll
                        for (int t = 0; t < QT; ++t) st[t][kt][r] = -INFINITY;
                    }
        }
        bf16x8_t pb[QT][2];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float mx;
            if (PERM) {
                mx = vmax3(st[t][0][0], st[t][0][1], st[t][0][2]);
                mx = vmax3(mx, st[t][0][3], st[t][1][0]);
                mx = vmax3(mx, st[t][1][1], st[t][1][2]);
                mx = vmax3(mx, st[t][1][3], st[t][2][0]);
                mx = vmax3(mx, st[t][2][1], st[t][2][2]);
                mx = vmax3(mx, st[t][2][3], st[t][3][0]);
                mx = vmax3(mx, st[t][3][1], st[t][3][2]);
                mx = fmaxf(mx, st[t][3][3]);   // (a compiler-visible VALU write in front of the swap: it places the hazard wait states)
                mx = xmax16(mx);
                mx = xmax32(mx);
            } else {
                mx = st[t][0][0];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[t][kt][r]);
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
            }
            const float mn = PERM ? vmax2(m[t], mx) : fmaxf(m[t], mx);
            const float alpha = __builtin_amdgcn_exp2f((m[t] - mn) * c);
            m[t] = mn;
            const float mc = mn * c;
            float ps = 0.f;
            float p[4][4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[kt][r] = X_NOEXP ? st[t][kt][r] * c - mc : __builtin_amdgcn_exp2f(st[t][kt][r] * c - mc);
                    if (!ONES) ps += p[kt][r];
                }
            if (!ONES) l[t] = l[t] * alpha + ps;
            if (ALWAYS || !__all(alpha == 1.0f)) {
#pragma unroll
                for (int d = 0; d < 3; ++d)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[t][d][r] *= alpha;
                if (ONES) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) ls[t][r] *= alpha;
                }
            }
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int j = 0; j < 8; ++j) pb[t][hh][j] = (__bf16)p[2 * hh + (j >> 2)][j & 3];
            if (PER_T) {
                if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
                    for (int d = 0; d < 3; ++d) o[t][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va[d][hh], pb[t][hh], o[t][d], 0, 0, 0);
                    if (ONES) ls[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pb[t][hh], ls[t], 0, 0, 0);
                }
                if (PRIO) __builtin_amdgcn_s_setprio(0);
            }
        }
        if (!PER_T) {
            if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const unsigned char* vr = vt_l + (d * 16 + (X_UNIFV ? 0 : c16)) * kVStride;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    union { bf16x8_t v; uint2 u[2]; } a;
                    a.u[0] = *reinterpret_cast<const uint2*>(vr + (X_UNIFV ? 8 * hh : vgran(8 * hh + g)));
                    a.u[1] = *reinterpret_cast<const uint2*>(vr + (X_UNIFV ? 8 * hh + 32 : vgran(8 * hh + g + 4)));
#pragma unroll
                    for (int t = 0; t < QT; ++t) o[t][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, pb[t][hh], o[t][d], 0, 0, 0);
                }
            }
            if (ONES) {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                    for (int t = 0; t < QT; ++t) ls[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pb[t][hh], ls[t], 0, 0, 0);
            }
            if (PRIO) __builtin_amdgcn_s_setprio(0);
        }
        if (!GLDS && !X_NOLOAD && it + 1 < ntile) {
            lstore((it + 1) & 1);
            if (it + 2 < ntile) gload(k0 + 128);
        }
        if (!X_NOBAR) __syncthreads();   // (GLDS: the barrier's release waits for this wave's LDS-bound loads)
    };

    if (GLDS) {
        __syncthreads();   // zero chunks written
        issue(0, 0);
        __syncthreads();
    } else {
        gload(0);
        __syncthreads();
        lstore(0);
        if (ntile > 1) gload(64);
        __syncthreads();
    }
    const bool tail = (S & 63) != 0;
    const int nfull = tail ? ntile - 1 : ntile;
    for (int it = 0; it < nfull; ++it) tile(std::false_type{}, it);
    if (tail) tile(std::true_type{}, ntile - 1);
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        float lt;
        if (ONES) {
            lt = ls[t][0];            // every row of the ones block holds the same sum over all keys
        } else {
            lt = l[t];
            lt += __shfl_xor(lt, 16);
            lt += __shfl_xor(lt, 32);
        }
        const float inv = 1.0f / lt;
        const int qr = q0 + t * 16 + c16;
        if (qr < S) {
            bf16_t* yr = y + ((long)f * S + qr) * E + h * kHeadDim + 4 * g;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                float v[4] = {o[t][d][0] * inv, o[t][d][1] * inv, o[t][d][2] * inv, o[t][d][3] * inv};
                store4(yr + d * 16, v);
            }
        }
    }
}

#endif   // UMGEN_ATTN_VARIANTS

template <typename TT>
void launch_attn_spatial_mfma(hipStream_t s, const TT* qk, const TT* vt, TT* y, int F, int S, int S_pad, int H) {
    constexpr int QT = UMGEN_ATTN_QT;
    const int nq = (S + 4 * QT * 16 - 1) / (4 * QT * 16);
    // F*H pairs; groups of 8 pairs need F*H % 8 == 0 -- pad the pair count up and let the surplus blocks exit
    const int pairs = ((F * H + 7) / 8) * 8;
    const dim3 grid(pairs * nq), block(256);
    if (BgRecorder* rec = g_bg_rec) {      // two blocks of the virtual grid per unit (the two halves of a 512-thread worker)
        const double unit_flops = 2.0 * 4.0 * (4 * QT * 16) * (double)S * kHeadDim;
        BgOp& o = rec->add(BG_ATTN_S, 0, ((long)pairs * nq + 1) / 2, 4, (unsigned)(unit_flops / 1.3e12 * 1e8));
        o.h.i0 = S; o.h.i1 = S_pad; o.h.i2 = H; o.h.i3 = nq; o.a.i4 = F * H; o.a.i5 = pairs * nq;
        o.a.p0 = const_cast<TT*>(qk); o.a.p1 = const_cast<TT*>(vt); o.a.p2 = y;
        return;
    }
#ifdef UMGEN_ATTN_VARIANTS
#define UMGEN_ATTN_CASE(OPT) \
    case OPT: { \
        const int nw = ((OPT) & 8192) ? 8 : 4, nq2 = (S + nw * QT * 16 - 1) / (nw * QT * 16); \
        hipLaunchKernelGGL((attn_spatial_mfma2_kernel<QT, OPT>), dim3(pairs * nq2), dim3(nw * 64), 0, s, qk, vt, y, S, S_pad, H, nq2, F * H); \
        return; }
    if constexpr (sizeof(TT) == 2 && !__is_same(TT, f16_t)) {   // (the variant kernels exist for bf16 only)
    switch (g_attn_variant) {
        UMGEN_ATTN_CASE(1) UMGEN_ATTN_CASE(2) UMGEN_ATTN_CASE(3) UMGEN_ATTN_CASE(7) UMGEN_ATTN_CASE(11) UMGEN_ATTN_CASE(15)
        UMGEN_ATTN_CASE(19) UMGEN_ATTN_CASE(23) UMGEN_ATTN_CASE(27) UMGEN_ATTN_CASE(31) UMGEN_ATTN_CASE(32)
        UMGEN_ATTN_CASE(71) UMGEN_ATTN_CASE(327) UMGEN_ATTN_CASE(135) UMGEN_ATTN_CASE(455) UMGEN_ATTN_CASE(967) UMGEN_ATTN_CASE(1479) UMGEN_ATTN_CASE(1991)
        UMGEN_ATTN_CASE(2048) UMGEN_ATTN_CASE(2051) UMGEN_ATTN_CASE(2055) UMGEN_ATTN_CASE(2503)
        UMGEN_ATTN_CASE(4096) UMGEN_ATTN_CASE(4099) UMGEN_ATTN_CASE(4103) UMGEN_ATTN_CASE(4107) UMGEN_ATTN_CASE(4115)
        UMGEN_ATTN_CASE(12288) UMGEN_ATTN_CASE(12291) UMGEN_ATTN_CASE(12295)
        default: break;
    }
    }
#undef UMGEN_ATTN_CASE
#endif
    hipLaunchKernelGGL((attn_spatial_mfma_kernel<QT, TT>), grid, block, 0, s, qk, vt, y, S, S_pad, H, nq, F * H);
}
template void launch_attn_spatial_mfma<bf16_t>(hipStream_t, const bf16_t*, const bf16_t*, bf16_t*, int, int, int, int);
template void launch_attn_spatial_mfma<f16_t>(hipStream_t, const f16_t*, const f16_t*, f16_t*, int, int, int, int);
template <typename TT>
void launch_attn_causal_mfma(hipStream_t s, const TT* qk, const TT* vt, TT* y, int F, int S, int S_pad, int H) {
    constexpr int QT = UMGEN_ATTN_QT;
    const int nq = (S + 4 * QT * 16 - 1) / (4 * QT * 16);
    const int pairs = ((F * H + 7) / 8) * 8;
    hipLaunchKernelGGL((attn_spatial_mfma_kernel<QT, TT, true>), dim3(pairs * nq), dim3(256), 0, s, qk, vt, y, S, S_pad, H, nq, F * H);
}
template void launch_attn_causal_mfma<bf16_t>(hipStream_t, const bf16_t*, const bf16_t*, bf16_t*, int, int, int, int);
template void launch_attn_causal_mfma<f16_t>(hipStream_t, const f16_t*, const f16_t*, f16_t*, int, int, int, int);

// ---------------------------------------------------------------------------------------------------------
// spatial, generic VALU (parity mode): one thread per query, keys streamed through LDS in tiles of 32
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(128) void attn_spatial_valu_kernel(const T* __restrict__ qk, const T* __restrict__ vt, T* __restrict__ y,
                                                                int S, int S_pad, int H, int causal = 0) {
    __shared__ float sk[32][kHeadDim + 1];
    __shared__ float sv[32][kHeadDim + 1];
    const int E = H * kHeadDim;
    const int f = blockIdx.z, h = blockIdx.y;
    const int qi = blockIdx.x * 128 + threadIdx.x;
    const long ld = 2L * E;
    const T* qbase = qk + (long)f * S * ld + h * kHeadDim;
    const T* kbase = qbase + E;
    const T* vbase = vt + ((long)f * H + h) * kHeadDim * S_pad;
    float q[kHeadDim], o[kHeadDim];
    const int qr = min(qi, S - 1);
#pragma unroll
    for (int d = 0; d < kHeadDim; ++d) {
        q[d] = Cvt<T>::to_f(qbase[qr * ld + d]);
        o[d] = 0.f;
    }
    float m = -INFINITY, l = 0.f;
    const int k_end = causal ? min(S, (int)blockIdx.x * 128 + 128) : S;     // (causal: the keys up to the workgroup's last query)
    for (int k0 = 0; k0 < k_end; k0 += 32) {
        __syncthreads();
        for (int e = threadIdx.x; e < 32 * kHeadDim; e += 128) {
            const int kk = e / kHeadDim, d = e % kHeadDim;
            const int kr = min(k0 + kk, S - 1);
            sk[kk][d] = Cvt<T>::to_f(kbase[kr * ld + d]);
            const int kk2 = e % 32, d2 = e / 32;
            sv[kk2][d2] = (k0 + kk2 < S) ? Cvt<T>::to_f(vbase[(long)d2 * S_pad + k0 + kk2]) : 0.f;
        }
        __syncthreads();
        float sc[32];
        float mx = -INFINITY;
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            float a = 0.f;
#pragma unroll
            for (int d = 0; d < kHeadDim; ++d) a = fmaf(q[d], sk[kk][d], a);
            a = (k0 + kk < S && !(causal && k0 + kk > qi)) ? a * kScale : -INFINITY;
            sc[kk] = a;
            mx = fmaxf(mx, a);
        }
        const float mn = fmaxf(m, mx);
        const float alpha = expf(m - mn);
        m = mn;
        l *= alpha;
#pragma unroll
        for (int d = 0; d < kHeadDim; ++d) o[d] *= alpha;
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            const float p = expf(sc[kk] - mn);
            l += p;
#pragma unroll
            for (int d = 0; d < kHeadDim; ++d) o[d] = fmaf(p, sv[kk][d], o[d]);
        }
    }
    if (qi < S) {
        const float inv = 1.0f / l;
        T* yr = y + ((long)f * S + qi) * E + h * kHeadDim;
#pragma unroll
        for (int d = 0; d < kHeadDim; ++d) yr[d] = Cvt<T>::from_f(o[d] * inv);
    }
}

// ---------------------------------------------------------------------------------------------------------
// spatial, fp32 on the matrix cores (parity mode): v_mfma_f32_32x32x2_f32 -- exact fp32 products and sums at the fp32 vector peak,
// where the one-thread-per-query VALU kernel above reaches 8 TFLOP/s (it was 4.7 s of a 10.7 s fp32 frame).  Same arithmetic
// contract (fp32 operands, fp32 softmax with expf, exact 1/l); another summation order of the keys, like every other attention here.
//   workgroup = 4 waves x 32 queries of one (frame, head); keys in tiles of 32 through LDS, dim-major: Kt[48][33], Vt[48][33]
//   S^T = K Q^T  ("swapped", so a query's scores are lane-local): A = K (row = key l % 32, k = dim 2 s + l / 32), B = Q^T held in
//         24 registers per lane; C: column = query l % 32, row = key 8 (r / 4) + 4 (l / 32) + r % 4 -- 16 of the tile's 32 keys per lane,
//         the other 16 in lane l ^ 32 (one exchange of the tile maximum per tile, the row sum is folded once at the end)
//   O^T += V^T P^T: A = V^T (row = dim, k = key), B = P^T straight from the score registers: the MFMA step that consumes keys
//         8 a + b (lanes < 32) and 8 a + 4 + b (lanes >= 32) takes register 4 a + b of every lane; 48 dims = one full and one
//         half-used 32-row tile
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_spatial_f32_mfma_kernel(const float* __restrict__ qk, const float* __restrict__ vt, float* __restrict__ y,
                                                                    int S, int S_pad, int H, int causal = 0) {
    __shared__ float sK[kHeadDim][33];
    __shared__ float sV[kHeadDim][33];
    const int E = H * kHeadDim;
    const int f = blockIdx.z, h = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l32 = lane & 31, half = lane >> 5;
    const long ld = 2L * E;
    const float* qbase = qk + (long)f * S * ld + h * kHeadDim;
    const float* kbase = qbase + E;
    const float* vbase = vt + ((long)f * H + h) * kHeadDim * S_pad;
    const int qi = blockIdx.x * 128 + wave * 32 + l32;
    const int qr = min(qi, S - 1);
    float qreg[24];
#pragma unroll
    for (int s2 = 0; s2 < 24; ++s2) qreg[s2] = qbase[qr * ld + 2 * s2 + half];
    f32x16_t o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m = -INFINITY, l = 0.f;
    // staging: 384 float4 of K (32 keys x 12 chunks) and 384 float4 of V^T (48 dims x 8 chunks) per tile, 256 threads
    float4 rk[2], rv[2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int e = tid + 256 * it;
            if (e < 384) {
                const int kk = e / 12, c = e % 12;
                rk[it] = *reinterpret_cast<const float4*>(kbase + (long)min(k0 + kk, S - 1) * ld + 4 * c);
                const int d = e / 8, c2 = e % 8;
                rv[it] = *reinterpret_cast<const float4*>(vbase + (long)d * S_pad + k0 + 4 * c2);     // pad columns are zero (engine.hip)
            }
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int e = tid + 256 * it;
            if (e < 384) {
                const int kk = e / 12, c = e % 12;
                sK[4 * c + 0][kk] = rk[it].x; sK[4 * c + 1][kk] = rk[it].y; sK[4 * c + 2][kk] = rk[it].z; sK[4 * c + 3][kk] = rk[it].w;
                const int d = e / 8, c2 = e % 8;
                sV[d][4 * c2 + 0] = rv[it].x; sV[d][4 * c2 + 1] = rv[it].y; sV[d][4 * c2 + 2] = rv[it].z; sV[d][4 * c2 + 3] = rv[it].w;
            }
        }
    };
    fetch(0);
    const int k_end = causal ? min(S, (int)blockIdx.x * 128 + 128) : S;     // (causal: the keys up to the workgroup's last query)
    for (int k0 = 0; k0 < k_end; k0 += 32) {
        stash();
        __syncthreads();
        if (k0 + 32 < k_end) fetch(k0 + 32);
        f32x16_t sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int s2 = 0; s2 < 24; ++s2) sc = __builtin_amdgcn_mfma_f32_32x32x2f32(sK[2 * s2 + half][l32], qreg[s2], sc, 0, 0, 0);
#endif
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + 8 * (r >> 2) + 4 * half + (r & 3);
            sc[r] = (key < S && !(causal && key > qi)) ? sc[r] * kScale : -INFINITY;
            mx = fmaxf(mx, sc[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mn = fmaxf(m, mx);
        const float alpha = expf(m - mn);
        m = mn;
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sc[r] = expf(sc[r] - mn);
            ps += sc[r];
        }
        l = l * alpha + ps;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int a4 = 0; a4 < 4; ++a4)
#pragma unroll
            for (int b4 = 0; b4 < 4; ++b4) {
                const int key = 8 * a4 + b4 + 4 * half;
                o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(sV[l32][key], sc[4 * a4 + b4], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(sV[32 + (l32 & 15)][key], sc[4 * a4 + b4], o1, 0, 0, 0);
            }
#endif
        __syncthreads();
    }
    l += __shfl_xor(l, 32);
    if (qi < S) {
        const float inv = 1.0f / l;
        float* yr = y + ((long)f * S + qi) * E + h * kHeadDim;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {      // rows (dims) 8 q4 + 4 half + 0..3 of the 32-row tile
            const float4 v = make_float4(o0[4 * q4] * inv, o0[4 * q4 + 1] * inv, o0[4 * q4 + 2] * inv, o0[4 * q4 + 3] * inv);
            *reinterpret_cast<float4*>(yr + 8 * q4 + 4 * half) = v;
        }
#pragma unroll
        for (int q4 = 0; q4 < 2; ++q4) {      // dims 32 .. 47: rows 0 .. 15 of the second tile
            const float4 v = make_float4(o1[4 * q4] * inv, o1[4 * q4 + 1] * inv, o1[4 * q4 + 2] * inv, o1[4 * q4 + 3] * inv);
            *reinterpret_cast<float4*>(yr + 32 + 8 * q4 + 4 * half) = v;
        }
    }
}

template <typename T>
void launch_attn_spatial_valu(hipStream_t s, const T* qk, const T* vt, T* y, int F, int S, int S_pad, int H) {
    dim3 grid((S + 127) / 128, H, F);
    hipLaunchKernelGGL(attn_spatial_valu_kernel<T>, grid, dim3(128), 0, s, qk, vt, y, S, S_pad, H);
}
void launch_attn_spatial_f32_mfma(hipStream_t s, const float* qk, const float* vt, float* y, int F, int S, int S_pad, int H) {
    static const bool off = getenv("UMGEN_FP32_MFMA") && getenv("UMGEN_FP32_MFMA")[0] == '0';   // 0: the one-thread-per-query VALU kernel (A/B)
    if (off) { launch_attn_spatial_valu<float>(s, qk, vt, y, F, S, S_pad, H); return; }
    dim3 grid((S + 127) / 128, H, F);
    hipLaunchKernelGGL(attn_spatial_f32_mfma_kernel, grid, dim3(256), 0, s, qk, vt, y, S, S_pad, H);
}
// causal S x S attention in fp32 (parity mode) / generic VALU: the OAR prefix pass
void launch_attn_causal_f32(hipStream_t s, const float* qk, const float* vt, float* y, int F, int S, int S_pad, int H) {
    static const bool off = getenv("UMGEN_FP32_MFMA") && getenv("UMGEN_FP32_MFMA")[0] == '0';
    dim3 grid((S + 127) / 128, H, F);
    if (off) hipLaunchKernelGGL(attn_spatial_valu_kernel<float>, grid, dim3(128), 0, s, qk, vt, y, S, S_pad, H, 1);
    else hipLaunchKernelGGL(attn_spatial_f32_mfma_kernel, grid, dim3(256), 0, s, qk, vt, y, S, S_pad, H, 1);
}
template void launch_attn_spatial_valu<float>(hipStream_t, const float*, const float*, float*, int, int, int, int);
template void launch_attn_spatial_valu<bf16_t>(hipStream_t, const bf16_t*, const bf16_t*, bf16_t*, int, int, int, int);
template void launch_attn_spatial_valu<f16_t>(hipStream_t, const f16_t*, const f16_t*, f16_t*, int, int, int, int);

// ---------------------------------------------------------------------------------------------------------
// temporal: causal attention over the T (<= 32) history frames of one spatial position.
// One workgroup = (scene b, position s, group of HG heads): the q | k | v segments of all T frames are staged once into LDS
// (every byte of the [R][3E] buffer is read exactly once -- the first version, one thread per (t, s, h) reading K/V straight
// from global, measured 5.5x the algorithmic HBM traffic with FETCH_SIZE), then thread (head, tq) runs its causal row.
// ---------------------------------------------------------------------------------------------------------
template <typename T>
void launch_attn_temporal(hipStream_t s, const T* qkv, T* y, int B, int Tn, int S, int H, TemporalRange tr) {
    // up to 32 history slots: 4 heads per workgroup when H allows; 33..64 slots (the 2x-context stress configuration): 64 query slots
    // per head, 2 heads per workgroup
    const int T_ = tr.t0 + Tn;
    if (BgRecorder* rec = g_bg_rec) {      // one block (scene, position, 4 heads) per unit
        if (sizeof(T) != 2 || H % 4 != 0 || T_ > kTmax) { rec->failed = "temporal attention variant"; return; }
        BgOp& o = rec->add(BG_ATTN_T, T_ <= 20 ? 20 : 32, (long)B * S * (H / 4), 16, 400);
        o.h.i0 = Tn; o.h.i1 = S; o.h.i2 = H; o.a.i4 = tr.t0; o.a.i5 = tr.Tcap; o.a.i6 = tr.write; o.a.i7 = tr.q0;
        o.a.p0 = const_cast<T*>(qkv); o.a.p1 = y; o.a.p2 = tr.cache;
        return;
    }
    if (T_ > kTmax) {
        const size_t shm = (size_t)T_ * 3 * 2 * kHeadDim * sizeof(float);
        if (H % 2 == 0)
            hipLaunchKernelGGL((attn_temporal_kernel<T, 2, 64>), dim3((unsigned)((long)B * S * (H / 2))), dim3(2 * 64 * 4), shm, s, qkv, y, Tn, S, H, tr);
        else
            hipLaunchKernelGGL((attn_temporal_kernel<T, 1, 64>), dim3((unsigned)((long)B * S * H)), dim3(64 * 4), shm / 2, s, qkv, y, Tn, S, H, tr);
        return;
    }
    if (H % 4 == 0) {
        const size_t shm = (size_t)T_ * 3 * 4 * kHeadDim * sizeof(float);
        // the evaluation window is 20 slots: with 20 query slots per head every lane of the 5 waves has a query (32 slots: 12 of 32 idle)
        if (T_ <= 20)
            hipLaunchKernelGGL((attn_temporal_kernel<T, 4, 20>), dim3((unsigned)((long)B * S * (H / 4))), dim3(4 * 20 * 4), shm, s, qkv, y, Tn, S, H, tr);
        else
            hipLaunchKernelGGL((attn_temporal_kernel<T, 4>), dim3((unsigned)((long)B * S * (H / 4))), dim3(4 * kTmax * 4), shm, s, qkv, y, Tn, S, H, tr);
    } else if (H % 2 == 0) {
        const size_t shm = (size_t)T_ * 3 * 2 * kHeadDim * sizeof(float);
        hipLaunchKernelGGL((attn_temporal_kernel<T, 2>), dim3((unsigned)((long)B * S * (H / 2))), dim3(2 * kTmax * 4), shm, s, qkv, y, Tn, S, H, tr);
    } else {
        const size_t shm = (size_t)T_ * 3 * 1 * kHeadDim * sizeof(float);
        hipLaunchKernelGGL((attn_temporal_kernel<T, 1>), dim3((unsigned)((long)B * S * H)), dim3(1 * kTmax * 4), shm, s, qkv, y, Tn, S, H, tr);
    }
}
template void launch_attn_temporal<float>(hipStream_t, const float*, float*, int, int, int, int, TemporalRange);
template void launch_attn_temporal<bf16_t>(hipStream_t, const bf16_t*, bf16_t*, int, int, int, int, TemporalRange);
template void launch_attn_temporal<f16_t>(hipStream_t, const f16_t*, f16_t*, int, int, int, int, TemporalRange);

}  // namespace umgen
