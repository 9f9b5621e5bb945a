// Shared device/host helpers for libumgen_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace umgen {

constexpr int kHeadDim = 48;      // n_embd / n_head for UMGen_Large (768/16), the 2x-width config (1536/32) and tests
constexpr int kSeq = 2207;        // scene positions per frame (infer_fun.py:118)
constexpr int kWave = 64;

typedef unsigned short bf16_t;    // raw bfloat16 bits
typedef _Float16 f16_t;           // IEEE half: the reference's own autocast dtype (UMGen.py:1604-1605), precision mode UMGEN_PREC_FP16
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__host__ __device__ inline float bf16_to_f32(bf16_t v) {
    union { uint32_t u; float f; } c;
    c.u = ((uint32_t)v) << 16;
    return c.f;
}
// round-to-nearest-even, NaN preserved (same as torch's float -> bfloat16).  On the device this is the hardware conversion
// (v_cvt_pk_bf16_f32, RNE; same bits for every non-NaN input) instead of six integer operations per value: the GEMM / LayerNorm /
// attention epilogues convert 32 values per lane and tile.
__host__ __device__ inline bf16_t f32_to_bf16(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(bf16_t, (__bf16)f);
#endif
    union { uint32_t u; float f; } c;
    c.f = f;
    uint32_t u = c.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

// float -> IEEE half bits on the HOST, round-to-nearest-even like torch's .half() (the integer formulation: without F16C flags the
// compiler's (_Float16) cast is a library call per element -- 2.4 G weights took 19 s instead of 6; checked bit for bit against that
// cast on 2 x 10^8 random and every 37th float of the exponents -26 .. 17)
inline uint16_t f32_to_f16_bits_host(float f) {
    union { uint32_t u; float f; } c;
    c.f = f;
    const uint32_t sign = c.u & 0x80000000u;
    c.u ^= sign;
    uint16_t o;
    if (c.u >= ((127u + 16u) << 23)) {
        o = (c.u > (255u << 23)) ? 0x7e00 : 0x7c00;              // nan : inf (|f| >= 65536; 65520 .. 65536 round up to inf below)
    } else if (c.u < (113u << 23)) {                              // half subnormal or zero: let the fp32 adder do the rounding
        union { uint32_t u; float f; } m;
        m.u = ((127u - 15u) + (23u - 10u) + 1u) << 23;
        c.f += m.f;
        o = (uint16_t)(c.u - m.u);
    } else {
        const uint32_t odd = (c.u >> 13) & 1u;
        c.u += ((uint32_t)(15 - 127) << 23) + 0xfffu;
        c.u += odd;
        o = (uint16_t)(c.u >> 13);
    }
    return (uint16_t)(o | (sign >> 16));
}

template <typename T> struct Cvt;
template <> struct Cvt<float> {
    __host__ __device__ static inline float to_f(float v) { return v; }
    __host__ __device__ static inline float from_f(float v) { return v; }
};
template <> struct Cvt<bf16_t> {
    __host__ __device__ static inline float to_f(bf16_t v) { return bf16_to_f32(v); }
    __host__ __device__ static inline bf16_t from_f(float v) { return f32_to_bf16(v); }
};
template <> struct Cvt<f16_t> {   // v_cvt_f32_f16 / v_cvt_f16_f32 (round-to-nearest-even, overflow -> inf like torch's .half())
    __host__ __device__ static inline float to_f(f16_t v) { return (float)v; }
    __host__ __device__ static inline f16_t from_f(float v) { return (f16_t)v; }
};

// 16-bit MFMA operand traits: both types use v_mfma_f32_16x16x32_* / 32x32x16_* at the same rate and with the same fragment
// layouts, so every matrix-core kernel is one template over the operand type
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
template <typename T> struct Mma16;
template <> struct Mma16<bf16_t> {
    typedef bf16x8_t vec;
    typedef __bf16 elem;
    __device__ static inline f32x4_t mfma(vec a, vec b, f32x4_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#else
        return c;
#endif
    }
    __device__ static inline f32x16_t mfma32(vec a, vec b, f32x16_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#else
        return c;
#endif
    }
};
template <> struct Mma16<f16_t> {
    typedef f16x8_t vec;
    typedef _Float16 elem;
    __device__ static inline f32x4_t mfma(vec a, vec b, f32x4_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#else
        return c;
#endif
    }
    __device__ static inline f32x16_t mfma32(vec a, vec b, f32x16_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#else
        return c;
#endif
    }
};

// wave64 sum, result broadcast to all lanes.  DPP row shifts / broadcasts (7 VALU ops) instead of six dependent
// ds_bpermute round trips: on the decode path the reduction latency is a visible fraction of a ~3 us kernel.
__device__ inline float wave_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));   // row_shr:1
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, true));   // row_shr:2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xe, true));   // row_shr:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xc, true));   // row_shr:8  -> lane 15 of each row
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, true));   // row_bcast:15
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xc, 0xf, true));   // row_bcast:31 -> lane 63
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// wave64 max / integer min with the same DPP schedule (lanes without a valid source keep their own value)
#define UMGEN_DPP_STEP(op, T, ctrl, rmask, bmask) v = op(v, T##_from_int(__builtin_amdgcn_update_dpp(T##_to_int(v), T##_to_int(v), ctrl, rmask, bmask, false)))
__device__ inline int f32_to_int(float x) { return __float_as_int(x); }
__device__ inline float f32_from_int(int x) { return __int_as_float(x); }
__device__ inline int i32_to_int(int x) { return x; }
__device__ inline int i32_from_int(int x) { return x; }
__device__ inline float wave_max(float v) {
    UMGEN_DPP_STEP(fmaxf, f32, 0x111, 0xf, 0xf); UMGEN_DPP_STEP(fmaxf, f32, 0x112, 0xf, 0xf);
    UMGEN_DPP_STEP(fmaxf, f32, 0x114, 0xf, 0xe); UMGEN_DPP_STEP(fmaxf, f32, 0x118, 0xf, 0xc);
    UMGEN_DPP_STEP(fmaxf, f32, 0x142, 0xa, 0xf); UMGEN_DPP_STEP(fmaxf, f32, 0x143, 0xc, 0xf);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ inline int wave_min_i32(int v) {
    UMGEN_DPP_STEP(min, i32, 0x111, 0xf, 0xf); UMGEN_DPP_STEP(min, i32, 0x112, 0xf, 0xf);
    UMGEN_DPP_STEP(min, i32, 0x114, 0xf, 0xe); UMGEN_DPP_STEP(min, i32, 0x118, 0xf, 0xc);
    UMGEN_DPP_STEP(min, i32, 0x142, 0xa, 0xf); UMGEN_DPP_STEP(min, i32, 0x143, 0xc, 0xf);
    return __builtin_amdgcn_readlane(v, 63);
}

// DPP lane exchanges inside a row of 16 lanes (one VALU op each, no LDS crossbar round trip)
__device__ inline float dpp_xor1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true)); }   // quad_perm [1,0,3,2]
__device__ inline float dpp_xor2(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true)); }   // quad_perm [2,3,0,1]
__device__ inline float dpp_half_mirror(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true)); }   // lane i <- lane 7-i (per 8)
__device__ inline float dpp_xor8(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, true)); }   // row_ror:8

// exact GELU (nn.GELU default, module.py:239): 0.5 x (1 + erf(x / sqrt(2)))
__device__ inline float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// GELU of the 16-bit GEMM epilogues: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 absolute: 1 rcp + 1 exp2 + 8 FMA / mul instead
// of erff's ~40 instructions).  The erf-GELU epilogue of the fc GEMM costs a wave 128 evaluations per 256 x 256 tile -- with erff
// that was as long as the tile's MFMAs (fc + GELU 713 vs 942 TFLOP/s without, profiles/r03_gemm_bench_256tile_stagger.txt).  The
// result is rounded to 16 bits right behind it (relative 2^-9 / 2^-12), so the 1.5e-7 are fp32-noise class; fp32 mode keeps erff.
__device__ inline float gelu_fast(float x) {
    const float z = x * 0.70710678118654752440f, az = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
    const float p = fmaf(fmaf(fmaf(fmaf(1.061405429f, t, -1.453152027f), t, 1.421413741f), t, -0.284496736f), t, 0.254829592f) * t;
    const float e = __builtin_amdgcn_exp2f(-az * az * 1.4426950408889634f);
    const float erf_abs = 1.0f - p * e;
    return 0.5f * x * (1.0f + copysignf(erf_abs, z));
}
template <typename T> __device__ inline float gelu_for(float x) { return sizeof(T) == 2 ? gelu_fast(x) : gelu_erf(x); }

// loads 8 consecutive elements as fp32
__device__ inline void load8(const float* p, float (&o)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
__device__ inline void load8(const bf16_t* p, float (&o)[8]) {
    const uint4 a = *reinterpret_cast<const uint4*>(p);
    o[0] = __uint_as_float(a.x << 16); o[1] = __uint_as_float(a.x & 0xffff0000u);
    o[2] = __uint_as_float(a.y << 16); o[3] = __uint_as_float(a.y & 0xffff0000u);
    o[4] = __uint_as_float(a.z << 16); o[5] = __uint_as_float(a.z & 0xffff0000u);
    o[6] = __uint_as_float(a.w << 16); o[7] = __uint_as_float(a.w & 0xffff0000u);
}
__device__ inline void load8(const f16_t* p, float (&o)[8]) {
    const f16x8_t a = *reinterpret_cast<const f16x8_t*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)a[e];
}
__device__ inline void load4(const float* p, float (&o)[4]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
}
__device__ inline void load4(const bf16_t* p, float (&o)[4]) {
    const uint2 a = *reinterpret_cast<const uint2*>(p);
    o[0] = __uint_as_float(a.x << 16); o[1] = __uint_as_float(a.x & 0xffff0000u);
    o[2] = __uint_as_float(a.y << 16); o[3] = __uint_as_float(a.y & 0xffff0000u);
}
__device__ inline void load4(const f16_t* p, float (&o)[4]) {
    typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
    const f16x4_t a = *reinterpret_cast<const f16x4_t*>(p);
    o[0] = (float)a[0]; o[1] = (float)a[1]; o[2] = (float)a[2]; o[3] = (float)a[3];
}
__device__ inline void store4(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ inline void store4(bf16_t* p, const float (&v)[4]) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
    const bf16x4_t o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};   // two v_cvt_pk_bf16_f32
    *reinterpret_cast<uint2*>(p) = __builtin_bit_cast(uint2, o);
}

__device__ inline void store4(f16_t* p, const float (&v)[4]) {
    typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
    const f16x4_t o = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    *reinterpret_cast<uint2*>(p) = __builtin_bit_cast(uint2, o);
}

// ---- build-owned counter-based RNG: bit-for-bit the oracle's rng_u24 (oracle/umgen_oracle.py) ----
__host__ __device__ inline uint32_t rng_u24(uint64_t seed, int frame, int pos, int draw) {
    uint64_t x = seed ^ ((uint64_t)(frame + 1) * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)(pos + 1) * 0xBF58476D1CE4E5B9ull) ^
                 ((uint64_t)(draw + 1) * 0x94D049BB133111EBull);
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return (uint32_t)(x >> 40);
}
__host__ __device__ inline float rng_uniform(uint64_t seed, int frame, int pos, int draw) {
    return (float)rng_u24(seed, frame, pos, draw) * 5.9604644775390625e-08f;  // 2^-24
}
enum { DRAW_MAIN = 0, DRAW_PAD_AVOID = 1, DRAW_CONTROL = 2 };

// exp(x) for x <= 0 that is BIT-IDENTICAL on the device and in the numpy oracle (oracle/umgen_oracle.py: exp_det): the samplers'
// softmax feeds an inverse-CDF walk, where a 1-ulp difference between two libm expf implementations can move a draw across a CDF
// boundary.  2^n * P(f) with t = x log2(e), n = floor(t), f = t - n in [0, 1), P = degree-9 Taylor polynomial of 2^f evaluated by
// Horner with SEPARATELY rounded fp32 multiplies and adds (no FMA contraction); the scaling by 2^n is exact.
__device__ inline float exp_det(float x) {
    if (!(x > -87.0f)) return 0.f;   // underflow, -inf and NaN
    const float t = __fmul_rn(x, 1.44269504088896341f);
    const float n = floorf(t);
    const float f = __fsub_rn(t, n);
    const float c[10] = {1.0f, 0.693147180559945f, 0.240226506959101f, 0.0555041086648216f, 0.00961812910762848f,
                         0.00133335581464284f, 1.54035303933816e-4f, 1.52527338040598e-5f, 1.32154867901443e-6f, 1.01780860092397e-7f};
    float p = c[9];
#pragma unroll
    for (int k = 8; k >= 0; --k) p = __fadd_rn(__fmul_rn(p, f), c[k]);
    return ldexpf(p, (int)n);
}

}  // namespace umgen
