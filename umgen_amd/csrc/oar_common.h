// Helpers shared by the decode engines (oar_engine.hip: XCD-resident, one scene per work item; oar_engine_wide.hip: chip-wide, 2x width): XCD identification, {tag, value} hand-off granules and their bounded polling, wave-uniform-base loads, 16-bit weight
// widening, transposed / whole-wave reductions, matrix-core weight fragments.  Everything here is internal to the two engine files.
#pragma once
#include "frame.h"
#include "kernels.h"

namespace umgen {

namespace {

typedef unsigned long long u64;
typedef unsigned int u32;
typedef u32 u32x4_t __attribute__((ext_vector_type(4)));

constexpr int E = kEngE, H = kEngH, F = 4 * kEngE;
constexpr int NT = kEngThreads, NW = kEngThreads / 64, CU = kEngGroup;
constexpr u32 kSpinLimit = 2000000;   // bounded polls: ~1 s worst case, then the give-up code is published
constexpr float kScaleQK = 0.14433756729740643f;   // float32(1/sqrt(48)), module.py:196-198


__device__ inline u32 xcc_id() {
    u32 x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 15u;
}
__device__ inline u64 gran(u32 tag, float v) { return ((u64)tag << 32) | (u64)__float_as_uint(v); }
// in-group edge: plain store, stays in the XCD's L2 (readers bypass their L1 with sc1 loads)
// (every global access below is `wave-uniform base [32-bit per-lane index]`: the saddr + voffset form needs no 64-bit pointer per
// lane; spilled pointers cost a scratch reload whose s_waitcnt vmcnt(0) also waits for every K/V and weight request in flight)
__device__ inline void put_local(u64* g, u32 i, u32 tag, float v) { __hip_atomic_store(g + i, gran(tag, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// cross-group edge: write-through
__device__ inline void put_far(u64* g, u32 i, u32 tag, float v) { __hip_atomic_store(g + i, gran(tag, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline u64 get(const u64* g, u32 i) { return __hip_atomic_load(g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct Ctx {
    u32* err;
    bool failed;
};

// Workgroup barrier for LDS hand-offs only.  __syncthreads() carries a release fence, and on gfx950 loads and stores share vmcnt: the
// fence becomes s_waitcnt vmcnt(0), i.e. EVERY barrier drains the weight requests in flight -- harmless while they arrive during a
// group's idle wait, but on an item that has none (a launch's first item; every item when a group runs a whole scene) the first
// barrier of P1 waited for the whole 14 MB.  Nothing here needs global-memory ordering at a barrier: granules are self-validating
// (tag + value in one 8-byte store) and the K/V rows written are read by later launches.
__device__ inline void wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Polling with TWO requests of every granule in flight, half a round trip apart.  With one (round 2) a poll that just misses the
// producer's store costs a whole further round trip (L2: ~0.7 us, another XCD: ~1.5 us), on average half of one per hand-off and five
// hand-offs per layer; a wave's loads return in order, so `check(older)` waits for the older request only (s_waitcnt vmcnt(PER)) and
// the next one leaves as soon as it is back: the initial stagger sustains itself.  Loads are unconditional (granules already
// received are simply requested again) so that the loop is straight-line code and the wait counts are exact.
#ifndef UMGEN_ENG_POLL2
#define UMGEN_ENG_POLL2 0
#endif
#ifndef UMGEN_ENG_POLL_STAGGER
#define UMGEN_ENG_POLL_STAGGER 6      // s_sleep units of 64 clocks between the first two requests
#endif
// slot k of thread tid (bit k of need) waits for granule idx(k) and writes its value to dst[tid + k * NT]
template <int PER, typename IDX>
__device__ inline void poll_granules(Ctx& c, int tid, const u64* g, u32 need, IDX idx, u32 tag, float* dst) {
    if (c.failed || !__any(need != 0u)) return;
    u32 got = 0;
    u32 ix[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) ix[k] = idx(k);
    u64 va[PER], vb[PER];
    auto issue = [&](u64 (&v)[PER]) {
#pragma unroll
        for (int k = 0; k < PER; ++k) v[k] = get(g, ix[k]);
    };
    auto check = [&](const u64 (&v)[PER]) {
#pragma unroll
        for (int k = 0; k < PER; ++k)
            if ((((need & ~got) >> k) & 1u) && (u32)(v[k] >> 32) == tag) { dst[tid + k * NT] = __uint_as_float((u32)v[k]); got |= 1u << k; }
        return !__any(got != need);
    };
    issue(va);
    if (UMGEN_ENG_POLL2) __builtin_amdgcn_s_sleep(UMGEN_ENG_POLL_STAGGER);
    for (u32 spins = 0;;) {
        if (UMGEN_ENG_POLL2) {
            issue(vb);
            if (check(va)) break;
            issue(va);
            if (check(vb)) break;
        } else {
            if (check(va)) break;
            issue(va);
        }
        if (++spins > kSpinLimit) { if ((tid & 63) == 0) atomicExch(c.err, tag | 0x80000000u); c.failed = true; break; }
        if ((spins & 255u) == 0 && __hip_atomic_load(c.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { c.failed = true; break; }
    }
}

// the workgroup gathers granules [0, n) of g into dst[0, n)
template <int PER>
__device__ inline void gather(Ctx& c, int tid, const u64* g, int n, u32 tag, float* dst) {
    u32 need = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k)
        if (tid + k * NT < n) need |= 1u << k;
    poll_granules<PER>(c, tid, g, need, [&](int k) { return (u32)min(tid + k * NT, n - 1); }, tag, dst);
    wg_barrier();
}

// Pointers read out of the layer table are generic to the compiler (flat loads): cast them to the global address space.
#define UMGEN_GLOBAL __attribute__((address_space(1)))
// wave-uniform base + 32-bit per-lane element offset (global_load ... saddr form: one VGPR of address per load)
__device__ inline u32x4_t ldwu(const bf16_t* ubase, u32 off) {
    return __builtin_nontemporal_load((const UMGEN_GLOBAL u32x4_t*)(ubase + off));
}
// the same with the default cache policy: rows that are read again from this XCD's L2 (the systolic schedule's q|k|v rows)
__device__ inline u32x4_t ldwk(const bf16_t* ubase, u32 off) { return *(const UMGEN_GLOBAL u32x4_t*)(ubase + off); }
__device__ inline float ldg(const float* p) { return *(const UMGEN_GLOBAL float*)p; }
__device__ inline void ldg8(const float* p, float (&o)[8]) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v x = *(const UMGEN_GLOBAL f4v*)p;
    const f4v y = *(const UMGEN_GLOBAL f4v*)(p + 4);
    o[0] = x.x; o[1] = x.y; o[2] = x.z; o[3] = x.w; o[4] = y.x; o[5] = y.y; o[6] = y.z; o[7] = y.w;
}
// 8 16-bit weights (TT = bf16_t: raw bfloat16 bits widened by a shift / mask; TT = f16_t: IEEE half through v_cvt_f32_f16) x 8 fp32
// activations on the packed fp32 FMA (v_pk_fma_f32: two MACs per instruction): the even / odd elements accumulate in the two
// halves of acc
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <typename TT> __device__ inline f32x2_t up2(u32 w);
template <> __device__ inline f32x2_t up2<bf16_t>(u32 w) { return f32x2_t{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)}; }
template <> __device__ inline f32x2_t up2<f16_t>(u32 w) {
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    const f16x2_t h = __builtin_bit_cast(f16x2_t, w);
    return f32x2_t{(float)h.x, (float)h.y};
}
template <typename TT>
__device__ inline void unpack8(const u32x4_t& w, float (&o)[8]) {
    const f32x2_t a = up2<TT>(w.x), b = up2<TT>(w.y), c = up2<TT>(w.z), d = up2<TT>(w.w);
    o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y; o[4] = c.x; o[5] = c.y; o[6] = d.x; o[7] = d.y;
}
// one weight pair x one activation pair: bf16 -> two widening instructions + one packed FMA (3 per 2 MACs); IEEE half -> two
// v_fma_mix_f32 (the f16 -> f32 conversion is part of the FMA: 2 per 2 MACs).  The same fp32 FMAs in the same order either way.
template <typename TT> __device__ inline f32x2_t mac2(u32 w, f32x2_t x, f32x2_t acc);
template <> __device__ inline f32x2_t mac2<bf16_t>(u32 w, f32x2_t x, f32x2_t acc) { return __builtin_elementwise_fma(up2<bf16_t>(w), x, acc); }
template <> __device__ inline f32x2_t mac2<f16_t>(u32 w, f32x2_t x, f32x2_t acc) {
    // (the compiler does not form the mix instruction from fma(fpext(half), ..) on this target: written out)
    f32x2_t d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d.x) : "v"(w), "v"(x.x), "v"(acc.x));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d.y) : "v"(w), "v"(x.y), "v"(acc.y));
    return d;
}
template <typename TT>
__device__ inline f32x2_t dot8(const u32x4_t& w, const f32x2_t (&x)[4], f32x2_t acc) {
    acc = mac2<TT>(w.x, x[0], acc);
    acc = mac2<TT>(w.y, x[1], acc);
    acc = mac2<TT>(w.z, x[2], acc);
    acc = mac2<TT>(w.w, x[3], acc);
    return acc;
}
// value as the 16-bit K/V cache will hold it, and its raw bits
template <typename TT> __device__ inline float round16(float v) { return Cvt<TT>::to_f(Cvt<TT>::from_f(v)); }
template <typename TT> __device__ inline bf16_t bits16(float v) { return __builtin_bit_cast(bf16_t, Cvt<TT>::from_f(v)); }
__device__ inline void load8p(const float* p, f32x2_t (&o)[4]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    o[0] = f32x2_t{a.x, a.y}; o[1] = f32x2_t{a.z, a.w}; o[2] = f32x2_t{b.x, b.y}; o[3] = f32x2_t{b.z, b.w};
}

// Attention lane mapping: LPK lanes per key, KPW keys per wave pass; a lane holds 12 of a key's 48 values: 16 bytes + 8 bytes
#ifndef UMGEN_ENG_KP
#define UMGEN_ENG_KP 1
#endif
#ifndef UMGEN_ENG_NBM
#define UMGEN_ENG_NBM 2        // matrix-core attention: register buffers of 32 keys (28 VGPRs each)
#endif
#ifndef UMGEN_ENG_NB
// measured (profiles/r03_engine_experiments.txt): VALU row products 2: 476 us per launch, 3: 476, 4: 471, 5 (15 spilled VGPRs): 504; with the c_fc
// rows as matrix-core fragments (aligned register tuples) 4 buffers spill 8 VGPRs (462 us), 3 do not (443)
#define UMGEN_ENG_NB ((UMGEN_ENG_MFMA & 4) ? 3 : 4)
#endif
#ifndef UMGEN_ENG_NB_SYS
#define UMGEN_ENG_NB_SYS 3    // the systolic kernel keeps more of a layer live: 4 buffers spill 2-4 VGPRs there
#endif
constexpr int LPK = 4, KPW = 64 / LPK;
typedef u32 u32x2_t __attribute__((ext_vector_type(2)));
struct KVPiece {
    u32x4_t a;   // dimensions 8 piece .. 8 piece + 7
    u32x2_t b;   // dimensions 32 + 4 piece .. 32 + 4 piece + 3
};
__device__ inline u32x2_t ldwu2(const bf16_t* ubase, u32 off) {
    return __builtin_nontemporal_load((const UMGEN_GLOBAL u32x2_t*)(ubase + off));
}
template <typename TT>
__device__ inline void unpack12(const KVPiece& w, f32x2_t (&o)[6]) {
    o[0] = up2<TT>(w.a.x); o[1] = up2<TT>(w.a.y); o[2] = up2<TT>(w.a.z); o[3] = up2<TT>(w.a.w); o[4] = up2<TT>(w.b.x); o[5] = up2<TT>(w.b.y);
}

template <int CTRL> __device__ inline float dpp_mov(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true)); }
__device__ inline float sum_rows16(float v) {   // lane-wise sum of the wave's four 16-lane rows, in every lane
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// wave total in every lane: four fused DPP adds inside the 16-lane rows (quad_perm x 2, row_half_mirror, row_mirror), then the four
// rows through the permlane swaps -- 10 instructions and no readlane, against 6 DPP steps of mov + add + wait states (common.h wave_sum)
__device__ inline float wave_max_all(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));       // (bound_ctrl zero fill never applies: these controls are permutations)
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ inline float wave_sum_all(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return sum_rows16(v);
}

template <int NTILE> struct WFrags { u32x4_t f[NTILE][3]; };
template <int NTILE, bool KEEP, int T0 = 0, int T1 = NTILE>
__device__ inline void req_frags(WFrags<NTILE>& w, const bf16_t* W, int row0, int nvalid, int wave, int lane) {
    const bf16_t* base = W + (long)row0 * E + 96 * wave;
#pragma unroll
    for (int t = T0; t < T1; ++t) {
        const u32 ro = (u32)min(16 * t + (lane & 15), nvalid - 1) * (u32)E + (u32)(lane >> 4) * 8u;
#pragma unroll
        for (int j = 0; j < 3; ++j) w.f[t][j] = KEEP ? ldwk(base, ro + 32u * j) : ldwu(base, ro + 32u * j);
    }
}
// the same fragments out of a repacked copy [..][NTILE x 3 fragments][64 lanes][8]: 1 KB contiguous per request
template <int NTILE, bool KEEP, int T0 = 0, int T1 = NTILE>
__device__ inline void req_frags_packed(WFrags<NTILE>& w, const bf16_t* P, int lane) {
#pragma unroll
    for (int t = T0; t < T1; ++t)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const u32 off = (u32)((3 * t + j) * 64 + lane) * 8u;
            w.f[t][j] = KEEP ? ldwk(P, off) : ldwu(P, off);
        }
}
// value -> (hi, lo) in the operand type, as raw 16-bit patterns
template <typename TT>
__device__ inline void split16(float v, unsigned short& hi, unsigned short& lo) {
    const auto h = Cvt<TT>::from_f(v);
    hi = __builtin_bit_cast(unsigned short, h);
    lo = __builtin_bit_cast(unsigned short, Cvt<TT>::from_f(v - Cvt<TT>::to_f(h)));
}
// Granule i of a WAVE-UNIFORM base as `saddr + 32-bit byte offset`: one VGPR of address per request in flight (a gather of this engine keeps
// 12 requests per lane in flight; as 64-bit flat pointers -- what the compiler makes of an agent-scope atomic load -- their addresses alone
// were 24 VGPRs and pushed 12 of the resident c_fc fragments out to scratch memory).  Written as inline assembly, so the requests are
// invisible to the compiler's wait-count bookkeeping: poll_wait() is the explicit s_waitcnt every consumer goes through (a wave's loads
// return in order: vmcnt(0) is exactly what waiting for the youngest request means).
__device__ inline const u64* uniform_ptr(const u64* g) {      // (the base IS wave-uniform: pinned to scalar registers for the "s" operand below)
    const u64 b = reinterpret_cast<u64>(g);
    const u32 lo = __builtin_amdgcn_readfirstlane((u32)b), hi = __builtin_amdgcn_readfirstlane((u32)(b >> 32));
    return reinterpret_cast<const u64*>(((u64)hi << 32) | (u64)lo);
}
__device__ inline void poll_issue(u64& v, const u64* g, u32 i) {
    asm volatile("global_load_dwordx2 %0, %1, %2 sc1" : "=v"(v) : "v"(i << 3), "s"(g) : "memory");
}
template <int PER>
__device__ inline void poll_wait(u64 (&v)[PER]) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < PER; ++k) asm volatile("" : "+v"(v[k]));     // (the values are defined behind the wait, not behind the request)
}
// slot k of thread tid (bit k of need) waits for granule idx(k) with `tag` and hands its value to sink(k, value).  Round 1 requests every
// slot; while something is missing a lane asks for ONE of its missing granules per round with a short sleep in between (the gathers
// of this engine are up to 12 granules per thread and 32 CUs: polling all of them would put 1.5 MB per round on the L2 / the fabric
// while a group waits for its predecessor), then requests all its missing slots again.
// ALL: every round requests every slot again (one round trip behind the producers; for the hand-offs at which the whole group
// is waiting anyway -- nobody's K/V stream shares the L2 with the polls).
template <int PER, bool ALL = false, typename IDX, typename SINK>
__device__ inline void poll_ms(Ctx& c, int tid, const u64* g, u32 need, IDX idx, u32 tag, SINK sink) {
    if (c.failed || !__any(need != 0u)) return;
    g = uniform_ptr(g);
    u32 got = 0;
    for (u32 spins = 0;;) {
        u64 v[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) poll_issue(v[k], g, idx(k));      // (unconditional: slots that are not needed re-read a valid index)
        poll_wait<PER>(v);
#pragma unroll
        for (int k = 0; k < PER; ++k)
            if ((((need & ~got) >> k) & 1u) && (u32)(v[k] >> 32) == tag) { sink(k, __uint_as_float((u32)v[k])); got |= 1u << k; }
        if (!__any(got != need)) break;
        if (ALL) {
            if (++spins > kSpinLimit / 8) { if ((tid & 63) == 0) atomicExch(c.err, tag | 0x80000000u); c.failed = true; break; }
            if ((spins & 63u) == 0 && __hip_atomic_load(c.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { c.failed = true; break; }
            continue;
        }
        // one missing granule per lane until it is there (lanes that have everything re-read a slot of theirs)
        const u32 miss = need & ~got;
        const u32 i1 = idx(miss ? __ffs((int)miss) - 1 : 0);
        for (;;) {
            __builtin_amdgcn_s_sleep(2);
            u64 v1[1];
            poll_issue(v1[0], g, i1);
            poll_wait<1>(v1);
            if (!__any(miss != 0u && (u32)(v1[0] >> 32) != tag)) break;
            if (++spins > kSpinLimit) { if ((tid & 63) == 0) atomicExch(c.err, tag | 0x80000000u); c.failed = true; break; }
            if ((spins & 255u) == 0 && __hip_atomic_load(c.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { c.failed = true; break; }
        }
        if (c.failed) break;
    }
}

// The same hand-off with TWO requests of every slot in flight, `stagger` x 64 clocks apart (half a fabric round trip): a request that just misses the
// producer's store is followed by one half a round trip behind it instead of a whole one.  A wave's loads return in order: s_waitcnt vmcnt(PER) is "the
// older set is back".  Every round requests every slot (exact counts); both sets are drained before the registers are given back.
template <int PER, typename IDX, typename SINK>
__device__ inline void poll_stag(Ctx& c, int tid, const u64* g, u32 need, IDX idx, u32 tag, int stagger, SINK sink) {
    if (c.failed || !__any(need != 0u)) return;
    g = uniform_ptr(g);
    u32 got = 0;
    u64 va[PER], vb[PER];
    auto issue = [&](u64 (&v)[PER]) {
#pragma unroll
        for (int k = 0; k < PER; ++k) poll_issue(v[k], g, idx(k));
    };
    auto older_back = [&](u64 (&v)[PER]) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
#pragma unroll
        for (int k = 0; k < PER; ++k) asm volatile("" : "+v"(v[k]));
    };
    auto check = [&](const u64 (&v)[PER]) {
#pragma unroll
        for (int k = 0; k < PER; ++k)
            if ((((need & ~got) >> k) & 1u) && (u32)(v[k] >> 32) == tag) { sink(k, __uint_as_float((u32)v[k])); got |= 1u << k; }
        return !__any(got != need);
    };
    issue(va);
    for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(1);
    for (u32 spins = 0;;) {
        issue(vb);
        older_back(va);
        if (check(va)) break;
        issue(va);
        older_back(vb);
        if (check(vb)) break;
        if (++spins > kSpinLimit / 8) { if ((tid & 63) == 0) atomicExch(c.err, tag | 0x80000000u); c.failed = true; break; }
        if ((spins & 63u) == 0 && __hip_atomic_load(c.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { c.failed = true; break; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < PER; ++k) { asm volatile("" : "+v"(va[k])); asm volatile("" : "+v"(vb[k])); }
}

}  // namespace

}  // namespace umgen
