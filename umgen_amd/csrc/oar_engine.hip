// XCD-resident OAR decode engine: the 36 BlockOAR layers of one decode step (module.py:378-428) in ONE launch.
//
// Why this shape (measured on MI355X, profiles/r02_seam_bench_*.txt): a decode layer at one scene is 14 MB of weights and five
// all-to-all hand-offs (x -> q|k|v -> attention -> x' -> h -> x'').  As five launches the layer costs 22 us, of which ~9 us are
// kernel boundaries; as one chip-wide persistent kernel the hand-offs alone cost 12-17 us per layer, because every hop between
// XCDs pays the fabric twice (write-through store, L2-missing load).  Inside ONE XCD the L2 is coherent for its 32 CUs: a plain
// 8-byte {tag, value} store lands in the L2 and an sc1 load (L1 bypass) reads it back in 0.8-1.7 us per edge.  So:
//   * a GROUP = the 32 workgroups (one per CU, 512 threads) that landed on one XCD; a work item = (scene, layer); all five
//     hand-offs of an item stay inside the group, in group-private granule buffers;
//   * layers are dealt round-robin over the D groups that serve a scene (layer l -> group l % D): only the 768-float x vector
//     crosses the fabric between layers (sc1 store + sc1 load, ~2.8 us), and while the other D-1 groups work, a group's loads
//     for its next layer are already in flight (weights are requested one phase ahead into registers, the first phase's right
//     after the previous item);
//   * several scenes: scene s of a round owns groups [s*D, s*D + D) -- with 8 scenes every XCD runs a whole scene.
// Workgroups find their XCD with s_getreg HW_REG_XCC_ID and take a rank from a per-group ticket; the host has checked with a
// census launch (oar_engine_census) that the stream's CUs give exactly 32 workgroups on each of NG XCDs -- otherwise the engine
// is not used and the decode step runs as the five-launch form (gemv.hip).  Every poll is bounded; a give-up is reported
// through OarEngineArgs::err and fails the frame loudly.
//
// The MLP is split by HIDDEN UNITS: CU c owns c_fc rows 96c..96c+95 and the matching 96 columns of the mlp c_proj, so gelu(c_fc)
// never leaves the CU; what is exchanged are the CUs' partial sums of the 768 outputs (each CU then adds the 32 partials of its
// own 24 rows): a 768-granule gather instead of a 3072-granule one, and the projection needs no cross-lane reduction.
//
// Arithmetic (fixed, independent of B / D / group placement, so scenes are batch-invariant): fp32 activations, bf16 weights and
// bf16 K/V cache, fp32 accumulation.  Row dot products: lane l owns k = 8l..8l+7 (+512 i), 8 sequential FMAs per chunk, DPP
// wave sum.  Attention of a head: its keys are split in two halves (two CUs), each half in 8 wave spans, each span in groups
// of 8 lanes per key with an online softmax per lane group; the 64 group partials of a half, then the two halves, are merged in
// a fixed order.
#include "frame.h"
#include "kernels.h"

namespace umgen {

namespace {

typedef unsigned long long u64;
typedef unsigned int u32;
typedef u32 u32x4_t __attribute__((ext_vector_type(4)));

constexpr int E = kEngE, H = kEngH, F = 4 * kEngE;
constexpr int NT = kEngThreads, NW = kEngThreads / 64, CU = kEngGroup;
constexpr int RQ = 3 * E / CU / NW;   // 9 q|k|v rows per wave
constexpr int RO = E / CU / NW;       // 3 c_proj rows per wave
constexpr int RF = F / CU / NW;       // 12 c_fc rows per wave
constexpr int RP = E / CU / NW;       // 3 mlp c_proj rows per wave
static_assert(RQ * NW * CU == 3 * E && RO * NW * CU == E && RF * NW * CU == F, "row partition");
constexpr u32 kSpinLimit = 2000000;   // bounded polls: ~1 s worst case, then the give-up code is published
constexpr float kScaleQK = 0.14433756729740643f;   // float32(1/sqrt(48)), module.py:196-198

// LDS carve (floats)
constexpr int L_XS = 0;                    // x of the item (kept until the attention projection's residual)   [768]
constexpr int L_XB = L_XS + E;             // x' (kept until the MLP projection's residual)                    [768]
constexpr int L_AS = L_XB + E;             // merged attention output                                            [768]
constexpr int L_HS = L_AS + E;             // this CU's 96 gelu(c_fc) values [96] | half-row sums [256] | gathered mlp partial sums [32][24]   [1152]
constexpr int L_QKV = L_HS + 1152;          // q_h | k_h | v_h of this CU's head                                  [144 -> 160]
constexpr int L_GP = L_QKV + 160;          // gathered half partials [32][50]                                   [1600]
constexpr int L_SM = L_GP + 2 * H * 50;    // per lane-group m [64], l [64], weights [64]                        [192]
constexpr int L_SO = L_SM + 192;           // per lane-group o [64][48]                                          [3072]
constexpr int L_MISC = L_SO + 64 * 48;     // rank / scratch                                                      [16]
constexpr int L_LN = L_MISC + 16;          // ln_1 | ln_2 weights of the item                                     [1536]
constexpr int L_W2 = L_LN + 2 * E;         // parked mlp c_proj units 0..11 of every thread: [12][512] x 16 B      [24576]
// K/V staging of the attention: UMGEN_ENG_LDS_KEYS more keys per wave in flight than the two register buffers hold, brought in by
// LDS-DMA (global_load_lds: no VGPRs) while the group waits for x.  Private to each wave: [K rows 16 x 96 B | V rows 16 x 96 B]
// MEASURED (profiles/r03_engine_experiments.txt): attention 3.67 vs 3.68 us per item, 561 vs 563 us per launch -- the phase is not
// waiting for K/V (its ~100 VALU instructions per 16-key pass and their DPP / exp dependency chains are what it costs).  Off.
#ifndef UMGEN_ENG_LDS_KEYS
#define UMGEN_ENG_LDS_KEYS 0
#endif
constexpr int kStageKeys = UMGEN_ENG_LDS_KEYS;           // 0 or 16
static_assert(kStageKeys == 0 || kStageKeys == 16, "staging is written for 16 keys (two 8-key passes)");
constexpr int L_KV = L_W2 + 12 * NT * 4;                 // [8 waves][768 floats = 3 KB]
constexpr int L_TOTAL = L_KV + (kStageKeys ? NW * 768 : 0);
static_assert(L_TOTAL * 4 <= 160 * 1024, "LDS budget");

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

// the workgroup gathers granules [0, n) of g into dst[0, n)
template <int PER>
__device__ inline void gather(Ctx& c, int tid, const u64* g, int n, u32 tag, float* dst) {
    u32 got = 0, need = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k)
        if (tid + k * NT < n) need |= 1u << k;
    if (!c.failed) {
        for (u32 spins = 0;;) {
            u64 v[PER];
#pragma unroll
            for (int k = 0; k < PER; ++k)
                if (((need & ~got) >> k) & 1u) v[k] = get(g, (u32)tid + (u32)(k * NT));
#pragma unroll
            for (int k = 0; k < PER; ++k)
                if (((need & ~got) >> k) & 1u) {
                    if ((u32)(v[k] >> 32) == tag) { dst[tid + k * NT] = __uint_as_float((u32)v[k]); got |= 1u << k; }
                }
            if (!__any(got != need)) break;
            if (++spins > kSpinLimit) { if ((tid & 63) == 0) atomicExch(c.err, tag | 0x80000000u); c.failed = true; break; }
            if ((spins & 255u) == 0 && __hip_atomic_load(c.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { c.failed = true; break; }
        }
    }
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
template <typename TT>
__device__ inline f32x2_t dot8(const u32x4_t& w, const f32x2_t (&x)[4], f32x2_t acc) {
    acc = __builtin_elementwise_fma(up2<TT>(w.x), x[0], acc);
    acc = __builtin_elementwise_fma(up2<TT>(w.y), x[1], acc);
    acc = __builtin_elementwise_fma(up2<TT>(w.z), x[2], acc);
    acc = __builtin_elementwise_fma(up2<TT>(w.w), x[3], acc);
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

// R rows of a [N][768] matrix held by one wave: chunk a[r] = k 8l..8l+7 of row r; the 256 tail columns of rows (2j, 2j+1) are
// shared by the two half-waves: lanes 0-31 hold row 2j's, lanes 32-63 row 2j+1's, k = 512 + 8 (l & 31)
template <int R>
struct Rows768 {
    u32x4_t a[R];
    u32x4_t b[(R + 1) / 2];
};
template <int R, bool KEEP = false>
__device__ inline void req768(Rows768<R>& w, const bf16_t* W, int row0, int lane) {
#pragma unroll
    for (int r = 0; r < R; ++r) w.a[r] = KEEP ? ldwk(W + (long)(row0 + r) * E, (u32)lane * 8u) : ldwu(W + (long)(row0 + r) * E, (u32)lane * 8u);
#pragma unroll
    for (int j = 0; j < (R + 1) / 2; ++j) {
        const bool both = 2 * j + 1 < R;   // odd R: the upper half-wave re-reads the last row's tail, its copy is ignored
        const bf16_t* base = W + (long)(row0 + 2 * j) * E + 512;
        const u32 off = (u32)(lane & 31) * 8u + (both ? (u32)(lane >> 5) * (u32)E : 0u);
        w.b[j] = KEEP ? ldwk(base, off) : ldwu(base, off);
    }
}
// the same for rows that are not consecutive in W: row r of the wave is W row rowof(r)
template <int R, bool KEEP, typename F>
__device__ inline void req768_rows(Rows768<R>& w, const bf16_t* W, F rowof, int lane) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const bf16_t* base = W + (long)rowof(r) * E;
        w.a[r] = KEEP ? ldwk(base, (u32)lane * 8u) : ldwu(base, (u32)lane * 8u);
    }
#pragma unroll
    for (int j = 0; j < (R + 1) / 2; ++j) {
        const int r1 = 2 * j + 1 < R ? 2 * j + 1 : 2 * j;     // odd R: the upper half-wave re-reads the last row's tail, its copy is ignored
        const bf16_t* base = W + (long)(lane < 32 ? rowof(2 * j) : rowof(r1)) * E + 512;
        const u32 off = (u32)(lane & 31) * 8u;
        w.b[j] = KEEP ? *(const UMGEN_GLOBAL u32x4_t*)(base + off) : __builtin_nontemporal_load((const UMGEN_GLOBAL u32x4_t*)(base + off));
    }
}
// dot products of rows [R0, R1) only (a pair's shared tail chunk is multiplied by whichever range needs one of its rows)
template <typename TT, int R, int R0, int R1>
__device__ inline void dot768_range(const Rows768<R>& w, const f32x2_t (&x1)[4], const f32x2_t (&x2)[4], int lane, float (&out)[R1 - R0]) {
    const f32x2_t zero = {0.f, 0.f};
    f32x2_t acc[R1 - R0];
#pragma unroll
    for (int r = R0; r < R1; ++r) acc[r - R0] = dot8<TT>(w.a[r], x1, zero);
#pragma unroll
    for (int j = R0 / 2; j < (R1 + 1) / 2; ++j) {
        const f32x2_t p = dot8<TT>(w.b[j], x2, zero);
        if (2 * j >= R0 && 2 * j < R1) acc[2 * j - R0] += (lane < 32) ? p : zero;
        if (2 * j + 1 >= R0 && 2 * j + 1 < R1) acc[2 * j + 1 - R0] += (lane >= 32) ? p : zero;
    }
#pragma unroll
    for (int r = 0; r < R1 - R0; ++r) out[r] = wave_sum(acc[r].x + acc[r].y);
}

template <typename TT, int R>
__device__ inline void dot768(const Rows768<R>& w, const f32x2_t (&x1)[4], const f32x2_t (&x2)[4], int lane, float (&out)[R]) {
    const f32x2_t zero = {0.f, 0.f};
    f32x2_t acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = dot8<TT>(w.a[r], x1, zero);
#pragma unroll
    for (int j = 0; j < (R + 1) / 2; ++j) {
        const f32x2_t p = dot8<TT>(w.b[j], x2, zero);
        acc[2 * j] += (lane < 32) ? p : zero;
        if (2 * j + 1 < R) acc[2 * j + 1] += (lane >= 32) ? p : zero;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) out[r] = wave_sum(acc[r].x + acc[r].y);
}

// LayerNorm (weight only, eps 1e-5, module.py:26-37) of the 768-vector in LDS, in the lane's dot-product layout
__device__ inline void ln768(const float* xs, const float* lnw, int lane, f32x2_t (&x1)[4], f32x2_t (&x2)[4]) {
    f32x2_t l1[4], l2[4];
    load8p(lnw + lane * 8, l1);
    load8p(lnw + 512 + (lane & 31) * 8, l2);
    load8p(xs + lane * 8, x1);
    load8p(xs + 512 + (lane & 31) * 8, x2);
    f32x2_t s1 = (x1[0] + x1[1]) + (x1[2] + x1[3]);
    f32x2_t s2 = (x2[0] + x2[1]) + (x2[2] + x2[3]);
    float s = s1.x + s1.y;
    s += (lane < 32) ? (s2.x + s2.y) : 0.f;
    const float mean = wave_sum(s) / (float)E;
    const f32x2_t mean2 = {mean, mean};
    f32x2_t q1 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) { const f32x2_t d = x1[e] - mean2; q1 = __builtin_elementwise_fma(d, d, q1); }
#pragma unroll
    for (int e = 0; e < 4; ++e) { const f32x2_t d = x2[e] - mean2; q2 = __builtin_elementwise_fma(d, d, q2); }
    float q = q1.x + q1.y;
    q += (lane < 32) ? (q2.x + q2.y) : 0.f;
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)E + 1e-5f);
    const f32x2_t rstd2 = {rstd, rstd};
#pragma unroll
    for (int e = 0; e < 4; ++e) { x1[e] = (x1[e] - mean2) * rstd2 * l1[e]; x2[e] = (x2[e] - mean2) * rstd2 * l2[e]; }
}


}  // namespace

// systolic schedule: which matrices stay in registers over the scenes of a layer (the others are requested again by every item)
// (measured with -Rpass-analysis=kernel-resource-usage: the 512-thread kernel sits at 254 of 256 VGPRs; keeping the c_proj rows costs 8
//  spilled VGPRs, the c_fc rows 58, both 77 -- the rows the compiler cannot hold are then reloaded from scratch memory every item)
#ifndef UMGEN_SYS_KEEP_WO
#define UMGEN_SYS_KEEP_WO 0
#endif
#ifndef UMGEN_SYS_KEEP_WF
#define UMGEN_SYS_KEEP_WF 0
#endif
constexpr bool kSysKeepWo = UMGEN_SYS_KEEP_WO, kSysKeepWf = UMGEN_SYS_KEEP_WF;
// q rows first: every wave owns 3 q, 3 k and 3 v rows (instead of 9 consecutive rows of the packed c_attn matrix), computes and
// publishes its q rows, THEN its k | v rows: the latency of the q hand-off (one L2 round trip, 1.1 us) runs beside the k | v row
// products instead of behind all nine, and the new token's own k | v -- only one more key of the softmax -- is merged after the
// cached keys.  0: round-2 order (nine consecutive rows, one hand-off of q | k | v, the new key inside the key spans)
// MEASURED (profiles/r03_engine_experiments.txt): wait q 1.14 -> 0.75 us, but the own key's extra poll + pass puts the head's second
// half 1 us behind the first (wait partials 1.09 -> 2.05 us): 603 vs 565 us per launch.  Kept as a build option, off.
#ifndef UMGEN_ENG_QFIRST
#define UMGEN_ENG_QFIRST 0
#endif
constexpr bool kQFirst = UMGEN_ENG_QFIRST;
#ifndef UMGEN_ENG_STAGGER_US
#define UMGEN_ENG_STAGGER_US 10
#endif
constexpr int kStaggerTicks = UMGEN_ENG_STAGGER_US * 100;   // wall_clock64 ticks (100 MHz)

// STAMPS: per-phase 100 MHz time stamps of (group 0, rank 0) into OarEngineArgs::stamps (UMGEN_DEBUG_TIMING); compiled out otherwise
// SYS (systolic schedule, several scenes): group g keeps layers g, g + 8, ... RESIDENT -- their weights are requested once per step
// and layer, into the same registers / parked LDS rows -- and the B scenes of the batch flow through the eight groups one behind
// the other (item (layer l, scene s) on group l % 8 needs x of (l - 1, s) from group (l - 1) % 8).  A step costs
// (n_layers + B - 1) item times instead of B x n_layers / 8 x (item + weight stream): with one scene per XCD (round 2) every
// group streamed all 36 layers, 8x the algorithmic weight traffic through the fabric, and waited 11 of 29 us per item for it.
template <bool STAMPS, typename TT, bool SYS>
__global__ __launch_bounds__(kEngThreads) void oar_engine_kernel(OarEngineArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid0 = threadIdx.x;
    const unsigned long long t_k0 = wall_clock64();   // 100 MHz
    // ---- who am I: group (XCD) and rank inside it ----
    const u32 xcc = xcc_id();
    const int g = a.xcc_group[xcc];
    if (g >= a.NG) return;   // (the census guarantees this never happens)
    if (tid0 == 0) reinterpret_cast<u32*>(lds + L_MISC)[0] = atomicAdd(a.ticket + g, 1u) & (u32)(CU - 1);
    wg_barrier();
    const int w0 = __builtin_amdgcn_readfirstlane((int)reinterpret_cast<u32*>(lds + L_MISC)[0]);
    Ctx c{a.err, false};
    const bool timer = STAMPS && a.stamps != nullptr && g == 0 && w0 == 0 && tid0 == 0;
    unsigned long long t_prev = 0;
    unsigned long long t_entry = (STAMPS && timer) ? wall_clock64() : 0ull;
    if (STAMPS && timer) a.stamps[12] += t_entry - t_k0;
    bool first_item = true;
    auto stamp = [&](int p) {
        if (STAMPS && timer) {
            const unsigned long long t = wall_clock64();
            if (p >= 0) a.stamps[p] += t - t_prev;
            if (first_item) {
                if (p == 0) { a.stamps[14] += t - t_k0; a.stamps[15] += 1; first_item = false; }   // kernel entry -> the first item's P1 can start
            }
            t_prev = t;
        }
    };
    const int Lk = a.st->step;              // cached keys before this step == position of the new token
    const u32 ep = a.st->epoch;
    if (STAMPS && timer) {
        asm volatile("" ::"v"(Lk + (int)ep));   // (both loads have returned)
        const unsigned long long t = wall_clock64();
        a.stamps[12] += t - t_entry;
        t_entry = t;
    }
    const int R = SYS ? 1 : a.R, D = SYS ? a.NG : a.D;
    const int rounds = (a.B + R - 1) / R;
    const int pipe = g / D, q = g % D;      // pipeline (scene slot of the round) and position in it
    float* xs = lds + L_XS;
    float* xb = lds + L_XB;
    float* as = lds + L_AS;
    // the weight rows of a wave (requested at the start of an item).  SYS: c_proj / c_fc rows and the parked mlp rows at the first
    // scene of a layer only -- they stay in their registers / LDS rows for the other scenes; the q|k|v rows (56 VGPRs, dead after P1)
    // do not fit beside them through the attention (142 spilled VGPRs when kept), so every item requests them again, with the
    // default cache policy: after the layer's first scene they come out of this XCD's L2
    Rows768<RO> wo;
    Rows768<RF> wf;

    // items of this group in the order it works through them: (round rd, layer l) -- scene rd * R + pipe.
    //   !SYS: rounds outside, this pipeline's layers (q, q + D, ...) inside;  SYS: layers outside, every scene of the batch inside
    const int n_lay = (a.n_layers - q + D - 1) / D;
    const int n_items = n_lay > 0 ? n_lay * rounds : 0;
    for (int item = 0; item < n_items; ++item) {
        const int rd = SYS ? item % rounds : item / n_lay;
        const int l = q + D * (SYS ? item / rounds : item % n_lay);
        const int s = rd * R + pipe;
        const bool load_w = !SYS || rd == 0;          // this item requests the layer's weights
        if (s >= a.B) continue;
        {
            // Launch-time stagger: every group would request its first layer's 14 MB at kernel entry -- 114 MB at once, HBM-bound, and
            // the one stream that is on the critical path (group 0, layer 0: nothing to hide it behind) took 19 us instead of the
            // 11 us of its XCD port.  Group q's first request waits q x UMGEN_ENG_STAGGER_US: its x is q layers away anyway.
            if (item == 0 && q > 0 && D > 1) {
                const unsigned long long until = t_k0 + (unsigned long long)(q * kStaggerTicks);
                while (wall_clock64() < until) __builtin_amdgcn_s_sleep(8);
            }
            // Everything below is derived from these four values INSIDE the item: laundering them keeps the compiler from hoisting
            // ~100 VGPRs / SGPRs of loop-invariant addresses out of the layer loop (they spilled to scratch, on the critical path)
            int tid = tid0, w = w0, gl = g;
            asm volatile("" : "+v"(tid));
            asm volatile("" : "+s"(w), "+s"(gl));
            const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: row addresses stay in SGPRs)
            u64* gqkv = a.gloc + (long)gl * kEngLocStride;
            u64* gpart = gqkv + 3 * E;
            u64* gxb = gpart + 2 * H * 50;
            u64* gpy = gxb + E;                      // mlp partial sums [32 producers][768 rows]
            u64* gxl = gpy + CU * E;                 // in-group x edge (D == 1)
            Rows768<RQ> wq;
            u32x4_t wpl[6];                          // units 12..17 of the mlp c_proj slice (requested after the attention)
            const int rowq = (w * NW + wave) * RQ, rowo = (w * NW + wave) * RO, rowf = (w * NW + wave) * RF;
            const int rq0 = (w * NW + wave) * 3;
            // row r (0..8) of this wave in the packed q | k | v matrix
            auto qkv_row = [&](int r) { return kQFirst ? (r / 3) * E + rq0 + r % 3 : rowq + r; };
            const OarLayerDev lw = a.layers[l];
            const u32 tg = ep + (u32)((rd * 64 + l) * 8);
            // q|k|v, attention-projection and c_fc rows of this wave are requested NOW: they are in flight while the group waits
            // for x (the other D - 1 groups are working); the mlp projection's follow once the attention has freed its registers
            u32x4_t* w2p = reinterpret_cast<u32x4_t*>(lds + L_W2) + tid;
            const bf16_t* wp2 = lw.Wp2 + (long)w * kEngWpUnits * NT * 8;
            // A launch's first item (layer 0) has no idle wait to hide its 14 MB behind, and a wave's loads return in order: what it needs
            // first is requested first -- x, the LN weights and biases, then the q|k|v rows -- so that P1 starts after ~6 MB of the
            // stream instead of behind all of it (measured: 18 us from kernel entry to the start of P1 with the weights requested first).
            float lnr[3];   // ln_1 | ln_2 weights (1536 floats over 512 threads), staged through LDS once x is here
            if (load_w) {
#pragma unroll
                for (int k = 0; k < 3; ++k) lnr[k] = ldg((tid + k * NT < E ? lw.ln_a : lw.ln_b - E) + tid + k * NT);
            }
            float bq = 0.f, bo = 0.f;
            if (lane < RQ) bq = ldg(lw.bqkv + qkv_row(lane));
            if (lane < RO) bo = ldg(lw.bo + rowo + lane);
            float x_first[2] = {0.f, 0.f};
            if (l == 0) {
                x_first[0] = ldg(a.xdec + (long)s * E + tid);
                if (tid + NT < E) x_first[1] = ldg(a.xdec + (long)s * E + tid + NT);
            }
            if (SYS) {
                // layer switch of a resident group (once per layer and step): the parked mlp rows go through 6 staging registers at a
                // time BEFORE anything else is requested (the q|k|v rows, c_proj / c_fc rows and K/V buffers then fill the registers)
                if (load_w) {
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb) {
                        u32x4_t wp[6];
#pragma unroll
                        for (int j = 0; j < 6; ++j) wp[j] = ldwu(wp2 + (long)(6 * hb + j) * NT * 8, (u32)tid * 8u);
#pragma unroll
                        for (int j = 0; j < 6; ++j) w2p[(6 * hb + j) * NT] = wp[j];
                    }
                }
                req768_rows<RQ, true>(wq, lw.Wqkv, qkv_row, lane);
                if (load_w || !kSysKeepWo) req768<RO, !kSysKeepWo>(wo, lw.Wo, rowo, lane);
                if (load_w || !kSysKeepWf) req768<RF, !kSysKeepWf>(wf, lw.Wfc, rowf, lane);
            } else {
                u32x4_t wp[12];
                req768_rows<RQ, false>(wq, lw.Wqkv, qkv_row, lane);
#pragma unroll
                for (int j = 0; j < 12; ++j) wp[j] = ldwu(wp2 + (long)j * NT * 8, (u32)tid * 8u);

                // the first 12 units of the mlp c_proj slice (this thread's full row) are parked in LDS until P4 (the attention needs the
                // registers); only the q|k|v rows were requested before them, so this waits for those two -- the rest stays in flight
                // (only this thread reads its parked units back).  Units 12..17 are requested once the attention has freed its registers.
#pragma unroll
                for (int j = 0; j < 12; ++j) w2p[j * NT] = wp[j];
                // (the other two matrices only now: every wave's q|k|v and parked rows reach the memory system ahead of anybody's c_proj / c_fc rows)
                req768(wo, lw.Wo, rowo, lane);
                req768(wf, lw.Wfc, rowf, lane);
                if (STAMPS && timer && first_item) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); a.stamps[13] += wall_clock64() - t_k0; }
            }
            // attention geometry of this CU: head hh, half of the L + 1 keys, split in 8 wave spans of 8-key passes
            const int hh = w >> 1, half = w & 1;
            const int nk = kQFirst ? Lk : Lk + 1;     // keys of the spans: the cached ones (the new token's own key is merged behind them) / all
            const int n0 = min(nk, (((nk + 1) >> 1) + 7) & ~7);
            const int ka = half ? n0 : 0, kb = half ? nk : n0;
            const int span = ((((kb - ka) + NW - 1) / NW) + 7) & ~7;
            const int k_lo = ka + wave * span;
            int k_hi = min(kb, k_lo + span);
            const int piece = lane & 7, kg = lane >> 3;
            const bool pact = piece < 6;
            const bf16_t* kbase = a.kvcache + (long)l * a.kv_layer_stride + (long)s * a.kv_scene_stride + (long)hh * a.Lmax * kHeadDim;
            const bf16_t* vbase = kbase + (long)H * a.Lmax * kHeadDim;
            // With D > 1 this group now waits for the other groups: pull this CU's share of the cached K / V rows (two contiguous
            // byte ranges, head-major cache) into the XCD's L2 meanwhile -- one dword per 128-byte line, default cache policy, issued
            // BEHIND the non-temporal weight requests so that the weight stream does not push them out again.  The attention's own
            // loads then hit the L2 (4.3 TB/s per XCD) instead of the fabric port (1.3 TB/s): K/V was 2.6 us of the layer at L = 1100.
            // (Not on layer 0: a launch's first item has no idle wait, and the touch loop consumes its loads -- which return behind the whole
            //  weight stream, a wave's loads being in order: P1 started 6.6 us late.)
            u32 touched = 0;
            if (!SYS && D > 1 && l != 0) {
                const int n_lines = ((kb - ka) * kHeadDim * 2 + 127) >> 7;
                const char* k0p = reinterpret_cast<const char*>(kbase + (long)ka * kHeadDim);
                const char* v0p = reinterpret_cast<const char*>(vbase + (long)ka * kHeadDim);
                for (int ln = tid; ln < n_lines; ln += NT) {
                    touched ^= *(const UMGEN_GLOBAL u32*)(k0p + ((long)ln << 7));
                    touched ^= *(const UMGEN_GLOBAL u32*)(v0p + ((long)ln << 7));
                }
            }
            // 16 more keys of this wave's span (behind the 32 that the two register buffers take) go straight into its LDS strip: 4 LDS-DMA
            // instructions, 3 KB, issued HERE -- in front of the idle wait -- because the compiler drains every load in flight
            // (s_waitcnt vmcnt(0)) in front of the first LDS read behind an LDS-DMA.  With the two register buffers alone every later
            // 16-key chunk waited ~0.3 us of an L2 round trip behind 0.3 us of work (attention 3.55 us at L = 1100).  Not where no idle
            // wait follows (a launch's first item, a busy systolic group): there the drain would be the whole weight stream.
            const bool staged = kStageKeys != 0 && !SYS && D > 1 && l != 0 && k_lo + 32 < k_hi;
            float* kvs = lds + L_KV + wave * 768;
            if (staged) {
                const u32 eoff = (u32)(k_lo + 32) * (u32)kHeadDim + (u32)lane * 8u;   // element offset of this lane's 16 bytes
                typedef __attribute__((address_space(3))) void* lds_ptr;
                __builtin_amdgcn_global_load_lds((const UMGEN_GLOBAL void*)(kbase + eoff), (lds_ptr)kvs, 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const UMGEN_GLOBAL void*)(vbase + eoff), (lds_ptr)(kvs + 384), 16, 0, 0);
                if (lane < 32) {   // keys 10.67 .. 16 of the strip: the second half-kilobyte
                    __builtin_amdgcn_global_load_lds((const UMGEN_GLOBAL void*)(kbase + eoff + 512), (lds_ptr)(kvs + 256), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds((const UMGEN_GLOBAL void*)(vbase + eoff + 512), (lds_ptr)(kvs + 384 + 256), 16, 0, 0);
                }
            }
            const int kst = staged ? kStageKeys : 0;
            stamp(-1);
            // ================= P1: x -> LN -> q | k | v =================
            float* lnw = lds + L_LN;
            if (load_w) {   // (nobody reads the previous item's LayerNorm weights any more: its last use is in front of P4's barrier)
#pragma unroll
                for (int k = 0; k < 3; ++k) lnw[tid + k * NT] = lnr[k];
            }
            if (l == 0) {
                xs[tid] = x_first[0];
                if (tid + NT < E) xs[tid + NT] = x_first[1];
                wg_barrier();
            } else {
                gather<2>(c, tid, D == 1 ? gxl : a.gx + (long)s * E, E, tg + 0, xs);   // (ends with the workgroup barrier)
            }
            if (kStageKeys && !SYS) {
                // the compiler drains every load in flight in front of the first LDS read behind an LDS-DMA: take that wait HERE, where
                // everything has landed during the idle wait, and not in front of the LayerNorm's LDS reads behind the K/V requests below
                __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) (expcnt / lgkmcnt untouched), as an instruction the wait-count pass models
            }
            stamp(0);   // waited for x
#ifndef UMGEN_ENG_NB
#define UMGEN_ENG_NB 2
#endif
            // 8-key passes per register buffer, buffers (NB * KP * 8 keys of a wave in flight).  NB = 3 costs 30 spilled VGPRs whose
            // reloads (s_waitcnt vmcnt(0)) also wait for every request in flight; with the K/V rows already in the L2 two suffice
            constexpr int KP = 2, NB = UMGEN_ENG_NB;
            u32x4_t kc[NB][KP], vc[NB][KP];
            auto kv_req = [&](int buf, int k0) {
#pragma unroll
                for (int i = 0; i < KP; ++i) {
                    const u32 off = (u32)min(k0 + 8 * i + kg, a.Lmax - 1) * (u32)kHeadDim + (u32)piece * 8u;
                    if (pact) { kc[buf][i] = ldwu(kbase, off); vc[buf][i] = ldwu(vbase, off); }
                    else { kc[buf][i] = u32x4_t{0, 0, 0, 0}; vc[buf][i] = u32x4_t{0, 0, 0, 0}; }
                }
            };
            if (k_lo < k_hi) kv_req(0, k_lo);
            if (k_lo + 8 * KP < k_hi) kv_req(1, k_lo + 8 * KP);
            {
                f32x2_t x1[4], x2[4];
                ln768(xs, lnw, lane, x1, x2);
                const int n = qkv_row(min(lane, RQ - 1));
                float v = 0.f;
                if (kQFirst) {
                    float oq[3];
                    dot768_range<TT, RQ, 0, 3>(wq, x1, x2, lane, oq);
#pragma unroll
                    for (int r = 0; r < 3; ++r) v = (lane == r) ? oq[r] : v;
                    if (lane < 3) put_local(gqkv, (u32)n, tg + 1, v + bq);       // q rows are on their way while the k | v rows are multiplied
                    float okv[6];
                    dot768_range<TT, RQ, 3, 9>(wq, x1, x2, lane, okv);
#pragma unroll
                    for (int r = 0; r < 6; ++r) v = (lane == 3 + r) ? okv[r] : v;
                } else {
                    float out[RQ];
                    dot768<TT, RQ>(wq, x1, x2, lane, out);
#pragma unroll
                    for (int r = 0; r < RQ; ++r) v = (lane == r) ? out[r] : v;
                }
                v += bq;
                if (lane < RQ && !(kQFirst && lane < 3)) {
                    put_local(gqkv, (u32)n, tg + 1, v);
                    if (n >= E) {   // K / V rows of the new token: bf16 into the cache (head-major [2][H][Lmax][48])
                        const int cc = n - E, kvsel = cc / E, hc = cc % E;
                        (a.kvcache + (long)l * a.kv_layer_stride + (long)s * a.kv_scene_stride)[
                            (u32)(((kvsel * H + hc / kHeadDim) * a.Lmax + Lk) * kHeadDim + hc % kHeadDim)] = bits16<TT>(v);
                    }
                }
            }
            if (NB > 2 && k_lo + 16 * KP < k_hi) kv_req(NB > 2 ? 2 : 0, k_lo + 16 * KP);   // (the q|k|v rows' registers are free now)
            stamp(1);   // LN + q|k|v rows
            // (never true for bf16 K/V bit patterns XORed; keeps the L2 touch loads alive.  Consumed HERE, not before P1: a wave's loads return
            //  in order, so on a launch's first item the touches arrive behind the whole weight stream -- P1 waited 6.6 us for them)
            if (touched == 0x7ff00123u) (lds + L_SM)[0] = 0.f;
            // ================= P2: attention of (head hh, half) =================
            {
                // q_h | k_h | v_h of the new token
                float* qs = lds + L_QKV;
                // values [lo, hi) of q_h | k_h | v_h (48 each) of this CU's head out of the group's q | k | v granules
                auto poll_head = [&](int lo, int hi) {
                    if (!c.failed) {
                        const int src = (tid / kHeadDim) * E + hh * kHeadDim + tid % kHeadDim;
                        const bool mine = tid >= lo && tid < hi;
                        for (u32 spins = 0;;) {
                            bool ok = true;
                            u64 v = 0;
                            if (mine) { v = get(gqkv, (u32)src); ok = (u32)(v >> 32) == tg + 1; }
                            if (ok && mine) qs[tid] = __uint_as_float((u32)v);
                            if (!__any(!ok)) break;
                            if (++spins > kSpinLimit) { if ((tid & 63) == 0) atomicExch(c.err, (tg + 1) | 0x80000000u); c.failed = true; break; }
                            if ((spins & 255u) == 0 && __hip_atomic_load(c.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { c.failed = true; break; }
                        }
                    }
                    wg_barrier();
                };
                poll_head(0, kQFirst ? kHeadDim : 3 * kHeadDim);
                stamp(2);   // waited for q_h (| k_h | v_h)
                float q8[8];   // this lane's piece of q
#pragma unroll
                for (int e = 0; e < 8; ++e) q8[e] = pact ? qs[piece * 8 + e] : 0.f;
                float m_run = -INFINITY, l_run = 0.f, o8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o8[e] = 0.f;
                auto chunk = [&](const u32x4_t (&kcb)[KP], const u32x4_t (&vcb)[KP], int k0) {
                    float sc[KP];
                    float mc = -INFINITY;
#pragma unroll
                    for (int i = 0; i < KP; ++i) {
                        const int k = k0 + 8 * i + kg;
                        float kf[8];
                        unpack8<TT>(kcb[i], kf);
                        if (k == Lk) {   // the new token's own k, as the cache will hold it (bf16)
#pragma unroll
                            for (int e = 0; e < 8; ++e) kf[e] = pact ? round16<TT>(qs[kHeadDim + piece * 8 + e]) : 0.f;
                        }
                        float d = 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) d = fmaf(q8[e], kf[e], d);
                        d += dpp_xor1(d);
                        d += dpp_xor2(d);
                        d += dpp_half_mirror(d);
                        d = (k < k_hi) ? d * kScaleQK : -INFINITY;
                        sc[i] = d;
                        mc = fmaxf(mc, d);
                    }
                    const float m_new = fmaxf(m_run, mc);
                    if (m_new > -INFINITY) {
                        const float scale = __expf(m_run - m_new);   // exp(-inf) = 0 on the first chunk
                        l_run *= scale;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o8[e] *= scale;
#pragma unroll
                        for (int i = 0; i < KP; ++i) {
                            const int k = k0 + 8 * i + kg;
                            float vf[8];
                            unpack8<TT>(vcb[i], vf);
                            if (k == Lk) {
#pragma unroll
                                for (int e = 0; e < 8; ++e) vf[e] = pact ? round16<TT>(qs[2 * kHeadDim + piece * 8 + e]) : 0.f;
                            }
                            const float p = __expf(sc[i] - m_new);
                            l_run += p;
#pragma unroll
                            for (int e = 0; e < 8; ++e) o8[e] = fmaf(p, vf[e], o8[e]);
                        }
                        m_run = m_new;
                    }
                };
                // chunks of 8 KP keys in NB register buffers, all requested before q|k|v were exchanged; a buffer is requested again as
                // soon as it has been consumed
                for (int k0 = k_lo; k0 < k_hi; k0 += 8 * KP * NB) {
                    if (k0 == k_lo + 8 * KP * NB && kst) {
                        // the staged keys sit between the first round of the register buffers and their reloads
                        u32x4_t kl[KP], vl[KP];
#pragma unroll
                        for (int i = 0; i < KP; ++i) {
                            const float* kr = kvs + (8 * i + kg) * 24 + piece * 4;       // key j of the strip: 96 bytes at 24 floats
                            kl[i] = pact ? *reinterpret_cast<const u32x4_t*>(kr) : u32x4_t{0, 0, 0, 0};
                            vl[i] = pact ? *reinterpret_cast<const u32x4_t*>(kr + 384) : u32x4_t{0, 0, 0, 0};
                        }
                        chunk(kl, vl, k0);
                        k0 += kst;
                        if (k0 >= k_hi) break;
                    }
                    chunk(kc[0], vc[0], k0);
                    if (k0 + 8 * KP * NB + (k0 == k_lo ? kst : 0) < k_hi) kv_req(0, k0 + 8 * KP * NB + (k0 == k_lo ? kst : 0));
                    if (k0 + 8 * KP < k_hi) {
                        chunk(kc[1], vc[1], k0 + 8 * KP);
                        if (k0 + 8 * KP * (NB + 1) + (k0 == k_lo ? kst : 0) < k_hi) kv_req(1, k0 + 8 * KP * (NB + 1) + (k0 == k_lo ? kst : 0));
                    }
                    if (NB > 2 && k0 + 16 * KP < k_hi) {
                        chunk(kc[NB > 2 ? 2 : 0], vc[NB > 2 ? 2 : 0], k0 + 16 * KP);
                        if (k0 + 8 * KP * (NB + 2) < k_hi) kv_req(NB > 2 ? 2 : 0, k0 + 8 * KP * (NB + 2));
                    }
                }
                if (kQFirst) {
                    // the new token's own key: k_h | v_h have been on their way since P1; one wave of the head's second half adds the
                    // key as one more term of its online softmax (as the cache will hold it: rounded to 16 bits)
                    poll_head(kHeadDim, 3 * kHeadDim);
                    if (half == 1 && wave == NW - 1) {
                        k_hi = Lk + 1;
                        u32x4_t kz[KP], vz[KP];
#pragma unroll
                        for (int i = 0; i < KP; ++i) { kz[i] = u32x4_t{0, 0, 0, 0}; vz[i] = u32x4_t{0, 0, 0, 0}; }
                        chunk(kz, vz, Lk);
                    }
                }
                // 64 lane-group partials of this CU -> LDS -> one half partial (m, l, o[48]) published by wave 0
                float* sm = lds + L_SM;
                float* so = lds + L_SO;
                const int gi = wave * 8 + kg;
                if (piece == 0) { sm[gi] = m_run; sm[64 + gi] = l_run; }
                if (pact) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) so[gi * kHeadDim + piece * 8 + e] = o8[e];
                }
                wg_barrier();
                {
                    // every wave recomputes the 64 merge weights (cheap), then thread (jg, d) folds 8 of the 64 partial rows of
                    // column d; 48 threads add the 8 folds in a fixed order and publish the half partial (m, l, o[48])
                    const float mg = sm[lane];
                    const float M = wave_max(mg);
                    const float wg = (M > -INFINITY) ? __expf(mg - M) : 0.f;
                    const float Ls = wave_sum(wg * sm[64 + lane]);
                    float* fold = lds + L_HS;   // (the MLP's strip: free until P4 -- so no barrier is needed between the folds' readers and P3's gather)
                    if (tid < 8 * kHeadDim) {
                        const int jg = tid / kHeadDim, d = tid % kHeadDim;
                        float o = 0.f;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float wj = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * (8 * jg + j), __float_as_int(wg)));
                            o = fmaf(wj, so[(8 * jg + j) * kHeadDim + d], o);
                        }
                        fold[jg * kHeadDim + d] = o;
                    }
                    wg_barrier();
                    if (tid < kHeadDim) {
                        float o = 0.f;
#pragma unroll
                        for (int jg = 0; jg < 8; ++jg) o += fold[jg * kHeadDim + tid];
                        u64* gp = gpart + (hh * 2 + half) * 50;
                        put_local(gp, (u32)tid, tg + 2, o);
                        if (tid == 0) { put_local(gp, 48u, tg + 2, M); put_local(gp, 49u, tg + 2, Ls); }
                    }
                }
            }
            stamp(3);   // attention of this CU's half
            // ================= P3: merge the halves -> c_proj -> x' =================
            gather<4>(c, tid, gpart, 2 * H * 50, tg + 2, lds + L_GP);
            stamp(4);   // waited for the half partials
#pragma unroll
            for (int j = 0; j < 6; ++j) wpl[j] = ldwu(wp2 + (long)(12 + j) * NT * 8, (u32)tid * 8u);
            {
                const float* gp = lds + L_GP;
                for (int col = tid; col < E; col += NT) {
                    const int h2 = col / kHeadDim, d = col % kHeadDim;
                    const float* p0 = gp + (h2 * 2) * 50;
                    const float* p1 = p0 + 50;
                    const float m0 = p0[48], m1 = p1[48];
                    const float M = fmaxf(m0, m1);
                    const float e0 = (m0 > -INFINITY) ? expf(m0 - M) : 0.f, e1 = (m1 > -INFINITY) ? expf(m1 - M) : 0.f;
                    const float Ls = fmaf(e1, p1[49], e0 * p0[49]);
                    as[col] = fmaf(e1, p1[d], e0 * p0[d]) / Ls;
                }
                wg_barrier();
                f32x2_t x1[4], x2[4];
                float out[RO];
                load8p(as + lane * 8, x1);
                load8p(as + 512 + (lane & 31) * 8, x2);
                dot768<TT, RO>(wo, x1, x2, lane, out);
                float v = 0.f;
#pragma unroll
                for (int r = 0; r < RO; ++r) v = (lane == r) ? out[r] : v;
                if (lane < RO) {
                    const int n = rowo + lane;
                    put_local(gxb, (u32)n, tg + 3, xs[n] + (v + bo));
                }
            }
            stamp(5);   // merge + c_proj rows
            // ================= P4: x' -> LN -> c_fc -> GELU -> this CU's partial sums of the mlp c_proj =================
            gather<2>(c, tid, gxb, E, tg + 3, xb);
            stamp(6);   // waited for x'
            float* hsl = lds + L_HS;              // [96] gelu(c_fc) of this CU's hidden units
            float* hrow = hsl + 96;               // [256] second halves of the shared rows 512..767
            float* part = hrow + 256;             // [32][24] gathered partial sums (P5)
            {
                f32x2_t x1[4], x2[4];
                float out[RF];
                ln768(xb, lnw + E, lane, x1, x2);
                dot768<TT, RF>(wf, x1, x2, lane, out);
                float v = 0.f;
#pragma unroll
                for (int r = 0; r < RF; ++r) v = (lane == r) ? out[r] : v;
                if (lane < RF) hsl[wave * RF + lane] = gelu_erf(v);
            }
            wg_barrier();
            stamp(11);  // LN + c_fc rows + GELU
            {
                // thread t: all 96 columns of output row t (units 0..11, parked in LDS) + half of the columns of row 512 + (t & 255)
                // (units 12..17: columns 48 (t >> 8) .. +47); every lane reads the same h values (LDS broadcast)
                const f32x2_t zero = {0.f, 0.f};
                f32x2_t accA = zero, accB = zero;
#pragma unroll
                for (int j = 0; j < 12; ++j) {
                    f32x2_t xv[4];
                    load8p(hsl + 8 * j, xv);
                    accA = dot8<TT>(w2p[j * NT], xv, accA);
                    // SYS keeps the c_proj / c_fc rows (92 VGPRs) live through this phase: stop the scheduler from hoisting all 36 LDS
                    // reads (144 VGPRs) to the front, which pushed those rows out to scratch memory
                    if (SYS && (j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
                const int cb = 6 * (tid >> 8);
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    f32x2_t xv[4];
                    load8p(hsl + 8 * (cb + j), xv);
                    accB = dot8<TT>(wpl[j], xv, accB);
                }
                const float yA = accA.x + accA.y;
                float yB = accB.x + accB.y;
                if (tid >= 256) hrow[tid - 256] = yB;
                wg_barrier();
                u64* mine = gpy + (long)w * E;
                put_local(mine, (u32)tid, tg + 4, yA);
                if (tid < 256) put_local(mine, 512u + (u32)tid, tg + 4, yB + hrow[tid]);
            }
            stamp(7);   // LN + c_fc rows + partial sums
            // ================= P5: the 32 partial sums of this CU's 24 rows -> x'' (next layer's x) =================
            if (!c.failed) {
                // producer p's partials of rows 24 w .. 24 w + 23 sit at gpy[p * 768 + 24 w + r]: 768 granules in 32 runs of 192 B
                u32 got = 0;
                const u32 need = (tid < 256) ? 3u : 1u;
                for (u32 spins = 0;;) {
                    u64 v[2];
#pragma unroll
                    for (int k = 0; k < 2; ++k)
                        if (((need & ~got) >> k) & 1u) { const u32 i = (u32)tid + (u32)(k * NT); v[k] = get(gpy + 24 * w, (i / 24u) * (u32)E + i % 24u); }
#pragma unroll
                    for (int k = 0; k < 2; ++k)
                        if (((need & ~got) >> k) & 1u) {
                            if ((u32)(v[k] >> 32) == tg + 4) { part[tid + k * NT] = __uint_as_float((u32)v[k]); got |= 1u << k; }
                        }
                    if (!__any(got != need)) break;
                    if (++spins > kSpinLimit) { if ((tid & 63) == 0) atomicExch(c.err, (tg + 4) | 0x80000000u); c.failed = true; break; }
                    if ((spins & 255u) == 0 && __hip_atomic_load(c.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { c.failed = true; break; }
                }
            }
            wg_barrier();
            stamp(8);   // waited for the partial sums
            if (tid < 24) {
                float sum = 0.f;
#pragma unroll
                for (int p = 0; p < CU; ++p) sum += part[p * 24 + tid];      // fixed order: producer 0, 1, ..., 31
                const int n = 24 * w + tid;
                const float xn = xb[n] + sum;
                if (l + 1 == a.n_layers) (a.xdec + (long)s * E)[(u32)n] = xn;
                else if (D == 1) put_local(gxl, (u32)n, tg + 8, xn);
                else put_far(a.gx + (long)s * E, (u32)n, tg + 8, xn);
            }
            stamp(9);   // mlp c_proj rows
            if (STAMPS && timer) a.stamps[10] += 1;
        }
    }
}

// census: which XCD did each workgroup of an engine-shaped launch land on?
__global__ __launch_bounds__(kEngThreads) void oar_engine_census_kernel(u32* counts) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (threadIdx.x == 0) {
        lds[0] = 0.f;
        atomicAdd(counts + xcc_id(), 1u);
    }
}

size_t oar_engine_lds_bytes() {
    const size_t need = (size_t)L_TOTAL * sizeof(float);   // (with the K/V strips: 159.2 KB of the CU's 160)
    return need > (size_t)(96 << 10) ? need : (size_t)(96 << 10);   // > 80 KB: never two engine workgroups on one CU
}

hipError_t launch_oar_engine_census(hipStream_t s, int n_groups, unsigned int* d_counts16) {
    hipLaunchKernelGGL(oar_engine_census_kernel, dim3(n_groups * kEngGroup), dim3(kEngThreads), oar_engine_lds_bytes(), s, d_counts16);
    return hipGetLastError();
}

// once per device (umgen_create): the engine's dynamic LDS exceeds the default limit
hipError_t oar_engine_prepare() {
    for (const void* f : {reinterpret_cast<const void*>(oar_engine_kernel<false, bf16_t, false>), reinterpret_cast<const void*>(oar_engine_kernel<true, bf16_t, false>),
                          reinterpret_cast<const void*>(oar_engine_kernel<false, f16_t, false>), reinterpret_cast<const void*>(oar_engine_kernel<true, f16_t, false>),
                          reinterpret_cast<const void*>(oar_engine_kernel<false, bf16_t, true>), reinterpret_cast<const void*>(oar_engine_kernel<true, bf16_t, true>),
                          reinterpret_cast<const void*>(oar_engine_kernel<false, f16_t, true>), reinterpret_cast<const void*>(oar_engine_kernel<true, f16_t, true>),
                          reinterpret_cast<const void*>(oar_engine_census_kernel)}) {
        hipError_t rc = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)oar_engine_lds_bytes());
        if (rc != hipSuccess) return rc;
    }
    return hipSuccess;
}

template <typename TT, bool SYS>
static void launch_engine_t(hipStream_t s, const OarEngineArgs& a) {
    const dim3 grid(a.NG * kEngGroup), block(kEngThreads);
    const size_t shm = oar_engine_lds_bytes();
    if (a.stamps) hipLaunchKernelGGL((oar_engine_kernel<true, TT, SYS>), grid, block, shm, s, a);
    else hipLaunchKernelGGL((oar_engine_kernel<false, TT, SYS>), grid, block, shm, s, a);
}

hipError_t launch_oar_engine(hipStream_t s, const OarEngineArgs& a) {
    if (a.fp16) { if (a.systolic) launch_engine_t<f16_t, true>(s, a); else launch_engine_t<f16_t, false>(s, a); }
    else { if (a.systolic) launch_engine_t<bf16_t, true>(s, a); else launch_engine_t<bf16_t, false>(s, a); }
    return hipGetLastError();
}

}  // namespace umgen
