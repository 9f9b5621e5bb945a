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
// Arithmetic (fixed, independent of B / D / group placement, so scenes are batch-invariant): fp32 activations, 16-bit weights (bf16 or
// IEEE half) and 16-bit K/V cache, fp32 accumulation.  VALU row dot products (q|k|v, c_proj): lane l owns k = 8l..8l+7 (+512 i), packed
// fp32 FMAs, transposed wave sums (rows_sum).  Matrix-core row products (c_fc, mlp partial sums; UMGEN_ENG_MFMA): the activations enter
// as hi + lo 16-bit columns of one v_mfma_f32_16x16x32 (2^-17 relative), k split over the 8 waves, partial sums added in wave order.
// Attention of a head: its keys are split in two halves (two CUs), each half in 8 wave spans, each span in groups of 4 lanes per key
// (16 keys per pass) with an online softmax per lane group; the 16 lane groups of a wave, the 64 partials of a half, then the two
// halves, are merged in a fixed order.
#include "oar_common.h"
#include "bg_queue.h"
#include "bg_worker.h"

namespace umgen {

namespace {

constexpr int RQ = 3 * E / CU / NW;   // 9 q|k|v rows per wave
constexpr int RO = E / CU / NW;       // 3 c_proj rows per wave
constexpr int RF = F / CU / NW;       // 12 c_fc rows per wave
constexpr int RP = E / CU / NW;       // 3 mlp c_proj rows per wave
static_assert(RQ * NW * CU == 3 * E && RO * NW * CU == E && RF * NW * CU == F, "row partition");
// UMGEN_ENG_MFMA bits: 1 q|k|v rows (P1), 2 c_proj rows (P3), 4 c_fc rows (P4), 8 mlp partial sums (P4b; changes the repacked layout)
constexpr bool kMfma = UMGEN_ENG_MFMA != 0, kMfmaQ = (UMGEN_ENG_MFMA & 1) != 0, kMfmaO = (UMGEN_ENG_MFMA & 2) != 0, kMfmaF = (UMGEN_ENG_MFMA & 4) != 0,
               kMfmaP = (UMGEN_ENG_MFMA & 8) != 0, kMfmaA = (UMGEN_ENG_MFMA & 16) != 0;   // 16: attention (needs the dim-major V cache)
#ifndef UMGEN_ENG_MFMA_LATE_TILE
#define UMGEN_ENG_MFMA_LATE_TILE 0
#endif
constexpr bool kLateTile = UMGEN_ENG_MFMA_LATE_TILE != 0;
// LDS carve (floats)
constexpr int L_XS = 0;                    // x of the item (kept until the attention projection's residual)   [768]
constexpr int L_XB = L_XS + E;             // x' (kept until the MLP projection's residual)                    [768]
constexpr int L_AS = L_XB + E;             // merged attention output                                            [768]
constexpr int L_HS = L_AS + E;             // this CU's 96 gelu(c_fc) values [96] | half-row sums [256] | gathered mlp partial sums [32][24]   [1152]
constexpr int L_QKV = L_HS + 1152;          // q_h | k_h | v_h of this CU's head                                  [144 -> 160]
constexpr int L_GP = L_QKV + 160;          // gathered half partials [32][50]                                   [1600]
constexpr int L_SM = L_GP + 2 * H * 50;    // per lane-group m [64], l [64], weights [64]                        [192]
constexpr int L_SO = L_SM + 192;           // per lane-group o [64][48]                                          [3072]
constexpr int L_MISC = L_SO + (kMfmaA ? NW : 64) * 48;   // rank / scratch (matrix-core attention: one partial per wave)        [16]
constexpr int L_LN = L_MISC + 16;          // ln_1 | ln_2 weights of the item                                     [1536]
constexpr int L_W2 = L_LN + 2 * E;         // parked mlp c_proj units 0..11 of every thread: [12][512] x 16 B      [24576]
// Row dot products on the matrix cores (UMGEN_ENG_MFMA, frame.h): activation split x = hi + lo in the operand type (16-bit each) [768 + 768],
// the same for the CU's 96 gelu(c_fc) values [96 + 96], a zero strip for the 14 unused operand columns, per-wave partial row sums [8][96]
constexpr int L_XH = L_W2 + 12 * NT * 4;    // hi halves of the 768 activations (16-bit)     [384 floats]
constexpr int L_XL = L_XH + E / 2;          // lo halves                                      [384]
constexpr int L_HH = L_XL + E / 2;          // hi | lo of the 96 hidden values                 [48 + 48]
constexpr int L_ZR = L_HH + 96;             // zeros                                            [64]
constexpr int L_PT = L_ZR + 64;             // partial row sums of the 8 waves' k ranges        [8][96]
constexpr int L_PS = L_PT + NW * 96;        // attention on the matrix cores: per wave the 32 probabilities of a pass as hi [32] | lo [32] 16-bit   [8][32 floats]
constexpr int L_FT = L_PS + NW * 32;       // matrix-core attention: the sixth c_fc tile's 3 fragments of every thread, parked like the mlp rows  [3][512] x 16 B
constexpr bool kParkFT = kMfmaA && kMfmaF;
constexpr int L_TOTAL = kMfma ? L_FT + (kParkFT ? 3 * NT * 4 : 0) : L_XH;
static_assert(L_TOTAL * 4 <= 160 * 1024, "LDS budget");

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
// Transposed wave sums of R rows (R <= 16): a[r] = this lane's partial of row r; returns, in every lane, the wave total of row
// (lane & 15).  Instead of R independent 6-step DPP reductions (7 R instructions + R selects to put row r into lane r), each step
// halves the number of live values: a lane keeps the rows whose index bit matches its lane bit and hands the others to its partner
// (bit 3: row_ror:8, bit 2: row_half_mirror -- BEFORE the quad steps, as it also flips bits 0 / 1 and partners must hold the same row
// subset -- bit 1 / 0: quad_perm), 3 instructions per row pair; the last two steps add the four 16-lane rows with gfx950's permlane swaps.
// Lanes whose row index is >= R end with sums of other rows (never read).  Fixed order, the same for every B / D / placement.
#ifndef UMGEN_ENG_TREDUCE
#define UMGEN_ENG_TREDUCE 1
#endif
template <int N, int R, int CTRL>   // N (power of two) slots of which R hold rows -> N / 2 slots
__device__ inline void rows_step(float (&v)[16], bool hi) {
#pragma unroll
    for (int r = 0; r < N / 2; ++r) {
        if (r + N / 2 < R) {
            const float keep = hi ? v[r + N / 2] : v[r], send = hi ? v[r] : v[r + N / 2];
            v[r] = keep + dpp_mov<CTRL>(send);
        } else if (r < R) {
            v[r] = v[r] + dpp_mov<CTRL>(v[r]);
        }
    }
}
template <int R>
__device__ inline float rows_sum(const f32x2_t (&acc)[R], int lane) {
    static_assert(R <= 16, "rows_sum");
    float v[16];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = acc[r].x + acc[r].y;
    constexpr int R8 = R > 8 ? 8 : R, R4 = R > 4 ? 4 : R, R2 = R > 2 ? 2 : R;
    if (R > 8) rows_step<16, R, 0x128>(v, (lane & 8) != 0);          // row_ror:8: lane i <-> i ^ 8
    if (R > 4) rows_step<8, R8, 0x141>(v, (lane & 4) != 0);          // row_half_mirror: i <-> 7 - i (flips bits 0..2: before the quad steps)
    if (R > 2) rows_step<4, R4, 0x4E>(v, (lane & 2) != 0);           // quad_perm [2,3,0,1]
    else { v[0] += dpp_mov<0x4E>(v[0]); if (R > 1) v[1] += dpp_mov<0x4E>(v[1]); }
    if (R > 1) rows_step<2, R2, 0xB1>(v, (lane & 1) != 0);           // quad_perm [1,0,3,2]
    else v[0] += dpp_mov<0xB1>(v[0]);
    if (R <= 4) v[0] += dpp_mov<0x124>(v[0]);                        // row_ror:4 (every quad holds the same rows)
    if (R <= 8) v[0] += dpp_mov<0x128>(v[0]);                        // row_ror:8
    return sum_rows16(v[0]);
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
    for (int r = 0; r < R1 - R0; ++r) out[r] = wave_sum(acc[r].x + acc[r].y);      // (only the q-first experiment: one reduction per row)
}

template <typename TT, int R>
__device__ inline void dot768(const Rows768<R>& w, const f32x2_t (&x1)[4], const f32x2_t (&x2)[4], int lane, float (&out)[R]) {
    const f32x2_t zero = {0.f, 0.f};
    // the shared tail chunk of a row pair goes to the lower / upper half-wave's row through a 1 / 0 multiplier: one packed FMA per
    // row (p x 1 + acc == acc + p exactly) instead of two selects and a packed add
    const float flo = lane < 32 ? 1.f : 0.f;
    const f32x2_t mlo = {flo, flo}, mhi = {1.f - flo, 1.f - flo};
    f32x2_t acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = dot8<TT>(w.a[r], x1, zero);
#pragma unroll
    for (int j = 0; j < (R + 1) / 2; ++j) {
        const f32x2_t p = dot8<TT>(w.b[j], x2, zero);
        acc[2 * j] = __builtin_elementwise_fma(p, mlo, acc[2 * j]);
        if (2 * j + 1 < R) acc[2 * j + 1] = __builtin_elementwise_fma(p, mhi, acc[2 * j + 1]);
    }
    if (UMGEN_ENG_TREDUCE) {
        const float t = rows_sum<R>(acc, lane);
#pragma unroll
        for (int r = 0; r < R; ++r) out[r] = t;            // (lane r of every 16 holds row r: the callers pick out[r] in lane r)
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) out[r] = wave_sum(acc[r].x + acc[r].y);
    }
}

// LayerNorm (weight only, eps 1e-5, module.py:26-37) of the 768-vector in LDS, in the lane's dot-product layout.  1 / 768 as a
// multiplication and the hardware's reciprocal square root (1 ulp): the IEEE division / square-root sequences were ~50 dependent
// instructions on every wave's critical path, twice per layer
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
    const float mean = wave_sum_all(s) * (1.0f / (float)E);
    const f32x2_t mean2 = {mean, mean};
    f32x2_t q1 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) { const f32x2_t d = x1[e] - mean2; q1 = __builtin_elementwise_fma(d, d, q1); }
#pragma unroll
    for (int e = 0; e < 4; ++e) { const f32x2_t d = x2[e] - mean2; q2 = __builtin_elementwise_fma(d, d, q2); }
    float q = q1.x + q1.y;
    q += (lane < 32) ? (q2.x + q2.y) : 0.f;
    const float rstd = __builtin_amdgcn_rsqf(fmaf(wave_sum_all(q), 1.0f / (float)E, 1e-5f));
    const f32x2_t rstd2 = {rstd, rstd};
#pragma unroll
    for (int e = 0; e < 4; ++e) { x1[e] = (x1[e] - mean2) * rstd2 * l1[e]; x2[e] = (x2[e] - mean2) * rstd2 * l2[e]; }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Row dot products on the matrix cores (kMfma).  A wave's rows x 768 products were ~2/3 of an item's VALU instructions (a third of
// them the 16-bit -> fp32 widening of the weights).  v_mfma_f32_16x16x32 takes the 16-bit weights as they are: A = 16 weight rows x
// 32 k, B = 32 k x 16 columns of which TWO are used -- column 0 holds the activations' 16-bit hi parts, column 1 their lo parts
// (x = hi + lo to 2^-17 relative in bf16, 2^-22 in fp16: fp32-class against the weights' own 2^-9 / 2^-12) -- so one instruction
// per 512 weights gives W . hi and W . lo, added afterwards.  The k range is split over the CU's 8 waves (96 k each: 3 instructions
// per 16-row tile) so that a wave's fragments are as many registers as its whole rows were; the 8 partial sums of a row meet in LDS.
//   fragment (tile t, k-step j) of lane l: 8 weights W[row0 + 16 t + l % 16][96 wave + 32 j + 8 (l / 16) ..]
//   result of tile t: lane l holds column l % 16 of rows 4 (l / 16) .. + 3
// ---------------------------------------------------------------------------------------------------------------------------
// the B operand of k-step j: column 0 = hi parts, column 1 = lo parts, columns 2..15 = 0; k0 = first k of the wave's range in xh / xl
template <typename TT>
__device__ inline void load_bfrags(const float* lds, int hi_off, int lo_off, int k0, int lane, typename Mma16<TT>::vec (&b)[3]) {
    const int n = lane & 15;
    const unsigned char* base = reinterpret_cast<const unsigned char*>(lds + (n == 0 ? hi_off : n == 1 ? lo_off : L_ZR)) + (n < 2 ? 2 * k0 : 0) + 16 * (lane >> 4);
#pragma unroll
    for (int j = 0; j < 3; ++j) b[j] = *reinterpret_cast<const typename Mma16<TT>::vec*>(base + 64 * j);
}
// NTILE tiles x 3 k-steps; hi + lo columns added; the wave's partial sums of rows 16 t + 4 (l / 16) .. + 3 -> part[row] (lanes of column 0)
template <typename TT, int NTILE>
__device__ inline void mfma_rows(const WFrags<NTILE>& w, const typename Mma16<TT>::vec (&b)[3], int lane, f32x4_t (&acc)[NTILE]) {
    typedef typename Mma16<TT>::vec vec;
#pragma unroll
    for (int t = 0; t < NTILE; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int t = 0; t < NTILE; ++t) acc[t] = Mma16<TT>::mfma(__builtin_bit_cast(vec, w.f[t][j]), b[j], acc[t]);
#pragma unroll
    for (int t = 0; t < NTILE; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] += dpp_mov<0x101>(acc[t][r]);       // row_shl:1: column 1 (lo) onto column 0 (hi)
}
template <int NTILE>
__device__ inline void store_partials(float* part, int lane, const f32x4_t (&acc)[NTILE]) {
    if ((lane & 15) == 0) {
#pragma unroll
        for (int t = 0; t < NTILE; ++t) *reinterpret_cast<f32x4_t*>(part + 16 * t + 4 * (lane >> 4)) = acc[t];
    }
}
// LayerNorm of the 768-vector in LDS (statistics by every wave, as ln768), normalised values of THIS wave's 96 k split into hi / lo
template <typename TT>
__device__ inline void ln_split(const float* xs, const float* lnw, int lane, int wave, float* lds) {
    f32x2_t x1[4], x2[4];
    load8p(xs + lane * 8, x1);
    load8p(xs + 512 + (lane & 31) * 8, x2);
    f32x2_t s1 = (x1[0] + x1[1]) + (x1[2] + x1[3]);
    f32x2_t s2 = (x2[0] + x2[1]) + (x2[2] + x2[3]);
    float s = s1.x + s1.y;
    s += (lane < 32) ? (s2.x + s2.y) : 0.f;
    const float mean = wave_sum_all(s) * (1.0f / (float)E);
    const f32x2_t mean2 = {mean, mean};
    f32x2_t q1 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) { const f32x2_t d = x1[e] - mean2; q1 = __builtin_elementwise_fma(d, d, q1); }
#pragma unroll
    for (int e = 0; e < 4; ++e) { const f32x2_t d = x2[e] - mean2; q2 = __builtin_elementwise_fma(d, d, q2); }
    float q = q1.x + q1.y;
    q += (lane < 32) ? (q2.x + q2.y) : 0.f;
    const float rstd = __builtin_amdgcn_rsqf(fmaf(wave_sum_all(q), 1.0f / (float)E, 1e-5f));
    unsigned short* xh = reinterpret_cast<unsigned short*>(lds + L_XH);
    unsigned short* xl = reinterpret_cast<unsigned short*>(lds + L_XL);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int k = 96 * wave + 64 * i + lane;
        if (i == 0 || lane < 32) {
            unsigned short hi, lo;
            split16<TT>((xs[k] - mean) * rstd * lnw[k], hi, lo);
            xh[k] = hi; xl[k] = lo;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the fragment reads below are this wave's own k range: no workgroup barrier)
}



}  // namespace

// systolic schedule: which matrices stay in registers over the scenes of a layer (the others are requested again by every item).
// Round 3 history (-Rpass-analysis=kernel-resource-usage): with the 8-lane attention and the one-row-per-thread mlp phase the kernel sat
// at 254 of 256 VGPRs and keeping the c_proj rows cost 8 spilled VGPRs, the c_fc rows 58, both 77 (slower than re-requesting them).
// With 12-VGPR K/V pieces and the four-rows-per-four-lanes mlp phase both fit (252 VGPRs, no spill, 3 K/V buffers): a layer's c_proj,
// c_fc and parked mlp rows are requested ONCE per step; only the q|k|v rows (3.5 MB, out of this XCD's L2 after the first scene)
// are requested per item.  8 scenes: 961 -> 739 us per launch, 16 scenes: 1916 -> 1377 (profiles/r03_systolic.txt).
#ifndef UMGEN_SYS_KEEP_WO
#define UMGEN_SYS_KEEP_WO 1
#endif
#ifndef UMGEN_SYS_KEEP_WF
#define UMGEN_SYS_KEEP_WF 1
#endif
constexpr bool kSysKeepWo = UMGEN_SYS_KEEP_WO, kSysKeepWf = UMGEN_SYS_KEEP_WF;
#ifndef UMGEN_SYS_LATE_PARK
#define UMGEN_SYS_LATE_PARK 0
#endif
constexpr bool kLatePark = UMGEN_SYS_LATE_PARK;
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
// BG (one scene, !SYS): the workgroups of the XCD groups no scene uses are the background workers of bg_worker.h (the next frame's TAR / ego pass)
template <bool STAMPS, typename TT, bool SYS, bool BG = false>
__global__ __launch_bounds__(kEngThreads) void oar_engine_kernel(OarEngineArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid0 = threadIdx.x;
    const unsigned long long t_k0 = wall_clock64();   // 100 MHz
    // ---- who am I: group (XCD) and rank inside it ----
    const u32 xcc = xcc_id();
    const int g = a.xcc_group[xcc];
    if (g >= a.NG) return;   // (the census guarantees this never happens)
    if (tid0 == 0) reinterpret_cast<u32*>(lds + L_MISC)[0] = atomicAdd(a.ticket + g, 1u) & (u32)(CU - 1);
    if (kMfma && tid0 < 64) lds[L_ZR + tid0] = 0.f;     // the B operand's 14 unused columns
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
    if constexpr (BG) {
        if (a.bg != nullptr && a.B == 1 && g >= D) {
            bg_worker<TT>(a.bg, (g - D) * CU + w0, (a.NG - D) * CU, t_k0, a.bg_only != 0);
            return;
        }
        if (a.bg_only) return;
    }
#ifdef UMGEN_ENG_BURN
    if (!SYS && a.burn_ticks > 0 && pipe >= a.B) {      // measurement: an idle pipeline's XCDs under a synthetic load
        typedef typename Mma16<TT>::vec vec8;
        f32x4_t acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        vec8 fa, fb;
#pragma unroll
        for (int j = 0; j < 8; ++j) { fa[j] = (typename Mma16<TT>::elem)(float)((tid0 + j) & 3); fb[j] = (typename Mma16<TT>::elem)(float)((tid0 * 3 + j) & 1); }
        const unsigned long long until = t_k0 + (unsigned long long)a.burn_ticks;
        const u32x4_t* buf = reinterpret_cast<const u32x4_t*>(a.burn_buf);
        const long n16 = (long)a.burn_kb * 64;     // 16-byte pieces of the buffer
        long pos = ((long)blockIdx.x * NT + tid0) % (n16 > 0 ? n16 : 1);
        u32x4_t sink{0u, 0u, 0u, 0u};
        while (wall_clock64() < until) {
            for (int i = 0; i < a.burn_mfma; i += 8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = Mma16<TT>::mfma(fa, fb, acc[j]);
            }
            if (n16 > 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const u32x4_t v = __builtin_nontemporal_load(buf + pos);
                    sink.x ^= v.x; sink.y ^= v.y; sink.z ^= v.z; sink.w ^= v.w;
                    pos += (long)gridDim.x * NT;
                    if (pos >= n16) pos -= n16;
                }
            }
            for (int i = 0; i < a.burn_sleep; ++i) __builtin_amdgcn_s_sleep(4);
        }
        float tot = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) tot += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
        if (tot == 1234.5f || (sink.x ^ sink.y ^ sink.z ^ sink.w) == 0x12345u) a.err[1] = 1u;     // (never true: keeps the loop alive)
        return;
    }
#endif
    float* xs = lds + L_XS;
    float* xb = lds + L_XB;
    float* as = lds + L_AS;
    // the weight rows of a wave (requested at the start of an item).  SYS: c_proj / c_fc rows and the parked mlp rows at the first
    // scene of a layer only -- they stay in their registers / LDS rows for the other scenes; the q|k|v rows (56 VGPRs, dead after P1)
    // do not fit beside them through the attention (142 spilled VGPRs when kept), so every item requests them again, with the
    // default cache policy: after the layer's first scene they come out of this XCD's L2
    Rows768<RO> wo;
    Rows768<RF> wf;
    Rows768<RQ> wq;
    WFrags<5> fq;            // kMfma: the CU's 72 q|k|v rows as 5 tiles of 16 (this wave's 96 k), c_proj 24 rows as 2 tiles, c_fc 96 rows as 6
    WFrags<2> fo;
    WFrags<6> ff;
    static_assert(!(kMfma && kQFirst), "the q-first experiment is written for the VALU row products");
    bool wq_ahead = false;   // SYS: the q|k|v rows of this item were requested during the previous item's mlp phase

    // items of this group in the order it works through them: (round rd, layer l) -- scene rd * R + pipe.
    //   !SYS: rounds outside, this pipeline's layers (q, q + D, ...) inside;
    //    SYS: layers outside, every scene of the batch inside.  The layers that do not fill a whole round of the D groups (36 = 4 x 8 + 4)
    //         are SHARED: with `rem` such layers, D / rem groups hold layer D n_full + q % rem each and take B rem / D of the scenes
    //         (36 layers: groups g and g + 4 both keep layer 32 + g, one for the first half of the batch, one for the second), so
    //         that every group works through 4.5 B items instead of 5 B / 4 B (the step was as long as the 5 B groups' work)
    const int n_lay = (a.n_layers - q + D - 1) / D;
    int n_items = n_lay > 0 ? n_lay * rounds : 0;
    int n_full = 0, tail_l = -1, ts0 = 0, ts1 = 0;
    if (SYS) {
        n_full = a.n_layers / D;
        const int rem = a.n_layers - n_full * D;
        if (rem > 0) {
            if (D % rem == 0) {
                const int share = D / rem, part = q / rem;
                tail_l = n_full * D + q % rem;
                ts0 = part * a.B / share;
                ts1 = (part + 1) * a.B / share;
            } else if (q < rem) {
                tail_l = n_full * D + q;
                ts1 = a.B;
            }
        }
        n_items = n_full * a.B + (ts1 - ts0);
    }
    for (int item = 0; item < n_items; ++item) {
        const bool tail = SYS && item >= n_full * a.B;
        const int rd = SYS ? (tail ? ts0 + item - n_full * a.B : item % a.B) : item / n_lay;
        const int l = SYS ? (tail ? tail_l : q + D * (item / a.B)) : q + D * (item % n_lay);
        const int s = rd * R + pipe;
        const bool load_w = !SYS || rd == (tail ? ts0 : 0);          // this item requests the layer's weights
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
            // (matrix-core form: thread r finishes row r of the CU's 72 q|k|v / 24 c_proj rows)
            if (kMfmaQ) { if (tid < 72) bq = ldg(lw.bqkv + 72 * w + tid); }
            else if (lane < RQ) bq = ldg(lw.bqkv + qkv_row(lane));
            if (kMfmaO) { if (tid < 24) bo = ldg(lw.bo + 24 * w + tid); }
            else if (lane < RO) bo = ldg(lw.bo + rowo + lane);
            float x_first[2] = {0.f, 0.f};
            if (l == 0) {
                x_first[0] = ldg(a.xdec + (long)s * E + tid);
                if (tid + NT < E) x_first[1] = ldg(a.xdec + (long)s * E + tid + NT);
            }
            if (SYS) {
                // layer switch of a resident group (once per layer and step): the parked mlp rows go through 6 staging registers at a
                // time in front of everything else (the q|k|v rows, c_proj / c_fc rows and K/V buffers then fill the registers).
                // EXPERIMENT, off (kLatePark): the parked rows in three batches of 4 units behind P1, the key loop and P3's gather, so that a
                // layer's first scene starts P1 as soon as its q|k|v rows are there: 8 scenes 748 vs 743 us, 5 / 6 scenes 673 / 692 vs
                // 599 / 630 -- with idle time in front of the item the up-front staging was free, and at 8 scenes the switch's cost is
                // the q|k|v rows' own trip from HBM (profiles/r03_engine_experiments.txt, session K).
                if (load_w && !kLatePark) {
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb) {
                        u32x4_t wp[6];
#pragma unroll
                        for (int j = 0; j < 6; ++j) wp[j] = ldwu(wp2 + (long)(6 * hb + j) * NT * 8, (u32)tid * 8u);
#pragma unroll
                        for (int j = 0; j < 6; ++j) w2p[(6 * hb + j) * NT] = wp[j];
                    }
                }
                if (kMfmaQ) req_frags<5, true>(fq, lw.Wqkv, 72 * w, 72, wave, lane);
                else if (!wq_ahead) req768_rows<RQ, true>(wq, lw.Wqkv, qkv_row, lane);
                if (load_w || !kSysKeepWo) {
                    if (kMfmaO) req_frags<2, !kSysKeepWo>(fo, lw.Wo, 24 * w, 24, wave, lane);
                    else req768<RO, !kSysKeepWo>(wo, lw.Wo, rowo, lane);
                }
                if (load_w || !kSysKeepWf) {
                    if (kMfmaF) {
                        const bf16_t* pf = lw.Wf2 + (long)(w * NW + wave) * 18 * 64 * 8;
                        if (kParkFT) {   // five tiles in registers, the sixth through three staging registers into LDS (read back by this thread in P4)
                            u32x4_t st[3];
#pragma unroll
                            for (int j = 0; j < 3; ++j) st[j] = ldwu(pf, (u32)((15 + j) * 64 + lane) * 8u);
                            req_frags_packed<6, !kSysKeepWf, 0, 5>(ff, pf, lane);
#pragma unroll
                            for (int j = 0; j < 3; ++j) reinterpret_cast<u32x4_t*>(lds + L_FT)[j * NT + tid] = st[j];
                        } else {
                            req_frags_packed<6, !kSysKeepWf>(ff, pf, lane);
                        }
                    }
                    else req768<RF, !kSysKeepWf>(wf, lw.Wfc, rowf, lane);
                }
            } else {
                u32x4_t wp[12];
                if (kMfmaQ) req_frags<5, false>(fq, lw.Wqkv, 72 * w, 72, wave, lane);
                else req768_rows<RQ, false>(wq, lw.Wqkv, qkv_row, lane);
#pragma unroll
                for (int j = 0; j < 12; ++j) wp[j] = ldwu(wp2 + (long)j * NT * 8, (u32)tid * 8u);

                // the first 12 units of the mlp c_proj slice (this thread's full row) are parked in LDS until P4 (the attention needs the
                // registers); only the q|k|v rows were requested before them, so this waits for those two -- the rest stays in flight
                // (only this thread reads its parked units back).  Units 12..17 are requested once the attention has freed its registers.
#pragma unroll
                for (int j = 0; j < 12; ++j) w2p[j * NT] = wp[j];
                // (the other two matrices only now: every wave's q|k|v and parked rows reach the memory system ahead of anybody's c_proj / c_fc rows)
                if (kMfmaO) req_frags<2, false>(fo, lw.Wo, 24 * w, 24, wave, lane);
                else req768(wo, lw.Wo, rowo, lane);
                // (kLateTile, measured and off: the sixth c_fc tile behind the attention so that 4 K/V buffers fit without spilling --
                //  its HBM round trip then sits in front of P4's poll: 472.6 vs 443.9 us with 3 buffers and all 18 fragments up front)
                if (kMfmaF) {
                    const bf16_t* pf = lw.Wf2 + (long)(w * NW + wave) * 18 * 64 * 8;
                    if (kLateTile) req_frags<6, false, 0, 5>(ff, lw.Wfc, 96 * w, 96, wave, lane);
                    else if (kParkFT) {
                        u32x4_t st[3];
#pragma unroll
                        for (int j = 0; j < 3; ++j) st[j] = ldwu(pf, (u32)((15 + j) * 64 + lane) * 8u);
                        req_frags_packed<6, false, 0, 5>(ff, pf, lane);
#pragma unroll
                        for (int j = 0; j < 3; ++j) reinterpret_cast<u32x4_t*>(lds + L_FT)[j * NT + tid] = st[j];
                    } else req_frags_packed<6, false>(ff, pf, lane);
                }
                else req768(wf, lw.Wfc, rowf, lane);
                if (STAMPS && timer && first_item) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); a.stamps[13] += wall_clock64() - t_k0; }
            }
            // attention geometry of this CU: head hh, half of the L + 1 keys, split in 8 wave spans of 16-key passes (4 lanes per key)
            const int hh = w >> 1, half = w & 1;
            const int nk = kQFirst ? Lk : Lk + 1;     // keys of the spans: the cached ones (the new token's own key is merged behind them) / all
            constexpr int KPA = kMfmaA ? 32 : KPW;    // keys per pass (matrix-core attention: 32)
            const int n0 = min(nk, (((nk + 1) >> 1) + KPA - 1) & ~(KPA - 1));
            const int ka = half ? n0 : 0, kb = half ? nk : n0;
            const int span = ((((kb - ka) + NW - 1) / NW) + KPA - 1) & ~(KPA - 1);
            const int k_lo = ka + wave * span;
            int k_hi = min(kb, k_lo + span);
            const int piece = lane & (LPK - 1), kg = lane / LPK;
            const bf16_t* kbase = a.kvcache + (long)l * a.kv_layer_stride + (long)s * a.kv_scene_stride + (long)hh * a.Lmax * kHeadDim;
            const bf16_t* vbase = kbase + (long)H * a.Lmax * kHeadDim;
            // matrix-core attention: V of this head dim-major [48][Lmax] (8 consecutive keys of one dimension are one 16-byte request)
            const bf16_t* vtbase = kMfmaA ? a.vtcache + (long)l * a.vt_layer_stride + (long)s * a.vt_scene_stride + (long)hh * kHeadDim * a.Lmax : nullptr;
            // With D > 1 this group now waits for the other groups: pull this CU's share of the cached K / V rows (two contiguous
            // byte ranges, head-major cache) into the XCD's L2 meanwhile -- one dword per 128-byte line, default cache policy, issued
            // BEHIND the non-temporal weight requests so that the weight stream does not push them out again.  The attention's own
            // loads then hit the L2 (4.3 TB/s per XCD) instead of the fabric port (1.3 TB/s): K/V was 2.6 us of the layer at L = 1100.
            // (Not on layer 0: a launch's first item has no idle wait, and the touch loop consumes its loads -- which return behind the whole
            //  weight stream, a wave's loads being in order: P1 started 6.6 us late.)
            u32 touched = 0;
#ifndef UMGEN_SYS_TOUCH
#define UMGEN_SYS_TOUCH 0
#endif
            if ((!SYS || UMGEN_SYS_TOUCH) && D > 1 && l != 0) {
                const int n_lines = ((kb - ka) * kHeadDim * 2 + 127) >> 7;
                const char* k0p = reinterpret_cast<const char*>(kbase + (long)ka * kHeadDim);
                const char* v0p = reinterpret_cast<const char*>(vbase + (long)ka * kHeadDim);
                if (kMfmaA) {
                    // K as before; V dim-major: 48 rows of (kb - ka) keys
                    for (int ln = tid; ln < n_lines; ln += NT) touched ^= *(const UMGEN_GLOBAL u32*)(k0p + ((long)ln << 7));
                    const int per_dim = ((kb - ka) * 2 + 127) >> 7;
                    for (int i = tid; i < per_dim * kHeadDim; i += NT) {
                        const int d = i / per_dim, ln = i - d * per_dim;
                        touched ^= *(const UMGEN_GLOBAL u32*)(reinterpret_cast<const char*>(vtbase + (long)d * a.Lmax + ka) + ((long)ln << 7));
                    }
                } else
                for (int ln = tid; ln < n_lines; ln += NT) {
                    touched ^= *(const UMGEN_GLOBAL u32*)(k0p + ((long)ln << 7));
                    touched ^= *(const UMGEN_GLOBAL u32*)(v0p + ((long)ln << 7));
                }
            }
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
            stamp(0);   // waited for x
            // 16-key passes per register buffer, buffers (NB * KP * 16 keys of a wave in flight; a key's 96 bytes over 4 lanes: 16 + 8 each)
            constexpr int KP = UMGEN_ENG_KP, NB = SYS ? UMGEN_ENG_NB_SYS : UMGEN_ENG_NB;
            KVPiece kc[NB][KP], vc[NB][KP];
            auto kv_req = [&](int buf, int k0) {
#pragma unroll
                for (int i = 0; i < KP; ++i) {
                    const u32 off = (u32)min(k0 + KPW * i + kg, a.Lmax - 1) * (u32)kHeadDim;
                    kc[buf][i].a = ldwu(kbase, off + (u32)piece * 8u);
                    kc[buf][i].b = ldwu2(kbase, off + 32u + (u32)piece * 4u);
                    vc[buf][i].a = ldwu(vbase, off + (u32)piece * 8u);
                    vc[buf][i].b = ldwu2(vbase, off + 32u + (u32)piece * 4u);
                }
            };
            // matrix-core attention: a 32-key pass = K as the B operand of q . K^T (2 key tiles x 2 k-steps of 32 dims; the second k-step's
            // upper half is the zero padding 48..63 of q, any finite filler will do) + V^T as the B operand of P . V (3 dim tiles x 32 keys)
            // (the k-step-1 fragment is SHARED by the two key tiles: its lower half-wave holds tile 0's dims 32..47, its upper half-wave
            //  tile 1's -- a v_permlane32_swap brings the latter down; whatever sits in the other half meets q's zero padding)
            struct KVM { u32x4_t k0[2]; u32x4_t k1; u32x4_t v[3]; };
            constexpr int NBM = UMGEN_ENG_NBM;
            KVM km[NBM];
            auto kvm_req = [&](int buf, int k0) {
                const int cg = lane >> 4;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    const u32 key = (u32)min(k0 + 16 * kt + (lane & 15), a.Lmax - 1) * (u32)kHeadDim;
                    km[buf].k0[kt] = ldwu(kbase, key + 8u * cg);
                }
                {
                    const u32 key = (u32)min(k0 + 16 * (cg >> 1) + (lane & 15), a.Lmax - 1) * (u32)kHeadDim;
                    km[buf].k1 = ldwu(kbase, key + 32u + 8u * (cg & 1));
                }
#pragma unroll
                for (int dt = 0; dt < 3; ++dt) km[buf].v[dt] = ldwu(vtbase, (u32)(16 * dt + (lane & 15)) * (u32)a.Lmax + (u32)k0 + 8u * cg);
            };
            if (kMfmaA) {
                if (k_lo < k_hi) kvm_req(0, k_lo);      // (the other buffers behind P1: 28 VGPRs each, and the q|k|v rows are still live here)
            } else {
                if (k_lo < k_hi) kv_req(0, k_lo);
                if (NB > 1 && k_lo + KPW * KP < k_hi) kv_req(NB > 1 ? 1 : 0, k_lo + KPW * KP);
            }
            if (kMfmaQ) {
                typename Mma16<TT>::vec bx[3];
                f32x4_t acc[5];
                ln_split<TT>(xs, lnw, lane, wave, lds);
                load_bfrags<TT>(lds, L_XH, L_XL, 96 * wave, lane, bx);
                mfma_rows<TT, 5>(fq, bx, lane, acc);
                store_partials<5>(lds + L_PT + wave * 96, lane, acc);
                wg_barrier();
                if (tid < 72) {
                    float v = bq;
#pragma unroll
                    for (int ww = 0; ww < NW; ++ww) v += (lds + L_PT)[ww * 96 + tid];      // fixed order: k ranges 0, 1, ..., 7
                    const int n = 72 * w + tid;
                    put_local(gqkv, (u32)n, tg + 1, v);
                    if (n >= E) {   // K / V rows of the new token: 16 bits into the cache (head-major [2][H][Lmax][48])
                        const int cc = n - E, kvsel = cc / E, hc = cc % E;
                        (a.kvcache + (long)l * a.kv_layer_stride + (long)s * a.kv_scene_stride)[
                            (u32)(((kvsel * H + hc / kHeadDim) * a.Lmax + Lk) * kHeadDim + hc % kHeadDim)] = bits16<TT>(v);
                        if (kMfmaA && kvsel == 1)   // the same value dim-major for the matrix-core attention
                            (a.vtcache + (long)l * a.vt_layer_stride + (long)s * a.vt_scene_stride)[(u32)(hc * a.Lmax + Lk)] = bits16<TT>(v);
                    }
                }
            } else
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
                        if (kMfmaA && kvsel == 1)   // the same value dim-major for the matrix-core attention
                            (a.vtcache + (long)l * a.vt_layer_stride + (long)s * a.vt_scene_stride)[(u32)(hc * a.Lmax + Lk)] = bits16<TT>(v);
                    }
                }
            }
            if (kMfmaA) {
#pragma unroll
                for (int bfr = 1; bfr < NBM; ++bfr)
                    if (k_lo + bfr * 32 < k_hi) kvm_req(bfr, k_lo + bfr * 32);
            } else {
#pragma unroll
                for (int bfr = 2; bfr < NB; ++bfr)     // (the q|k|v rows' registers are free now: these fly while q | k | v are exchanged)
                    if (k_lo + bfr * KPW * KP < k_hi) kv_req(bfr, k_lo + bfr * KPW * KP);
            }
            u32x4_t wps[4];                        // SYS, first scene of a layer: staging of the parked mlp rows (three batches of 4 units)
            const bool late_park = SYS && kLatePark && load_w;
            if (late_park) {
#pragma unroll
                for (int j = 0; j < 4; ++j) wps[j] = ldwu(wp2 + (long)j * NT * 8, (u32)tid * 8u);
            }
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
                    const u32 src = (u32)((min(tid, 3 * kHeadDim - 1) / kHeadDim) * E + hh * kHeadDim + tid % kHeadDim);
                    poll_granules<1>(c, tid, gqkv, (tid >= lo && tid < hi) ? 1u : 0u, [&](int) { return src; }, tg + 1, qs);
                    wg_barrier();
                };
                poll_head(0, kQFirst ? kHeadDim : 3 * kHeadDim);
                stamp(2);   // waited for q_h (| k_h | v_h)
                if (kMfmaA) {
                    typedef typename Mma16<TT>::vec vec;
                    const int cg = lane >> 4, mrow = lane & 15;
                    // A operand of q . K^T: row 0 = q's hi parts, row 1 = its lo parts, rows 2..15 zero; k-step 1 holds dims 32..47 and zeros
                    vec qa[2];
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const bool valid = ks == 0 || cg < 2;
                        const float* qp = qs + (valid ? 32 * ks + 8 * cg : 0);
                        u32 pk[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            unsigned short h0, l0, h1, l1;
                            split16<TT>(qp[2 * e], h0, l0);
                            split16<TT>(qp[2 * e + 1], h1, l1);
                            const u32 hi = (u32)h0 | ((u32)h1 << 16), lo = (u32)l0 | ((u32)l1 << 16);
                            pk[e] = !valid ? 0u : (mrow == 0 ? hi : (mrow == 1 ? lo : 0u));
                        }
                        qa[ks] = __builtin_bit_cast(vec, u32x4_t{pk[0], pk[1], pk[2], pk[3]});
                    }
                    // the new token's own key / value (not in the caches yet): score and value row from the head's q | k | v exchange, 16 bits
                    const float own_d = lane < kHeadDim ? qs[lane] * round16<TT>(qs[kHeadDim + lane]) : 0.f;
                    const float s_own = wave_sum_all(own_d) * kScaleQK;
                    float vn[3];
#pragma unroll
                    for (int dt = 0; dt < 3; ++dt) vn[dt] = round16<TT>(qs[2 * kHeadDim + 16 * dt + mrow]);
                    float m_run = -INFINITY, l_run = 0.f;      // (wave-uniform)
                    f32x4_t oacc[3];
#pragma unroll
                    for (int dt = 0; dt < 3; ++dt) oacc[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    unsigned short* pstrip = reinterpret_cast<unsigned short*>(lds + L_PS + wave * 32);   // hi [32] | lo [32]
                    auto rowmax16 = [](float v) {   // over lanes 0..15 (every lane of the row ends with it)
                        v = fmaxf(v, dpp_mov<0xB1>(v)); v = fmaxf(v, dpp_mov<0x4E>(v)); v = fmaxf(v, dpp_mov<0x141>(v)); v = fmaxf(v, dpp_mov<0x140>(v));
                        return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
                    };
                    auto rowsum16 = [](float v) {
                        v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); v += dpp_mov<0x141>(v); v += dpp_mov<0x140>(v);
                        return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
                    };
                    auto pass = [&](const KVM& kv, int k0) {
                        f32x4_t s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
                        s0 = Mma16<TT>::mfma(qa[0], __builtin_bit_cast(vec, kv.k0[0]), s0);
                        s1 = Mma16<TT>::mfma(qa[0], __builtin_bit_cast(vec, kv.k0[1]), s1);
                        {
                            u32x4_t up;     // the upper half-wave's registers in every lane
#pragma unroll
                            for (int e = 0; e < 4; ++e) up[e] = __builtin_amdgcn_permlane32_swap(kv.k1[e], kv.k1[e], false, false)[1];
                            s0 = Mma16<TT>::mfma(qa[1], __builtin_bit_cast(vec, kv.k1), s0);
                            s1 = Mma16<TT>::mfma(qa[1], __builtin_bit_cast(vec, up), s1);
                        }
                        // lanes 0..15: rows 0 (q hi) + 1 (q lo) of key k0 + lane (tile 0) / k0 + 16 + lane (tile 1)
                        const int io = Lk - k0;                 // position of the new token's own key in this pass (if 0 <= io < 32)
                        float c0 = (s0[0] + s0[1]) * kScaleQK, c1 = (s1[0] + s1[1]) * kScaleQK;
                        if (io >= 0 && io < 16 && lane == io) c0 = s_own;
                        if (io >= 16 && io < 32 && lane == io - 16) c1 = s_own;
                        c0 = (lane < 16 && k0 + lane < k_hi) ? c0 : -INFINITY;
                        c1 = (lane < 16 && k0 + 16 + lane < k_hi) ? c1 : -INFINITY;
                        const float m_new = fmaxf(m_run, rowmax16(fmaxf(c0, c1)));
                        if (m_new > -INFINITY) {
                            const float scale = __expf(m_run - m_new);   // exp(-inf) = 0 on the first pass
                            float p0 = __expf(c0 - m_new), p1 = __expf(c1 - m_new);
                            l_run = fmaf(l_run, scale, rowsum16(p0 + p1));
                            float p_own = 0.f;
                            if (io >= 0 && io < 32 && Lk < k_hi) {        // (wave-uniform)
                                p_own = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(io < 16 ? p0 : p1), io & 15));
                                if (io < 16) p0 = (lane == io) ? 0.f : p0; else p1 = (lane == io - 16) ? 0.f : p1;   // its V row is not in the cache
                            }
                            if (lane < 16) {
                                unsigned short h, lo2;
                                split16<TT>(p0, h, lo2); pstrip[lane] = h; pstrip[32 + lane] = lo2;
                                split16<TT>(p1, h, lo2); pstrip[16 + lane] = h; pstrip[48 + lane] = lo2;
                            }
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            // A operand of P . V: row 0 = hi parts, row 1 = lo parts of the 32 probabilities (8 keys per lane group)
                            const unsigned char* pb = mrow == 0 ? reinterpret_cast<const unsigned char*>(pstrip)
                                                    : mrow == 1 ? reinterpret_cast<const unsigned char*>(pstrip + 32)
                                                                : reinterpret_cast<const unsigned char*>(lds + L_ZR);
                            const vec pa = *reinterpret_cast<const vec*>(pb + 16 * cg);
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the strip is rewritten by the next pass)
#pragma unroll
                            for (int dt = 0; dt < 3; ++dt) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) oacc[dt][r] *= scale;
                                oacc[dt] = Mma16<TT>::mfma(pa, __builtin_bit_cast(vec, kv.v[dt]), oacc[dt]);
                                oacc[dt][0] = fmaf(p_own, vn[dt], oacc[dt][0]);
                            }
                            m_run = m_new;
                        }
                    };
                    for (int k0 = k_lo; k0 < k_hi; k0 += 32 * NBM) {
#pragma unroll
                        for (int bfr = 0; bfr < NBM; ++bfr) {
                            if (k0 + 32 * bfr < k_hi) {
                                pass(km[bfr], k0 + 32 * bfr);
                                if (k0 + 32 * (NBM + bfr) < k_hi) kvm_req(bfr, k0 + 32 * (NBM + bfr));
                            }
                        }
                    }
                    // one partial per wave: (m, l, o[48] = rows 0 + 1 of the three dim tiles, lanes 0..15) -> LDS -> wave 0 merges the 8 and publishes
                    float* sm = lds + L_SM;
                    float* so = lds + L_SO;
                    if (lane == 0) { sm[wave] = m_run; sm[8 + wave] = l_run; }
                    if (lane < 16) {
#pragma unroll
                        for (int dt = 0; dt < 3; ++dt) so[wave * kHeadDim + 16 * dt + lane] = oacc[dt][0] + oacc[dt][1];
                    }
                    wg_barrier();
                    if (tid < kHeadDim) {
                        float M = sm[0];
#pragma unroll
                        for (int ww = 1; ww < NW; ++ww) M = fmaxf(M, sm[ww]);
                        float Ls = 0.f, o = 0.f;
#pragma unroll
                        for (int ww = 0; ww < NW; ++ww) {
                            const float e = (M > -INFINITY) ? __expf(sm[ww] - M) : 0.f;
                            Ls = fmaf(e, sm[8 + ww], Ls);
                            o = fmaf(e, so[ww * kHeadDim + tid], o);
                        }
                        u64* gp = gpart + (hh * 2 + half) * 50;
                        put_local(gp, (u32)tid, tg + 2, o);
                        if (tid == 0) { put_local(gp, 48u, tg + 2, M); put_local(gp, 49u, tg + 2, Ls); }
                    }
                } else {
                // this lane's 12 of the head's 48 dimensions: 8 piece .. 8 piece + 7 and 32 + 4 piece .. + 3, as 6 packed pairs
                auto dim_of = [&](int j) { return j < 4 ? piece * 8 + 2 * j : 32 + piece * 4 + 2 * (j - 4); };
                auto own16 = [&](const float* src, f32x2_t (&o)[6]) {   // the new token's own k / v, as the cache will hold it (16 bits)
#pragma unroll
                    for (int j = 0; j < 6; ++j) o[j] = f32x2_t{round16<TT>(src[dim_of(j)]), round16<TT>(src[dim_of(j) + 1])};
                };
                f32x2_t q2[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) q2[j] = f32x2_t{qs[dim_of(j)], qs[dim_of(j) + 1]};
                float m_run = -INFINITY, l_run = 0.f;
                f32x2_t o2[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) o2[j] = f32x2_t{0.f, 0.f};
                auto chunk = [&](const KVPiece (&kcb)[KP], const KVPiece (&vcb)[KP], int k0) {
                    float sc[KP];
                    float mc = -INFINITY;
#pragma unroll
                    for (int i = 0; i < KP; ++i) {
                        const int k = k0 + KPW * i + kg;
                        const u32 kw[6] = {kcb[i].a.x, kcb[i].a.y, kcb[i].a.z, kcb[i].a.w, kcb[i].b.x, kcb[i].b.y};
                        f32x2_t acc = {0.f, 0.f};
#pragma unroll
                        for (int j = 0; j < 6; ++j) acc = mac2<TT>(kw[j], q2[j], acc);
                        if (k == Lk) {   // the new token's own key is not in the cache yet: from the head's q | k | v exchange
                            f32x2_t kf[6];
                            own16(qs + kHeadDim, kf);
                            acc = f32x2_t{0.f, 0.f};
#pragma unroll
                            for (int j = 0; j < 6; ++j) acc = __builtin_elementwise_fma(kf[j], q2[j], acc);
                        }
                        float d = acc.x + acc.y;
                        d += dpp_xor1(d);
                        d += dpp_xor2(d);
                        d = (k < k_hi) ? d * kScaleQK : -INFINITY;
                        sc[i] = d;
                        mc = fmaxf(mc, d);
                    }
                    const float m_new = fmaxf(m_run, mc);
                    if (m_new > -INFINITY) {
                        const float scale = __expf(m_run - m_new);   // exp(-inf) = 0 on the first chunk
                        const f32x2_t scale2 = {scale, scale};
                        l_run *= scale;
#pragma unroll
                        for (int j = 0; j < 6; ++j) o2[j] *= scale2;
#pragma unroll
                        for (int i = 0; i < KP; ++i) {
                            const int k = k0 + KPW * i + kg;
                            const u32 vw[6] = {vcb[i].a.x, vcb[i].a.y, vcb[i].a.z, vcb[i].a.w, vcb[i].b.x, vcb[i].b.y};
                            const float p = __expf(sc[i] - m_new);
                            const f32x2_t p2 = {p, p};
                            l_run += p;
                            if (k == Lk) {
                                f32x2_t vf[6];
                                own16(qs + 2 * kHeadDim, vf);
#pragma unroll
                                for (int j = 0; j < 6; ++j) o2[j] = __builtin_elementwise_fma(p2, vf[j], o2[j]);
                            } else {
#pragma unroll
                                for (int j = 0; j < 6; ++j) o2[j] = mac2<TT>(vw[j], p2, o2[j]);
                            }
                        }
                        m_run = m_new;
                    }
                };
                // chunks of KPW * KP keys in NB register buffers, all requested before q|k|v were exchanged; a buffer is requested again as
                // soon as it has been consumed
                constexpr int CK = KPW * KP;
                for (int k0 = k_lo; k0 < k_hi; k0 += CK * NB) {
#pragma unroll
                    for (int bfr = 0; bfr < NB; ++bfr) {
                        if (k0 + CK * bfr < k_hi) {
                            chunk(kc[bfr], vc[bfr], k0 + CK * bfr);
                            if (k0 + CK * (NB + bfr) < k_hi) kv_req(bfr, k0 + CK * (NB + bfr));
                        }
                    }
                }
                if (kQFirst) {
                    // the new token's own key: k_h | v_h have been on their way since P1; one wave of the head's second half adds the
                    // key as one more term of its online softmax (as the cache will hold it: rounded to 16 bits)
                    poll_head(kHeadDim, 3 * kHeadDim);
                    if (half == 1 && wave == NW - 1) {
                        k_hi = Lk + 1;
                        KVPiece kz[KP], vz[KP];
#pragma unroll
                        for (int i = 0; i < KP; ++i) { kz[i] = KVPiece{u32x4_t{0, 0, 0, 0}, u32x2_t{0, 0}}; vz[i] = kz[i]; }
                        chunk(kz, vz, Lk);
                    }
                }
                if (late_park) {   // batch 1 has arrived during the key loop: park it, request batch 2 (it flies during the merge and P3's gather)
#pragma unroll
                    for (int j = 0; j < 4; ++j) w2p[j * NT] = wps[j];
#pragma unroll
                    for (int j = 0; j < 4; ++j) wps[j] = ldwu(wp2 + (long)(4 + j) * NT * 8, (u32)tid * 8u);
                }
                // the wave's 16 lane groups fold to 8 (group kg + 8 into group kg: lanes l + 32 into l, fixed order), then the 64 partials
                // of this CU -> LDS -> one half partial (m, l, o[48]) published by wave 0
                {
                    // v_permlane32_swap of a value with itself: (the lower half-wave's values, the upper half-wave's) in every lane
                    auto halves = [](float v, float& lo, float& hi) {
                        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                        lo = __uint_as_float(r[0]); hi = __uint_as_float(r[1]);
                    };
                    float m_a, m_b, l_a, l_b;
                    halves(m_run, m_a, m_b);
                    halves(l_run, l_a, l_b);
                    const float M2 = fmaxf(m_a, m_b);
                    const float e_a = (M2 > -INFINITY) ? __expf(m_a - M2) : 0.f, e_b = (M2 > -INFINITY) ? __expf(m_b - M2) : 0.f;
                    l_run = fmaf(e_b, l_b, e_a * l_a);
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        float xa, xb2, ya, yb;
                        halves(o2[j].x, xa, xb2);
                        halves(o2[j].y, ya, yb);
                        o2[j] = f32x2_t{fmaf(e_b, xb2, e_a * xa), fmaf(e_b, yb, e_a * ya)};
                    }
                    m_run = M2;
                }
                float* sm = lds + L_SM;
                float* so = lds + L_SO;
                const int gi = wave * 8 + (kg & 7);
                if (lane < 32) {
                    if (piece == 0) { sm[gi] = m_run; sm[64 + gi] = l_run; }
#pragma unroll
                    for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x2_t*>(so + gi * kHeadDim + dim_of(j)) = o2[j];
                }
                wg_barrier();
                {
                    // every wave recomputes the 64 merge weights (cheap), then thread (jg, d) folds 8 of the 64 partial rows of
                    // column d; 48 threads add the 8 folds in a fixed order and publish the half partial (m, l, o[48])
                    const float mg = sm[lane];
                    const float M = wave_max_all(mg);
                    const float wg = (M > -INFINITY) ? __expf(mg - M) : 0.f;
                    const float Ls = wave_sum_all(wg * sm[64 + lane]);
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
            }
            stamp(3);   // attention of this CU's half
            // ================= P3: merge the halves -> c_proj -> x' =================
            gather<4>(c, tid, gpart, 2 * H * 50, tg + 2, lds + L_GP);
            stamp(4);   // waited for the half partials
            if (late_park) {   // batch 2 -> LDS, batch 3 requested (with the mlp rows' last 6 units below: parked behind P4's gather)
#pragma unroll
                for (int j = 0; j < 4; ++j) w2p[(4 + j) * NT] = wps[j];
#pragma unroll
                for (int j = 0; j < 4; ++j) wps[j] = ldwu(wp2 + (long)(8 + j) * NT * 8, (u32)tid * 8u);
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) wpl[j] = ldwu(wp2 + (long)(12 + j) * NT * 8, (u32)tid * 8u);
            if (kMfmaF && kLateTile && !SYS) req_frags<6, false, 5, 6>(ff, lw.Wfc, 96 * w, 96, wave, lane);
            {
                const float* gp = lds + L_GP;
                for (int col = tid; col < E; col += NT) {
                    const int h2 = col / kHeadDim, d = col % kHeadDim;
                    const float* p0 = gp + (h2 * 2) * 50;
                    const float* p1 = p0 + 50;
                    const float m0 = p0[48], m1 = p1[48];
                    const float M = fmaxf(m0, m1);
                    const float e0 = (m0 > -INFINITY) ? __expf(m0 - M) : 0.f, e1 = (m1 > -INFINITY) ? __expf(m1 - M) : 0.f;
                    const float Ls = fmaf(e1, p1[49], e0 * p0[49]);
                    const float av = fmaf(e1, p1[d], e0 * p0[d]) * __builtin_amdgcn_rcpf(Ls);
                    if (kMfmaO) {
                        unsigned short hi, lo;
                        split16<TT>(av, hi, lo);
                        reinterpret_cast<unsigned short*>(lds + L_XH)[col] = hi;
                        reinterpret_cast<unsigned short*>(lds + L_XL)[col] = lo;
                    } else {
                        as[col] = av;
                    }
                }
                wg_barrier();
                if (kMfmaO) {
                    typename Mma16<TT>::vec bx[3];
                    f32x4_t acc[2];
                    load_bfrags<TT>(lds, L_XH, L_XL, 96 * wave, lane, bx);
                    mfma_rows<TT, 2>(fo, bx, lane, acc);
                    store_partials<2>(lds + L_PT + wave * 96, lane, acc);
                    wg_barrier();
                    if (tid < 24) {
                        float v = 0.f;
#pragma unroll
                        for (int ww = 0; ww < NW; ++ww) v += (lds + L_PT)[ww * 96 + tid];
                        const int n = 24 * w + tid;
                        put_local(gxb, (u32)n, tg + 3, xs[n] + (v + bo));
                    }
                } else {
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
            }
            stamp(5);   // merge + c_proj rows
            // ================= P4: x' -> LN -> c_fc -> GELU -> this CU's partial sums of the mlp c_proj =================
            gather<2>(c, tid, gxb, E, tg + 3, xb);
            stamp(6);   // waited for x'
            if (late_park) {
#pragma unroll
                for (int j = 0; j < 4; ++j) w2p[(8 + j) * NT] = wps[j];
            }
            float* hsl = lds + L_HS;              // [96] gelu(c_fc) of this CU's hidden units
            float* hrow = hsl + 96;               // [256] second halves of the shared rows 512..767
            float* part = hrow + 256;             // [32][24] gathered partial sums (P5)
            if (kMfmaF) {
                typename Mma16<TT>::vec bx[3];
                ln_split<TT>(xb, lnw + E, lane, wave, lds);
                load_bfrags<TT>(lds, L_XH, L_XL, 96 * wave, lane, bx);
                {   // two passes of three tiles: 12 accumulator registers live instead of 24
                    typedef typename Mma16<TT>::vec vec;
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        f32x4_t a3[3];
#pragma unroll
                        for (int t = 0; t < 3; ++t) a3[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int j = 0; j < 3; ++j)
#pragma unroll
                            for (int t = 0; t < 3; ++t) {
                                const u32x4_t fr = (kParkFT && 3 * hf + t == 5) ? reinterpret_cast<const u32x4_t*>(lds + L_FT)[j * NT + tid] : ff.f[3 * hf + t][j];
                                a3[t] = Mma16<TT>::mfma(__builtin_bit_cast(vec, fr), bx[j], a3[t]);
                            }
#pragma unroll
                        for (int t = 0; t < 3; ++t) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) a3[t][r] += dpp_mov<0x101>(a3[t][r]);
                            if ((lane & 15) == 0) *reinterpret_cast<f32x4_t*>(lds + L_PT + wave * 96 + 16 * (3 * hf + t) + 4 * (lane >> 4)) = a3[t];
                        }
                    }
                }
                wg_barrier();
                if (tid < 96) {   // hidden unit 96 w + tid: sum of the 8 k ranges -> exact GELU -> hi / lo for the mlp projection's B operand
                    float v = 0.f;
#pragma unroll
                    for (int ww = 0; ww < NW; ++ww) v += (lds + L_PT)[ww * 96 + tid];
                    const float hv = gelu_erf(v);
                    if (kMfmaP) {
                        unsigned short hi, lo;
                        split16<TT>(hv, hi, lo);
                        reinterpret_cast<unsigned short*>(lds + L_HH)[tid] = hi;
                        reinterpret_cast<unsigned short*>(lds + L_HH)[96 + tid] = lo;
                    } else {
                        hsl[tid] = hv;
                    }
                }
            } else {
                f32x2_t x1[4], x2[4];
                float out[RF];
                ln768(xb, lnw + E, lane, x1, x2);
                dot768<TT, RF>(wf, x1, x2, lane, out);
                float v = 0.f;
#pragma unroll
                for (int r = 0; r < RF; ++r) v = (lane == r) ? out[r] : v;
                if (lane < RF) {
                    const float hv = gelu_erf(v);
                    if (kMfmaP) {
                        unsigned short hi, lo;
                        split16<TT>(hv, hi, lo);
                        reinterpret_cast<unsigned short*>(lds + L_HH)[wave * RF + lane] = hi;
                        reinterpret_cast<unsigned short*>(lds + L_HH)[96 + wave * RF + lane] = lo;
                    } else {
                        hsl[wave * RF + lane] = hv;
                    }
                }
            }
            wg_barrier();
            stamp(11);  // LN + c_fc rows + GELU
#ifndef UMGEN_SYS_WQ_AHEAD
#define UMGEN_SYS_WQ_AHEAD 0
#endif
            if (SYS && UMGEN_SYS_WQ_AHEAD) {
                // EXPERIMENT, off: a busy systolic group has no idle wait in front of its next item -- its q|k|v rows (which no register
                // can hold through the attention) are requested at the item's start and P1 waits for them (L2 latency + 110 KB per CU).
                // Requesting them HERE, one item ahead, keeps 56 more VGPRs live over the loop's back edge: 18 spilled VGPRs with 3 K/V
                // buffers (8 scenes: 909 vs 730 us per launch), 8 with 2 (803), none with 1 (740): never a gain
                // (profiles/r03_engine_experiments.txt, session I).
                wq_ahead = item + 1 < n_items;
                if (wq_ahead) {
                    const bool tail2 = item + 1 >= n_full * a.B;
                    const int l2 = tail2 ? tail_l : q + D * ((item + 1) / a.B);
                    req768_rows<RQ, true>(wq, a.layers[l2].Wqkv, qkv_row, lane);
                }
            }
            if (kMfmaP) {
                // this CU's partial sums of the 768 mlp c_proj outputs over its 96 hidden units: wave w takes rows 96 w .. 96 w + 95 (6 tiles
                // of 16) x all 96 k (3 k-steps); the 18 fragments are the thread's 18 repacked units (12 parked in LDS, 6 in registers).
                // The results sit in 4 lanes x 4 rows per tile: they go through the wave's strip of the partial-sum buffer so that every
                // lane publishes one or two rows (24 store instructions with 4 active lanes each cost more than the 18 MFMAs)
                typename Mma16<TT>::vec bh[3];
                typedef typename Mma16<TT>::vec vec;
                load_bfrags<TT>(lds, L_HH, L_HH + 48, 0, lane, bh);
                float* strip = lds + L_PT + wave * 96;
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int f = 3 * t + j;
                        const u32x4_t wfrag = f < 12 ? w2p[f * NT] : wpl[f < 12 ? 0 : f - 12];
                        acc = Mma16<TT>::mfma(__builtin_bit_cast(vec, wfrag), bh[j], acc);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] += dpp_mov<0x101>(acc[r]);
                    if ((lane & 15) == 0) *reinterpret_cast<f32x4_t*>(strip + 16 * t + 4 * (lane >> 4)) = acc;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the wave's own strip: no workgroup barrier)
                u64* mine = gpy + (long)w * E;
                put_local(mine, (u32)(96 * wave + lane), tg + 4, strip[lane]);
                if (lane < 32) put_local(mine, (u32)(96 * wave + 64 + lane), tg + 4, strip[64 + lane]);
            } else {
                // this CU's partial sums of the 768 mlp c_proj outputs over its 96 hidden units.  Four lanes share four rows: thread t
                // multiplies rows 4 (t / 4) .. + 3 by columns 24 (t % 4) .. + 23 (units 0..11, parked in LDS) and -- eight lanes per four
                // rows -- rows 512 + 4 (t / 8) .. + 3 by columns 12 (t % 8) .. + 11 (units 12..17); transposed quad sums leave row t's
                // total in thread t.  (Round 2: thread t = all 96 columns of row t.  Every lane then read all 96 h values -- 36
                // broadcast ds_read_b128 per wave, 8 clocks each whatever the addresses: ~1 us of LDS time per item; now 9.)
                const f32x2_t zero = {0.f, 0.f};
                float ya[16], yb[16];
                {
                    f32x2_t hq[3][4];
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) load8p(hsl + 24 * (tid & 3) + 8 * cc, hq[cc]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        f32x2_t acc = zero;
#pragma unroll
                        for (int cc = 0; cc < 3; ++cc) acc = dot8<TT>(w2p[(3 * r + cc) * NT], hq[cc], acc);
                        ya[r] = acc.x + acc.y;
                        // SYS keeps the c_proj / c_fc rows (92 VGPRs) live through this phase: stop the scheduler from hoisting all the LDS
                        // reads to the front, which pushed those rows out to scratch memory
                        if (SYS) __builtin_amdgcn_sched_barrier(0);
                    }
                }
                {
                    f32x2_t h2[6], wv[24];
                    const float4* hp = reinterpret_cast<const float4*>(hsl + 12 * (tid & 7));
#pragma unroll
                    for (int i = 0; i < 3; ++i) { const float4 v = hp[i]; h2[2 * i] = f32x2_t{v.x, v.y}; h2[2 * i + 1] = f32x2_t{v.z, v.w}; }
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        wv[4 * j] = up2<TT>(wpl[j].x); wv[4 * j + 1] = up2<TT>(wpl[j].y); wv[4 * j + 2] = up2<TT>(wpl[j].z); wv[4 * j + 3] = up2<TT>(wpl[j].w);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        f32x2_t acc = zero;
#pragma unroll
                        for (int i = 0; i < 6; ++i) acc = __builtin_elementwise_fma(wv[6 * r + i], h2[i], acc);
                        yb[r] = acc.x + acc.y;
                    }
                }
                rows_step<4, 4, 0x4E>(ya, (tid & 2) != 0);
                rows_step<2, 2, 0xB1>(ya, (tid & 1) != 0);
                rows_step<4, 4, 0x4E>(yb, (tid & 2) != 0);
                rows_step<2, 2, 0xB1>(yb, (tid & 1) != 0);
                const float yB = yb[0] + dpp_mov<0x104>(yb[0]);     // row_shl:4: lane i + 4 (same row of the group's other four lanes)
                u64* mine = gpy + (long)w * E;
                put_local(mine, (u32)tid, tg + 4, ya[0]);
                if ((tid & 7) < 4) put_local(mine, 512u + 4u * (u32)(tid >> 3) + (u32)(tid & 7), tg + 4, yB);
            }
            stamp(7);   // LN + c_fc rows + partial sums
            // ================= P5: the 32 partial sums of this CU's 24 rows -> x'' (next layer's x) =================
            // producer p's partials of rows 24 w .. 24 w + 23 sit at gpy[p * 768 + 24 w + r]: 768 granules in 32 runs of 192 B
            poll_granules<2>(c, tid, gpy + 24 * w, (tid < 256) ? 3u : 1u,
                             [&](int k) { const u32 i = (u32)min(tid + k * NT, E - 1); return (i / 24u) * (u32)E + i % 24u; }, tg + 4, part);
            wg_barrier();
            stamp(8);   // waited for the partial sums
            if (tid < 96) {
                // row tid / 4: four lanes add 8 producers each (producer 8 j, ..., 8 j + 7), then the quad (fixed order)
                const int row = tid >> 2, j8 = (tid & 3) * 8;
                float sum = 0.f;
#pragma unroll
                for (int p = 0; p < 8; ++p) sum += part[(j8 + p) * 24 + row];
                sum += dpp_xor1(sum);
                sum += dpp_xor2(sum);
                const int n = 24 * w + row;
                const float xn = xb[n] + sum;
                if ((tid & 3) == 0) {
                if (l + 1 == a.n_layers) {
                    (a.xdec + (long)s * E)[(u32)n] = xn;
                    // the engine part of this launch ends here: the background workers of the next launch plan against this duration
                    if (BG && a.bg != nullptr && w == 0 && tid == 0) a.bg->engine_ticks = (u32)(wall_clock64() - t_k0);
                }
                else if (D == 1) put_local(gxl, (u32)n, tg + 8, xn);
                else put_far(a.gx + (long)s * E, (u32)n, tg + 8, xn);
                }
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

size_t oar_engine_bg_lds_bytes() { return std::max(oar_engine_lds_bytes(), (size_t)kLds256); }

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
    for (const void* f : {reinterpret_cast<const void*>(oar_engine_kernel<false, bf16_t, false, true>), reinterpret_cast<const void*>(oar_engine_kernel<false, f16_t, false, true>)}) {
        hipError_t rc = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)oar_engine_bg_lds_bytes());
        if (rc != hipSuccess) return rc;
    }
    return hipSuccess;
}

template <typename TT, bool SYS>
static void launch_engine_t(hipStream_t s, const OarEngineArgs& a) {
    const dim3 grid(a.NG * kEngGroup), block(kEngThreads);
    const size_t shm = oar_engine_lds_bytes();
    if (!SYS && a.bg != nullptr && !a.stamps) {      // with background workers: the workers' GEMM tiles need the whole 160 KB
        hipLaunchKernelGGL((oar_engine_kernel<false, TT, false, true>), grid, block, oar_engine_bg_lds_bytes(), s, a);
        return;
    }
    if (a.stamps) hipLaunchKernelGGL((oar_engine_kernel<true, TT, SYS>), grid, block, shm, s, a);
    else hipLaunchKernelGGL((oar_engine_kernel<false, TT, SYS>), grid, block, shm, s, a);
}

hipError_t launch_oar_engine(hipStream_t s, const OarEngineArgs& a) {
    if (a.fp16) { if (a.systolic) launch_engine_t<f16_t, true>(s, a); else launch_engine_t<f16_t, false>(s, a); }
    else { if (a.systolic) launch_engine_t<bf16_t, true>(s, a); else launch_engine_t<bf16_t, false>(s, a); }
    return hipGetLastError();
}

}  // namespace umgen
