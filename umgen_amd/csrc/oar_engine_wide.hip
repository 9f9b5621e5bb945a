// Chip-wide decode engine for WIDE BlockOAR layers (configs[4]: n_embd 1536, 32 heads): the n_oar_layer layers of one decode step (module.py:378-428)
// of ONE scene in one persistent launch of 256 workgroups (one per CU).
//
// At 2x width a layer is 57 MB: through ONE XCD's memory link (the XCD-resident engine, oar_engine.hip) that is 45-85 us per layer, as five launches
// (gemv.hip) 33.5 us of which 11 us are the weight stream and the rest kernel boundaries.  Here all 256 CUs work on every layer: 23.3 us per layer
// (profiles/r05_wide2x_engine.txt: every step from 33 to 23 us with its measurement).
//
// OWNERSHIP.  Rank r = 32 x XCD + arrival order on the XCD (per-XCD tickets; the census at umgen_create has found 32 workgroups on each of the 8 XCDs).
// It owns 18 of the 576 q|k|v rows of ITS XCD's four heads (heads 4 g .. 4 g + 3 on XCD g), c_proj rows 6 r .., the hidden units 24 r .. of c_fc and the
// matching 24 columns of the mlp c_proj (hidden-unit split: what is exchanged are the ranks' partial sums of the 1536 outputs).  Ranks 0..15 of an XCD are
// the attention ranks of its heads (head = rank / 4, key quarter = rank % 4; the six compute waves split the quarter's keys).
//
// ROLES.  The first form of this engine ran at the five launches' 33 us per layer: a wave's loads return in order, so a hand-off poll's s_waitcnt also
// waited for the weight requests in front of it.  Wait counters are PER WAVE: waves 6, 7 of a workgroup are POLL waves that issue nothing but granule
// polls (and the layer's LayerNorm weights) and hand the gathered vectors over through LDS + the workgroup barrier (which does not drain vmcnt,
// oar_common.h wg_barrier); waves 0..5 are COMPUTE waves that never poll.  The two roles are two loops over the layers that execute the same nine
// barriers per layer.
//   per layer: x -> [P1 LN + 18 q|k|v rows] -> q|k|v (XCD-local) -> [attention of a key quarter] -> 4 partials per head (XCD-local) -> [owner merges]
//              -> attention output -> [P3 c_proj + residual] -> x' -> [P4 LN + 24 hidden units + GELU + partial sums] -> 256 partials per row -> [P5 adds] -> x
//   Hand-offs are 8-byte {tag, value} granules: inside the XCD's L2 0.8-1.0 us, across the fabric 2.1-2.3 us PROVIDED the producer writes whole 64-byte
//   pieces with one store instruction (x and x' live in one 128-byte line per rank for that; 8-byte stores of six waves into shared sectors: 3.5-5.3 us).
//   Requests: a poll's answer queues behind whatever the XCD's link still has to deliver -- the weight groups go out as LATE as their consumer allows
//   (UMGEN_WIDE_AT_* slots below), not as early as their registers are free.
// Arithmetic (fixed, independent of the placement): fp32 activations, 16-bit weights and K/V cache, fp32 accumulation; row dot products as in gemv.hip
// (lane l owns k = 512 c + 8 l .. + 7, packed fp32 FMAs, wave sum); weight-only LayerNorm (eps 1e-5) with gemv.hip's statistics; exact erf-GELU; the
// attention's new key / value out of the q|k|v exchange rounded to 16 bits.  Against the five-launch form only fp32 summation orders differ.
#include "oar_common.h"

namespace umgen {

namespace {

constexpr int WE = kWideE, WH = kWideE / kHeadDim, WF = 4 * kWideE;
constexpr int NWG = kWideGroups;                 // 256 workgroups = ranks
constexpr int WT = kEngThreads;                  // 512 threads
constexpr int CW = 6, PW = 2;                    // compute waves, poll waves
constexpr int CT = CW * 64, PT = PW * 64;        // 384 compute threads, 128 poll threads
constexpr int KC = WE / 512;                     // 16-byte weight chunks per lane and row (3)
constexpr int WRQ = 3 * WE / NWG;                // 18 q|k|v rows per rank: 3 per compute wave
constexpr int WRO = WE / NWG;                    // 6 c_proj rows: 1 per compute wave
constexpr int WRF = WF / NWG;                    // 24 hidden units: 4 per compute wave
constexpr int NSP = kWideSplits;                 // key splits per head (H x NSP attention ranks)
constexpr int NXCD = 8, RPX = NWG / NXCD;        // XCDs, ranks per XCD
constexpr int HPX = WH / NXCD;                   // heads per XCD (4): RPX / 2 attention ranks per XCD
constexpr int KG = NSP == 8 ? 3 : 5;                            // 16-key passes of a wave in flight at once
constexpr int PREC = 52;                         // floats per attention partial record: o[48] | m | l | pad
static_assert(WE == 1536 && WRQ == 3 * CW && WRO == CW && WRF == 4 * CW && WE == 4 * CT && WH * NSP <= NWG && (NSP == 4 || NSP == 8), "wide engine geometry");
static_assert(WE == 12 * PT, "gathers of 1536 granules: 12 per poll lane");

// LDS carve (floats)
constexpr int W_XS = 0;                       // x of the layer [1536]
constexpr int W_XB = W_XS + WE;               // x' [1536]
constexpr int W_AS = W_XB + WE;               // attention output [1536]
constexpr int W_LN = W_AS + WE;               // ln_1 | ln_2 weights of the layer [2][1536]
constexpr int W_PT = W_LN + 2 * WE;           // gathered mlp partial sums [256][6]
constexpr int W_SB = W_PT + NWG * WRO;        // the four key quarters' partials of this owner's head [4][52]
constexpr int W_QS = W_SB + NSP * PREC;       // q_h | k_h | v_h [144 -> 160]
constexpr int W_WP = W_QS + 160;              // the compute waves' attention partials [6][52]
constexpr int W_HS = W_WP + CW * PREC;        // gelu(c_fc) of this rank's 24 hidden units [24 -> 32]
constexpr int W_XO = W_HS + 32;              // the six c_proj results of the workgroup (padded x' layouts) [16]
constexpr int W_MISC = W_XO + 16;
constexpr int W_TOTAL = W_MISC + 16;

// Hand-off polls of the poll waves: 0 = request everything once, then one missing granule per lane until it is there, then everything missing again
// (two fabric round trips behind the producers); 1 = every round requests every slot again (one round trip behind); 2 = the far hand-offs (x, attention
// output, x', mlp partial sums) with two requests of every slot in flight, UMGEN_WIDE_STAGGER x 64 clocks apart (oar_common.h poll_stag)
#ifndef UMGEN_WIDE_POLL
#define UMGEN_WIDE_POLL 1
#endif
constexpr bool kPollAll = UMGEN_WIDE_POLL != 0;
constexpr bool kPollStag = UMGEN_WIDE_POLL == 2;
#ifndef UMGEN_WIDE_STAGGER
#define UMGEN_WIDE_STAGGER 24
#endif
template <typename IDX, typename SINK>
__device__ inline void poll_far(Ctx& c, int tid, const u64* g, IDX idx, u32 tag, SINK sink) {
    if (kPollStag) poll_stag<12>(c, tid, g, 0xfffu, idx, tag, UMGEN_WIDE_STAGGER, sink);
    else poll_ms<12, kPollAll>(c, tid, g, 0xfffu, idx, tag, sink);
}
// Measurement builds (tools/build_variant.sh; results are garbage, only the step time means something):
//   UMGEN_WIDE_EXP_NOPOLL: the poll waves do not wait (a layer without its six fabric hops)
//   UMGEN_WIDE_EXP_NOLOAD: no weight / K/V request is made (a layer without its 57 MB)
#ifndef UMGEN_WIDE_EXP_NOPOLL
#define UMGEN_WIDE_EXP_NOPOLL 0
#endif
#ifndef UMGEN_WIDE_EXP_NOLOAD
#define UMGEN_WIDE_EXP_NOLOAD 0
#endif
// Granules per rank in the x / x' buffers: 8 / 16 = every rank writes whole 64 / 128-byte pieces with ONE store instruction (x': its six rows' results
// collected through LDS behind one more workgroup barrier).  Compact (6 per rank: 48 bytes that share 64-byte sectors with the neighbours', written by six
// waves' lane 0 one after the other) the x' hand-off took 5.3 us instead of 2.5 (profiles/r05_wide2x_engine.txt).
#ifndef UMGEN_WIDE_PAD
#define UMGEN_WIDE_PAD 16
#endif
// Where the requests of a layer go out -- a poll's answer queues behind whatever the XCD's memory link still has to deliver, and a compute wave's loads
// return in the order they were made.  Slots: 0 before B1 (under the x hand-off), 1 behind B1, 2 behind P1 (under q|k|v), 3 behind the keys, 4 behind B3's
// merge (under the quarters), 5 behind B4's merge (under the attention output), 6 behind B5, 7 behind B6 (x' is there), 8 behind B7, 9 behind B8.
// Measured (profiles/r05_wide2x_engine.txt): K/V + the c_proj row at 0, the c_fc rows at 3, the mlp slice at 5, the next layer's q|k|v rows at 7 -- "as late
// as the registers' consumer allows" beats "as early as the registers are free": what is in the link's queue delays the polls.
#ifndef UMGEN_WIDE_AT_KV
#define UMGEN_WIDE_AT_KV 0
#endif
#ifndef UMGEN_WIDE_AT_O
#define UMGEN_WIDE_AT_O 0
#endif
#ifndef UMGEN_WIDE_AT_F
#define UMGEN_WIDE_AT_F 3
#endif
#ifndef UMGEN_WIDE_AT_P
#define UMGEN_WIDE_AT_P 5
#endif
#ifndef UMGEN_WIDE_AT_Q
#define UMGEN_WIDE_AT_Q 7
#endif
constexpr int XPAD = UMGEN_WIDE_PAD;
constexpr int XGR = 256 * 16;      // granules reserved for each of the x / x' buffers
static_assert(XPAD == 8 || XPAD == 16, "x / x' granules per rank");
__device__ inline u32 xslot(u32 n) { return (n / 6u) * (u32)XPAD + n % 6u; }
__device__ inline u32x4_t wld(const bf16_t* ubase, u32 off) {
    if (UMGEN_WIDE_EXP_NOLOAD) { u32x4_t z; asm volatile("" : "=v"(z)); return z; }
    return ldwu(ubase, off);
}

struct WRow { u32x4_t c[KC]; };
__device__ inline void req_row(WRow& w, const bf16_t* W, long row, int lane) {
#pragma unroll
    for (int c = 0; c < KC; ++c) w.c[c] = wld(W + row * WE, (u32)(512 * c + 8 * lane));
}
struct XRegs { f32x2_t v[KC][4]; };
// lane l's values k = 512 c + 8 l .. + 7 of a 1536-vector
__device__ inline void load_x(const float* xs, int lane, XRegs& x) {
#pragma unroll
    for (int c = 0; c < KC; ++c) load8p(xs + 512 * c + 8 * lane, x.v[c]);
}
// weight-only LayerNorm (module.py:26-37) of the whole row held by the wave (every wave computes the statistics itself), gemv.hip's form
__device__ inline void ln_regs(XRegs& x, const float* lnw, int lane) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += x.v[c][e].x + x.v[c][e].y;
    const float mean = wave_sum_all(s) / (float)WE;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d0 = x.v[c][e].x - mean, d1 = x.v[c][e].y - mean; q += d0 * d0 + d1 * d1; }
    const float rstd = 1.0f / sqrtf(wave_sum_all(q) / (float)WE + 1e-5f);
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        f32x2_t lw[4];
        load8p(lnw + 512 * c + 8 * lane, lw);
#pragma unroll
        for (int e = 0; e < 4; ++e) x.v[c][e] = f32x2_t{(x.v[c][e].x - mean) * rstd * lw[e].x, (x.v[c][e].y - mean) * rstd * lw[e].y};
    }
}
template <typename TT>
__device__ inline float row_dot(const WRow& w, const XRegs& x) {
    f32x2_t acc = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < KC; ++c) acc = dot8<TT>(w.c[c], x.v[c], acc);
    return wave_sum_all(acc.x + acc.y);
}

// online-softmax state of a lane (its 12 of the head's 48 dimensions) and its folds
struct WAtt { float m, l; f32x2_t o[6]; };
__device__ inline void watt_merge(WAtt& a, float mb, float lb, const f32x2_t (&ob)[6]) {
    const float M = fmaxf(a.m, mb);
    const float ea = (M > -INFINITY) ? __expf(a.m - M) : 0.f, eb = (M > -INFINITY) ? __expf(mb - M) : 0.f;
    a.l = fmaf(eb, lb, ea * a.l);
#pragma unroll
    for (int j = 0; j < 6; ++j) a.o[j] = f32x2_t{fmaf(eb, ob[j].x, ea * a.o[j].x), fmaf(eb, ob[j].y, ea * a.o[j].y)};
    a.m = M;
}
template <int CTRL>
__device__ inline void watt_fold_dpp(WAtt& a) {
    f32x2_t ob[6];
    const float mb = dpp_mov<CTRL>(a.m), lb = dpp_mov<CTRL>(a.l);
#pragma unroll
    for (int j = 0; j < 6; ++j) ob[j] = f32x2_t{dpp_mov<CTRL>(a.o[j].x), dpp_mov<CTRL>(a.o[j].y)};
    watt_merge(a, mb, lb, ob);
}
template <bool ROW32>
__device__ inline void watt_fold_swap(WAtt& a) {
    auto sw = [](float v, float& x, float& y) {
        if (ROW32) { auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false); x = __uint_as_float(r[0]); y = __uint_as_float(r[1]); }
        else { auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false); x = __uint_as_float(r[0]); y = __uint_as_float(r[1]); }
    };
    WAtt x, y;
    sw(a.m, x.m, y.m);
    sw(a.l, x.l, y.l);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float x0, y0, x1, y1;
        sw(a.o[j].x, x0, y0);
        sw(a.o[j].y, x1, y1);
        x.o[j] = f32x2_t{x0, x1};
        y.o[j] = f32x2_t{y0, y1};
    }
    watt_merge(x, y.m, y.l, y.o);
    a = x;
}

}  // namespace

template <bool STAMPS, typename TT>
__global__ __launch_bounds__(kEngThreads) void oar_engine_wide_kernel(OarWideArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    u32* ldsu = reinterpret_cast<u32*>(lds);
    const int tid0 = threadIdx.x;
    // rank = 32 x XCD + arrival order on the XCD (the census, engine.hip, has found 32 workgroups on each of the 8 XCDs): the q|k|v rows of heads 4 g .. 4 g + 3
    // and their attention are on XCD g, so the q|k|v hand-off and the key quarters' merge stay inside the XCD's L2
    const int xg = (int)xcc_id() & (NXCD - 1);
    if (tid0 == 0) ldsu[W_MISC] = atomicAdd(a.ticket + xg, 1u) & (u32)(RPX - 1);
    wg_barrier();
    const int r0 = __builtin_amdgcn_readfirstlane(RPX * xg + (int)ldsu[W_MISC]);     // this workgroup's rank
    // (wave-uniform, and the compiler is told so: the two roles are two loops over the layers, branched to once -- as `if (poller)` blocks inside ONE
    //  loop every register of the compute waves that is assigned in one block and read in a later one stayed allocated around the whole loop, 40-90
    //  spills with six K/V passes in flight.  The eight workgroup barriers of a layer are executed by both loops in the same order.)
    const bool poller = __builtin_amdgcn_readfirstlane(tid0 >> 6) >= CW;
    Ctx c{a.err, false};
    const bool timer = STAMPS && a.stamps != nullptr && r0 == 0 && tid0 == 0 && a.scene == 0;
    unsigned long long t_prev = 0;
    auto stamp = [&](int p) {      // UMGEN_DEBUG_TIMING: 100 MHz ticks per phase of rank 0's compute wave 0
        if (STAMPS && timer) {
            const unsigned long long t = wall_clock64();
            if (p >= 0) a.stamps[p] += t - t_prev;
            t_prev = t;
        }
    };
    stamp(-1);
    const int Lk = a.st->step;
    const u32 ep = a.st->epoch;
    float* xs = lds + W_XS;
    float* xb = lds + W_XB;
    float* as = lds + W_AS;
    float* hs = lds + W_HS;
    float* qs = lds + W_QS;
    // granule buffers (shared by a call's scenes, one launch behind the other): x | q|k|v | key-quarter partials | attention output | x' | mlp partial sums [256][1536]
    u64* gx = a.gran;
    u64* gqkv = gx + XGR;
    u64* gpart = gqkv + 3 * WE;
    u64* gatt = gpart + WH * NSP * PREC;
    u64* gxb = gatt + WE;
    u64* gpy = gxb + XGR;
    const long kv_scene = (long)a.scene * a.kv_scene_stride;

    if (poller) {
        // =====================================================  POLL WAVES  =====================================================
        for (int l = 0; l < a.n_layers; ++l) {
            int tid = tid0, r = r0;
            asm volatile("" : "+v"(tid));
            asm volatile("" : "+s"(r));
            const int pt = tid - CT;                                              // poll thread 0 .. 127
            const int rl = r & (RPX - 1);
            const bool att_rank = rl < HPX * NSP, owner = att_rank && (rl % NSP) == 0;
            const int hh = HPX * (r / RPX) + rl / NSP;
            const OarLayerDev lw = a.layers[l];
            const u32 tg = ep + (u32)((a.scene * 64 + l) * 8);
            // hand-off 1: x (+ the layer's ln_1 | ln_2 weights -> LDS: 3072 floats over 128 lanes)
            {
                float lnv[24];
#pragma unroll
                for (int k = 0; k < 24; ++k) lnv[k] = ldg((pt + k * PT < WE ? lw.ln_a : lw.ln_b - WE) + pt + k * PT);
                if (l == 0) {
#pragma unroll
                    for (int k = 0; k < 12; ++k) xs[pt + k * PT] = ldg(a.xdec + (long)a.scene * WE + pt + k * PT);
                } else {
                    if (!UMGEN_WIDE_EXP_NOPOLL) poll_far(c, tid, gx, [&](int k) { return xslot((u32)(pt + k * PT)); }, tg + 0, [&](int k, float v) { xs[pt + k * PT] = v; });
                }
#pragma unroll
                for (int k = 0; k < 24; ++k) lds[W_LN + pt + k * PT] = lnv[k];
            }
            wg_barrier();      // B1
            // hand-off 2: q_h | k_h | v_h of the attention ranks (lanes 0 .. 127: values 0 .. 127, lanes 0 .. 15 also 128 .. 143)
            if (att_rank) {
                auto src = [&](int k) { const int e = min(pt + k * PT, 3 * kHeadDim - 1); return (u32)((e / kHeadDim) * WE + hh * kHeadDim + e % kHeadDim); };
                if (!UMGEN_WIDE_EXP_NOPOLL) poll_ms<2, kPollAll>(c, tid, gqkv, pt < 16 ? 3u : 1u, src, tg + 1, [&](int k, float v) { qs[pt + k * PT] = v; });
            }
            wg_barrier();      // B2
            wg_barrier();      // B3
            // hand-off 3: the owner's four quarters (4 records x 50 values = 200 granules over 128 lanes)
            if (owner) {
                constexpr int NREC = NSP * (kHeadDim + 2), PERQ = (NREC + PT - 1) / PT;      // NSP records x 50 values over 128 lanes
                auto src = [&](int k) { const int f = min(pt + k * PT, NREC - 1); return (u32)((f / (kHeadDim + 2)) * PREC + f % (kHeadDim + 2)); };
                u32 need = 0;
#pragma unroll
                for (int k = 0; k < PERQ; ++k) need |= (pt + k * PT < NREC) ? 1u << k : 0u;
                if (!UMGEN_WIDE_EXP_NOPOLL) poll_ms<PERQ, kPollAll>(c, tid, gpart + (long)(hh * NSP) * PREC, need, src, tg + 2,
                                     [&](int k, float v) { const int f = pt + k * PT; lds[W_SB + (f / (kHeadDim + 2)) * PREC + f % (kHeadDim + 2)] = v; });
            }
            wg_barrier();      // B4
            // hand-off 4: the 1536 attention outputs
            if (!UMGEN_WIDE_EXP_NOPOLL) poll_far(c, tid, gatt, [&](int k) { return (u32)(pt + k * PT); }, tg + 3, [&](int k, float v) { as[pt + k * PT] = v; });
            wg_barrier();      // B5
            wg_barrier();      // B5b
            // hand-off 5: x'
            if (!UMGEN_WIDE_EXP_NOPOLL) poll_far(c, tid, gxb, [&](int k) { return xslot((u32)(pt + k * PT)); }, tg + 4, [&](int k, float v) { xb[pt + k * PT] = v; });
            wg_barrier();      // B6
            wg_barrier();      // B7
            // hand-off 6: the 256 partials of this rank's 6 rows (slot f = producer p x 6 + row i: 1536 granules, 12 per poll lane)
            if (!UMGEN_WIDE_EXP_NOPOLL) poll_far(c, tid, gpy + WRO * r, [&](int k) { const u32 f = (u32)(pt + k * PT); return (f / (u32)WRO) * (u32)WE + f % (u32)WRO; }, tg + 5,
                                  [&](int k, float v) { lds[W_PT + pt + k * PT] = v; });
            wg_barrier();      // B8
        }
        return;
    }

    // =====================================================  COMPUTE WAVES  =====================================================
    // the weights of the phases ahead, in registers
    WRow wq[3];        // (the only weights alive across the layer boundary)
    {
        const int lane = tid0 & 63, wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int j = WRQ * (r0 & (RPX - 1)) + wave + CW * i;
            req_row(wq[i], a.layers[0].Wqkv, (long)((j / (HPX * kHeadDim)) * WE + HPX * kHeadDim * (r0 / RPX) + j % (HPX * kHeadDim)), lane);
        }
    }
    for (int l = 0; l < a.n_layers; ++l) {
        // (everything of a layer is derived from these INSIDE the layer: laundered, so that no loop-invariant row / granule address is hoisted into
        //  registers -- 64-bit addresses in VGPRs were what the register allocator spilled)
        int tid = tid0, r = r0;
        asm volatile("" : "+v"(tid));
        asm volatile("" : "+s"(r));
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int rl = r & (RPX - 1), xq = r / RPX;                             // rank on the XCD, XCD
        const bool att_rank = rl < HPX * NSP, owner = att_rank && (rl % NSP) == 0;
        const int hh = HPX * xq + rl / NSP, sp = rl % NSP;                    // (attention ranks: head, key quarter)
        // q|k|v rows of this rank: 18 of the 576 rows {q, k, v} x heads 4 xq .. 4 xq + 3 (3 per compute wave)
        auto qkv_row = [&](int i) { const int j = WRQ * rl + wave + CW * i; return (j / (HPX * kHeadDim)) * WE + HPX * kHeadDim * xq + j % (HPX * kHeadDim); };
        const OarLayerDev lw = a.layers[l];
        const u32 tg = ep + (u32)((a.scene * 64 + l) * 8);
        // attention geometry of this rank (head hh, key quarter sp): the compute waves split the quarter's keys in passes of 16 keys (4 lanes per key)
        const int nk = Lk + 1;
        const int spn = ((((nk + NSP - 1) / NSP) + KPW - 1) / KPW) * KPW;        // keys per quarter
        const int span = ((((spn + CW - 1) / CW) + KPW - 1) / KPW) * KPW;        // keys per wave
        const int nch = span / KPW;
        const int k_lo = sp * spn + wave * span, k_hi = min(min(nk, (sp + 1) * spn), k_lo + span);
        const int piece = lane & (LPK - 1), kg = lane / LPK;
        auto dim_of = [&](int j) { return j < 4 ? piece * 8 + 2 * j : 32 + piece * 4 + 2 * (j - 4); };
        auto kv_req = [&](KVPiece& kk, KVPiece& vv, int ci) {
            const bf16_t* kbase = a.kvcache + (long)l * a.kv_layer_stride + kv_scene + (long)hh * a.Lmax * kHeadDim;
            const bf16_t* vbase = kbase + (long)WH * a.Lmax * kHeadDim;
            const u32 off = (u32)min(k_lo + KPW * min(ci, nch - 1) + kg, a.Lmax - 1) * (u32)kHeadDim;
            kk.a = wld(kbase, off + (u32)piece * 8u);
            vv.a = wld(vbase, off + (u32)piece * 8u);
            if (UMGEN_WIDE_EXP_NOLOAD) { asm volatile("" : "=v"(kk.b)); asm volatile("" : "=v"(vv.b)); }
            else { kk.b = ldwu2(kbase, off + 32u + (u32)piece * 4u); vv.b = ldwu2(vbase, off + 32u + (u32)piece * 4u); }
        };
        // ---------------- (hand-off 1: x) ----------------
        // (K/V of the cached keys do not depend on this layer's q: the first KG passes, 80 keys per wave = 1920 positions, are requested ahead; the passes
        //  behind them pay a round trip each.  Where the requests go out: the UMGEN_WIDE_AT_* slots above.)
        float bq[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) bq[i] = ldg(lw.bqkv + qkv_row(i));
        const float bo = ldg(lw.bo + WRO * r + wave);
        KVPiece kc[KG], vc[KG];
        WRow wo, wf[4];
        u32x4_t wp[4][3];      // mlp c_proj slice of this rank, repacked [256 ranks][1536 rows][24]: rows tid, tid + 384, ...
        auto req_kv = [&]() {
            if (att_rank) {
#pragma unroll
                for (int i = 0; i < KG; ++i)
                    if (i < nch) kv_req(kc[i], vc[i], i);      // (only the passes this wave has: requests at clamped addresses for the others cost 0.9 us per layer)
            }
        };
        auto req_o = [&]() { req_row(wo, lw.Wo, (long)WRO * r + wave, lane); };
        auto req_f = [&]() {
#pragma unroll
            for (int i = 0; i < 4; ++i) req_row(wf[i], lw.Wfc, (long)WRF * r + wave + CW * i, lane);
        };
        auto req_p = [&]() {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) wp[i][j] = wld(lw.Wp2 + ((long)r * WE + tid + CT * i) * WRF, (u32)(8 * j));
        };
        auto req_q_next = [&]() {
            const OarLayerDev& ln = a.layers[min(l + 1, a.n_layers - 1)];      // (the last layer requests its own rows again: unconditional, no second loop form)
#pragma unroll
            for (int i = 0; i < 3; ++i) req_row(wq[i], ln.Wqkv, (long)qkv_row(i), lane);
        };
        auto at = [&](int slot) {      // (compile-time slots: each group's requests appear once)
            if (slot == UMGEN_WIDE_AT_KV) req_kv();
            if (slot == UMGEN_WIDE_AT_O) req_o();
            if (slot == UMGEN_WIDE_AT_F) req_f();
            if (slot == UMGEN_WIDE_AT_P) req_p();
            if (slot == UMGEN_WIDE_AT_Q) req_q_next();
        };
        at(0);
        wg_barrier();      // B1
        stamp(0);
        at(1);
        // ---------------- P1: LN + this rank's 18 q|k|v rows (hand-off 2: q_h | k_h | v_h of the attention ranks) ----------------
        {
            XRegs x;
            load_x(xs, lane, x);
            ln_regs(x, lds + W_LN, lane);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float v = row_dot<TT>(wq[i], x) + bq[i];
                const int n = qkv_row(i);
                if (lane == 0) {
                    put_local(gqkv, (u32)n, tg + 1, v);
                    if (n >= WE) {   // K / V rows of the new token: 16 bits into the cache (head-major [2][H][Lmax][48])
                        const int cc = n - WE, kvsel = cc / WE, hc = cc % WE;
                        (a.kvcache + (long)l * a.kv_layer_stride + kv_scene)[(u32)(((kvsel * WH + hc / kHeadDim) * a.Lmax + Lk) * kHeadDim + hc % kHeadDim)] = bits16<TT>(v);
                    }
                }
            }
        }
        at(2);
        stamp(1);
        wg_barrier();      // B2
        stamp(2);
        // ---------------- P2: attention of (head hh, key quarter sp): the compute waves split the quarter's keys ----------------
        if (att_rank) {
            WAtt st;
            st.m = -INFINITY; st.l = 0.f;
            f32x2_t q2[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) { st.o[j] = f32x2_t{0.f, 0.f}; q2[j] = f32x2_t{qs[dim_of(j)], qs[dim_of(j) + 1]}; }
            auto chunk = [&](const KVPiece& kcb, const KVPiece& vcb, int ci) {
                const int k = k_lo + KPW * ci + kg;
                const u32 kw[6] = {kcb.a.x, kcb.a.y, kcb.a.z, kcb.a.w, kcb.b.x, kcb.b.y};
                f32x2_t acc = {0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 6; ++j) acc = mac2<TT>(kw[j], q2[j], acc);
                if (k == Lk) {   // the new token's own key is not in the cache yet: from the q | k | v exchange, as the cache will hold it
                    acc = f32x2_t{0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 6; ++j)
                        acc = __builtin_elementwise_fma(f32x2_t{round16<TT>(qs[kHeadDim + dim_of(j)]), round16<TT>(qs[kHeadDim + dim_of(j) + 1])}, q2[j], acc);
                }
                float d = acc.x + acc.y;
                d += dpp_xor1(d);
                d += dpp_xor2(d);
                d = (k < k_hi) ? d * kScaleQK : -INFINITY;
                const float m_new = fmaxf(st.m, d);
                if (m_new > -INFINITY) {
                    const float scale = __expf(st.m - m_new);
                    const f32x2_t scale2 = {scale, scale};
                    st.l *= scale;
#pragma unroll
                    for (int j = 0; j < 6; ++j) st.o[j] *= scale2;
                    const u32 vw[6] = {vcb.a.x, vcb.a.y, vcb.a.z, vcb.a.w, vcb.b.x, vcb.b.y};
                    const float p = (k < k_hi) ? __expf(d - m_new) : 0.f;
                    const f32x2_t p2 = {p, p};
                    st.l += p;
                    if (k == Lk) {
#pragma unroll
                        for (int j = 0; j < 6; ++j)
                            st.o[j] = __builtin_elementwise_fma(p2, f32x2_t{round16<TT>(qs[2 * kHeadDim + dim_of(j)]), round16<TT>(qs[2 * kHeadDim + dim_of(j) + 1])}, st.o[j]);
                    } else if (k < k_hi) {
#pragma unroll
                        for (int j = 0; j < 6; ++j) st.o[j] = mac2<TT>(vw[j], p2, st.o[j]);
                    }
                    st.m = m_new;
                }
            };
#pragma unroll
            for (int i = 0; i < KG; ++i)
                if (i < nch) chunk(kc[i], vc[i], i);
            for (int ci = KG; ci < nch; ++ci) {      // (contexts beyond KG x 16 keys per wave: a round trip per pass)
                KVPiece kt, vt;
                kv_req(kt, vt, ci);
                chunk(kt, vt, ci);
            }
            watt_fold_dpp<0x124>(st);
            watt_fold_dpp<0x128>(st);
            watt_fold_swap<false>(st);
            watt_fold_swap<true>(st);
            float* wpz = lds + W_WP + wave * PREC;
            if (lane < LPK) {
#pragma unroll
                for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x2_t*>(wpz + dim_of(j)) = st.o[j];
                if (lane == 0) { wpz[48] = st.m; wpz[49] = st.l; }
            }
        }
        at(3);
        stamp(3);
        wg_barrier();      // B3
        stamp(4);
        // ---------------- the compute waves' partials merged in wave order -> this quarter's partial (unnormalised o, m, l) (hand-off 3: the owner's four quarters) ----------------
        if (att_rank && tid < kHeadDim + 2) {
            const float* wz = lds + W_WP;
            float M = wz[48];
#pragma unroll
            for (int ww = 1; ww < CW; ++ww) M = fmaxf(M, wz[ww * PREC + 48]);
            float Ls = 0.f, o = 0.f;
#pragma unroll
            for (int ww = 0; ww < CW; ++ww) {
                const float e = (M > -INFINITY) ? __expf(wz[ww * PREC + 48] - M) : 0.f;
                Ls = fmaf(e, wz[ww * PREC + 49], Ls);
                if (tid < kHeadDim) o = fmaf(e, wz[ww * PREC + tid], o);
            }
            put_local(gpart + (long)(hh * NSP + sp) * PREC, (u32)tid, tg + 2, tid < kHeadDim ? o : (tid == kHeadDim ? M : Ls));
        }
        at(4);
        wg_barrier();      // B4
        stamp(5);
        // ---------------- the owner merges its head's four quarters -> attention output (hand-off 4: the 1536 attention outputs) ----------------
        if (owner && tid < kHeadDim) {
            const float* p0 = lds + W_SB;
            float M = p0[48];
#pragma unroll
            for (int s2 = 1; s2 < NSP; ++s2) M = fmaxf(M, p0[s2 * PREC + 48]);
            float Ls = 0.f, o = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < NSP; ++s2) {
                const float e = (p0[s2 * PREC + 48] > -INFINITY) ? __expf(p0[s2 * PREC + 48] - M) : 0.f;
                Ls = fmaf(e, p0[s2 * PREC + 49], Ls);
                o = fmaf(e, p0[s2 * PREC + tid], o);
            }
            put_far(gatt, (u32)(hh * kHeadDim + tid), tg + 3, o / Ls);
        }
        at(5);
        wg_barrier();      // B5
        stamp(6);
        at(6);
        // ---------------- P3: c_proj row 6 r + wave + residual -> x' (hand-off 5: x') ----------------
        {
            XRegs x;
            load_x(as, lane, x);
            const int n = WRO * r + wave;
            const float v = row_dot<TT>(wo, x) + bo;
            if (lane == 0) lds[W_XO + wave] = xs[n] + v;
        }
        wg_barrier();      // B5b
        if (tid < XPAD) put_far(gxb, (u32)(XPAD * r + tid), tg + 4, tid < WRO ? lds[W_XO + tid] : 0.f);      // (ONE store of whole 64-byte pieces)
        stamp(7);
        wg_barrier();      // B6
        stamp(8);
        at(7);
        // ---------------- P4: LN + this rank's 24 hidden units + GELU ----------------
        {
            XRegs x;
            load_x(xb, lane, x);
            ln_regs(x, lds + W_LN + WE, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = row_dot<TT>(wf[i], x);
                if (lane == 0) hs[wave + CW * i] = gelu_erf(v);
            }
        }
        stamp(9);
        wg_barrier();      // B7
        stamp(10);
        at(8);
        // ---------------- this rank's partial sums of the 1536 mlp c_proj outputs (hand-off 6: the 256 partials of this rank's 6 rows) ----------------
        {
            f32x2_t hq[3][4];
#pragma unroll
            for (int j = 0; j < 3; ++j) load8p(hs + 8 * j, hq[j]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x2_t acc = {0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 3; ++j) acc = dot8<TT>(wp[i][j], hq[j], acc);
                put_far(gpy + (long)r * WE, (u32)(tid + CT * i), tg + 5, acc.x + acc.y);
            }
        }
        stamp(11);
        wg_barrier();      // B8
        stamp(12);
        at(9);
        // ---------------- P5: x'' = x' + the 256 partial sums (eight lanes add 32 producers each, ascending; then the eight in a fixed order) ----------------
        if (tid < 64) {      // wave 0: eight lanes per row (rows 6, 7: copies of row 5, never polled), 32 producers per lane
            const int i = min(tid >> 3, WRO - 1), g8 = tid & 7;
            float s = 0.f;
#pragma unroll 8
            for (int p = 0; p < NWG / 8; ++p) s += lds[W_PT + (g8 * (NWG / 8) + p) * WRO + i];
            s += dpp_xor1(s);
            s += dpp_xor2(s);
            s += dpp_mov<0x141>(s);      // row_half_mirror: the other quad of the eight
            const int n = WRO * r + i;
            const float xn = xb[n] + s;
            if (g8 == 0) {      // lanes 0, 8, .. 56: ONE store of 8 consecutive granules (64 bytes)
                if (l + 1 == a.n_layers) { if ((tid >> 3) < WRO) a.xdec[(long)a.scene * WE + n] = xn; }
                else put_far(gx, (u32)(XPAD * r + (tid >> 3)), tg + 8, (tid >> 3) < WRO ? xn : 0.f);
            }
        }
        stamp(13);
        if (STAMPS && timer) a.stamps[15] += 1;
        // (no barrier: the next layer's gathers cannot complete before every rank -- this one included -- has published its x'' rows, and
        //  W_PT / xb are next written behind B6 / B8 of the next layer)
    }
}

size_t oar_engine_wide_lds_bytes() {
    const size_t need = (size_t)W_TOTAL * sizeof(float);
    return need > (size_t)(96 << 10) ? need : (size_t)(96 << 10);      // > 80 KB: never two workgroups on one CU
}
size_t oar_engine_wide_granules() { return (size_t)XGR + 3 * WE + (size_t)WH * NSP * PREC + WE + XGR + (size_t)NWG * WE; }

__global__ __launch_bounds__(kEngThreads) void oar_engine_wide_census_kernel(unsigned int* count) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (threadIdx.x == 0) {
        lds[0] = 0.f;
        atomicAdd(count + xcc_id(), 1u);
    }
}
hipError_t launch_oar_engine_wide_census(hipStream_t s, unsigned int* d_counts16) {
    hipLaunchKernelGGL(oar_engine_wide_census_kernel, dim3(NWG), dim3(kEngThreads), oar_engine_wide_lds_bytes(), s, d_counts16);
    return hipGetLastError();
}

hipError_t oar_engine_wide_prepare() {
    for (const void* f : {reinterpret_cast<const void*>(oar_engine_wide_kernel<false, bf16_t>), reinterpret_cast<const void*>(oar_engine_wide_kernel<false, f16_t>),
                          reinterpret_cast<const void*>(oar_engine_wide_kernel<true, bf16_t>), reinterpret_cast<const void*>(oar_engine_wide_kernel<true, f16_t>),
                          reinterpret_cast<const void*>(oar_engine_wide_census_kernel)}) {
        hipError_t rc = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)oar_engine_wide_lds_bytes());
        if (rc != hipSuccess) return rc;
    }
    return hipSuccess;
}

hipError_t launch_oar_engine_wide(hipStream_t s, const OarWideArgs& a) {
    const dim3 grid(NWG), block(kEngThreads);
    const size_t shm = oar_engine_wide_lds_bytes();
    if (a.stamps) {
        if (a.fp16) hipLaunchKernelGGL((oar_engine_wide_kernel<true, f16_t>), grid, block, shm, s, a);
        else hipLaunchKernelGGL((oar_engine_wide_kernel<true, bf16_t>), grid, block, shm, s, a);
    } else {
        if (a.fp16) hipLaunchKernelGGL((oar_engine_wide_kernel<false, f16_t>), grid, block, shm, s, a);
        else hipLaunchKernelGGL((oar_engine_wide_kernel<false, bf16_t>), grid, block, shm, s, a);
    }
    return hipGetLastError();
}

}  // namespace umgen
