// Multi-scene XCD-resident decode engine: the 36 BlockOAR layers (module.py:378-428) of one decode step (UMGen.py:1209-1262) for MANY
// scenes per GPU in ONE launch.  Same machine as oar_engine.hip -- 8 groups of 32 workgroups (one per CU of an XCD), the all-to-all
// hand-offs of a layer as 8-byte {tag, value} granules inside the group's coherent L2, layers resident on their groups (layer l on group
// l % 8) with the work flowing systolically through them -- but a work item is (BLOCK OF SCENES, layer) instead of (scene, layer):
//   * the ns <= 8 scenes of a block are the 16 B-columns of v_mfma_f32_16x16x32 as (hi, lo) 16-bit pairs of their fp32 activations
//     (column 2 s = hi parts of scene s, column 2 s + 1 = lo parts; x = hi + lo to 2^-17 relative in bf16, 2^-22 in fp16), so q|k|v,
//     c_proj, c_fc and the mlp partial sums cost ONE pass over the layer's weights per item whatever ns is -- the one-scene engine
//     spends 12.8 us per (scene, layer) whatever the batch is, which is why its HBM fraction FALLS with the batch;
//   * the five hand-offs of an item (x, q|k|v, attention output, x', mlp partial sums) are shared by the ns scenes;
//   * only the attention scales with ns: the block's ns x 16 (scene, head) pairs are dealt over the group's 32 CUs (pair p on CU
//     p % 32), a CU walks a pair's keys once in 8 wave spans (4 lanes per key, 16 keys per wave pass, UMGEN_MS_NB passes in flight) and
//     the K/V stream of its next pair is already requested while the current pair's 8 wave partials are merged.
// ns = ceil(B / 8), so a step is at most 8 blocks: (35 + blocks) item slots per step.
//
// Arithmetic per scene (independent of B, ns, the block a scene sits in and its column: batch-invariant bit for bit WITHIN this engine):
// fp32 activations, 16-bit weights and K/V cache, fp32 accumulation.  Weight-only LayerNorm (eps 1e-5) by one wave per scene with the
// statistics of oar_engine.hip's ln_split; every row product = the matrix-core instruction over this wave's 96 k (3 k-steps), hi and lo
// columns added, then the 8 waves' k ranges added in wave order (+ bias); exact erf-GELU; the mlp split by hidden units with the 32 CUs'
// partial sums added in four groups of eight.  Attention of a (scene, head): online softmax per lane group, the 16 lane groups of a
// wave folded in a fixed butterfly, the 8 waves merged in wave order; the new token's own key / value come out of the q|k|v exchange
// rounded to 16 bits (what the cache will hold).  Against the one-scene engine (VALU q|k|v / c_proj rows, two CUs per head) logits
// differ by ~1e-4 teacher-forced (tests/test_gpu_decode_engine.py), the bar of DESIGN.md section 3 is 1e-3.
#include "oar_common.h"

namespace umgen {

namespace {

#ifndef UMGEN_MS_NB
#define UMGEN_MS_NB 2            // 16-key passes of a wave in flight (12 VGPRs each for K, 12 for V).  The key loop is bound by the XCD's memory link (0.67 TB/s
                                 // with 2, 3, 4 or 6 passes in flight, profiles/r05_ms_experiments.txt); 4 spill 32 VGPRs, 2 none
#endif
#ifndef UMGEN_MS_POLL_ALL_X
#define UMGEN_MS_POLL_ALL_X false      // re-request every missing x granule every round while the group waits for its predecessor (cross-XCD)
#endif
#ifndef UMGEN_MS_POLL_ALL_ATT
#define UMGEN_MS_POLL_ALL_ATT false    // ... every missing attention output while other CUs of the group still stream K/V through the same L2
#endif
constexpr int MS = kEngMsScenes;  // scenes per work item (16 matrix-core columns = 8 x (hi, lo))
constexpr int XST = E + 4;        // scene stride (dwords) of the activation buffer: 772 = 4 mod 32 banks, the 32 (scene, k-group) chunks of a B-fragment read spread over all banks
constexpr int RS = 100;           // row stride of a (wave, scene) strip of partial sums
constexpr int HST = 100;          // scene stride of the packed gelu(c_fc) values
constexpr int QST = 160, WPS = 52;
// LDS carve (dwords)
constexpr int M_XS = 0;                        // activations of the item's scenes [8][772]: fp32 as gathered, then packed (hi | lo << 16) in place
constexpr int M_ST = M_XS + MS * XST;          // partial sums of the 8 waves' k ranges [8 waves][8 scenes][100]
constexpr int M_W2 = M_ST + NW * MS * RS;      // parked mlp c_proj fragments 0..11 of every thread [12][512] x 16 B
constexpr int M_LN = M_W2 + 12 * NT * 4;       // ln_1 | ln_2 weights of the layer [1536]
constexpr int M_XR = M_LN + 2 * E;             // x of this CU's 24 c_proj rows [8][24] (the attention projection's residual)
constexpr int M_HH = M_XR + MS * 24;           // packed gelu(c_fc) of this CU's 96 hidden units [8][100]
constexpr int M_MISC = M_HH + MS * HST;
constexpr int M_TOTAL = M_MISC + 16;
// inside M_XS while the attention runs (the LayerNorm-ed x is dead behind P1's last barrier): q_h | k_h | v_h of this CU's pairs [4][160],
// the 8 waves' partial (o[48], m, l) of a pair, double-buffered [2][8][52]
constexpr int M_QS = M_XS;
constexpr int M_WP = M_QS + 4 * QST;
static_assert(M_WP + 2 * NW * WPS <= M_ST, "attention scratch inside the activation buffer");
static_assert(CU * MS * 24 <= MS * XST, "gathered mlp partial sums inside the activation buffer");
static_assert(M_TOTAL * 4 <= 160 * 1024, "LDS budget");

// slot k of thread tid (bit k of need) waits for granule idx(k) with `tag` and hands its value to sink(k, value).  Round 1 requests every
// slot; while something is missing a lane asks for ONE of its missing granules per round with a short sleep in between (the gathers
// of this engine are up to 12 granules per thread and 32 CUs: polling all of them would put 1.5 MB per round on the L2 / the fabric
// while a group waits for its predecessor), then requests all its missing slots again.
// value -> hi | lo << 16 in the operand type
template <typename TT>
__device__ inline u32 pack16(float v) {
    unsigned short hi, lo;
    split16<TT>(v, hi, lo);
    return (u32)hi | ((u32)lo << 16);
}

// LayerNorm (weight only, eps 1e-5, module.py:26-37) of one scene's 768 fp32 values by ONE wave, result packed (hi | lo << 16) in place.
// Statistics exactly as oar_engine.hip's ln_split.  XR: also leave x of this CU's 24 c_proj rows (24 w ..) in xr[0..23] (fp32, the residual).
template <typename TT, bool XR>
__device__ inline void ln_pack(float* xs, const float* lnw, int lane, int w, float* xr) {
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
    float xv[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) xv[i] = xs[lane + 64 * i];
    u32* xu = reinterpret_cast<u32*>(xs);
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const int k = lane + 64 * i;
        if (XR) { const int r = k - 24 * w; if (r >= 0 && r < 24) xr[r] = xv[i]; }
        xu[k] = pack16<TT>((xv[i] - mean) * rstd * lnw[k]);
    }
}

// B operand of the wave's 3 k-steps out of a packed activation buffer: lane l = column l % 16 (scene (l % 16) / 2, hi or lo part), the 8
// values k0 + 32 j + 8 (l / 16) .. + 7.  Columns past the block's scenes repeat its last scene (their results are never stored).
template <typename TT>
__device__ inline void load_bfrags_ms(const u32* xu, int stride, int ns, int k0, int lane, typename Mma16<TT>::vec (&b)[3]) {
    typedef typename Mma16<TT>::vec vec;
    const int col = lane & 15;
    const u32* p = xu + min(col >> 1, ns - 1) * stride + k0 + 8 * (lane >> 4);
    const u32 sel = (col & 1) ? 0x07060302u : 0x05040100u;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const u32x4_t a0 = *reinterpret_cast<const u32x4_t*>(p + 32 * j);
        const u32x4_t a1 = *reinterpret_cast<const u32x4_t*>(p + 32 * j + 4);
        const u32x4_t o = {__builtin_amdgcn_perm(a0.y, a0.x, sel), __builtin_amdgcn_perm(a0.w, a0.z, sel),
                           __builtin_amdgcn_perm(a1.y, a1.x, sel), __builtin_amdgcn_perm(a1.w, a1.z, sel)};
        b[j] = __builtin_bit_cast(vec, o);
    }
}
// hi + lo columns added (into the even column), rows 4 (l / 16) .. + 3 of tile t of scene (l % 16) / 2 -> the wave's strip
__device__ inline void store_tile(float* strip_wave, int ns, int lane, int t, f32x4_t acc) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] += dpp_mov<0x101>(acc[r]);       // row_shl:1: column 2 s + 1 (lo) onto column 2 s (hi)
    const int col = lane & 15;
    if (!(col & 1) && (col >> 1) < ns) *reinterpret_cast<f32x4_t*>(strip_wave + (col >> 1) * RS + 16 * t + 4 * (lane >> 4)) = acc;
}

// online-softmax state of a lane (its 12 of the head's 48 dimensions as 6 packed pairs)
struct AttState { float m, l; f32x2_t o[6]; };
__device__ inline void att_merge(AttState& a, float mb, float lb, const f32x2_t (&ob)[6]) {
    const float M = fmaxf(a.m, mb);
    const float ea = (M > -INFINITY) ? __expf(a.m - M) : 0.f, eb = (M > -INFINITY) ? __expf(mb - M) : 0.f;
    a.l = fmaf(eb, lb, ea * a.l);
#pragma unroll
    for (int j = 0; j < 6; ++j) a.o[j] = f32x2_t{fmaf(eb, ob[j].x, ea * a.o[j].x), fmaf(eb, ob[j].y, ea * a.o[j].y)};
    a.m = M;
}
template <int CTRL>
__device__ inline void att_fold_dpp(AttState& a) {       // with the lane CTRL brings in (same dimensions, another key group of the 16-lane row)
    f32x2_t ob[6];
    const float mb = dpp_mov<CTRL>(a.m), lb = dpp_mov<CTRL>(a.l);
#pragma unroll
    for (int j = 0; j < 6; ++j) ob[j] = f32x2_t{dpp_mov<CTRL>(a.o[j].x), dpp_mov<CTRL>(a.o[j].y)};
    att_merge(a, mb, lb, ob);
}
template <bool ROW32>
__device__ inline void att_fold_swap(AttState& a) {      // with the neighbouring 16-lane row (ROW32: the other half-wave)
    auto sw = [](float v, float& x, float& y) {
        if (ROW32) { auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false); x = __uint_as_float(r[0]); y = __uint_as_float(r[1]); }
        else { auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false); x = __uint_as_float(r[0]); y = __uint_as_float(r[1]); }
    };
    AttState x, y;
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
    att_merge(x, y.m, y.l, y.o);
    a = x;
}

}  // namespace

template <bool STAMPS, typename TT>
__global__ __launch_bounds__(kEngThreads) void oar_engine_ms_kernel(OarMsArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    u32* ldsu = reinterpret_cast<u32*>(lds);
    const int tid0 = threadIdx.x;
    const unsigned long long t_k0 = wall_clock64();   // 100 MHz
    const u32 xcc = xcc_id();
    const int g = a.xcc_group[xcc];
    if (g >= a.NG) return;   // (the census guarantees this never happens)
    if (tid0 == 0) ldsu[M_MISC] = atomicAdd(a.ticket + g, 1u) & (u32)(CU - 1);
    wg_barrier();
    const int w0 = __builtin_amdgcn_readfirstlane((int)ldsu[M_MISC]);
    Ctx c{a.err, false};
    const bool timer = STAMPS && a.stamps != nullptr && g == 0 && w0 == 0 && tid0 == 0;
    unsigned long long t_prev = 0;
    auto stamp = [&](int p) {
        if (STAMPS && timer) {
            const unsigned long long t = wall_clock64();
            if (p >= 0) a.stamps[p] += t - t_prev;
            t_prev = t;
        }
    };
    const int Lk = a.st->step;              // cached keys before this step == position of the new token
    const u32 ep = a.st->epoch;
    const int D = a.NG, q = g, nb = a.nb;
    // items of this group: layers outside (q, q + D, ...), the blocks of the batch inside; the layers that do not fill a whole round of
    // the D groups are shared like in oar_engine.hip's systolic schedule (36 = 4 x 8 + 4: groups g and g + 4 both keep layer 32 + g % 4,
    // one for the first half of the blocks, one for the second)
    const int n_full = a.n_layers / D;
    int tail_l = -1, ts0 = 0, ts1 = 0;
    {
        const int rem = a.n_layers - n_full * D;
        if (rem > 0) {
            if (D % rem == 0) {
                const int share = D / rem, part = q / rem;
                tail_l = n_full * D + q % rem;
                ts0 = part * nb / share;
                ts1 = (part + 1) * nb / share;
            } else if (q < rem) {
                tail_l = n_full * D + q;
                ts1 = nb;
            }
        }
    }
    const int n_items = n_full * nb + (ts1 - ts0);
    WFrags<4> ff;            // c_fc: tiles 0..3 of this CU's 96 rows (6 tiles)                         (resident over the blocks of a layer; tiles 4, 5 per item)
    WFrags<5> fq;            // q|k|v: this CU's 72 rows as 5 tiles                                     (requested by every item: out of the L2 after the layer's first block)
    float xres2 = 0.f;       // x' of (scene tid / 24, row 24 w + tid % 24): the mlp projection's residual
    // (block, layer) of this group's item number `item`
    auto sched = [&](int item, int& blk, int& l, bool& load_w) {
        const bool tail = item >= n_full * nb;
        blk = tail ? ts0 + item - n_full * nb : item % nb;
        l = tail ? tail_l : q + D * (item / nb);
        load_w = blk == (tail ? ts0 : 0);          // this item requests the layer's resident weights
    };
    for (int item = 0; item < n_items; ++item) {
        int blk, l;
        bool load_w;
        sched(item, blk, l, load_w);
        const int s0 = blk * a.ns;
        const int ns = min(a.ns, a.B - s0);
        if (ns <= 0) continue;
        if (item == 0 && q > 0) {       // launch-time stagger of the groups' first weight streams (oar_engine.hip)
            const unsigned long long until = t_k0 + (unsigned long long)(q * 1000);
            while (wall_clock64() < until) __builtin_amdgcn_s_sleep(8);
        }
        // Everything below is derived from these values INSIDE the item (laundered: nothing loop-invariant is hoisted into registers)
        int tid = tid0, w = w0, gl = g;
        asm volatile("" : "+v"(tid));
        asm volatile("" : "+s"(w), "+s"(gl));
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        u64* gqkv = a.gloc + (long)gl * kEngMsLocStride;     // q | k | v of the block's scenes [8][2304]
        u64* gatt = gqkv + MS * 3 * E;                       // attention outputs [8][768]
        u64* gxb = gatt + MS * E;                            // x' [8][768]
        u64* gpy = gxb + MS * E;                             // mlp partial sums [32 producers][8][768]
        const OarLayerDev lw = a.layers[l];
        const u32 tg = ep + (u32)((blk * 64 + l) * 8);
        u32x4_t* w2p = reinterpret_cast<u32x4_t*>(ldsu + M_W2) + tid;
        const bf16_t* wp2 = lw.Wp2 + (long)w * kEngWpUnits * NT * 8;
        float lnr[3];
        if (load_w) {
#pragma unroll
            for (int k = 0; k < 3; ++k) lnr[k] = ldg((tid + k * NT < E ? lw.ln_a : lw.ln_b - E) + tid + k * NT);
        }
        // biases of the outputs this thread finishes: q|k|v row (tid + 512 it) % 72, c_proj row tid % 24
        float bq[2], bo = 0.f;
#pragma unroll
        for (int it = 0; it < 2; ++it) { const int o = tid + it * NT; bq[it] = o < 72 * ns ? ldg(lw.bqkv + 72 * w + o % 72) : 0.f; }
        if (tid < 24 * ns) bo = ldg(lw.bo + 24 * w + tid % 24);
        if (load_w) {
            // layer switch (once per layer and step): the parked mlp fragments through 6 staging registers at a time, in front of everything else
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                u32x4_t wp[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) wp[j] = ldwu(wp2 + (long)(6 * hb + j) * NT * 8, (u32)tid * 8u);
#pragma unroll
                for (int j = 0; j < 6; ++j) w2p[(6 * hb + j) * NT] = wp[j];
            }
        }
        req_frags<5, true>(fq, lw.Wqkv, 72 * w, 72, wave, lane);
        const bf16_t* pf = lw.Wf2 + (long)(w * NW + wave) * 18 * 64 * 8;
        if (load_w) req_frags_packed<4, false>(ff, pf, lane);
        stamp(-1);
        // ================= P1: x -> LN -> q | k | v of the block's scenes =================
        if (load_w) {
#pragma unroll
            for (int k = 0; k < 3; ++k) lds[M_LN + tid + k * NT] = lnr[k];
        }
        if (l == 0) {      // (the sampler's / the first-input kernel's rows: plain loads, once per block and step)
#pragma unroll
            for (int k = 0; k < 12; ++k) { const int f = tid + k * NT; if (f < ns * E) lds[M_XS + (f / E) * XST + f % E] = ldg(a.xdec + (long)s0 * E + f); }
        } else {
            u32 need = 0;
#pragma unroll
            for (int k = 0; k < 12; ++k) if (tid + k * NT < ns * E) need |= 1u << k;
            poll_ms<12, UMGEN_MS_POLL_ALL_X>(c, tid, a.gx + (long)s0 * E, need, [&](int k) { return (u32)min(tid + k * NT, ns * E - 1); }, tg + 0,
                        [&](int k, float v) { const int f = tid + k * NT; lds[M_XS + (f / E) * XST + f % E] = v; });
        }
        wg_barrier();
        stamp(0);   // waited for x
        if (wave < ns) ln_pack<TT, true>(lds + M_XS + wave * XST, lds + M_LN, lane, w, lds + M_XR + wave * 24);
        wg_barrier();
        {
            typename Mma16<TT>::vec bx[3];
            typedef typename Mma16<TT>::vec vec;
            load_bfrags_ms<TT>(ldsu + M_XS, XST, ns, 96 * wave, lane, bx);
            f32x4_t acc[5];
#pragma unroll
            for (int t = 0; t < 5; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int t = 0; t < 5; ++t) acc[t] = Mma16<TT>::mfma(__builtin_bit_cast(vec, fq.f[t][j]), bx[j], acc[t]);
#pragma unroll
            for (int t = 0; t < 5; ++t) store_tile(lds + M_ST + wave * MS * RS, ns, lane, t, acc[t]);
        }
        wg_barrier();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int o = tid + it * NT;
            if (o < 72 * ns) {
                const int si = o / 72, r = o - 72 * si;
                float v = bq[it];
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) v += lds[M_ST + (ww * MS + si) * RS + r];      // fixed order: k ranges 0, 1, ..., 7
                const int n = 72 * w + r;
                put_local(gqkv + si * 3 * E, (u32)n, tg + 1, v);
                if (n >= E) {   // K / V rows of the new token: 16 bits into the cache (head-major [2][H][Lmax][48])
                    const int cc = n - E, kvsel = cc / E, hc = cc % E;
                    (a.kvcache + (long)l * a.kv_layer_stride + (long)(s0 + si) * a.kv_scene_stride)[
                        (u32)(((kvsel * H + hc / kHeadDim) * a.Lmax + Lk) * kHeadDim + hc % kHeadDim)] = bits16<TT>(v);
                }
            }
        }
        stamp(1);   // LN + q|k|v rows
        // ================= P2: attention of this CU's (scene, head) pairs: pair p = w, w + 32, ... < 16 ns =================
        {
            const int npair = (16 * ns - w + 31) >> 5;          // (0 for w >= 16 ns)
            float* qs = lds + M_QS;
            {
                u32 need = 0;
#pragma unroll
                for (int k = 0; k < 2; ++k) if (tid + k * NT < 144 * npair) need |= 1u << k;
                auto src = [&](int k) {
                    const int f = min(tid + k * NT, max(144 * npair - 1, 0));
                    const int pi = f / 144, e = f - 144 * pi, p = w + 32 * pi;
                    return (u32)((p >> 4) * 3 * E + (e / kHeadDim) * E + (p & 15) * kHeadDim + e % kHeadDim);
                };
                poll_ms<2, true>(c, tid, gqkv, need, src, tg + 1, [&](int k, float v) { const int f = tid + k * NT; qs[(f / 144) * QST + f % 144] = v; });
                wg_barrier();
            }
            stamp(2);   // waited for the pairs' q_h | k_h | v_h
            const int nk = Lk + 1;
            const int span = ((((nk + NW - 1) / NW) + KPW - 1) / KPW) * KPW;     // keys per wave (the same for every pair: they share the step)
            const int nch = span / KPW, total = npair * nch;
            const int k_lo = wave * span, k_hi = min(nk, k_lo + span);
            const int piece = lane & (LPK - 1), kg = lane / LPK;
            auto dim_of = [&](int j) { return j < 4 ? piece * 8 + 2 * j : 32 + piece * 4 + 2 * (j - 4); };
            const bf16_t* kv_layer = a.kvcache + (long)l * a.kv_layer_stride + (long)s0 * a.kv_scene_stride;
            constexpr int NB = UMGEN_MS_NB;
            KVPiece kc[NB], vc[NB];
            // Chunk c of the flat (pair, 16-key pass) sequence.  The K/V requests are INLINE ASSEMBLY with hand-counted waits: written as
            // compiler-visible loads under `if (chunk exists)` the wait-count pass has to assume the shortest path and emitted
            // s_waitcnt vmcnt(3..0) in front of every chunk -- every pass waited for ALL the requests in flight, the key loop ran at one
            // memory latency per 16 keys whatever UMGEN_MS_NB was (profiles/r05_ms_nb_sweep.txt: 10.6 us per pair for NB = 2, 3, 4, 6).
            // Here every slot of the ring ALWAYS issues its four requests (slots past the end of the sequence re-request the last
            // chunk: cache hits), so exactly 4 (NB - 1) requests are younger than the oldest buffer's and `s_waitcnt vmcnt(4 (NB - 1))`
            // is exact (a wave's loads return in order; the one granule store wave 0 has in flight can only make the wait longer).
            int rq_pi = 0, rq_ci = 0;
            auto kv_req = [&](KVPiece& kq, KVPiece& vq) {
                const int p = w + 32 * rq_pi;
                const bf16_t* kbase = kv_layer + (long)(p >> 4) * a.kv_scene_stride + (long)(p & 15) * a.Lmax * kHeadDim;
                const bf16_t* vbase = kbase + (long)H * a.Lmax * kHeadDim;
                const u32 off = ((u32)min(k_lo + KPW * rq_ci + kg, a.Lmax - 1) * (u32)kHeadDim + (u32)piece * 8u) * 2u;      // bytes
                const u32 off2 = ((u32)min(k_lo + KPW * rq_ci + kg, a.Lmax - 1) * (u32)kHeadDim + 32u + (u32)piece * 4u) * 2u;
#ifdef UMGEN_MS_EXP_NOLOAD      // measurement builds only (tools/build_variant.sh): the key loop without its K/V requests / without its arithmetic
                kq.a = u32x4_t{off, off2, off, off2}; kq.b = u32x2_t{off, off2}; vq.a = kq.a; vq.b = kq.b;
                if (rq_pi * nch + rq_ci + 1 < total) { if (++rq_ci == nch) { rq_ci = 0; ++rq_pi; } }
                return;
#endif
                asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(kq.a) : "v"(off), "s"(kbase));
                asm volatile("global_load_dwordx2 %0, %1, %2 nt" : "=v"(kq.b) : "v"(off2), "s"(kbase));
                asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(vq.a) : "v"(off), "s"(vbase));
                asm volatile("global_load_dwordx2 %0, %1, %2 nt" : "=v"(vq.b) : "v"(off2), "s"(vbase));
                if (rq_pi * nch + rq_ci + 1 < total) { if (++rq_ci == nch) { rq_ci = 0; ++rq_pi; } }     // (stays on the last chunk behind the end)
            };
            auto kv_wait = [&](KVPiece& kq, KVPiece& vq) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NB - 1)));
                asm volatile("" : "+v"(kq.a), "+v"(kq.b), "+v"(vq.a), "+v"(vq.b));     // (defined behind the wait, not behind the request)
            };
            if (total > 0) {
#pragma unroll
                for (int bfr = 0; bfr < NB; ++bfr) kv_req(kc[bfr], vc[bfr]);
            }
            int pi = 0, ci = 0;
            AttState st;
            f32x2_t q2[6];
            auto chunk = [&](const KVPiece& kcb, const KVPiece& vcb) {
                const float* qp = qs + pi * QST;
#ifdef UMGEN_MS_EXP_NOMATH
                if (ci == 0) { st.m = 0.f; st.l = 1.f;
#pragma unroll
                    for (int j = 0; j < 6; ++j) { st.o[j] = f32x2_t{0.f, 0.f}; q2[j] = st.o[j]; } }
                st.o[0].x += __uint_as_float((kcb.a.x ^ kcb.a.y ^ kcb.a.z ^ kcb.a.w ^ kcb.b.x ^ kcb.b.y ^ vcb.a.x ^ vcb.a.y ^ vcb.a.z ^ vcb.a.w ^ vcb.b.x ^ vcb.b.y) & 0x3f800000u);
                return;
#endif
                if (ci == 0) {
                    st.m = -INFINITY; st.l = 0.f;
#pragma unroll
                    for (int j = 0; j < 6; ++j) { st.o[j] = f32x2_t{0.f, 0.f}; q2[j] = f32x2_t{qp[dim_of(j)], qp[dim_of(j) + 1]}; }
                }
                const int k = k_lo + KPW * ci + kg;
                const u32 kw[6] = {kcb.a.x, kcb.a.y, kcb.a.z, kcb.a.w, kcb.b.x, kcb.b.y};
                f32x2_t acc = {0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 6; ++j) acc = mac2<TT>(kw[j], q2[j], acc);
                if (k == Lk) {   // the new token's own key is not in the cache yet: from the q | k | v exchange, as the cache will hold it
                    acc = f32x2_t{0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 6; ++j)
                        acc = __builtin_elementwise_fma(f32x2_t{round16<TT>(qp[kHeadDim + dim_of(j)]), round16<TT>(qp[kHeadDim + dim_of(j) + 1])}, q2[j], acc);
                }
                float d = acc.x + acc.y;
                d += dpp_xor1(d);
                d += dpp_xor2(d);
                d = (k < k_hi) ? d * kScaleQK : -INFINITY;
                const float m_new = fmaxf(st.m, d);
                if (m_new > -INFINITY) {
                    const float scale = __expf(st.m - m_new);   // exp(-inf) = 0 on the first chunk
                    const f32x2_t scale2 = {scale, scale};
                    st.l *= scale;
#pragma unroll
                    for (int j = 0; j < 6; ++j) st.o[j] *= scale2;
                    const u32 vw[6] = {vcb.a.x, vcb.a.y, vcb.a.z, vcb.a.w, vcb.b.x, vcb.b.y};
                    const float p = __expf(d - m_new);
                    const f32x2_t p2 = {p, p};
                    st.l += p;
                    if (k == Lk) {
#pragma unroll
                        for (int j = 0; j < 6; ++j)
                            st.o[j] = __builtin_elementwise_fma(p2, f32x2_t{round16<TT>(qp[2 * kHeadDim + dim_of(j)]), round16<TT>(qp[2 * kHeadDim + dim_of(j) + 1])}, st.o[j]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 6; ++j) st.o[j] = mac2<TT>(vw[j], p2, st.o[j]);
                    }
                    st.m = m_new;
                }
            };
            for (int c0 = 0; c0 < total; c0 += NB) {
#pragma unroll
                for (int bfr = 0; bfr < NB; ++bfr) {
                    if (c0 + bfr < total) {
                        kv_wait(kc[bfr], vc[bfr]);
                        chunk(kc[bfr], vc[bfr]);
                        kv_req(kc[bfr], vc[bfr]);
                        if (++ci == nch) {
                            // the pair is complete: the wave's 16 lane groups fold to one (same dimensions: lanes l, l + 4, l + 8, l + 12 of a
                            // 16-lane row, then the four rows), the 8 waves' partials meet in LDS and 48 threads merge them in wave order
                            att_fold_dpp<0x124>(st);      // row_ror:4
                            att_fold_dpp<0x128>(st);      // row_ror:8
                            att_fold_swap<false>(st);
                            att_fold_swap<true>(st);
                            float* wp = lds + M_WP + ((pi & 1) * NW + wave) * WPS;
                            if (lane < LPK) {
#pragma unroll
                                for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x2_t*>(wp + dim_of(j)) = st.o[j];
                                if (lane == 0) { wp[48] = st.m; wp[49] = st.l; }
                            }
                            wg_barrier();
                            if (tid < kHeadDim) {
                                const float* wq = lds + M_WP + (pi & 1) * NW * WPS;
                                float M = wq[48];
#pragma unroll
                                for (int ww = 1; ww < NW; ++ww) M = fmaxf(M, wq[ww * WPS + 48]);
                                float Ls = 0.f, o = 0.f;
#pragma unroll
                                for (int ww = 0; ww < NW; ++ww) {
                                    const float e = (M > -INFINITY) ? __expf(wq[ww * WPS + 48] - M) : 0.f;
                                    Ls = fmaf(e, wq[ww * WPS + 49], Ls);
                                    o = fmaf(e, wq[ww * WPS + tid], o);
                                }
                                const int p = w + 32 * pi;
                                put_local(gatt + (p >> 4) * E, (u32)((p & 15) * kHeadDim + tid), tg + 2, o * __builtin_amdgcn_rcpf(Ls));
                            }
                            ci = 0;
                            ++pi;
                        }
                    }
                }
            }
            if (total > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the ring's last, unused requests: their registers are free again)
            wg_barrier();     // (the activation buffer is rewritten by P3's gather: the last pair's merge must be through with its scratch)
        }
        stamp(3);   // attention of this CU's pairs
        // c_proj: this CU's 24 rows as 2 tiles x this wave's 3 k-steps, fragments 12..17 of the mlp c_proj slice and c_fc tiles 4, 5: requested by
        // every item as soon as the key loop has freed its registers -- they fly while the group waits for the attention outputs (1.2 +
        // 1.2 + 1.6 MB per item and group, out of the L2 / the Infinity Cache after the layer's first block; resident they would cost
        // 72 VGPRs through the key loop, and the compiler parked 12 c_fc fragments in scratch memory instead)
        WFrags<2> fo;
        req_frags<2, true>(fo, lw.Wo, 24 * w, 24, wave, lane);
        u32x4_t wpl[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) wpl[j] = ldwk(wp2 + (long)(12 + j) * NT * 8, (u32)tid * 8u);
        WFrags<2> ffi;           // c_fc tiles 4, 5 (the register file holds four of the six tiles through the key loop)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 3; ++j) ffi.f[t][j] = ldwk(pf, (u32)((3 * (4 + t) + j) * 64 + lane) * 8u);
        // ================= P3: attention outputs -> c_proj -> x' =================
        {
            u32 need = 0;
#pragma unroll
            for (int k = 0; k < 12; ++k) if (tid + k * NT < ns * E) need |= 1u << k;
            poll_ms<12, UMGEN_MS_POLL_ALL_ATT>(c, tid, gatt, need, [&](int k) { return (u32)min(tid + k * NT, ns * E - 1); }, tg + 2,
                        [&](int k, float v) { const int f = tid + k * NT; ldsu[M_XS + (f / E) * XST + f % E] = pack16<TT>(v); });
            wg_barrier();
        }
        stamp(4);   // waited for the attention outputs
        {
            typename Mma16<TT>::vec bx[3];
            typedef typename Mma16<TT>::vec vec;
            load_bfrags_ms<TT>(ldsu + M_XS, XST, ns, 96 * wave, lane, bx);
            f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t] = Mma16<TT>::mfma(__builtin_bit_cast(vec, fo.f[t][j]), bx[j], acc[t]);
#pragma unroll
            for (int t = 0; t < 2; ++t) store_tile(lds + M_ST + wave * MS * RS, ns, lane, t, acc[t]);
        }
        wg_barrier();
        if (tid < 24 * ns) {
            const int si = tid / 24, r = tid - 24 * si;
            float v = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) v += lds[M_ST + (ww * MS + si) * RS + r];
            xres2 = lds[M_XR + si * 24 + r] + (v + bo);
            put_local(gxb + si * E, (u32)(24 * w + r), tg + 3, xres2);
        }
        stamp(5);   // c_proj rows
        // ================= P4: x' -> LN -> c_fc -> GELU -> this CU's partial sums of the mlp c_proj =================
        {
            u32 need = 0;
#pragma unroll
            for (int k = 0; k < 12; ++k) if (tid + k * NT < ns * E) need |= 1u << k;
            poll_ms<12, true>(c, tid, gxb, need, [&](int k) { return (u32)min(tid + k * NT, ns * E - 1); }, tg + 3,
                        [&](int k, float v) { const int f = tid + k * NT; lds[M_XS + (f / E) * XST + f % E] = v; });
            wg_barrier();
        }
        stamp(6);   // waited for x'
        if (wave < ns) ln_pack<TT, false>(lds + M_XS + wave * XST, lds + M_LN + E, lane, w, nullptr);
        wg_barrier();
        {
            typename Mma16<TT>::vec bx[3];
            typedef typename Mma16<TT>::vec vec;
            load_bfrags_ms<TT>(ldsu + M_XS, XST, ns, 96 * wave, lane, bx);
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {     // two passes of three tiles: 12 accumulator registers live instead of 24
                f32x4_t a3[3] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        const int tt = 3 * hf + t;
                        a3[t] = Mma16<TT>::mfma(__builtin_bit_cast(vec, tt < 4 ? ff.f[tt < 4 ? tt : 0][j] : ffi.f[tt < 4 ? 0 : tt - 4][j]), bx[j], a3[t]);
                    }
#pragma unroll
                for (int t = 0; t < 3; ++t) store_tile(lds + M_ST + wave * MS * RS, ns, lane, 3 * hf + t, a3[t]);
            }
        }
        wg_barrier();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int o = tid + it * NT;
            if (o < 96 * ns) {   // hidden unit 96 w + r of scene si: sum of the 8 k ranges -> exact GELU -> packed for the mlp projection's B operand
                const int si = o / 96, r = o - 96 * si;
                float v = 0.f;
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) v += lds[M_ST + (ww * MS + si) * RS + r];
                ldsu[M_HH + si * HST + r] = pack16<TT>(gelu_erf(v));
            }
        }
        wg_barrier();
        stamp(7);   // LN + c_fc rows + GELU
        {
            // this CU's partial sums of the 768 mlp c_proj outputs over its 96 hidden units: wave `wave` takes rows 96 wave .. + 95 (6 tiles) x
            // all 96 k (3 k-steps); its 18 fragments are the thread's repacked units (12 parked in LDS, 6 in registers)
            typename Mma16<TT>::vec bh[3];
            typedef typename Mma16<TT>::vec vec;
            load_bfrags_ms<TT>(ldsu + M_HH, HST, ns, 0, lane, bh);
            float* strip = lds + M_ST + wave * MS * RS;
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int f = 3 * t + j;
                    const u32x4_t wfrag = f < 12 ? w2p[f * NT] : wpl[f < 12 ? 0 : f - 12];
                    acc = Mma16<TT>::mfma(__builtin_bit_cast(vec, wfrag), bh[j], acc);
                }
                store_tile(strip, ns, lane, t, acc);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the wave's own strip: no workgroup barrier)
            for (int i = lane; i < 96 * ns; i += 64) {
                const int si = i / 96, r = i - 96 * si;
                put_local(gpy + (long)(w * MS + si) * E, (u32)(96 * wave + r), tg + 4, strip[si * RS + r]);
            }
        }
        stamp(8);   // mlp partial sums
        // ================= P5: the 32 partial sums of this CU's 24 rows -> x'' (next layer's x) =================
        {
            // slot f = (producer p, scene si, row r) in a fixed [32][8][24] index space (slots of scenes past the block are not waited for)
            float* part = lds + M_XS;
            u32 need = 0;
#pragma unroll
            for (int k = 0; k < 12; ++k) if (((tid + k * NT) / 24) % MS < ns) need |= 1u << k;
            poll_ms<12, true>(c, tid, gpy + 24 * w, need,
                        [&](int k) { const u32 f = (u32)(tid + k * NT); return (f / 192u) * (u32)(MS * E) + ((f / 24u) % (u32)MS) * (u32)E + f % 24u; }, tg + 4,
                        [&](int k, float v) { part[tid + k * NT] = v; });
            wg_barrier();
            stamp(9);   // waited for the partial sums
            if (tid < 24 * ns) {
                const int si = tid / 24, r = tid - 24 * si;
                float gsum[4];
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    float s = 0.f;
#pragma unroll
                    for (int p = 0; p < 8; ++p) s += part[((8 * gq + p) * MS + si) * 24 + r];
                    gsum[gq] = s;
                }
                const float xn = xres2 + ((gsum[0] + gsum[1]) + (gsum[2] + gsum[3]));
                const int n = 24 * w + r;
                if (l + 1 == a.n_layers) {
                    (a.xdec + (long)(s0 + si) * E)[(u32)n] = xn;
                    if (a.xfrag) a.xfrag[frag_index(s0 + si, n)] = xn;          // the copy the head launch (rows_mfma_kernel) streams
                } else {
                    put_far(a.gx + (long)(s0 + si) * E, (u32)n, tg + 8, xn);
                }
            }
            wg_barrier();     // (the next item's gather rewrites the activation buffer)
        }
        stamp(10);  // mlp c_proj rows
        if (STAMPS && timer) a.stamps[15] += 1;
    }
}

size_t oar_engine_ms_lds_bytes() { return (size_t)M_TOTAL * sizeof(float); }   // > 80 KB: never two workgroups on one CU

hipError_t oar_engine_ms_prepare() {
    for (const void* f : {reinterpret_cast<const void*>(oar_engine_ms_kernel<false, bf16_t>), reinterpret_cast<const void*>(oar_engine_ms_kernel<true, bf16_t>),
                          reinterpret_cast<const void*>(oar_engine_ms_kernel<false, f16_t>), reinterpret_cast<const void*>(oar_engine_ms_kernel<true, f16_t>)}) {
        hipError_t rc = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)oar_engine_ms_lds_bytes());
        if (rc != hipSuccess) return rc;
    }
    return hipSuccess;
}

template <typename TT>
static void launch_ms_t(hipStream_t s, const OarMsArgs& a) {
    const dim3 grid(a.NG * kEngGroup), block(kEngThreads);
    const size_t shm = oar_engine_ms_lds_bytes();
    if (a.stamps) hipLaunchKernelGGL((oar_engine_ms_kernel<true, TT>), grid, block, shm, s, a);
    else hipLaunchKernelGGL((oar_engine_ms_kernel<false, TT>), grid, block, shm, s, a);
}

hipError_t launch_oar_engine_ms(hipStream_t s, const OarMsArgs& a) {
    if (a.fp16) launch_ms_t<f16_t>(s, a); else launch_ms_t<bf16_t>(s, a);
    return hipGetLastError();
}

}  // namespace umgen
