// Host side of libumgen_hip.so: the C ABI of include/umgen.h and the per-frame orchestration of
// UMGen._inference (UMGen.py:1406-1540): ego net -> pose shift -> three TAR stacks -> conditioning rows -> OAR decode loop.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdarg>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <map>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/umgen.h"
#include "bg_queue.h"
#include "frame.h"
#include "kernels.h"

using namespace umgen;

namespace umgen { thread_local BgRecorder* g_bg_rec = nullptr; }      // bg_queue.h: installed around the recording of a background pass

#define HIPCHK(e, call)                                                                              \
    do {                                                                                             \
        hipError_t _err = (call);                                                                    \
        if (_err != hipSuccess) return (e)->fail(UMGEN_E_HIP, "%s -> %s", #call, hipGetErrorString(_err)); \
    } while (0)

namespace {

struct AttnW { void* Wqkv; float* bqkv; void* Wo; float* bo; };
struct MlpW { void* Wfc; void* Wproj; };
struct SubW { float* ln_a; AttnW attn; float* ln_b; MlpW mlp; };
struct TarW { SubW sub[3]; };
struct DecW { float* ln1; AttnW self; float *ln2, *ln3; void* Wq; float* bq; void* Wkv; float* bkv; void* Wco; float* bco; float* ln4; MlpW mlp; };

struct Slot {            // destination of one state-dict entry
    void* dst;
    std::vector<int64_t> shape;
    int kind;            // 0: fp32 param, 1: T (precision dtype) weight, 2: bf16 table
    bool loaded;
    bool optional;
};

}  // namespace

struct umgen_engine {
    umgen_config cfg{};
    int E = 0, H = 0;
    size_t tsz = 4;                      // sizeof(T)
    hipStream_t stream = nullptr;
    std::string err = "";
    std::map<std::string, Slot> slots;
    std::vector<void*> allocs;
    bool finalized = false;
    // weights
    std::vector<TarW> stk[4];            // STACK_EGO, STACK_MAP, STACK_BOX, STACK_TAR
    std::vector<SubW> oar;
    std::vector<DecW> dec;
    float *ln_ego_tar = nullptr, *ln_ego = nullptr, *ln_tar = nullptr, *ln_oar = nullptr, *ln_map_tar = nullptr, *ln_box_tar = nullptr;
    void *head_ego = nullptr, *head_ar_map = nullptr, *head_ar_box = nullptr, *head_tar_box = nullptr, *head_ar_img = nullptr;
    void *map_fc = nullptr, *map_proj = nullptr, *img_fc = nullptr, *img_proj = nullptr;
    float *map_cb = nullptr, *img_cb = nullptr;
    EmbedTables tb{};
    // workspace
    float *X = nullptr, *mapfeat = nullptr, *warped_last = nullptr, *cond = nullptr, *pose_diff = nullptr, *pego = nullptr;
    void *A = nullptr, *QKV = nullptr, *VT = nullptr, *Hb = nullptr;
    float *xdec = nullptr, *qdec = nullptr, *part = nullptr, *hdec = nullptr, *logits = nullptr, *logits_tar = nullptr, *qkv3 = nullptr;
    void* kvcache = nullptr;
    void* vtcache = nullptr;            // UMGEN_ENG_MFMA & 16: the decode engine's dim-major copy of V [layer][scene][H][48][Lmax]
    long vt_layer_stride = 0, vt_scene_stride = 0;
    long kv_layer_stride = 0, kv_scene_stride = 0;
    int Lmax = kAttnSplit * kAttnChunk, S_pad = 2240;   // cache rows per head: every split's fixed key range is addressable
    int *d_pose = nullptr, *d_pose_shift = nullptr, *d_map = nullptr, *d_box = nullptr, *d_img = nullptr;
    int *d_tokens = nullptr, *d_prev_box = nullptr, *d_forced = nullptr, *d_counters = nullptr, *d_nboxes = nullptr, *d_ego_tok = nullptr;
    unsigned char* d_control = nullptr;
    double* d_boxes = nullptr;
    unsigned long long* d_seeds = nullptr;
    OarState* d_state = nullptr;
    // Overlapped TAR pass.  Causal temporal attention + frame-local spatial attention make every history
    // slot but the last one of the NEXT frame's window independent of the frame being decoded, so those slots are pushed through
    // the ego / map / box / TAR stacks on `bg_stream` (a CU-masked stream) while the latency-bound decode loop runs on the
    // other CUs; their temporal k | v rows are kept per layer in `tcache`.  The next frame then only computes its last slot.
    bool overlap = false, overlap_suspended = false;
    bool conc_stacks = false;            // plain path: the map / box stacks on side streams beside the TAR stack (UMGEN_CONCURRENT_STACKS, default on)
    int last_B = 0;
    float last_full_pre_ms = 0.f, last_oar_ms = 0.f;   // ego + TAR phase of the last whole-window frame / decode loop of the last frame
    int overlap_mode = 1;                // UMGEN_OVERLAP: 0 off, 1 on for one scene per GPU (default), 2 always
    hipStream_t bg_stream = nullptr;
    // the last-slot passes of the map / box stacks run beside the TAR stack's on their own streams and 1-slot workspaces
    struct Work { float* X; void *A, *QKV, *VT, *Hb; float* mapfeat; };
    Work w_main{}, w_side[2] = {};
    hipStream_t side_stream[2] = {nullptr, nullptr};
    hipEvent_t ev_side_in = nullptr, ev_side_done[2] = {nullptr, nullptr};
    void set_work(const Work& w) { X = w.X; A = w.A; QKV = w.QKV; VT = w.VT; Hb = w.Hb; mapfeat = w.mapfeat; }
    hipStream_t full_stream = nullptr;   // unmasked: whole-window passes, profiling frames and rollouts that do not overlap use all CUs
    hipEvent_t ev_pre_done = nullptr;
    hipEvent_t ev_tar_done = nullptr, ev_bg_done = nullptr, ev_bg0 = nullptr;
    bool bg_pending = false;
    // The overlapped pass ON THE DECODE ENGINE'S IDLE XCDs (round 6; bg_worker.h): one scene per GPU runs the engine on 4 of the 8 XCD groups (same step
    // time) and the engine workgroups of the other four execute the pass as an op list, recorded from the very launchers of the stand-alone kernels
    // (BgRecorder).  No second stream, no CU masks: the pass advances inside the decode steps' launches and is drained behind the frame's last step.
    bool bg_engine = false;
    BgQueue* d_bgq = nullptr;
    struct BgHead { unsigned w[4]; EmbedTables tb; } bg_head{};      // staging of the queue's header (must outlive the asynchronous upload)
    BgRecorder bg_rec;                   // host copy of the op list in flight (kept: the asynchronous uploads read it, and the next pass is compared with it)
    std::vector<unsigned> bg_state_host; // worker states read back behind the drain
    hipEvent_t ev_drain0 = nullptr, ev_drain1 = nullptr;
    std::vector<void*> tcache[4];        // per stack, per BlockTAR: [max_batch][max_cond_frames][S_stack][2E] of T
    // Growing window in the FOREGROUND (control mode starts with 13 history frames and grows to the cap,
    // infer_fun.py:64-71, UMGen.py:1600-1603): while the window grows, slot t of frame n + 1's window is slot t of frame n's window --
    // same tokens (the control overwrite of the last bbox3d frame persists, UMGen.py:1465-1467), same tpe row, and every later block
    // sees it through frame-local spatial sub-blocks and a CAUSAL temporal sub-block (module.py:332-359) -- so a frame that is followed
    // by a longer window leaves the temporal k | v rows of all its slots in the slot caches (allocated on first use), and the next
    // frame pushes only its new last slot through the stacks.  No second stream: this is the production path with the decode engine.
    bool grow_cache = true;              // UMGEN_GROW_CACHE=0: recompute the whole window every frame, like the reference
    int tcache_state = 0;                // 0 not tried, 1 allocated, -1 does not fit (plain path)
    struct Prefix {
        bool valid = false, has_ego = false;
        int B = 0, P = 0, Tfull = 0;
        std::vector<int> pose, map, box, img;   // slots 0..P-1 of the window the pass was computed for, [B][P][S_mod]
        std::vector<int> pose_next;             // [B][3] pose tokens the new frame must carry (they sit in shifted slot P-1)
    } px;
    std::vector<int> px_up[4], px_pshift;   // host staging of the background pass's uploads (must outlive the async copies)
    std::vector<float> px_pdiff;
    // timing
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    umgen_timings tm{};
    bool profiling = false;
    int rows_per_block = -1;              // few-row launches: rows per workgroup; -1 = by row count (1 up to 6 rows, else 2), 0 = row loop
    bool dbg_same_layer = false;          // UMGEN_DEBUG_SAME_LAYER=1: timing experiment, every decode layer reads layer 0's weights
    // decode step graphs per (kind: fixed / map / bbox3d / image, number of attention key splits 1..8)
    hipGraphExec_t step_graph[4][kAttnSplit + 1] = {};
    int step_graph_B = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> gemm_ev, attn_ev, layer_ev;
    size_t gemm_ev_used = 0, attn_ev_used = 0, layer_ev_used = 0;
    // XCD-resident decode engine (oar_engine.hip): one launch per decode step instead of five per layer
    struct EngStream { bool ok = false; int NG = 0; unsigned char map[16]; };
    bool eng_enabled = false, eng_fallback = false;   // eng_fallback: wanted, but the census failed (umgen_timings::engine_fallback)
    EngStream eng_fg, eng_full;          // census of the decode stream (CU-masked when the overlap exists) and of the unmasked stream
    OarLayerDev* d_layers = nullptr;
    unsigned long long *eng_gx = nullptr, *eng_gloc = nullptr;
    unsigned int *eng_ticket = nullptr, *eng_err = nullptr;
    std::vector<void*> eng_wp2;             // per BlockOAR: mlp c_proj repacked for the engine's hidden-unit split (repack_mlp_proj)
    std::vector<void*> eng_wf2;             // per BlockOAR: c_fc as matrix-core fragments (UMGEN_ENG_MFMA & 4)
    unsigned long long* eng_stamps = nullptr;   // UMGEN_DEBUG_TIMING: per-phase ticks of the engine (printed at destroy)
    size_t eng_gloc_bytes = 0;
    // Chip-wide decode engine for wide layers (oar_engine_wide.hip; n_embd 1536; one launch per scene and step).  Default: engines created for ONE scene per
    // call (two scenes as two launches: step 1696 us against 1523 on five launches per layer, four: 3382 against 2217 -- profiles/r05_wide2x_engine.txt);
    // UMGEN_DECODE_WIDE=n (1..4): engines of up to n scenes per call; =0: five launches per layer
    bool wide_enabled = false;
    OarLayerDev* d_layers_wide = nullptr;
    std::vector<void*> wide_wp2;            // per BlockOAR: mlp c_proj repacked [256 ranks][E rows][24 hidden units of the rank]
    unsigned long long* wide_gran = nullptr;
    unsigned int *wide_ticket = nullptr, *wide_err = nullptr;
    unsigned long long* wide_stamps = nullptr;
    bool use_wide(int B) const { return wide_enabled && tsz == 2 && B <= 4; }
    void* burn_buf = nullptr;             // UMGEN_DEBUG_BURN (measurement builds): the synthetic load's stream buffer
    int fg_xcds = 8;
    unsigned eng_epoch = 16u;             // first hand-off tag of the next frame (see run_frame)
    int step_graph_NG = -1;
    bool in_capture = false;
    const EngStream* eng_for(hipStream_t s) const {
        if (!eng_enabled) return nullptr;
        const EngStream* es = (full_stream && s == full_stream) ? &eng_full : &eng_fg;
        return es->ok ? es : nullptr;
    }
    double gemm_flops_pending = 0, attn_flops_pending = 0;
    // Batched decode layer (decode_batched.hip) from `batched_min` scenes per launch on (UMGEN_DECODE_BATCHED=n; 0 = never): the weights
    // once per step for the whole batch, the scenes as the matrix-core instruction's B-columns
    int batched_min = 24;               // measured crossover with the engine (profiles/r04_lanes_sweep.txt): 20 scenes 1546 (engine) vs 1674 us per step, 24: 1843 vs 1708, 28: 2104 vs 1775
    float *xfrag = nullptr, *afrag = nullptr, *hfrag = nullptr;   // fragment-major x / attention output [64 E], gelu(c_fc) [64 x 4E] of the batched layer
    bool use_batched(int B) const {      // (in_lanes: a lane's sub-batch of a batch that qualified)
        return tsz == 2 && (in_lanes || (batched_min > 0 && B >= batched_min)) && B <= kRowsMaxM && E % 32 == 0 && E <= 768;
    }
    // Decode LANES: the scenes of a batch are independent until the frame is complete (own K/V rows, own sampler state, own RNG
    // stream), and a batched layer launch for <= 16 scenes is latency-bound (5 dependent launches per layer, 48 - 96 workgroups each,
    // 34 us per layer whatever the batch is) while its attention launch is the only part at the HBM roof.  The batch is therefore cut
    // into `lanes` sub-batches, each with its own stream, OarState, fragment buffers and step graphs, forked once behind the TAR stacks
    // and joined once before the token download: one lane's weight GEMMs run in the shadow of another lane's K/V stream.  Tokens are
    // those of the single-lane batched layer bit for bit (a scene's column never mixes with another's, decode_batched.hip).
    static constexpr int kMaxLanes = 8;
    struct DecLane {
        hipStream_t s = nullptr;
        hipEvent_t done = nullptr;
        OarState* st = nullptr;
        float *xfrag = nullptr, *afrag = nullptr, *hfrag = nullptr;
        hipGraphExec_t graph[4][3] = {};
    };
    DecLane lane[kMaxLanes];
    hipEvent_t ev_lane_fork = nullptr;
    int lanes_env = -1;                  // UMGEN_DECODE_LANES=n: n lanes whenever the batched layer runs (1 = off); -1: by batch size
    int lane_graph_B = 0, lane_graph_n = 0;
    bool in_lanes = false;               // enqueueing a lane's steps (a profiled frame times them around the graph launches, not inside enqueue_step)
    int lane_count(int B) const {
        if (!use_batched(B) || !lane[0].s) return 1;
        // measured (profiles/r04_lanes_sweep.txt): lanes of 16 scenes (one full column block of the matrix-core instruction) are best --
        // 32 scenes 2132 / 1829 / 2040 us per step on 1 / 2 / 4 lanes, 64 scenes 3284 / 2689 / 2532 on 1 / 2 / 4; the device runs four
        // streams' kernels at a time (8 lanes: two rounds, 3784 / 4063 us)
        // ; from the threshold of 24 scenes on at least two lanes (24 scenes: 1978 us on one lane, 1708 on two)
        // ; lanes of at most 16 scenes: 40 scenes 2285 / 2020 / 2136 us on 2 / 3 / 4 lanes, 48: 2182 / 2231 on 3 / 4, 56: 2570 / 2369 on 3 / 4
        int n = lanes_env > 0 ? lanes_env : std::min(4, std::max(B >= 24 ? 2 : 1, (B + 15) / 16));
        return std::max(1, std::min(std::min(n, kMaxLanes), B));
    }
    hipError_t launch_status = hipSuccess;   // first refused kernel launch of the frame (hipGetLastError behind the GEMM launches): fails the frame

    int fail(int code, const char* fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }
};

namespace {

int dev_alloc(umgen_engine* e, void** p, size_t bytes) {
    HIPCHK(e, hipMalloc(p, bytes ? bytes : 16));
    e->allocs.push_back(*p);
    return 0;
}
template <typename P>
int dalloc(umgen_engine* e, P** p, size_t n) { return dev_alloc(e, reinterpret_cast<void**>(p), n * sizeof(P)); }

void reg(umgen_engine* e, const std::string& key, void* dst, std::vector<int64_t> shape, int kind, bool optional = false) {
    e->slots[key] = Slot{dst, std::move(shape), kind, false, optional};
}

int alloc_f32(umgen_engine* e, const std::string& key, float** p, std::vector<int64_t> shape) {
    size_t n = 1;
    for (auto d : shape) n *= (size_t)d;
    if (int rc = dalloc(e, p, n)) return rc;
    reg(e, key, *p, shape, 0);
    return 0;
}
int alloc_w(umgen_engine* e, const std::string& key, void** p, std::vector<int64_t> shape) {
    size_t n = 1;
    for (auto d : shape) n *= (size_t)d;
    if (int rc = dev_alloc(e, p, n * e->tsz)) return rc;
    reg(e, key, *p, shape, 1);
    return 0;
}

int alloc_attn(umgen_engine* e, const std::string& pre, AttnW& a) {
    const int64_t E = e->E;
    if (int rc = alloc_w(e, pre + ".c_attn.weight", &a.Wqkv, {3 * E, E})) return rc;
    if (int rc = alloc_f32(e, pre + ".c_attn.bias", &a.bqkv, {3 * E})) return rc;
    if (int rc = alloc_w(e, pre + ".c_proj.weight", &a.Wo, {E, E})) return rc;
    return alloc_f32(e, pre + ".c_proj.bias", &a.bo, {E});
}
int alloc_mlp(umgen_engine* e, const std::string& pre, MlpW& m) {
    const int64_t E = e->E;
    if (int rc = alloc_w(e, pre + ".c_fc.weight", &m.Wfc, {4 * E, E})) return rc;
    return alloc_w(e, pre + ".c_proj.weight", &m.Wproj, {E, 4 * E});
}
int alloc_sub(umgen_engine* e, const std::string& pre, const char* ln_a, const char* attn, const char* ln_b, const char* mlp, SubW& s) {
    const int64_t E = e->E;
    if (int rc = alloc_f32(e, pre + "." + ln_a + ".weight", &s.ln_a, {E})) return rc;
    if (int rc = alloc_attn(e, pre + "." + attn, s.attn)) return rc;
    if (int rc = alloc_f32(e, pre + "." + ln_b + ".weight", &s.ln_b, {E})) return rc;
    return alloc_mlp(e, pre + "." + mlp, s.mlp);
}

// ---- numpy-faithful host helpers (unfused double arithmetic) ----------------------------------------------
#pragma clang fp contract(off)
double lin_bin(int i, double start, double stop, int n) {   // np.linspace(start, stop, n)[i]
    if (i >= n - 1) return stop;
    const double step = (stop - start) / (double)(n - 1);
    volatile double t = (double)i * step;
    return t + start;
}
// UMGen.decode_pose (UMGen.py:1008-1024): DigitalBinsTokenizer.decode + Normalize_Standard.unnormalize_ego
float decode_pose_value(int tok, int axis) {
    const float stdv[3] = {10.0f, 4.0f, 1.0f};
    const float inv_std = 1.0f / stdv[axis];                 // np.float32 division (normalize.py:26)
    const int right = std::min(std::max(tok, 0), 1023), left = std::min(std::max(tok - 1, 0), 1023);
    volatile double s = lin_bin(left, -1.0, 1.0, 1024) + lin_bin(right, -1.0, 1.0, 1024);
    volatile double v = s / 2.0;
    volatile double u = v / (double)inv_std;
    return (float)(u + 0.0);
}
// module.py:746-768 position_encoding_init -> bf16
void sinusoid_table(int n_position, int E, int start_index, std::vector<bf16_t>& out) {
    out.assign((size_t)n_position * E, 0);
    for (int pos = 1; pos < n_position; ++pos)
        for (int j = 0; j < E; ++j) {
            const double denom = std::pow(10000.0, 2.0 * (double)(j / 2) / (double)E);
            const double a = (double)(pos + start_index) / denom;
            const double v = (j % 2 == 0) ? std::sin(a) : std::cos(a);
            // double -> bf16 round-to-nearest-even (via the exactly-representable float when possible)
            float f = (float)v;
            // correct double rounding: if the float rounding moved across a bf16 tie, fix it up
            bf16_t b = f32_to_bf16(f);
            const double lo = (double)bf16_to_f32((bf16_t)(b - 1)), hi = (double)bf16_to_f32((bf16_t)(b + 1)), mid = (double)bf16_to_f32(b);
            // choose nearest of the three candidates to v (ties to even mantissa)
            double best = mid;
            bf16_t bb = b;
            const double cands[2] = {lo, hi};
            const bf16_t cb[2] = {(bf16_t)(b - 1), (bf16_t)(b + 1)};
            for (int c = 0; c < 2; ++c) {
                const double d1 = std::fabs(cands[c] - v), d0 = std::fabs(best - v);
                if (d1 < d0 || (d1 == d0 && (cb[c] & 1) == 0 && (bb & 1) == 1)) { best = cands[c]; bb = cb[c]; }
            }
            out[(size_t)pos * E + j] = bb;
        }
}

template <typename T> void convert_to(const void* src, int dtype, size_t n, T* dst);
float load_as_f32(const void* src, int dtype, size_t i) {
    switch (dtype) {
        case UMGEN_DT_F32: return reinterpret_cast<const float*>(src)[i];
        case UMGEN_DT_BF16: return bf16_to_f32(reinterpret_cast<const bf16_t*>(src)[i]);
        case UMGEN_DT_F64: return (float)reinterpret_cast<const double*>(src)[i];
        case UMGEN_DT_F16: {
            const uint16_t h = reinterpret_cast<const uint16_t*>(src)[i];
            const uint32_t sign = (h >> 15) & 1, ex = (h >> 10) & 31, man = h & 1023;
            float v;
            if (ex == 0) v = std::ldexp((float)man, -24);
            else if (ex == 31) v = man ? NAN : INFINITY;
            else v = std::ldexp((float)(man | 1024), (int)ex - 25);
            return sign ? -v : v;
        }
    }
    return 0.f;
}

// ---- templated compute path ---------------------------------------------------------------------------------
template <typename T> struct Path;
template <> struct Path<float> {
    static void gemm(hipStream_t s, const GemmArgs& a) { launch_gemm_valu<float, float>(s, a); }
    static void attn_spatial(hipStream_t s, const float* qk, const float* vt, float* y, int F, int S, int Sp, int H) {
        launch_attn_spatial_f32_mfma(s, qk, vt, y, F, S, Sp, H);
    }
    static void attn_causal(hipStream_t s, const float* qk, const float* vt, float* y, int F, int S, int Sp, int H) {
        launch_attn_causal_f32(s, qk, vt, y, F, S, Sp, H);
    }
    static void gemm_w_f32act(hipStream_t s, const GemmArgs& a) { launch_gemm_valu<float, float>(s, a); }
};
template <typename TT> struct Path16 {   // bf16_t / f16_t: the same matrix-core kernels with the other operand type
    static void gemm(hipStream_t s, const GemmArgs& a) { launch_gemm_mfma<TT>(s, a); }
    static void attn_spatial(hipStream_t s, const TT* qk, const TT* vt, TT* y, int F, int S, int Sp, int H) {
        launch_attn_spatial_mfma<TT>(s, qk, vt, y, F, S, Sp, H);
    }
    static void attn_causal(hipStream_t s, const TT* qk, const TT* vt, TT* y, int F, int S, int Sp, int H) {
        launch_attn_causal_mfma<TT>(s, qk, vt, y, F, S, Sp, H);
    }
    static void gemm_w_f32act(hipStream_t s, const GemmArgs& a) { launch_gemm_valu<TT, float>(s, a); }
};
template <> struct Path<bf16_t> : Path16<bf16_t> {};
template <> struct Path<f16_t> : Path16<f16_t> {};

template <typename T>
void gemm_timed(umgen_engine* e, const GemmArgs& a) {
    if (e->profiling) {
        if (e->gemm_ev_used == e->gemm_ev.size()) {
            hipEvent_t a0, a1;
            hipEventCreate(&a0);
            hipEventCreate(&a1);
            e->gemm_ev.emplace_back(a0, a1);
        }
        auto& pr = e->gemm_ev[e->gemm_ev_used++];
        hipEventRecord(pr.first, e->stream);
        Path<T>::gemm(e->stream, a);
        hipEventRecord(pr.second, e->stream);
        e->gemm_flops_pending += 2.0 * (double)a.Mi * (double)a.Nj * (double)a.K * (double)a.batch;
    } else {
        Path<T>::gemm(e->stream, a);
    }
    // a refused launch (e.g. a dynamic-LDS attribute missing on this device) would leave stale workspace data behind: never silently
    if (!e->in_capture) { const hipError_t le = hipGetLastError(); if (le != hipSuccess && e->launch_status == hipSuccess) e->launch_status = le; }
}

// out[tokens R][N] (T) = A[R][K] . W[N][K]^T + bias  (optionally GELU)
template <typename T>
void linear_store(umgen_engine* e, const void* W, const float* bias, int N, int K, const void* A, long R, void* out, long ldo, int gelu) {
    GemmArgs g{};
    g.P = W; g.Q = A; g.Mi = N; g.Nj = (int)R; g.K = K; g.ldp = K; g.ldq = K; g.batch = 1;
    g.mode = GEMM_STORE; g.bias = bias; g.gelu = gelu; g.out = out; g.ldo = ldo;
    gemm_timed<T>(e, g);
}
// X[R][N] (fp32) += A[R][K] . W[N][K]^T + bias
template <typename T>
void linear_resid(umgen_engine* e, const void* W, const float* bias, int N, int K, const void* A, long R, float* X) {
    GemmArgs g{};
    g.P = W; g.Q = A; g.Mi = N; g.Nj = (int)R; g.K = K; g.ldp = K; g.ldq = K; g.batch = 1;
    g.mode = GEMM_RESID; g.bias = bias; g.out = X; g.ldo = N;
    gemm_timed<T>(e, g);
}

// one (LayerNorm -> attention -> residual -> LayerNorm -> MLP -> residual) sub-block of BlockTAR (module.py:332-359)
//
// `tail`: the reference consumes only the LAST frame of every stack's output (UMGen.py:1227-1231 takes
// [:, -1] of each TAR output, 1002 of the ego stack), the spatial attention / LayerNorm / MLPs are frame-local and the temporal
// attention is causal (module.py:332-359) -- so in a stack's FINAL block everything behind the temporal attention's k | v rows is
// evaluated for the last frame only: identical outputs (per-row arithmetic does not depend on which other rows are in the launch).
//   tail 0: all rows;  tail 1 (temporal sub-block): LN + k|v of all rows, q / attention output / projection / MLP of the last frame;
//   tail 2 (the spatial sub-block behind it): the whole sub-block on the last frame's rows.
template <typename T>
void tar_sub(umgen_engine* e, const SubW& w, int B, int Tn, int S, bool temporal, TemporalRange tr = TemporalRange{0, nullptr, 0, 0}, int tail = 0) {
    const int E = e->E, H = e->H;
    const long R = (long)B * Tn * S;
    T* A = reinterpret_cast<T*>(e->A);
    T* QKV = reinterpret_cast<T*>(e->QKV);
    T* Hb = reinterpret_cast<T*>(e->Hb);
    const char* Wqkv = reinterpret_cast<const char*>(w.attn.Wqkv);
    const size_t wrow = (size_t)E * sizeof(T);                      // bytes of one weight row of c_attn
    // row ranges the "rest" of the sub-block (projection, MLP) runs on: everything, or the last frame of each scene
    struct Range { long row0, rows; };
    std::vector<Range> rest;
    if (tail == 0) rest.push_back(Range{0, R});
    else for (int b = 0; b < B; ++b) rest.push_back(Range{((long)b * Tn + (Tn - 1)) * S, (long)S});
    auto spatial_attention = [&](long frame0, int frames) {         // q | k row-major, V transposed per (frame, head) for the attention kernel
        const long r0 = frame0 * S;
        linear_store<T>(e, Wqkv, w.attn.bqkv, 2 * E, E, A + r0 * E, (long)frames * S, QKV + r0 * 2 * E, 2L * E, 0);
        T* vt = reinterpret_cast<T*>(e->VT) + frame0 * (long)E * e->S_pad;
        GemmArgs g{};
        g.P = A + r0 * E; g.Q = Wqkv + (size_t)2 * E * wrow;
        g.Mi = S; g.Nj = E; g.K = E; g.ldp = E; g.ldq = E; g.strideP = (long)S * E; g.strideQ = 0; g.batch = frames;
        g.mode = GEMM_VT; g.bias = w.attn.bqkv + 2 * E; g.out = vt; g.ldo = e->S_pad; g.H = H;
        gemm_timed<T>(e, g);
        hipEvent_t t0 = nullptr, t1 = nullptr;
        if (e->profiling) {
            if (e->attn_ev_used == e->attn_ev.size()) {
                hipEvent_t a0, a1;
                hipEventCreate(&a0);
                hipEventCreate(&a1);
                e->attn_ev.emplace_back(a0, a1);
            }
            auto& pr = e->attn_ev[e->attn_ev_used++];
            t0 = pr.first; t1 = pr.second;
            hipEventRecord(t0, e->stream);
            e->attn_flops_pending += 4.0 * (double)S * (double)S * kHeadDim * (double)H * (double)frames;
        }
        Path<T>::attn_spatial(e->stream, QKV + r0 * 2 * E, vt, A + r0 * E, frames, S, e->S_pad, H);
        if (t1) hipEventRecord(t1, e->stream);
    };
    if (temporal) {
        launch_layernorm<T>(e->stream, e->X, E, R, E, w.ln_a, A);
        if (tail == 0) {
            linear_store<T>(e, Wqkv, w.attn.bqkv, 3 * E, E, A, R, QKV, 3L * E, 0);
        } else {   // k | v rows of every frame, q rows of the last frame only (the same [R][3E] row layout)
            linear_store<T>(e, Wqkv + (size_t)E * wrow, w.attn.bqkv + E, 2 * E, E, A, R, QKV + E, 3L * E, 0);
            for (const Range& r : rest) linear_store<T>(e, Wqkv, w.attn.bqkv, E, E, A + r.row0 * E, r.rows, QKV + r.row0 * 3 * E, 3L * E, 0);
            tr.q0 = tr.t0 + Tn - 1;
        }
        launch_attn_temporal<T>(e->stream, QKV, A, B, Tn, S, H, tr);
    } else if (tail == 0) {
        launch_layernorm<T>(e->stream, e->X, E, R, E, w.ln_a, A);
        spatial_attention(0, B * Tn);
    } else {
        for (const Range& r : rest) {
            launch_layernorm<T>(e->stream, e->X + r.row0 * E, E, r.rows, E, w.ln_a, A + r.row0 * E);
            spatial_attention(r.row0 / S, 1);
        }
    }
    for (const Range& r : rest) {
        float* X = e->X + r.row0 * E;
        linear_resid<T>(e, w.attn.Wo, w.attn.bo, E, E, A + r.row0 * E, r.rows, X);
        launch_layernorm<T>(e->stream, X, E, r.rows, E, w.ln_b, A + r.row0 * E);
        linear_store<T>(e, w.mlp.Wfc, nullptr, 4 * E, E, A + r.row0 * E, r.rows, Hb + r.row0 * 4 * E, 4L * E, 1);
        linear_resid<T>(e, w.mlp.Wproj, nullptr, E, 4 * E, Hb + r.row0 * 4 * E, r.rows, X);
    }
}

// cache_mode: 0 = one pass over the whole window; 1 = prefix pass (slots [0, w.T), k | v appended to the slot caches);
// 2 = last-slot pass (slots [w.t0, w.t0 + w.T) against the caches); 3 = whole window like 0, and its k | v rows are left in the slot
// caches for a longer window that follows; 4 = last-slot pass like 2 that appends its own k | v rows (the window keeps growing)
template <typename T>
void run_stack(umgen_engine* e, int stack, const WindowTokens& w, int cache_mode = 0) {
    const int S = stack_len(stack);
    launch_embed_stack(e->stream, stack, e->tb, w, e->X, e->mapfeat);
    if (stack != STACK_EGO) launch_warp_map(e->stream, stack, e->tb, w.B, w.T, e->mapfeat, e->pose_diff, e->X,
                                            stack == STACK_MAP ? e->warped_last : nullptr, w.Tfull, w.t0);
    static const bool no_tail = getenv("UMGEN_NO_TAIL") != nullptr;   // measurement: evaluate every block on every frame like the reference
    for (size_t i = 0; i < e->stk[stack].size(); ++i) {
        const TarW& blk = e->stk[stack][i];
        TemporalRange tr{w.t0, cache_mode ? e->tcache[stack][i] : nullptr, e->cfg.max_cond_frames, (cache_mode == 1 || cache_mode >= 3) ? 1 : 0};
        // the final block's tail on the last frame only (f-3): whole-window passes with more than one slot (the temporal sub-block still
        // evaluates the k | v rows of every slot, so a pass that fills the slot caches keeps the shortcut)
        const bool last = i + 1 == e->stk[stack].size() && (cache_mode == 0 || cache_mode == 3) && w.T > 1 && !no_tail;
        tar_sub<T>(e, blk.sub[0], w.B, w.T, S, false);
        tar_sub<T>(e, blk.sub[1], w.B, w.T, S, true, tr, last ? 1 : 0);
        tar_sub<T>(e, blk.sub[2], w.B, w.T, S, false, TemporalRange{0, nullptr, 0, 0}, last ? 2 : 0);
    }
}

// Few-row launches with several rows (scenes, ego queries): one row per workgroup keeps the single-row dependency chain (4 scenes:
// 2.58 vs 3.12 s of decode per frame) but re-reads the weights from L2 once per row; from 7 rows on, 2 rows per workgroup win
// (8 scenes: 3.53 vs 3.77 s).  UMGEN_ROWS_PER_BLOCK overrides (0 = every workgroup loops over all rows).
inline int rows_per_block_for(const umgen_engine* e, int M) { return e->rows_per_block >= 0 ? e->rows_per_block : (M <= 6 ? 1 : 2); }

// GemvArgs helpers
template <typename T>
void gemv(umgen_engine* e, const float* x, long ldx, const float* ln_w, const void* W, const float* bias, int N, int K, int M,
          int mode, float* out, long ldo) {
    GemvArgs a{};
    a.rows_per_block = rows_per_block_for(e, M);
    a.x = x; a.ldx = ldx; a.ln_w = ln_w; a.W = W; a.bias = bias; a.N = N; a.K = K; a.M = M; a.out_mode = mode; a.out = out; a.ldo = ldo;
    a.E = e->E;
    launch_gemv<T>(e->stream, a);
}
template <typename T>
void gemv_resid(umgen_engine* e, const float* a_in, long lda, const float* part, const void* W, const float* bias, int N, int K, int M,
                float* x, long ldx, int ns = 1) {
    GemvResidArgs a{};
    a.rows_per_block = rows_per_block_for(e, M);
    a.a = a_in; a.lda = lda; a.part = part; a.H = e->H; a.ns = ns; a.W = W; a.bias = bias; a.N = N; a.K = K; a.M = M; a.x = x; a.ldx = ldx;
    launch_gemv_resid<T>(e->stream, a);
}

// infer_ego_net / forward_ego_net (UMGen.py:994-1005, 634-687).  The Decoder is frame-local and only t = -1 is consumed
// (UMGen.py:1002), so the 12 decoder blocks run on the last frame only -- identical outputs, 1/T of the work.
template <typename T>
void run_ego(umgen_engine* e, const WindowTokens& w, const SamplerParams& sp, int frame_idx, bool forced, float* trace_logits,
             int cache_mode = 0) {
    const int E = e->E, H = e->H, B = w.B, Tn = w.T;   // Tn: slots in this pass (the last one is the window's last frame)
    run_stack<T>(e, STACK_EGO, w, cache_mode);
    // p = ln_ego_tar(x) of the last frame, kept in fp32 (every decoder block re-normalises it with its own ln_3)
    for (int b = 0; b < B; ++b)
        launch_layernorm<float>(e->stream, e->X + (((long)b * Tn + (Tn - 1)) * kSeq) * E, E, kSeq, E, e->ln_ego_tar,
                                e->pego + (long)b * kSeq * E);
    float* x = e->xdec;   // [3B][E] ego queries
    launch_ego_queries(e->stream, e->tb, B, w.Tfull ? w.Tfull : Tn, x);
    const int M = 3 * B;
    T* PN = reinterpret_cast<T*>(e->A);         // ln_3(p)           [B*2207][E]
    T* KV = reinterpret_cast<T*>(e->QKV);       // k | v of ln_3(p)  [B*2207][2E]
    for (const DecW& d : e->dec) {               // Decoder.forward_func (module.py:662-683)
        gemv<T>(e, x, E, d.ln1, d.self.Wqkv, d.self.bqkv, 3 * E, E, M, GEMV_OUT_F32, e->qkv3, 3L * E);
        // self-attention among the 3 ego queries of a scene (non-causal); q rows gathered out of the packed q|k|v rows
        hipMemcpy2DAsync(e->qdec, (size_t)E * 4, e->qkv3, (size_t)3 * E * 4, (size_t)E * 4, M, hipMemcpyDeviceToDevice, e->stream);
        launch_attn_partial<float>(e->stream, e->qdec, e->qkv3 + E, 3L * 3 * E, kHeadDim, 3L * E, E, M, 3, H, nullptr, 3, 1, e->part);
        gemv_resid<T>(e, nullptr, 0, e->part, d.self.Wo, d.self.bo, E, E, M, x, E, 1);
        // cross attention to the frame's 2207 scene tokens (FlashCrossAttention.forward, module.py:482-509)
        gemv<T>(e, x, E, d.ln2, d.Wq, d.bq, E, E, M, GEMV_OUT_F32, e->qdec, E);
        launch_layernorm<T>(e->stream, e->pego, E, (long)B * kSeq, E, d.ln3, PN);
        linear_store<T>(e, d.Wkv, d.bkv, 2 * E, E, PN, (long)B * kSeq, KV, 2L * E, 0);
        launch_attn_partial<T>(e->stream, e->qdec, KV, (long)kSeq * 2 * E, kHeadDim, 2L * E, E, M, 3, H, nullptr, kSeq, attn_nsplit(kSeq), e->part);
        gemv_resid<T>(e, nullptr, 0, e->part, d.Wco, d.bco, E, E, M, x, E, attn_nsplit(kSeq));
        gemv<T>(e, x, E, d.ln4, d.mlp.Wfc, nullptr, 4 * E, E, M, GEMV_OUT_GELU, e->hdec, 4L * E);
        gemv_resid<T>(e, e->hdec, 4L * E, nullptr, d.mlp.Wproj, nullptr, E, 4 * E, M, x, E);
    }
    gemv<T>(e, x, E, e->ln_ego, e->head_ego, nullptr, e->cfg.pose_vocab, E, M, GEMV_OUT_F32, e->logits, e->cfg.pose_vocab);
    if (trace_logits) hipMemcpyAsync(trace_logits, e->logits, (size_t)3 * e->cfg.pose_vocab * 4, hipMemcpyDeviceToHost, e->stream);
    launch_sample_ego(e->stream, e->logits, e->cfg.pose_vocab, sp, e->d_seeds, frame_idx, forced ? e->d_forced : nullptr, e->d_ego_tok, B,
                      e->d_counters + 7);
}

// one OAR decode step through the 36 BlockOAR layers (module.py:402-416) for the B scenes
template <typename T>
int oar_layers(umgen_engine* e, int B, int ns) {
    const int E = e->E, H = e->H;
    const int* d_len = &e->d_state->step;
    if constexpr (sizeof(T) == 2) {
        if (e->use_batched(B)) {      // five launches per layer for the whole batch: LN + q|k|v, attention, c_proj (+x), LN + c_fc + GELU, mlp c_proj (+x)
            // activations between the launches are fragment-major (decode_batched.hip); x also stays row-major in xdec (sampler, residual)
            launch_rows_to_frag(e->stream, e->xdec, E, B, E, e->xfrag);
            for (size_t li = 0; li < e->oar.size(); ++li) {
                const SubW& w = e->oar[e->dbg_same_layer ? 0 : li];
                T* cache = reinterpret_cast<T*>(e->kvcache) + (long)li * e->kv_layer_stride;
                RowsArgs r{};
                r.x = e->xfrag; r.M = B; r.ln_w = w.ln_a; r.W = w.attn.Wqkv; r.bias = w.attn.bqkv; r.N = 3 * E; r.K = E; r.mode = ROWS_QKV;
                r.out = e->qdec; r.ldo = E; r.cache = cache; r.scene_stride = e->kv_scene_stride; r.d_len = d_len; r.Lmax = e->Lmax; r.E = E;
                launch_rows_mfma<T>(e->stream, r);
                launch_attn_decode_batched<T>(e->stream, e->qdec, cache, e->kv_scene_stride, B, H, e->Lmax, d_len, e->afrag);
                RowsArgs p{};
                p.x = e->afrag; p.M = B; p.W = w.attn.Wo; p.bias = w.attn.bo; p.N = E; p.K = E; p.mode = ROWS_RESID; p.out = e->xdec; p.ldo = E; p.out_frag = e->xfrag; p.E = E;
                launch_rows_mfma<T>(e->stream, p);
                RowsArgs f{};
                f.x = e->xfrag; f.M = B; f.ln_w = w.ln_b; f.W = w.mlp.Wfc; f.N = 4 * E; f.K = E; f.mode = ROWS_GELU; f.out_frag = e->hfrag; f.E = E;
                launch_rows_mfma<T>(e->stream, f);
                RowsArgs q{};
                q.x = e->hfrag; q.M = B; q.W = w.mlp.Wproj; q.N = E; q.K = 4 * E; q.mode = ROWS_RESID; q.out = e->xdec; q.ldo = E; q.out_frag = e->xfrag; q.E = E;
                launch_rows_mfma<T>(e->stream, q);
            }
            return 0;
        }
    }
    if (sizeof(T) == 2 && e->use_wide(B)) {
        OarWideArgs a{};
        a.layers = e->d_layers_wide; a.n_layers = (int)e->oar.size();
        a.kvcache = reinterpret_cast<bf16_t*>(e->kvcache); a.kv_layer_stride = e->kv_layer_stride; a.kv_scene_stride = e->kv_scene_stride; a.Lmax = e->Lmax;
        a.xdec = e->xdec; a.st = e->d_state; a.gran = e->wide_gran; a.ticket = e->wide_ticket; a.err = e->wide_err;
        a.fp16 = std::is_same<T, f16_t>::value ? 1 : 0;
        a.stamps = e->wide_stamps;
        for (int b = 0; b < B; ++b) {
            a.scene = b;
            HIPCHK(e, launch_oar_engine_wide(e->stream, a));
        }
        return 0;
    }
    if (const umgen_engine::EngStream* es = sizeof(T) == 2 ? e->eng_for(e->stream) : nullptr) {
        OarEngineArgs a{};
        a.layers = e->d_layers; a.n_layers = (int)e->oar.size();
        a.kvcache = reinterpret_cast<bf16_t*>(e->kvcache); a.kv_layer_stride = e->kv_layer_stride; a.kv_scene_stride = e->kv_scene_stride; a.Lmax = e->Lmax;
        a.vtcache = reinterpret_cast<bf16_t*>(e->vtcache); a.vt_layer_stride = e->vt_layer_stride; a.vt_scene_stride = e->vt_scene_stride;
        a.xdec = e->xdec; a.st = e->d_state; a.gx = e->eng_gx; a.gloc = e->eng_gloc; a.ticket = e->eng_ticket; a.err = e->eng_err;
        a.B = B; a.NG = es->NG;
        a.R = es->NG;
        for (int r = 1; r <= es->NG; ++r)
            if (es->NG % r == 0 && r >= std::min(B, es->NG)) { a.R = r; break; }
        if (const char* dd = getenv("UMGEN_DEBUG_ENGINE_D")) {   // debugging: fewer groups per scene than the batch size asks for
            const int want = atoi(dd);
            if (want >= 1 && es->NG % want == 0 && es->NG / want >= std::min(B, es->NG)) a.R = es->NG / want;
        }
        a.D = es->NG / a.R;
        if (e->bg_engine && B == 1 && es->NG == 8 && !e->eng_stamps) {      // one scene on 4 XCD groups, the other 4 XCDs' workgroups are background workers (bg_worker.h)
            a.R = 2; a.D = 4;
            a.bg = e->d_bgq;
        }
        memcpy(a.xcc_group, es->map, 16);
        if (const char* bs = getenv("UMGEN_DEBUG_BURN")) {      // measurement builds (-DUMGEN_ENG_BURN): "us,mfma,sleep,kb"
            int us = 0, mf = 0, sl = 0, kb = 0;
            if (sscanf(bs, "%d,%d,%d,%d", &us, &mf, &sl, &kb) >= 1) {
                a.burn_ticks = us * 100; a.burn_mfma = mf; a.burn_sleep = sl; a.burn_kb = e->burn_buf ? kb : 0; a.burn_buf = e->burn_buf;
            }
        }
        a.stamps = e->eng_stamps;
        a.fp16 = std::is_same<T, f16_t>::value ? 1 : 0;
        // More than 4 scenes: the systolic schedule (oar_engine.hip) -- the scenes flow through the 8 groups, group g working on layers
        // g, g + 8, ...; up to 4 scenes keep disjoint group sets per scene (8 / 4 / 2 groups each), whose per-scene latency is lower.
        // Measured (profiles/r03_systolic.txt): 5 scenes 704 vs 1014 us per step, 8 scenes 1034 vs 1037 us (the register file cannot
        // keep a layer resident, so every item still streams 11 of its 14 MB).  UMGEN_ENGINE_SYSTOLIC=0/1 forces either form.
        static const char* sys_env = getenv("UMGEN_ENGINE_SYSTOLIC");
        a.systolic = sys_env ? (sys_env[0] == '1' && B > 1) : (B > 4);
        if (a.systolic && B > kEngMaxSystolic) a.systolic = 0;
        // hand-off tags of one step: (round or scene, layer, edge) must fit kEpochPerStep (umgen_create bounds n_oar_layer and max_batch)
        const int slots = a.systolic ? B : (B + a.R - 1) / a.R;
        if ((unsigned)(slots * 64 * 8) > kEpochPerStep) return e->fail(UMGEN_E_UNSUPPORTED, "decode engine: %d scenes need more hand-off tags than one step has", B);
        HIPCHK(e, launch_oar_engine(e->stream, a));
        return 0;
    }
    for (size_t li = 0; li < e->oar.size(); ++li) {
        const SubW& w = e->oar[e->dbg_same_layer ? 0 : li];
        T* cache = reinterpret_cast<T*>(e->kvcache) + (long)li * e->kv_layer_stride;
        {
            GemvArgs a{};
            a.x = e->xdec; a.ldx = E; a.ln_w = w.ln_a; a.W = w.attn.Wqkv; a.bias = w.attn.bqkv; a.N = 3 * E; a.K = E; a.M = B;
            a.out_mode = GEMV_OUT_QKV; a.out = e->qdec; a.ldo = E; a.cache = cache; a.scene_stride = e->kv_scene_stride; a.d_len = d_len;
            a.Lmax = e->Lmax; a.E = E; a.rows_per_block = rows_per_block_for(e, B);
            launch_gemv<T>(e->stream, a);
            launch_attn_partial<T>(e->stream, e->qdec, cache, e->kv_scene_stride, (long)e->Lmax * kHeadDim, kHeadDim,
                                   (long)H * e->Lmax * kHeadDim, B, 1, H, d_len, 1, ns, e->part);
            gemv_resid<T>(e, nullptr, 0, e->part, w.attn.Wo, w.attn.bo, E, E, B, e->xdec, E, ns);
        }
        gemv<T>(e, e->xdec, E, w.ln_b, w.mlp.Wfc, nullptr, 4 * E, E, B, GEMV_OUT_GELU, e->hdec, 4L * E);
        gemv_resid<T>(e, e->hdec, 4L * E, nullptr, w.mlp.Wproj, nullptr, E, 4 * E, B, e->xdec, E);
    }
    return 0;
}

// The GIVEN-token prefix of a frame as ONE forward pass (infer_oar_net's first iteration pushes the whole predefined prefix through the 36
// layers, UMGen.py:1184-1201, 1234-1237; rounds 1-4 replayed it as up to 1693 single decode steps, ~0.46 s per frame for a given map).
// Positions 0 .. P - 2 of every scene are the rows of the TAR stacks' own kernels -- LayerNorm, q|k and V^T GEMMs, S x S attention with the
// causal mask, projection + MLP with the residual epilogues -- in the stacks' workspaces (idle while the decode runs); every layer leaves its
// k | v rows in the decode cache.  Position P - 1 stays a decode step: its input goes to xdec and the step loop starts there.
// Arithmetic: the stacks' contract (16-bit GEMM operands in the 16-bit modes, exact fp32 chains in fp32 mode) instead of the decode
// step's fp32 activations -- the reference computes the prefix in one fp16-autocast pass as well.
template <typename T>
int run_prefix_prefill(umgen_engine* e, int B, int P) {
    const int E = e->E, H = e->H, S = P - 1;
    if (S < 1 || S > e->S_pad) return e->fail(UMGEN_E_INVALID, "prefix pass over %d positions", S);
    hipStream_t st = e->stream;
    const long R = (long)B * S;
    T* A = reinterpret_cast<T*>(e->A);
    T* QKV = reinterpret_cast<T*>(e->QKV);
    T* Hb = reinterpret_cast<T*>(e->Hb);
    T* vt = reinterpret_cast<T*>(e->VT);
    launch_prefix_rows(st, e->tb, e->tb.tske + (long)e->cfg.task_id * E, e->cond, e->d_tokens, B, P, e->X, e->xdec);
    const size_t wrow = (size_t)E * sizeof(T);
    for (size_t li = 0; li < e->oar.size(); ++li) {
        const SubW& w = e->oar[li];
        const char* Wqkv = reinterpret_cast<const char*>(w.attn.Wqkv);
        launch_layernorm<T>(st, e->X, E, R, E, w.ln_a, A);
        linear_store<T>(e, Wqkv, w.attn.bqkv, 2 * E, E, A, R, QKV, 2L * E, 0);           // q | k rows
        GemmArgs g{};                                                                     // V^T per (scene, head): [B][H][48][S_pad]
        g.P = A; g.Q = Wqkv + (size_t)2 * E * wrow;
        g.Mi = S; g.Nj = E; g.K = E; g.ldp = E; g.ldq = E; g.strideP = (long)S * E; g.strideQ = 0; g.batch = B;
        g.mode = GEMM_VT; g.bias = w.attn.bqkv + 2 * E; g.out = vt; g.ldo = e->S_pad; g.H = H;
        gemm_timed<T>(e, g);
        launch_prefix_kv_to_cache<T>(st, QKV, vt, B, S, e->S_pad, H, e->Lmax, reinterpret_cast<T*>(e->kvcache) + (long)li * e->kv_layer_stride,
                                     e->kv_scene_stride);
        Path<T>::attn_causal(st, QKV, vt, A, B, S, e->S_pad, H);
        linear_resid<T>(e, w.attn.Wo, w.attn.bo, E, E, A, R, e->X);
        launch_layernorm<T>(st, e->X, E, R, E, w.ln_b, A);
        linear_store<T>(e, w.mlp.Wfc, nullptr, 4 * E, E, A, R, Hb, 4L * E, 1);
        linear_resid<T>(e, w.mlp.Wproj, nullptr, E, 4 * E, Hb, R, e->X);
    }
    return 0;
}

// Slot caches of the temporal sub-blocks: per stack and block [max_batch][max_cond_frames][S_stack][2E] of T (10.5 GB per scene for
// UMGen_Large in 16 bits -- what the 288 GB are for).  Allocated once, at create for the overlapped pass or on the first frame whose
// window will grow; when they would take more than half of the free memory the engine keeps recomputing the window (tcache_state -1).
bool ensure_tcache(umgen_engine* e) {
    if (e->tcache_state) return e->tcache_state > 0;
    const size_t Bm = e->cfg.max_batch, Tm = e->cfg.max_cond_frames;
    size_t free_b = 0, total_b = 0, need = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); e->tcache_state = -1; return false; }
    for (int st = 0; st < 4; ++st) need += e->stk[st].size() * Bm * Tm * (size_t)stack_len(st) * 2 * e->E * e->tsz;
    if (need > free_b / 2) { e->tcache_state = -1; return false; }
    for (int st = 0; st < 4; ++st) {
        e->tcache[st].assign(e->stk[st].size(), nullptr);
        for (auto& c : e->tcache[st])
            if (dev_alloc(e, &c, Bm * Tm * (size_t)stack_len(st) * 2 * e->E * e->tsz)) {
                // partial failure: the caches allocated so far would stay reserved and unused for the engine's life -- give them back
                (void)hipGetLastError();
                for (int s2 = 0; s2 <= st; ++s2) {
                    for (void*& p : e->tcache[s2])
                        if (p) {
                            e->allocs.erase(std::remove(e->allocs.begin(), e->allocs.end(), p), e->allocs.end());
                            (void)hipFree(p);
                            p = nullptr;
                        }
                    e->tcache[s2].clear();
                }
                e->tcache_state = -1;
                return false;
            }
    }
    e->tcache_state = 1;
    return true;
}

struct FrameIO {
    int B, T;
    const int *pose, *map, *box, *img;          // host window tokens [B][T][S_mod] (box already control-overwritten)
    const int* ctrl_pose;                        // host [B][3] or nullptr: pose given (init_tokens["pose"])
    const unsigned char* control_slot;           // host [B][60] or nullptr
    int frame_idx;
    const umgen_sampling* smp;
    const umgen_trace* trace;                    // B == 1 only
    int* out_tokens;                             // host [B][2199]
    int cond_cap = 0;                            // window cap (cond_frames) of the rollout; 0 = single frame, nothing follows
    bool next_follows = false;                   // another frame of the same rollout follows: run its prefix pass beside the decode
    bool next_has_ctrl_pose = false;             // ... and its pose is given, so the ego net's prefix is not needed
    const int* given_map = nullptr;              // host [B][1024] or nullptr: the new frame's map is given (predefined-token prefix)
    const int* given_box = nullptr;              // host [B][660] or nullptr: ... and its boxes (only behind a given map)
};

// Does the prefix pass that ran beside the previous frame's decode cover exactly this window's slots 0..P-1 ?
bool prefix_matches(const umgen_engine* e, const FrameIO& io) {
    const umgen_engine::Prefix& px = e->px;
    if (!px.valid || io.trace || io.B != px.B || io.T != px.Tfull || px.P != io.T - 1 || px.P < 1) return false;
    if (!io.ctrl_pose && !px.has_ego) return false;
    const int S[4] = {kNPose, kNMap, kNBox, kNImg};
    const int* cur[4] = {io.pose, io.map, io.box, io.img};
    const std::vector<int>* old[4] = {&px.pose, &px.map, &px.box, &px.img};
    for (int m = 0; m < 4; ++m)
        for (int b = 0; b < io.B; ++b)
            if (memcmp(cur[m] + (size_t)b * io.T * S[m], old[m]->data() + (size_t)b * px.P * S[m], (size_t)px.P * S[m] * sizeof(int))) return false;
    for (int b = 0; b < io.B; ++b)   // the last frame's pose tokens were already baked into shifted slot P-1
        if (memcmp(io.pose + ((size_t)b * io.T + px.P) * kNPose, &px.pose_next[(size_t)b * 3], 3 * sizeof(int))) return false;
    return true;
}

void decode_pose_shift(const int* pose, const int* ego, int B, int Tn, std::vector<int>& pshift, std::vector<float>& pdiff) {
    // pose shifted one frame ahead (UMGen.py:1445-1452) and its decoded (dx, dy, dtheta) for the map warp
    pshift.resize((size_t)B * Tn * 3);
    pdiff.resize((size_t)B * Tn * 3);
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < Tn; ++t)
            for (int a = 0; a < 3; ++a) {
                const int v = (t + 1 < Tn) ? pose[((size_t)b * Tn + t + 1) * 3 + a] : ego[b * 3 + a];
                pshift[((size_t)b * Tn + t) * 3 + a] = v;
                pdiff[((size_t)b * Tn + t) * 3 + a] = decode_pose_value(v, a);
            }
}

// What a decode step reads and writes per scene, as the engine's members: a decode lane swaps in the view of its sub-batch (scenes
// b0 .. b0 + nb - 1 of every per-scene array, its own stream / step state / fragment buffers) around enqueue_step.
struct DecView {
    hipStream_t stream;
    float *xdec, *qdec, *logits, *logits_tar, *cond, *xfrag, *afrag, *hfrag;
    void* kvcache;
    int *d_tokens, *d_prev_box, *d_nboxes;
    unsigned char* d_control;
    double* d_boxes;
    unsigned long long* d_seeds;
    OarState* d_state;
};
DecView current_view(const umgen_engine* e) {
    return DecView{e->stream, e->xdec, e->qdec, e->logits, e->logits_tar, e->cond, e->xfrag, e->afrag, e->hfrag, e->kvcache,
                   e->d_tokens, e->d_prev_box, e->d_nboxes, e->d_control, e->d_boxes, e->d_seeds, e->d_state};
}
void apply_view(umgen_engine* e, const DecView& v) {
    e->stream = v.stream; e->xdec = v.xdec; e->qdec = v.qdec; e->logits = v.logits; e->logits_tar = v.logits_tar; e->cond = v.cond;
    e->xfrag = v.xfrag; e->afrag = v.afrag; e->hfrag = v.hfrag; e->kvcache = v.kvcache; e->d_tokens = v.d_tokens; e->d_prev_box = v.d_prev_box;
    e->d_nboxes = v.d_nboxes; e->d_control = v.d_control; e->d_boxes = v.d_boxes; e->d_seeds = v.d_seeds; e->d_state = v.d_state;
}
DecView lane_view(const umgen_engine* e, const DecView& all, const umgen_engine::DecLane& ln, int b0) {
    const long E = e->E;
    DecView v = all;
    v.stream = ln.s; v.d_state = ln.st; v.xfrag = ln.xfrag; v.afrag = ln.afrag; v.hfrag = ln.hfrag;
    v.xdec = all.xdec + b0 * E; v.qdec = all.qdec + b0 * E; v.logits = all.logits + (long)b0 * 8192;
    v.logits_tar = all.logits_tar + (long)b0 * kNBox * e->cfg.bbox3d_vocab; v.cond = all.cond + (long)b0 * kSeq * E;
    v.kvcache = static_cast<unsigned char*>(all.kvcache) + (size_t)b0 * e->kv_scene_stride * e->tsz;
    v.d_tokens = all.d_tokens + (long)b0 * kTokPerFrame; v.d_prev_box = all.d_prev_box + (long)b0 * kNBox; v.d_nboxes = all.d_nboxes + b0;
    v.d_control = all.d_control + (long)b0 * kSlots; v.d_boxes = all.d_boxes + (long)b0 * 64 * 10; v.d_seeds = all.d_seeds + b0;
    return v;
}

// kernels of one decode step of kind mod (0 fixed token, 1 map, 2 bbox3d, 3 image) for B scenes
template <typename T>
int enqueue_step(umgen_engine* e, int B, int mod, int ns, const umgen_trace* tr, int j) {
    const int E = e->E;
    hipStream_t st = e->stream;
    const bool time_layers = e->profiling && !e->in_capture && !e->in_lanes;
    if (time_layers) {
        if (e->layer_ev_used == e->layer_ev.size()) {
            hipEvent_t a0, a1;
            hipEventCreate(&a0);
            hipEventCreate(&a1);
            e->layer_ev.emplace_back(a0, a1);
        }
        hipEventRecord(e->layer_ev[e->layer_ev_used].first, st);
    }
    if (int rc = oar_layers<T>(e, B, ns)) return rc;
    if (time_layers) hipEventRecord(e->layer_ev[e->layer_ev_used++].second, st);
    static FILE* dump = getenv("UMGEN_DEBUG_DUMP_X") ? fopen(getenv("UMGEN_DEBUG_DUMP_X"), "wb") : nullptr;   // debugging only (eager launches)
    if (dump && tr == nullptr && !e->cfg.use_graphs) {
        std::vector<float> hx((size_t)E);
        hipMemcpyAsync(hx.data(), e->xdec, (size_t)E * 4, hipMemcpyDeviceToHost, st);
        hipStreamSynchronize(st);
        fwrite(hx.data(), 4, E, dump);
        fflush(dump);
    }
    SampleArgs sa{};
    sa.st = e->d_state; sa.tb = e->tb; sa.logits = e->logits; sa.logits_tar = e->logits_tar; sa.ld_logits = 8192; sa.ld_tar = e->cfg.bbox3d_vocab;
    sa.cond = e->cond; sa.x_next = e->xdec; sa.tokens = e->d_tokens; sa.prev_box = e->d_prev_box; sa.control_slot = e->d_control;
    sa.boxes = e->d_boxes; sa.n_boxes = e->d_nboxes; sa.seeds = e->d_seeds; sa.forced = e->d_forced; sa.counters = e->d_counters;
    if (mod == 0) {
        launch_fixed_token(st, sa, B);
    } else {
        const void* head = mod == 1 ? e->head_ar_map : (mod == 2 ? e->head_ar_box : e->head_ar_img);
        const int V = mod == 1 ? e->cfg.map_vocab : (mod == 2 ? e->cfg.bbox3d_vocab : e->cfg.img_vocab);
        bool head_done = false;
        if constexpr (sizeof(T) == 2) {
            if (e->use_batched(B)) {
                RowsArgs r{};
                r.x = e->xfrag; r.M = B; r.ln_w = e->ln_oar; r.W = head; r.N = V; r.K = E; r.mode = ROWS_F32; r.out = e->logits; r.ldo = sa.ld_logits; r.E = E;   // (xfrag: the last layer's copy of x)
                launch_rows_mfma<T>(e->stream, r);
                head_done = true;
            }
        }
        if (!head_done) gemv<T>(e, e->xdec, E, e->ln_oar, head, nullptr, V, E, B, GEMV_OUT_F32, e->logits, sa.ld_logits);
        // (head_tar_bbox3d on the conditioning rows, UMGen.py:1087,1103: the rows do not depend on the decoded tokens, so all 660
        //  positions were multiplied once before the loop -- tar_head_logits -- instead of one GEMV launch per bbox3d step)
        if (tr) {
            float* dst = mod == 1 ? tr->logits_map : (mod == 2 ? tr->logits_bbox3d : tr->logits_image);
            const int k = mod == 1 ? j - kMapC0 : (mod == 2 ? j - kBoxC0 : j - kImgC0);
            if (dst) HIPCHK(e, hipMemcpyAsync(dst + (size_t)k * V, e->logits, (size_t)V * 4, hipMemcpyDeviceToHost, st));
        }
        sa.mod = mod;
        sa.vocab = V;
        launch_sample_token(st, sa, B);
    }
    return 0;
}

// Background pass for the NEXT frame of the rollout: its window is this window moved on by one frame, and all its slots but the
// last are known now (the new frame's pose tokens `ego` included).  They run through the four stacks on bg_stream while the
// decode loop of the current frame owns the other CUs; the temporal k | v rows of every layer land in the slot caches.
template <typename T>
int launch_prefix(umgen_engine* e, const FrameIO& io, const std::vector<int>& ego) {
    const int B = io.B, Tn = io.T;
    const int Tnext = std::min(Tn + 1, io.cond_cap);
    const int off = (Tn + 1 > io.cond_cap) ? 1 : 0;     // the window slides (UMGen.py:1600-1603) or still grows
    const int P = Tnext - 1;
    if (P < 1 || Tnext > e->cfg.max_cond_frames) return 0;
    umgen_engine::Prefix& px = e->px;
    px.valid = false;
    px.B = B; px.P = P; px.Tfull = Tnext; px.has_ego = !io.next_has_ctrl_pose;
    const int S[4] = {kNPose, kNMap, kNBox, kNImg};
    const int* cur[4] = {io.pose, io.map, io.box, io.img};
    std::vector<int>* keep[4] = {&px.pose, &px.map, &px.box, &px.img};
    std::vector<int>* up[4] = {&e->px_up[0], &e->px_up[1], &e->px_up[2], &e->px_up[3]};
    for (int m = 0; m < 4; ++m) {
        keep[m]->assign((size_t)B * P * S[m], 0);
        up[m]->assign((size_t)B * Tnext * S[m], 0);
        for (int b = 0; b < B; ++b) {
            const int* src = cur[m] + ((size_t)b * Tn + off) * S[m];
            memcpy(keep[m]->data() + (size_t)b * P * S[m], src, (size_t)P * S[m] * sizeof(int));
            memcpy(up[m]->data() + (size_t)b * Tnext * S[m], src, (size_t)P * S[m] * sizeof(int));
        }
    }
    px.pose_next.assign(ego.begin(), ego.end());
    for (int b = 0; b < B; ++b)
        for (int a = 0; a < 3; ++a) (*up[0])[((size_t)b * Tnext + P) * 3 + a] = ego[b * 3 + a];
    std::vector<int> zero((size_t)B * 3, 0);
    decode_pose_shift(up[0]->data(), zero.data(), B, Tnext, e->px_pshift, e->px_pdiff);   // slot P (unknown) is not touched by this pass

    hipStream_t fg = e->stream, bg = e->bg_engine ? e->stream : e->bg_stream;
    if (!e->bg_engine) {
        HIPCHK(e, hipEventRecord(e->ev_tar_done, fg));
        HIPCHK(e, hipStreamWaitEvent(bg, e->ev_tar_done, 0));
    }
    HIPCHK(e, hipMemcpyAsync(e->d_pose, up[0]->data(), up[0]->size() * 4, hipMemcpyHostToDevice, bg));
    HIPCHK(e, hipMemcpyAsync(e->d_map, up[1]->data(), up[1]->size() * 4, hipMemcpyHostToDevice, bg));
    HIPCHK(e, hipMemcpyAsync(e->d_box, up[2]->data(), up[2]->size() * 4, hipMemcpyHostToDevice, bg));
    HIPCHK(e, hipMemcpyAsync(e->d_img, up[3]->data(), up[3]->size() * 4, hipMemcpyHostToDevice, bg));
    HIPCHK(e, hipMemcpyAsync(e->d_pose_shift, e->px_pshift.data(), e->px_pshift.size() * 4, hipMemcpyHostToDevice, bg));
    HIPCHK(e, hipMemcpyAsync(e->pose_diff, e->px_pdiff.data(), e->px_pdiff.size() * 4, hipMemcpyHostToDevice, bg));
    const WindowTokens ws{e->d_pose_shift, e->d_map, e->d_box, e->d_img, B, P, Tnext, 0};
    if (e->bg_engine) {
        // the pass as an op list for the decode engine's background workers: the same run_stack calls, with the launchers recording instead of launching
        BgRecorder rec;
        g_bg_rec = &rec;
        if (px.has_ego) run_stack<T>(e, STACK_EGO, WindowTokens{e->d_pose, e->d_map, e->d_box, e->d_img, B, P, Tnext, 0}, 1);
        run_stack<T>(e, STACK_MAP, ws, 1);
        run_stack<T>(e, STACK_BOX, ws, 1);
        run_stack<T>(e, STACK_TAR, ws, 1);
        g_bg_rec = nullptr;
        if (rec.failed || rec.ops.empty() || rec.ops.size() > (size_t)kBgMaxOps) {
            if (getenv("UMGEN_DEBUG_TIMING")) fprintf(stderr, "[umgen] background pass not recordable (%s): this frame's successor computes its whole window\n", rec.failed ? rec.failed : "op count");
            return 0;      // (px.valid stays false: the next frame takes the plain path)
        }
        // unit times: the workers' own measurements survive from pass to pass while the list keeps its shape; a new shape starts from the host's guesses
        bool same = rec.ops.size() == e->bg_rec.ops.size();
        for (size_t i = 0; same && i < rec.ops.size(); ++i) same = memcmp(&rec.ops[i].h, &e->bg_rec.ops[i].h, sizeof(BgOpHead)) == 0;
        e->bg_rec = std::move(rec);
        // header: op count, engine_ticks = 0 (the first launch measures, the workers start with the second), margin 10 us, the embedding tables
        e->bg_head.w[0] = (unsigned)e->bg_rec.ops.size(); e->bg_head.w[1] = 0u; e->bg_head.w[2] = (unsigned)(getenv("UMGEN_BG_MARGIN_US") ? atoi(getenv("UMGEN_BG_MARGIN_US")) * 100 : 1000); e->bg_head.w[3] = 0u;
        e->bg_head.tb = e->tb;
        static_assert(offsetof(BgQueue, tb) == 16 && offsetof(BgQueue, state) == 16 + sizeof(EmbedTables), "BgQueue header layout");
        HIPCHK(e, hipMemcpyAsync(&e->d_bgq->n_ops, &e->bg_head, sizeof(e->bg_head), hipMemcpyHostToDevice, fg));
        HIPCHK(e, hipMemsetAsync(&e->d_bgq->state[0][0], 0, sizeof(e->d_bgq->state) + sizeof(e->d_bgq->arrive), fg));
        if (!same) HIPCHK(e, hipMemcpyAsync(&e->d_bgq->est[0], e->bg_rec.est.data(), e->bg_rec.est.size() * sizeof(unsigned), hipMemcpyHostToDevice, fg));
        HIPCHK(e, hipMemcpyAsync(&e->d_bgq->ops[0], e->bg_rec.ops.data(), e->bg_rec.ops.size() * sizeof(BgOp), hipMemcpyHostToDevice, fg));
        e->bg_pending = true;      // (px.valid: once the drain behind the frame's last step has seen every worker at the end of the list, run_frame)
        return 0;
    }
    HIPCHK(e, hipEventRecord(e->ev_bg0, bg));
    e->stream = bg;
    if (px.has_ego) run_stack<T>(e, STACK_EGO, WindowTokens{e->d_pose, e->d_map, e->d_box, e->d_img, B, P, Tnext, 0}, 1);
    run_stack<T>(e, STACK_MAP, ws, 1);
    run_stack<T>(e, STACK_BOX, ws, 1);
    run_stack<T>(e, STACK_TAR, ws, 1);
    e->stream = fg;
    HIPCHK(e, hipEventRecord(e->ev_bg_done, bg));
    e->bg_pending = true;
    px.valid = true;
    return 0;
}

// UMGen._inference (UMGen.py:1406-1540) for B scenes
template <typename T>
int run_frame(umgen_engine* e, const FrameIO& io) {
    const int E = e->E, B = io.B, Tn = io.T;
    hipStream_t const fg = e->stream;   // the decode stream of overlapped rollouts (6 of the 8 XCDs when the overlap exists)
    e->launch_status = hipSuccess;
    (void)hipGetLastError();            // a stale error another HIP user of this thread left behind (torch, RCCL) is not this frame's
    struct RestoreStream { umgen_engine* e; hipStream_t s; ~RestoreStream() { e->stream = s; e->set_work(e->w_main); } } restore{e, fg};
    hipStream_t st = fg;
    SamplerParams sp{io.smp->method, io.smp->top_k, io.smp->top_k_map, io.smp->topk_image, io.smp->p, io.smp->p_map, io.smp->temperature,
                     io.smp->rule_constrain, io.smp->merge_ar_tar, io.smp->only_ar};
    const umgen_trace* tr = io.trace;
    const bool forced = tr && tr->forced_map;
    if (e->bg_pending && e->bg_engine) {
        // a pass whose frame failed before its drain: whatever the workers have done of it is worthless -- empty the queue, the slot caches are not to be trusted
        HIPCHK(e, hipMemsetAsync(&e->d_bgq->n_ops, 0, 4, fg));
        e->bg_pending = false;
        e->px.valid = false;
    }
    if (e->bg_pending && !e->bg_engine) {   // the background pass reads the token arrays and owns the TAR scratch buffers until it is done
        const auto tw0 = std::chrono::steady_clock::now();
        HIPCHK(e, hipEventSynchronize(e->ev_bg_done));
        const double waited = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count();
        // the pass only pays while it hides behind the decode loop: if the foreground had to wait for it (many scenes per GPU,
        // a much wider model), later frames go back to the plain one-pass path
        if (waited > 25.0 && e->overlap_mode != 2) e->overlap_suspended = true;
        if (getenv("UMGEN_DEBUG_TIMING")) fprintf(stderr, "[umgen] waited %.1f ms for the background pass\n", waited);
        float bms = 0.f;
        hipEventElapsedTime(&bms, e->ev_bg0, e->ev_bg_done);
        e->tm.bg_ms += bms;
        e->bg_pending = false;
    }
    // overlapped TAR pass: when the previous frame's background pass covered slots 0..Tn-2 of this very window, only the last slot
    // is pushed through the stacks now (against the per-layer slot caches)
    const bool use_px = prefix_matches(e, io);
    // the background pass runs on a quarter of the CUs (measured 3.1x the whole-window time of the same stacks on all CUs): only
    // worth launching when that is expected to hide behind the decode loop (numbers of the previous frames of this engine)
    if (e->last_B != B) { e->last_B = B; e->last_full_pre_ms = 0.f; }
    // (On the decode engine's idle XCDs the pass of a one-scene frame takes ~1800 of its 2206 decode steps, 3.3 x its foreground time (profiles/r06_bg_worker_trace.txt), and what
    //  the steps leave of it drains at 2.7 x: starting it pays as long as about two thirds of it hide -- 2.3 x the foreground time inside the decode loop.  A tighter rule (3.4 x, 5 %
    //  slack) switched the pass OFF on a box whose stacks ran 12 % slower: fp16 1737 instead of 2041 scene-tokens/s.)
    const bool hides = e->overlap_mode == 2 || e->last_full_pre_ms <= 0.f || (e->bg_engine ? 2.3f * e->last_full_pre_ms < e->last_oar_ms : 3.3f * e->last_full_pre_ms < 0.95f * e->last_oar_ms);
    const bool ov_active = e->overlap && !e->overlap_suspended && hides && !e->profiling && !tr && (B == 1 || e->overlap_mode == 2);
    // the ego / TAR phase runs on all CUs; the decode loop leaves the background stream's XCDs alone only when a pass can follow
    hipStream_t const pre = e->full_stream ? e->full_stream : fg;   // (the background stream is idle until this phase is over)
    hipStream_t const dec = (e->full_stream && !ov_active) ? e->full_stream : fg;
    st = pre;
    e->stream = pre;
    HIPCHK(e, hipMemcpyAsync(e->d_pose, io.pose, (size_t)B * Tn * 3 * 4, hipMemcpyHostToDevice, st));
    HIPCHK(e, hipMemcpyAsync(e->d_map, io.map, (size_t)B * Tn * kNMap * 4, hipMemcpyHostToDevice, st));
    HIPCHK(e, hipMemcpyAsync(e->d_box, io.box, (size_t)B * Tn * kNBox * 4, hipMemcpyHostToDevice, st));
    HIPCHK(e, hipMemcpyAsync(e->d_img, io.img, (size_t)B * Tn * kNImg * 4, hipMemcpyHostToDevice, st));
    std::vector<unsigned long long> seeds(B);
    for (int b = 0; b < B; ++b) seeds[b] = io.smp->seeds ? io.smp->seeds[b] : 0ull;
    HIPCHK(e, hipMemcpyAsync(e->d_seeds, seeds.data(), (size_t)B * 8, hipMemcpyHostToDevice, st));
    std::vector<int> forced_host;
    if (forced) {
        forced_host.resize(kTokPerFrame);
        for (int i = 0; i < kNPose; ++i) forced_host[i] = (int)tr->forced_pose[i];
        for (int i = 0; i < kNMap; ++i) forced_host[kOffMap + i] = (int)tr->forced_map[i];
        for (int i = 0; i < kNBox; ++i) forced_host[kOffBox + i] = (int)tr->forced_bbox3d[i];
        for (int i = 0; i < kNImg; ++i) forced_host[kOffImg + i] = (int)tr->forced_image[i];
        HIPCHK(e, hipMemcpyAsync(e->d_forced, forced_host.data(), (size_t)kTokPerFrame * 4, hipMemcpyHostToDevice, st));
    }
    HIPCHK(e, hipMemsetAsync(e->d_counters, 0, 8 * sizeof(int), st));
    HIPCHK(e, hipMemsetAsync(e->d_nboxes, 0, (size_t)B * sizeof(int), st));

    // foreground growing-window reuse (f-3): the next frame's window is this one plus one slot (it grows, it does not slide), so this
    // frame's passes leave their temporal k | v rows in the slot caches
    const bool fg_write = !e->overlap && e->grow_cache && io.next_follows && io.cond_cap > 0 && Tn + 1 <= io.cond_cap &&
                          Tn + 1 <= e->cfg.max_cond_frames && !tr && ensure_tcache(e);
    const int t0 = use_px ? Tn - 1 : 0, Tc = use_px ? 1 : Tn, cmode = use_px ? (fg_write ? 4 : 2) : (fg_write ? 3 : 0);
    e->px.valid = false;
    WindowTokens w{e->d_pose, e->d_map, e->d_box, e->d_img, B, Tc, Tn, t0};
    // Step 1: ego pose tokens (UMGen.py:1440-1455)
    std::vector<int> ego(B * 3);
    HIPCHK(e, hipEventRecord(e->ev[0], st));
    if (io.ctrl_pose) {
        for (int i = 0; i < B * 3; ++i) ego[i] = io.ctrl_pose[i];
    } else {
        run_ego<T>(e, w, sp, io.frame_idx, forced, tr ? tr->ego_logits : nullptr, cmode);
        HIPCHK(e, hipMemcpyAsync(ego.data(), e->d_ego_tok, (size_t)B * 3 * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(e, hipStreamSynchronize(st));
    }
    HIPCHK(e, hipEventRecord(e->ev[1], st));
    std::vector<int> pshift;
    std::vector<float> pdiff;
    decode_pose_shift(io.pose, ego.data(), B, Tn, pshift, pdiff);
    HIPCHK(e, hipMemcpyAsync(e->d_pose_shift, pshift.data(), pshift.size() * 4, hipMemcpyHostToDevice, st));
    HIPCHK(e, hipMemcpyAsync(e->pose_diff, pdiff.data(), pdiff.size() * 4, hipMemcpyHostToDevice, st));
    // new-frame token buffer: pose = ego tokens; previous frame's bbox3d tokens; control mask
    std::vector<int> tok0((size_t)B * kTokPerFrame, 0);
    std::vector<int> prevbox((size_t)B * kNBox);
    for (int b = 0; b < B; ++b) {
        for (int a = 0; a < 3; ++a) tok0[(size_t)b * kTokPerFrame + a] = ego[b * 3 + a];
        memcpy(&prevbox[(size_t)b * kNBox], io.box + ((size_t)b * Tn + (Tn - 1)) * kNBox, kNBox * sizeof(int));
        if (io.given_map) memcpy(&tok0[(size_t)b * kTokPerFrame + kOffMap], io.given_map + (size_t)b * kNMap, kNMap * sizeof(int));
        if (io.given_box) memcpy(&tok0[(size_t)b * kTokPerFrame + kOffBox], io.given_box + (size_t)b * kNBox, kNBox * sizeof(int));
    }
    // scene positions whose tokens are GIVEN (UMGen.py:1184-1201): the pose prefix always; the map, or the map and the boxes, when the
    // caller provides them.  Their steps replay the token (fixed_token_kernel), the sampled steps start behind them.
    const int given_end = io.given_box ? kBoxEos + 1 : (io.given_map ? kMapEos + 1 : kPoseEos + 1);
    HIPCHK(e, hipMemcpyAsync(e->d_tokens, tok0.data(), tok0.size() * 4, hipMemcpyHostToDevice, st));
    HIPCHK(e, hipMemcpyAsync(e->d_prev_box, prevbox.data(), prevbox.size() * 4, hipMemcpyHostToDevice, st));
    if (io.control_slot) HIPCHK(e, hipMemcpyAsync(e->d_control, io.control_slot, (size_t)B * kSlots, hipMemcpyHostToDevice, st));
    // measurement knob (never set in production): UMGEN_DEBUG_OAR_STEPS="a:b" runs only decode steps j in [a, b) so that
    // per-dispatch PMC collection (which serialises every kernel) stays bounded; results are meaningless then.
    int j_begin = 0, j_end = kImgEos;
    if (const char* dbg = getenv("UMGEN_DEBUG_OAR_STEPS")) {
        int a0 = 0, b0 = kImgEos;
        if (sscanf(dbg, "%d:%d", &a0, &b0) == 2 && a0 >= 0 && b0 <= kImgEos && a0 < b0) { j_begin = a0; j_end = b0; }
    }
    // Hand-off tags of the decode engine never repeat while a copy of an old granule can survive anywhere (a group's L2 keeps its
    // plain-stored granules across launches): the epoch runs on monotonically over the engine's lifetime.  Before the 32-bit
    // counter would wrap (~100 frames at 16384 tags per step) everything is drained and the granule buffers are cleared.
    if (e->wide_enabled && e->eng_epoch > 0xE0000000u) {
        HIPCHK(e, hipDeviceSynchronize());
        HIPCHK(e, hipMemset(e->wide_gran, 0, oar_engine_wide_granules() * 8));
        e->eng_epoch = 16u;
    }
    if (e->eng_enabled && e->eng_epoch > 0xE0000000u) {
        HIPCHK(e, hipDeviceSynchronize());
        HIPCHK(e, hipMemset(e->eng_gx, 0, (size_t)e->cfg.max_batch * kEngE * 8));
        HIPCHK(e, hipMemset(e->eng_gloc, 0, e->eng_gloc_bytes));
        e->eng_epoch = 16u;
    }
    OarState s0{j_begin, io.frame_idx, forced ? 1 : 0, io.control_slot ? 1 : 0, 0, e->eng_epoch, sp};
    e->eng_epoch += (unsigned)(kImgEos + 1) * kEpochPerStep;
    HIPCHK(e, hipMemcpyAsync(e->d_state, &s0, sizeof(s0), hipMemcpyHostToDevice, st));
    const int n_lanes = (sizeof(T) == 2 && !tr) ? e->lane_count(B) : 1;      // decode lanes: every lane steps its own copy of the state
    for (int l = 0; l < n_lanes && n_lanes > 1; ++l) HIPCHK(e, hipMemcpyAsync(e->lane[l].st, &s0, sizeof(s0), hipMemcpyHostToDevice, st));

    // Step 2: the three TAR stacks (UMGen.py:1484-1494) and the conditioning rows (1496-1511)
    WindowTokens ws{e->d_pose_shift, e->d_map, e->d_box, e->d_img, B, Tc, Tn, t0};
    // (a profiled frame keeps the stacks one behind the other: its per-launch events are meant to time each kernel alone)
    if ((use_px || (e->conc_stacks && !e->profiling)) && e->side_stream[0]) {
        // the three stacks are independent -- map and box run on side streams (own workspaces: 1 slot for the last-slot passes of the
        // overlapped path, where 2207 rows per scene leave most CUs idle; the whole window otherwise) beside the TAR stack
        struct Restore { umgen_engine* e; hipStream_t s; ~Restore() { e->stream = s; e->set_work(e->w_main); } } restore{e, st};
        HIPCHK(e, hipEventRecord(e->ev_side_in, st));
        const int side_stack[2] = {STACK_MAP, STACK_BOX};
        for (int i = 0; i < 2; ++i) {
            HIPCHK(e, hipStreamWaitEvent(e->side_stream[i], e->ev_side_in, 0));
            e->stream = e->side_stream[i];
            e->set_work(e->w_side[i]);
            run_stack<T>(e, side_stack[i], ws, cmode);
            launch_cond_rows(e->stream, side_stack[i], B, Tc, E, e->X, i == 0 ? e->ln_map_tar : e->ln_box_tar, i == 0 ? e->warped_last : nullptr,
                             e->cond);
            HIPCHK(e, hipEventRecord(e->ev_side_done[i], e->side_stream[i]));
        }
        e->stream = st;
        e->set_work(e->w_main);
        run_stack<T>(e, STACK_TAR, ws, cmode);
        launch_cond_rows(st, STACK_TAR, B, Tc, E, e->X, e->ln_tar, nullptr, e->cond);
        for (int i = 0; i < 2; ++i) HIPCHK(e, hipStreamWaitEvent(st, e->ev_side_done[i], 0));
    } else {
        run_stack<T>(e, STACK_MAP, ws, cmode);
        launch_cond_rows(st, STACK_MAP, B, Tc, E, e->X, e->ln_map_tar, e->warped_last, e->cond);
        run_stack<T>(e, STACK_BOX, ws, cmode);
        launch_cond_rows(st, STACK_BOX, B, Tc, E, e->X, e->ln_box_tar, nullptr, e->cond);
        run_stack<T>(e, STACK_TAR, ws, cmode);
        launch_cond_rows(st, STACK_TAR, B, Tc, E, e->X, e->ln_tar, nullptr, e->cond);
    }
    if (tr && tr->cond) HIPCHK(e, hipMemcpyAsync(tr->cond, e->cond, (size_t)kSeq * E * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(e, hipEventRecord(e->ev[2], st));
    if (fg_write) {   // the slot caches now hold slots 0 .. Tn-1 of this window: what the next frame must find unchanged (prefix_matches)
        umgen_engine::Prefix& px = e->px;
        px.B = B; px.P = Tn; px.Tfull = Tn + 1; px.has_ego = !io.ctrl_pose;
        px.pose.assign(io.pose, io.pose + (size_t)B * Tn * kNPose);
        px.map.assign(io.map, io.map + (size_t)B * Tn * kNMap);
        px.box.assign(io.box, io.box + (size_t)B * Tn * kNBox);
        px.img.assign(io.img, io.img + (size_t)B * Tn * kNImg);
        px.pose_next.assign(ego.begin(), ego.end());   // slot Tn-1 was evaluated with the new frame's pose (the shift of UMGen.py:1445-1452)
        px.valid = true;
    }

    // Step 3: OAR decode loop (infer_oar_net, UMGen.py:1151-1273).  Step j consumes scene position j (KV length j) and
    // emits scene token j; j = 0..4 replays the given pose prefix, bos/eos are fixed, everything else is sampled.
    // A step is a fixed kernel sequence with fixed arguments (all per-step state is device resident), replayed from a
    // hipGraph per step kind; trace mode launches directly so the logits can be copied out between kernels.
    if (dec != pre) {
        HIPCHK(e, hipEventRecord(e->ev_pre_done, pre));
        HIPCHK(e, hipStreamWaitEvent(dec, e->ev_pre_done, 0));
    }
    st = dec;
    e->stream = dec;
    launch_first_input(st, B, E, e->tb.tske + (long)e->cfg.task_id * E, e->cond, e->xdec);
    {   // tar_head_logits: logits_tar[b][k][:] = head_tar_bbox3d . cond[b][kBoxC0 + k][:]  (exact fp32 FMA chain, bf16 or fp32 weights)
        GemmArgs g{};
        g.P = e->head_tar_box; g.Q = e->cond + (long)kBoxC0 * E; g.Mi = e->cfg.bbox3d_vocab; g.Nj = kNBox; g.K = E; g.ldp = E; g.ldq = E;
        g.strideP = 0; g.strideQ = (long)kSeq * E; g.batch = B; g.mode = GEMM_STORE_F32; g.out = e->logits_tar; g.ldo = e->cfg.bbox3d_vocab;
        g.strideO = (long)kNBox * e->cfg.bbox3d_vocab;
        launch_gemm_valu<T, float>(st, g);
    }
    if (ov_active && io.next_follows) {
        const auto tp0 = std::chrono::steady_clock::now();
        if (int rc = launch_prefix<T>(e, io, ego)) return rc;
        if (getenv("UMGEN_DEBUG_TIMING"))
            fprintf(stderr, "[umgen] host time to enqueue the background pass: %.1f ms\n",
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp0).count());
    }
    // given map (/ boxes): positions 0 .. given_end - 2 as one pass, the step loop starts at given_end - 1 (UMGEN_PREFIX_PASS=0: replay every
    // given position as a decode step, rounds 1-4).  WHICH form runs is a property of the engine, never of the frame: the pass works in the stacks'
    // buffers, so an engine created with the overlapped background TAR pass ON A SECOND STREAM (UMGEN_OVERLAP=1) replays on every frame, and every
    // other engine -- the decode engine's background workers included: their pass starts with the first decode step, behind this one -- takes the
    // pass on every frame -- traced or not, followed by another frame or not -- so that rollout(n) stays a token prefix of rollout(n + 1) in the
    // 16-bit modes, whose two forms differ in arithmetic.
    const char* ppe = getenv("UMGEN_PREFIX_PASS");      // (read per frame: the tests compare both forms in one process)
    const bool prefix_pass_off = ppe && ppe[0] == '0';
    OarState s1 = s0;                                   // (function scope: the asynchronous uploads below read it until the stream is drained)
    if (given_end > kPoseEos + 1 && !prefix_pass_off && (!e->overlap || e->bg_engine) && j_begin == 0) {
        if (int rc = run_prefix_prefill<T>(e, B, given_end)) return rc;
        j_begin = given_end - 1;
        s1.step = j_begin;
        HIPCHK(e, hipMemcpyAsync(e->d_state, &s1, sizeof(s1), hipMemcpyHostToDevice, st));
        for (int l = 0; l < n_lanes && n_lanes > 1; ++l) HIPCHK(e, hipMemcpyAsync(e->lane[l].st, &s1, sizeof(s1), hipMemcpyHostToDevice, st));
        e->tm.prefix_passes += 1;
    }
    const int steps_run = j_end - j_begin;             // decode steps of this frame (a prefix pass replaces the given positions' steps)
    const bool graphs = e->cfg.use_graphs && !tr && !e->profiling;   // profiled frames time every decode step's layer kernel(s) with events
    const bool batched = sizeof(T) == 2 && e->use_batched(B);       // the batched decode layer takes the step (oar_layers)
    const bool wide = sizeof(T) == 2 && e->use_wide(B);           // the chip-wide engine of the wide layers: one launch per scene and step
    static const umgen_engine::EngStream wide_stream{true, 8, {}};
    const umgen_engine::EngStream* eng = wide ? &wide_stream : ((sizeof(T) == 2 && !batched) ? e->eng_for(st) : nullptr);
    const int eng_ng = wide ? -3 : (eng ? eng->NG + (e->bg_engine && B == 1 ? 100 : 0) : (batched ? -2 : 0));     // the engine's grid depends on the stream's XCDs: graphs are per (B, NG)
    if (graphs && (e->step_graph_B != B || e->step_graph_NG != eng_ng)) {
        for (auto& row : e->step_graph)
            for (auto& g : row)
                if (g) { hipGraphExecDestroy(g); g = nullptr; }
        e->step_graph_B = B;
        e->step_graph_NG = eng_ng;
    }
    if (n_lanes > 1) {
        // Decode lanes (umgen_engine::DecLane): fork behind the first input, every lane replays the step runs for its scenes on its own
        // stream, join before the token download.  Step runs are graphs in every mode but --no-graphs (a profiled frame times lane 0's
        // runs around the graph launches: eager launches of n lanes x 182 kernels per step would measure the host).
        auto kind_of = [given_end](int jj) {
            if (jj < given_end) return 0;
            return (jj >= kMapC0 && jj < kMapEos) ? 1 : (jj >= kBoxC0 && jj < kBoxEos) ? 2 : (jj >= kImgC0 && jj < kImgEos) ? 3 : 0;
        };
        const bool lane_graphs = e->cfg.use_graphs;
        if (e->lane_graph_B != B || e->lane_graph_n != n_lanes) {
            for (auto& ln : e->lane)
                for (auto& row : ln.graph)
                    for (auto& g : row)
                        if (g) { hipGraphExecDestroy(g); g = nullptr; }
            e->lane_graph_B = B;
            e->lane_graph_n = n_lanes;
        }
        const DecView all = current_view(e);
        struct RestoreView { umgen_engine* e; DecView v; ~RestoreView() { apply_view(e, v); e->in_lanes = false; e->in_capture = false; } } restore_view{e, all};
        e->in_lanes = true;
        HIPCHK(e, hipEventRecord(e->ev_lane_fork, st));
        int b0s[umgen_engine::kMaxLanes], nbs[umgen_engine::kMaxLanes];
        for (int l = 0, b0 = 0; l < n_lanes; ++l) {
            nbs[l] = B / n_lanes + (l < B % n_lanes ? 1 : 0);
            b0s[l] = b0;
            b0 += nbs[l];
            HIPCHK(e, hipStreamWaitEvent(e->lane[l].s, e->ev_lane_fork, 0));
        }
        for (int j = j_begin; j < j_end; ++j) {
            const int mod = kind_of(j);
            int same = 1;
            while (same < 16 && j + same < j_end && kind_of(j + same) == mod) ++same;
            const int run = !lane_graphs ? 1 : (same >= 16 ? 16 : (same >= 4 ? 4 : 1));
            const int ri = run == 16 ? 2 : (run == 4 ? 1 : 0);
            for (int l = 0; l < n_lanes; ++l) {
                umgen_engine::DecLane& ln = e->lane[l];
                apply_view(e, lane_view(e, all, ln, b0s[l]));
                const bool timed = e->profiling && l == 0;
                if (timed) {
                    if (e->layer_ev_used == e->layer_ev.size()) {
                        hipEvent_t a0, a1;
                        hipEventCreate(&a0);
                        hipEventCreate(&a1);
                        e->layer_ev.emplace_back(a0, a1);
                    }
                    hipEventRecord(e->layer_ev[e->layer_ev_used].first, ln.s);
                }
                if (lane_graphs) {
                    hipGraphExec_t& ge = ln.graph[mod][ri];
                    if (!ge) {
                        hipGraph_t g;
                        HIPCHK(e, hipStreamBeginCapture(ln.s, hipStreamCaptureModeThreadLocal));
                        e->in_capture = true;
                        int crc = 0;
                        for (int r = 0; r < run && !crc; ++r) crc = enqueue_step<T>(e, nbs[l], mod, 1, nullptr, 0);
                        e->in_capture = false;
                        if (crc) return crc;   // (run_frame_any ends the open capture)
                        HIPCHK(e, hipStreamEndCapture(ln.s, &g));
                        HIPCHK(e, hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                        HIPCHK(e, hipGraphDestroy(g));
                    }
                    HIPCHK(e, hipGraphLaunch(ge, ln.s));
                } else if (int rc = enqueue_step<T>(e, nbs[l], mod, 1, nullptr, j)) {
                    return rc;
                }
                if (timed) {
                    hipEventRecord(e->layer_ev[e->layer_ev_used++].second, ln.s);
                    e->tm.layers_launches += run - 1;      // (the frame's bookkeeping below adds one per event pair)
                }
                e->tm.oar_kernels += (int64_t)run * (5 * (int64_t)e->oar.size() + (mod == 0 ? 1 : (mod == 2 ? 3 : 2)));
            }
            j += run - 1;
        }
        for (int l = 0; l < n_lanes; ++l) {
            HIPCHK(e, hipEventRecord(e->lane[l].done, e->lane[l].s));
            HIPCHK(e, hipStreamWaitEvent(st, e->lane[l].done, 0));
        }
        j_begin = j_end;     // (the single-stream loop below has nothing left to do)
    }
    for (int j = j_begin; j < j_end; ++j) {   // the img-eos step (j = 2206) produces nothing that is consumed
        auto kind_of = [given_end](int jj) {
            if (jj < given_end) return 0;
            return (jj >= kMapC0 && jj < kMapEos) ? 1 : (jj >= kBoxC0 && jj < kBoxEos) ? 2 : (jj >= kImgC0 && jj < kImgEos) ? 3 : 0;
        };
        const int mod = kind_of(j);
        const int ns = attn_nsplit(j + 1);          // key splits over the j cached keys + the new one
        const int gkey = (eng || batched) ? 0 : ns;   // the engine / the batched layer derive their key geometry from the device-side step
        if (graphs) {
            // With the decode engine a step is 3 kernel nodes and ~600 us, and a graph launch costs ~7 us on the device (the gap between
            // the sampler of one replay and the engine of the next, rocprofv3 kernel trace): runs of steps of the same kind are replayed
            // 16 or 4 at a time (gkey 1 / 2; the engine derives everything else from the device-side step counter).
            int run = 1;
            if (eng || batched) {
                int same = 1;
                while (same < 16 && j + same < j_end) {
                    if (kind_of(j + same) != mod) break;
                    ++same;
                }
                run = same >= 16 ? 16 : (same >= 4 ? 4 : 1);
            }
            hipGraphExec_t& ge = e->step_graph[mod][(eng || batched) ? (run == 16 ? 2 : (run == 4 ? 1 : 0)) : gkey];
            if (!ge) {
                hipGraph_t g;
                HIPCHK(e, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                e->in_capture = true;
                int crc = 0;
                for (int r = 0; r < run && !crc; ++r) crc = enqueue_step<T>(e, B, mod, ns, nullptr, 0);
                e->in_capture = false;
                if (crc) return crc;   // (run_frame_any ends the open capture)
                HIPCHK(e, hipStreamEndCapture(st, &g));
                HIPCHK(e, hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                HIPCHK(e, hipGraphDestroy(g));
            }
            HIPCHK(e, hipGraphLaunch(ge, st));
            e->tm.oar_kernels += (int64_t)(run - 1) * ((eng ? (wide ? B : 1) : 5 * (int64_t)e->oar.size()) + (mod == 0 ? 1 : (mod == 2 ? 3 : 2)));
            j += run - 1;
        } else if (int rc = enqueue_step<T>(e, B, mod, ns, tr, j)) {
            return rc;
        }
        e->tm.oar_kernels += (eng ? (wide ? B : 1) : 5 * (int64_t)e->oar.size()) + (mod == 0 ? 1 : (mod == 2 ? 3 : 2));
    }
    e->tm.oar_steps += steps_run;
    const bool drained = e->bg_engine && e->bg_pending;
    if (drained) {      // what the decode steps' launches left of the background pass: one launch without an engine part runs it to the end
        HIPCHK(e, hipEventRecord(e->ev_drain0, st));
        OarEngineArgs a{};
        a.NG = 8; a.R = 2; a.D = 4; a.B = 1; a.bg = e->d_bgq; a.bg_only = 1;
        a.st = e->d_state; a.ticket = e->eng_ticket; a.err = e->eng_err;
        memcpy(a.xcc_group, e->eng_fg.map, 16);
        a.fp16 = std::is_same<T, f16_t>::value ? 1 : 0;
        HIPCHK(e, launch_oar_engine(st, a));
        HIPCHK(e, hipEventRecord(e->ev_drain1, st));
        e->bg_state_host.assign((size_t)kBgMaxWorkers * 4, 0u);
        HIPCHK(e, hipMemcpyAsync(e->bg_state_host.data(), &e->d_bgq->state[0][0], sizeof(e->d_bgq->state), hipMemcpyDeviceToHost, st));
    }
    HIPCHK(e, hipEventRecord(e->ev[3], st));
    HIPCHK(e, hipMemcpyAsync(io.out_tokens, e->d_tokens, (size_t)B * kTokPerFrame * 4, hipMemcpyDeviceToHost, st));
    int counters[8] = {};
    HIPCHK(e, hipMemcpyAsync(counters, e->d_counters, 8 * sizeof(int), hipMemcpyDeviceToHost, st));
    unsigned eng_err = 0;
    if (eng) HIPCHK(e, hipMemcpyAsync(&eng_err, wide ? e->wide_err : e->eng_err, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HIPCHK(e, hipStreamSynchronize(st));
    {
        const hipError_t le = e->launch_status != hipSuccess ? e->launch_status : hipGetLastError();
        e->launch_status = hipSuccess;
        if (le != hipSuccess) return e->fail(UMGEN_E_HIP, "a kernel launch of this frame was refused: %s", hipGetErrorString(le));
    }
    if (drained) {
        e->bg_pending = false;
        float dms = 0.f;
        hipEventElapsedTime(&dms, e->ev_drain0, e->ev_drain1);
        e->tm.bg_ms += dms;
        for (int wk = 0; wk < kBgMaxWorkers; ++wk)
            if (e->bg_state_host[(size_t)wk * 4] < e->bg_rec.ops.size()) {
                e->px.valid = false;
                if (getenv("UMGEN_DEBUG_TIMING")) {
                    fprintf(stderr, "[umgen] worker states (op, units done | arrived):");
                    for (int k = 0; k < kBgMaxWorkers; ++k) fprintf(stderr, " %u:%x", e->bg_state_host[(size_t)k * 4], e->bg_state_host[(size_t)k * 4 + 1]);
                    const unsigned o0 = e->bg_state_host[(size_t)wk * 4];
                    unsigned arr[4] = {};
                    (void)hipMemcpy(arr, &e->d_bgq->arrive[o0 > 0 ? o0 - 1 : 0], sizeof(arr), hipMemcpyDeviceToHost);
                    const BgOpHead& h = e->bg_rec.ops[o0].h;
                    fprintf(stderr, "\n[umgen] arrivals from op %u on: %u %u %u %u; op %u: kind %d mode %d units %d chunk %d i0..3 %d %d %d %d\n", o0 > 0 ? o0 - 1 : 0, arr[0], arr[1], arr[2], arr[3], o0,
                            h.kind, h.mode, h.n_units, h.chunk, h.i0, h.i1, h.i2, h.i3);
                }
                return e->fail(UMGEN_E_HIP, "background pass incomplete: worker %d stopped at op %u of %zu", wk, e->bg_state_host[(size_t)wk * 4], e->bg_rec.ops.size());
            }
        e->px.valid = true;
        if (getenv("UMGEN_DEBUG_TIMING")) {
            fprintf(stderr, "[umgen] background pass: %zu ops, drain launch %.2f ms\n", e->bg_rec.ops.size(), dms);
            // where a worker's time goes, by kind: sum over the ops of (units per worker) x (the slowest batch's time per unit) -- an upper bound
            std::vector<unsigned> est(e->bg_rec.ops.size());
            if (hipMemcpy(est.data(), &e->d_bgq->est[0], est.size() * 4, hipMemcpyDeviceToHost) == hipSuccess) {
                double by_kind[8] = {}, gemm_by_mode[4] = {};
                for (size_t i = 0; i < est.size(); ++i) {
                    const BgOpHead& h = e->bg_rec.ops[i].h;
                    const double us = std::ceil((double)h.n_units / 128.0) * ((double)est[i] - 20.0) / 1.125 / 100.0;
                    by_kind[h.kind & 7] += us;
                    if (h.kind == BG_GEMM) gemm_by_mode[h.mode & 3] += us;
                }
                fprintf(stderr, "[umgen] per-worker time by kind (ms, upper bound): gemm %.1f (store %.1f, resid %.1f, vt %.1f) ln %.1f attn_s %.1f attn_t %.1f embed %.1f warp %.1f\n", by_kind[BG_GEMM] / 1e3,
                        gemm_by_mode[GEMM_STORE] / 1e3, gemm_by_mode[GEMM_RESID] / 1e3, gemm_by_mode[GEMM_VT] / 1e3, by_kind[BG_LN] / 1e3, by_kind[BG_ATTN_S] / 1e3, by_kind[BG_ATTN_T] / 1e3,
                        by_kind[BG_EMBED] / 1e3, by_kind[BG_WARP] / 1e3);
            }
        }
    }
    if (eng_err) {   // a hand-off of the decode engine timed out (e.g. two engines sharing one GPU): never return tokens from such a frame
        (void)hipMemset(wide ? e->wide_err : e->eng_err, 0, sizeof(unsigned));
        return e->fail(UMGEN_E_HIP, "decode engine gave up waiting for hand-off tag 0x%08x (is another persistent kernel using this GPU?)", eng_err);
    }
    if (tr && tr->counters) memcpy(tr->counters, counters, sizeof(counters));
    float ms;
    hipEventElapsedTime(&ms, e->ev[0], e->ev[1]); e->tm.ego_ms += ms;
    hipEventElapsedTime(&ms, e->ev[1], e->ev[2]); e->tm.tar_ms += ms;
    hipEventElapsedTime(&ms, e->ev[2], e->ev[3]); e->tm.oar_ms += ms;
    e->last_oar_ms = ms;
    if (!use_px) {
        float pre = 0.f;
        hipEventElapsedTime(&pre, e->ev[0], e->ev[2]);
        e->last_full_pre_ms = pre;
    }
    hipEventElapsedTime(&ms, e->ev[0], e->ev[3]); e->tm.total_ms += ms;
    e->tm.frames += 1;
    e->tm.decode_engine = eng ? (wide ? 3 : 1) : 0;      // 1: the XCD-resident engine, 3: the chip-wide engine of the wide layers (2: the retired multi-scene engine of round 5)
    e->tm.decode_batched = batched ? 1 : 0;
    e->tm.decode_lanes = batched ? n_lanes : 0;
    e->tm.engine_fallback = e->eng_fallback ? 1 : 0;
    if (use_px) e->tm.overlapped_frames += 1;
    if (e->profiling) {
        for (size_t i = 0; i < e->gemm_ev_used; ++i) {
            hipEventElapsedTime(&ms, e->gemm_ev[i].first, e->gemm_ev[i].second);
            e->tm.gemm_ms += ms;
        }
        e->tm.gemm_launches += (int64_t)e->gemm_ev_used;
        e->tm.gemm_flops += e->gemm_flops_pending;
        e->gemm_ev_used = 0;
        e->gemm_flops_pending = 0;
        for (size_t i = 0; i < e->attn_ev_used; ++i) {
            hipEventElapsedTime(&ms, e->attn_ev[i].first, e->attn_ev[i].second);
            e->tm.attn_ms += ms;
        }
        for (size_t i = 0; i < e->layer_ev_used; ++i) {
            hipEventElapsedTime(&ms, e->layer_ev[i].first, e->layer_ev[i].second);
            e->tm.layers_ms += ms;
        }
        e->tm.layers_launches += (int64_t)e->layer_ev_used;
        e->layer_ev_used = 0;
        e->tm.attn_launches += (int64_t)e->attn_ev_used;
        e->tm.attn_flops += e->attn_flops_pending;
        e->attn_ev_used = 0;
        e->attn_flops_pending = 0;
    }
    // algorithmic HBM bytes of this frame's decode steps: weights once per step + KV read + KV write
    {
        const double w_layer = (double)e->tsz * (12.0 * E * E) + 4.0 * (6.0 * E);   // 3E*E + E*E + 4E*E + 4E*E weights, biases/LN
        const double w_oar = w_layer * (double)e->oar.size();
        double bytes = 0;
        for (int j = 0; j < kImgEos; ++j) {
            bytes += w_oar + (double)B * (double)e->oar.size() * (double)e->tsz * 2.0 * E * (double)(j + 1);
            const int V = (j >= kMapC0 && j < kMapEos) ? e->cfg.map_vocab : (j >= kBoxC0 && j < kBoxEos) ? e->cfg.bbox3d_vocab
                        : (j >= kImgC0 && j < kImgEos) ? e->cfg.img_vocab : 0;
            bytes += (double)V * E * (double)e->tsz;
        }
        bytes += (double)e->cfg.bbox3d_vocab * E * (double)e->tsz;   // head_tar_bbox3d: one GEMM per frame (tar_head_logits), not one GEMV per bbox3d step
        e->tm.oar_bytes += bytes;
    }
    return 0;
}

int run_frame_any(umgen_engine* e, const FrameIO& io) {
    const int rc = e->cfg.precision == UMGEN_PREC_BF16 ? run_frame<bf16_t>(e, io)
                 : e->cfg.precision == UMGEN_PREC_FP16 ? run_frame<f16_t>(e, io) : run_frame<float>(e, io);
    if (rc != UMGEN_OK) {
        // A failed frame may have left a stream capture open, async copies in flight that read this call's host buffers, and a
        // background pass the next call would wait for: drain everything and forget the pass (the error message is kept).
        const std::string msg = e->err;
        for (hipStream_t s : {e->stream, e->full_stream, e->bg_stream, e->side_stream[0], e->side_stream[1], e->lane[0].s, e->lane[1].s, e->lane[2].s,
                              e->lane[3].s, e->lane[4].s, e->lane[5].s, e->lane[6].s, e->lane[7].s}) {
            if (!s) continue;
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
                hipGraph_t g = nullptr;
                (void)hipStreamEndCapture(s, &g);
                if (g) (void)hipGraphDestroy(g);
            }
        }
        (void)hipDeviceSynchronize();
        (void)hipGetLastError();
        e->bg_pending = false;
        e->px.valid = false;
        e->err = msg;
    }
    return rc;
}

template <typename T>
int build_tables(umgen_engine* e) {
    // GMLP(codebook) rows (module.py:710-743 applied once to each of the 8192 codes): fp32 activations, exact FMA chain
    const int E = e->E;
    for (int which = 0; which < 2; ++which) {
        const int V = which ? e->cfg.img_vocab : e->cfg.map_vocab;
        const int C = which ? e->cfg.n_img_embd : e->cfg.n_map_embd;
        float* table;
        if (int rc = dalloc(e, &table, (size_t)V * E)) return rc;
        float* hid;
        HIPCHK(e, hipMalloc(&hid, (size_t)V * 4 * E * 4));
        GemmArgs g{};
        g.P = which ? e->img_fc : e->map_fc; g.Q = which ? e->img_cb : e->map_cb; g.Mi = 4 * E; g.Nj = V; g.K = C; g.ldp = C; g.ldq = C;
        g.batch = 1; g.mode = GEMM_STORE; g.gelu = 1; g.out = hid; g.ldo = 4L * E;
        Path<T>::gemm_w_f32act(e->stream, g);
        GemmArgs g2{};
        g2.P = which ? e->img_proj : e->map_proj; g2.Q = hid; g2.Mi = E; g2.Nj = V; g2.K = 4 * E; g2.ldp = 4 * E; g2.ldq = 4 * E;
        g2.batch = 1; g2.mode = GEMM_STORE; g2.out = table; g2.ldo = E;
        Path<T>::gemm_w_f32act(e->stream, g2);
        HIPCHK(e, hipStreamSynchronize(e->stream));
        HIPCHK(e, hipFree(hid));
        if (which) e->tb.gimg = table; else e->tb.gmap = table;
    }
    return 0;
}

// Decode engine (oar_engine.hip): the mlp c_proj of every BlockOAR, repacked for the hidden-unit split.  CU c of a group owns
// hidden units 96 c .. 96 c + 95; thread t of its workgroup holds, as 16-byte units of 8 bf16 in the order it requests them,
//   unit j < 12 : W[4 (t / 4) + j / 3][96 c + 24 (t % 4) + 8 (j % 3) .. + 7]      (four lanes share rows 4 (t / 4) .. + 3)
//   units 12..17: 48 weights, weight i = W[512 + 4 (t / 8) + i / 12][96 c + 12 (t % 8) + i % 12]   (eight lanes share four of the rows 512..767)
// layout [32 CUs][18 units][512 threads][8]: a wave's request of one unit is 1 KB contiguous.
int repack_mlp_proj(umgen_engine* e) {
    const int E = e->E, F4 = 4 * E;
    std::vector<bf16_t> src((size_t)E * F4), dst((size_t)kEngGroup * kEngWpUnits * kEngThreads * 8);
    std::vector<OarLayerDev> hl(e->oar.size());
    HIPCHK(e, hipMemcpy(hl.data(), e->d_layers, hl.size() * sizeof(OarLayerDev), hipMemcpyDeviceToHost));
    if (e->eng_wp2.size() != e->oar.size()) {
        e->eng_wp2.assign(e->oar.size(), nullptr);
        for (auto& p : e->eng_wp2)
            if (int rc = dev_alloc(e, &p, dst.size() * sizeof(bf16_t))) return rc;
    }
    for (size_t li = 0; li < e->oar.size(); ++li) {
        HIPCHK(e, hipMemcpy(src.data(), e->oar[li].mlp.Wproj, src.size() * sizeof(bf16_t), hipMemcpyDeviceToHost));
        for (int c = 0; c < kEngGroup; ++c)
            for (int j = 0; j < kEngWpUnits; ++j)
                for (int t = 0; t < kEngThreads; ++t) {
                    bf16_t* d8 = &dst[(((size_t)c * kEngWpUnits + j) * kEngThreads + t) * 8];
                    if (UMGEN_ENG_MFMA & 8) {
                        // matrix-core form (oar_engine.hip, kMfmaP): unit f = 3 tile + kstep of lane (t % 64) of wave (t / 64) is the A fragment
                        // W[96 wave + 16 tile + lane % 16][96 c + 32 kstep + 8 (lane / 16) .. + 7]
                        const int wave = t / 64, lane = t % 64, tile = j / 3, ks = j % 3;
                        memcpy(d8, &src[(size_t)(96 * wave + 16 * tile + lane % 16) * F4 + 96 * c + 32 * ks + 8 * (lane / 16)], 8 * sizeof(bf16_t));
                    } else if (j < 12) {
                        memcpy(d8, &src[(size_t)(4 * (t / 4) + j / 3) * F4 + 96 * c + 24 * (t % 4) + 8 * (j % 3)], 8 * sizeof(bf16_t));
                    } else {
                        for (int k = 0; k < 8; ++k) {
                            const int i = 8 * (j - 12) + k;
                            d8[k] = src[(size_t)(512 + 4 * (t / 8) + i / 12) * F4 + 96 * c + 12 * (t % 8) + i % 12];
                        }
                    }
                }
        HIPCHK(e, hipMemcpy(e->eng_wp2[li], dst.data(), dst.size() * sizeof(bf16_t), hipMemcpyHostToDevice));
        hl[li].Wp2 = reinterpret_cast<const bf16_t*>(e->eng_wp2[li]);
        hl[li].Wf2 = nullptr;
        if (UMGEN_ENG_MFMA & 4) {
            // c_fc [4E][E] as the engine's A fragments: CU c owns rows 96 c .. + 95 (6 tiles of 16), wave v the k range 96 v .. + 95
            // (3 k-steps of 32); fragment f = 3 tile + kstep of lane l = W[96 c + 16 tile + l % 16][96 v + 32 kstep + 8 (l / 16) .. + 7],
            // stored [c][v][f][l][8]: a wave's request of one fragment is 1 KB contiguous (row-strided 64-byte pieces streamed at
            // two thirds of the rate where the weight stream is not hidden: 4 scenes 564 vs 511 us per launch)
            if (e->eng_wf2.size() != e->oar.size()) e->eng_wf2.assign(e->oar.size(), nullptr);
            std::vector<bf16_t> fsrc((size_t)F4 * E), fdst((size_t)F4 * E);
            if (!e->eng_wf2[li])
                if (int rc = dev_alloc(e, &e->eng_wf2[li], fdst.size() * sizeof(bf16_t))) return rc;
            HIPCHK(e, hipMemcpy(fsrc.data(), e->oar[li].mlp.Wfc, fsrc.size() * sizeof(bf16_t), hipMemcpyDeviceToHost));
            for (int c = 0; c < kEngGroup; ++c)
                for (int v = 0; v < 8; ++v)
                    for (int f = 0; f < 18; ++f)
                        for (int ln = 0; ln < 64; ++ln)
                            memcpy(&fdst[((((size_t)c * 8 + v) * 18 + f) * 64 + ln) * 8],
                                   &fsrc[(size_t)(96 * c + 16 * (f / 3) + ln % 16) * E + 96 * v + 32 * (f % 3) + 8 * (ln / 16)], 8 * sizeof(bf16_t));
            HIPCHK(e, hipMemcpy(e->eng_wf2[li], fdst.data(), fdst.size() * sizeof(bf16_t), hipMemcpyHostToDevice));
            hl[li].Wf2 = reinterpret_cast<const bf16_t*>(e->eng_wf2[li]);
        }
    }
    HIPCHK(e, hipMemcpy(e->d_layers, hl.data(), hl.size() * sizeof(OarLayerDev), hipMemcpyHostToDevice));
    return 0;
}

// Layer table of the chip-wide engine (oar_engine_wide.hip) + its mlp c_proj slices: rank r multiplies its 24 hidden units h[24 r ..] into ALL E
// output rows, so its slice is W[row][24 r .. 24 r + 23] for every row: [256 ranks][E rows][24], 48 contiguous bytes per (rank, row).
int repack_wide(umgen_engine* e) {
    const int E = e->E, F4 = 4 * E, RF = F4 / kWideGroups;
    std::vector<bf16_t> src((size_t)E * F4), dst((size_t)E * F4);
    std::vector<OarLayerDev> hl(e->oar.size());
    if (e->wide_wp2.size() != e->oar.size()) {
        e->wide_wp2.assign(e->oar.size(), nullptr);
        for (auto& p : e->wide_wp2)
            if (int rc = dev_alloc(e, &p, dst.size() * sizeof(bf16_t))) return rc;
    }
    for (size_t li = 0; li < e->oar.size(); ++li) {
        const SubW& w = e->oar[li];
        HIPCHK(e, hipMemcpy(src.data(), w.mlp.Wproj, src.size() * sizeof(bf16_t), hipMemcpyDeviceToHost));
        for (int r = 0; r < kWideGroups; ++r)
            for (int row = 0; row < E; ++row)
                memcpy(&dst[((size_t)r * E + row) * RF], &src[(size_t)row * F4 + (size_t)RF * r], RF * sizeof(bf16_t));
        HIPCHK(e, hipMemcpy(e->wide_wp2[li], dst.data(), dst.size() * sizeof(bf16_t), hipMemcpyHostToDevice));
        hl[li] = OarLayerDev{reinterpret_cast<const bf16_t*>(w.attn.Wqkv), reinterpret_cast<const bf16_t*>(w.attn.Wo),
                             reinterpret_cast<const bf16_t*>(w.mlp.Wfc), reinterpret_cast<const bf16_t*>(w.mlp.Wproj),
                             reinterpret_cast<const bf16_t*>(e->wide_wp2[li]), w.attn.bqkv, w.attn.bo, w.ln_a, w.ln_b, nullptr};
    }
    HIPCHK(e, hipMemcpy(e->d_layers_wide, hl.data(), hl.size() * sizeof(OarLayerDev), hipMemcpyHostToDevice));
    return 0;
}

}  // namespace

// =============================================================================================================
// C ABI
// =============================================================================================================
extern "C" {

#ifndef UMGEN_SRC_HASH
#define UMGEN_SRC_HASH "unknown"
#endif
const char* umgen_version(void) { return "umgen_hip 0.2 (gfx950) src " UMGEN_SRC_HASH; }
const char* umgen_last_error(const umgen_engine* e) { return e ? e->err.c_str() : "null engine"; }

int umgen_create(const umgen_config* cfg, umgen_engine** out) {
    if (!cfg || !out) return UMGEN_E_INVALID;
    *out = nullptr;
    umgen_engine* e = new umgen_engine();
    *out = e;   // returned even on failure so the caller can read umgen_last_error()
    e->cfg = *cfg;
    if (cfg->abi_version != UMGEN_ABI_VERSION) return e->fail(UMGEN_E_INVALID, "abi_version %d != %d", cfg->abi_version, UMGEN_ABI_VERSION);
    if (cfg->n_head <= 0 || cfg->n_embd != cfg->n_head * kHeadDim)
        return e->fail(UMGEN_E_UNSUPPORTED, "head_dim must be %d (n_embd=%d, n_head=%d)", kHeadDim, cfg->n_embd, cfg->n_head);
    if (cfg->n_embd > 1536) return e->fail(UMGEN_E_UNSUPPORTED, "n_embd <= 1536 supported");
    if (cfg->map_vocab > 8192 || cfg->img_vocab > 8192 || cfg->bbox3d_vocab != 1028 || cfg->pose_vocab > 8192)
        return e->fail(UMGEN_E_UNSUPPORTED, "vocab sizes out of range");
    if (cfg->max_cond_frames > 64) return e->fail(UMGEN_E_UNSUPPORTED, "max_cond_frames <= 64 supported (temporal attention tile)");
    if (cfg->max_batch < 1 || cfg->max_cond_frames < 1 || cfg->max_cond_frames > cfg->max_frame_len)
        return e->fail(UMGEN_E_INVALID, "max_batch / max_cond_frames invalid");
    if (cfg->precision != UMGEN_PREC_FP32 && cfg->precision != UMGEN_PREC_BF16 && cfg->precision != UMGEN_PREC_FP16)
        return e->fail(UMGEN_E_INVALID, "precision %d (UMGEN_PREC_FP32 / _BF16 / _FP16)", cfg->precision);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return e->fail(UMGEN_E_HIP, "no HIP device visible: libumgen_hip has no CPU fallback");
    HIPCHK(e, hipSetDevice(cfg->device));
    HIPCHK(e, gemm256_prepare());   // per device: dynamic-LDS attribute + CU count of the 256 x 256 GEMM (a second GPU in one process gets its own)
    // overlapped TAR pass (UMGEN_OVERLAP=0 disables it): the decode stream and the background stream get disjoint CU masks --
    // measured on MI355X, a decode loop sharing CUs with a concurrent GEMM stream runs at a quarter of its speed, with disjoint
    // masks (64 background CUs) it loses 8 %
    // The decode engine (oar_engine.hip) needs whole XCDs: 32 workgroups, one per CU, on each of them.  A CU mask cannot give
    // that -- measured with the engine's census: the mask bits are striped over the XCDs (64 background CUs = 8 CUs of EVERY
    // XCD), so a masked decode stream has 24 CUs per XCD.  The engine halves the decode loop, which is worth more than hiding the
    // TAR pass behind a launch-bound loop: when the engine can be used the overlap is off unless UMGEN_OVERLAP asks for it
    // (then the decode step is the five-launch form again).
    const char* de_env = getenv("UMGEN_DECODE_ENGINE");
    const char* ov_env = getenv("UMGEN_OVERLAP");
    // hand-off tags: (round or scene, layer, edge) of one step must fit kEpochPerStep (oar_engine.hip): <= 64 layers; up to 32 scenes flow
    // through the systolic schedule, more run as rounds of 8 whole-scene groups (<= 32 rounds)
    const bool engine_wanted = cfg->precision != UMGEN_PREC_FP32 && cfg->n_embd == kEngE && cfg->n_head == kEngH && cfg->n_oar_layer <= 64 &&
                               cfg->max_batch <= 256 && !(de_env && de_env[0] == '0') && !(ov_env && ov_env[0] != '0');
    if (engine_wanted) {
        // Census FIRST, on the plain stream the engine would use: an engine-shaped launch (one 512-thread workgroup per CU) must put
        // exactly 32 workgroups on each of 8 XCDs, twice in a row with the same XCD map.  Only when that holds is the overlap given
        // up for the engine; otherwise (partitioned GPU, another SKU, CUs busy with somebody else's persistent kernel) the engine
        // falls back to the five-launch decode layer WITH the overlapped TAR pass, and says so.
        HIPCHK(e, hipStreamCreate(&e->stream));
        HIPCHK(e, oar_engine_prepare());
        unsigned* d_cnt = nullptr;
        HIPCHK(e, hipMalloc(&d_cnt, 64));
        umgen_engine::EngStream& es = e->eng_fg;
        es.ok = true;
        for (int rep = 0; rep < 2 && es.ok; ++rep) {
            unsigned cnt[16] = {};
            if (hipMemsetAsync(d_cnt, 0, 64, e->stream) != hipSuccess || launch_oar_engine_census(e->stream, 8, d_cnt) != hipSuccess ||
                hipMemcpyAsync(cnt, d_cnt, 64, hipMemcpyDeviceToHost, e->stream) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess) {
                (void)hipGetLastError();
                es.ok = false;
                break;
            }
            if (getenv("UMGEN_DEBUG_TIMING")) {
                fprintf(stderr, "[umgen] engine census:");
                for (int x = 0; x < 16; ++x) fprintf(stderr, " %u", cnt[x]);
                fprintf(stderr, "\n");
            }
            int groups = 0;
            unsigned char map[16];
            for (int x = 0; x < 16; ++x) {
                map[x] = 0xff;
                if (cnt[x] == (unsigned)kEngGroup) map[x] = (unsigned char)groups++;
                else if (cnt[x] != 0) es.ok = false;
            }
            if (groups != 8 || (rep == 1 && memcmp(map, es.map, 16))) es.ok = false;
            memcpy(es.map, map, 16);
        }
        (void)hipFree(d_cnt);
        es.NG = 8;
        e->eng_enabled = es.ok;
        if (!es.ok) {
            e->eng_fallback = true;
            fprintf(stderr, "[umgen] WARNING: the XCD-resident decode engine cannot be used on device %d (its census did not find 32 workgroups on each "
                            "of 8 XCDs); decode steps run as five launches per layer (~30 %% slower at one scene per GPU)\n", cfg->device);
            HIPCHK(e, hipStreamDestroy(e->stream));
            e->stream = nullptr;
        }
    }
    // Wide layers (n_embd 1536): the chip-wide engine needs all 256 CUs of the decode stream at once (one persistent workgroup per CU: the same
    // census as the XCD-resident engine's, 32 workgroups on each of 8 XCDs, twice) and, like it, gives up the CU-masked background TAR pass.
    const char* dw_env = getenv("UMGEN_DECODE_WIDE");
    if (cfg->precision != UMGEN_PREC_FP32 && cfg->n_embd == kWideE && cfg->n_head == kWideE / kHeadDim && cfg->n_oar_layer <= 64 &&
        cfg->max_batch <= std::max(0, std::min(4, dw_env ? atoi(dw_env) : 1)) && !(ov_env && ov_env[0] != '0')) {
        HIPCHK(e, hipStreamCreate(&e->stream));
        HIPCHK(e, oar_engine_wide_prepare());
        unsigned* d_cnt = nullptr;
        HIPCHK(e, hipMalloc(&d_cnt, 64));
        bool ok = true;
        for (int rep2 = 0; rep2 < 2 && ok; ++rep2) {
            unsigned cnt[16] = {};
            if (hipMemsetAsync(d_cnt, 0, 64, e->stream) != hipSuccess || launch_oar_engine_wide_census(e->stream, d_cnt) != hipSuccess ||
                hipMemcpyAsync(cnt, d_cnt, 64, hipMemcpyDeviceToHost, e->stream) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess) {
                (void)hipGetLastError();
                ok = false;
                break;
            }
            int groups = 0;
            for (int x = 0; x < 16; ++x) {
                if (cnt[x] == (unsigned)kEngGroup) ++groups;
                else if (cnt[x] != 0) ok = false;
            }
            if (groups != 8) ok = false;
        }
        (void)hipFree(d_cnt);
        e->wide_enabled = ok;
        if (!ok) {
            fprintf(stderr, "[umgen] WARNING: the chip-wide decode engine cannot be used on device %d (its census did not find one workgroup on each of 256 "
                            "CUs); decode steps run as five launches per layer\n", cfg->device);
            HIPCHK(e, hipStreamDestroy(e->stream));
            e->stream = nullptr;
        }
    }
    e->overlap = cfg->max_cond_frames >= 2 && !e->eng_enabled && !e->wide_enabled;
    if (ov_env) { e->overlap_mode = ov_env[0] - '0'; e->overlap = cfg->max_cond_frames >= 2 && ov_env[0] != '0'; }
    // One scene per GPU on the XCD-resident engine: the overlapped pass runs on the engine's idle XCDs (bg_worker.h; UMGEN_BG_ENGINE=0: the engine on all
    // eight groups and every frame's whole window in the foreground, as in rounds 2-5).  Engines for more scenes keep every XCD busy with the decode.
    const char* be_env = getenv("UMGEN_BG_ENGINE");
    e->bg_engine = e->eng_enabled && cfg->max_batch == 1 && cfg->max_cond_frames >= 2 && cfg->max_cond_frames <= 32 && !ov_env && !(be_env && be_env[0] == '0');
    if (e->bg_engine) { e->overlap = true; e->overlap_mode = 1; }
    int bg_cus = 64;   // mask bits are striped over the 8 XCDs: 64 = 8 CUs of each XCD for the background stream
    if (const char* bc = getenv("UMGEN_BG_CUS")) bg_cus = std::max(32, std::min(128, atoi(bc)));
    if (e->bg_engine) {
        void* qp = nullptr;
        if (int rc = dev_alloc(e, &qp, sizeof(BgQueue))) return rc;
        e->d_bgq = reinterpret_cast<BgQueue*>(qp);
        HIPCHK(e, hipMemset(e->d_bgq, 0, sizeof(BgQueue)));
        HIPCHK(e, hipEventCreate(&e->ev_drain0));
        HIPCHK(e, hipEventCreate(&e->ev_drain1));
    }
    if (e->overlap && !e->bg_engine) {
        hipDeviceProp_t prop;
        HIPCHK(e, hipGetDeviceProperties(&prop, cfg->device));
        const int ncu = prop.multiProcessorCount;
        if (ncu < 2 * bg_cus) e->overlap = false;
        else {
            std::vector<uint32_t> mbg((ncu + 31) / 32, 0u), mfg((ncu + 31) / 32, 0u);
            int fg_cus = ncu - bg_cus;   // UMGEN_FG_CUS: experiment, decode loop on fewer CUs
            if (const char* fc = getenv("UMGEN_FG_CUS")) fg_cus = std::max(32, std::min(ncu - bg_cus, atoi(fc)));
            e->fg_xcds = fg_cus / 32;
            for (int cu = 0; cu < ncu; ++cu) {
                if (cu < bg_cus) mbg[cu / 32] |= 1u << (cu % 32);
                else if (cu < bg_cus + fg_cus) mfg[cu / 32] |= 1u << (cu % 32);
            }
            // a device that refuses CU masks (e.g. a partitioned GPU) simply runs the plain one-stream path: same HIP kernels, same tokens
            if (hipExtStreamCreateWithCUMask(&e->stream, (uint32_t)mfg.size(), mfg.data()) != hipSuccess ||
                hipExtStreamCreateWithCUMask(&e->bg_stream, (uint32_t)mbg.size(), mbg.data()) != hipSuccess) {
                (void)hipGetLastError();
                if (e->stream) { hipStreamDestroy(e->stream); e->stream = nullptr; }
                if (e->bg_stream) { hipStreamDestroy(e->bg_stream); e->bg_stream = nullptr; }
                e->overlap = false;
            } else {
                HIPCHK(e, hipStreamCreateWithFlags(&e->full_stream, hipStreamNonBlocking));
                HIPCHK(e, hipEventCreate(&e->ev_pre_done));
                HIPCHK(e, hipEventCreate(&e->ev_tar_done));
                HIPCHK(e, hipEventCreate(&e->ev_bg_done));
                HIPCHK(e, hipEventCreate(&e->ev_bg0));
            }
        }
    }
    if (!e->stream) HIPCHK(e, hipStreamCreate(&e->stream));
    for (auto& ev : e->ev) HIPCHK(e, hipEventCreate(&ev));
    e->E = cfg->n_embd;
    e->H = cfg->n_head;
    if (const char* sl = getenv("UMGEN_DEBUG_SAME_LAYER")) e->dbg_same_layer = sl[0] == '1';
    if (const char* rb = getenv("UMGEN_ROWS_PER_BLOCK")) e->rows_per_block = rb[0] - '0';
    e->tsz = cfg->precision == UMGEN_PREC_FP32 ? 4 : 2;
    const int64_t E = e->E;
    const std::string t = "transformer.";
    // ---- parameters (names = the reference state-dict keys, UMGen.py:176-261) ----
    float* tmp;
    if (int rc = alloc_f32(e, t + "egoe.weight", &tmp, {3, E})) return rc; e->tb.egoe = tmp;
    if (int rc = alloc_f32(e, t + "axe.weight", &tmp, {cfg->aux_vocab, E})) return rc; e->tb.axe = tmp;
    if (int rc = alloc_f32(e, t + "be.weight", &tmp, {cfg->bbox3d_vocab, E})) return rc; e->tb.be = tmp;
    if (int rc = alloc_f32(e, t + "tpe.weight", &tmp, {cfg->max_frame_len, E})) return rc; e->tb.tpe = tmp;
    if (int rc = alloc_f32(e, t + "spe.weight", &tmp, {kSeq, E})) return rc; e->tb.spe = tmp;
    if (int rc = alloc_f32(e, t + "tske.weight", &tmp, {cfg->task_num, E})) return rc; e->tb.tske = tmp;
    e->tb.E = e->E;
    const char* stack_name[4] = {"ego_tar", "map_tar", "box_tar", "TAR"};
    const int stack_n[4] = {cfg->n_ego_tar_layer, cfg->n_map_tar_layer, cfg->n_box_tar_layer, cfg->n_tar_layer};
    for (int s = 0; s < 4; ++s) {
        e->stk[s].resize(stack_n[s]);
        for (int i = 0; i < stack_n[s]; ++i) {
            const std::string pre = t + stack_name[s] + "." + std::to_string(i);
            TarW& b = e->stk[s][i];
            if (int rc = alloc_sub(e, pre, "ln_1", "spatial_attn_1", "ln_2", "mlp1", b.sub[0])) return rc;
            if (int rc = alloc_sub(e, pre, "ln_3", "temporal_attn", "ln_4", "mlp2", b.sub[1])) return rc;
            if (int rc = alloc_sub(e, pre, "ln_5", "spatial_attn_2", "ln_6", "mlp3", b.sub[2])) return rc;
        }
    }
    e->oar.resize(cfg->n_oar_layer);
    for (int i = 0; i < cfg->n_oar_layer; ++i)
        if (int rc = alloc_sub(e, t + "OAR." + std::to_string(i), "ln_1", "temporal_attn", "ln_2", "mlp", e->oar[i])) return rc;
    e->dec.resize(cfg->n_ego_ca_layer);
    for (int i = 0; i < cfg->n_ego_ca_layer; ++i) {
        const std::string pre = t + "ego_cross_attn." + std::to_string(i);
        DecW& d = e->dec[i];
        if (int rc = alloc_f32(e, pre + ".ln_1.weight", &d.ln1, {E})) return rc;
        if (int rc = alloc_attn(e, pre + ".self_attn", d.self)) return rc;
        if (int rc = alloc_f32(e, pre + ".ln_2.weight", &d.ln2, {E})) return rc;
        if (int rc = alloc_f32(e, pre + ".ln_3.weight", &d.ln3, {E})) return rc;
        if (int rc = alloc_w(e, pre + ".cross_attn.q_attn.weight", &d.Wq, {E, E})) return rc;
        if (int rc = alloc_f32(e, pre + ".cross_attn.q_attn.bias", &d.bq, {E})) return rc;
        // k_attn | v_attn packed into one [2E][E] projection
        if (int rc = dev_alloc(e, &d.Wkv, (size_t)2 * E * E * e->tsz)) return rc;
        if (int rc = dalloc(e, &d.bkv, (size_t)2 * E)) return rc;
        reg(e, pre + ".cross_attn.k_attn.weight", d.Wkv, {E, E}, 1);
        reg(e, pre + ".cross_attn.v_attn.weight", reinterpret_cast<char*>(d.Wkv) + (size_t)E * E * e->tsz, {E, E}, 1);
        reg(e, pre + ".cross_attn.k_attn.bias", d.bkv, {E}, 0);
        reg(e, pre + ".cross_attn.v_attn.bias", d.bkv + E, {E}, 0);
        if (int rc = alloc_w(e, pre + ".cross_attn.c_proj.weight", &d.Wco, {E, E})) return rc;
        if (int rc = alloc_f32(e, pre + ".cross_attn.c_proj.bias", &d.bco, {E})) return rc;
        if (int rc = alloc_f32(e, pre + ".ln_4.weight", &d.ln4, {E})) return rc;
        if (int rc = alloc_mlp(e, pre + ".mlp1", d.mlp)) return rc;
    }
    if (int rc = alloc_f32(e, t + "ln_ego_tar.weight", &e->ln_ego_tar, {E})) return rc;
    if (int rc = alloc_f32(e, t + "ln_ego.weight", &e->ln_ego, {E})) return rc;
    if (int rc = alloc_f32(e, t + "ln_tar.weight", &e->ln_tar, {E})) return rc;
    if (int rc = alloc_f32(e, t + "ln_oar.weight", &e->ln_oar, {E})) return rc;
    if (int rc = alloc_f32(e, t + "ln_map_tar.weight", &e->ln_map_tar, {E})) return rc;
    if (int rc = alloc_f32(e, t + "ln_box_tar.weight", &e->ln_box_tar, {E})) return rc;
    if (int rc = alloc_w(e, t + "head_ego.weight", &e->head_ego, {cfg->pose_vocab, E})) return rc;
    if (int rc = alloc_w(e, t + "head_ar_map.weight", &e->head_ar_map, {cfg->map_vocab, E})) return rc;
    if (int rc = alloc_w(e, t + "head_ar_bbox3d.weight", &e->head_ar_box, {cfg->bbox3d_vocab, E})) return rc;
    if (int rc = alloc_w(e, t + "head_tar_bbox3d.weight", &e->head_tar_box, {cfg->bbox3d_vocab, E})) return rc;
    if (int rc = alloc_w(e, t + "head_ar_img.weight", &e->head_ar_img, {cfg->img_vocab, E})) return rc;
    if (int rc = alloc_w(e, "map_mlp_pre.c_fc.weight", &e->map_fc, {4 * E, cfg->n_map_embd})) return rc;
    if (int rc = alloc_w(e, "map_mlp_pre.c_proj.weight", &e->map_proj, {E, 4 * E})) return rc;
    if (int rc = alloc_w(e, "img_mlp_pre.c_fc.weight", &e->img_fc, {4 * E, cfg->n_img_embd})) return rc;
    if (int rc = alloc_w(e, "img_mlp_pre.c_proj.weight", &e->img_proj, {E, 4 * E})) return rc;
    if (int rc = alloc_f32(e, "map_codebook.weight", &e->map_cb, {cfg->map_vocab, cfg->n_map_embd})) return rc;
    if (int rc = alloc_f32(e, "img_codebook.weight", &e->img_cb, {cfg->img_vocab, cfg->n_img_embd})) return rc;
    // bf16 constant tables: computed at finalize unless a checkpoint provides them (UMGen.py:257-261)
    bf16_t *fp, *po, *gp;
    if (int rc = dalloc(e, &fp, (size_t)1024 * E)) return rc;
    if (int rc = dalloc(e, &po, (size_t)1030 * E)) return rc;
    if (int rc = dalloc(e, &gp, (size_t)1024 * E)) return rc;
    e->tb.fouier_pe = fp; e->tb.posi = po; e->tb.grid_posi = gp;
    reg(e, "fouier_pe", fp, {1024, E}, 2, true);
    reg(e, "bbox3d_spatial_posi", po, {1030, E}, 2, true);
    reg(e, "grid_center_posi_embedding", gp, {1024, E}, 2, true);

    // ---- workspace ----
    const size_t Bm = cfg->max_batch, Tm = cfg->max_cond_frames;
    const size_t R = Bm * Tm * kSeq;
    e->S_pad = ((kSeq + 63) / 64) * 64;
    if (int rc = dalloc(e, &e->X, R * E)) return rc;
    if (int rc = dev_alloc(e, &e->A, R * E * e->tsz)) return rc;
    if (int rc = dev_alloc(e, &e->QKV, R * 3 * E * e->tsz)) return rc;
    if (int rc = dev_alloc(e, &e->VT, Bm * Tm * E * e->S_pad * e->tsz)) return rc;
    HIPCHK(e, hipMemset(e->VT, 0, Bm * Tm * E * e->S_pad * e->tsz));   // pad columns stay zero forever
    if (int rc = dev_alloc(e, &e->Hb, R * 4 * E * e->tsz)) return rc;
    if (int rc = dalloc(e, &e->mapfeat, Bm * Tm * kNMap * E)) return rc;
    e->w_main = umgen_engine::Work{e->X, e->A, e->QKV, e->VT, e->Hb, e->mapfeat};
    // The three TAR stacks of a frame are independent (UMGen.py:1484-1494 feeds each the same window): in the plain path the map and box
    // stacks run on two side streams with whole-window workspaces of their own, so that the tail of every launch (a persistent GEMM's
    // last partial round of tiles, the ragged last attention blocks, the gaps between dependent launches) is filled by the other
    // stacks' workgroups instead of idling.  (The overlapped pass of round 1 uses the same streams with 1-slot workspaces.)
    const char* cs_env = getenv("UMGEN_CONCURRENT_STACKS");
    e->conc_stacks = !e->overlap && (cs_env ? cs_env[0] != '0' : true);
    if (e->overlap || e->conc_stacks) {   // 1-slot (overlap) / whole-window (concurrent stacks) workspaces + streams
        const size_t slots = e->overlap ? 1 : Tm;
        // the side workspaces hold the map stack (1031 rows per frame) and the box stack (1693), not 2207; when they do not fit beside
        // the main workspace and the caches (a large max_batch), the stacks simply run one behind the other on one stream
        const int side_len[2] = {stack_len(STACK_MAP), stack_len(STACK_BOX)};
        size_t need = 0, free_b = 0, total_b = 0;
        for (int i = 0; i < 2; ++i)
            need += Bm * slots * ((size_t)side_len[i] * E * (4 + 8 * e->tsz) + (size_t)E * e->S_pad * e->tsz + (size_t)kNMap * E * 4);
        HIPCHK(e, hipMemGetInfo(&free_b, &total_b));
        const size_t kv_need = (size_t)cfg->n_oar_layer * Bm * (size_t)e->Lmax * 2 * E * e->tsz;
        if (!e->overlap && need + kv_need > free_b - free_b / 8) {
            e->conc_stacks = false;
            fprintf(stderr, "[umgen] note: %.1f GB of side workspaces for the concurrent map / box stacks do not fit (%.1f GB free): the three TAR stacks run "
                            "one behind the other\n", (double)need / 1e9, (double)free_b / 1e9);
        }
    }
    if (e->overlap || e->conc_stacks) {
        const size_t slots = e->overlap ? 1 : Tm;
        const int side_len[2] = {stack_len(STACK_MAP), stack_len(STACK_BOX)};
        for (int i = 0; i < 2; ++i) {
            const size_t R1 = Bm * slots * (size_t)(e->overlap ? kSeq : side_len[i]);
            umgen_engine::Work& w = e->w_side[i];
            if (int rc = dalloc(e, &w.X, R1 * E)) return rc;
            if (int rc = dev_alloc(e, &w.A, R1 * E * e->tsz)) return rc;
            if (int rc = dev_alloc(e, &w.QKV, R1 * 3 * E * e->tsz)) return rc;
            if (int rc = dev_alloc(e, &w.VT, Bm * slots * E * e->S_pad * e->tsz)) return rc;
            HIPCHK(e, hipMemset(w.VT, 0, Bm * slots * E * e->S_pad * e->tsz));
            if (int rc = dev_alloc(e, &w.Hb, R1 * 4 * E * e->tsz)) return rc;
            if (int rc = dalloc(e, &w.mapfeat, Bm * slots * kNMap * E)) return rc;
            HIPCHK(e, hipStreamCreateWithFlags(&e->side_stream[i], hipStreamNonBlocking));
            HIPCHK(e, hipEventCreate(&e->ev_side_done[i]));
        }
        HIPCHK(e, hipEventCreate(&e->ev_side_in));
    }
    if (int rc = dalloc(e, &e->warped_last, Bm * kNMap * E)) return rc;
    if (int rc = dalloc(e, &e->cond, Bm * kSeq * E)) return rc;
    if (int rc = dalloc(e, &e->pego, Bm * kSeq * E)) return rc;
    if (int rc = dalloc(e, &e->pose_diff, Bm * Tm * 3)) return rc;
    if (int rc = dalloc(e, &e->xdec, 3 * Bm * E)) return rc;
    if (int rc = dalloc(e, &e->qdec, 3 * Bm * E)) return rc;
    if (int rc = dalloc(e, &e->qkv3, 3 * Bm * 3 * E)) return rc;
    if (int rc = dalloc(e, &e->part, 3 * Bm * e->H * kAttnRec)) return rc;
    HIPCHK(e, hipMemset(e->part, 0, 3 * Bm * e->H * kAttnRec * sizeof(float)));   // never-written split slots are read with weight 0
    if (int rc = dalloc(e, &e->hdec, 3 * Bm * 4 * E)) return rc;
    if (int rc = dalloc(e, &e->xfrag, (size_t)kRowsMaxM * E)) return rc;
    if (int rc = dalloc(e, &e->afrag, (size_t)kRowsMaxM * E)) return rc;
    if (int rc = dalloc(e, &e->hfrag, (size_t)kRowsMaxM * 4 * E)) return rc;
    HIPCHK(e, hipMemset(e->xfrag, 0, (size_t)kRowsMaxM * E * 4));       // (columns past the batch are computed, never stored: keep them finite)
    HIPCHK(e, hipMemset(e->afrag, 0, (size_t)kRowsMaxM * E * 4));
    HIPCHK(e, hipMemset(e->hfrag, 0, (size_t)kRowsMaxM * 4 * E * 4));
    if (const char* bd = getenv("UMGEN_DECODE_BATCHED")) e->batched_min = atoi(bd);
    if (const char* dl = getenv("UMGEN_DECODE_LANES")) e->lanes_env = atoi(dl);
    if (e->tsz == 2 && Bm >= 2 && e->lanes_env != 1) {      // decode lanes: streams, step states and fragment buffers (1.2 MB per lane)
        for (auto& ln : e->lane) {
            HIPCHK(e, hipStreamCreateWithFlags(&ln.s, hipStreamNonBlocking));
            HIPCHK(e, hipEventCreateWithFlags(&ln.done, hipEventDisableTiming));
            if (int rc = dalloc(e, &ln.st, (size_t)1)) return rc;
            if (int rc = dalloc(e, &ln.xfrag, (size_t)kRowsMaxM * E)) return rc;
            if (int rc = dalloc(e, &ln.afrag, (size_t)kRowsMaxM * E)) return rc;
            if (int rc = dalloc(e, &ln.hfrag, (size_t)kRowsMaxM * 4 * E)) return rc;
            HIPCHK(e, hipMemset(ln.xfrag, 0, (size_t)kRowsMaxM * E * 4));
            HIPCHK(e, hipMemset(ln.afrag, 0, (size_t)kRowsMaxM * E * 4));
            HIPCHK(e, hipMemset(ln.hfrag, 0, (size_t)kRowsMaxM * 4 * E * 4));
        }
        HIPCHK(e, hipEventCreateWithFlags(&e->ev_lane_fork, hipEventDisableTiming));
    }
    if (int rc = dalloc(e, &e->logits, 3 * Bm * 8192)) return rc;
    if (int rc = dalloc(e, &e->logits_tar, Bm * kNBox * (size_t)cfg->bbox3d_vocab)) return rc;
    e->kv_scene_stride = (long)e->Lmax * 2 * E;
    e->kv_layer_stride = (long)Bm * e->kv_scene_stride;
    if (int rc = dev_alloc(e, &e->kvcache, (size_t)cfg->n_oar_layer * e->kv_layer_stride * e->tsz)) return rc;
    // (the engines' key loops request whole 16-key passes and mask the keys past the step: p = 0 times whatever bits lie there must be 0, not NaN)
    HIPCHK(e, hipMemset(e->kvcache, 0, (size_t)cfg->n_oar_layer * e->kv_layer_stride * e->tsz));
    if ((UMGEN_ENG_MFMA & 16) && e->tsz == 2) {
        // the matrix-core attention multiplies whole 32-key tiles, masked keys included: no NaN bit patterns may sit behind the mask
        HIPCHK(e, hipMemset(e->kvcache, 0, (size_t)cfg->n_oar_layer * e->kv_layer_stride * e->tsz));
        e->vt_scene_stride = (long)e->H * kHeadDim * e->Lmax;
        e->vt_layer_stride = (long)Bm * e->vt_scene_stride;
        if (int rc = dev_alloc(e, &e->vtcache, (size_t)cfg->n_oar_layer * e->vt_layer_stride * e->tsz)) return rc;
        HIPCHK(e, hipMemset(e->vtcache, 0, (size_t)cfg->n_oar_layer * e->vt_layer_stride * e->tsz));
    }
    // slot caches of the overlapped TAR pass: k | v rows of every temporal sub-block, all history slots (the foreground's growing-window
    // reuse allocates the same caches on first use, run_frame)
    if (e->overlap && !ensure_tcache(e)) e->overlap = false;     // keep the plain path rather than crowding the KV caches out
    if (const char* gc = getenv("UMGEN_GROW_CACHE")) e->grow_cache = gc[0] != '0';
    if (int rc = dalloc(e, &e->d_pose, Bm * Tm * 3)) return rc;
    if (int rc = dalloc(e, &e->d_pose_shift, Bm * Tm * 3)) return rc;
    if (int rc = dalloc(e, &e->d_map, Bm * Tm * kNMap)) return rc;
    if (int rc = dalloc(e, &e->d_box, Bm * Tm * kNBox)) return rc;
    if (int rc = dalloc(e, &e->d_img, Bm * Tm * kNImg)) return rc;
    if (int rc = dalloc(e, &e->d_tokens, Bm * kTokPerFrame)) return rc;
    if (int rc = dalloc(e, &e->d_prev_box, Bm * kNBox)) return rc;
    if (int rc = dalloc(e, &e->d_forced, Bm * kTokPerFrame)) return rc;
    if (int rc = dalloc(e, &e->d_counters, (size_t)8)) return rc;
    if (int rc = dalloc(e, &e->d_nboxes, Bm)) return rc;
    if (int rc = dalloc(e, &e->d_ego_tok, Bm * 3)) return rc;
    if (int rc = dalloc(e, &e->d_control, Bm * kSlots)) return rc;
    if (int rc = dalloc(e, &e->d_boxes, Bm * 64 * 10)) return rc;
    if (int rc = dalloc(e, &e->d_seeds, Bm)) return rc;
    if (int rc = dalloc(e, &e->d_state, (size_t)1)) return rc;
    // ---- XCD-resident decode engine (UMGEN_DECODE_ENGINE=0 keeps the five-launch decode layer) ----
    if (e->eng_enabled) {
        if (int rc = dalloc(e, &e->d_layers, (size_t)cfg->n_oar_layer)) return rc;
        std::vector<OarLayerDev> hl(cfg->n_oar_layer);
        for (int i = 0; i < cfg->n_oar_layer; ++i) {
            const SubW& w = e->oar[i];
            hl[i] = OarLayerDev{reinterpret_cast<const bf16_t*>(w.attn.Wqkv), reinterpret_cast<const bf16_t*>(w.attn.Wo),
                                reinterpret_cast<const bf16_t*>(w.mlp.Wfc), reinterpret_cast<const bf16_t*>(w.mlp.Wproj), nullptr,
                                w.attn.bqkv, w.attn.bo, w.ln_a, w.ln_b};
        }
        HIPCHK(e, hipMemcpy(e->d_layers, hl.data(), hl.size() * sizeof(OarLayerDev), hipMemcpyHostToDevice));
        e->eng_gloc_bytes = (size_t)16 * kEngLocStride * 8;
        if (int rc = dalloc(e, &e->eng_gx, Bm * kEngE)) return rc;
        if (int rc = dev_alloc(e, reinterpret_cast<void**>(&e->eng_gloc), e->eng_gloc_bytes)) return rc;
        if (int rc = dalloc(e, &e->eng_ticket, (size_t)16)) return rc;
        if (int rc = dalloc(e, &e->eng_err, (size_t)4)) return rc;
        HIPCHK(e, hipMemset(e->eng_ticket, 0, 64));
        if (getenv("UMGEN_DEBUG_TIMING") && !e->bg_engine) {      // (per-phase stamps: the instantiation without background workers -- UMGEN_BG_ENGINE=0 for a one-scene engine)
            if (int rc = dalloc(e, &e->eng_stamps, (size_t)16)) return rc;
            HIPCHK(e, hipMemset(e->eng_stamps, 0, 128));
        }
        HIPCHK(e, hipMemset(e->eng_gx, 0, Bm * kEngE * 8));
        HIPCHK(e, hipMemset(e->eng_gloc, 0, e->eng_gloc_bytes));
        HIPCHK(e, hipMemset(e->eng_err, 0, 16));
        if (const char* bs = getenv("UMGEN_DEBUG_BURN")) {
            int us = 0, mf = 0, sl = 0, kb = 0;
            if (sscanf(bs, "%d,%d,%d,%d", &us, &mf, &sl, &kb) == 4 && kb > 0) {
                if (int rc = dev_alloc(e, &e->burn_buf, (size_t)kb << 10)) return rc;
                HIPCHK(e, hipMemset(e->burn_buf, 1, (size_t)kb << 10));
            }
        }
        if (getenv("UMGEN_DEBUG_TIMING")) fprintf(stderr, "[umgen] decode engine: on (8 XCD groups)\n");
    }
    if (e->wide_enabled) {
        if (int rc = dalloc(e, &e->d_layers_wide, (size_t)cfg->n_oar_layer)) return rc;
        if (int rc = dalloc(e, &e->wide_gran, oar_engine_wide_granules())) return rc;
        if (int rc = dalloc(e, &e->wide_ticket, (size_t)16)) return rc;      // one arrival counter per XCD
        if (int rc = dalloc(e, &e->wide_err, (size_t)4)) return rc;
        HIPCHK(e, hipMemset(e->wide_gran, 0, oar_engine_wide_granules() * 8));
        HIPCHK(e, hipMemset(e->wide_ticket, 0, 64));
        HIPCHK(e, hipMemset(e->wide_err, 0, 16));
        if (getenv("UMGEN_DEBUG_TIMING")) {
            if (int rc = dalloc(e, &e->wide_stamps, (size_t)16)) return rc;
            HIPCHK(e, hipMemset(e->wide_stamps, 0, 128));
        }
    }
    return UMGEN_OK;
}

int umgen_load_tensor(umgen_engine* e, const char* key, const void* data, int32_t dtype, const int64_t* shape, int32_t ndim) {
    if (!e || !key || !data) return UMGEN_E_INVALID;
    auto it = e->slots.find(key);
    if (it == e->slots.end()) return 1;   // not consumed by the rollout (e.g. head_tar_pose, *.scale buffers)
    Slot& s = it->second;
    if ((size_t)ndim != s.shape.size()) return e->fail(UMGEN_E_INVALID, "%s: ndim %d, expected %zu", key, ndim, s.shape.size());
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        if (shape[i] != s.shape[i]) return e->fail(UMGEN_E_INVALID, "%s: dim %d is %lld, expected %lld", key, i, (long long)shape[i], (long long)s.shape[i]);
        n *= (size_t)shape[i];
    }
    if (dtype < 0 || dtype > UMGEN_DT_F64) return e->fail(UMGEN_E_INVALID, "%s: dtype %d", key, dtype);
    const bool to_bf16 = (s.kind == 2) || (s.kind == 1 && e->cfg.precision == UMGEN_PREC_BF16);
    if (s.kind == 1 && e->cfg.precision == UMGEN_PREC_FP16) {   // round-to-nearest-even to IEEE half, like torch's .half()
        std::vector<f16_t> h(n);
        if (dtype == UMGEN_DT_F16) memcpy(h.data(), data, n * 2);
        else {
            unsigned overflow = 0;                              // (no early exit in the conversion loop: it stays vectorisable)
            for (size_t i = 0; i < n; ++i) {
                const float v = load_as_f32(data, dtype, i);
                const uint16_t hb = f32_to_f16_bits_host(v);
                // the CONVERTED value decides, like torch's .half(): (65504, 65520) still rounds to 65504, only >= 65520 becomes inf
                overflow |= (unsigned)(((hb & 0x7fffu) == 0x7c00u) & (std::fabs(v) <= 3.4e38f));
                memcpy(&h[i], &hb, 2);
            }
            if (overflow) {                                     // a finite weight would become inf: refuse rather than decode garbage
                for (size_t i = 0; i < n; ++i) {
                    const float v = load_as_f32(data, dtype, i);
                    if (std::isfinite(v) && (f32_to_f16_bits_host(v) & 0x7fffu) == 0x7c00u)
                        return e->fail(UMGEN_E_INVALID, "%s[%zu] = %g does not fit fp16 (precision fp16 needs |w| < 65520)", key, i, (double)v);
                }
            }
        }
        HIPCHK(e, hipMemcpy(s.dst, h.data(), n * 2, hipMemcpyHostToDevice));
    } else if (to_bf16) {
        std::vector<bf16_t> h(n);
        if (dtype == UMGEN_DT_BF16) memcpy(h.data(), data, n * 2);
        else for (size_t i = 0; i < n; ++i) h[i] = f32_to_bf16(load_as_f32(data, dtype, i));
        HIPCHK(e, hipMemcpy(s.dst, h.data(), n * 2, hipMemcpyHostToDevice));
    } else {
        if (dtype == UMGEN_DT_F32) {
            HIPCHK(e, hipMemcpy(s.dst, data, n * 4, hipMemcpyHostToDevice));
        } else {
            std::vector<float> h(n);
            for (size_t i = 0; i < n; ++i) h[i] = load_as_f32(data, dtype, i);
            HIPCHK(e, hipMemcpy(s.dst, h.data(), n * 4, hipMemcpyHostToDevice));
        }
    }
    s.loaded = true;
    e->finalized = false;
    return UMGEN_OK;
}

int umgen_finalize_weights(umgen_engine* e) {
    if (!e) return UMGEN_E_INVALID;
    std::string missing;
    int nmiss = 0;
    for (auto& kv : e->slots)
        if (!kv.second.loaded && !kv.second.optional) {
            if (nmiss < 4) missing += (nmiss ? ", " : "") + kv.first;
            ++nmiss;
        }
    if (nmiss) return e->fail(UMGEN_E_STATE, "%d state-dict entries not loaded (e.g. %s)", nmiss, missing.c_str());
    const int E = e->E;
    std::vector<bf16_t> posi;
    const bool have_posi = e->slots["bbox3d_spatial_posi"].loaded;
    if (!e->slots["fouier_pe"].loaded) {
        std::vector<bf16_t> t;
        sinusoid_table(1024, E, 0, t);
        HIPCHK(e, hipMemcpy(const_cast<bf16_t*>(e->tb.fouier_pe), t.data(), t.size() * 2, hipMemcpyHostToDevice));
    }
    if (!have_posi) {
        sinusoid_table(1030, E, 1024, posi);
        HIPCHK(e, hipMemcpy(const_cast<bf16_t*>(e->tb.posi), posi.data(), posi.size() * 2, hipMemcpyHostToDevice));
    } else {
        posi.resize((size_t)1030 * E);
        HIPCHK(e, hipMemcpy(posi.data(), e->tb.posi, posi.size() * 2, hipMemcpyDeviceToHost));
    }
    if (!e->slots["grid_center_posi_embedding"].loaded) {
        // UMGen.py:140-153, 357-383: token of grid centre c = 62 - 4g is np.digitize((c + 64)/128, linspace(0,1,1024))
        std::vector<bf16_t> gp((size_t)1024 * E);
        int tok[32];
        for (int g = 0; g < 32; ++g) {
            const double x = ((double)(62 - 4 * g) + 64.0) / 128.0;
            int c = 0;
            for (int i = 0; i < 1024; ++i) if (lin_bin(i, 0.0, 1.0, 1024) <= x) ++c;
            tok[g] = c;
        }
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j)
                for (int c = 0; c < E; ++c)
                    gp[((size_t)i * 32 + j) * E + c] = f32_to_bf16(bf16_to_f32(posi[(size_t)tok[i] * E + c]) + bf16_to_f32(posi[(size_t)tok[j] * E + c]));
        HIPCHK(e, hipMemcpy(const_cast<bf16_t*>(e->tb.grid_posi), gp.data(), gp.size() * 2, hipMemcpyHostToDevice));
    }
    const int rc = e->cfg.precision == UMGEN_PREC_BF16 ? build_tables<bf16_t>(e)
                 : e->cfg.precision == UMGEN_PREC_FP16 ? build_tables<f16_t>(e) : build_tables<float>(e);
    if (rc) return rc;
    if (e->eng_enabled) { if (int rc2 = repack_mlp_proj(e)) return rc2; }
    if (e->wide_enabled) { if (int rc2 = repack_wide(e)) return rc2; }
    e->px.valid = false;   // slot caches filled with other weights are not a prefix of anything
    e->finalized = true;
    return UMGEN_OK;
}

int umgen_set_profiling(umgen_engine* e, int32_t enable) {
    if (!e) return UMGEN_E_INVALID;
    e->profiling = enable != 0;
    return UMGEN_OK;
}
int umgen_get_timings(umgen_engine* e, umgen_timings* out) {
    if (!e || !out) return UMGEN_E_INVALID;
    *out = e->tm;
    return UMGEN_OK;
}

static int check_sampling(umgen_engine* e, const umgen_sampling* s) {
    if (!s) return e->fail(UMGEN_E_INVALID, "sampling is null");
    if (s->method != UMGEN_SAMPLE_TOPK && s->method != UMGEN_SAMPLE_TOPP) return e->fail(UMGEN_E_INVALID, "sample method %d", s->method);
    if (s->method == UMGEN_SAMPLE_TOPP && !(s->p > 0.f && s->p_map > 0.f)) return e->fail(UMGEN_E_INVALID, "top-p mass must be > 0");
    if (s->top_k < 1 || s->top_k > 16 || s->top_k_map < 1 || s->top_k_map > 16 || s->topk_image < 1 || s->topk_image > 16)
        return e->fail(UMGEN_E_INVALID, "top-k values must be in [1, 16] (reference: 5 / 5 / 16)");
    if (!(s->temperature > 0.f)) return e->fail(UMGEN_E_INVALID, "temperature must be > 0");
    return 0;
}

// Every token that becomes a gather index is checked at the ABI boundary (history: embed_stack_kernel's table rows; control
// pose: the fouier_pe rows and decode_pose_value; control bbox3d: -1 = "free", anything else a be / posi row).
static int check_tokens(umgen_engine* e, const char* what, const int64_t* p, size_t n, int64_t vocab, bool allow_free) {
    for (size_t i = 0; i < n; ++i) {
        const int64_t v = p[i];
        if (v >= 0 && v < vocab) continue;
        if (allow_free && v == -1) continue;
        return e->fail(UMGEN_E_INVALID, "%s token %lld at flat index %zu is outside [0, %lld)%s", what, (long long)v, i, (long long)vocab,
                       allow_free ? " (and is not -1 = free)" : "");
    }
    return 0;
}
static int check_scene_tokens(umgen_engine* e, size_t frames, const int64_t* pose, const int64_t* map, const int64_t* bbox3d, const int64_t* image) {
    if (int rc = check_tokens(e, "pose", pose, frames * kNPose, e->cfg.pose_vocab, false)) return rc;
    if (int rc = check_tokens(e, "map", map, frames * kNMap, e->cfg.map_vocab, false)) return rc;
    if (int rc = check_tokens(e, "bbox3d", bbox3d, frames * kNBox, e->cfg.bbox3d_vocab, false)) return rc;
    if (int rc = check_tokens(e, "image", image, frames * kNImg, e->cfg.img_vocab, false)) return rc;
    // the x / y attribute tokens of a slot index the 1030-row spatial table; attribute tokens are bins (< 1024) or pad
    return 0;
}

int umgen_frame(umgen_engine* e, int32_t T, const int64_t* pose, const int64_t* map, const int64_t* bbox3d, const int64_t* image,
                const int64_t* ctrl_pose, const int64_t* ctrl_bbox3d, int32_t control_test, const umgen_sampling* sampling,
                int32_t frame_idx, const umgen_trace* trace, int64_t* out_pose, int64_t* out_map, int64_t* out_bbox3d, int64_t* out_image) {
    if (!e) return UMGEN_E_INVALID;
    if (!e->finalized) return e->fail(UMGEN_E_STATE, "umgen_finalize_weights has not been called");
    if (T < 1 || T > e->cfg.max_cond_frames) return e->fail(UMGEN_E_INVALID, "T=%d out of range [1,%d]", T, e->cfg.max_cond_frames);
    if (int rc = check_sampling(e, sampling)) return rc;
    if (!pose || !map || !bbox3d || !image || !out_pose || !out_map || !out_bbox3d || !out_image) return e->fail(UMGEN_E_INVALID, "null token buffer");
    if (int rc = check_scene_tokens(e, (size_t)T, pose, map, bbox3d, image)) return rc;
    if (ctrl_pose) { if (int rc = check_tokens(e, "control pose", ctrl_pose, 3, e->cfg.pose_vocab, false)) return rc; }
    if (ctrl_bbox3d) { if (int rc = check_tokens(e, "control bbox3d", ctrl_bbox3d, kNBox, e->cfg.bbox3d_vocab, true)) return rc; }
    if (ctrl_bbox3d && !control_test) return e->fail(UMGEN_E_UNSUPPORTED, "init_tokens['bbox3d'] without control_test is not a supported reference path");
    std::vector<int> p((size_t)T * 3), m((size_t)T * kNMap), bx((size_t)T * kNBox), im((size_t)T * kNImg);
    for (size_t i = 0; i < p.size(); ++i) p[i] = (int)pose[i];
    for (size_t i = 0; i < m.size(); ++i) m[i] = (int)map[i];
    for (size_t i = 0; i < bx.size(); ++i) bx[i] = (int)bbox3d[i];
    for (size_t i = 0; i < im.size(); ++i) im[i] = (int)image[i];
    std::vector<int> cp;
    std::vector<unsigned char> cs;
    if (ctrl_pose) { cp.resize(3); for (int i = 0; i < 3; ++i) cp[i] = (int)ctrl_pose[i]; }
    if (ctrl_bbox3d && control_test) {   // UMGen.py:1458-1473
        cs.assign(kSlots, 0);
        for (int i = 0; i < kNBox; ++i)
            if (ctrl_bbox3d[i] != -1) { bx[(size_t)(T - 1) * kNBox + i] = (int)ctrl_bbox3d[i]; cs[i / kSlotLen] = 1; }
    }
    std::vector<int> out(kTokPerFrame);
    FrameIO io{1, T, p.data(), m.data(), bx.data(), im.data(), ctrl_pose ? cp.data() : nullptr, cs.empty() ? nullptr : cs.data(),
               frame_idx, sampling, trace, out.data()};
    // the frame's GIVEN tokens (umgen_rollout's given_* for one frame: the predefined-token prefix of infer_oar_net, UMGen.py:1184-1201)
    std::vector<int> gm, gb;
    if (trace && trace->given_bbox3d && !trace->given_map)
        return e->fail(UMGEN_E_UNSUPPORTED, "given bbox3d tokens without a given map: the reference would put them on the map's positions (UMGen.py:1190-1201)");
    if (trace && trace->given_map) {
        if (control_test) return e->fail(UMGEN_E_UNSUPPORTED, "given tokens and control_test exclude each other");
        if (int rc = check_tokens(e, "given map", trace->given_map, kNMap, e->cfg.map_vocab, false)) return rc;
        gm.resize(kNMap);
        for (int i = 0; i < kNMap; ++i) gm[i] = (int)trace->given_map[i];
        io.given_map = gm.data();
        if (trace->given_bbox3d) {
            if (int rc = check_tokens(e, "given bbox3d", trace->given_bbox3d, kNBox, e->cfg.bbox3d_vocab, false)) return rc;
            gb.resize(kNBox);
            for (int i = 0; i < kNBox; ++i) gb[i] = (int)trace->given_bbox3d[i];
            io.given_box = gb.data();
        }
    }
    if (int rc = run_frame_any(e, io)) return rc;
    for (int i = 0; i < kNPose; ++i) out_pose[i] = out[i];
    for (int i = 0; i < kNMap; ++i) out_map[i] = out[kOffMap + i];
    for (int i = 0; i < kNBox; ++i) out_bbox3d[i] = out[kOffBox + i];
    for (int i = 0; i < kNImg; ++i) out_image[i] = out[kOffImg + i];
    return UMGEN_OK;
}

// UMGen.inference (UMGen.py:1542-1671)
int umgen_rollout(umgen_engine* e, int32_t B, int32_t T_in, int32_t new_frames, int32_t cond_frames, const int64_t* pose,
                  const int64_t* map, const int64_t* bbox3d, const int64_t* image, int32_t T_ctl, const int64_t* ctrl_pose,
                  const int64_t* ctrl_bbox3d, int32_t control_test, const int64_t* given_map, const int64_t* given_bbox3d,
                  const umgen_sampling* sampling, int64_t* out_pose, int64_t* out_map, int64_t* out_bbox3d, int64_t* out_image) {
    if (!e) return UMGEN_E_INVALID;
    if (!e->finalized) return e->fail(UMGEN_E_STATE, "umgen_finalize_weights has not been called");
    if (B < 1 || B > e->cfg.max_batch) return e->fail(UMGEN_E_INVALID, "B=%d out of range [1,%d]", B, e->cfg.max_batch);
    if (T_in < 1 || new_frames < 0 || cond_frames < 1 || cond_frames > e->cfg.max_cond_frames)
        return e->fail(UMGEN_E_INVALID, "T_in=%d new_frames=%d cond_frames=%d (max %d)", T_in, new_frames, cond_frames, e->cfg.max_cond_frames);
    if (int rc = check_sampling(e, sampling)) return rc;
    if (!pose || !map || !bbox3d || !image || !out_pose || !out_map || !out_bbox3d || !out_image) return e->fail(UMGEN_E_INVALID, "null token buffer");
    if (int rc = check_scene_tokens(e, (size_t)B * T_in, pose, map, bbox3d, image)) return rc;
    if ((ctrl_pose || ctrl_bbox3d) && T_ctl < 1) return e->fail(UMGEN_E_INVALID, "control tokens given with T_ctl=%d", T_ctl);
    if (ctrl_pose) { if (int rc = check_tokens(e, "control pose", ctrl_pose, (size_t)B * T_ctl * 3, e->cfg.pose_vocab, false)) return rc; }
    if (ctrl_bbox3d) { if (int rc = check_tokens(e, "control bbox3d", ctrl_bbox3d, (size_t)B * T_ctl * kNBox, e->cfg.bbox3d_vocab, true)) return rc; }
    // given (not generated) modalities of the new frames: infer_oar_net's predefined-token prefix (UMGen.py:1184-1201)
    if ((given_map || given_bbox3d) && T_ctl < 1) return e->fail(UMGEN_E_INVALID, "given tokens with T_ctl=%d", T_ctl);
    if (given_bbox3d && !given_map)
        return e->fail(UMGEN_E_UNSUPPORTED, "init_tokens['bbox3d'] without init_tokens['map'] (and without control_test): the reference concatenates the given "
                                            "modalities back to back behind the pose, so the boxes would sit on the map's positions -- not a meaningful path");
    if (given_bbox3d && (ctrl_bbox3d || control_test)) return e->fail(UMGEN_E_INVALID, "given bbox3d tokens and bbox3d control exclude each other");
    if (given_map) { if (int rc = check_tokens(e, "given map", given_map, (size_t)B * T_ctl * kNMap, e->cfg.map_vocab, false)) return rc; }
    if (given_bbox3d) { if (int rc = check_tokens(e, "given bbox3d", given_bbox3d, (size_t)B * T_ctl * kNBox, e->cfg.bbox3d_vocab, false)) return rc; }
    e->tm = umgen_timings{};
    e->overlap_suspended = false;
    const int T_out = T_in + new_frames;
    const int S[4] = {kNPose, kNMap, kNBox, kNImg};
    const int64_t* in[4] = {pose, map, bbox3d, image};
    int64_t* out[4] = {out_pose, out_map, out_bbox3d, out_image};
    // history: out_tokens and cond_tokens both start as the first T_in frames (UMGen.py:1581-1595)
    std::vector<int> hist[4];   // [B][T_cur][S] "cond_tokens" window (control mode mutates its last bbox3d frame in place)
    for (int m = 0; m < 4; ++m) {
        hist[m].resize((size_t)B * T_in * S[m]);
        for (int b = 0; b < B; ++b)
            for (int t = 0; t < T_in; ++t)
                for (int i = 0; i < S[m]; ++i) {
                    const int64_t v = in[m][((size_t)b * T_in + t) * S[m] + i];
                    hist[m][((size_t)b * T_in + t) * S[m] + i] = (int)v;
                    out[m][((size_t)b * T_out + t) * S[m] + i] = v;
                }
    }
    int T_cur = T_in;
    // Control tokens (UMGen.py:1605-1619, 1438-1473).  With pose tokens the rollout leaves control mode for good once they are used
    // up (init_tokens = None, control_test = False); bbox3d tokens alone (agents controlled, ego inferred by the ego net) simply stop
    // applying after their last frame (get_mod_tokens returns None past the end).
    bool have_ctl = (ctrl_pose != nullptr) && T_ctl > 0;
    bool have_box = (ctrl_bbox3d != nullptr) && T_ctl > 0;
    bool have_given = (given_map != nullptr) && T_ctl > 0;   // like every init_tokens entry: None past its last frame (get_mod_tokens), and gone
                                                              // for good with the pose tokens (UMGen.py:1613-1619)
    std::vector<int> frame_out((size_t)B * kTokPerFrame);
    for (int idx = 0; idx < new_frames; ++idx) {
        if (T_cur > cond_frames) {   // sliding window (UMGen.py:1600-1603)
            for (int m = 0; m < 4; ++m) {
                std::vector<int> nw((size_t)B * cond_frames * S[m]);
                for (int b = 0; b < B; ++b)
                    memcpy(&nw[(size_t)b * cond_frames * S[m]], &hist[m][((size_t)b * T_cur + (T_cur - cond_frames)) * S[m]],
                           (size_t)cond_frames * S[m] * sizeof(int));
                hist[m].swap(nw);
            }
            T_cur = cond_frames;
        }
        if (have_ctl && idx >= T_ctl) { have_ctl = false; have_box = false; have_given = false; control_test = 0; }   // control tokens exhausted (UMGen.py:1613-1619)
        if (have_box && idx >= T_ctl) have_box = false;
        if (have_given && idx >= T_ctl) have_given = false;
        std::vector<int> gm, gb;
        if (have_given) {
            gm.resize((size_t)B * kNMap);
            for (int b = 0; b < B; ++b)
                for (int i = 0; i < kNMap; ++i) gm[(size_t)b * kNMap + i] = (int)given_map[((size_t)b * T_ctl + idx) * kNMap + i];
            if (given_bbox3d) {
                gb.resize((size_t)B * kNBox);
                for (int b = 0; b < B; ++b)
                    for (int i = 0; i < kNBox; ++i) gb[(size_t)b * kNBox + i] = (int)given_bbox3d[((size_t)b * T_ctl + idx) * kNBox + i];
            }
        }
        std::vector<int> cp;
        std::vector<unsigned char> cs;
        if (have_ctl) {
            cp.resize((size_t)B * 3);
            for (int b = 0; b < B; ++b)
                for (int a = 0; a < 3; ++a) cp[b * 3 + a] = (int)ctrl_pose[((size_t)b * T_ctl + idx) * 3 + a];
        }
        if (have_box) {
            if (control_test) {
                cs.assign((size_t)B * kSlots, 0);
                for (int b = 0; b < B; ++b)
                    for (int i = 0; i < kNBox; ++i) {
                        const int64_t v = ctrl_bbox3d[((size_t)b * T_ctl + idx) * kNBox + i];
                        if (v != -1) { hist[2][((size_t)b * T_cur + (T_cur - 1)) * kNBox + i] = (int)v; cs[(size_t)b * kSlots + i / kSlotLen] = 1; }
                    }
            } else {
                return e->fail(UMGEN_E_UNSUPPORTED, "init_tokens['bbox3d'] without control_test is not a supported reference path");
            }
        }
        FrameIO io{B, T_cur, hist[0].data(), hist[1].data(), hist[2].data(), hist[3].data(), have_ctl ? cp.data() : nullptr,
                   cs.empty() ? nullptr : cs.data(), idx, sampling, nullptr, frame_out.data()};
        io.cond_cap = cond_frames;
        io.next_follows = idx + 1 < new_frames;
        io.next_has_ctrl_pose = have_ctl && idx + 1 < T_ctl;
        io.given_map = gm.empty() ? nullptr : gm.data();
        io.given_box = gb.empty() ? nullptr : gb.data();
        if (int rc = run_frame_any(e, io)) return rc;
        // append (UMGen.py:1636-1666): control pose tokens are copied verbatim; everything else is what was generated
        const int off[4] = {0, kOffMap, kOffBox, kOffImg};
        for (int m = 0; m < 4; ++m) {
            std::vector<int> nw((size_t)B * (T_cur + 1) * S[m]);
            for (int b = 0; b < B; ++b) {
                memcpy(&nw[(size_t)b * (T_cur + 1) * S[m]], &hist[m][(size_t)b * T_cur * S[m]], (size_t)T_cur * S[m] * sizeof(int));
                for (int i = 0; i < S[m]; ++i) {
                    const int v = frame_out[(size_t)b * kTokPerFrame + off[m] + i];
                    nw[((size_t)b * (T_cur + 1) + T_cur) * S[m] + i] = v;
                    out[m][((size_t)b * T_out + T_in + idx) * S[m] + i] = v;
                }
            }
            hist[m].swap(nw);
        }
        T_cur += 1;
    }
    return UMGEN_OK;
}

// Test hook: ONE decode step through the BlockOAR layers (no head, no sampler) on caller-provided inputs, either as the five-launch
// layer form or through the decode engine.  The K/V rows of position L are appended to the cache, so a test drives L = 0, 1, 2, ...
int umgen_dbg_oar_step(umgen_engine* e, int32_t B, int32_t L, const float* x_in, float* x_out, int32_t use_engine, int32_t unmasked) {
    if (!e || !x_in || !x_out) return UMGEN_E_INVALID;
    if (!e->finalized) return e->fail(UMGEN_E_STATE, "umgen_finalize_weights has not been called");
    if (e->cfg.precision == UMGEN_PREC_FP32) return e->fail(UMGEN_E_UNSUPPORTED, "16-bit engines only");
    if (B < 1 || B > e->cfg.max_batch || L < 0 || L >= e->Lmax) return e->fail(UMGEN_E_INVALID, "B=%d L=%d", B, L);
    if (use_engine != 0 && use_engine != 1 && use_engine != 3) return e->fail(UMGEN_E_INVALID, "use_engine %d (0: five launches per layer, 1: XCD-resident engine, 3: chip-wide engine)", use_engine);
    if (use_engine == 1 && !e->eng_enabled) return e->fail(UMGEN_E_UNSUPPORTED, "decode engine not available on this engine");
    if (e->eng_epoch > 0xE0000000u) {   // same wrap rule as run_frame, for both engines' granule buffers
        HIPCHK(e, hipDeviceSynchronize());
        if (e->eng_enabled) {
            HIPCHK(e, hipMemset(e->eng_gx, 0, (size_t)e->cfg.max_batch * kEngE * 8));
            HIPCHK(e, hipMemset(e->eng_gloc, 0, e->eng_gloc_bytes));
        }
        if (e->wide_enabled) HIPCHK(e, hipMemset(e->wide_gran, 0, oar_engine_wide_granules() * 8));
        e->eng_epoch = 16u;
    }
    const unsigned epoch = e->eng_epoch;   // tags never repeat across calls
    e->eng_epoch += kEpochPerStep;
    hipStream_t const keep = e->stream;
    hipStream_t const st = (unmasked && e->full_stream) ? e->full_stream : e->stream;
    OarState s0{L, 0, 0, 0, 0, epoch, SamplerParams{}};
    HIPCHK(e, hipMemcpyAsync(e->d_state, &s0, sizeof(s0), hipMemcpyHostToDevice, st));
    HIPCHK(e, hipMemcpyAsync(e->xdec, x_in, (size_t)B * e->E * 4, hipMemcpyHostToDevice, st));
    const bool en = e->eng_enabled, wide_keep = e->wide_enabled;
    if (use_engine == 3 && !e->wide_enabled) return e->fail(UMGEN_E_UNSUPPORTED, "chip-wide decode engine not available on this engine");
    e->wide_enabled = wide_keep && use_engine == 3;
    e->eng_enabled = en && use_engine == 1;
    e->stream = st;
    const int lrc = e->cfg.precision == UMGEN_PREC_FP16 ? oar_layers<f16_t>(e, B, attn_nsplit(L + 1)) : oar_layers<bf16_t>(e, B, attn_nsplit(L + 1));
    e->stream = keep;
    e->eng_enabled = en;
    e->wide_enabled = wide_keep;
    if (lrc) return lrc;
    HIPCHK(e, hipMemcpyAsync(x_out, e->xdec, (size_t)B * e->E * 4, hipMemcpyDeviceToHost, st));
    unsigned eng_err = 0;
    if (use_engine) HIPCHK(e, hipMemcpyAsync(&eng_err, use_engine == 3 ? e->wide_err : e->eng_err, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HIPCHK(e, hipStreamSynchronize(st));
    if (eng_err) {      // clear the word of the engine that ran (the other engine's may not exist: ADVICE r5)
        (void)hipMemset(use_engine == 3 ? e->wide_err : e->eng_err, 0, sizeof(unsigned));
        return e->fail(UMGEN_E_HIP, "decode engine gave up waiting for hand-off tag 0x%08x", eng_err);
    }
    return UMGEN_OK;
}

int umgen_destroy(umgen_engine* e) {
    if (!e) return UMGEN_OK;
    (void)hipSetDevice(e->cfg.device);
    (void)hipDeviceSynchronize();   // every stream of this engine (decode, background, side, unmasked) is idle before anything is freed
    if (e->wide_stamps) {
        unsigned long long st[16];
        if (hipMemcpy(st, e->wide_stamps, 128, hipMemcpyDeviceToHost) == hipSuccess && st[15]) {
            const char* nm[14] = {"wait x", "LN + qkv rows", "wait qkv", "attention", "wait waves", "quarter out + wait quarters", "merge + wait att", "c_proj", "wait x'",
                                  "LN + c_fc + GELU", "wait waves", "mlp partial sums", "wait partial sums", "add partials"};
            fprintf(stderr, "[umgen] chip-wide decode engine, rank 0 wave 0, us per layer over %llu layers:", st[15]);
            double tot = 0;
            for (int p = 0; p < 14; ++p) { fprintf(stderr, " %s %.2f", nm[p], (double)st[p] / 100.0 / (double)st[15]); tot += (double)st[p] / 100.0 / (double)st[15]; }
            fprintf(stderr, " | total %.2f\n", tot);
        }
    }
    if (e->eng_stamps) {
        unsigned long long st[16];
        if (hipMemcpy(st, e->eng_stamps, 128, hipMemcpyDeviceToHost) == hipSuccess && st[10]) {
            const char* nm[10] = {"wait x", "qkv rows", "wait qkv", "attention", "wait partials", "c_proj", "wait x'", "c_fc + partial sums", "wait partial sums", "add partials"};
            fprintf(stderr, "[umgen] decode engine, group 0 rank 0, us per item over %llu items:", st[10]);
            double tot = 0;
            for (int p = 0; p < 10; ++p) { fprintf(stderr, " %s %.2f", nm[p], (double)st[p] / 100.0 / (double)st[10]); tot += (double)st[p] / 100.0 / (double)st[10]; }
            fprintf(stderr, " (c_fc part %.2f)", (double)st[11] / 100.0 / (double)st[10]);
            fprintf(stderr, " | total %.2f\n", tot);
            if (st[15])
                fprintf(stderr, "[umgen] decode engine, prologue of a launch (%llu launches): kernel entry -> rank, step, epoch known %.2f us; kernel entry -> first item's q|k|v + parked mlp rows there %.2f us"
                        "; kernel entry -> first item's x, LN weights and q|k|v rows there %.2f us\n", st[15], (double)st[12] / 100.0 / (double)st[15],
                        (double)st[13] / 100.0 / (double)st[15], (double)st[14] / 100.0 / (double)st[15]);
        }
    }
    for (auto& row : e->step_graph)
        for (auto& g : row)
            if (g) hipGraphExecDestroy(g);
    for (auto& ln : e->lane) {
        for (auto& row : ln.graph)
            for (auto& g : row)
                if (g) hipGraphExecDestroy(g);
        if (ln.s) { hipStreamSynchronize(ln.s); hipStreamDestroy(ln.s); }
        if (ln.done) hipEventDestroy(ln.done);
    }
    if (e->ev_lane_fork) hipEventDestroy(e->ev_lane_fork);
    for (void* p : e->allocs) hipFree(p);
    for (auto& ev : e->ev) if (ev) hipEventDestroy(ev);
    for (auto& pr : e->gemm_ev) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    for (auto& pr : e->attn_ev) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    for (auto& pr : e->layer_ev) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    if (e->tb.gmap) {}   // tables are in allocs
    if (e->bg_stream) { hipStreamSynchronize(e->bg_stream); hipStreamDestroy(e->bg_stream); }
    if (e->full_stream) hipStreamDestroy(e->full_stream);
    for (int i = 0; i < 2; ++i) {
        if (e->side_stream[i]) hipStreamDestroy(e->side_stream[i]);
        if (e->ev_side_done[i]) hipEventDestroy(e->ev_side_done[i]);
    }
    if (e->ev_side_in) hipEventDestroy(e->ev_side_in);
    for (hipEvent_t ev : {e->ev_tar_done, e->ev_bg_done, e->ev_bg0, e->ev_pre_done, e->ev_drain0, e->ev_drain1}) if (ev) hipEventDestroy(ev);
    if (e->stream) hipStreamDestroy(e->stream);
    delete e;
    return UMGEN_OK;
}

}  // extern "C"
