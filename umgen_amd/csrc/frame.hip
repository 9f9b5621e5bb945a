// Frame-level kernels: token embedding of the TAR/ego stacks, action-aware map warp, conditioning rows, and the
// per-token sampler with the bbox3d control flow (pad-avoid resample, control resample, rule-based constraint) that the
// reference runs on the host between decode steps (UMGen.py:1029-1139, 1275-1383) -- here it stays on the device so a
// frame's 2206 decode steps need no host round trip.
#include "bg_queue.h"
#include "frame_body.h"

namespace umgen {

__global__ __launch_bounds__(128) void embed_stack_kernel(int stack, EmbedTables tb, WindowTokens w, float* __restrict__ X,
                                                          float* __restrict__ mapfeat) {
    embed_stack_row(stack, tb, w, X, mapfeat, (long)blockIdx.x, (int)threadIdx.x);
}

__global__ __launch_bounds__(128) void warp_map_kernel(int stack, EmbedTables tb, int T, const float* __restrict__ mapfeat,
                                                       const float* __restrict__ pose_diff, float* __restrict__ X,
                                                       float* __restrict__ warped_last, int Tfull, int t0) {
    warp_map_cell(stack, tb, T, mapfeat, pose_diff, X, warped_last, Tfull, t0, (long)blockIdx.x, (int)threadIdx.x);
}

void launch_embed_stack(hipStream_t s, int stack, const EmbedTables& tb, const WindowTokens& w, float* X, float* mapfeat) {
    const long rows = (long)w.B * w.T * stack_len(stack);
    if (BgRecorder* rec = g_bg_rec) {      // (the tables are the engine's: BgQueue::tb)
        BgOp& o = rec->add(BG_EMBED, 0, (rows + 3) / 4, 128, 60);
        o.h.i0 = stack; o.h.i1 = w.B; o.h.i2 = w.T; o.h.i3 = w.Tfull; o.a.i4 = w.t0; o.a.l1 = rows;
        o.a.p0 = const_cast<int*>(w.pose); o.a.p1 = const_cast<int*>(w.map); o.a.p2 = const_cast<int*>(w.box); o.a.p3 = const_cast<int*>(w.img);
        o.a.p4 = X; o.a.p5 = mapfeat;
        return;
    }
    hipLaunchKernelGGL(embed_stack_kernel, dim3((unsigned)rows), dim3(128), 0, s, stack, tb, w, X, mapfeat);
}


void launch_warp_map(hipStream_t s, int stack, const EmbedTables& tb, int B, int T, const float* mapfeat, const float* pose_diff,
                     float* X, float* warped_last, int Tfull, int t0) {
    if (BgRecorder* rec = g_bg_rec) {
        const long cells = (long)B * T * kNMap;
        BgOp& o = rec->add(BG_WARP, 0, (cells + 3) / 4, 128, 80);
        o.h.i0 = stack; o.h.i1 = T; o.h.i2 = Tfull ? Tfull : T; o.h.i3 = t0; o.a.l1 = cells;
        o.a.p0 = const_cast<float*>(mapfeat); o.a.p1 = const_cast<float*>(pose_diff); o.a.p2 = X; o.a.p3 = warped_last;
        return;
    }
    hipLaunchKernelGGL(warp_map_kernel, dim3((unsigned)((long)B * T * kNMap)), dim3(128), 0, s, stack, tb, T, mapfeat, pose_diff, X,
                       warped_last, Tfull ? Tfull : T, t0);
}

// ---------------------------------------------------------------------------------------------------------
// conditioning rows of the OAR (UMGen.py:1496-1511, 1227-1231): last history frame of each stack after its final LN
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cond_rows_kernel(int stack, int T, int E, const float* __restrict__ X, const float* __restrict__ ln_w,
                                                        const float* __restrict__ warped_last, float* __restrict__ cond, int n_rows) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.y;
    if (r >= n_rows) return;
    const int lane = threadIdx.x & 63;
    int s;
    if (stack == STACK_MAP) s = kMapBos + r;
    else if (stack == STACK_BOX) s = kBoxBos + r;
    else s = (r < 5) ? r : (kImgBos + (r - 5));
    const int SS = stack_len(stack);
    const float* xr = X + (((long)b * T + (T - 1)) * SS + s) * E;
    float sum = 0.f;
    for (int c = lane; c < E; c += 64) sum += xr[c];
    const float mean = wave_sum(sum) / (float)E;
    float q = 0.f;
    for (int c = lane; c < E; c += 64) { const float d = xr[c] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)E + 1e-5f);
    float* o = cond + ((long)b * kSeq + s) * E;
    const float* wl = (stack == STACK_MAP && s >= kMapC0 && s < kMapEos) ? warped_last + ((long)b * kNMap + (s - kMapC0)) * E : nullptr;
    for (int c = lane; c < E; c += 64) {
        float v = (xr[c] - mean) * rstd * ln_w[c];
        if (wl) v += wl[c];
        o[c] = v;
    }
}

void launch_cond_rows(hipStream_t s, int stack, int B, int T, int E, const float* X, const float* ln_w, const float* warped_last,
                      float* cond) {
    const int n_rows = stack == STACK_MAP ? 1026 : (stack == STACK_BOX ? 662 : 5 + 514);
    hipLaunchKernelGGL(cond_rows_kernel, dim3((n_rows + 3) / 4, B), dim3(256), 0, s, stack, T, E, X, ln_w, warped_last, cond, n_rows);
}

__global__ void first_input_kernel(int E, const float* __restrict__ row, const float* __restrict__ cond, float* __restrict__ x) {
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < E; c += blockDim.x) x[(long)b * E + c] = row[c] + cond[(long)b * kSeq * E + c];
}
void launch_first_input(hipStream_t s, int B, int E, const float* tske_row, const float* cond, float* x) {
    hipLaunchKernelGGL(first_input_kernel, dim3(B), dim3(256), 0, s, E, tske_row, cond, x);
}

__global__ void ego_queries_kernel(EmbedTables tb, int T, float* __restrict__ x) {
    const int j = blockIdx.x % 3;
    const int E = tb.E;
    for (int c = threadIdx.x; c < E; c += blockDim.x)
        x[(long)blockIdx.x * E + c] = (tb.egoe[(long)j * E + c] + tb.spe[(long)j * E + c]) + tb.tpe[(long)(T - 1) * E + c];
}
void launch_ego_queries(hipStream_t s, const EmbedTables& tb, int B, int T, float* x) {
    hipLaunchKernelGGL(ego_queries_kernel, dim3(B * 3), dim3(256), 0, s, tb, T, x);
}

// Last kernel of a decode step: every block has read st->step before it arrives here; the last block to arrive advances it.
__device__ inline void finish_step(OarState* st) {
    __syncthreads();
    if (gridDim.x == 1) {   // one scene: no arrival count needed (saves a returning atomic at the end of every step)
        if (threadIdx.x == 0) { st->step += 1; st->epoch += kEpochPerStep; }
        return;
    }
    if (threadIdx.x == 0) {
        const int t = atomicAdd(&st->done, 1);
        if (t == (int)gridDim.x - 1) { st->done = 0; st->step += 1; st->epoch += kEpochPerStep; }
    }
}

// ---------------------------------------------------------------------------------------------------------
// sampler: topk (UMGen.py:899-913) + sfmx_temp_sampling (967-974); torch.multinomial is replaced by inverse-CDF
// sampling on the build's counter-based uniform (bit-for-bit the oracle's OracleUMGen.sample)
// ---------------------------------------------------------------------------------------------------------
constexpr int kMaxKept = 64;
constexpr int kMaxK = 16;   // top-k <= 16 (the reference uses 5 / 5 / 16): 4 waves x 16 candidates = one per lane of wave 0
struct SampleShared {
    float cand_v[4 * kMaxK];
    int cand_i[4 * kMaxK];
    float kth;
    int n_kept, n_cand;
    int kept_i[kMaxKept];
    float kept_v[kMaxKept];
    int result;
};

// all 256 threads call; returns the sampled index.  mask_idx (>=0) is treated as -inf.
// k-th largest value by threshold selection (round 3; was k arg-max rounds over every thread's 32 registers, ~230 instructions per
// round: 12 / 14 / 25 us per map / bbox3d / image token): T = k-th largest of the per-thread maxima, then the exact k-th largest among
// the few logits >= T; the exhaustive rounds remain as the fallback for > 64 such logits.  The k-th value -- and with it the kept set,
// its order and the draw -- is the same number either way.  The final softmax / inverse-CDF walk is sequential in ascending token
// index (bit-for-bit the oracle's order) but runs on register values fetched with v_readlane, not on LDS round trips.
__device__ int block_sample_topk(const float* __restrict__ logits, int V, int k, float temp, float u, int mask_idx, SampleShared& sh, int* overflow) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float v0[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int idx = tid + 256 * i;
        v0[i] = (idx < V && idx != mask_idx) ? logits[idx] : -INFINITY;
    }
    const int kk = min(min(k, V), kMaxK);
    if (tid < 4 * kMaxK) { sh.cand_v[tid] = -INFINITY; sh.cand_i[tid] = 0x7fffffff; }
    if (tid == 0) { sh.n_kept = 0; sh.n_cand = 0; }
    __syncthreads();
    // ---- the k-th largest logit, in two cheap selections instead of k arg-max rounds over every thread's 32 registers ----
    // (1) T = k-th largest of the 256 per-thread maxima.  k different logits are >= T, so the k-th largest logit is >= T and every
    //     logit of the top k is among the (few) logits >= T.  Per wave: k rounds of wave-max over ONE value per lane; wave 0 ranks
    //     the 4k wave candidates (ties between equal maxima of different threads are separate candidates: order by slot).
    {
        float cur = v0[0];
#pragma unroll
        for (int i = 1; i < 32; ++i) cur = fmaxf(cur, v0[i]);
        for (int it = 0; it < kk; ++it) {
            const float wv = wave_max(cur);
            const int first = __ffsll((unsigned long long)__ballot(cur == wv)) - 1;   // lowest lane holding the wave maximum
            if (lane == 0) { sh.cand_v[wave * kMaxK + it] = wv; sh.cand_i[wave * kMaxK + it] = wave * kMaxK + it; }
            if (lane == first) cur = -INFINITY;
        }
    }
    __syncthreads();
    // rank of each candidate of sh.cand_* (one per lane of wave 0; slots [16 w, 16 w + per_wave)) = number of candidates that precede
    // it (greater value, or equal value and lower index); the candidate of rank kk - 1 is written to sh.kth
    auto rank_candidates = [&](int per_wave) {
        const float cv = sh.cand_v[lane];
        const int ci = sh.cand_i[lane];
        int rank = 0;
        const int ncand = 4 * per_wave;
        for (int jj = 0, w4 = 0, it = 0; jj < ncand; ++jj) {
            const int j = w4 * kMaxK + it;
            const float ov = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cv), j));
            const int oi = __builtin_amdgcn_readlane(ci, j);
            rank += (ov > cv || (ov == cv && oi < ci)) ? 1 : 0;
            if (++it == per_wave) { it = 0; ++w4; }
        }
        if (rank == kk - 1) sh.kth = cv;   // exactly one lane (ranks are a permutation; the unused slots hold -inf / int-max and rank last)
    };
    if (wave == 0) rank_candidates(kk);
    __syncthreads();
    // (2) the logits >= T (k of them, plus whatever else a top thread holds above T): the exact k-th largest among them
    {
        const float T = sh.kth;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (v0[i] >= T && v0[i] > -INFINITY) {
                const int slot = atomicAdd(&sh.n_cand, 1);
                if (slot < kMaxKept) { sh.kept_i[slot] = tid + 256 * i; sh.kept_v[slot] = v0[i]; }
            }
        }
    }
    __syncthreads();
    const int n_cand = sh.n_cand;
    if (n_cand <= kMaxKept) {
        if (wave == 0) {
            const float cv = lane < n_cand ? sh.kept_v[lane] : -INFINITY;
            const int ci = lane < n_cand ? sh.kept_i[lane] : 0x7fffffff;
            int rank = 0;
            for (int j = 0; j < n_cand; ++j) {
                const float ov = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cv), j));
                const int oi = __builtin_amdgcn_readlane(ci, j);
                rank += (ov > cv || (ov == cv && oi < ci)) ? 1 : 0;
            }
            if (lane < n_cand && rank == kk - 1) sh.kth = cv;
        }
    } else {
        // more than 64 logits >= T (T = -inf with fewer than k finite logits, or massive ties): the exhaustive selection -- every wave
        // extracts its own top k by k rounds of register arg-max + two wave reductions, wave 0 ranks the 4k candidates
        float v[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = v0[i];
        if (tid < 4 * kMaxK) { sh.cand_v[tid] = -INFINITY; sh.cand_i[tid] = 0x7fffffff; }
        __syncthreads();
        for (int it = 0; it < kk; ++it) {
            float bv = -INFINITY;
            int bi = 0x7fffffff;
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int idx = tid + 256 * i;
                if (v[i] > bv) { bv = v[i]; bi = idx; }   // ascending idx within a thread => lowest index kept on ties
            }
            const float wv = wave_max(bv);
            const int wi = wave_min_i32(bv == wv ? bi : 0x7fffffff);   // lowest index among the lanes holding the wave maximum
            if (lane == 0) { sh.cand_v[wave * kMaxK + it] = wv; sh.cand_i[wave * kMaxK + it] = wi; }
            if (bi == wi) {
#pragma unroll
                for (int i = 0; i < 32; ++i)
                    if (tid + 256 * i == wi) v[i] = -INFINITY;
            }
        }
        __syncthreads();
        if (wave == 0) rank_candidates(kk);
    }
    __syncthreads();
    const float kth = sh.kth;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int idx = tid + 256 * i;
        const float l = v0[i];
        if (l >= kth && l > -INFINITY) {
            const int slot = atomicAdd(&sh.n_kept, 1);
            if (slot < kMaxKept) { sh.kept_i[slot] = idx; sh.kept_v[slot] = l; }
        }
    }
    __syncthreads();
    (void)overflow;
    if (sh.n_kept > kMaxKept) {
        // More logits tie with the k-th largest one than the kept-set buffer holds (a degenerate head: e.g. zero-initialised weights give V
        // equal logits).  The reference's topk keeps EVERY tie and samples among them (UMGen.py:899-913), so does this walk: one thread goes
        // over the logits in index order -- maximum, sequential fp32 softmax denominator, inverse-CDF -- with exactly the arithmetic of the
        // kept-set path below (and of the oracle).  Slow (three passes over V values by one lane) and never taken by a trained model.
        if (tid == 0) {
            float zmax = -INFINITY;
            for (int idx = 0; idx < V; ++idx) {
                const float l = idx != mask_idx ? logits[idx] : -INFINITY;
                if (l >= kth && l > -INFINITY) zmax = fmaxf(zmax, l / temp);
            }
            float total = 0.f;
            int last = 0;
            for (int idx = 0; idx < V; ++idx) {
                const float l = idx != mask_idx ? logits[idx] : -INFINITY;
                if (l >= kth && l > -INFINITY) { total = __fadd_rn(total, exp_det(__fsub_rn(l / temp, zmax))); last = idx; }
            }
            const float target = __fmul_rn(u, total);
            float c = 0.f;
            int res = last;
            for (int idx = 0; idx < V; ++idx) {
                const float l = idx != mask_idx ? logits[idx] : -INFINITY;
                if (l >= kth && l > -INFINITY) {
                    c = __fadd_rn(c, exp_det(__fsub_rn(l / temp, zmax)));
                    if (c > target) { res = idx; break; }
                }
            }
            sh.result = res;
        }
    } else if (wave == 0) {
        const int n = min(sh.n_kept, kMaxKept);
        // sort the kept set by token index: rank by uniform readlane loop, scatter through LDS, read back sorted
        const int ki = lane < n ? sh.kept_i[lane] : 0x7fffffff;
        const float kv = lane < n ? sh.kept_v[lane] : -INFINITY;
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += (__builtin_amdgcn_readlane(ki, j) < ki) ? 1 : 0;
        if (lane < n) { sh.cand_i[rank] = ki; sh.cand_v[rank] = kv; }
        __builtin_amdgcn_wave_barrier();
        const int si = lane < n ? sh.cand_i[lane] : 0;
        const float z = lane < n ? sh.cand_v[lane] / temp : -INFINITY;
        const float zmax = wave_max(z);
        const float ev = lane < n ? exp_det(__fsub_rn(z, zmax)) : 0.f;
        float total = 0.f;
        for (int a = 0; a < n; ++a) total = __fadd_rn(total, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ev), a)));
        const float target = __fmul_rn(u, total);
        float c = 0.f;
        int res = __builtin_amdgcn_readlane(si, n - 1);
        for (int a = 0; a < n; ++a) {
            c = __fadd_rn(c, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ev), a)));
            if (c > target) { res = __builtin_amdgcn_readlane(si, a); break; }
        }
        if (lane == 0) sh.result = res;
    }
    __syncthreads();
    const int r = sh.result;
    __syncthreads();
    return r;
}

// ---------------------------------------------------------------------------------------------------------
// sample_top_p (UMGen.py:915-965): softmax -> sort descending -> keep while (cumsum - p_j) <= p -> renormalise -> draw.
// Mirrors OracleUMGen.sample bit for bit: sequential fp32 sums (index order for the softmax denominator, sorted order for
// the nucleus), stable descending order (ties by ascending index) from a block-wide bitonic sort in LDS.
// Not the default path (evaluate.py resolves sample_method="topk"); ~0.1 ms per token.
// ---------------------------------------------------------------------------------------------------------
constexpr int kTopPMax = 8192;
struct TopPShared {
    float pr[kTopPMax];
    int ix[kTopPMax];
    float red[4];
    float total;
    int result;
};
__device__ int block_sample_topp(const float* __restrict__ logits, int V, float p, float temp, float u, int mask_idx, TopPShared& sh) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float mx = -INFINITY;
    for (int i = tid; i < kTopPMax; i += 256) {
        const float z = (i < V && i != mask_idx) ? logits[i] / temp : -INFINITY;
        sh.pr[i] = z;
        sh.ix[i] = i;
        mx = fmaxf(mx, z);
    }
    mx = wave_max(mx);
    if (lane == 0) sh.red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sh.red[0], sh.red[1]), fmaxf(sh.red[2], sh.red[3]));
    for (int i = tid; i < kTopPMax; i += 256) sh.pr[i] = exp_det(__fsub_rn(sh.pr[i], mx));   // exp(-inf) = 0 for padding / masked
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int i = 0; i < V; ++i) t = __fadd_rn(t, sh.pr[i]);
        sh.total = t;
    }
    __syncthreads();
    const float total = sh.total;
    for (int i = tid; i < kTopPMax; i += 256) sh.pr[i] = (i < V) ? sh.pr[i] / total : -1.f;   // padding sorts last
    __syncthreads();
    // bitonic sort, descending by probability, ascending index among equals
    for (int k2 = 2; k2 <= kTopPMax; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < kTopPMax; i += 256) {
                const int l2 = i ^ j;
                if (l2 > i) {
                    const float a = sh.pr[i], b2 = sh.pr[l2];
                    const int ia = sh.ix[i], ib = sh.ix[l2];
                    const bool a_first = (a > b2) || (a == b2 && ia < ib);   // a should precede b in the final order
                    const bool up = ((i & k2) == 0);
                    if (up ? !a_first : a_first) { sh.pr[i] = b2; sh.pr[l2] = a; sh.ix[i] = ib; sh.ix[l2] = ia; }
                }
            }
            __syncthreads();
        }
    }
    if (tid == 0) {
        float c = 0.f;
        int n = 0;
        while (n < V && !(c > p)) { c = __fadd_rn(c, sh.pr[n]); ++n; }      // (cumsum - p_j) > p masks entry j
        float t2 = 0.f;
        for (int i = 0; i < n; ++i) t2 = __fadd_rn(t2, sh.pr[i]);
        const float target = __fmul_rn(u, t2);
        float cc = 0.f;
        int res = sh.ix[n - 1];
        for (int i = 0; i < n; ++i) {
            cc = __fadd_rn(cc, sh.pr[i]);
            if (cc > target) { res = sh.ix[i]; break; }
        }
        sh.result = res;
    }
    __syncthreads();
    const int r = sh.result;
    __syncthreads();
    return r;
}

union SamplerLds {
    SampleShared k;
    TopPShared p;
};
// dispatch on the sampling method (UMGen.py:119-126): kparam is the top-k value, pparam the nucleus mass
__device__ inline int block_sample(const SamplerParams& sp, const float* logits, int V, int kparam, float pparam, float u, int mask_idx,
                                   SamplerLds& sh, int* overflow) {
    return sp.method == 0 ? block_sample_topk(logits, V, kparam, sp.temperature, u, mask_idx, sh.k, overflow)
                          : block_sample_topp(logits, V, pparam, sp.temperature, u, mask_idx, sh.p);
}

// ---------------------------------------------------------------------------------------------------------
// rule-based constraint (UMGen.py:1275-1383) and its helpers, evaluated by one thread
//   decode: BBox3DTokenizer.decode_single_objects (tokenizer.py:679-687) + Normalize.unnormalize_bbox3d (normalize.py:136-229)
//   collision: BoxOverlap.check_collision(fliter=True) (misc.py:591-630), bbox3d2bevcorners (143-177), box_collision_test (203-311)
// fp64 / fp32 operations are issued unfused (__dmul_rn, __fsub_rn ...) to follow numpy's evaluation order.
// ---------------------------------------------------------------------------------------------------------
__device__ inline double box_bin(int i) {   // np.linspace(0, 1, 1024)[i]
    return (i >= 1023) ? 1.0 : __dadd_rn(__dmul_rn((double)i, 1.0 / 1023.0), 0.0);
}
__constant__ double kBoxLo[10] = {-64, -64, -5, 0, 0, 0, -3.14, -20, -15, -0.3};
__constant__ double kBoxHi[10] = {64, 64, 5, 15, 4, 5, 3.14, 20, 15, 0.3};

__device__ inline bool gt_cross(float a1, float a0, float b1, float b0) {   // a1*a0 > b1*b0 in fp32
    return __fmul_rn(a1, a0) > __fmul_rn(b1, b0);
}

__device__ bool check_collision_dev(const double* boxes, int n, float (*cor)[8]) {
    if (n == 1) return false;
    int m = 0;
    for (int i = 0; i < n; ++i) {
        const double* bx = boxes + i * 10;
        if (bx[0] >= 63.0) continue;   // fliter_and_map_object (misc.py:475-481)
        const double l = bx[3], w = bx[4], ang = -bx[6];
        const double sn = sin(ang), cs = cos(ang);
        const double tx[4] = {-0.5, -0.5, 0.5, 0.5}, ty[4] = {-0.5, 0.5, 0.5, -0.5};
        for (int k = 0; k < 4; ++k) {
            const double x = __dmul_rn(tx[k], l), y = __dmul_rn(ty[k], w);
            const double rx = __dadd_rn(__dmul_rn(x, cs), __dmul_rn(y, -sn));
            const double ry = __dadd_rn(__dmul_rn(x, sn), __dmul_rn(y, cs));
            cor[m][2 * k] = (float)__dadd_rn(rx, bx[0]);
            cor[m][2 * k + 1] = (float)__dadd_rn(ry, bx[1]);
        }
        ++m;
    }
    if (m <= 1) return false;
    const float* q = cor[m - 1];
    float qx0 = q[0], qx1 = q[0], qy0 = q[1], qy1 = q[1];
    for (int k = 1; k < 4; ++k) { qx0 = fminf(qx0, q[2 * k]); qx1 = fmaxf(qx1, q[2 * k]); qy0 = fminf(qy0, q[2 * k + 1]); qy1 = fmaxf(qy1, q[2 * k + 1]); }
    for (int i = 0; i < m; ++i) {
        const float* bq = cor[i];
        float bx0 = bq[0], bx1 = bq[0], by0 = bq[1], by1 = bq[1];
        for (int k = 1; k < 4; ++k) { bx0 = fminf(bx0, bq[2 * k]); bx1 = fmaxf(bx1, bq[2 * k]); by0 = fminf(by0, bq[2 * k + 1]); by1 = fmaxf(by1, bq[2 * k + 1]); }
        const float iw = __fsub_rn(fminf(bx1, qx1), fmaxf(bx0, qx0));
        if (!(iw > 0.f)) continue;
        const float ih = __fsub_rn(fminf(by1, qy1), fmaxf(by0, qy0));
        if (!(ih > 0.f)) continue;
        for (int k = 0; k < 4; ++k) {
            const float A0 = bq[2 * k], A1 = bq[2 * k + 1], B0 = bq[2 * ((k + 1) & 3)], B1 = bq[2 * ((k + 1) & 3) + 1];
            for (int l2 = 0; l2 < 4; ++l2) {
                const float C0 = q[2 * l2], C1 = q[2 * l2 + 1], D0 = q[2 * ((l2 + 1) & 3)], D1 = q[2 * ((l2 + 1) & 3) + 1];
                const bool acd = gt_cross(__fsub_rn(D1, A1), __fsub_rn(C0, A0), __fsub_rn(C1, A1), __fsub_rn(D0, A0));
                const bool bcd = gt_cross(__fsub_rn(D1, B1), __fsub_rn(C0, B0), __fsub_rn(C1, B1), __fsub_rn(D0, B0));
                if (acd != bcd) {
                    const bool abc = gt_cross(__fsub_rn(C1, A1), __fsub_rn(B0, A0), __fsub_rn(B1, A1), __fsub_rn(C0, A0));
                    const bool abd = gt_cross(__fsub_rn(D1, A1), __fsub_rn(B0, A0), __fsub_rn(B1, A1), __fsub_rn(D0, A0));
                    if (abc != abd) return true;
                }
            }
        }
    }
    return false;
}

__device__ inline void write_next_input(const SampleArgs& a, int b, int j, const float* emb_f, const bf16_t* emb_b) {
    const int E = a.tb.E;
    const float* cr = a.cond + ((long)b * kSeq + (j + 1)) * E;
    float* xo = a.x_next + (long)b * E;
    for (int c = threadIdx.x; c < E; c += blockDim.x) xo[c] = (emb_f ? emb_f[c] : bf16_to_f32(emb_b[c])) + cr[c];
}
// the same with the conditioning row already in registers (requested at the start of the sampler kernel: one memory round trip
// less behind the sampled token); E <= 6 * 256
struct CondRow { float v[6]; };
__device__ inline CondRow load_cond_row(const SampleArgs& a, int b, int j) {
    CondRow r;
    const float* cr = a.cond + ((long)b * kSeq + (j + 1)) * a.tb.E;
#pragma unroll
    for (int i = 0; i < 6; ++i) { const int c = threadIdx.x + 256 * i; r.v[i] = c < a.tb.E ? cr[c] : 0.f; }
    return r;
}
__device__ inline void write_next_input(const SampleArgs& a, int b, const float* emb_f, const CondRow& r) {
    float* xo = a.x_next + (long)b * a.tb.E;
#pragma unroll
    for (int i = 0; i < 6; ++i) { const int c = threadIdx.x + 256 * i; if (c < a.tb.E) xo[c] = emb_f[c] + r.v[i]; }
}

// steps whose emitted scene token is known: pose prefix (bos, 3 pose tokens, eos) and every bos/eos (d_token_pos,
// UMGen.py:976-984, 1046-1050)
__global__ __launch_bounds__(256) void fixed_token_kernel(SampleArgs a) {
    const int b = blockIdx.x, j = a.st->step, E = a.tb.E;
    const int aux = fixed_aux_id(j);
    const int* toks = a.tokens + (long)b * kTokPerFrame;
    if (aux >= 0) write_next_input(a, b, j, a.tb.axe + (long)aux * E, nullptr);
    else if (j < kPoseEos) write_next_input(a, b, j, nullptr, a.tb.fouier_pe + (long)toks[j - 1] * E);
    // GIVEN map / bbox3d tokens (infer_oar_net's predefined-token prefix, UMGen.py:1184-1201): get_mod_emb_pre of the given token --
    // the GMLP(codebook) row / the be row, no position table -- exactly what sample_token_kernel feeds back for a sampled one
    else if (j < kMapEos) write_next_input(a, b, j, a.tb.gmap + (long)toks[kOffMap + (j - kMapC0)] * E, nullptr);
    else write_next_input(a, b, j, a.tb.be + (long)toks[kOffBox + (j - kBoxC0)] * E, nullptr);
    finish_step(a.st);
}
void launch_fixed_token(hipStream_t s, const SampleArgs& a, int B) { hipLaunchKernelGGL(fixed_token_kernel, dim3(B), dim3(256), 0, s, a); }

// The decode inputs of the GIVEN positions 0 .. P - 1 of every scene at once (infer_oar_net's predefined-token prefix, UMGen.py:1184-1201): row j is
// what the step loop would have fed into step j -- the task row + cond[0] for j = 0 (first_input_kernel), the embedding of the token at
// position j - 1 + cond[j] behind it (fixed_token_kernel) -- none of which depends on a layer output.  Rows 0 .. P - 2 go to X (the prefix
// pass pushes them through the BlockOAR layers as the rows of GEMMs), row P - 1 to x_last (the input of the first step the loop still runs).
__global__ __launch_bounds__(256) void prefix_rows_kernel(EmbedTables tb, const float* __restrict__ tske_row, const float* __restrict__ cond,
                                                          const int* __restrict__ tokens, int P, float* __restrict__ X, float* __restrict__ x_last) {
    const int b = blockIdx.x / P, j = blockIdx.x % P, E = tb.E;
    const float* cr = cond + ((long)b * kSeq + j) * E;
    float* xo = j + 1 < P ? X + ((long)b * (P - 1) + j) * E : x_last + (long)b * E;
    const int* toks = tokens + (long)b * kTokPerFrame;
    const float* ef = nullptr;
    const bf16_t* eb = nullptr;
    if (j == 0) ef = tske_row;
    else {
        const int jp = j - 1, aux = fixed_aux_id(jp);      // the token at position j - 1
        if (aux >= 0) ef = tb.axe + (long)aux * E;
        else if (jp < kPoseEos) eb = tb.fouier_pe + (long)toks[jp - 1] * E;
        else if (jp < kMapEos) ef = tb.gmap + (long)toks[kOffMap + (jp - kMapC0)] * E;
        else ef = tb.be + (long)toks[kOffBox + (jp - kBoxC0)] * E;
    }
    for (int c = threadIdx.x; c < E; c += blockDim.x) xo[c] = (ef ? ef[c] : bf16_to_f32(eb[c])) + cr[c];
}
void launch_prefix_rows(hipStream_t s, const EmbedTables& tb, const float* tske_row, const float* cond, const int* tokens, int B, int P, float* X,
                        float* x_last) {
    hipLaunchKernelGGL(prefix_rows_kernel, dim3(B * P), dim3(256), 0, s, tb, tske_row, cond, tokens, P, X, x_last);
}

// k rows (row-major [B S][2E]: q | k) and V^T ([B][H][48][S_pad]) of a prefix pass -> rows 0 .. S - 1 of one layer's decode cache
// ([scene][2][H][Lmax][48], the type the pass computed in)
template <typename T>
__global__ __launch_bounds__(256) void prefix_kv_to_cache_kernel(const T* __restrict__ qk, const T* __restrict__ vt, int S, int S_pad, int H, int Lmax,
                                                                  T* __restrict__ cache, long scene_stride) {
    const int E = H * kHeadDim;
    const int b = blockIdx.y;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;       // (key, head, dim)
    if (i >= (long)S * E) return;
    const int j = (int)(i / E), hd = (int)(i % E), h = hd / kHeadDim, d = hd % kHeadDim;
    T* c = cache + (long)b * scene_stride;
    c[((long)h * Lmax + j) * kHeadDim + d] = qk[((long)b * S + j) * 2 * E + E + hd];
    c[((long)(H + h) * Lmax + j) * kHeadDim + d] = vt[(((long)b * H + h) * kHeadDim + d) * S_pad + j];
}
template <typename T>
void launch_prefix_kv_to_cache(hipStream_t s, const T* qk, const T* vt, int B, int S, int S_pad, int H, int Lmax, T* cache, long scene_stride) {
    const long n = (long)S * H * kHeadDim;
    hipLaunchKernelGGL(prefix_kv_to_cache_kernel<T>, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, s, qk, vt, S, S_pad, H, Lmax, cache, scene_stride);
}
template void launch_prefix_kv_to_cache<float>(hipStream_t, const float*, const float*, int, int, int, int, int, float*, long);
template void launch_prefix_kv_to_cache<bf16_t>(hipStream_t, const bf16_t*, const bf16_t*, int, int, int, int, int, bf16_t*, long);
template void launch_prefix_kv_to_cache<f16_t>(hipStream_t, const f16_t*, const f16_t*, int, int, int, int, int, f16_t*, long);

__global__ __launch_bounds__(256) void sample_token_kernel(SampleArgs a) {
    __shared__ SamplerLds sh;
    __shared__ float cor[64][8];
    __shared__ int s_tok;
    const int b = blockIdx.x, j = a.st->step, frame = a.st->frame_idx, E = a.tb.E;
    const SamplerParams sp = a.st->sp;
    const bool use_forced = a.st->use_forced != 0, use_control = a.st->use_control != 0;
    const int pos1 = j + 1;   // the reference's 1-based curr_seq_len
    const unsigned long long seed = a.seeds[b];
    const CondRow crow = load_cond_row(a, b, j);
    const float* lg = a.logits + (long)b * a.ld_logits;
    const int topk = a.mod == 1 ? sp.top_k_map : (a.mod == 3 ? sp.topk_image : sp.top_k);
    // top-p: the image head receives topk_image as its "p" (UMGen.py:1133) => the whole distribution is kept
    const float topp = a.mod == 1 ? sp.p_map : (a.mod == 3 ? (float)sp.topk_image : sp.p);
    int tok = block_sample(sp, lg, a.vocab, topk, topp, rng_uniform(seed, frame, pos1, DRAW_MAIN), -1, sh, a.counters + 7);
    int off, k;
    if (a.mod == 1) { off = kOffMap; k = j - kMapC0; }
    else if (a.mod == 2) { off = kOffBox; k = j - kBoxC0; }
    else { off = kOffImg; k = j - kImgC0; }
    int* toks = a.tokens + (long)b * kTokPerFrame;
    if (a.mod == 2) {
        const float* lt = a.logits_tar + ((long)b * kNBox + k) * a.ld_tar;
        const int prev = a.prev_box[(long)b * kNBox + k];
        if (use_control) {   // UMGen.py:1083-1089
            const int object_id = (pos1 - 1032) / kSlotLen;
            if (object_id < kSlots && a.control_slot[b * kSlots + object_id]) {
                tok = block_sample(sp, lt, a.vocab, sp.top_k, sp.p, rng_uniform(seed, frame, pos1, DRAW_CONTROL), a.vocab - 1, sh, a.counters + 7);
                if (threadIdx.x == 0) atomicAdd(a.counters + 1, 1);
            }
        }
        if (tok == kBoxPad && sp.merge_ar_tar && prev != kBoxPad && !sp.only_ar) {   // UMGen.py:1092-1104
            tok = block_sample(sp, lt, a.vocab, sp.top_k, sp.p, rng_uniform(seed, frame, pos1, DRAW_PAD_AVOID), -1, sh, a.counters + 7);
            if (threadIdx.x == 0) atomicAdd(a.counters + 0, 1);
        }
        if (sp.rule_constrain && !use_forced && tok != kBoxPad && (pos1 - 1032) % kSlotLen == 0) {   // UMGen.py:1116-1123
            if (threadIdx.x == 0) {
                double* boxes = a.boxes + (long)b * 64 * 10;
                int n = a.n_boxes[b];
                if (n == 0) {
                    const double ego[10] = {0, 0, 0, 5.176, 2.297, 1.777, 0, 0, 0, 0};
                    for (int q = 0; q < 10; ++q) boxes[q] = ego[q];
                    n = 1;
                }
                double* nb = boxes + n * 10;
                for (int q = 0; q < 10; ++q) {
                    const int t = toks[off + k - 10 + q];
                    const int right = min(max(t, 0), 1023), left = min(max(t - 1, 0), 1023);
                    const double v = __dadd_rn(box_bin(left), box_bin(right)) / 2.0;
                    nb[q] = __dadd_rn(__dmul_rn(v, kBoxHi[q] - kBoxLo[q]), kBoxLo[q]);
                }
                ++n;
                const bool collision = check_collision_dev(boxes, n, cor);
                const bool newborn = (prev == kBoxPad);
                atomicAdd(a.counters + 2, 1);
                if (collision) atomicAdd(a.counters + 3, 1);
                int t2 = tok;
                if ((newborn && collision) || (n > 30 && newborn)) {
                    for (int q = 1; q <= 10; ++q) toks[off + k - q] = kBoxPad;
                    --n;
                    t2 = kBoxPad;
                    atomicAdd(a.counters + 4, 1);
                }
                a.n_boxes[b] = n;
                s_tok = t2;
            }
            __syncthreads();
            tok = s_tok;
        }
    }
    if (use_forced) {
        const int ft = a.forced[(long)b * kTokPerFrame + off + k];
        if (threadIdx.x == 0 && ft != tok) atomicAdd(a.counters + 5, 1);
        tok = ft;
    }
    if (threadIdx.x == 0) toks[off + k] = tok;
    const float* emb = (a.mod == 1) ? a.tb.gmap + (long)tok * E : (a.mod == 3 ? a.tb.gimg + (long)tok * E : a.tb.be + (long)tok * E);
    write_next_input(a, b, emb, crow);
    finish_step(a.st);
}
void launch_sample_token(hipStream_t s, const SampleArgs& a, int B) { hipLaunchKernelGGL(sample_token_kernel, dim3(B), dim3(256), 0, s, a); }

__global__ __launch_bounds__(256) void sample_ego_kernel(const float* __restrict__ logits, int vocab, SamplerParams sp,
                                                         const unsigned long long* __restrict__ seeds, int frame_idx,
                                                         const int* __restrict__ forced, int* __restrict__ out_tokens, int* overflow) {
    __shared__ SamplerLds sh;
    const int b = blockIdx.x / 3, jq = blockIdx.x % 3;
    int tok = block_sample(sp, logits + (long)blockIdx.x * vocab, vocab, sp.top_k, sp.p,
                           rng_uniform(seeds[b], frame_idx, kSeq + jq, DRAW_MAIN), -1, sh, overflow);
    if (forced) tok = forced[(long)b * kTokPerFrame + jq];
    if (threadIdx.x == 0) out_tokens[b * 3 + jq] = tok;
}
void launch_sample_ego(hipStream_t s, const float* logits, int vocab, SamplerParams sp, const unsigned long long* seeds, int frame_idx,
                       const int* forced, int* out_tokens, int B, int* overflow) {
    hipLaunchKernelGGL(sample_ego_kernel, dim3(B * 3), dim3(256), 0, s, logits, vocab, sp, seeds, frame_idx, forced, out_tokens, overflow);
}

// test hook (debug_api.hip): the top-k sampler on n independent logit rows with given uniforms
__global__ __launch_bounds__(256) void sample_rows_kernel(const float* __restrict__ logits, int V, int k, float temp, const float* __restrict__ u,
                                                          int* __restrict__ out, int* overflow) {
    __shared__ SamplerLds sh;
    const int tok = block_sample_topk(logits + (long)blockIdx.x * V, V, k, temp, u[blockIdx.x], -1, sh.k, overflow);
    if (threadIdx.x == 0) out[blockIdx.x] = tok;
}
void launch_sample_rows(hipStream_t s, const float* logits, int V, int k, float temp, const float* u, int* out, int* overflow, int n) {
    hipLaunchKernelGGL(sample_rows_kernel, dim3(n), dim3(256), 0, s, logits, V, k, temp, u, out, overflow);
}

}  // namespace umgen
