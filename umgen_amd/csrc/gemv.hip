// Few-row linear layers for the OAR decode step and the ego decoder (M = scenes or 3*scenes rows).
// Replaces F.linear at module.py:206,229 (c_attn, c_proj), 246-248 (MLP) and the heads (UMGen.py:1062,1072,1087,1132)
// for single-token inputs, with LayerNorm (module.py:34-37), exact GELU and the residual add fused in.
//
// These launches are latency-bound weight streams (a decode layer is 14 MB; the whole chip drains that in ~2.5 us), so the
// kernels are written around the dependency chain, not around bandwidth:
//   * every weight byte of the launch is requested in the first instructions of each wave (16 B per lane, whole rows held in
//     registers) -- no loop-carried load->use->load chain, no LDS staging, no barrier on the weight path;
//   * the activation row(s) live in registers in exactly the k-chunks the lane needs for its dot products; LayerNorm statistics
//     are wave reductions over those registers (each wave recomputes them: 768 floats, cheaper than a barrier);
//   * weights stay bf16 in HBM (precision mode bf16) and are widened in-register; activations and accumulation are fp32.
#include "kernels.h"

namespace umgen {

template <typename T> struct WChunk;                      // 8 consecutive weights of one row
template <> struct WChunk<bf16_t> { uint4 v; };
template <> struct WChunk<f16_t> { uint4 v; };
template <> struct WChunk<float> { float4 a, b; };

// weights are streamed exactly once per launch: non-temporal loads (MI355X_MICROARCH.md "nt-weights": issue->landed -18 %)
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4v_t __attribute__((ext_vector_type(4)));
// (NT = false when several workgroups of one XCD read the same rows: one scene per workgroup, the others hit the L2)
template <bool NT = true>
__device__ inline void wload(WChunk<bf16_t>& w, const bf16_t* p) {
#ifndef UMGEN_NO_NT
    if (NT) {
        const u32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
        w.v = make_uint4(t.x, t.y, t.z, t.w);
        return;
    }
#endif
    w.v = *reinterpret_cast<const uint4*>(p);
}
template <bool NT = true>
__device__ inline void wload(WChunk<f16_t>& w, const f16_t* p) {
    WChunk<bf16_t> t;
    wload<NT>(t, reinterpret_cast<const bf16_t*>(p));
    w.v = t.v;
}
template <bool NT = true>
__device__ inline void wload(WChunk<float>& w, const float* p) {
#ifndef UMGEN_NO_NT
    if (NT) {
        const f32x4v_t a = __builtin_nontemporal_load(reinterpret_cast<const f32x4v_t*>(p));
        const f32x4v_t b = __builtin_nontemporal_load(reinterpret_cast<const f32x4v_t*>(p + 4));
        w.a = make_float4(a.x, a.y, a.z, a.w);
        w.b = make_float4(b.x, b.y, b.z, b.w);
        return;
    }
#endif
    w.a = *reinterpret_cast<const float4*>(p);
    w.b = *reinterpret_cast<const float4*>(p + 4);
}

// One (feature tile, row) per workgroup with XCD affinity: workgroup b runs on XCD b % 8 (observed), so the rows of one tile are
// given to workgroups b, b + 8, b + 16, ... -- the tile's weights are fetched from HBM into ONE L2 and hit there for the other rows.
__device__ inline void tile_row_of_block(int bid, int rows, int& tile, int& row) {
    const int q = bid >> 3;
    row = q % rows;
    tile = (q / rows) * 8 + (bid & 7);
}
inline int tile_row_grid(int tiles, int rows) { return ((tiles + 7) / 8) * 8 * rows; }
__device__ inline void wzero(WChunk<bf16_t>& w) { w.v = make_uint4(0, 0, 0, 0); }
__device__ inline void wzero(WChunk<f16_t>& w) { w.v = make_uint4(0, 0, 0, 0); }
__device__ inline void wzero(WChunk<float>& w) { w.a = make_float4(0, 0, 0, 0); w.b = w.a; }
__device__ inline void wunpack(const WChunk<bf16_t>& w, float (&o)[8]) {
    o[0] = __uint_as_float(w.v.x << 16); o[1] = __uint_as_float(w.v.x & 0xffff0000u);
    o[2] = __uint_as_float(w.v.y << 16); o[3] = __uint_as_float(w.v.y & 0xffff0000u);
    o[4] = __uint_as_float(w.v.z << 16); o[5] = __uint_as_float(w.v.z & 0xffff0000u);
    o[6] = __uint_as_float(w.v.w << 16); o[7] = __uint_as_float(w.v.w & 0xffff0000u);
}
__device__ inline void wunpack(const WChunk<f16_t>& w, float (&o)[8]) {
    const f16x8_t h = __builtin_bit_cast(f16x8_t, w.v);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)h[e];
}
__device__ inline void wunpack(const WChunk<float>& w, float (&o)[8]) {
    o[0] = w.a.x; o[1] = w.a.y; o[2] = w.a.z; o[3] = w.a.w; o[4] = w.b.x; o[5] = w.b.y; o[6] = w.b.z; o[7] = w.b.w;
}

// ---------------------------------------------------------------------------------------------------------
// out[m][n] = LN(x[m]) . W[n] + bias[n]     K = n_embd (<= 1536): NCH = ceil(K / 512) chunks of 8 per lane
// ---------------------------------------------------------------------------------------------------------
template <typename T, int MB, int NCH, int RPW, bool PERROW = false>
__device__ __forceinline__ void gemv_ln_body(const GemvArgs& a, int bid) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int K = a.K;
    const T* W = reinterpret_cast<const T*>(a.W);
    int tile = bid, mb = 0;                       // mb: first input row of this workgroup
    if (PERROW) { tile_row_of_block(bid, (a.M + MB - 1) / MB, tile, mb); mb *= MB; }
    const int Mloc = PERROW ? min(MB, a.M - mb) : a.M;
    const int n0 = (tile * 4 + wave) * RPW;
    if (n0 >= a.N) return;
    WChunk<T> w[RPW][NCH];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const T* wr = W + (long)min(n0 + r, a.N - 1) * K;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane * 8 + 512 * i;
            if (c < K) wload<!PERROW>(w[r][i], wr + c); else wzero(w[r][i]);
        }
    }
    float bias[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) bias[r] = (a.bias && n0 + r < a.N) ? a.bias[n0 + r] : 0.f;
    const int pos = a.d_len ? *a.d_len : 0;
    const float* xbase = a.x + (a.d_xoff ? (long)(*a.d_xoff) * a.xoff_mul : 0L);
    float lw[NCH][8];
    if (a.ln_w) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane * 8 + 512 * i;
            if (c < K) load8(a.ln_w + c, lw[i]);
        }
    }
    for (int m0 = 0; m0 < Mloc; m0 += MB) {
        float xv[MB][NCH][8];
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const float* xr = xbase + (long)(mb + min(m0 + m, Mloc - 1)) * a.ldx;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = lane * 8 + 512 * i;
                if (c < K) load8(xr + c, xv[m][i]);
                else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) xv[m][i][e] = 0.f;
                }
            }
        }
        if (a.ln_w) {
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < NCH; ++i)
#pragma unroll
                    for (int e = 0; e < 8; ++e) s += xv[m][i][e];
                const float mean = wave_sum(s) / (float)K;
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    if (lane * 8 + 512 * i < K) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) { const float d = xv[m][i][e] - mean; q += d * d; }
                    }
                }
                const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)K + 1e-5f);
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    if (lane * 8 + 512 * i < K) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) xv[m][i][e] = (xv[m][i][e] - mean) * rstd * lw[i][e];
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            float acc[MB];
#pragma unroll
            for (int m = 0; m < MB; ++m) acc[m] = 0.f;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                float w8[8];
                wunpack(w[r][i], w8);
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[m] = fmaf(w8[e], xv[m][i][e], acc[m]);
            }
#pragma unroll
            for (int m = 0; m < MB; ++m) acc[m] = wave_sum(acc[m]);
            const int n = n0 + r;
            if (lane == 0 && n < a.N) {
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    const int mm = mb + m0 + m;
                    if (m0 + m < Mloc) {
                        const float v = acc[m] + bias[r];
                        if (a.out_mode == GEMV_OUT_QKV) {
                            if (n < a.E) __builtin_nontemporal_store(v, &a.out[(long)mm * a.ldo + n]);
                            else {
                                const int c = n - a.E, kvsel = c / a.E, hc = c % a.E;   // kvsel 0 = K, 1 = V
                                const long H = a.E / kHeadDim;
                                reinterpret_cast<T*>(a.cache)[(long)mm * a.scene_stride + ((kvsel * H + hc / kHeadDim) * a.Lmax + pos) * kHeadDim +
                                                              hc % kHeadDim] = Cvt<T>::from_f(v);
                            }
                        } else {
                            // (head logits are read by ONE workgroup right away: a plain store keeps them in the L2)
                            if (a.out_mode == GEMV_OUT_GELU) __builtin_nontemporal_store(gelu_erf(v), &a.out[(long)mm * a.ldo + n]);
                            else a.out[(long)mm * a.ldo + n] = v;
                        }
                    }
                }
            }
        }
    }
}

template <typename T, int MB, int NCH, int RPW, bool PERROW = false>
__global__ __launch_bounds__(256) void gemv_ln_kernel(GemvArgs a) {
#ifdef UMGEN_DRY_DECODE
    return;   // launch-floor experiment: same graph, no work
#endif
    gemv_ln_body<T, MB, NCH, RPW, PERROW>(a, blockIdx.x);
}

template <typename T, int NCH>
static void launch_gemv_nch(hipStream_t s, const GemvArgs& a) {
    constexpr int RPW = 2;   // measured: 2 rows per wave (288 workgroups for N=2304) beats 1 and 4
    const int grid = (a.N + 4 * RPW - 1) / (4 * RPW);
    // several rows (scenes): one row per workgroup keeps every launch on the single-row dependency chain (measured at 4 scenes:
    // 8.3 us for the row-looping form vs 4.8 us for one row) at the price of L2 re-reads of the weights
    if (a.M > 1 && a.rows_per_block == 1)
        hipLaunchKernelGGL((gemv_ln_kernel<T, 1, NCH, RPW, true>), dim3(tile_row_grid(grid, a.M)), dim3(256), 0, s, a);
    else if (a.M > 2 && a.rows_per_block == 2)
        hipLaunchKernelGGL((gemv_ln_kernel<T, 2, NCH, RPW, true>), dim3(tile_row_grid(grid, (a.M + 1) / 2)), dim3(256), 0, s, a);
    else if (a.M == 1) hipLaunchKernelGGL((gemv_ln_kernel<T, 1, NCH, RPW>), dim3(grid), dim3(256), 0, s, a);
    else if (a.M == 2) hipLaunchKernelGGL((gemv_ln_kernel<T, 2, NCH, RPW>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((gemv_ln_kernel<T, 4, NCH, RPW>), dim3(grid), dim3(256), 0, s, a);
}

template <typename T>
void launch_gemv(hipStream_t s, const GemvArgs& a) {
    if (a.K <= 512) launch_gemv_nch<T, 1>(s, a);
    else if (a.K <= 1024) launch_gemv_nch<T, 2>(s, a);
    else launch_gemv_nch<T, 3>(s, a);
}
template void launch_gemv<float>(hipStream_t, const GemvArgs&);
template void launch_gemv<bf16_t>(hipStream_t, const GemvArgs&);
template void launch_gemv<f16_t>(hipStream_t, const GemvArgs&);

// ---------------------------------------------------------------------------------------------------------
// x[m][n] += (sum_k a[m][k] W[n][k]) + bias[n]    one weight row per wave; K = n_embd or 4 n_embd (NCH chunks)
// COMBINE: a[m][:] is first merged from the attention partials (K == H*48) into LDS, after the weight loads are in flight
// ---------------------------------------------------------------------------------------------------------

template <typename T, int MB, int NCH, bool COMBINE, bool PERROW = false>
__global__ __launch_bounds__(256) void gemv_resid_kernel(GemvResidArgs a_in) {
#ifdef UMGEN_DRY_DECODE
    return;
#endif
    extern __shared__ __attribute__((aligned(16))) float as[];   // [M][K] when COMBINE
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    GemvResidArgs a = a_in;
    int tile = blockIdx.x;
    if (PERROW) {   // this workgroup: one (feature tile, row); everything below sees a one-row problem
        int mb;
        tile_row_of_block(blockIdx.x, (a_in.M + MB - 1) / MB, tile, mb);
        mb *= MB;
        a.M = min(MB, a_in.M - mb);
        a.x += (long)mb * a.ldx;
        if (a.a) a.a += (long)mb * a.lda;
        if (a.part) a.part += (long)mb * a.H * kAttnRec;
    }
    const int K = a.K;
    const T* W = reinterpret_cast<const T*>(a.W);
    const int n = tile * 4 + wave;
    const bool active = n < a.N;
    WChunk<T> w[NCH];
    {
        const T* wr = W + (long)min(n, a.N - 1) * K;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane * 8 + 512 * i;
            if (c < K) wload<!PERROW>(w[i], wr + c); else wzero(w[i]);
        }
    }
    const float bias = (a.bias && active) ? a.bias[n] : 0.f;
    if (COMBINE) {
        // merge the split-softmax partials (m, l, o[48]) of every (row, head): o = sum_s e^{m_s - M} o_s / sum_s e^{m_s - M} l_s.
        // Stage A: the (row, head, split) statistics go to LDS; stage B: every output column folds its own weights from LDS
        // (<= 18 LDS reads) and streams the o_s values with independent loads.
        float* s_w = as + (long)a.M * K;                      // [M][H][kAttnPad] normalised split weights e^{m_s - M} / L
        const int ns = a.ns;
        const int tid = threadIdx.x;
        const int n_el = a.M * K;
        // 1. request the first 768 columns' o_s values (independent of the weights) before anything else
        float4 ov[3][kAttnPad / 4];
        auto load_chunk = [&](int base) {
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const int e = base + tid + 256 * it;
                if (e < n_el) {
                    const int m = e / K, col = e % K;
                    const float4* po = reinterpret_cast<const float4*>(a.part + (long)(m * a.H + col / kHeadDim) * kAttnRec + 2 * kAttnPad +
                                                                       (col % kHeadDim) * kAttnPad);
#pragma unroll
                    for (int i = 0; i < kAttnPad / 4; ++i) ov[it][i] = (4 * i < ns) ? po[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        };
        load_chunk(0);
        // 2. one thread per (row, head): softmax-merge weights
        if (tid < a.M * a.H) {
            const float* p = a.part + (long)tid * kAttnRec;
            float pm[kAttnPad], pl[kAttnPad];
#pragma unroll
            for (int i = 0; i < kAttnPad / 4; ++i) {
                const float4 m4 = (4 * i < ns) ? reinterpret_cast<const float4*>(p)[i] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
                const float4 l4 = (4 * i < ns) ? reinterpret_cast<const float4*>(p + kAttnPad)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                pm[4 * i] = m4.x; pm[4 * i + 1] = m4.y; pm[4 * i + 2] = m4.z; pm[4 * i + 3] = m4.w;
                pl[4 * i] = l4.x; pl[4 * i + 1] = l4.y; pl[4 * i + 2] = l4.z; pl[4 * i + 3] = l4.w;
            }
            float mx = -INFINITY;
#pragma unroll
            for (int sp = 0; sp < kAttnPad; ++sp) { if (sp >= ns) pm[sp] = -INFINITY; mx = fmaxf(mx, pm[sp]); }
            float l = 0.f;
#pragma unroll
            for (int sp = 0; sp < kAttnPad; ++sp) { pm[sp] = expf(pm[sp] - mx); l = fmaf(pm[sp], (sp < ns) ? pl[sp] : 0.f, l); }
            const float inv = 1.0f / l;
#pragma unroll
            for (int sp = 0; sp < kAttnPad - 1; ++sp) s_w[tid * kAttnPad + sp] = pm[sp] * inv;
            s_w[tid * kAttnPad + kAttnPad - 1] = 0.f;              // kAttnSplit <= kAttnPad - 1, so this slot is never a split
        }
        __syncthreads();
        // 3. fold, 768 columns at a time
        for (int base = 0; base < n_el; base += 768) {
            if (base) load_chunk(base);
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const int e = base + tid + 256 * it;
                if (e < n_el) {
                    const int m = e / K, col = e % K;
                    const float* w = s_w + (m * a.H + col / kHeadDim) * kAttnPad;
                    float o = 0.f;
#pragma unroll
                    for (int i = 0; i < kAttnPad / 4; ++i) {
                        o = fmaf(w[4 * i], ov[it][i].x, o);
                        o = fmaf(w[4 * i + 1], ov[it][i].y, o);
                        o = fmaf(w[4 * i + 2], ov[it][i].z, o);
                        if (4 * i + 3 < kAttnPad - 1) o = fmaf(w[4 * i + 3], ov[it][i].w, o);
                    }
                    as[e] = o;
                }
            }
        }
        __syncthreads();
    }
    if (!active) return;
    for (int m0 = 0; m0 < a.M; m0 += MB) {
        float xold[MB];
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < MB; ++m) xold[m] = (m0 + m < a.M) ? a.x[(long)(m0 + m) * a.ldx + n] : 0.f;
        }
        float acc[MB];
#pragma unroll
        for (int m = 0; m < MB; ++m) acc[m] = 0.f;
        float xv[MB][NCH][8];
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const int mm = min(m0 + m, a.M - 1);
            const float* xr = COMBINE ? (as + (long)mm * K) : (a.a + (long)mm * a.lda);
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = lane * 8 + 512 * i;
                if (c < K) load8(xr + c, xv[m][i]);
                else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) xv[m][i][e] = 0.f;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            float w8[8];
            wunpack(w[i], w8);
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[m] = fmaf(w8[e], xv[m][i][e], acc[m]);
        }
#pragma unroll
        for (int m = 0; m < MB; ++m) acc[m] = wave_sum(acc[m]);
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
                if (m0 + m < a.M) __builtin_nontemporal_store(xold[m] + (acc[m] + bias), &a.x[(long)(m0 + m) * a.ldx + n]);
        }
    }
}

template <typename T, int NCH, bool COMBINE>
static void launch_resid_nch(hipStream_t s, const GemvResidArgs& a) {
    const int grid = (a.N + 3) / 4;
    if (a.M > 1 && a.rows_per_block == 1) {   // one (feature tile, row) per workgroup, see launch_gemv_nch
        const size_t shm1 = COMBINE ? ((size_t)a.K + (size_t)a.H * kAttnPad) * sizeof(float) : 0;
        hipLaunchKernelGGL((gemv_resid_kernel<T, 1, NCH, COMBINE, true>), dim3(tile_row_grid(grid, a.M)), dim3(256), shm1, s, a);
        return;
    }
    if (a.M > 2 && a.rows_per_block == 2 && NCH <= 6) {
        const size_t shm2 = COMBINE ? 2 * ((size_t)a.K + (size_t)a.H * kAttnPad) * sizeof(float) : 0;
        hipLaunchKernelGGL((gemv_resid_kernel<T, 2, NCH, COMBINE, true>), dim3(tile_row_grid(grid, (a.M + 1) / 2)), dim3(256), shm2, s, a);
        return;
    }
    const size_t shm = COMBINE ? ((size_t)a.M * a.K + (size_t)a.M * a.H * kAttnPad) * sizeof(float) : 0;
    constexpr int MBmax = (NCH <= 3) ? 4 : (NCH <= 6 ? 2 : 1);
    if (a.M == 1 || MBmax == 1) hipLaunchKernelGGL((gemv_resid_kernel<T, 1, NCH, COMBINE>), dim3(grid), dim3(256), shm, s, a);
    else if (a.M == 2 || MBmax == 2) hipLaunchKernelGGL((gemv_resid_kernel<T, (MBmax >= 2 ? 2 : 1), NCH, COMBINE>), dim3(grid), dim3(256), shm, s, a);
    else hipLaunchKernelGGL((gemv_resid_kernel<T, MBmax, NCH, COMBINE>), dim3(grid), dim3(256), shm, s, a);
}

template <typename T>
void launch_gemv_resid(hipStream_t s, const GemvResidArgs& a0) {
    GemvResidArgs a = a0;
    const int nch = (a.K + 511) / 512;
    if (a.part && a.M > 8 && a.rows_per_block != 1) {   // the merged attention rows are staged in LDS: at most 8 rows per launch
        for (int m = 0; m < a0.M; m += 8) {
            GemvResidArgs b = a0;
            b.M = min(8, a0.M - m);
            b.part = a0.part + (long)m * a0.H * kAttnRec;
            b.x = a0.x + (long)m * a0.ldx;
            launch_gemv_resid<T>(s, b);
        }
        return;
    }
    if (a.part) {
        if (nch <= 1) launch_resid_nch<T, 1, true>(s, a);
        else if (nch <= 2) launch_resid_nch<T, 2, true>(s, a);
        else launch_resid_nch<T, 3, true>(s, a);
    } else {
        if (nch <= 1) launch_resid_nch<T, 1, false>(s, a);
        else if (nch <= 2) launch_resid_nch<T, 2, false>(s, a);
        else if (nch <= 3) launch_resid_nch<T, 3, false>(s, a);
        else if (nch <= 6) launch_resid_nch<T, 6, false>(s, a);
        else launch_resid_nch<T, 12, false>(s, a);
    }
}
template void launch_gemv_resid<float>(hipStream_t, const GemvResidArgs&);
template void launch_gemv_resid<bf16_t>(hipStream_t, const GemvResidArgs&);
template void launch_gemv_resid<f16_t>(hipStream_t, const GemvResidArgs&);

// ---------------------------------------------------------------------------------------------------------
// few-query attention, partial pass over one slice of <= kAttnChunk keys (OAR decode step; ego-decoder self/cross attention).
// 8 lanes own one key: lane piece p < 6 holds the 8 head-dim values [8p, 8p+8) of BOTH the K row and the V row of its keys,
// all requested up front (16 B per lane per row); scores are 8-lane shuffle sums, the softmax statistics and the P.V
// accumulation are wave shuffles + one LDS exchange across the 4 waves.  Only ceil(L / kAttnChunk) splits do work.
// ---------------------------------------------------------------------------------------------------------
constexpr float kScale = 0.14433756729740643f;   // float32(1/sqrt(48)), module.py:196-198
constexpr int kKeyPass = kAttnChunk / 32;        // keys per thread (32 keys per pass of the 256 threads)

struct AttnGeom {
    const float* q;            // [NQ][E]
    const void* kv_base; long scene_stride, head_stride, key_stride, v_off;
    int q_per_scene, H;
    const int* d_len; int len_add, kmax;
    float* part;
};

// One (head h, key split, query qi) block.
template <typename T>
__device__ __forceinline__ void attn_body(const AttnGeom& g, int h, int split, int qi) {
    __shared__ float s_max[4];
    __shared__ float s_sum[16];
    __shared__ float s_o[16][kHeadDim];          // one partial per (wave, row of 16 lanes)
    const int H = g.H;
    const int E = H * kHeadDim;
    const int k0 = split * kAttnChunk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int piece = tid & 7, kg = tid >> 3;        // 32 key groups
    const bool pact = piece < 6;
    const T* base = reinterpret_cast<const T*>(g.kv_base) + (long)(qi / g.q_per_scene) * g.scene_stride + h * g.head_stride + piece * 8;
    float kf[kKeyPass][8], vf[kKeyPass][8];
#pragma unroll
    for (int i = 0; i < kKeyPass; ++i) {
        const int k = min(k0 + kg + 32 * i, g.kmax - 1);
        if (pact) {   // cached K/V rows are read once per step: non-temporal, like the weights
            WChunk<T> ck, cv;
            wload(ck, base + (long)k * g.key_stride);
            wload(cv, base + (long)k * g.key_stride + g.v_off);
            wunpack(ck, kf[i]);
            wunpack(cv, vf[i]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { kf[i][e] = 0.f; vf[i][e] = 0.f; }
        }
    }
    float q8[8];
    if (pact) load8(g.q + (long)qi * E + h * kHeadDim + piece * 8, q8);
    else {
#pragma unroll
        for (int e = 0; e < 8; ++e) q8[e] = 0.f;
    }
    const int L = (g.d_len ? *g.d_len : 0) + g.len_add;
    const int k1 = min(L, k0 + kAttnChunk);
    float sc[kKeyPass];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < kKeyPass; ++i) {
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) d = fmaf(q8[e], kf[i][e], d);
        d += dpp_xor1(d);            // sum over the 8 lanes of the key (pieces 6, 7 hold zeros)
        d += dpp_xor2(d);
        d += dpp_half_mirror(d);
        d = (k0 + kg + 32 * i < k1) ? d * kScale : -INFINITY;
        sc[i] = d;
        mx = fmaxf(mx, d);
    }
    mx = wave_max(mx);
    if (lane == 0) s_max[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
    float ls = 0.f;
    float o8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o8[e] = 0.f;
#pragma unroll
    for (int i = 0; i < kKeyPass; ++i) {
        const float p = (k0 + kg + 32 * i < k1) ? expf(sc[i] - mx) : 0.f;   // keys past the slice contribute nothing
        ls += p;
#pragma unroll
        for (int e = 0; e < 8; ++e) o8[e] = fmaf(p, vf[i][e], o8[e]);
    }
    // the two key groups of a 16-lane row are folded with one DPP rotate; the 4 rows x 4 waves meet in LDS
#pragma unroll
    for (int e = 0; e < 8; ++e) o8[e] += dpp_xor8(o8[e]);
    ls += dpp_xor8(ls);
    const int row = wave * 4 + (lane >> 4);
    if ((lane & 15) < 6) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s_o[row][(lane & 15) * 8 + e] = o8[e];
    }
    if ((lane & 15) == 0) s_sum[row] = ls;
    __syncthreads();
    float* out = g.part + ((long)qi * H + h) * kAttnRec;
    if (tid < kHeadDim) {
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc += s_o[r][tid];
        out[2 * kAttnPad + tid * kAttnPad + split] = acc;
    }
    if (tid == 0) {
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc += s_sum[r];
        out[split] = mx;
        out[kAttnPad + split] = acc;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_partial_kernel(AttnGeom g) {
#ifdef UMGEN_DRY_DECODE
    return;
#endif
    attn_body<T>(g, blockIdx.x, blockIdx.y, blockIdx.z);
}

template <typename T>
void launch_attn_partial(hipStream_t s, const float* q, const T* kv_base, long scene_stride, long head_stride, long key_stride, long v_off,
                         int NQ, int q_per_scene, int H, const int* d_len, int len_add, int ns, float* part) {
    // kmax: rows that exist behind kv_base for one (scene, head): loads are clamped to it, the softmax masks by the true length
    const int kmax = d_len ? kAttnSplit * kAttnChunk : len_add;
    AttnGeom g{q, kv_base, scene_stride, head_stride, key_stride, v_off, q_per_scene, H, d_len, len_add, kmax, part};
    hipLaunchKernelGGL(attn_partial_kernel<T>, dim3(H, ns, NQ), dim3(256), 0, s, g);
}
template void launch_attn_partial<float>(hipStream_t, const float*, const float*, long, long, long, long, int, int, int, const int*, int, int, float*);
template void launch_attn_partial<bf16_t>(hipStream_t, const float*, const bf16_t*, long, long, long, long, int, int, int, const int*, int, int, float*);
template void launch_attn_partial<f16_t>(hipStream_t, const float*, const f16_t*, long, long, long, long, int, int, int, const int*, int, int, float*);

}  // namespace umgen
