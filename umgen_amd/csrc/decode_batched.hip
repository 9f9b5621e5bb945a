// Batched decode layer for MANY scenes per GPU (24 and more; UMGEN_DECODE_BATCHED): BlockOAR (module.py:378-428) for B scenes as five
// launches per layer whose cost is what the roofline says a large batch should cost -- the layer's weights ONCE per step plus every
// scene's K/V rows -- instead of the XCD-resident engine's one (scene, layer) item after the other (12.8 us per item whatever B is:
// 0.24 of HBM peak at 32 scenes, profiles/r03_bench_b32.json).
//   rows_mfma_kernel   out[m][n] = act[m][:] . W[n][:]  for the M <= 64 scenes of the batch on the matrix cores: the scenes are the 16
//                      B-columns of v_mfma_f32_16x16x32, ONE instruction multiplies 16 weight rows x 32 k by 16 scenes.  Activations
//                      stay fp32 like everywhere in the decode step: they enter as hi + lo 16-bit pairs (two MFMAs
//                      per k-step; 2^-17 relative in bf16, 2^-22 in fp16 -- the engine's own matrix-core phases do the same).
//                      Workgroup = 16 weight rows x all of K, K split over its 8 waves (k-steps w, w + 8, ...), the 8 partial sums of an
//                      output meet in LDS in wave order.  Prologue: LayerNorm (weight only, eps 1e-5, two-pass).  Activations live in
//                      FRAGMENT-MAJOR buffers (frag_index below): the order the matrix cores consume them.
//                      Epilogues: q rows + K/V cache rows, exact GELU, residual add, plain fp32 (heads).
//   attn_decode_batched_kernel   one workgroup per (scene, head) walks the head's L + 1 cached keys once: 4 lanes per key (12 dims
//                      each), 16 keys per wave pass, four passes in flight, online softmax per lane group, the 64 group states of the
//                      workgroup merged in LDS in a fixed order.  No key splits: with >= 16 scenes x 16 heads the launch fills the chip.
// Every output element of scene m is a function of scene m's column alone (the MFMA's columns do not mix) and its summation order
// does not depend on M: a batch of B scenes equals the B one-scene runs of the SAME path bit for bit (tests/test_gpu_decode_engine.py).
#include "kernels.h"

namespace umgen {

namespace {

template <typename TT>
__device__ inline void split8(const float (&v)[8], typename Mma16<TT>::vec& hi, typename Mma16<TT>::vec& lo) {
    typedef typename Mma16<TT>::elem elem;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const auto h = Cvt<TT>::from_f(v[e]);
        const float r = v[e] - Cvt<TT>::to_f(h);
        hi[e] = __builtin_bit_cast(elem, h);
        lo[e] = __builtin_bit_cast(elem, Cvt<TT>::from_f(r));
    }
}

constexpr int kRowsThreads = 512, kRowsWaves = 8;

// FRAGMENT-MAJOR activations (frag_index, kernels.h).  A wave's B operand of k-step s and column block nb is "lane l: the 8 values
// k = 32 s + 8 (l / 16) .. + 7 of scene 16 nb + l % 16".  Out of row-major [scene][K] rows that is 64 scattered 32-byte pieces per
// request, and the launches were bound by exactly that gather (every workgroup re-reads all M x K activations: 1.4 TB/s out of the L2
// at 64 scenes).  The batched launches therefore keep their activations in the order the matrix cores consume them, so a request is
// 2 KB contiguous.  Producers scatter single elements into it (their volume is tiny), consumers stream it.
__global__ void rows_to_frag_kernel(const float* __restrict__ x, long ldx, int M, int K, float* __restrict__ xf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M * K) xf[frag_index(i / K, i % K)] = x[(long)(i / K) * ldx + i % K];
}

// NB: column blocks of 16 scenes (M <= 16 NB); RT: 16-row tiles of W per workgroup (the tiles share every activation fragment: the
// per-workgroup L2 traffic is the activations, M x K x 4 bytes, so wide outputs take more rows per workgroup); JB: k-steps of a wave whose
// loads are in flight together (one memory round trip per batch of JB: K = 768 is ONE batch of 3 per wave, K = 3072 two of 6).
// LN: weight-only LayerNorm of every scene's row first (module.py:26-37) -- the statistics come out of the SAME fragments the wave
// multiplies (its k-steps' partial sums, folded over the 4 k-groups of a column and over the 8 waves in LDS, in a fixed order), so
// the activations are read once; needs every k-step of the wave in registers: K <= 256 JB.
template <typename TT, int MODE, int NB, int RT, int JB, bool LN>
__global__ __launch_bounds__(kRowsThreads) void rows_mfma_kernel(RowsArgs a) {
    typedef typename Mma16<TT>::vec vec;
    __shared__ float s_red[2][kRowsWaves][NB][16];
    __shared__ __attribute__((aligned(16))) float s_part[kRowsWaves][NB * RT][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int M = a.M, K = a.K;
    const int n0 = blockIdx.x * 16 * RT;
    const int row = lane & 15, kg = lane >> 4;
    const TT* W = reinterpret_cast<const TT*>(a.W);
    const int nsteps = K >> 5;                 // 32 k per MFMA (launcher: K % 32 == 0)
    const TT* wrow[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) wrow[t] = W + (long)min(n0 + 16 * t + row, a.N - 1) * K + 8 * kg;
    // everything the workgroup needs goes out in ONE round trip: the weight fragments, the activation fragments and LayerNorm weights of
    // the wave's first batch of k-steps, the epilogue's cache position / bias / residual values
    vec wf[RT][JB];
    float4 xa[JB][NB][2], lw[JB][2];
    auto req = [&](int j0) {
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            const int s = min(wave + 8 * (j0 + j), nsteps - 1);
#pragma unroll
            for (int t = 0; t < RT; ++t) wf[t][j] = *reinterpret_cast<const vec*>(wrow[t] + 32 * s);
            if (LN) {
                lw[j][0] = *reinterpret_cast<const float4*>(a.ln_w + 32 * s + 8 * kg);
                lw[j][1] = *reinterpret_cast<const float4*>(a.ln_w + 32 * s + 8 * kg + 4);
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const float* xp = a.x + ((((long)s * kRowsMaxNB + nb) * 64 + lane) << 3);
                xa[j][nb][0] = *reinterpret_cast<const float4*>(xp);
                xa[j][nb][1] = *reinterpret_cast<const float4*>(xp + 4);
            }
        }
    };
    req(0);
    const int pos = (MODE == ROWS_QKV && a.d_len) ? *a.d_len : 0;
    constexpr int kEpi = (RT * NB * 256 + kRowsThreads - 1) / kRowsThreads;     // outputs per thread
    float bias_e[kEpi];
#pragma unroll
    for (int q = 0; q < kEpi; ++q) {
        const int e = tid + q * kRowsThreads;
        const int n = n0 + 16 * ((e >> 8) / NB) + (e & 15);
        bias_e[q] = (a.bias && n < a.N) ? a.bias[n] : 0.f;
        if (MODE == ROWS_RESID) {      // the residual value it will be added to (read by this thread only): same round trip
            const int m = 16 * ((e >> 8) % NB) + ((e >> 4) & 15);
            if (e < RT * NB * 256 && m < M && n < a.N) bias_e[q] += a.out[(long)m * a.ldo + n];
        }
    }
    if (LN) {
        // statistics of scene (16 nb + l % 16) from this wave's fragments: sum over its k-steps and the lane's 8 values, then over the 4
        // k-groups of the column (lanes l, l ^ 16, l ^ 32, l ^ 48), then over the 8 waves through LDS -- twice (mean, then squared deviations)
        float part[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < JB; ++j)
                if (wave + 8 * j < nsteps)
                    sum += ((xa[j][nb][0].x + xa[j][nb][0].y) + (xa[j][nb][0].z + xa[j][nb][0].w)) + ((xa[j][nb][1].x + xa[j][nb][1].y) + (xa[j][nb][1].z + xa[j][nb][1].w));
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            part[nb] = sum;
        }
        if (kg == 0) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) s_red[0][wave][nb][row] = part[nb];
        }
        __syncthreads();
        float mean[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float t = s_red[0][0][nb][row];
#pragma unroll
            for (int w = 1; w < kRowsWaves; ++w) t += s_red[0][w][nb][row];
            mean[nb] = t / (float)K;
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < JB; ++j)
                if (wave + 8 * j < nsteps) {
                    const float4 u0 = xa[j][nb][0], u1 = xa[j][nb][1];
                    const float d0 = u0.x - mean[nb], d1 = u0.y - mean[nb], d2 = u0.z - mean[nb], d3 = u0.w - mean[nb];
                    const float d4 = u1.x - mean[nb], d5 = u1.y - mean[nb], d6 = u1.z - mean[nb], d7 = u1.w - mean[nb];
                    q += ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) + ((d4 * d4 + d5 * d5) + (d6 * d6 + d7 * d7));
                }
            q += __shfl_xor(q, 16);
            q += __shfl_xor(q, 32);
            part[nb] = q;
        }
        if (kg == 0) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) s_red[1][wave][nb][row] = part[nb];
        }
        __syncthreads();
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float t = s_red[1][0][nb][row];
#pragma unroll
            for (int w = 1; w < kRowsWaves; ++w) t += s_red[1][w][nb][row];
            const float rstd = 1.0f / sqrtf(t / (float)K + 1e-5f);
#pragma unroll
            for (int j = 0; j < JB; ++j) {
                float4& u0 = xa[j][nb][0];
                float4& u1 = xa[j][nb][1];
                u0.x = (u0.x - mean[nb]) * rstd * lw[j][0].x; u0.y = (u0.y - mean[nb]) * rstd * lw[j][0].y;
                u0.z = (u0.z - mean[nb]) * rstd * lw[j][0].z; u0.w = (u0.w - mean[nb]) * rstd * lw[j][0].w;
                u1.x = (u1.x - mean[nb]) * rstd * lw[j][1].x; u1.y = (u1.y - mean[nb]) * rstd * lw[j][1].y;
                u1.z = (u1.z - mean[nb]) * rstd * lw[j][1].z; u1.w = (u1.w - mean[nb]) * rstd * lw[j][1].w;
            }
        }
    }
    f32x4_t acc[RT][NB];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[t][nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int j0 = 0; wave + 8 * j0 < nsteps; j0 += JB) {
        if (j0 > 0) req(j0);
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            if (wave + 8 * (j0 + j) < nsteps) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const float v[8] = {xa[j][nb][0].x, xa[j][nb][0].y, xa[j][nb][0].z, xa[j][nb][0].w, xa[j][nb][1].x, xa[j][nb][1].y, xa[j][nb][1].z, xa[j][nb][1].w};
                    vec hi, lo;
                    split8<TT>(v, hi, lo);
#pragma unroll
                    for (int t = 0; t < RT; ++t) {
                        acc[t][nb] = Mma16<TT>::mfma(wf[t][j], hi, acc[t][nb]);
                        acc[t][nb] = Mma16<TT>::mfma(wf[t][j], lo, acc[t][nb]);
                    }
                }
            }
        }
    }
    // the 8 waves' partial sums of output (row r, scene c): lane holds rows 4 kg .. + 3 of column `row` (its l % 16)
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) s_part[wave][t * NB + nb][(lane & 15) * 16 + 4 * kg + r] = acc[t][nb][r];     // [scene column][weight row]
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kEpi; ++q) {
        const int e = tid + q * kRowsThreads;
        if (e >= RT * NB * 256) break;
        const int tb = e >> 8, col = (e >> 4) & 15, r = e & 15;
        const int t = tb / NB, nb = tb % NB;
        const int m = 16 * nb + col, n = n0 + 16 * t + r;
        if (m >= M || n >= a.N) continue;
        float v = s_part[0][tb][col * 16 + r];
#pragma unroll
        for (int w = 1; w < kRowsWaves; ++w) v += s_part[w][tb][col * 16 + r];      // fixed order: wave 0, 1, .. 7
        v += bias_e[q];
        if (MODE == ROWS_QKV) {
            if (n < a.E) a.out[(long)m * a.ldo + n] = v;
            else {
                const int c = n - a.E, kvsel = c / a.E, hc = c % a.E;   // kvsel 0 = K, 1 = V; head-major cache [2][H][Lmax][48]
                const long H = a.E / kHeadDim;
                reinterpret_cast<TT*>(a.cache)[(long)m * a.scene_stride + ((kvsel * H + hc / kHeadDim) * a.Lmax + pos) * kHeadDim + hc % kHeadDim] =
                    Cvt<TT>::from_f(v);
            }
        } else if (MODE == ROWS_GELU) {
            a.out_frag[frag_index(m, n)] = gelu_erf(v);       // consumed by the mlp c_proj launch only
        } else if (MODE == ROWS_RESID) {
            a.out[(long)m * a.ldo + n] = v;                   // (x + bias were added above)
            a.out_frag[frag_index(m, n)] = v;                 // the copy the next LayerNorm + GEMM launch streams
        } else {
            a.out[(long)m * a.ldo + n] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// decode attention, one query per (scene, head), keys 0 .. L (the new token's own row was written by the q | k | v launch)
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int kABThreads = 256, kABWaves = 4, kABFlight = 4;
constexpr float kScaleQKb = 0.14433756729740643f;   // float32(1 / sqrt(48)), module.py:196-198

template <typename TT>
__global__ __launch_bounds__(kABThreads) void attn_decode_batched_kernel(const float* __restrict__ q, const TT* __restrict__ cache, long scene_stride,
                                                                        int H, int Lmax, const int* __restrict__ d_len, float* __restrict__ y) {
    __shared__ float s_m[64], s_l[64], s_o[64][kHeadDim];
    const int h = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int part = lane & 3, grp = lane >> 2;              // 4 lanes per key: dims 12 part .. + 11; 16 keys per wave pass
    const int E = H * kHeadDim;
    const int L = *d_len + 1;
    const TT* kb = cache + (long)b * scene_stride + (long)h * Lmax * kHeadDim + 12 * part;
    const TT* vb = kb + (long)H * Lmax * kHeadDim;
    float qv[12];
    {
        const float* qp = q + (long)b * E + h * kHeadDim + 12 * part;
#pragma unroll
        for (int d = 0; d < 12; d += 4) load4(qp + d, *reinterpret_cast<float(*)[4]>(&qv[d]));
    }
    float m = -INFINITY, l = 0.f, o[12];
#pragma unroll
    for (int d = 0; d < 12; ++d) o[d] = 0.f;
    // pass p of this wave: keys 16 (wave + 4 p) + grp; kABFlight passes requested before the first is used
    const int npass = (L + 63) / 64;     // passes per wave (the last ones may be partly or wholly past L: masked)
    for (int p0 = 0; p0 < npass; p0 += kABFlight) {
        float kf[kABFlight][12], vf[kABFlight][12];
#pragma unroll
        for (int f = 0; f < kABFlight; ++f) {
            const int key = min(16 * (wave + 4 * (p0 + f)) + grp, L - 1);
            const TT* kp = kb + (long)key * kHeadDim;
            const TT* vp = vb + (long)key * kHeadDim;
#pragma unroll
            for (int d = 0; d < 12; d += 4) {
                load4(kp + d, *reinterpret_cast<float(*)[4]>(&kf[f][d]));
                load4(vp + d, *reinterpret_cast<float(*)[4]>(&vf[f][d]));
            }
        }
#pragma unroll
        for (int f = 0; f < kABFlight; ++f) {
            const int key = 16 * (wave + 4 * (p0 + f)) + grp;
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < 12; ++d) s = fmaf(qv[d], kf[f][d], s);
            s += dpp_xor1(s);
            s += dpp_xor2(s);
            if (key < L) {
                s *= kScaleQKb;
                const float mn = fmaxf(m, s);
                const float alpha = __expf(m - mn), pr = __expf(s - mn);
                m = mn;
                l = l * alpha + pr;
#pragma unroll
                for (int d = 0; d < 12; ++d) o[d] = fmaf(pr, vf[f][d], o[d] * alpha);
            }
        }
    }
    // merge the workgroup's 64 lane-group states (16 per wave) in a fixed order
    const int g = wave * 16 + grp;
    if (part == 0) { s_m[g] = m; s_l[g] = l; }
#pragma unroll
    for (int d = 0; d < 12; ++d) s_o[g][12 * part + d] = o[d];
    __syncthreads();
    if (tid < kHeadDim) {
        float mx = -INFINITY;
        for (int i = 0; i < 64; ++i) mx = fmaxf(mx, s_m[i]);
        float lt = 0.f, ot = 0.f;
        for (int i = 0; i < 64; ++i) {
            const float w = s_m[i] == -INFINITY ? 0.f : __expf(s_m[i] - mx);
            lt = fmaf(s_l[i], w, lt);
            ot = fmaf(s_o[i][tid], w, ot);
        }
        y[frag_index(b, h * kHeadDim + tid)] = ot / lt;     // fragment-major: the c_proj launch streams it
    }
}


}  // namespace

template <typename TT, int MODE, int NB, int RT, int JB, bool LN>
static void launch_rows_one(hipStream_t s, const RowsArgs& a) {
    const dim3 grid((a.N + 16 * RT - 1) / (16 * RT)), block(kRowsThreads);
    hipLaunchKernelGGL((rows_mfma_kernel<TT, MODE, NB, RT, JB, LN>), grid, block, 0, s, a);
}
template <typename TT, int NB>
static void launch_rows_nb(hipStream_t s, const RowsArgs& a) {
    // rows per workgroup by output width (the projections: 16; q|k|v and c_fc: 32; the heads: 64 -- fewer from 3 column blocks on, where
    // the wider tiles would spill registers); k-steps in flight by K (K = 3072: 6 up to 32 scenes, 3 beyond)
    switch (a.mode) {
        case ROWS_QKV: launch_rows_one<TT, ROWS_QKV, NB, (NB <= 3 ? 2 : 1), 3, true>(s, a); break;
        case ROWS_GELU: launch_rows_one<TT, ROWS_GELU, NB, (NB <= 3 ? 2 : 1), 3, true>(s, a); break;
        case ROWS_RESID:
            if (a.K > 768) launch_rows_one<TT, ROWS_RESID, NB, 1, (NB <= 2 ? 6 : 3), false>(s, a); else launch_rows_one<TT, ROWS_RESID, NB, 1, 3, false>(s, a);
            break;
        default: launch_rows_one<TT, ROWS_F32, NB, (NB <= 2 ? 4 : 2), 3, true>(s, a); break;
    }
}
template <typename TT>
void launch_rows_mfma(hipStream_t s, const RowsArgs& a) {
    switch ((a.M + 15) / 16) {
        case 1: launch_rows_nb<TT, 1>(s, a); break;
        case 2: launch_rows_nb<TT, 2>(s, a); break;
        case 3: launch_rows_nb<TT, 3>(s, a); break;
        default: launch_rows_nb<TT, 4>(s, a); break;
    }
}
void launch_rows_to_frag(hipStream_t s, const float* x, long ldx, int M, int K, float* xf) {
    hipLaunchKernelGGL(rows_to_frag_kernel, dim3((M * K + 255) / 256), dim3(256), 0, s, x, ldx, M, K, xf);
}
template void launch_rows_mfma<bf16_t>(hipStream_t, const RowsArgs&);
template void launch_rows_mfma<f16_t>(hipStream_t, const RowsArgs&);

template <typename TT>
void launch_attn_decode_batched(hipStream_t s, const float* q, const TT* cache, long scene_stride, int B, int H, int Lmax, const int* d_len, float* y) {
    hipLaunchKernelGGL((attn_decode_batched_kernel<TT>), dim3(H, B), dim3(kABThreads), 0, s, q, cache, scene_stride, H, Lmax, d_len, y);
}
template void launch_attn_decode_batched<bf16_t>(hipStream_t, const float*, const bf16_t*, long, int, int, int, const int*, float*);
template void launch_attn_decode_batched<f16_t>(hipStream_t, const float*, const f16_t*, long, int, int, int, const int*, float*);

}  // namespace umgen
