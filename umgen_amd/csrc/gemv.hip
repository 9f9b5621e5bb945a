// Few-row linear layers for the OAR decode step and the ego decoder (M = scenes or 3*scenes rows).
// These are HBM-bound weight streams: each weight row is read exactly once per launch, 16 B per lane, fp32 accumulate,
// activations stay fp32 (only the stored weights / KV cache are bf16 in the bf16 precision mode).
// Replaces F.linear at module.py:206,229 (c_attn, c_proj), 246-248 (MLP) and the heads (UMGen.py:1062,1072,1087,1132)
// for single-token inputs, with LayerNorm (module.py:34-37), exact GELU and the residual add fused in.
#include "kernels.h"

namespace umgen {

constexpr int MB = 8;    // activation rows processed per pass
constexpr int RPW = 2;   // weight rows per wave

template <typename T>
__global__ __launch_bounds__(256) void gemv_kernel(GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float xs[];   // [MB][K]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.K;
    const T* W = reinterpret_cast<const T*>(a.W);
    const int pos = a.d_len ? *a.d_len : 0;
    const float* xbase = a.x + (a.d_xoff ? (long)(*a.d_xoff) * a.xoff_mul : 0L);
    for (int m0 = 0; m0 < a.M; m0 += MB) {
        const int mc = min(MB, a.M - m0);
        __syncthreads();
        // stage (and LayerNorm) the activation rows: one wave per row
        for (int m = wave; m < mc; m += 4) {
            const float* xr = xbase + (long)(m0 + m) * a.ldx;
            float* dst = xs + (long)m * K;
            if (a.ln_w) {
                float s = 0.f;
                for (int c = lane; c < K; c += 64) s += xr[c];
                const float mean = wave_sum(s) / (float)K;
                float q = 0.f;
                for (int c = lane; c < K; c += 64) { const float d = xr[c] - mean; q += d * d; }
                const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)K + 1e-5f);
                for (int c = lane; c < K; c += 64) dst[c] = (xr[c] - mean) * rstd * a.ln_w[c];
            } else {
                for (int c = lane; c < K; c += 64) dst[c] = xr[c];
            }
        }
        __syncthreads();
        const int n0 = (blockIdx.x * 4 + wave) * RPW;
#pragma unroll 1
        for (int r = 0; r < RPW; ++r) {
            const int n = n0 + r;
            if (n >= a.N) break;
            float acc[MB];
#pragma unroll
            for (int m = 0; m < MB; ++m) acc[m] = 0.f;
            const T* wr = W + (long)n * K;
            for (int c = lane * 8; c < K; c += 512) {
                float w8[8];
                load8(wr + c, w8);
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    if (m < mc) {
                        float x8[8];
                        load8(xs + (long)m * K + c, x8);
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[m] = fmaf(w8[e], x8[e], acc[m]);
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < MB; ++m) acc[m] = wave_sum(acc[m]);
            if (lane == 0) {
                const float b = a.bias ? a.bias[n] : 0.f;
                for (int m = 0; m < mc; ++m) {
                    const float v = acc[m] + b;
                    const int mm = m0 + m;
                    if (a.out_mode == GEMV_OUT_QKV) {
                        if (n < a.E) a.out[(long)mm * a.ldo + n] = v;
                        else reinterpret_cast<T*>(a.cache)[(long)mm * a.scene_stride + (long)pos * 2 * a.E + (n - a.E)] = Cvt<T>::from_f(v);
                    } else {
                        a.out[(long)mm * a.ldo + n] = (a.out_mode == GEMV_OUT_GELU) ? gelu_erf(v) : v;
                    }
                }
            }
        }
    }
}

template <typename T>
void launch_gemv(hipStream_t s, const GemvArgs& a) {
    const int grid = (a.N + 4 * RPW - 1) / (4 * RPW);
    const size_t shm = (size_t)MB * a.K * sizeof(float);
    hipLaunchKernelGGL(gemv_kernel<T>, dim3(grid), dim3(256), shm, s, a);
}
template void launch_gemv<float>(hipStream_t, const GemvArgs&);
template void launch_gemv<bf16_t>(hipStream_t, const GemvArgs&);

// x[m][n] += (sum_k a[m][k] W[n][k]) + bias[n]
template <typename T, bool COMBINE>
__global__ __launch_bounds__(256) void gemv_resid_kernel(GemvResidArgs a) {
    extern __shared__ __attribute__((aligned(16))) float as[];   // [MB][K] when COMBINE
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.K;
    const T* W = reinterpret_cast<const T*>(a.W);
    for (int m0 = 0; m0 < a.M; m0 += MB) {
        const int mc = min(MB, a.M - m0);
        if (COMBINE) {
            // merge the kAttnSplit partial softmax results of every (row, head):  (m, l, o[48]) -> o / l
            __syncthreads();
            for (int e = tid; e < mc * K; e += 256) {
                const int m = e / K, col = e % K;
                const int h = col / kHeadDim, d = col % kHeadDim;
                const float* p = a.part + (((long)(m0 + m) * a.H + h) * kAttnSplit) * kAttnPart;
                float mx = -INFINITY;
#pragma unroll
                for (int sp = 0; sp < kAttnSplit; ++sp) mx = fmaxf(mx, p[sp * kAttnPart]);
                float l = 0.f, o = 0.f;
#pragma unroll
                for (int sp = 0; sp < kAttnSplit; ++sp) {
                    const float w = expf(p[sp * kAttnPart] - mx);
                    l = fmaf(w, p[sp * kAttnPart + 1], l);
                    o = fmaf(w, p[sp * kAttnPart + 2 + d], o);
                }
                as[e] = o / l;
            }
            __syncthreads();
        }
        const int n0 = (blockIdx.x * 4 + wave) * RPW;
        float acc[RPW][MB];
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int m = 0; m < MB; ++m) acc[r][m] = 0.f;
        for (int c = lane * 8; c < K; c += 512) {
            float w8[RPW][8];
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int n = min(n0 + r, a.N - 1);
                load8(W + (long)n * K + c, w8[r]);
            }
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                if (m < mc) {
                    float x8[8];
                    if (COMBINE) load8(as + (long)m * K + c, x8);
                    else load8(a.a + (long)(m0 + m) * a.lda + c, x8);
#pragma unroll
                    for (int r = 0; r < RPW; ++r)
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[r][m] = fmaf(w8[r][e], x8[e], acc[r][m]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int m = 0; m < MB; ++m) acc[r][m] = wave_sum(acc[r][m]);
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int n = n0 + r;
                if (n < a.N) {
                    const float b = a.bias ? a.bias[n] : 0.f;
                    for (int m = 0; m < mc; ++m) a.x[(long)(m0 + m) * a.ldx + n] += acc[r][m] + b;
                }
            }
        }
    }
}

template <typename T>
void launch_gemv_resid(hipStream_t s, const GemvResidArgs& a) {
    const int grid = (a.N + 4 * RPW - 1) / (4 * RPW);
    if (a.part) {
        const size_t shm = (size_t)MB * a.K * sizeof(float);
        hipLaunchKernelGGL((gemv_resid_kernel<T, true>), dim3(grid), dim3(256), shm, s, a);
    } else {
        hipLaunchKernelGGL((gemv_resid_kernel<T, false>), dim3(grid), dim3(256), 0, s, a);
    }
}
template void launch_gemv_resid<float>(hipStream_t, const GemvResidArgs&);
template void launch_gemv_resid<bf16_t>(hipStream_t, const GemvResidArgs&);

}  // namespace umgen
