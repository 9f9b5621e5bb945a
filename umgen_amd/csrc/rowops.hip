// Row-wise kernels: LayerNorm (module.py:26-37: weight only, eps 1e-5, fp32 statistics).
#include "kernels.h"

namespace umgen {

constexpr int kMaxPerLane = 24;   // E <= 1536

template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, long row_stride, long n_rows, int E,
                                                         const float* __restrict__ w, T* __restrict__ out) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + row * row_stride;
    float v[kMaxPerLane];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxPerLane; ++i) {
        const int c = lane + 64 * i;
        v[i] = (c < E) ? xr[c] : 0.f;
        s += v[i];
    }
    const float mean = wave_sum(s) / (float)E;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxPerLane; ++i) {
        const int c = lane + 64 * i;
        const float d = (c < E) ? v[i] - mean : 0.f;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)E + 1e-5f);
    T* o = out + row * (long)E;
#pragma unroll
    for (int i = 0; i < kMaxPerLane; ++i) {
        const int c = lane + 64 * i;
        if (c < E) o[c] = Cvt<T>::from_f((v[i] - mean) * rstd * w[c]);
    }
}

template <typename T>
void launch_layernorm(hipStream_t s, const float* x, long row_stride, long n_rows, int E, const float* w, T* out) {
    if (n_rows <= 0) return;
    hipLaunchKernelGGL(layernorm_kernel<T>, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, s, x, row_stride, n_rows, E, w, out);
}
template void launch_layernorm<float>(hipStream_t, const float*, long, long, int, const float*, float*);
template void launch_layernorm<bf16_t>(hipStream_t, const float*, long, long, int, const float*, bf16_t*);
template void launch_layernorm<f16_t>(hipStream_t, const float*, long, long, int, const float*, f16_t*);

}  // namespace umgen
