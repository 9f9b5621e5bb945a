// LayerNorm of one row by one wave (module.py:26-37: weight only, eps 1e-5, fp32 statistics): rowops.hip's kernel and the decode engine's background workers.
#pragma once
#include "kernels.h"

namespace umgen {

constexpr int kMaxPerLane = 24;   // E <= 1536

// one wave normalises row `row` (the kernel below: 4 rows per 256-thread block; the decode engine's background workers, bg_worker.h: 8 per workgroup)
template <typename T>
__device__ __forceinline__ void layernorm_row(const float* __restrict__ x, long row_stride, int E, const float* __restrict__ w, T* __restrict__ out, long row,
                                              int lane) {
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

// NR rows by one wave with all their loads in flight together (the background workers: 8 waves per CU instead of the stand-alone kernel's 32, so a wave
// has to carry the memory-level parallelism itself: 78 -> see DESIGN.md 5.9 ms of LayerNorm per worker and pass).  Per row exactly layernorm_row's
// arithmetic in its order: the same bits.  Rows past n_rows are skipped (their slots read row n_rows - 1 and store nothing).
template <typename T, int NR, int NPL>
__device__ __forceinline__ void layernorm_rows(const float* __restrict__ x, long row_stride, int E, const float* __restrict__ w, T* __restrict__ out,
                                               const long (&rows)[NR], long n_rows, int lane) {
    float v[NR][NPL];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const float* xr = x + (rows[r] < n_rows ? rows[r] : n_rows - 1) * row_stride;
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int c = lane + 64 * i;
            v[r][i] = (c < E) ? xr[c] : 0.f;
        }
    }
    float wv[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) { const int c = lane + 64 * i; wv[i] = (c < E) ? w[c] : 0.f; }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NPL; ++i) s += v[r][i];
        const float mean = wave_sum(s) / (float)E;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int c = lane + 64 * i;
            const float d = (c < E) ? v[r][i] - mean : 0.f;
            q += d * d;
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)E + 1e-5f);
        if (rows[r] < n_rows) {
            T* o = out + rows[r] * (long)E;
#pragma unroll
            for (int i = 0; i < NPL; ++i) {
                const int c = lane + 64 * i;
                if (c < E) o[c] = Cvt<T>::from_f((v[r][i] - mean) * rstd * wv[i]);
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, long row_stride, long n_rows, int E,
                                                         const float* __restrict__ w, T* __restrict__ out) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    layernorm_row<T>(x, row_stride, E, w, out, row, threadIdx.x & 63);
}

}  // namespace umgen
