// MFMA issue-rate microbenchmark (gfx950): N independent v_mfma_f32_16x16x32_bf16 / 32x32x16 chains per wave, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int KIND>
__global__ __launch_bounds__(256) void spin(float* sink, int iters) {
    b8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (i + 1)); }
    if (KIND == 0) {
        f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
        }
        if (c0[0] + c1[0] + c2[0] + c3[0] == 12345.f) sink[0] = 1.f;
    } else {
        f16v c0 = {}, c1 = {};
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        }
        if (c0[0] + c1[0] == 12345.f) sink[0] = 1.f;
    }
}
int main() {
    float* d; hipMalloc(&d, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 200000;
    for (int kind = 0; kind < 2; ++kind)
        for (int wps = 1; wps <= 2; ++wps) {          // waves per SIMD: blocks of 256 threads = 1 wave per SIMD each
            dim3 grid(256 * wps);
            if (kind == 0) hipLaunchKernelGGL(spin<0>, grid, dim3(256), 0, 0, d, 1000); else hipLaunchKernelGGL(spin<1>, grid, dim3(256), 0, 0, d, 1000);
            hipEventRecord(e0);
            if (kind == 0) hipLaunchKernelGGL(spin<0>, grid, dim3(256), 0, 0, d, iters); else hipLaunchKernelGGL(spin<1>, grid, dim3(256), 0, 0, d, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)grid.x * 4 * iters * 4.0 * (kind == 0 ? 16.0 * 16 * 32 * 2 : 32.0 * 32 * 16 * 2);
            printf("%s waves/SIMD=%d: %.2f ms  %.0f TFLOP/s\n", kind == 0 ? "16x16x32" : "32x32x16", wps, ms, flops / ms / 1e9);
        }
    return 0;
}
