// Do the matrix pipe and the VALU of one SIMD run concurrently when they are fed by DIFFERENT waves?  And by ONE wave?
// 512-thread workgroups (2 waves per SIMD), 1 workgroup per CU.  mode 0: waves 0-3 run NM independent-accumulator MFMAs, waves 4-7 exit;
// mode 1: waves 4-7 run NV v_fma_f32 (4 chains), waves 0-3 exit; mode 2: both; mode 3: every wave runs MFMA and VALU interleaved
// (1 MFMA : R VALU) at half the counts; mode 4: both kinds in every wave, phase-separated (all MFMAs, then all VALU).
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/coissue tools/micro/coissue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int MODE, int R>
__global__ __launch_bounds__(512) void k(float* out, int nm, int nv) {
    const int wave = threadIdx.x >> 6;
    bf16x8_t a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x & 3); b[j] = (__bf16)1.0f; }
    f32x4_t acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float v[4] = {1.f, 2.f, 3.f, 4.f};
    const float c = 1.0001f, d = 0.5f;
    const bool do_m = MODE == 0 ? wave < 4 : MODE == 1 ? false : MODE == 2 ? wave < 4 : true;
    const bool do_v = MODE == 0 ? false : MODE == 1 ? wave >= 4 : MODE == 2 ? wave >= 4 : true;
    if (MODE <= 2) {
        if (do_m)
            for (int i = 0; i < nm; i += 4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[q], 0, 0, 0);
            }
        if (do_v)
            for (int i = 0; i < nv; i += 16) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q]) : "v"(c), "v"(d));
            }
    } else if (MODE == 3) {
        for (int i = 0; i < nm / 2; i += 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[q], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < R; ++u) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[u & 3]) : "v"(c), "v"(d));
            }
        }
    } else {
        for (int i = 0; i < nm / 2; i += 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[q], 0, 0, 0);
        }
        for (int i = 0; i < nv / 2; i += 16) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q]) : "v"(c), "v"(d));
        }
    }
    float s = v[0] + v[1] + v[2] + v[3];
    for (int q = 0; q < 4; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int MODE, int R>
static void run(const char* name, float* d, int nm, int nv) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, R>), dim3(256), dim3(512), 0, nullptr, d, nm, nv);
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<MODE, R>), dim3(256), dim3(512), 0, nullptr, d, nm, nv);
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s %8.1f us\n", name, ms * 200.0);
}

int main() {
    float* d;
    hipMalloc(&d, 4096);
    const int nm = 1 << 16, nv = 1 << 18;   // 65536 MFMAs (16 cycles each) ~ 1.05 M cycles; 262144 v_fma (4 cycles each) ~ 1.05 M cycles
    run<0, 0>("MFMA waves only (1 wave / SIMD)", d, nm, nv);
    run<1, 0>("VALU waves only (1 wave / SIMD)", d, nm, nv);
    run<2, 0>("MFMA wave + VALU wave on every SIMD", d, nm, nv);
    run<3, 4>("2 waves / SIMD, each: 1 MFMA : 4 VALU interleaved", d, nm, nv);
    run<4, 0>("2 waves / SIMD, each: all its MFMAs, then all its VALU", d, nm, nv);
    run<3, 2>("2 waves / SIMD, each: 1 MFMA : 2 VALU (half the VALU)", d, nm, nv);
    run<3, 8>("2 waves / SIMD, each: 1 MFMA : 8 VALU (twice the VALU)", d, nm, nv);
    return 0;
}
