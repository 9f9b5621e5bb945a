// Follow-up of xcd_pingpong.hip: the round trip of a hand-off between two XCDs as a function of WHERE the 8-byte granule lives
// (byte offset inside one allocation): is the latency a property of the XCD pair, or of the pair and the granule's home channel?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
constexpr int N = 200;
__global__ __launch_bounds__(64) void pp(unsigned* ticket, char* base, const long* offs, int n_off, int xa, int xb, unsigned long long* out, unsigned* err) {
    extern __shared__ float lds[];
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    x &= 15u;
    __shared__ unsigned rank;
    if (threadIdx.x == 0) rank = atomicAdd(ticket + x, 1u);
    __syncthreads();
    if (rank != 0 || threadIdx.x != 0 || ((int)x != xa && (int)x != xb)) return;
    for (int o = 0; o < n_off; ++o) {
        u64* pa = reinterpret_cast<u64*>(base + offs[o]);
        u64* pb = pa + 1;
        unsigned long long t0 = 0;
        for (int i = 1; i <= N + 8; ++i) {
            if (i == 9) t0 = wall_clock64();
            if ((int)x == xa) {
                __hip_atomic_store(pa, (u64)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned spins = 0;
                while (__hip_atomic_load(pb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (u64)i) if (++spins > 50000000u) { atomicExch(err, 1u); return; }
            } else {
                unsigned spins = 0;
                while (__hip_atomic_load(pa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (u64)i) if (++spins > 50000000u) { atomicExch(err, 1u); return; }
                __hip_atomic_store(pb, (u64)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if ((int)x == xa) out[o] = wall_clock64() - t0;
    }
}
int main(int argc, char** argv) {
    const int xa = argc > 1 ? atoi(argv[1]) : 4, xb = argc > 2 ? atoi(argv[2]) : 6;
    const long step = argc > 3 ? atol(argv[3]) : 256, n = argc > 4 ? atol(argv[4]) : 64;
    unsigned *ticket, *err; char* base; long* offs; unsigned long long* out;
    hipMalloc(&ticket, 64); hipMalloc(&err, 4);
    hipMalloc(&base, step * n + 4096); hipMemset(base, 0, step * n + 4096);
    hipMalloc(&offs, n * 8); hipMalloc(&out, n * 8); hipMemset(out, 0, n * 8);
    long* ho = (long*)malloc(n * 8);
    for (long i = 0; i < n; ++i) ho[i] = i * step;
    hipMemcpy(offs, ho, n * 8, hipMemcpyHostToDevice);
    hipMemset(ticket, 0, 64); hipMemset(err, 0, 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(pp), hipFuncAttributeMaxDynamicSharedMemorySize, 96 << 10);
    hipLaunchKernelGGL(pp, dim3(256), dim3(64), 96 << 10, 0, ticket, base, offs, (int)n, xa, xb, out, err);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    unsigned long long* h = (unsigned long long*)malloc(n * 8);
    hipMemcpy(h, out, n * 8, hipMemcpyDeviceToHost);
    printf("XCD %d <-> %d, granule at offset k x %ld bytes: round trip ns\n", xa, xb, step);
    for (long i = 0; i < n; ++i) printf("%5.0f%s", (double)h[i] * 10.0 / N, (i % 16 == 15) ? "\n" : " ");
    printf("\n");
    return 0;
}
