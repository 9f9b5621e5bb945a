// Round-trip latency of an 8-byte {tag, value} hand-off between every PAIR of XCDs (sc1 store -> sc1 poll -> sc1 store -> sc1 poll),
// one representative workgroup per XCD.  Question: are some XCD pairs closer than others (same IO die), so that the decode engine's
// layer -> XCD order could put cheap hops between consecutive layers?   hipcc --offload-arch=gfx950 -O3 -o xcd_pingpong xcd_pingpong.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef unsigned long long u64;
constexpr int N = 400;          // round trips per pair
__global__ __launch_bounds__(64) void pingpong(unsigned* ticket, u64* buf, unsigned long long* out, unsigned* err) {
    extern __shared__ float lds[];   // > 80 KB: one workgroup per CU
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    x &= 15u;
    __shared__ unsigned rank;
    if (threadIdx.x == 0) rank = atomicAdd(ticket + x, 1u);
    __syncthreads();
    if (rank != 0 || x >= 8 || threadIdx.x != 0) return;       // one thread of the first workgroup of each XCD plays
    const int me = (int)x;
    for (int a = 0; a < 8; ++a)
        for (int b = a + 1; b < 8; ++b) {
            if (me != a && me != b) continue;
            u64* pa = buf + (a * 8 + b) * 2;      // a -> b
            u64* pb = pa + 1;                     // b -> a
            unsigned long long t0 = 0;
            for (int i = 1; i <= N + 8; ++i) {
                if (i == 9) t0 = wall_clock64();
                if (me == a) {
                    __hip_atomic_store(pa, (u64)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    unsigned spins = 0;
                    while (__hip_atomic_load(pb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (u64)i)
                        if (++spins > 50000000u) { atomicExch(err, 1u); return; }
                } else {
                    unsigned spins = 0;
                    while (__hip_atomic_load(pa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (u64)i)
                        if (++spins > 50000000u) { atomicExch(err, 1u); return; }
                    __hip_atomic_store(pb, (u64)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (me == a) out[a * 8 + b] = wall_clock64() - t0;     // 100 MHz ticks for N round trips
        }
}
int main() {
    unsigned *ticket, *err; u64* buf; unsigned long long* out;
    hipMalloc(&ticket, 64); hipMemset(ticket, 0, 64);
    hipMalloc(&err, 4); hipMemset(err, 0, 4);
    hipMalloc(&buf, 64 * 2 * 8); hipMemset(buf, 0, 64 * 2 * 8);
    hipMalloc(&out, 64 * 8); hipMemset(out, 0, 64 * 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(pingpong), hipFuncAttributeMaxDynamicSharedMemorySize, 96 << 10);
    hipLaunchKernelGGL(pingpong, dim3(256), dim3(64), 96 << 10, 0, ticket, buf, out, err);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    unsigned long long h[64]; unsigned e;
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
    printf("round trip (two hand-offs) between XCD a (row) and b (column), ns%s\n      ", e ? "  [GAVE UP]" : "");
    for (int b = 0; b < 8; ++b) printf("%6d", b);
    printf("\n");
    for (int a = 0; a < 8; ++a) {
        printf("  %d : ", a);
        for (int b = 0; b < 8; ++b) {
            const unsigned long long t = a < b ? h[a * 8 + b] : (a > b ? h[b * 8 + a] : 0);
            if (a == b) printf("     -"); else printf("%6.0f", (double)t * 10.0 / N);
        }
        printf("\n");
    }
    return 0;
}
