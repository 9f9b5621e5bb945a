// XCD-local seams (gfx950): can the five all-to-all hand-offs of a decode layer be kept inside ONE XCD, whose L2 is coherent
// for its own 32 CUs, so that only the layer-to-layer hand-off of x crosses the fabric?
//
//   test A  "xcd-local layer": the 32 workgroups that landed on XCD 0 run 36 layers x 5 seams among themselves; in-XCD edges
//           are PLAIN 8-byte {tag, value} stores (they stay in the XCD's L2) read back with sc1 loads (bypass L1, L2-served).
//   test B  "layer pipeline": layer l runs on XCD l % D; its four inner seams are XCD-local as in A, the x vector of the next
//           layer is published write-through (sc1) and gathered by the next XCD's workgroups.
//   test C  streaming bandwidth of ONE XCD (32 workgroups, non-temporal 16-byte loads) and of all 8.
// Workgroups find their XCD with s_getreg_b32 HW_REG_XCC_ID and take a rank inside it from a per-XCD ticket, so nothing
// depends on the blockIdx -> XCD mapping; the run aborts (census printed) if an XCD did not get exactly 32 workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned long long u64;
typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

constexpr int kLayers = 36;
constexpr u32 kSpinLimit = 400000;
constexpr int kNS = 8;
constexpr int kPart = 16 * kNS * 50;

struct Args {
    u64* gx;                 // [768] cross-XCD (sc1)
    u64* loc;                // per XCD: gqkv[2304] | gpart[kPart] | gxb[768] | gh[3072]
    u32* err;                // [0] give-up code, [1] mismatches
    u32* census;             // [8] tickets per XCD, [8] arrivals total, [9..] scratch
    u32 base;
    int D;                   // XCDs taking part in the layer pipeline (test B); 0 = test A (XCD 0 only)
    int local_sc1;           // 1: in-XCD edges also written with sc1 stores (control experiment)
    unsigned long long* stamps;
};
constexpr int kLocStride = 2304 + kPart + 768 + 3072;

__device__ inline u32 xcc_id() {
    u32 x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 15u;
}
__device__ inline void put_sc1(u64* g, u32 epoch, float v) {
    __hip_atomic_store(g, ((u64)epoch << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void put_plain(u64* g, u32 epoch, float v) {
    __hip_atomic_store(g, ((u64)epoch << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ inline void put(u64* g, u32 epoch, float v, int sc1) { if (sc1) put_sc1(g, epoch, v); else put_plain(g, epoch, v); }

template <int PER>
__device__ inline void gather(const u64* g, int n, u32 epoch, float* lds, u32* err, bool& failed) {
    const int tid = threadIdx.x;
    u32 got = 0, need = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k)
        if (tid + k * 256 < n) need |= 1u << k;
    if (!failed) {
        for (u32 spins = 0;;) {
            u64 v[PER];
#pragma unroll
            for (int k = 0; k < PER; ++k)
                if (((need & ~got) >> k) & 1u) v[k] = __hip_atomic_load(g + tid + k * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int k = 0; k < PER; ++k)
                if (((need & ~got) >> k) & 1u) {
                    if ((u32)(v[k] >> 32) == epoch) { lds[tid + k * 256] = __uint_as_float((u32)v[k]); got |= 1u << k; }
                }
            if (!__any(got != need)) break;
            if (++spins > kSpinLimit) { if ((tid & 63) == 0) atomicExch(err, epoch); failed = true; break; }
            if ((spins & 127u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { failed = true; break; }
        }
    }
    __syncthreads();
}

__device__ inline float expect(int l, int e, int i) { return (float)((l * 8 + e) * 8192 + i); }

__global__ __launch_bounds__(256) void xcd_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ u32 s_rank;
    const int tid = threadIdx.x;
    const u32 xcc = xcc_id();
    if (tid == 0) {
        s_rank = atomicAdd(a.census + xcc, 1u);
        __hip_atomic_fetch_add(a.census + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int w = (int)s_rank;
    const int G = 32;
    if (w >= G) return;                                 // uneven placement: the host sees it in the census
    const int D = a.D ? a.D : 1;
    if ((int)xcc >= D) return;
    bool failed = false;
    u32 bad = 0;
    u64* gqkv = a.loc + (long)xcc * kLocStride;
    u64* gpart = gqkv + 2304;
    u64* gxb = gpart + kPart;
    u64* gh = gxb + 768;
    const int sc = a.local_sc1;
    unsigned long long t_prev = 0, acc[5] = {0, 0, 0, 0, 0};
    const bool timer = (xcc == 0 && w == 0 && tid == 0);
    if (timer) t_prev = wall_clock64();
    for (int l = (int)xcc; l < kLayers; l += D) {
        const u32 eb = a.base + l * 8;
        if (l > 0) {
            gather<3>(a.gx, 768, eb + 0, lds, a.err, failed);
            if (!failed && tid < 3) bad += lds[tid * 255 + w % 3] != expect(l, 0, tid * 255 + w % 3);
        }
        if (timer) { const unsigned long long t = wall_clock64(); acc[0] += t - t_prev; t_prev = t; }
        for (int i = w + G * tid; i < 2304; i += G * 256) put(gqkv + i, eb + 1, expect(l, 1, i), sc);
        for (int u = w; u < 16 * kNS; u += G) {
            const int h = u % 16;
            if (!failed) {
                const int src = (tid / 48) * 768 + h * 48 + tid % 48;
                u64 v = 0;
                for (u32 spins = 0;;) {
                    bool have = true;
                    if (tid < 144) { v = __hip_atomic_load(gqkv + src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); have = (u32)(v >> 32) == eb + 1; }
                    if (!__any(!have)) break;
                    if (++spins > kSpinLimit) { if ((tid & 63) == 0) atomicExch(a.err, eb + 1); failed = true; break; }
                    if ((spins & 127u) == 0 && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { failed = true; break; }
                }
                if (!failed && tid < 144) bad += __uint_as_float((u32)v) != expect(l, 1, src);
            }
            __syncthreads();
            if (tid < 50) put(gpart + u * 50 + tid, eb + 2, expect(l, 2, u * 50 + tid), sc);
        }
        if (timer) { const unsigned long long t = wall_clock64(); acc[1] += t - t_prev; t_prev = t; }
        gather<25>(gpart, kPart, eb + 2, lds, a.err, failed);
        if (!failed && tid < 8) bad += lds[(tid * 97 + w) % kPart] != expect(l, 2, (tid * 97 + w) % kPart);
        if (timer) { const unsigned long long t = wall_clock64(); acc[2] += t - t_prev; t_prev = t; }
        for (int i = w + G * tid; i < 768; i += G * 256) put(gxb + i, eb + 3, expect(l, 3, i), sc);
        gather<3>(gxb, 768, eb + 3, lds, a.err, failed);
        if (!failed && tid < 3) bad += lds[tid * 255 + w % 3] != expect(l, 3, tid * 255 + w % 3);
        if (timer) { const unsigned long long t = wall_clock64(); acc[3] += t - t_prev; t_prev = t; }
        for (int i = w + G * tid; i < 3072; i += G * 256) put(gh + i, eb + 4, expect(l, 4, i), sc);
        gather<12>(gh, 3072, eb + 4, lds, a.err, failed);
        if (!failed && tid < 8) bad += lds[(tid * 383 + w) % 3072] != expect(l, 4, (tid * 383 + w) % 3072);
        if (timer) { const unsigned long long t = wall_clock64(); acc[4] += t - t_prev; t_prev = t; }
        // next layer's x: crosses to another XCD unless D == 1 (then it is one more XCD-local edge, but it shares the buffer, so sc1)
        for (int i = w + G * tid; i < 768; i += G * 256) put_sc1(a.gx + i, eb + 8, expect(l + 1, 0, i));
    }
    if (bad) atomicAdd(a.err + 1, bad);
    if (timer)
        for (int p = 0; p < 5; ++p) a.stamps[p] += acc[p];
}

__global__ __launch_bounds__(256) void stream_kernel(const u32x4* W, long n16_per_wg, int only_xcc, u32* sink) {
    const u32 xcc = xcc_id();
    if (only_xcc >= 0 && (int)xcc != only_xcc) return;
    const u32x4* p = W + (long)blockIdx.x * n16_per_wg + threadIdx.x;
    u32 acc = 0;
    for (long i = 0; i + 8 * 256 <= n16_per_wg; i += 8 * 256) {
        u32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(p + i + k * 256);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    }
    if (acc == 0x12345u) sink[0] = acc;
}

static void run_seams(Args a, int D, int local_sc1, const char* tag) {
    a.D = D;
    a.local_sc1 = local_sc1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t lds_bytes = 96 << 10;
    hipFuncSetAttribute(reinterpret_cast<const void*>(xcd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    const int iters = 20;
    static u32 base = 16;
    hipMemset(a.err, 0, 8);
    u32 census[16] = {};
    for (int it = -2; it < iters; ++it) {
        if (it == 0) { hipMemset(a.stamps, 0, 64); hipDeviceSynchronize(); hipEventRecord(e0); }
        hipMemsetAsync(a.census, 0, 64, 0);
        a.base = base;
        base += 512;
        hipLaunchKernelGGL(xcd_kernel, dim3(256), dim3(256), lds_bytes, 0, a);
        if (it == -2) { hipDeviceSynchronize(); hipMemcpy(census, a.census, 64, hipMemcpyDeviceToHost); }
    }
    hipEventRecord(e1);
    hipError_t rc = hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    u32 err[2];
    hipMemcpy(err, a.err, 8, hipMemcpyDeviceToHost);
    unsigned long long st[8];
    hipMemcpy(st, a.stamps, 64, hipMemcpyDeviceToHost);
    const double per_layer = ms * 1e3 / (iters * kLayers);
    const int mine = D ? (kLayers + D - 1) / D : kLayers;    // layers timed by XCD 0's workgroup 0
    printf("%-34s D=%d: %7.2f us/layer (%7.1f us/step)  XCD0 phases[x,qkv,part,x',h] us:", tag, D, per_layer, per_layer * kLayers);
    for (int p = 0; p < 5; ++p) printf(" %.2f", (double)st[p] / 100.0 / (iters * mine));
    printf("  giveup=%u mismatches=%u rc=%d census=", err[0], err[1], (int)rc);
    for (int k = 0; k < 8; ++k) printf("%u ", census[k]);
    printf("\n");
    fflush(stdout);
}

int main() {
    Args a{};
    hipMalloc(&a.gx, 768 * 8);
    hipMemset(a.gx, 0, 768 * 8);
    hipMalloc(&a.loc, (size_t)8 * kLocStride * 8);
    hipMemset(a.loc, 0, (size_t)8 * kLocStride * 8);
    hipMalloc(&a.err, 8);
    hipMalloc(&a.census, 64);
    hipMalloc(&a.stamps, 64);
    run_seams(a, 0, 0, "A: xcd-local layer (plain stores)");
    run_seams(a, 0, 1, "A': xcd-local layer (sc1 stores)");
    run_seams(a, 2, 0, "B: layer pipeline");
    run_seams(a, 4, 0, "B: layer pipeline");
    run_seams(a, 8, 0, "B: layer pipeline");
    run_seams(a, 8, 1, "B': layer pipeline, sc1 everywhere");
    // C: streaming bandwidth
    const size_t bytes = (size_t)2 << 30;
    void* W;
    hipMalloc(&W, bytes);
    hipMemset(W, 1, bytes);
    u32* sink;
    hipMalloc(&sink, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int only = -1; only < 1; ++only)
        for (int wgs_per_cu = 1; wgs_per_cu <= 4; wgs_per_cu *= 2) {
            const int grid = 256 * wgs_per_cu;
            const long n16 = (long)(bytes / 16 / grid) / 2048 * 2048;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(stream_kernel, dim3(grid), dim3(256), 0, 0, (const u32x4*)W, n16, only, sink);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double moved = (double)n16 * 16 * (only < 0 ? grid : grid / 8);
            printf("C: stream %s, %d wg/CU: %.1f MB in %.3f ms = %.0f GB/s\n", only < 0 ? "all XCDs" : "XCD 0 only", wgs_per_cu, moved / 1e6, ms, moved / ms / 1e6);
        }
    return 0;
}
