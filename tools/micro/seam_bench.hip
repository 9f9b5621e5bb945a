// Seam micro-benchmark for a persistent OAR decode step (gfx950): what do the five all-to-all hand-offs of a decode layer
// cost INSIDE one launch, as a function of how many workgroups take part?
//
// One launch = 36 "layers" x 5 phases.  A phase = every workgroup gathers the previous phase's output vector (8-byte
// {tag, value} granules written with one relaxed agent-scope store each, MI355X_MICROARCH.md "allgather"), then publishes its
// slice of the next vector.  Edge sizes are those of UMGen_Large's BlockOAR at one scene: x 768, q|k|v 2304 (an attention
// unit only reads its head's 144), attention partials 16 heads x NS splits x 50, x' 768, h 3072.
// STREAM adds the layer's weight stream (14.2 MB bf16 per layer) as register prefetch: the loads of phase p+1 are issued
// right after the gather of phase p and consumed after the gather of phase p+1.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o seam_bench seam_bench.hip ; run: ./seam_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;
typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

constexpr int kLayers = 36;
constexpr u32 kSpinLimit = 400000;

struct Args {
    u64 *gx, *gqkv, *gpart, *gxb, *gh;   // granule buffers
    u32* err;                            // [0] give-up code, [1] payload mismatches
    u32 base;                            // epoch base of this launch
    int G, NS;                           // workgroups, attention key splits per head
    const u32x4* W;                      // weight stream (STREAM)
    float* sink;
    unsigned long long* stamps;          // [8] accumulated 100 MHz ticks per phase (workgroup 0)
};

__device__ inline void put(u64* g, u32 epoch, float v) {
    __hip_atomic_store(g, ((u64)epoch << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the workgroup gathers granules [0, n) of g into lds[0, n); returns false after a give-up
template <int PER>
__device__ inline bool gather(const u64* g, int n, u32 epoch, float* lds, u32* err, bool& failed) {
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
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    return !failed;
}

__device__ inline float expect(int l, int e, int i) { return (float)((l * 8 + e) * 8192 + i); }

template <int N>
__device__ inline void wissue(u32x4 (&w)[N], const u32x4* p, long stride) {
#pragma unroll
    for (int k = 0; k < N; ++k) w[k] = __builtin_nontemporal_load(p + k * stride);
}
template <int N>
__device__ inline u32 wfold(const u32x4 (&w)[N]) {
    u32 a = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) a ^= w[k].x ^ w[k].y ^ w[k].z ^ w[k].w;
    return a;
}

// G workgroups of 256 threads; GW = compile-time G (register array sizes of the weight stream)
template <int GW, bool STREAM>
__global__ __launch_bounds__(256) void seam_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, w = blockIdx.x, G = a.G;
    bool failed = false;
    u32 bad = 0, wx = 0;
    const int nunits = 16 * a.NS;
    const int npart = nunits * 50;
    // per-thread 16-byte loads of each phase's weight slice (bytes / (G * 256 * 16))
    constexpr int LQ = STREAM ? (2304 * 768 * 2 + GW * 4096 - 1) / (GW * 4096) : 1;
    constexpr int LO = STREAM ? (768 * 768 * 2 + GW * 4096 - 1) / (GW * 4096) : 1;
    constexpr int LF = STREAM ? (3072 * 768 * 2 + GW * 4096 - 1) / (GW * 4096) : 1;
    u32x4 wq[LQ], wo[LO], wf[LF], w2[LF];
    const long stride = (long)G * 256;
    const long layer16 = (long)(12 * 768 * 768 * 2) / 16;
    const u32x4* wbase = a.W + (long)w * 256 + tid;
    unsigned long long t_prev = 0, acc[5] = {0, 0, 0, 0, 0};
    if (STREAM) wissue(wq, wbase, stride);
    if (w == 0 && tid == 0) t_prev = wall_clock64();
    for (int l = 0; l < kLayers; ++l) {
        const u32 eb = a.base + l * 8;
        const u32x4* wl = wbase + (long)l * layer16;
        // P1: gather x, publish q|k|v slice
        if (l > 0) {
            gather<3>(a.gx, 768, eb + 0, lds, a.err, failed);
            if (!failed && tid < 3) bad += lds[tid * 255 + w % 3] != expect(l, 0, tid * 255 + w % 3);
        }
        if (w == 0 && tid == 0) { const unsigned long long t = wall_clock64(); acc[0] += t - t_prev; t_prev = t; }
        if (STREAM) wissue(wo, wl + (2304L * 768 * 2) / 16, stride);
        if (STREAM) wx ^= wfold(wq);
        for (int i = w + G * tid; i < 2304; i += G * 256) put(a.gqkv + i, eb + 1, expect(l, 1, i) + (wx == 0xdeadbeefu ? 1.f : 0.f));
        // P2: attention units gather their head's q|k|v (144 values), publish 50 partial values
        for (int u = w; u < nunits; u += G) {
            const int h = u % 16;
            // q_h, k_h, v_h live at [h*48, +48), [768 + h*48, +48), [1536 + h*48, +48)
            bool ok = !failed;
            if (ok) {
                const int src = (tid / 48) * 768 + h * 48 + tid % 48;
                u64 v = 0;
                for (u32 spins = 0;;) {
                    bool have = true;
                    if (tid < 144) { v = __hip_atomic_load(a.gqkv + src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); have = (u32)(v >> 32) == eb + 1; }
                    if (!__any(!have)) break;
                    if (++spins > kSpinLimit) { if ((tid & 63) == 0) atomicExch(a.err, eb + 1); failed = true; break; }
                    if ((spins & 127u) == 0 && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { failed = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (!failed && tid < 144) bad += __uint_as_float((u32)v) != expect(l, 1, src);
            }
            __syncthreads();
            if (tid < 50) put(a.gpart + u * 50 + tid, eb + 2, expect(l, 2, u * 50 + tid));
        }
        if (w == 0 && tid == 0) { const unsigned long long t = wall_clock64(); acc[1] += t - t_prev; t_prev = t; }
        // P3: gather all partials, publish x' slice
        gather<25>(a.gpart, npart, eb + 2, lds, a.err, failed);
        if (!failed && tid < 8) bad += lds[(tid * 97 + w) % npart] != expect(l, 2, (tid * 97 + w) % npart);
        if (w == 0 && tid == 0) { const unsigned long long t = wall_clock64(); acc[2] += t - t_prev; t_prev = t; }
        if (STREAM) wissue(wf, wl + (3072L * 768 * 2) / 16, stride);
        if (STREAM) wx ^= wfold(wo);
        for (int i = w + G * tid; i < 768; i += G * 256) put(a.gxb + i, eb + 3, expect(l, 3, i) + (wx == 0xdeadbeefu ? 1.f : 0.f));
        // P4: gather x', publish h slice
        gather<3>(a.gxb, 768, eb + 3, lds, a.err, failed);
        if (!failed && tid < 3) bad += lds[tid * 255 + w % 3] != expect(l, 3, tid * 255 + w % 3);
        if (w == 0 && tid == 0) { const unsigned long long t = wall_clock64(); acc[3] += t - t_prev; t_prev = t; }
        if (STREAM) wissue(w2, wl + (7168L * 768 * 2) / 16, stride);
        if (STREAM) wx ^= wfold(wf);
        for (int i = w + G * tid; i < 3072; i += G * 256) put(a.gh + i, eb + 4, expect(l, 4, i) + (wx == 0xdeadbeefu ? 1.f : 0.f));
        // P5: gather h, publish next layer's x slice
        gather<12>(a.gh, 3072, eb + 4, lds, a.err, failed);
        if (!failed && tid < 8) bad += lds[(tid * 383 + w) % 3072] != expect(l, 4, (tid * 383 + w) % 3072);
        if (w == 0 && tid == 0) { const unsigned long long t = wall_clock64(); acc[4] += t - t_prev; t_prev = t; }
        if (STREAM && l + 1 < kLayers) wissue(wq, wl + layer16, stride);
        if (STREAM) wx ^= wfold(w2);
        for (int i = w + G * tid; i < 768; i += G * 256) put(a.gx + i, eb + 8, expect(l + 1, 0, i) + (wx == 0xdeadbeefu ? 1.f : 0.f));
    }
    if (bad) atomicAdd(a.err + 1, bad);
    if (wx == 0x12345u) a.sink[0] = 1.f;
    if (w == 0 && tid == 0)
        for (int p = 0; p < 5; ++p) a.stamps[p] += acc[p];
}

template <int GW, bool STREAM>
static void run(int NS, size_t lds_bytes, Args a, const char* tag) {
    a.G = GW;
    a.NS = NS;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(seam_kernel<GW, STREAM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    const int iters = 20;
    u32 base = 16;
    hipMemset(a.err, 0, 8);
    hipMemset(a.stamps, 0, 64);
    for (int it = -2; it < iters; ++it) {
        if (it == 0) { hipMemset(a.stamps, 0, 64); hipDeviceSynchronize(); hipEventRecord(e0); }
        a.base = base;
        base += 512;
        hipLaunchKernelGGL((seam_kernel<GW, STREAM>), dim3(GW), dim3(256), lds_bytes, 0, a);
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
    printf("%-28s G=%3d NS=%d lds=%3zuK: %7.2f us/layer (%7.1f us/step)  phases[x,qkv,part,x',h] us:", tag, GW, NS, lds_bytes >> 10, per_layer, per_layer * kLayers);
    for (int p = 0; p < 5; ++p) printf(" %.2f", (double)st[p] / 100.0 / (iters * kLayers));
    printf("  giveup=%u mismatches=%u rc=%d\n", err[0], err[1], (int)rc);
    fflush(stdout);
}

int main(int argc, char** argv) {
    Args a{};
    hipMalloc(&a.gx, 768 * 8);
    hipMalloc(&a.gqkv, 2304 * 8);
    hipMalloc(&a.gpart, 16 * 18 * 50 * 8);
    hipMalloc(&a.gxb, 768 * 8);
    hipMalloc(&a.gh, 3072 * 8);
    hipMemset(a.gx, 0, 768 * 8);
    hipMemset(a.gqkv, 0, 2304 * 8);
    hipMemset(a.gpart, 0, 16 * 18 * 50 * 8);
    hipMemset(a.gxb, 0, 768 * 8);
    hipMemset(a.gh, 0, 3072 * 8);
    hipMalloc(&a.err, 8);
    hipMalloc(&a.sink, 64);
    hipMalloc(&a.stamps, 64);
    const size_t wbytes = (size_t)kLayers * 12 * 768 * 768 * 2 + (64 << 20);
    void* W;
    hipMalloc(&W, wbytes);
    hipMemset(W, 1, wbytes);
    a.W = (const u32x4*)W;
    const size_t big = 96 << 10, small = 64 << 10;   // > 80 KB: one workgroup per CU
    const int NS = argc > 1 ? atoi(argv[1]) : 8;
    run<256, false>(NS, big, a, "seams only");
    run<128, false>(NS, big, a, "seams only");
    run<64, false>(NS, big, a, "seams only");
    run<32, false>(NS, big, a, "seams only");
    run<256, false>(4, big, a, "seams only");
    run<64, false>(4, big, a, "seams only");
    run<256, false>(NS, small, a, "seams only (2 wg/CU ok)");
    run<256, true>(NS, big, a, "seams + weight stream");
    run<128, true>(NS, big, a, "seams + weight stream");
    run<64, true>(NS, big, a, "seams + weight stream");
    run<64, true>(4, big, a, "seams + weight stream");
    return 0;
}
