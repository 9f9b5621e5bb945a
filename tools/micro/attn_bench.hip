// A/B bench of the spatial-attention kernel forms (umgen_amd/csrc/attn.hip) at the production shape: F history frames x H heads,
// S = 2207 tokens, head_dim 48, random bf16 operands.  Every variant's output is compared with variant 0's (the first form).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -o tools/micro/attn_bench tools/micro/attn_bench.hip
//   tools/micro/attn_bench [F] [variants...]
#define UMGEN_ATTN_VARIANTS 1
#include "../../umgen_amd/csrc/attn.hip"

#include <cmath>
#include <cstring>
#include <cstdio>
#include <random>
#include <vector>

using namespace umgen;

__host__ static float bf2f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; std::memcpy(&f, &u, 4); return f; }
__host__ static unsigned short f2bf(float f) { unsigned u; std::memcpy(&u, &f, 4); u += 0x7FFFu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }

int main(int argc, char** argv) {
    const int F = argc > 1 ? atoi(argv[1]) : 20, S = 2207, H = 16, E = H * 48, S_pad = ((S + 63) / 64) * 64;
    std::vector<int> variants;
    for (int i = 2; i < argc; ++i) variants.push_back(atoi(argv[i]));
    if (variants.empty()) variants = {0, 32, 1, 2, 3, 7, 11, 15, 19, 23, 27, 31, 0};
    const size_t nqk = (size_t)F * S * 2 * E, nvt = (size_t)F * H * 48 * S_pad, ny = (size_t)F * S * E;
    std::vector<unsigned short> hqk(nqk), hvt(nvt, 0), y0(ny), y1(ny);
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto& v : hqk) v = f2bf(nd(rng) * 1.5f);
    for (int fh = 0; fh < F * H; ++fh)
        for (int d = 0; d < 48; ++d)
            for (int s = 0; s < S; ++s) hvt[((size_t)fh * 48 + d) * S_pad + s] = f2bf(nd(rng));
    bf16_t *dqk, *dvt, *dy;
    hipMalloc(&dqk, nqk * 2); hipMalloc(&dvt, nvt * 2); hipMalloc(&dy, ny * 2);
    hipMemcpy(dqk, hqk.data(), nqk * 2, hipMemcpyHostToDevice);
    hipMemcpy(dvt, hvt.data(), nvt * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const double flops = 4.0 * (double)S * S * 48 * H * F;
    bool have0 = false;
    for (int v : variants) {
        g_attn_variant = v;
        hipMemset(dy, 0, ny * 2);
        for (int i = 0; i < 3; ++i) launch_attn_spatial_mfma<bf16_t>(nullptr, dqk, dvt, dy, F, S, S_pad, H);
        const int reps = 20;
        hipEventRecord(e0, nullptr);
        for (int i = 0; i < reps; ++i) launch_attn_spatial_mfma<bf16_t>(nullptr, dqk, dvt, dy, F, S, S_pad, H);
        hipEventRecord(e1, nullptr);
        if (hipEventSynchronize(e1) != hipSuccess) { printf("variant %d: launch failed\n", v); return 1; }
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1000.0 / reps;
        hipMemcpy(y1.data(), dy, ny * 2, hipMemcpyDeviceToHost);
        double maxd = 0.0, sumd = 0.0;
        if (!have0) { y0 = y1; have0 = true; }
        for (size_t i = 0; i < ny; ++i) {
            const double d = fabs((double)bf2f(y1[i]) - (double)bf2f(y0[i]));
            maxd = d > maxd ? d : maxd;
            sumd += d;
        }
        printf("variant %2d: %8.1f us  %7.1f TFLOP/s  (%.1f %% of 2.5 PF)   vs variant %d: max |d| %.3e  mean |d| %.3e\n", v, us,
               flops / us * 1e-6, flops / us * 1e-6 / 25.0, variants[0], maxd, sumd / ny);
        fflush(stdout);
    }
    return 0;
}
