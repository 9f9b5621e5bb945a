// Kernel-level test hooks of libumgen_hip.so (host pointers in, host pointers out).  Used only by tests/ to pin each
// HIP kernel against the CPU oracle at production width; never called by the product path.
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "../../include/umgen.h"
#include "frame.h"
#include "kernels.h"

using namespace umgen;

namespace {
struct DevBuf {
    void* p = nullptr;
    explicit DevBuf(size_t bytes) { if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) p = nullptr; }
    ~DevBuf() { if (p) (void)hipFree(p); }
};
inline int up(void* d, const void* h, size_t n) { return hipMemcpy(d, h, n, hipMemcpyHostToDevice) == hipSuccess ? 0 : UMGEN_E_HIP; }
inline int down(void* h, const void* d, size_t n) { return hipMemcpy(h, d, n, hipMemcpyDeviceToHost) == hipSuccess ? 0 : UMGEN_E_HIP; }
}  // namespace

extern "C" {

// out[R][N] = act[R][K] . W[N][K]^T + bias (+gelu) (+ residual into out when resid != 0).  bf16 != 0: operands are raw
// bf16 bits (bf16 == 2: IEEE half bits) and the MFMA kernel runs; else fp32 operands and the exact VALU kernel.  out is fp32 for resid, operand dtype otherwise.
int umgen_dbg_linear(int flags, const void* act, const void* W, const float* bias, int R, int N, int K, int gelu, int resid, void* out) {
    const int bf16 = flags & 3;                 // precision code; flag 16: force the 256 x 256 kernel, 32: never use it
    const size_t es = bf16 ? 2 : 4;
    DevBuf dA((size_t)R * K * es), dW((size_t)N * K * es), dB((size_t)N * 4), dO((size_t)R * N * 4);
    if (!dA.p || !dW.p || !dB.p || !dO.p) return UMGEN_E_NOMEM;
    if (up(dA.p, act, (size_t)R * K * es) || up(dW.p, W, (size_t)N * K * es)) return UMGEN_E_HIP;
    if (bias && up(dB.p, bias, (size_t)N * 4)) return UMGEN_E_HIP;
    const size_t osz = (size_t)R * N * (resid ? 4 : es);
    if (resid && up(dO.p, out, osz)) return UMGEN_E_HIP;
    GemmArgs g{};
    g.P = dW.p; g.Q = dA.p; g.Mi = N; g.Nj = R; g.K = K; g.ldp = K; g.ldq = K; g.batch = 1;
    g.mode = resid ? GEMM_RESID : GEMM_STORE; g.bias = bias ? (const float*)dB.p : nullptr; g.gelu = gelu; g.out = dO.p; g.ldo = N;
    g.tile256 = (flags & 16) ? 1 : ((flags & 32) ? -1 : 0);
    if (bf16 == 2) launch_gemm_mfma<f16_t>(nullptr, g); else if (bf16) launch_gemm_mfma<bf16_t>(nullptr, g); else launch_gemm_valu<float, float>(nullptr, g);
    if (hipDeviceSynchronize() != hipSuccess) return UMGEN_E_HIP;
    return down(out, dO.p, osz);
}

// V^T GEMM of the spatial attention (GEMM_VT): act [F*S][K] rows, W [N][K], bias [N] -> out [F][N][S_pad] of the operand type
// (S_pad = S rounded up to 64; pad columns are zero).  flag 16: force the 256 x 256 kernel, 32: the 128-tile kernels only.
int umgen_dbg_linear_vt(int flags, const void* act, const void* W, const float* bias, int F, int S, int N, int K, void* out) {
    const int prec = flags & 3;
    if (prec != 1 && prec != 2) return UMGEN_E_UNSUPPORTED;
    const int S_pad = ((S + 63) / 64) * 64;
    const size_t R = (size_t)F * S, osz = (size_t)F * N * S_pad * 2;
    DevBuf dA(R * K * 2), dW((size_t)N * K * 2), dB((size_t)N * 4), dO(osz);
    if (!dA.p || !dW.p || !dB.p || !dO.p) return UMGEN_E_NOMEM;
    if (up(dA.p, act, R * K * 2) || up(dW.p, W, (size_t)N * K * 2)) return UMGEN_E_HIP;
    if (bias && up(dB.p, bias, (size_t)N * 4)) return UMGEN_E_HIP;
    (void)hipMemset(dO.p, 0, osz);
    GemmArgs g{};
    g.P = dA.p; g.Q = dW.p; g.Mi = S; g.Nj = N; g.K = K; g.ldp = K; g.ldq = K; g.strideP = (long)S * K; g.strideQ = 0; g.batch = F;
    g.mode = GEMM_VT; g.bias = bias ? (const float*)dB.p : nullptr; g.out = dO.p; g.ldo = S_pad; g.H = N / kHeadDim;
    g.tile256 = (flags & 16) ? 1 : ((flags & 32) ? -1 : 0);
    if (prec == 2) launch_gemm_mfma<f16_t>(nullptr, g); else launch_gemm_mfma<bf16_t>(nullptr, g);
    if (hipDeviceSynchronize() != hipSuccess) return UMGEN_E_HIP;
    return down(out, dO.p, osz);
}

// spatial attention on q|k rows [F*S][2E] and v rows [F*S][E] (both row-major on the host; V is transposed on the device
// through the same GEMM_VT-layout the engine uses).  y [F*S][E].
int umgen_dbg_attn_spatial(int flags, const void* qk, const void* v, int F, int S, int H, void* y) {
    const int bf16 = flags & 3;                 // precision code; flag 32 (fp32 only): the VALU kernel instead of the matrix-core one
    const int E = H * kHeadDim, S_pad = ((S + 63) / 64) * 64;
    const size_t es = bf16 ? 2 : 4;
    const size_t R = (size_t)F * S;
    // host-side transpose of V into [F][H][48][S_pad]
    std::vector<unsigned char> vt((size_t)F * E * S_pad * es, 0);
    const unsigned char* vs = (const unsigned char*)v;
    for (int f = 0; f < F; ++f)
        for (int s = 0; s < S; ++s)
            for (int c = 0; c < E; ++c)
                memcpy(&vt[(((size_t)f * E + c) * S_pad + s) * es], &vs[(((size_t)f * S + s) * E + c) * es], es);
    DevBuf dQK(R * 2 * E * es), dVT(vt.size()), dY(R * E * es);
    if (!dQK.p || !dVT.p || !dY.p) return UMGEN_E_NOMEM;
    if (up(dQK.p, qk, R * 2 * E * es) || up(dVT.p, vt.data(), vt.size())) return UMGEN_E_HIP;
    if (bf16 == 2) launch_attn_spatial_mfma<f16_t>(nullptr, (const f16_t*)dQK.p, (const f16_t*)dVT.p, (f16_t*)dY.p, F, S, S_pad, H);
    else if (bf16) launch_attn_spatial_mfma<bf16_t>(nullptr, (const bf16_t*)dQK.p, (const bf16_t*)dVT.p, (bf16_t*)dY.p, F, S, S_pad, H);
    else if (flags & 32) launch_attn_spatial_valu<float>(nullptr, (const float*)dQK.p, (const float*)dVT.p, (float*)dY.p, F, S, S_pad, H);   // flag 32: the VALU kernel
    else launch_attn_spatial_f32_mfma(nullptr, (const float*)dQK.p, (const float*)dVT.p, (float*)dY.p, F, S, S_pad, H);
    if (hipDeviceSynchronize() != hipSuccess) return UMGEN_E_HIP;
    return down(y, dY.p, R * E * es);
}

// temporal causal attention on qkv rows [B*T*S][3E] -> y [B*T*S][E].  split = P > 0: slots 0..P-1 first (k | v appended to a
// slot cache), then slots P..T-1 against that cache -- must equal the single pass bit for bit.
int umgen_dbg_attn_temporal(int bf16, const void* qkv, int B, int T, int S, int H, int split, void* y) {
    const int E = H * kHeadDim;
    const size_t es = bf16 ? 2 : 4, R = (size_t)B * T * S;
    if (split <= 0 || split >= T) {
        DevBuf dQ(R * 3 * E * es), dY(R * E * es);
        if (!dQ.p || !dY.p) return UMGEN_E_NOMEM;
        if (up(dQ.p, qkv, R * 3 * E * es)) return UMGEN_E_HIP;
        if (bf16 == 2) launch_attn_temporal<f16_t>(nullptr, (const f16_t*)dQ.p, (f16_t*)dY.p, B, T, S, H);
        else if (bf16) launch_attn_temporal<bf16_t>(nullptr, (const bf16_t*)dQ.p, (bf16_t*)dY.p, B, T, S, H);
        else launch_attn_temporal<float>(nullptr, (const float*)dQ.p, (float*)dY.p, B, T, S, H);
        if (hipDeviceSynchronize() != hipSuccess) return UMGEN_E_HIP;
        return down(y, dY.p, R * E * es);
    }
    const int Tcap = T + 1;
    DevBuf dC((size_t)B * Tcap * S * 2 * E * es);
    if (!dC.p) return UMGEN_E_NOMEM;
    const int t0s[2] = {0, split}, tns[2] = {split, T - split};
    const size_t row = (size_t)3 * E * es, orow = (size_t)E * es;
    for (int pass = 0; pass < 2; ++pass) {
        const int t0 = t0s[pass], Tn = tns[pass];
        const size_t Rn = (size_t)B * Tn * S;
        std::vector<unsigned char> hq(Rn * row), hy(Rn * orow);
        for (int b = 0; b < B; ++b)   // gather the [b][t0 .. t0+Tn) slots into a compact [B][Tn][S] block
            memcpy(&hq[(size_t)b * Tn * S * row], (const unsigned char*)qkv + ((size_t)b * T + t0) * S * row, (size_t)Tn * S * row);
        DevBuf dQ(hq.size()), dY(hy.size());
        if (!dQ.p || !dY.p) return UMGEN_E_NOMEM;
        if (up(dQ.p, hq.data(), hq.size())) return UMGEN_E_HIP;
        TemporalRange tr{t0, dC.p, Tcap, pass == 0 ? 1 : 0};
        if (bf16 == 2) launch_attn_temporal<f16_t>(nullptr, (const f16_t*)dQ.p, (f16_t*)dY.p, B, Tn, S, H, tr);
        else if (bf16) launch_attn_temporal<bf16_t>(nullptr, (const bf16_t*)dQ.p, (bf16_t*)dY.p, B, Tn, S, H, tr);
        else launch_attn_temporal<float>(nullptr, (const float*)dQ.p, (float*)dY.p, B, Tn, S, H, tr);
        if (hipDeviceSynchronize() != hipSuccess) return UMGEN_E_HIP;
        if (down(hy.data(), dY.p, hy.size())) return UMGEN_E_HIP;
        for (int b = 0; b < B; ++b)
            memcpy((unsigned char*)y + ((size_t)b * T + t0) * S * orow, &hy[(size_t)b * Tn * S * orow], (size_t)Tn * S * orow);
    }
    return UMGEN_OK;
}

// decode-style attention: q [NQ][E] fp32, kv [L][2E] (k | v) of dtype bf16/fp32 shared by all queries -> y [NQ][E] fp32
// (partial pass + the combine that normally runs in the projection prologue, here through an identity projection)
int umgen_dbg_attn_decode(int bf16, const float* q, const void* kv, int NQ, int L, int H, float* y) {
    const int E = H * kHeadDim;
    const size_t es = bf16 ? 2 : 4;
    DevBuf dQ((size_t)NQ * E * 4), dKV((size_t)L * 2 * E * es), dP((size_t)NQ * H * kAttnRec * 4), dW((size_t)E * E * 4), dX((size_t)NQ * E * 4);
    if (!dQ.p || !dKV.p || !dP.p || !dW.p || !dX.p) return UMGEN_E_NOMEM;
    if (up(dQ.p, q, (size_t)NQ * E * 4) || up(dKV.p, kv, (size_t)L * 2 * E * es)) return UMGEN_E_HIP;
    std::vector<float> eye((size_t)E * E, 0.f);
    for (int i = 0; i < E; ++i) eye[(size_t)i * E + i] = 1.f;
    if (up(dW.p, eye.data(), eye.size() * 4)) return UMGEN_E_HIP;
    (void)hipMemset(dX.p, 0, (size_t)NQ * E * 4);
    if (bf16 == 2) launch_attn_partial<f16_t>(nullptr, (const float*)dQ.p, (const f16_t*)dKV.p, 0, kHeadDim, 2L * E, E, NQ, NQ, H, nullptr, L, attn_nsplit(L), (float*)dP.p);
    else if (bf16) launch_attn_partial<bf16_t>(nullptr, (const float*)dQ.p, (const bf16_t*)dKV.p, 0, kHeadDim, 2L * E, E, NQ, NQ, H, nullptr, L, attn_nsplit(L), (float*)dP.p);
    else launch_attn_partial<float>(nullptr, (const float*)dQ.p, (const float*)dKV.p, 0, kHeadDim, 2L * E, E, NQ, NQ, H, nullptr, L, attn_nsplit(L), (float*)dP.p);
    GemvResidArgs a{};
    a.part = (const float*)dP.p; a.H = H; a.ns = attn_nsplit(L); a.W = dW.p; a.N = E; a.K = E; a.M = NQ; a.x = (float*)dX.p; a.ldx = E;
    launch_gemv_resid<float>(nullptr, a);
    if (hipDeviceSynchronize() != hipSuccess) return UMGEN_E_HIP;
    return down(y, dX.p, (size_t)NQ * E * 4);
}

// times `iters` launches of the bf16 MFMA GEMM (mode: GEMM_STORE / GEMM_RESID) on device-resident random operands;
// returns the average milliseconds per launch through *ms.  tokens R, features N, reduction K.
int umgen_dbg_gemm_bench(int R, int N, int K, int mode, int iters, float* ms) {
    DevBuf dA((size_t)R * K * 2), dW((size_t)N * K * 2), dO((size_t)R * N * 4);
    if (!dA.p || !dW.p || !dO.p) return UMGEN_E_NOMEM;
    std::vector<bf16_t> h((size_t)R * K);
    unsigned x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = f32_to_bf16(((x >> 8) & 0xffff) / 32768.0f - 1.0f); }
    if (up(dA.p, h.data(), h.size() * 2)) return UMGEN_E_HIP;
    h.resize((size_t)N * K);
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = f32_to_bf16((((x >> 8) & 0xffff) / 32768.0f - 1.0f) * 0.05f); }
    if (up(dW.p, h.data(), h.size() * 2)) return UMGEN_E_HIP;
    (void)hipMemset(dO.p, 0, (size_t)R * N * 4);
    GemmArgs g{};
    g.P = dW.p; g.Q = dA.p; g.Mi = N; g.Nj = R; g.K = K; g.ldp = K; g.ldq = K; g.batch = 1;
    g.mode = mode & 15; g.gelu = (mode >> 4) & 1; g.out = dO.p; g.ldo = N;   // mode bit 4: erf-GELU epilogue
    g.tile256 = (mode & 64) ? -1 : ((mode & 32) ? 1 : 0);                     // bit 5: force the 256 x 256 kernel, bit 6: 128 x 128 kernels only
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch_gemm_mfma<bf16_t>(nullptr, g);
    (void)hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters; ++i) launch_gemm_mfma<bf16_t>(nullptr, g);
    (void)hipEventRecord(e1, nullptr);
    if (hipDeviceSynchronize() != hipSuccess) return UMGEN_E_HIP;
    float t = 0.f;
    (void)hipEventElapsedTime(&t, e0, e1);
    *ms = t / iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return 0;
}

// timing of the V^T GEMM (GEMM_VT) of F frames x S tokens; flag 32: force the 256-tile kernel, 64: the 128-tile kernels only
int umgen_dbg_gemm_vt_bench(int F, int S, int N, int K, int flag, int iters, float* ms) {
    const int S_pad = ((S + 63) / 64) * 64;
    const size_t R = (size_t)F * S;
    DevBuf dA(R * K * 2), dW((size_t)N * K * 2), dO((size_t)F * N * S_pad * 2);
    if (!dA.p || !dW.p || !dO.p) return UMGEN_E_NOMEM;
    std::vector<bf16_t> h(R * K);
    unsigned x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = f32_to_bf16(((x >> 8) & 0xffff) / 32768.0f - 1.0f); }
    if (up(dA.p, h.data(), h.size() * 2)) return UMGEN_E_HIP;
    h.resize((size_t)N * K);
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = f32_to_bf16((((x >> 8) & 0xffff) / 32768.0f - 1.0f) * 0.05f); }
    if (up(dW.p, h.data(), h.size() * 2)) return UMGEN_E_HIP;
    (void)hipMemset(dO.p, 0, (size_t)F * N * S_pad * 2);
    GemmArgs g{};
    g.P = dA.p; g.Q = dW.p; g.Mi = S; g.Nj = N; g.K = K; g.ldp = K; g.ldq = K; g.strideP = (long)S * K; g.strideQ = 0; g.batch = F;
    g.mode = GEMM_VT; g.out = dO.p; g.ldo = S_pad; g.H = N / kHeadDim;
    g.tile256 = (flag & 64) ? -1 : ((flag & 32) ? 1 : 0);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch_gemm_mfma<bf16_t>(nullptr, g);
    (void)hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters; ++i) launch_gemm_mfma<bf16_t>(nullptr, g);
    (void)hipEventRecord(e1, nullptr);
    if (hipDeviceSynchronize() != hipSuccess) return UMGEN_E_HIP;
    float t = 0.f;
    (void)hipEventElapsedTime(&t, e0, e1);
    *ms = t / iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return 0;
}

int umgen_dbg_gemm_stamps(unsigned long long* out16) { return gemm256_read_stamps(out16); }

// few-row linear: out[M][N] = LN(x[M][K]; ln_w) . W[N][K]^T + bias, optional GELU.  W dtype bf16/fp32, activations fp32.
int umgen_dbg_gemv(int bf16, const float* x, const float* ln_w, const void* W, const float* bias, int M, int N, int K, int gelu, float* out) {
    const size_t es = bf16 ? 2 : 4;
    DevBuf dX((size_t)M * K * 4), dL((size_t)K * 4), dW((size_t)N * K * es), dB((size_t)N * 4), dO((size_t)M * N * 4);
    if (!dX.p || !dL.p || !dW.p || !dB.p || !dO.p) return UMGEN_E_NOMEM;
    if (up(dX.p, x, (size_t)M * K * 4) || up(dW.p, W, (size_t)N * K * es)) return UMGEN_E_HIP;
    if (ln_w && up(dL.p, ln_w, (size_t)K * 4)) return UMGEN_E_HIP;
    if (bias && up(dB.p, bias, (size_t)N * 4)) return UMGEN_E_HIP;
    GemvArgs a{};
    a.x = (const float*)dX.p; a.ldx = K; a.ln_w = ln_w ? (const float*)dL.p : nullptr; a.W = dW.p; a.bias = bias ? (const float*)dB.p : nullptr;
    a.N = N; a.K = K; a.M = M; a.out_mode = gelu ? GEMV_OUT_GELU : GEMV_OUT_F32; a.out = (float*)dO.p; a.ldo = N; a.E = K;
    if (bf16 == 2) launch_gemv<f16_t>(nullptr, a); else if (bf16) launch_gemv<bf16_t>(nullptr, a); else launch_gemv<float>(nullptr, a);
    if (hipDeviceSynchronize() != hipSuccess) return UMGEN_E_HIP;
    if (int rc = down(out, dO.p, (size_t)M * N * 4)) return rc;
    if (M > 1) {   // the one-row-per-workgroup form (what the engine launches for several scenes) must give the same bits
        std::vector<float> alt((size_t)M * N);
        (void)hipMemset(dO.p, 0, (size_t)M * N * 4);
        a.rows_per_block = 1;
        if (bf16 == 2) launch_gemv<f16_t>(nullptr, a); else if (bf16) launch_gemv<bf16_t>(nullptr, a); else launch_gemv<float>(nullptr, a);
        if (hipDeviceSynchronize() != hipSuccess) return UMGEN_E_HIP;
        if (int rc = down(alt.data(), dO.p, (size_t)M * N * 4)) return rc;
        if (memcmp(alt.data(), out, (size_t)M * N * 4) != 0) return UMGEN_E_STATE;
    }
    return UMGEN_OK;
}

// the top-k sampler (frame.hip block_sample_topk: UMGen.py:899-913 + 967-974 on the build's uniforms) on n independent rows of V <= 8192
// logits; *overflow counts the rows whose kept set (ties at the k-th value) exceeded the sampler's 64 slots
int umgen_dbg_sample_topk(const float* logits, int n, int V, int k, float temp, const float* u, int32_t* tokens, int32_t* overflow) {
    if (V > 8192 || V < 1 || n < 1) return UMGEN_E_INVALID;
    DevBuf dL((size_t)n * V * 4), dU((size_t)n * 4), dT((size_t)n * 4), dO(4);
    if (!dL.p || !dU.p || !dT.p || !dO.p) return UMGEN_E_NOMEM;
    if (up(dL.p, logits, (size_t)n * V * 4) || up(dU.p, u, (size_t)n * 4)) return UMGEN_E_HIP;
    (void)hipMemset(dO.p, 0, 4);
    launch_sample_rows(nullptr, (const float*)dL.p, V, k, temp, (const float*)dU.p, (int*)dT.p, (int*)dO.p, n);
    if (hipDeviceSynchronize() != hipSuccess) return UMGEN_E_HIP;
    if (int rc = down(tokens, dT.p, (size_t)n * 4)) return rc;
    return down(overflow, dO.p, 4);
}

// Timing hook of the batched decode layer's kernels (decode_batched.hip) on random data: one BlockOAR layer's five launches at M scenes and
// KV length L, `iters` times back to back; us[0..4] = average microseconds of q|k|v, attention, c_proj, c_fc, mlp c_proj (HIP events
// around each launch), us[5] = the five as one sequence.
int umgen_dbg_batched_layer_bench(int prec, int M, int L, int iters, float* us) {
    const int E = 768, H = 16, Lmax = 2304;
    if (prec != 1 && prec != 2) return UMGEN_E_INVALID;
    if (M < 1 || M > kRowsMaxM || L < 1 || L >= Lmax) return UMGEN_E_INVALID;
    const size_t wsz = (size_t)12 * E * E * 2, csz = (size_t)M * 2 * H * Lmax * kHeadDim * 2;
    DevBuf dW(wsz), dC(csz), dx((size_t)64 * E * 4), dxr((size_t)64 * E * 4), dq((size_t)64 * E * 4), da((size_t)64 * E * 4), dh((size_t)64 * 4 * E * 4), dln((size_t)E * 4), db((size_t)4 * E * 4), dlen(16);
    if (!dW.p || !dC.p || !dx.p || !dxr.p || !dq.p || !da.p || !dh.p || !dln.p || !db.p || !dlen.p) return UMGEN_E_NOMEM;
    std::vector<unsigned short> hw(wsz / 2);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (unsigned short)(0x3c00 + (i * 2654435761u >> 24)) & (prec == 1 ? 0x3cff : 0x2fff);   // small positive values
    std::vector<float> hx((size_t)64 * E, 0.25f), hl(E, 1.f), hb(4 * E, 0.f);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = 0.25f + 1e-3f * (float)(i % 97);
    if (up(dW.p, hw.data(), wsz) || up(dx.p, hx.data(), hx.size() * 4) || up(dxr.p, hx.data(), hx.size() * 4) || hipMemset(da.p, 0, (size_t)64 * E * 4) != hipSuccess || hipMemset(dh.p, 0, (size_t)64 * 4 * E * 4) != hipSuccess || up(dln.p, hl.data(), E * 4) || up(db.p, hb.data(), 4 * E * 4)) return UMGEN_E_HIP;
    if (hipMemset(dC.p, 0, csz) != hipSuccess || hipMemcpy(dlen.p, &L, 4, hipMemcpyHostToDevice) != hipSuccess) return UMGEN_E_HIP;
    hipStream_t st;
    if (hipStreamCreate(&st) != hipSuccess) return UMGEN_E_HIP;
    const char* Wq = (const char*)dW.p;
    auto run = [&](int which, auto tag) {
        typedef decltype(tag) TT;
        RowsArgs r{};
        r.M = M; r.E = E;
        switch (which) {
            case 0: r.x = (float*)dx.p; r.ln_w = (float*)dln.p; r.W = Wq; r.bias = (float*)db.p; r.N = 3 * E; r.K = E; r.mode = ROWS_QKV; r.out = (float*)dq.p; r.ldo = E;
                    r.cache = dC.p; r.scene_stride = (long)2 * H * Lmax * kHeadDim; r.d_len = (int*)dlen.p; r.Lmax = Lmax; launch_rows_mfma<TT>(st, r); break;
            case 1: launch_attn_decode_batched<TT>(st, (float*)dq.p, (const TT*)dC.p, (long)2 * H * Lmax * kHeadDim, M, H, Lmax, (int*)dlen.p, (float*)da.p); break;
            case 2: r.x = (float*)da.p; r.W = Wq + (size_t)3 * E * E * 2; r.bias = (float*)db.p; r.N = E; r.K = E; r.mode = ROWS_RESID; r.out = (float*)dxr.p; r.ldo = E; r.out_frag = (float*)dx.p; launch_rows_mfma<TT>(st, r); break;
            case 3: r.x = (float*)dx.p; r.ln_w = (float*)dln.p; r.W = Wq + (size_t)4 * E * E * 2; r.N = 4 * E; r.K = E; r.mode = ROWS_GELU; r.out_frag = (float*)dh.p; launch_rows_mfma<TT>(st, r); break;
            default: r.x = (float*)dh.p; r.W = Wq + (size_t)8 * E * E * 2; r.N = E; r.K = 4 * E; r.mode = ROWS_RESID; r.out = (float*)dxr.p; r.ldo = E; r.out_frag = (float*)dx.p; launch_rows_mfma<TT>(st, r); break;
        }
    };
    auto run_p = [&](int which) { if (prec == 2) run(which, f16_t{}); else run(which, bf16_t{}); };
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int which = 0; which <= 5; ++which) {
        for (int w = 0; w < 3; ++w) { if (which < 5) run_p(which); else for (int k = 0; k < 5; ++k) run_p(k); }
        hipEventRecord(e0, st);
        for (int i = 0; i < iters; ++i) { if (which < 5) run_p(which); else for (int k = 0; k < 5; ++k) run_p(k); }
        hipEventRecord(e1, st);
        if (hipEventSynchronize(e1) != hipSuccess) return UMGEN_E_HIP;
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        us[which] = ms * 1000.f / (float)iters;
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipStreamDestroy(st);
    return UMGEN_OK;
}

}  // extern "C"
