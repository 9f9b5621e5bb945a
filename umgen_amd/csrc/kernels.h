// Launcher declarations for the HIP kernels of libumgen_hip (implemented in the *.hip files next to this header).
#pragma once
#include "common.h"

namespace umgen {

// ------------------------------------------------------------------------------------------------
// GEMM  C[i][j] = sum_k P[i][k] * Q[j][k]   (both operands K-contiguous, like nn.Linear weights)
// The lane that owns C holds 4 consecutive i for one j, so:
//   P = weights, Q = activations  -> out[token j][feature i..i+3]   (row-major activations out)
//   P = activations, Q = weights  -> Vt[feature j][token i..i+3]    (transposed V for spatial attention)
// ------------------------------------------------------------------------------------------------
enum GemmMode {
    GEMM_STORE = 0,      // outT[j*ldo + i] = acc + bias[i]   (optional exact GELU)          T = operand dtype
    GEMM_RESID = 1,      // X[j*ldo + i]  += acc + bias[i]                                   fp32 residual stream
    GEMM_STORE_F32 = 2,  // outF[j*ldo + i] = acc + bias[i]
    GEMM_VT = 3          // Vt[((z*H + j/48)*48 + j%48)*ldo + i] = acc + bias[j]             T, per-frame batch z
};
struct GemmArgs {
    const void* P; const void* Q;
    int Mi, Nj, K;
    long ldp, ldq;
    long strideP, strideQ;   // per batch z (elements)
    int batch;
    int mode;
    const float* bias;       // nullable
    int gelu;
    void* out;
    long ldo;
    long strideO;            // per batch z (GEMM_STORE/RESID/F32 only)
    int H;                   // heads (GEMM_VT)
    int tile256;             // 0: the launcher picks the kernel; 1: force the 256 x 256 kernel when it supports the shape; -1: never use it
};
// gemm256.hip: 256 x 256 x 64 deep-pipelined kernel (GEMM_STORE / RESID / STORE_F32, batch 1, Mi % 256 == 0, K % 128 == 0)
bool gemm256_supported(const GemmArgs& a);
hipError_t gemm256_prepare();                   // per device, behind hipSetDevice (dynamic-LDS attribute, CU count of that device)
int gemm256_read_stamps(unsigned long long* out16);   // measurement builds only (-2 otherwise)
template <typename TT> void launch_gemm256(hipStream_t s, const GemmArgs& a);
void record_gemm256(const GemmArgs& a);          // bg_queue.h: the same GEMM as an op of the decode engine's background workers
template <typename TT> void launch_gemm_mfma(hipStream_t s, const GemmArgs& a);   // P, Q of the 16-bit type TT (bf16_t / f16_t), MFMA 16x16x32
template <typename TP, typename TQ> void launch_gemm_valu(hipStream_t s, const GemmArgs& a);  // exact fp32 FMA chain

// ------------------------------------------------------------------------------------------------
// row ops
// ------------------------------------------------------------------------------------------------
// out[r] = LayerNorm(x[row_base + r*row_stride]) * w   (weight only, eps 1e-5, module.py:26-37)
template <typename T> void launch_layernorm(hipStream_t s, const float* x, long row_stride, long n_rows, int E, const float* w, T* out);

// ------------------------------------------------------------------------------------------------
// attention (head_dim 48)
// ------------------------------------------------------------------------------------------------
// spatial (non-causal) attention inside each frame: qk [F*S][2E] row-major (q | k), vt [F][H][48][S_pad], y [F*S][E]
template <typename TT> void launch_attn_spatial_mfma(hipStream_t s, const TT* qk, const TT* vt, TT* y, int F, int S, int S_pad, int H);   // TT = bf16_t / f16_t
template <typename T> void launch_attn_spatial_valu(hipStream_t s, const T* qk, const T* vt, T* y, int F, int S, int S_pad, int H);
// fp32 parity mode on v_mfma_f32_32x32x2_f32 (UMGEN_FP32_MFMA=0: the VALU kernel above)
void launch_attn_spatial_f32_mfma(hipStream_t s, const float* qk, const float* vt, float* y, int F, int S, int S_pad, int H);
// the same layouts with flash-attn's causal mask at q_len == k_len (query i sees keys 0 .. i): the OAR prefix pass over the given tokens
template <typename TT> void launch_attn_causal_mfma(hipStream_t s, const TT* qk, const TT* vt, TT* y, int F, int S, int S_pad, int H);
void launch_attn_causal_f32(hipStream_t s, const float* qk, const float* vt, float* y, int F, int S, int S_pad, int H);
// temporal causal attention over T frames per spatial position: qkv [B*T*S][3E] row-major, y [B*T*S][E]
// Temporal attention over history slots [t0, t0 + Tn) held in the qkv rows; k | v of slots [0, t0) are read from `cache`
// ([B][Tcap][S][2E], dtype T) and, when write != 0, the k | v rows of the new slots are appended to it (attn.hip).
struct TemporalRange { int t0; void* cache; int Tcap; int write; int q0 = 0; };   // q0: only query slots >= q0 are evaluated / written
template <typename T> void launch_attn_temporal(hipStream_t s, const T* qkv, T* y, int B, int Tn, int S, int H,
                                                TemporalRange tr = TemporalRange{0, nullptr, 0, 0});

// few-query attention over a key/value stream (OAR decode, ego decoder): partial pass, NSPLIT splits of the keys.
//   q   [NQ][E] fp32;  K row of (scene sc, head h, key k) = kv_base + sc*scene_stride + h*head_stride + k*key_stride (48 values),
//   V row at + v_off.  Row-major k|v rows [L][2E]: head_stride 48, key_stride 2E, v_off E.  Decode cache (head-major, so a
//   split's keys are one contiguous 96 B x n_keys run): [2][H][Lmax][48] -> head_stride Lmax*48, key_stride 48, v_off H*Lmax*48.
//   L   = *d_len + len_add  when d_len != nullptr, else len_add;  scene of query qi = qi / q_per_scene
//   part [NQ][H][kAttnRec]: per (query, head) the split statistics m[sp], l[sp] and o[d][sp] with the split index fastest, so
//   the merge in the projection prologue reads each output column's <= 18 partials with five 16-byte loads
constexpr int kAttnSplit = 18;
constexpr int kAttnChunk = 128;   // split s owns keys [128 s, 128 s + 128); the host launches ns = ceil(L / 128) splits (L <= 2304)
inline int attn_nsplit(int L) { return L <= 0 ? 1 : (L + kAttnChunk - 1) / kAttnChunk; }
constexpr int kAttnPad = 20;                    // split dimension padded so every o-row is 16-byte aligned
constexpr int kAttnRec = 2 * kAttnPad + kHeadDim * kAttnPad;   // floats per (query, head): m[20] | l[20] | o[48][20] (split fastest)
template <typename T> void launch_attn_partial(hipStream_t s, const float* q, const T* kv_base, long scene_stride, long head_stride,
                                               long key_stride, long v_off, int NQ, int q_per_scene, int H, const int* d_len, int len_add,
                                               int ns, float* part);

// ------------------------------------------------------------------------------------------------
// few-row linear layers (decode / ego decoder): weights W [N][K] of type T, activations fp32
// ------------------------------------------------------------------------------------------------
enum GemvOut {
    GEMV_OUT_F32 = 0,    // out[m*ldo + n] = v
    GEMV_OUT_GELU = 1,   // out[m*ldo + n] = gelu(v)
    GEMV_OUT_QKV = 2     // n < E: q[m][n] = v ; else K/V cache of scene m (head-major [2][H][Lmax][48]) at position *d_len
};
struct GemvArgs {
    const float* x; long ldx;      // [M][K] input rows
    const int* d_xoff; long xoff_mul;   // optional device-side row offset: x += (*d_xoff) * xoff_mul
    const float* ln_w;             // nullable: apply LayerNorm(weight only) to each input row first
    const void* W; const float* bias; int N, K, M;
    int out_mode;
    float* out; long ldo;
    void* cache; long scene_stride; const int* d_len; int Lmax;   // GEMV_OUT_QKV
    int rows_per_block;            // 0: a workgroup loops over all M rows; 1: one (feature tile, row) per workgroup (set by the launcher)
    int E;
};
template <typename T> void launch_gemv(hipStream_t s, const GemvArgs& a);

// x[m][n] += sum_k a[m][k] W[n][k] + bias[n];  if part != nullptr the input a is first combined from the attention
// partials (K must equal H*48)
struct GemvResidArgs {
    const float* a; long lda; const float* part; int H;
    int ns;                          // number of attention splits to merge (attn_nsplit of the key count)
    const void* W; const float* bias; int N, K, M;
    float* x; long ldx;
    int rows_per_block;              // as in GemvArgs
};
template <typename T> void launch_gemv_resid(hipStream_t s, const GemvResidArgs& a);

// ------------------------------------------------------------------------------------------------
// batched decode layer for many scenes per GPU (decode_batched.hip): the scenes are the B-columns of the matrix-core instruction
// ------------------------------------------------------------------------------------------------
enum RowsMode {
    ROWS_F32 = 0,    // out[m*ldo + n] = v
    ROWS_GELU = 1,   // out[m*ldo + n] = gelu(v)
    ROWS_QKV = 2,    // n < E: q[m][n] = v ; else K/V cache of scene m (head-major [2][H][Lmax][48]) at position *d_len
    ROWS_RESID = 3   // out[m*ldo + n] += v
};
constexpr int kRowsMaxM = 64;      // scenes per rows_mfma launch (4 column blocks of 16)
constexpr int kRowsMaxNB = 4;
// FRAGMENT-MAJOR activations of the batched launches: [k-step][4 column blocks][64 lanes][8] floats -- the order the matrix cores consume
// them in (a wave's B operand of k-step s and column block nb is 2 KB contiguous); decode_batched.hip
__host__ __device__ inline long frag_index(int m, int k) {
    return ((((long)(k >> 5) * kRowsMaxNB + (m >> 4)) * 64 + ((k & 31) >> 3) * 16 + (m & 15)) << 3) + (k & 7);
}
struct RowsArgs {
    const float* x; int M;             // fp32 activations of the M <= kRowsMaxM scenes, FRAGMENT-MAJOR (decode_batched.hip: frag_index; 64 K floats)
    const float* ln_w;                 // ROWS_QKV / _GELU / _F32: LayerNorm (weight only) of every scene's K values first (K <= 768)
    const void* W; const float* bias; int N, K;   // weights [N][K] of the 16-bit type, K % 32 == 0
    int mode;
    float* out; long ldo;              // row-major outputs: q rows (ROWS_QKV), x (ROWS_RESID: read-modify-write), logits (ROWS_F32)
    float* out_frag;                   // fragment-major outputs: gelu(c_fc) (ROWS_GELU), the copy of x the next launch streams (ROWS_RESID)
    void* cache; long scene_stride; const int* d_len; int Lmax; int E;   // ROWS_QKV
};
template <typename TT> void launch_rows_mfma(hipStream_t s, const RowsArgs& a);
void launch_rows_to_frag(hipStream_t s, const float* x, long ldx, int M, int K, float* xf);   // row-major [M][K] -> fragment-major
// y (fragment-major) of scene b, columns h*48 .. = softmax(q_bh . K_bh^T / sqrt(48)) V_bh over keys 0 .. *d_len of the head-major decode cache,
// one workgroup per (scene, head); q row-major [B][E]
template <typename TT> void launch_attn_decode_batched(hipStream_t s, const float* q, const TT* cache, long scene_stride, int B, int H, int Lmax,
                                                        const int* d_len, float* y);

// ------------------------------------------------------------------------------------------------
// XCD-resident decode engine (oar_engine.hip): all BlockOAR layers of one decode step in one launch, bf16 weights, n_embd 768
// ------------------------------------------------------------------------------------------------
constexpr int kEngE = 768, kEngH = 16;         // the width the engine is built for (UMGen_Large); other widths use the launches above
// UMGEN_ENG_MFMA: which of the decode engine's row dot products run on the matrix cores (oar_engine.hip; bits: 1 q|k|v rows, 2 c_proj rows,
// 4 c_fc rows, 8 mlp partial sums, 16 attention -- correct, but its 32-key register buffers leave room for one only: slower, off).  Bit 4 adds a fragment-ordered copy of c_fc, bit 8 changes the repacked mlp c_proj layout (engine.hip).
// Measured (profiles/r03_engine_experiments.txt, sessions N-R): 12 is the best set -- 441 vs 462 us per launch at one scene, 710 vs 743 at eight
#ifndef UMGEN_ENG_MFMA
#define UMGEN_ENG_MFMA 12
#endif
constexpr int kEngThreads = 512;               // one workgroup per CU
constexpr int kEngGroup = 32;                  // workgroups per group == CUs per XCD
constexpr int kEngWpUnits = 18;                // 16-byte units of the repacked mlp c_proj slice per thread (12 of its full row + 6 of a shared row)
constexpr int kEngLocStride = 3 * kEngE + 2 * kEngH * 50 + kEngE + kEngGroup * kEngE + kEngE;   // granules per group: q|k|v, half partials, x', mlp partial sums [32][768], x
struct OarLayerDev {                           // one BlockOAR's parameters (module.py:378-400)
    const bf16_t *Wqkv, *Wo, *Wfc, *Wproj;
    const bf16_t *Wp2;                         // mlp c_proj repacked for the hidden-unit split: [32 CUs][18 units][512 threads][8] (engine.hip repack_mlp_proj)
    const float *bqkv, *bo, *ln_a, *ln_b;
    const bf16_t *Wf2;                         // UMGEN_ENG_MFMA & 4: c_fc as matrix-core fragments [32 CUs][8 waves][6 tiles x 3 k-steps][64 lanes][8] (1 KB per request)
};
struct OarState;
struct BgQueue;
struct OarEngineArgs {
    const OarLayerDev* layers; int n_layers;
    bf16_t* kvcache; long kv_layer_stride, kv_scene_stride; int Lmax;   // [layer][scene][2][H][Lmax][48]
    bf16_t* vtcache; long vt_layer_stride, vt_scene_stride;             // UMGEN_ENG_MFMA & 16: V once more, dim-major [layer][scene][H][48][Lmax]
    float* xdec;                               // [B][E]: in = input of layer 0, out = output of the last layer
    const OarState* st;                        // step (cached keys) and epoch of the hand-off tags
    unsigned long long* gx;                    // [max_batch][E] cross-group x granules
    unsigned long long* gloc;                  // [NG][kEngLocStride] group-private granules
    unsigned int* ticket;                      // [16] monotonic arrival counters per group (rank = ticket % 32)
    unsigned int* err;                         // give-up word (0 = ok)
    int B, NG, R, D;                           // scenes; groups; scenes per round; groups per scene (NG == R * D)
    unsigned char xcc_group[16];               // physical XCC id -> group index (0xff: not taking part)
    unsigned long long* stamps;                // optional [8]: 100 MHz ticks per phase + item count, accumulated by rank 0 of group 0
    int fp16;                                  // 0: weights and K/V cache hold bfloat16 bits, 1: IEEE half (UMGEN_PREC_FP16)
    int systolic;                              // 1: layer-resident groups, the scenes flow through all NG groups (R, D unused); B * 64 * 8 tags
    // measurement builds only (-DUMGEN_ENG_BURN, UMGEN_DEBUG_BURN="us,mfma,sleep,kb"): the groups no scene uses run a synthetic matrix-core / streaming
    // load for burn_ticks (100 MHz) -- what the decode step costs while the other XCDs work (profiles/r06_engine_contention.txt)
    int burn_ticks, burn_mfma, burn_sleep, burn_kb;
    const void* burn_buf;
    // Background workers (bg_worker.h; one scene, D < NG): the workgroups of the XCD groups no scene uses execute the op list `bg` -- the next frame's TAR /
    // ego pass -- while the engine part runs; bg_only: a launch without an engine part that drains what is left of the list
    BgQueue* bg;
    int bg_only;
};
// ------------------------------------------------------------------------------------------------
// chip-wide decode engine for wide layers (oar_engine_wide.hip): n_embd 1536, one scene per launch, 256 workgroups (6 compute + 2 poll waves),
// hand-offs across the fabric
// ------------------------------------------------------------------------------------------------
constexpr int kWideE = 1536;                   // the width this engine is built for (configs[4]: 2x UMGen_Large)
constexpr int kWideGroups = 256;               // workgroups = ranks (one per CU)
#ifndef UMGEN_WIDE_SPLITS
#define UMGEN_WIDE_SPLITS 4
#endif
constexpr int kWideSplits = UMGEN_WIDE_SPLITS; // key splits per head in the attention phase: 4 = ranks 0..15 of an XCD (8 = every rank: 30 us per step slower below 1500 keys, 30 faster at 2200)
struct OarWideArgs {
    const OarLayerDev* layers; int n_layers;   // Wp2: mlp c_proj repacked [256 ranks][1536 rows][24 hidden units of the rank] (engine.hip repack_wide)
    bf16_t* kvcache; long kv_layer_stride, kv_scene_stride; int Lmax;   // [layer][scene][2][H][Lmax][48]
    float* xdec;                               // [B][E]: in = input of layer 0, out = output of the last layer
    int scene;                                 // the scene of this launch (a call's scenes: one launch each, one behind the other)
    const OarState* st;
    unsigned long long* gran;                  // oar_engine_wide_granules() hand-off granules
    unsigned int* ticket;                      // [8] monotonic arrival counters, one per XCD (rank = 32 x XCD + ticket % 32)
    unsigned int* err;
    int fp16;
    unsigned long long* stamps;                // optional [16]: 100 MHz ticks per phase + layer count, rank 0 / scene 0
};
size_t oar_engine_wide_lds_bytes();
size_t oar_engine_wide_granules();
hipError_t oar_engine_wide_prepare();
hipError_t launch_oar_engine_wide(hipStream_t s, const OarWideArgs& a);
hipError_t launch_oar_engine_wide_census(hipStream_t s, unsigned int* d_counts16);

size_t oar_engine_lds_bytes();
size_t oar_engine_bg_lds_bytes();      // launches with background workers (bg_worker.h)
hipError_t oar_engine_prepare();                // per device, before the first launch / census (dynamic-LDS attribute)
hipError_t launch_oar_engine(hipStream_t s, const OarEngineArgs& a);
hipError_t launch_oar_engine_census(hipStream_t s, int n_groups, unsigned int* d_counts16);

}  // namespace umgen
