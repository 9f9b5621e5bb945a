// Per-frame data layout shared by the host engine (engine.hip) and the frame kernels (frame.hip).
#pragma once
#include "common.h"

namespace umgen {

// scene-sequence positions (0-based) -- infer_fun.py:112-118, UMGen.py:976-992
constexpr int kPoseBos = 0, kPoseEos = 4, kMapBos = 5, kMapC0 = 6, kMapEos = 1030, kBoxBos = 1031, kBoxC0 = 1032, kBoxEos = 1692,
              kImgBos = 1693, kImgC0 = 1694, kImgEos = 2206;
constexpr int kNMap = 1024, kNBox = 660, kNImg = 512, kNPose = 3;
constexpr int kSlots = 60, kSlotLen = 11, kBoxPad = 1027;
constexpr int kTokPerFrame = kNPose + kNMap + kNBox + kNImg;   // 2199 content tokens, stored pose|map|bbox3d|image
constexpr int kOffMap = kNPose, kOffBox = kNPose + kNMap, kOffImg = kNPose + kNMap + kNBox;

enum Stack { STACK_EGO = 0, STACK_MAP = 1, STACK_BOX = 2, STACK_TAR = 3 };
__host__ __device__ inline int stack_len(int st) { return st == STACK_MAP ? 1031 : (st == STACK_BOX ? 1693 : 2207); }

// embedding tables (device pointers); fp32 unless noted
struct EmbedTables {
    const float *egoe, *axe, *be, *tpe, *spe, *tske;
    const bf16_t *fouier_pe, *posi, *grid_posi;   // bf16 sinusoid tables (UMGen.py:137-153)
    const float *gmap, *gimg;                     // GMLP(codebook) rows [vocab][E] (module.py:710-743 applied to every code once)
    int E;
};

// window tokens of the B scenes on the device: int32 [B][T][...]
struct WindowTokens {
    const int *pose;   // [B][T][3]   (already shifted one frame ahead for the TAR stacks, UMGen.py:1445-1452)
    const int *map;    // [B][T][1024]
    const int *box;    // [B][T][660]
    const int *img;    // [B][T][512]
    int B, T;          // scenes, history slots covered by this pass (rows of X are (b, t_local, s))
    int Tfull = 0;     // slots per scene in the token arrays / pose_diff (0: = T)
    int t0 = 0;        // first slot of the pass: tokens and tpe are indexed with t0 + t_local
};

constexpr unsigned kEpochPerStep = 16384;  // hand-off tags one decode step may use: (round or scene <= 32) x (layer <= 64) x (edge <= 8)
constexpr int kEngMaxSystolic = 32;        // scenes one systolic engine launch can carry (tag budget)

struct SamplerParams {
    int method, top_k, top_k_map, topk_image;
    float p, p_map, temperature;
    int rule_constrain, merge_ar_tar, only_ar;
};

// state of the OAR decode loop: device resident, so that one decode step is a fixed kernel sequence with fixed
// arguments and can be replayed from a hipGraph (everything that changes between steps / frames / calls lives here)
struct OarState {
    int step;        // input position j of the current step == KV length before the step
    int frame_idx;
    int use_forced;  // teacher forcing: take tokens from SampleArgs::forced
    int use_control; // control_test: SampleArgs::control_slot is valid
    int done;        // blocks of the step's last kernel that have finished (the last one advances `step`)
    unsigned epoch;  // base of the decode engine's hand-off tags for this step (advanced with `step`, oar_engine.hip)
    SamplerParams sp;
};

// everything the per-token sampler kernel touches
struct SampleArgs {
    OarState* st;
    EmbedTables tb;
    const float* logits;       // [B][ld_logits] current AR head
    const float* logits_tar;   // [B][660][ld_tar] head_tar_bbox3d on the conditioning rows of the 660 bbox3d positions (one GEMM per frame)
    int ld_logits;
    int ld_tar;
    int vocab;                 // vocab of the current AR head
    int mod;                   // 1 map, 2 bbox3d, 3 image
    const float* cond;         // [B][2207][E]
    float* x_next;             // [B][E] input of the next step
    int* tokens;               // [B][2199] tokens of the frame being generated
    const int* prev_box;       // [B][660] bbox3d tokens of the last history frame (after control overwrite)
    const unsigned char* control_slot;   // [B][60] (valid when st->use_control)
    double* boxes;             // [B][64][10] decoded boxes of this frame (rule constraint), boxes[b][0] = ego
    int* n_boxes;              // [B]
    const unsigned long long* seeds;   // [B]
    const int* forced;         // [B][2199] teacher forcing (valid when st->use_forced)
    int* counters;             // [8] per-frame event counters (pad_avoid, control, rule_checked, rule_collision, rule_blanked, sampled != forced,
                               //     -, 7: unused since round 5 -- more than 64 ties at the k-th logit are sampled by an exhaustive walk, frame.hip)
};

void launch_embed_stack(hipStream_t s, int stack, const EmbedTables& tb, const WindowTokens& w, float* X, float* mapfeat);
// warp the map features by the ego motion and finish the map rows of X (UMGen.py:321-354, 729-736, 799-802, 836-840)
void launch_warp_map(hipStream_t s, int stack, const EmbedTables& tb, int B, int T, const float* mapfeat, const float* pose_diff,
                     float* X, float* warped_last, int Tfull = 0, int t0 = 0);
// conditioning rows: LayerNorm of the last history frame of a stack (+ warped-map prior) -> cond[b][s] (UMGen.py:1496-1511)
void launch_cond_rows(hipStream_t s, int stack, int B, int T, int E, const float* X, const float* ln_w, const float* warped_last,
                      float* cond);
// x[b] = row (+ cond[b][pos])
void launch_first_input(hipStream_t s, int B, int E, const float* tske_row, const float* cond, float* x);
void launch_fixed_token(hipStream_t s, const SampleArgs& a, int B);             // bos/eos/pose prefix steps
// given-token prefix as one pass (engine.hip run_prefix_prefill): decode inputs of positions 0 .. P - 1, and a pass's k | V^T rows -> the decode cache
void launch_prefix_rows(hipStream_t s, const EmbedTables& tb, const float* tske_row, const float* cond, const int* tokens, int B, int P, float* X,
                        float* x_last);
template <typename T>
void launch_prefix_kv_to_cache(hipStream_t s, const T* qk, const T* vt, int B, int S, int S_pad, int H, int Lmax, T* cache, long scene_stride);
void launch_sample_token(hipStream_t s, const SampleArgs& a, int B);            // sampled steps
// ego head: sample 3 pose tokens per scene from logits [B*3][vocab]
void launch_sample_rows(hipStream_t s, const float* logits, int V, int k, float temp, const float* u, int* out, int* overflow, int n);   // test hook
void launch_sample_ego(hipStream_t s, const float* logits, int vocab, SamplerParams sp, const unsigned long long* seeds, int frame_idx,
                       const int* forced, int* out_tokens, int B, int* overflow);
// ego query rows: egoe[j] + spe[j] + tpe[T-1]
void launch_ego_queries(hipStream_t s, const EmbedTables& tb, int B, int T, float* x);

}  // namespace umgen
