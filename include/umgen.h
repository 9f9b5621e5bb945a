/*
 * umgen.h -- C ABI of libumgen_hip.so, the MI355X (gfx950) next-scene rollout engine for UMGen.
 *
 * This is the drop-in boundary for ONE hot path of the reference (YanhaoWu/UMGen):
 *     UMGen.inference(...)                       projects/models/UMGen.py:1542-1671
 * reached through the model registry             projects/registry.py:1-3, UMGen.py:51-52
 * from                                           projects/tools/model_pl.py:173-175, 237-239.
 * The reference has no FFI of its own (it is pure PyTorch); a maintainer binds these entry points
 * with ctypes from a `UMGen(nn.Module)` shim -- see INTEGRATION.md and umgen_amd/model.py.
 *
 * Conventions: plain C, no exceptions across the boundary.  Every function returns 0 on success
 * or a negative UMGEN_E_* code; umgen_last_error() gives the message.  All pointer arguments are
 * HOST pointers to contiguous arrays owned by the caller (the library copies); a handle owns one
 * HIP stream and is not thread-safe.  Token arrays are int64 like the reference's LongTensors.
 */
#ifndef UMGEN_H_
#define UMGEN_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UMGEN_ABI_VERSION 4   /* 2: umgen_rollout takes given_map / given_bbox3d; umgen_timings::decode_batched.  3: umgen_timings::prefix_passes; decode_engine values 1 / 3 (2 retired).
                               * 4: umgen_trace::given_map / given_bbox3d (umgen_frame with a given-token prefix: traced logits behind a prefix pass) */

enum {
    UMGEN_OK = 0,
    UMGEN_E_INVALID = -1,   /* bad argument / shape / key          */
    UMGEN_E_STATE = -2,     /* call order (e.g. rollout before finalize) */
    UMGEN_E_HIP = -3,       /* HIP runtime error                   */
    UMGEN_E_NOMEM = -4,
    UMGEN_E_UNSUPPORTED = -5
};

/* FP32: exact-fp32 parity mode.  BF16: bf16 weights / operands / KV cache (the bench mode, BASELINE.json configs[1]).
 * FP16: the same kernels with IEEE-half operands -- the reference's own arithmetic (torch.cuda.amp.autocast fp16, UMGen.py:1604-1605);
 *       umgen_load_tensor refuses (UMGEN_E_INVALID) a matrix weight whose magnitude exceeds 65504 instead of storing inf */
enum { UMGEN_PREC_FP32 = 0, UMGEN_PREC_BF16 = 1, UMGEN_PREC_FP16 = 2 };
enum { UMGEN_DT_F32 = 0, UMGEN_DT_BF16 = 1, UMGEN_DT_F16 = 2, UMGEN_DT_F64 = 3 };
enum { UMGEN_SAMPLE_TOPK = 0, UMGEN_SAMPLE_TOPP = 1 };

/* scene-sequence constants (infer_fun.py:112-118): content tokens per frame */
#define UMGEN_S_POSE 3
#define UMGEN_S_MAP 1024
#define UMGEN_S_BBOX3D 660
#define UMGEN_S_IMAGE 512
#define UMGEN_SEQ_LEN 2207

typedef struct umgen_engine umgen_engine; /* opaque */

/* Mirrors the resolved `config` Namespace read in UMGen.__init__ (UMGen.py:53-172). */
typedef struct umgen_config {
    int32_t abi_version; /* = UMGEN_ABI_VERSION */
    int32_t n_embd, n_head;
    int32_t n_ego_tar_layer, n_ego_ca_layer, n_map_tar_layer, n_box_tar_layer, n_tar_layer, n_oar_layer;
    int32_t pose_vocab, map_vocab, bbox3d_vocab, img_vocab, aux_vocab;
    int32_t n_map_embd, n_img_embd;
    int32_t max_frame_len; /* rows of tpe */
    int32_t task_num, task_id;
    int32_t precision;       /* UMGEN_PREC_* */
    int32_t max_batch;       /* scenes rolled out together on this GPU */
    int32_t max_cond_frames; /* history window T (reference: 20) */
    int32_t device;          /* HIP device ordinal */
    int32_t use_graphs;      /* 1 = replay the decode step from a hipGraph */
} umgen_config;

/* Sampler parameters (UMGen.py:99-126; infer_task_config, config.py:442-463). */
typedef struct umgen_sampling {
    int32_t method;                     /* UMGEN_SAMPLE_* */
    int32_t top_k, top_k_map, topk_image;
    float p, p_map, temperature;
    int32_t rule_constrain;             /* evaluate.py:59-63, UMGen.py:1116-1123 */
    int32_t merge_ar_tar, only_ar;      /* UMGen.py:1092-1104 */
    const uint64_t *seeds;              /* [B] per-scene RNG seed (counter-based) */
} umgen_sampling;

/* Optional per-frame trace of one scene (teacher-forced logit parity; replaces SURVEY's umgen_step_logits).
 * Any pointer may be NULL.  Sizes use the engine's n_embd / vocab sizes. */
typedef struct umgen_trace {
    float *cond;            /* [2207][n_embd]  conditioning rows fed to the OAR (UMGen.py:1227-1231) */
    float *ego_logits;      /* [3][pose_vocab]                                  (UMGen.py:1001-1002) */
    float *logits_map;      /* [1024][map_vocab]                                 (UMGen.py:1062)      */
    float *logits_bbox3d;   /* [660][bbox3d_vocab]                               (UMGen.py:1072)      */
    float *logits_image;    /* [512][img_vocab]                                  (UMGen.py:1132)      */
    const int64_t *forced_pose, *forced_map, *forced_bbox3d, *forced_image; /* teacher forcing, [S_mod] */
    int32_t *counters;      /* [8] events of the frame: 0 pad-avoid resamples, 1 control resamples, 2 rule checks,
                               3 rule collisions, 4 slots blanked, 5 sampled != forced token (teacher forcing only) */
    const int64_t *given_map, *given_bbox3d; /* NULL, or the frame's GIVEN map [1024] (and boxes [660], only behind a given map): umgen_rollout's
                               given_* for one frame (UMGen.py:1184-1201), so that the logits behind a given-token prefix can be traced */
} umgen_trace;

/* Event-timed phases of the last umgen_rollout call (milliseconds, HIP events on the engine stream). */
typedef struct umgen_timings {
    double total_ms, ego_ms, tar_ms, oar_ms;   /* foreground stream: ego net, TAR stacks (or their last slot), decode loop */
    int64_t frames, oar_steps, oar_kernels;
    double gemm_ms;         /* sum over launches of the TAR/ego GEMM kernel (when profiling enabled) */
    int64_t gemm_launches;
    double gemm_flops;      /* algorithmic FLOPs of those launches */
    double oar_bytes;       /* algorithmic HBM bytes of the decode steps */
    double attn_ms;         /* sum over launches of the spatial attention kernel (when profiling enabled) */
    int64_t attn_launches;
    double attn_flops;
    double bg_ms;           // busy time of the background stream (next frame's history slots run beside the decode loop)
    int64_t overlapped_frames; /* frames whose TAR stacks only had to compute their last history slot */
    double layers_ms;       /* sum over launches of the decode step's layer kernel(s): the decode engine's one launch per step, or the
                             * 5 x n_oar_layer launches of the five-launch form (when profiling enabled; HIP events on the decode stream) */
    int64_t layers_launches; /* decode steps timed that way */
    int32_t decode_engine;  /* which persistent decode engine ran the last frame's decode steps: 0 none (five launches per layer, or the batched layer:
                             * see decode_batched), 1 the XCD-resident engine (csrc/oar_engine.hip; n_embd 768, up to 23 scenes), 3 the chip-wide engine
                             * of the 2x-width layers (csrc/oar_engine_wide.hip; n_embd 1536).  2 was round 5's multi-scene engine: measured behind the
                             * other paths and removed in round 6 -- the value is not reused */
    int32_t engine_fallback; /* 1 when this configuration would use the decode engine (16-bit mode, n_embd 768) but its census failed at
                              * umgen_create: the five-launch decode layer runs instead (a warning is printed at create) */
    int32_t decode_batched; /* 1 when the last frame's decode steps ran on the batched decode layer (24 and more scenes per call: the scenes
                              * as the matrix-core instruction's columns, csrc/decode_batched.hip) */
    int32_t decode_lanes;   /* ... and on how many decode lanes (sub-batches on their own streams, forked behind the TAR stacks and joined at the
                              * end of the frame; 0 when the batched layer did not run) */
    int64_t prefix_passes;  /* frames whose GIVEN map / bbox3d positions went through the BlockOAR layers as one forward pass (UMGen.py:1184-1201)
                             * instead of one decode step per position */
} umgen_timings;

/* UMGen(config)  -- UMGen.py:53 */
int umgen_create(const umgen_config *cfg, umgen_engine **out);

/* model.load_state_dict(ckpt["module"], strict=False) -- infer_fun.py:43-50.  One call per state-dict entry;
 * unknown keys are ignored (returns 1), known keys are shape-checked and repacked to the kernel layouts. */
int umgen_load_tensor(umgen_engine *e, const char *key, const void *data, int32_t dtype,
                      const int64_t *shape, int32_t ndim);

/* Builds derived tables (GMLP(codebook) rows, sinusoid tables if not loaded) and checks that every
 * tensor the rollout reads has been loaded.  Corresponds to model.eval() readiness. */
int umgen_finalize_weights(umgen_engine *e);

/* UMGen.inference(new_frames, cond_frames, input_cond_frames=T_in, input_cond_tokens, init_tokens, control_test)
 * for B independent scenes (the reference is B = 1; scenes never interact).
 *   pose/map/bbox3d/image : [B][T_in][S_mod] int64 history tokens (first T_in frames are used)
 *   ctrl_pose/ctrl_bbox3d : NULL, or [B][T_ctl][3] / [B][T_ctl][660] control tokens (init_tokens; -1 = free).  Either may be given
 *                           alone: pose only (ego controlled), bbox3d only with control_test (agents controlled, the ego net
 *                           infers the pose; UMGen.py:1438-1473), or both (the reference's control pickles)
 *   given_map / given_bbox3d : NULL, or [B][T_ctl][1024] / [B][T_ctl][660] tokens of the new frames that are GIVEN, not generated
 *                           (init_tokens["map"], init_tokens["bbox3d"] without control_test: infer_oar_net's predefined-token prefix,
 *                           UMGen.py:1184-1201 -- "use the predefined tokens and don't infer these tokens any more").  They must continue
 *                           the pose prefix in scene order: the map, or the map and the boxes (the reference concatenates whatever is
 *                           given back to back, so boxes without the map would sit on the map's positions: refused).  The given
 *                           positions go through the BlockOAR layers as ONE forward pass like the reference's first iteration (no head, no
 *                           sampler; engines created with the overlapped background pass, UMGEN_OVERLAP=1, replay them as decode steps instead
 *                           -- a property of the engine, the same for every frame), sampling starts behind them; the given tokens are
 *                           returned verbatim (UMGen.py:1640-1651).  given_bbox3d and control_test exclude each other.
 *   out_*                 : [B][T_in + new_frames][S_mod], caller-allocated
 */
int umgen_rollout(umgen_engine *e, int32_t B, int32_t T_in, int32_t new_frames, int32_t cond_frames,
                  const int64_t *pose, const int64_t *map, const int64_t *bbox3d, const int64_t *image,
                  int32_t T_ctl, const int64_t *ctrl_pose, const int64_t *ctrl_bbox3d, int32_t control_test,
                  const int64_t *given_map, const int64_t *given_bbox3d,
                  const umgen_sampling *sampling,
                  int64_t *out_pose, int64_t *out_map, int64_t *out_bbox3d, int64_t *out_image);

/* One frame of one scene (UMGen._inference, UMGen.py:1406-1540) with optional trace / teacher forcing.
 * Window tokens are [T][S_mod]; ctrl_* are NULL or the [S_mod] control tokens of this frame.
 * out_* receive the new frame's tokens [S_mod]. */
int umgen_frame(umgen_engine *e, int32_t T, const int64_t *pose, const int64_t *map, const int64_t *bbox3d,
                const int64_t *image, const int64_t *ctrl_pose, const int64_t *ctrl_bbox3d, int32_t control_test,
                const umgen_sampling *sampling, int32_t frame_idx, const umgen_trace *trace,
                int64_t *out_pose, int64_t *out_map, int64_t *out_bbox3d, int64_t *out_image);

int umgen_set_profiling(umgen_engine *e, int32_t enable); /* per-launch HIP-event timing of the GEMM / attention kernels and of the
                                                            * decode step's layer kernel(s); profiled frames launch eagerly */
int umgen_get_timings(umgen_engine *e, umgen_timings *out);

/* ---- scene formats either side of the rollout (host only, no engine needed) ------------------
 * DigitalBinsTokenizer.encode/.decode (tokenizer.py:316-354) + Normalize_Standard (normalize.py:7-76) for the ego motion and
 * Normalize min-max (normalize.py:79-137, 189-229) + the attribute / category tokens of BBox3DTokenizer (tokenizer.py:515-600)
 * for agent boxes.  Arrays are caller-owned; return 0 or UMGEN_E_INVALID. */
int umgen_tokenize_ego(const double *pose_diff /*[n][3] (dx, dy, dheading)*/, int64_t n, int64_t *tokens /*[n][3]*/);
int umgen_detokenize_ego(const int64_t *tokens /*[n][3]*/, int64_t n, float *pose_diff /*[n][3]*/);   /* == UMGen.decode_pose */
int umgen_tokenize_boxes(const float *boxes /*[n][stride], first 10 columns used*/, int64_t n, int32_t stride,
                         const int32_t *category_index /*[n], 0..2*/, int64_t *tokens /*[n][11]*/);
int umgen_detokenize_boxes(const int64_t *slot_tokens /*[n][11]*/, int64_t n, double *boxes /*[n][10]*/);

/* ---- VQ decoders: map / image tokens -> rasters, fp32 ---------------------------------------------
 * NormVQModel.decode_code (projects/tokenizer/vq_model.py:88-103,126-150) = quantize.embedding lookup -> post_quant_conv ->
 * Decoder.forward (projects/tokenizer/vq_modules.py:293-415), as called by Mapdecoder.decode_maps / Imagedecoder.decode_images
 * (projects/tools/decode_map.py:110-183).  The config mirrors the `ddconfig` dicts of vq_model.py:153-202. */
typedef struct umgen_vq umgen_vq; /* opaque */
typedef struct umgen_vq_config {
    int32_t n_embed, embed_dim;        /* codebook: 8192 x 16 */
    int32_t z_channels, ch, out_ch;    /* ddconfig */
    int32_t n_levels;                  /* len(ch_mult) */
    int32_t ch_mult[8];
    int32_t num_res_blocks;
    int32_t n_attn_res;
    int32_t attn_resolutions[4];
    int32_t resolution;                /* ddconfig["resolution"] (only used to place the attention blocks, like the reference) */
    int32_t post_quant_ks, post_quant_pad; /* NormVQModel(stride=..., padding=...): kernel size / padding of post_quant_conv */
    int32_t token_h, token_w;          /* token grid of one frame: 32 x 32 (map), 16 x 32 (image) */
    int32_t device;
} umgen_vq_config;
int umgen_vq_create(const umgen_vq_config *cfg, umgen_vq **out);
/* one call per state-dict entry of the VQ checkpoint (fp32, reference key names: "decoder.conv_in.weight", "post_quant_conv.bias",
 * "quantize.embedding.weight", ...); entries the decode path does not read (encoder.*, quant_conv.*, EMA buffers) return 1 */
int umgen_vq_load_tensor(umgen_vq *d, const char *key, const float *data, const int64_t *shape, int32_t ndim);
int umgen_vq_finalize(umgen_vq *d);
/* codes [n][token_h][token_w] -> out [n][out_ch][token_h * 2^(n_levels-1)][token_w * 2^(n_levels-1)] */
int umgen_vq_decode(umgen_vq *d, int32_t n, const int64_t *codes, float *out);
const char *umgen_vq_last_error(const umgen_vq *d);
int umgen_vq_destroy(umgen_vq *d);

const char *umgen_last_error(const umgen_engine *e); /* never NULL */
const char *umgen_version(void);
int umgen_destroy(umgen_engine *e);

#ifdef __cplusplus
}
#endif
#endif /* UMGEN_H_ */
