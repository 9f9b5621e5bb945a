// VQ decoders on the GPU: the map VQ-VAE and image VQGAN decoders that turn the rollout's map / image
// tokens back into rasters.  Replaces, value for value in fp32:
//   NormVQModel.decode_code / indices_to_quant + decode      projects/tokenizer/vq_model.py:88-103, 126-150
//   Decoder.forward (conv_in, mid blocks, up levels, norm_out, conv_out)   projects/tokenizer/vq_modules.py:293-415
//   ResnetBlock / AttnBlock / Upsample / Normalize / nonlinearity           vq_modules.py:14-40, 63-176
// as called by Mapdecoder.decode_maps / Imagedecoder.decode_images (projects/tools/decode_map.py:110-183).
//
// Layout: activations are channels-last fp32 [pixel][channel] (one frame at a time), so every convolution is ONE GEMM of this
// library: a 3 x 3 convolution = im2col ([pixel][9 C_in], zero padded) x the repacked kernel [C_out][(ky, kx, c_in)], a 1 x 1
// convolution = the GEMM on the activation rows themselves; the residual add of a ResnetBlock is the GEMM's residual epilogue.
// Arithmetic: exact fp32 FMA chains (launch_gemm_valu<float, float>), fp32 GroupNorm statistics, expf-based sigmoid / softmax --
// the reference runs these decoders in fp32 (no autocast around model_pl.py:366-447).  The attention block (single head of
// C channels over H*W positions) is three GEMMs + a row softmax.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/umgen.h"
#include "kernels.h"

using namespace umgen;

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------------------
// z[p][c] = embedding[code[p]][c]
__global__ void vq_embed_kernel(const long long* __restrict__ codes, const float* __restrict__ emb, int C, long n_px, float* __restrict__ z) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_px * C) return;
    const long p = i / C;
    z[i] = emb[codes[p] * C + (i % C)];
}

// col[p][(ky * KS + kx) * C + c] = x[y + ky - pad][x + kx - pad][c]   (zero outside), KS x KS kernel, stride 1
__global__ void vq_im2col_kernel(const float* __restrict__ x, int H, int W, int C, int KS, int pad, float* __restrict__ col) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;          // one float4 of 4 channels
    const int C4 = C >> 2;
    const long total = (long)H * W * KS * KS * C4;
    if (i >= total) return;
    const int c4 = (int)(i % C4);
    const long r = i / C4;
    const int kk = (int)(r % (KS * KS));
    const long p = r / (KS * KS);
    const int px = (int)(p % W), py = (int)(p / W);
    const int sy = py + kk / KS - pad, sx = px + kk % KS - pad;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sy >= 0 && sy < H && sx >= 0 && sx < W) v = reinterpret_cast<const float4*>(x + ((long)sy * W + sx) * C)[c4];
    reinterpret_cast<float4*>(col + (p * KS * KS + kk) * C)[c4] = v;
}

// GroupNorm(32 groups, eps 1e-6, affine) statistics of one frame: stats[g] = (mean, rstd) over H*W x (C/32) values
__global__ __launch_bounds__(256) void vq_gn_stats_kernel(const float* __restrict__ x, long n_px, int C, float* __restrict__ stats) {
    __shared__ double s_sum[4], s_sq[4];
    const int g = blockIdx.x, cg = C / 32;
    const long n = n_px * cg;
    double sum = 0.0, sq = 0.0;      // (torch accumulates GroupNorm statistics in a wider type on the CPU too)
    for (long i = threadIdx.x; i < n; i += 256) {
        const float v = x[(i / cg) * C + g * cg + (i % cg)];
        sum += v;
        sq += (double)v * v;
    }
    for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o); sq += __shfl_xor(sq, o); }
    if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6] = sum; s_sq[threadIdx.x >> 6] = sq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double s = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3], q = s_sq[0] + s_sq[1] + s_sq[2] + s_sq[3];
        const double mean = s / (double)n;
        const double var = q / (double)n - mean * mean;
        stats[2 * g] = (float)mean;
        stats[2 * g + 1] = (float)(1.0 / sqrt(var + 1e-6));
    }
}
// y = GroupNorm(x) * gamma + beta, optionally followed by x * sigmoid(x) (nonlinearity, vq_modules.py:14-16)
__global__ void vq_gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, long n_px, int C, int swish, float* __restrict__ y) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_px * C) return;
    const int c = (int)(i % C), g = c / (C / 32);
    float v = (x[i] - stats[2 * g]) * stats[2 * g + 1] * gamma[c] + beta[c];
    if (swish) v = v / (1.0f + expf(-v));
    y[i] = v;
}

// nearest-neighbour x2 upsampling (F.interpolate(scale_factor=2, mode="nearest")), channels last
__global__ void vq_upsample_kernel(const float* __restrict__ x, int H, int W, int C, float* __restrict__ y) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int C4 = C >> 2;
    const long total = (long)4 * H * W * C4;
    if (i >= total) return;
    const int c4 = (int)(i % C4);
    const long p = i / C4;
    const int ox = (int)(p % (2 * W)), oy = (int)(p / (2 * W));
    reinterpret_cast<float4*>(y + p * C)[c4] = reinterpret_cast<const float4*>(x + ((long)(oy >> 1) * W + (ox >> 1)) * C)[c4];
}

// row softmax of the attention scores: w[i][:] = softmax(s[i][:] * scale)  (AttnBlock, vq_modules.py:158-160); one wave per row
__global__ __launch_bounds__(256) void vq_softmax_kernel(float* __restrict__ s, int n, float scale) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    float* r = s + (long)row * n;
    float mx = -INFINITY;
    for (int j = lane; j < n; j += 64) mx = fmaxf(mx, r[j] * scale);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < n; j += 64) { const float e = expf(r[j] * scale - mx); r[j] = e; sum += e; }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int j = lane; j < n; j += 64) r[j] *= inv;
}

// out[c][p] (channels first, the reference's output layout) = x[p][c]
__global__ void vq_to_nchw_kernel(const float* __restrict__ x, long n_px, int C, int ldx, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_px * C) return;
    const long p = i % n_px;
    const int c = (int)(i / n_px);
    out[i] = x[p * ldx + c];
}

inline dim3 grid1d(long n, int block = 256) { return dim3((unsigned)((n + block - 1) / block)); }

// ---------------------------------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------------------------------
struct Conv { float* w = nullptr; float* b = nullptr; int cin = 0, cout = 0, cout_pad = 0, ks = 0, pad = 0; bool loaded_w = false, loaded_b = false; };   // cout_pad: rows of w / b (multiple of 4, zero rows behind cout: the GEMM epilogues write 4 features at a time)
struct Norm { float* g = nullptr; float* b = nullptr; int c = 0; bool loaded_g = false, loaded_b = false; };
struct Res { Norm n1, n2; Conv c1, c2, nin; bool has_nin = false; };
struct Attn { Norm n; Conv q, k, v, proj; };
struct Level { std::vector<Res> block; std::vector<Attn> attn; Conv up; bool has_up = false; };

}  // namespace

struct umgen_vq {
    umgen_vq_config cfg{};
    std::string err;
    hipStream_t stream = nullptr;
    std::vector<void*> allocs;
    float* emb = nullptr; bool emb_loaded = false;
    Conv post_quant, conv_in, conv_out;
    Res mid1, mid2;
    Attn mid_attn;
    std::vector<Level> up;     // index = i_level (0 = finest), like Decoder.up
    Norm norm_out;
    // what load_tensor fills: key -> (destination, element count, expected shape, conv to repack or nullptr, flag)
    struct Slot { float* dst; std::vector<int64_t> shape; Conv* repack; bool* flag; };
    std::map<std::string, Slot> slots;
    bool finalized = false;
    // workspace (one frame)
    float *x = nullptr, *h = nullptr, *t = nullptr, *col = nullptr, *stats = nullptr, *scores = nullptr, *q = nullptr, *k = nullptr, *vt = nullptr;
    long long* d_codes = nullptr;
    float* d_out = nullptr;
    int out_h = 0, out_w = 0;

    int fail(int code, const char* fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }
};

#define VQCHK(e, call)                                                                               \
    do {                                                                                             \
        hipError_t _err = (call);                                                                    \
        if (_err != hipSuccess) return (e)->fail(UMGEN_E_HIP, "%s -> %s", #call, hipGetErrorString(_err)); \
    } while (0)

namespace {

int vq_alloc(umgen_vq* e, float** p, size_t n) {
    VQCHK(e, hipMalloc(reinterpret_cast<void**>(p), (n ? n : 4) * sizeof(float)));
    e->allocs.push_back(*p);
    return 0;
}
int reg_conv(umgen_vq* e, const std::string& key, Conv& c, int cin, int cout, int ks, int pad) {
    c.cin = cin; c.cout = cout; c.cout_pad = (cout + 3) & ~3; c.ks = ks; c.pad = pad;
    if (int rc = vq_alloc(e, &c.w, (size_t)c.cout_pad * cin * ks * ks)) return rc;
    if (int rc = vq_alloc(e, &c.b, (size_t)c.cout_pad)) return rc;
    VQCHK(e, hipMemset(c.w, 0, (size_t)c.cout_pad * cin * ks * ks * 4));
    VQCHK(e, hipMemset(c.b, 0, (size_t)c.cout_pad * 4));
    e->slots[key + ".weight"] = umgen_vq::Slot{c.w, {cout, cin, ks, ks}, &c, &c.loaded_w};
    e->slots[key + ".bias"] = umgen_vq::Slot{c.b, {cout}, nullptr, &c.loaded_b};
    return 0;
}
int reg_norm(umgen_vq* e, const std::string& key, Norm& n, int c) {
    n.c = c;
    if (int rc = vq_alloc(e, &n.g, (size_t)c)) return rc;
    if (int rc = vq_alloc(e, &n.b, (size_t)c)) return rc;
    e->slots[key + ".weight"] = umgen_vq::Slot{n.g, {c}, nullptr, &n.loaded_g};
    e->slots[key + ".bias"] = umgen_vq::Slot{n.b, {c}, nullptr, &n.loaded_b};
    return 0;
}
int reg_res(umgen_vq* e, const std::string& key, Res& r, int cin, int cout) {
    if (int rc = reg_norm(e, key + ".norm1", r.n1, cin)) return rc;
    if (int rc = reg_conv(e, key + ".conv1", r.c1, cin, cout, 3, 1)) return rc;
    if (int rc = reg_norm(e, key + ".norm2", r.n2, cout)) return rc;
    if (int rc = reg_conv(e, key + ".conv2", r.c2, cout, cout, 3, 1)) return rc;
    r.has_nin = cin != cout;
    if (r.has_nin) { if (int rc = reg_conv(e, key + ".nin_shortcut", r.nin, cin, cout, 1, 0)) return rc; }
    return 0;
}
int reg_attn(umgen_vq* e, const std::string& key, Attn& a, int c) {
    if (int rc = reg_norm(e, key + ".norm", a.n, c)) return rc;
    if (int rc = reg_conv(e, key + ".q", a.q, c, c, 1, 0)) return rc;
    if (int rc = reg_conv(e, key + ".k", a.k, c, c, 1, 0)) return rc;
    if (int rc = reg_conv(e, key + ".v", a.v, c, c, 1, 0)) return rc;
    return reg_conv(e, key + ".proj_out", a.proj, c, c, 1, 0);
}

// out[p][cout] (= or +=) conv(x)[p][cout] + bias
void conv(umgen_vq* e, const Conv& c, const float* x, int H, int W, float* out, bool residual) {
    const long n_px = (long)H * W;
    const float* act = x;
    int K = c.cin;
    if (c.ks > 1) {
        const long total = n_px * c.ks * c.ks * (c.cin / 4);
        hipLaunchKernelGGL(vq_im2col_kernel, grid1d(total), dim3(256), 0, e->stream, x, H, W, c.cin, c.ks, c.pad, e->col);
        act = e->col;
        K = c.cin * c.ks * c.ks;
    }
    GemmArgs g{};
    g.P = c.w; g.Q = act; g.Mi = c.cout_pad; g.Nj = (int)n_px; g.K = K; g.ldp = K; g.ldq = K; g.batch = 1;
    g.mode = residual ? GEMM_RESID : GEMM_STORE; g.bias = c.b; g.out = out; g.ldo = c.cout_pad;
    launch_gemm_valu<float, float>(e->stream, g);
}
void group_norm(umgen_vq* e, const Norm& n, const float* x, long n_px, bool swish, float* y) {
    hipLaunchKernelGGL(vq_gn_stats_kernel, dim3(32), dim3(256), 0, e->stream, x, n_px, n.c, e->stats);
    hipLaunchKernelGGL(vq_gn_apply_kernel, grid1d(n_px * n.c), dim3(256), 0, e->stream, x, e->stats, n.g, n.b, n_px, n.c, swish ? 1 : 0, y);
}
// ResnetBlock.forward (vq_modules.py:108-128), temb = None, dropout 0: x (in e->x, C_in) -> e->x (C_out)
void res_block(umgen_vq* e, const Res& r, int H, int W) {
    const long n_px = (long)H * W;
    group_norm(e, r.n1, e->x, n_px, true, e->h);
    conv(e, r.c1, e->h, H, W, e->t, false);
    group_norm(e, r.n2, e->t, n_px, true, e->h);
    if (r.has_nin) {
        conv(e, r.nin, e->x, H, W, e->t, false);      // x = nin_shortcut(x)
        std::swap(e->x, e->t);
    }
    conv(e, r.c2, e->h, H, W, e->x, true);            // x + conv2(h)
}
// AttnBlock.forward (vq_modules.py:150-176): x += proj_out(softmax(q k^T / sqrt(C)) v)
void attn_block(umgen_vq* e, const Attn& a, int H, int W) {
    const int n = H * W, C = a.n.c;
    group_norm(e, a.n, e->x, n, false, e->h);
    conv(e, a.q, e->h, H, W, e->q, false);
    conv(e, a.k, e->h, H, W, e->k, false);
    {   // v^T[c][j] = sum_k Wv[c][k] h[j][k] + bv[c]: channels-first so that it is the K-contiguous operand of the second product
        GemmArgs g{};
        g.P = e->h; g.Q = a.v.w; g.Mi = n; g.Nj = C; g.K = C; g.ldp = C; g.ldq = C; g.batch = 1;
        g.mode = GEMM_VT; g.bias = a.v.b; g.out = e->vt; g.ldo = n; g.H = C / kHeadDim;   // (row index (j / 48) * 48 + j % 48 = j)
        launch_gemm_valu<float, float>(e->stream, g);
    }
    {   // scores[i][j] = sum_c q[i][c] k[j][c]
        GemmArgs g{};
        g.P = e->k; g.Q = e->q; g.Mi = n; g.Nj = n; g.K = C; g.ldp = C; g.ldq = C; g.batch = 1;
        g.mode = GEMM_STORE; g.out = e->scores; g.ldo = n;
        launch_gemm_valu<float, float>(e->stream, g);
    }
    hipLaunchKernelGGL(vq_softmax_kernel, dim3((n + 3) / 4), dim3(256), 0, e->stream, e->scores, n, 1.0f / sqrtf((float)C));   // int(c) ** (-0.5)
    {   // h[i][c] = sum_j w[i][j] v^T[c][j]
        GemmArgs g{};
        g.P = e->vt; g.Q = e->scores; g.Mi = C; g.Nj = n; g.K = n; g.ldp = n; g.ldq = n; g.batch = 1;
        g.mode = GEMM_STORE; g.out = e->h; g.ldo = C;
        launch_gemm_valu<float, float>(e->stream, g);
    }
    conv(e, a.proj, e->h, H, W, e->x, true);
}

}  // namespace

extern "C" {

const char* umgen_vq_last_error(const umgen_vq* e) { return e ? e->err.c_str() : "null decoder"; }

int umgen_vq_create(const umgen_vq_config* cfg, umgen_vq** out) {
    if (!cfg || !out) return UMGEN_E_INVALID;
    *out = nullptr;
    umgen_vq* e = new umgen_vq();
    *out = e;
    e->cfg = *cfg;
    if (cfg->n_levels < 1 || cfg->n_levels > 8 || cfg->num_res_blocks < 1 || cfg->ch < 32 || cfg->ch % 32 != 0)
        return e->fail(UMGEN_E_INVALID, "levels %d / res blocks %d / ch %d", cfg->n_levels, cfg->num_res_blocks, cfg->ch);
    if (cfg->embed_dim % 4 != 0 || cfg->z_channels % 4 != 0) return e->fail(UMGEN_E_UNSUPPORTED, "embed_dim and z_channels must be multiples of 4");
    if (cfg->post_quant_ks != 1 && cfg->post_quant_ks != 3) return e->fail(UMGEN_E_UNSUPPORTED, "post_quant_conv kernel size 1 or 3");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return e->fail(UMGEN_E_HIP, "no HIP device visible: libumgen_hip has no CPU fallback");
    VQCHK(e, hipSetDevice(cfg->device));
    VQCHK(e, gemm256_prepare());   // per device (the decoder's convolutions are this library's GEMMs)
    VQCHK(e, hipStreamCreate(&e->stream));
    const int L = cfg->n_levels;
    // Decoder.__init__ (vq_modules.py:294-383)
    int block_in = cfg->ch * cfg->ch_mult[L - 1];
    int curr_res = cfg->resolution >> (L - 1);
    if (int rc = vq_alloc(e, &e->emb, (size_t)cfg->n_embed * cfg->embed_dim)) return rc;
    e->slots["quantize.embedding.weight"] = umgen_vq::Slot{e->emb, {cfg->n_embed, cfg->embed_dim}, nullptr, &e->emb_loaded};
    if (int rc = reg_conv(e, "post_quant_conv", e->post_quant, cfg->embed_dim, cfg->z_channels, cfg->post_quant_ks, cfg->post_quant_pad)) return rc;
    if (int rc = reg_conv(e, "decoder.conv_in", e->conv_in, cfg->z_channels, block_in, 3, 1)) return rc;
    if (int rc = reg_res(e, "decoder.mid.block_1", e->mid1, block_in, block_in)) return rc;
    if (int rc = reg_attn(e, "decoder.mid.attn_1", e->mid_attn, block_in)) return rc;
    if (int rc = reg_res(e, "decoder.mid.block_2", e->mid2, block_in, block_in)) return rc;
    e->up.resize(L);
    int max_c = block_in;
    for (int lv = L - 1; lv >= 0; --lv) {
        Level& u = e->up[lv];
        const int block_out = cfg->ch * cfg->ch_mult[lv];
        bool at = false;
        for (int a = 0; a < cfg->n_attn_res; ++a) at = at || cfg->attn_resolutions[a] == curr_res;
        u.block.resize(cfg->num_res_blocks + 1);
        if (at) u.attn.resize(cfg->num_res_blocks + 1);
        for (int b = 0; b <= cfg->num_res_blocks; ++b) {
            const std::string key = "decoder.up." + std::to_string(lv) + ".block." + std::to_string(b);
            if (int rc = reg_res(e, key, u.block[b], block_in, block_out)) return rc;
            block_in = block_out;
            if (at) { if (int rc = reg_attn(e, "decoder.up." + std::to_string(lv) + ".attn." + std::to_string(b), u.attn[b], block_in)) return rc; }
        }
        max_c = std::max(max_c, block_out);
        if (lv != 0) {
            u.has_up = true;
            if (int rc = reg_conv(e, "decoder.up." + std::to_string(lv) + ".upsample.conv", u.up, block_in, block_in, 3, 1)) return rc;
            curr_res *= 2;
        }
    }
    if (int rc = reg_norm(e, "decoder.norm_out", e->norm_out, block_in)) return rc;
    if (int rc = reg_conv(e, "decoder.conv_out", e->conv_out, block_in, cfg->out_ch, 3, 1)) return rc;
    // workspace for one frame at the finest level (the largest H * W * C products)
    e->out_h = cfg->token_h << (L - 1);
    e->out_w = cfg->token_w << (L - 1);
    size_t act = 0, colsz = 0;
    {
        int c_in = cfg->ch * cfg->ch_mult[L - 1], H = cfg->token_h, W = cfg->token_w;
        act = std::max(act, (size_t)H * W * std::max(c_in, std::max(cfg->embed_dim, cfg->z_channels)));
        colsz = std::max(colsz, (size_t)H * W * 9 * std::max(c_in, std::max(cfg->embed_dim, cfg->z_channels)));
        for (int lv = L - 1; lv >= 0; --lv) {
            const int c_out = cfg->ch * cfg->ch_mult[lv];
            act = std::max(act, (size_t)H * W * std::max(c_in, c_out));
            colsz = std::max(colsz, (size_t)H * W * 9 * std::max(c_in, c_out));
            c_in = c_out;
            if (lv != 0) {
                H *= 2; W *= 2;
                act = std::max(act, (size_t)H * W * c_in);
                colsz = std::max(colsz, (size_t)H * W * 9 * c_in);
            }
        }
    }
    for (float** p : {&e->x, &e->h, &e->t}) { if (int rc = vq_alloc(e, p, act)) return rc; }
    if (int rc = vq_alloc(e, &e->col, colsz)) return rc;
    if (int rc = vq_alloc(e, &e->stats, 64)) return rc;
    // attention workspaces: the positions of the coarsest level (attention only exists where curr_res is in attn_resolutions; the
    // mid block always has one): bounded by token_h * token_w * 4^(levels with attention) -- allocate for the finest attention level
    {
        long n_att = (long)cfg->token_h * cfg->token_w;
        int res = cfg->resolution >> (L - 1);
        long n = n_att;
        for (int lv = L - 1; lv >= 0; --lv) {
            for (int a = 0; a < cfg->n_attn_res; ++a)
                if (cfg->attn_resolutions[a] == res) n_att = std::max(n_att, n);
            if (lv != 0) { res *= 2; n *= 4; }
        }
        if (n_att > 16384) return e->fail(UMGEN_E_UNSUPPORTED, "attention over %ld positions (scores would need %ld MB)", n_att, n_att * n_att * 4 >> 20);
        if (int rc = vq_alloc(e, &e->scores, (size_t)n_att * n_att)) return rc;
        for (float** p : {&e->q, &e->k, &e->vt}) { if (int rc = vq_alloc(e, p, (size_t)n_att * max_c)) return rc; }
    }
    VQCHK(e, hipMalloc(reinterpret_cast<void**>(&e->d_codes), (size_t)cfg->token_h * cfg->token_w * sizeof(long long)));
    e->allocs.push_back(e->d_codes);
    if (int rc = vq_alloc(e, &e->d_out, (size_t)cfg->out_ch * e->out_h * e->out_w)) return rc;
    return UMGEN_OK;
}

int umgen_vq_load_tensor(umgen_vq* e, const char* key, const float* data, const int64_t* shape, int32_t ndim) {
    if (!e || !key || !data) return UMGEN_E_INVALID;
    auto it = e->slots.find(key);
    if (it == e->slots.end()) return 1;      // encoder.*, quant_conv.*, EMA buffers: not read by the decode path
    umgen_vq::Slot& s = it->second;
    if ((size_t)ndim != s.shape.size()) return e->fail(UMGEN_E_INVALID, "%s: ndim %d, expected %zu", key, ndim, s.shape.size());
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        if (shape[i] != s.shape[i]) return e->fail(UMGEN_E_INVALID, "%s: dim %d is %lld, expected %lld", key, i, (long long)shape[i], (long long)s.shape[i]);
        n *= (size_t)shape[i];
    }
    if (s.repack && s.repack->ks > 1) {     // [C_out][C_in][ky][kx] -> [C_out][(ky, kx, c_in)]: the im2col column order
        const Conv& c = *s.repack;
        std::vector<float> r(n);
        const int kk = c.ks * c.ks;
        for (int o = 0; o < c.cout; ++o)
            for (int i = 0; i < c.cin; ++i)
                for (int k = 0; k < kk; ++k) r[((size_t)o * kk + k) * c.cin + i] = data[((size_t)o * c.cin + i) * kk + k];
        VQCHK(e, hipMemcpy(s.dst, r.data(), n * 4, hipMemcpyHostToDevice));
    } else {
        VQCHK(e, hipMemcpy(s.dst, data, n * 4, hipMemcpyHostToDevice));
    }
    *s.flag = true;
    e->finalized = false;
    return UMGEN_OK;
}

int umgen_vq_finalize(umgen_vq* e) {
    if (!e) return UMGEN_E_INVALID;
    int nmiss = 0;
    std::string first;
    for (auto& kv : e->slots)
        if (!*kv.second.flag) { if (!nmiss) first = kv.first; ++nmiss; }
    if (nmiss) return e->fail(UMGEN_E_STATE, "%d decoder tensors not loaded (e.g. %s)", nmiss, first.c_str());
    e->finalized = true;
    return UMGEN_OK;
}

// codes [n][token_h][token_w] -> out [n][out_ch][H][W] (the layout NormVQModel.decode_code returns)
int umgen_vq_decode(umgen_vq* e, int32_t n, const int64_t* codes, float* out) {
    if (!e || !codes || !out || n < 0) return UMGEN_E_INVALID;
    if (!e->finalized) return e->fail(UMGEN_E_STATE, "umgen_vq_finalize has not been called");
    const umgen_vq_config& cfg = e->cfg;
    const int L = cfg.n_levels;
    const long n_tok = (long)cfg.token_h * cfg.token_w;
    for (long i = 0; i < (long)n * n_tok; ++i)
        if (codes[i] < 0 || codes[i] >= cfg.n_embed) return e->fail(UMGEN_E_INVALID, "code %lld at flat index %ld is outside [0, %d)", (long long)codes[i], i, cfg.n_embed);
    VQCHK(e, hipSetDevice(cfg.device));
    for (int f = 0; f < n; ++f) {
        int H = cfg.token_h, W = cfg.token_w;
        VQCHK(e, hipMemcpyAsync(e->d_codes, codes + (long)f * n_tok, n_tok * sizeof(long long), hipMemcpyHostToDevice, e->stream));
        hipLaunchKernelGGL(vq_embed_kernel, grid1d(n_tok * cfg.embed_dim), dim3(256), 0, e->stream, e->d_codes, e->emb, cfg.embed_dim, n_tok, e->h);
        conv(e, e->post_quant, e->h, H, W, e->t, false);                  // NormVQModel.decode: post_quant_conv
        conv(e, e->conv_in, e->t, H, W, e->x, false);                     // Decoder.forward
        res_block(e, e->mid1, H, W);
        attn_block(e, e->mid_attn, H, W);
        res_block(e, e->mid2, H, W);
        for (int lv = L - 1; lv >= 0; --lv) {
            const Level& u = e->up[lv];
            for (int b = 0; b <= cfg.num_res_blocks; ++b) {
                res_block(e, u.block[b], H, W);
                if (!u.attn.empty()) attn_block(e, u.attn[b], H, W);
            }
            if (u.has_up) {
                hipLaunchKernelGGL(vq_upsample_kernel, grid1d((long)4 * H * W * (u.up.cin / 4)), dim3(256), 0, e->stream, e->x, H, W, u.up.cin, e->h);
                H *= 2; W *= 2;
                conv(e, u.up, e->h, H, W, e->x, false);
            }
        }
        group_norm(e, e->norm_out, e->x, (long)H * W, true, e->h);
        conv(e, e->conv_out, e->h, H, W, e->t, false);
        hipLaunchKernelGGL(vq_to_nchw_kernel, grid1d((long)H * W * cfg.out_ch), dim3(256), 0, e->stream, e->t, (long)H * W, cfg.out_ch, e->conv_out.cout_pad, e->d_out);
        VQCHK(e, hipMemcpyAsync(out + (size_t)f * cfg.out_ch * H * W, e->d_out, (size_t)cfg.out_ch * H * W * 4, hipMemcpyDeviceToHost, e->stream));
        VQCHK(e, hipStreamSynchronize(e->stream));
    }
    VQCHK(e, hipGetLastError());
    return UMGEN_OK;
}

int umgen_vq_destroy(umgen_vq* e) {
    if (!e) return UMGEN_OK;
    (void)hipSetDevice(e->cfg.device);
    (void)hipDeviceSynchronize();
    for (void* p : e->allocs) (void)hipFree(p);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
    return UMGEN_OK;
}

}  // extern "C"
