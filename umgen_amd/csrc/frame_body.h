// Token embedding of the TAR / ego stacks and the action-aware map warp as device functions (frame.hip launches them as kernels, the decode engine's
// background workers, bg_worker.h, run the same code on their share of the rows).
#pragma once
#include "frame.h"

namespace umgen {

// ---------------------------------------------------------------------------------------------------------
// embeddings (get_mod_emb_pre / add_spatial_pos_emb / add_bos_eos / add_pos_emb, UMGen.py:411-515)
// ---------------------------------------------------------------------------------------------------------
__device__ inline int fixed_aux_id(int s) {   // bos/eos id of scene position s, or -1
    switch (s) {
        case kPoseBos: return 0; case kPoseEos: return 1; case kMapBos: return 2; case kMapEos: return 3;
        case kBoxBos: return 4; case kBoxEos: return 5; case kImgBos: return 6; case kImgEos: return 7;
        default: return -1;
    }
}

// row `row` of the pass by 128 threads (tid: thread of the 128)
__device__ __forceinline__ void embed_stack_row(int stack, const EmbedTables tb, const WindowTokens w, float* __restrict__ X, float* __restrict__ mapfeat,
                                                long row, int tid) {
    const int SS = stack_len(stack);
    const int s = (int)(row % SS);
    const int t = (int)((row / SS) % w.T);
    const int b = (int)(row / ((long)SS * w.T));
    const int E = tb.E;
    const long frl = (long)b * w.T + t;                                        // frame index inside this pass (mapfeat rows)
    const long fr = (long)b * (w.Tfull ? w.Tfull : w.T) + w.t0 + t;            // frame index in the token arrays
    float* xr = X + row * E;
    const float* spe = tb.spe + (long)s * E;
    const float* tpe = tb.tpe + (long)(w.t0 + t) * E;
    const int aux = fixed_aux_id(s);
    if (aux >= 0) {
        const float* a = tb.axe + (long)aux * E;
        for (int c = tid; c < E; c += 128) xr[c] = (a[c] + spe[c]) + tpe[c];
    } else if (s < kPoseEos) {
        const bf16_t* p = tb.fouier_pe + (long)w.pose[fr * kNPose + (s - 1)] * E;
        for (int c = tid; c < E; c += 128) xr[c] = (bf16_to_f32(p[c]) + spe[c]) + tpe[c];
    } else if (s < kMapEos) {
        const int k = s - kMapC0;
        const float* gm = tb.gmap + (long)w.map[fr * kNMap + k] * E;
        if (stack == STACK_EGO) {
            for (int c = tid; c < E; c += 128) xr[c] = (gm[c] + spe[c]) + tpe[c];
        } else {
            float* mf = mapfeat + (frl * kNMap + k) * E;
            if (stack == STACK_TAR) {   // grid-centre positional embedding only in forward_tar_net (UMGen.py:722-726)
                const bf16_t* gp = tb.grid_posi + (long)k * E;
                for (int c = tid; c < E; c += 128) mf[c] = gm[c] + bf16_to_f32(gp[c]);
            } else {
                for (int c = tid; c < E; c += 128) mf[c] = gm[c];
            }
        }
    } else if (s < kBoxEos) {
        const int k = s - kBoxC0;
        const int* bt = w.box + fr * kNBox;
        const int slot = k / kSlotLen;
        const float* be = tb.be + (long)bt[k] * E;
        const bf16_t* px = tb.posi + (long)bt[slot * kSlotLen] * E;
        const bf16_t* py = tb.posi + (long)bt[slot * kSlotLen + 1] * E;
        for (int c = tid; c < E; c += 128) {
            const float pe = bf16_to_f32(f32_to_bf16(bf16_to_f32(px[c]) + bf16_to_f32(py[c])));   // bf16 + bf16 -> bf16
            xr[c] = ((be[c] + pe) + spe[c]) + tpe[c];
        }
    } else {
        const float* gi = tb.gimg + (long)w.img[fr * kNImg + (s - kImgC0)] * E;
        for (int c = tid; c < E; c += 128) xr[c] = (gi[c] + spe[c]) + tpe[c];
    }
}

// ---------------------------------------------------------------------------------------------------------
// affine_transform (UMGen.py:310-354): F.affine_grid + F.grid_sample(bilinear, zeros, align_corners=False) on the
// 32x32 map-feature grid; theta = [[cos(-th), -sin(-th), -dy], [sin(-th), cos(-th), -dx]], dx = 2(dx_m/4)/32.
// ---------------------------------------------------------------------------------------------------------
__device__ inline float base_coord(int i) {   // at::linspace(-1, 1, 32) * 31 / 32  (fp32, symmetric linspace)
    const float step = 2.0f / 31.0f;
    const float v = (i < 16) ? (-1.0f + step * (float)i) : (1.0f - step * (float)(31 - i));
    return (v * 31.0f) / 32.0f;
}

// grid cell `cell` = (b*T + t_local)*1024 + k by 128 threads
__device__ __forceinline__ void warp_map_cell(int stack, const EmbedTables tb, int T, const float* __restrict__ mapfeat, const float* __restrict__ pose_diff,
                                              float* __restrict__ X, float* __restrict__ warped_last, int Tfull, int t0, long cell, int tid) {
    const int SS = stack_len(stack);
    const int k = (int)(cell % kNMap);
    const long fr = cell / kNMap;            // frame index inside this pass
    const int b = (int)(fr / T);
    const int t = t0 + (int)(fr % T);        // history slot
    const long frg = (long)b * Tfull + t;    // frame index in pose_diff
    const int E = tb.E;
    const int hy = k >> 5, wx = k & 31;
    const float dxm = pose_diff[frg * 3 + 0], dym = pose_diff[frg * 3 + 1], th = pose_diff[frg * 3 + 2];
    const float dx = 2.0f * (dxm / 4.0f) / 32.0f, dy = 2.0f * (dym / 4.0f) / 32.0f;
    const float cs = cosf(-th), sn = sinf(-th);
    const float bx = base_coord(wx), by = base_coord(hy);
    const float gx = fmaf(cs, bx, fmaf(-sn, by, -dy));
    const float gy = fmaf(sn, bx, fmaf(cs, by, -dx));
    const float ix = ((gx + 1.0f) * 32.0f - 1.0f) / 2.0f;
    const float iy = ((gy + 1.0f) * 32.0f - 1.0f) / 2.0f;
    const float x0f = floorf(ix), y0f = floorf(iy);
    const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    const float nw = (x0f + 1.0f - ix) * (y0f + 1.0f - iy), ne = (ix - x0f) * (y0f + 1.0f - iy);
    const float sw = (x0f + 1.0f - ix) * (iy - y0f), se = (ix - x0f) * (iy - y0f);
    const float* src = mapfeat + fr * kNMap * E;
    const bool in_nw = (x0 >= 0 && x0 < 32 && y0 >= 0 && y0 < 32), in_ne = (x1 >= 0 && x1 < 32 && y0 >= 0 && y0 < 32);
    const bool in_sw = (x0 >= 0 && x0 < 32 && y1 >= 0 && y1 < 32), in_se = (x1 >= 0 && x1 < 32 && y1 >= 0 && y1 < 32);
    const float* pnw = src + (long)(y0 * 32 + x0) * E;
    const float* pne = src + (long)(y0 * 32 + x1) * E;
    const float* psw = src + (long)(y1 * 32 + x0) * E;
    const float* pse = src + (long)(y1 * 32 + x1) * E;
    const float* f = src + (long)k * E;
    const int s = kMapC0 + k;
    float* xr = X + ((fr * SS) + s) * E;
    const float* spe = tb.spe + (long)s * E;
    const float* tpe = tb.tpe + (long)t * E;
    float* wl = (warped_last && t == Tfull - 1) ? warped_last + ((long)b * kNMap + k) * E : nullptr;
    for (int c = tid; c < E; c += 128) {
        float o = 0.f;
        if (in_nw) o += pnw[c] * nw;
        if (in_ne) o += pne[c] * ne;
        if (in_sw) o += psw[c] * sw;
        if (in_se) o += pse[c] * se;
        xr[c] = ((o + f[c]) + spe[c]) + tpe[c];
        if (wl) wl[c] = o;
    }
}

}  // namespace umgen
