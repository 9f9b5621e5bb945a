// Host-side (de)tokenisers and normalisers of the scene formats either side of the rollout, as
// C entry points of libumgen_hip.so.  They replace, value for value:
//   DigitalBinsTokenizer.encode / .decode   projects/plugin/data/transforms/tokenizer.py:316-330 / 332-354
//   Normalize_Standard (ego)                projects/plugin/data/transforms/normalize.py:7-76
//   Normalize.normalize_* / unnormalize_*   projects/plugin/data/transforms/normalize.py:79-137, 189-229   (min-max per attribute)
//   BBox3DTokenizer attribute / category tokens   tokenizer.py:515-600 (the per-box part; track slotting stays in scene_io.py)
// numpy's evaluation order and dtypes are followed operation by operation (float32 boxes, float64 ego motion, float64 bin edges,
// np.linspace's start + i * step with the last edge pinned to stop), with FP contraction off, so the token ids are identical:
// tests/test_scene_io.py pins them on vectors recorded from the reference's own dataset class.
#include <cmath>
#include <cstdint>

#include "../../include/umgen.h"

#pragma clang fp contract(off)

namespace {

constexpr int kBins = 1024;
const double kBoxLo[10] = {-64, -64, -5, 0, 0, 0, -3.14, -20, -15, -0.3};   // config.py:126-137
const double kBoxHi[10] = {64, 64, 5, 15, 4, 5, 3.14, 20, 15, 0.3};
const float kEgoStd[3] = {10.0f, 4.0f, 1.0f};                                // config.py:223-231

inline double edge(int i, double start, double stop) {   // np.linspace(start, stop, 1024)[i]
    if (i >= kBins - 1) return stop;
    const double step = (stop - start) / (double)(kBins - 1);
    volatile double t = (double)i * step;
    return t + start;
}
// np.digitize(x, bins) for increasing bins (right=False): number of edges <= x, then np.clip(., 0, 1023)
inline int64_t digitize_clip(double x, double start, double stop) {
    if (std::isnan(x)) return kBins - 1;          // np.digitize puts NaN past the last edge
    int lo = 0, hi = kBins;                       // first index with edge > x
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (edge(mid, start, stop) <= x) lo = mid + 1; else hi = mid;
    }
    return lo > kBins - 1 ? kBins - 1 : lo;
}
inline double midpoint(int64_t tok, double start, double stop) {   // (bins[clip(t - 1)] + bins[clip(t)]) / 2
    const int r = (int)(tok < 0 ? 0 : (tok > kBins - 1 ? kBins - 1 : tok));
    const int l = (int)(tok - 1 < 0 ? 0 : (tok - 1 > kBins - 1 ? kBins - 1 : tok - 1));
    volatile double s = edge(l, start, stop) + edge(r, start, stop);
    return s / 2.0;
}

}  // namespace

extern "C" {

int umgen_tokenize_ego(const double* pose_diff, int64_t n, int64_t* tokens) {
    if ((!pose_diff || !tokens) && n) return UMGEN_E_INVALID;
    for (int64_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) {
            const float inv_std = 1.0f / kEgoStd[a];                    // np.float32 reciprocal (normalize.py:26)
            volatile double d = pose_diff[i * 3 + a] - (double)0.0f;    // float64 - float32 mean
            const double v = d * (double)inv_std;
            tokens[i * 3 + a] = digitize_clip(v, -1.0, 1.0);
        }
    return UMGEN_OK;
}

int umgen_detokenize_ego(const int64_t* tokens, int64_t n, float* pose_diff) {
    if ((!pose_diff || !tokens) && n) return UMGEN_E_INVALID;
    for (int64_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) {
            const float inv_std = 1.0f / kEgoStd[a];
            volatile double u = midpoint(tokens[i * 3 + a], -1.0, 1.0) / (double)inv_std;
            pose_diff[i * 3 + a] = (float)(u + 0.0);
        }
    return UMGEN_OK;
}

int umgen_tokenize_boxes(const float* boxes, int64_t n, int32_t stride, const int32_t* category_index, int64_t* tokens) {
    if ((!boxes || !tokens || !category_index) && n) return UMGEN_E_INVALID;
    if (stride < 10) return UMGEN_E_INVALID;
    for (int64_t i = 0; i < n; ++i) {
        for (int a = 0; a < 10; ++a) {
            // float32 column, python-scalar range: (col - lo) / (hi - lo) stays float32 in numpy
            volatile float num = boxes[i * stride + a] - (float)kBoxLo[a];
            const float v = num / (float)(kBoxHi[a] - kBoxLo[a]);
            tokens[i * 11 + a] = digitize_clip((double)v, 0.0, 1.0);
        }
        if (category_index[i] < 0 || category_index[i] > 2) return UMGEN_E_INVALID;
        tokens[i * 11 + 10] = 1024 + category_index[i];
    }
    return UMGEN_OK;
}

int umgen_detokenize_boxes(const int64_t* slot_tokens, int64_t n, double* boxes) {
    if ((!boxes || !slot_tokens) && n) return UMGEN_E_INVALID;
    for (int64_t i = 0; i < n; ++i)
        for (int a = 0; a < 10; ++a) {
            volatile double p = midpoint(slot_tokens[i * 11 + a], 0.0, 1.0) * (kBoxHi[a] - kBoxLo[a]);
            boxes[i * 10 + a] = p + kBoxLo[a];
        }
    return UMGEN_OK;
}

}  // extern "C"
