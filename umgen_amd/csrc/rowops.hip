// Row-wise kernels: LayerNorm (module.py:26-37: weight only, eps 1e-5, fp32 statistics).
#include "bg_queue.h"
#include "rowops_body.h"

namespace umgen {

template <typename T>
void launch_layernorm(hipStream_t s, const float* x, long row_stride, long n_rows, int E, const float* w, T* out) {
    if (n_rows <= 0) return;
    if (BgRecorder* rec = g_bg_rec) {      // the next frame's pass on the decode engine's background workers: 8 rows (one per wave) per unit
        if (sizeof(T) != 2) { rec->failed = "fp32 LayerNorm output"; return; }
        BgOp& o = rec->add(BG_LN, 0, (n_rows + 7) / 8, 256, 40);
        o.h.i0 = E; o.a.l0 = row_stride; o.a.l1 = n_rows;
        o.a.p0 = const_cast<float*>(x); o.a.p1 = const_cast<float*>(w); o.a.p2 = out;
        return;
    }
    hipLaunchKernelGGL(layernorm_kernel<T>, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, s, x, row_stride, n_rows, E, w, out);
}
template void launch_layernorm<float>(hipStream_t, const float*, long, long, int, const float*, float*);
template void launch_layernorm<bf16_t>(hipStream_t, const float*, long, long, int, const float*, bf16_t*);
template void launch_layernorm<f16_t>(hipStream_t, const float*, long, long, int, const float*, f16_t*);

}  // namespace umgen
