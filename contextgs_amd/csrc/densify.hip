// Densification statistics of the training iteration (scene/gaussian_model.py:696-713, `training_statis`) —
// the direct consumer of viewspace_points.grad, radii > 0 and the expansion's
// selection mask.  The reference runs ~15 torch launches with three boolean-mask index_puts (each a device scan +
// scatter, two of them over all N*K offsets); here it is one pass over the visible slots, every accumulator
// element written by exactly one thread (no atomics).
#include "cgs_internal.h"

__global__ void __launch_bounds__(256)
    densify_stats_kernel(int64_t n_slots, int K, const int64_t *__restrict__ vis_idx, const float *__restrict__ opacity,
                         const uint8_t *__restrict__ sel, const int64_t *__restrict__ sel_pos,
                         const uint8_t *__restrict__ update_filter, const float *__restrict__ grad,
                         float *__restrict__ opacity_accum, float *__restrict__ anchor_demon,
                         float *__restrict__ offset_gradient_accum, float *__restrict__ offset_denom) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n_slots) return;
    const int64_t a = s / K;
    const int k = (int)(s - a * K);
    const int64_t anchor = vis_idx[a];
    if (sel[s]) {
        const int64_t j = sel_pos[s];                     // index of this slot's Gaussian among the selected ones
        if (update_filter[j]) {
            const float gx = grad[3 * j], gy = grad[3 * j + 1];
            offset_gradient_accum[anchor * K + k] += sqrtf(gx * gx + gy * gy);      // ||grad[:, :2]||  (:710)
            offset_denom[anchor * K + k] += 1.f;
        }
    }
    if (k == 0) {
        float sum = 0.f;
        for (int kk = 0; kk < K; ++kk) sum += fmaxf(opacity[a * K + kk], 0.f);   // temp_opacity[temp_opacity < 0] = 0; sum(dim=1)
        opacity_accum[anchor] += sum;
        anchor_demon[anchor] += 1.f;
    }
}

extern "C" int cgs_densify_stats(int64_t n_vis, int K, const int64_t *vis_idx, const float *opacity, const uint8_t *sel,
                                 const int64_t *sel_pos, const uint8_t *update_filter, const float *grad,
                                 float *opacity_accum, float *anchor_demon, float *offset_gradient_accum,
                                 float *offset_denom, void *stream) {
    if (n_vis < 0 || K < 1) { cgs_set_error("densify_stats: bad args"); return CGS_ERR_ARG; }
    if (n_vis == 0) return CGS_OK;
    if (!vis_idx || !opacity || !sel || !sel_pos || !update_filter || !grad || !opacity_accum || !anchor_demon ||
        !offset_gradient_accum || !offset_denom) {
        cgs_set_error("densify_stats: NULL");
        return CGS_ERR_ARG;
    }
    const int64_t n = n_vis * K;
    hipLaunchKernelGGL(densify_stats_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, K,
                       vis_idx, opacity, sel, sel_pos, update_filter, grad, opacity_accum, anchor_demon,
                       offset_gradient_accum, offset_denom);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}
