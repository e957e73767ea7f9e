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

// ---- anchor pruning / growing surgery (scene/gaussian_model.py:673-760, `cat_tensors_to_optimizer`,
// `_prune_anchor_optimizer`): dst_t[r] = src_t[idx[r]] for r < n_keep, for up to CGS_COMPACT_MAX row-major fp32 tensors
// (the eight per-anchor parameters with their two Adam moments, the four statistics buffers) in ONE launch — the
// reference indexes each of them with the same boolean mask, i.e. 24+ (mask scan + gather) launch pairs.
// clamp_col0 >= 0 (per tensor): columns >= clamp_col0 are clamped to <= clamp_max on the way (the log-scale cap that
// `_prune_anchor_optimizer` applies to `scaling[:, 3:]`, :741-745).
#define CGS_COMPACT_MAX 32
struct CompactArgs {
    const float *src[CGS_COMPACT_MAX];
    float *dst[CGS_COMPACT_MAX];
    int width[CGS_COMPACT_MAX];
    int clamp_col0[CGS_COMPACT_MAX];
    int nt;
    float clamp_max;
};

__global__ void __launch_bounds__(256) compact_rows_kernel(CompactArgs a, const int64_t *__restrict__ idx, int64_t n_keep) {
    // a workgroup walks a block of 64 destination rows through every tensor: the index loads are shared, and the
    // elements of one tensor's row block are contiguous in the destination (coalesced stores)
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int rows = (int)min((int64_t)64, n_keep - r0);
    __shared__ int64_t sidx[64];
    if (threadIdx.x < rows) sidx[threadIdx.x] = idx[r0 + threadIdx.x];
    __syncthreads();
    for (int t = 0; t < a.nt; ++t) {
        const float *src = nullptr; float *dst = nullptr; int w = 0, c0 = -1;
#pragma unroll
        for (int k = 0; k < CGS_COMPACT_MAX; ++k)
            if (k == t) { src = a.src[k]; dst = a.dst[k]; w = a.width[k]; c0 = a.clamp_col0[k]; }
        const int total = rows * w;
        for (int e = threadIdx.x; e < total; e += 256) {
            const int r = e / w, c = e - r * w;
            float v = src[sidx[r] * w + c];
            if (c0 >= 0 && c >= c0) v = fminf(v, a.clamp_max);
            dst[(r0 + r) * w + c] = v;
        }
    }
}

extern "C" int cgs_compact_rows(int nt, const float *const *src, float *const *dst, const int *width,
                                const int *clamp_col0, float clamp_max, const int64_t *idx, int64_t n_keep, void *stream) {
    if (nt < 0 || nt > CGS_COMPACT_MAX || n_keep < 0) { cgs_set_error("compact_rows: bad args"); return CGS_ERR_ARG; }
    if (nt == 0 || n_keep == 0) return CGS_OK;
    if (!src || !dst || !width || !idx) { cgs_set_error("compact_rows: NULL"); return CGS_ERR_ARG; }
    CompactArgs a;
    for (int t = 0; t < CGS_COMPACT_MAX; ++t) {
        const bool on = t < nt;
        if (on && (!src[t] || !dst[t] || width[t] < 1)) { cgs_set_error("compact_rows: NULL tensor or width < 1"); return CGS_ERR_ARG; }
        a.src[t] = on ? src[t] : nullptr; a.dst[t] = on ? dst[t] : nullptr; a.width[t] = on ? width[t] : 0;
        a.clamp_col0[t] = (on && clamp_col0) ? clamp_col0[t] : -1;
    }
    a.nt = nt;
    a.clamp_max = clamp_max;
    hipLaunchKernelGGL(compact_rows_kernel, dim3((unsigned)((n_keep + 63) / 64)), dim3(256), 0, (hipStream_t)stream, a, idx, n_keep);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}
