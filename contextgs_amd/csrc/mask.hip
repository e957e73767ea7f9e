// Offset-mask accessor of the model (scene/gaussian_model.py:295-310): get_mask = straight-through binarisation of
// sigmoid(_mask) at 0.01 and get_mask_anchor = "any offset of the anchor alive", evaluated by every training step.
// The torch composition is five element-wise launches for the mask, seven for the anchor flag and three on the way
// back; here it is one pass each way.  Same fp32 expression as the reference, term by term:
//     s = 1 / (1 + exp(-m));  hard = s > 0.01;  mask = (hard - s) + s          (:297-299, the STE value)
//     d m = (g * (1 - s)) * s                                                  (sigmoid backward through the "+ s")
#include "cgs_internal.h"

__device__ __forceinline__ float mask_sigmoid(float m) { return 1.f / (1.f + expf(-m)); }

#define MASK_APB 128      // anchors per workgroup

// element-parallel (coalesced: a workgroup streams the MASK_APB * K consecutive logits of its anchors); the per-anchor
// "any alive" flags are combined in LDS
__global__ void __launch_bounds__(256)
mask_ste_fwd_kernel(const float *__restrict__ logits, int64_t n, int K, float *__restrict__ mask,
                    uint8_t *__restrict__ any_alive) {
    __shared__ uint32_t alive[MASK_APB];
    const int64_t a0 = (int64_t)blockIdx.x * MASK_APB;
    const int na = (int)((n - a0) < MASK_APB ? (n - a0) : MASK_APB);
    if (threadIdx.x < MASK_APB) alive[threadIdx.x] = 0u;
    __syncthreads();
    const int64_t e0 = a0 * K;
    const int ne = na * K;
    for (int t = threadIdx.x; t < ne; t += 256) {
        const float s = mask_sigmoid(logits[e0 + t]);
        const float v = ((s > 0.01f ? 1.f : 0.f) - s) + s;
        if (mask) mask[e0 + t] = v;
        if (v > 0.f) alive[t / K] = 1u;            // benign race: every writer stores 1
    }
    __syncthreads();
    if (any_alive && threadIdx.x < na) any_alive[a0 + threadIdx.x] = alive[threadIdx.x] ? 1 : 0;
}

__global__ void __launch_bounds__(256)
mask_ste_bwd_kernel(const float *__restrict__ logits, const float *__restrict__ g, int64_t n_el, float *__restrict__ d_logits) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_el) return;
    const float s = mask_sigmoid(logits[i]);
    d_logits[i] = (g[i] * (1.f - s)) * s;
}

extern "C" int cgs_mask_ste_fwd(const float *logits, int64_t n, int K, float *mask, uint8_t *any_alive, void *stream) {
    if (n < 0 || K < 1 || (n > 0 && !logits)) { cgs_set_error("mask_ste_fwd: bad args"); return CGS_ERR_ARG; }
    if (n == 0 || (!mask && !any_alive)) return CGS_OK;
    hipLaunchKernelGGL(mask_ste_fwd_kernel, dim3((unsigned)((n + MASK_APB - 1) / MASK_APB)), dim3(256), 0, (hipStream_t)stream, logits, n, K,
                       mask, any_alive);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

extern "C" int cgs_mask_ste_bwd(const float *logits, const float *g, int64_t n_elements, float *d_logits, void *stream) {
    if (n_elements < 0 || (n_elements > 0 && (!logits || !g || !d_logits))) { cgs_set_error("mask_ste_bwd: bad args"); return CGS_ERR_ARG; }
    if (n_elements == 0) return CGS_OK;
    hipLaunchKernelGGL(mask_ste_bwd_kernel, dim3((unsigned)((n_elements + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       logits, g, n_elements, d_logits);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}
