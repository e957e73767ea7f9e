// The three parameter means the rate model passes to Entropy_gaussian as clamp centres
// (scene/gaussian_model.py:1664-1668: x_mean = _anchor_feat.mean(), get_scaling.mean(), _offset.mean()): one launch
// over the three tensors (exp applied to the scaling logits on the fly) instead of exp + three torch reductions.
// Deterministic: per-block partials in double, the last block to finish adds them in block order.
#include "cgs_internal.h"

#define M3_BLOCKS 1024

__device__ __forceinline__ double m3_block_sum(double v, double *sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ void __launch_bounds__(256)
means3_kernel(const float *__restrict__ a, int64_t na, const float *__restrict__ b, int64_t nb, int exp_b,
              const float *__restrict__ c, int64_t nc, double *__restrict__ partial, unsigned int *counter,
              float *__restrict__ out) {
    __shared__ double sh[4];
    __shared__ bool last;
    const int64_t stride = (int64_t)gridDim.x * 256, t0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double sa = 0.0, sb = 0.0, sc = 0.0;
    for (int64_t i = t0; i < na; i += stride) sa += (double)a[i];
    if (exp_b) { for (int64_t i = t0; i < nb; i += stride) sb += (double)expf(b[i]); }
    else       { for (int64_t i = t0; i < nb; i += stride) sb += (double)b[i]; }
    for (int64_t i = t0; i < nc; i += stride) sc += (double)c[i];
    sa = m3_block_sum(sa, sh);
    sb = m3_block_sum(sb, sh);
    sc = m3_block_sum(sc, sh);
    if (threadIdx.x == 0) {
        // (published by device-scope exchanges, no fence: cgs_ticket_last in cgs_internal.h)
        cgs_publish(&partial[blockIdx.x], sa);
        cgs_publish(&partial[gridDim.x + blockIdx.x], sb);
        cgs_publish(&partial[2 * gridDim.x + blockIdx.x], sc);
        last = cgs_ticket_last(counter);
    }
    __syncthreads();
    if (!last) return;
    const int64_t n[3] = {na, nb, nc};
    for (int k = 0; k < 3; ++k) {
        double v = 0.0;
        for (int j = threadIdx.x; j < (int)gridDim.x; j += 256) v += cgs_published(&partial[k * gridDim.x + j]);
        v = m3_block_sum(v, sh);
        if (threadIdx.x == 0) out[k] = n[k] > 0 ? (float)(v / (double)n[k]) : 0.f;
    }
}

extern "C" size_t cgs_means3_scratch_bytes(void) { return (size_t)3 * M3_BLOCKS * sizeof(double) + CGS_TICKET_BYTES; }

extern "C" int cgs_means3(const float *a, int64_t na, const float *b, int64_t nb, int exp_b, const float *c, int64_t nc,
                          void *scratch, size_t scratch_bytes, float *out3, void *stream) {
    if (na < 0 || nb < 0 || nc < 0 || !out3 || !scratch || scratch_bytes < cgs_means3_scratch_bytes()) {
        cgs_set_error("means3: bad args");
        return CGS_ERR_ARG;
    }
    if ((na && !a) || (nb && !b) || (nc && !c)) { cgs_set_error("means3: NULL input"); return CGS_ERR_ARG; }
    double *partial = (double *)scratch;
    unsigned int *counter = (unsigned int *)((char *)scratch + (size_t)3 * M3_BLOCKS * sizeof(double));
    CGS_CHECK_HIP(hipMemsetAsync(counter, 0, CGS_TICKET_BYTES, (hipStream_t)stream));
    hipLaunchKernelGGL(means3_kernel, dim3(M3_BLOCKS), dim3(256), 0, (hipStream_t)stream, a, na, b, nb, exp_b, c, nc, partial,
                       counter, out3);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}
