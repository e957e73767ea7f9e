// Quantisers and the Gaussian rate model as single
// fused, coalesced passes.  HBM-bound elementwise work: one read of each
// operand, one write of each result, no temporaries (the reference runs each
// of these as 6-15 separate torch kernels with full-size intermediates, and
// its Low_bound.backward bounces through host numpy).
//
// Reference semantics: utils/encodings.py:203-231 (STE_multistep,
// Quantize_anchor), utils/entropy_models.py:30-50,141-156 (Entropy_gaussian,
// Low_bound).
#include "cgs_internal.h"

#define EW_THREADS 256

// torch.div(a, b, rounding_mode='floor') for floats (c10::div_floor_floating).
__device__ __forceinline__ float div_floor(float a, float b) {
    if (b == 0.f) return a / b;
    const float mod = fmodf(a, b);
    float div = (a - mod) / b;
    if ((mod != 0.f) && ((b < 0.f) != (mod < 0.f))) div -= 1.f;
    float fl;
    if (div != 0.f) {
        fl = floorf(div);
        if (div - fl > 0.5f) fl += 1.f;
    } else {
        fl = copysignf(0.f, a / b);
    }
    return fl;
}

// anchors [N,3]; min_v/max_v [3]; outputs anchors_q [N,3], quantized [N,3]
__global__ void __launch_bounds__(EW_THREADS)
    quantize_anchor_kernel(int64_t n3, const float *__restrict__ anchors, const float *__restrict__ min_v,
                           const float *__restrict__ max_v, float q_anchor, float levels_minus_1,
                           float *__restrict__ anchors_q, float *__restrict__ quantized) {
    const int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x;
    if (i >= n3) return;
    const int c = (int)(i % 3);
    const float lo = min_v[c];
    const float interval = (max_v[c] - lo) * q_anchor + 1e-6f;
    float q = div_floor(anchors[i] - lo, interval);
    q = fminf(fmaxf(q, 0.f), levels_minus_1);
    quantized[i] = q;
    anchors_q[i] = q * interval + lo;
}

extern "C" int cgs_quantize_anchor(const float *anchors, const float *min_v, const float *max_v, int64_t N,
                                   int round_digits, float *anchors_q, float *quantized, void *stream) {
    if (N < 0 || round_digits < 1 || round_digits > 24) { cgs_set_error("quantize_anchor: bad args"); return CGS_ERR_ARG; }
    if (N == 0) return CGS_OK;
    if (!anchors || !min_v || !max_v || !anchors_q || !quantized) { cgs_set_error("quantize_anchor: NULL"); return CGS_ERR_ARG; }
    const double levels = (double)((1u << round_digits) - 1u);
    const int64_t n3 = 3 * N;
    hipLaunchKernelGGL(quantize_anchor_kernel, dim3((unsigned)((n3 + EW_THREADS - 1) / EW_THREADS)),
                       dim3(EW_THREADS), 0, (hipStream_t)stream, n3, anchors, min_v, max_v,
                       (float)(1.0 / levels), (float)levels, anchors_q, quantized);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

// x [rows*cols]; Q index = i / q_div (q_div = cols for one Q per row, 1 for elementwise Q)
__global__ void __launch_bounds__(EW_THREADS)
    ste_multistep_kernel(int64_t n, const float *__restrict__ x, const float *__restrict__ Q, int64_t q_div,
                         int use_clamp, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x;
    if (i >= n) return;
    const float q = Q[i / q_div];
    float v = x[i];
    if (use_clamp) v = fminf(fmaxf(v, -15000.f * q), 15000.f * q);
    out[i] = rintf(v / q) * q;
}

extern "C" int cgs_ste_multistep(const float *x, const float *Q, int64_t n, int64_t q_div, int use_clamp,
                                 float *out, void *stream) {
    if (n < 0 || q_div < 1) { cgs_set_error("ste_multistep: bad args"); return CGS_ERR_ARG; }
    if (n == 0) return CGS_OK;
    if (!x || !Q || !out) { cgs_set_error("ste_multistep: NULL"); return CGS_ERR_ARG; }
    hipLaunchKernelGGL(ste_multistep_kernel, dim3((unsigned)((n + EW_THREADS - 1) / EW_THREADS)), dim3(EW_THREADS),
                       0, (hipStream_t)stream, n, x, Q, q_div, use_clamp, out);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

#include "rate_math.h"

__global__ void __launch_bounds__(EW_THREADS)
    entropy_gaussian_fwd_kernel(int64_t n, const float *__restrict__ x, const float *__restrict__ mean,
                                const float *__restrict__ scale, const float *__restrict__ Q, int64_t q_div,
                                const float *__restrict__ x_mean, int use_clamp, float *__restrict__ bits) {
    const int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x;
    if (i >= n) return;
    const RateTerms t = rate_terms(x[i], mean[i], scale[i], Q[i / q_div], use_clamp ? x_mean[0] : 0.f, use_clamp);
    bits[i] = rate_bits(t);
}

__global__ void __launch_bounds__(EW_THREADS)
    entropy_gaussian_bwd_kernel(int64_t n, const float *__restrict__ x, const float *__restrict__ mean,
                                const float *__restrict__ scale, const float *__restrict__ Q, int64_t q_div,
                                const float *__restrict__ x_mean, int use_clamp, const float *__restrict__ g_bits,
                                float *__restrict__ g_x, float *__restrict__ g_mean, float *__restrict__ g_scale,
                                float *__restrict__ g_Q) {
    const int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x;
    if (i >= n) return;
    const float sc = scale[i];
    const RateTerms t = rate_terms(x[i], mean[i], sc, Q[i / q_div], use_clamp ? x_mean[0] : 0.f, use_clamp);
    const RateGrads g = rate_grads(t, sc, g_bits[i]);
    g_x[i] = g.gx;
    g_mean[i] = g.gm;
    g_scale[i] = g.gs;
    g_Q[i] = g.gq;
}

extern "C" int cgs_entropy_gaussian_fwd(const float *x, const float *mean, const float *scale, const float *Q,
                                        int64_t n, int64_t q_div, const float *x_mean, int use_clamp, float *bits,
                                        void *stream) {
    if (n < 0 || q_div < 1) { cgs_set_error("entropy_gaussian_fwd: bad args"); return CGS_ERR_ARG; }
    if (n == 0) return CGS_OK;
    if (!x || !mean || !scale || !Q || !bits || (use_clamp && !x_mean)) {
        cgs_set_error("entropy_gaussian_fwd: NULL");
        return CGS_ERR_ARG;
    }
    CgsProfScope prof(CGS_PROF_RATE_FWD, (hipStream_t)stream);
    hipLaunchKernelGGL(entropy_gaussian_fwd_kernel, dim3((unsigned)((n + EW_THREADS - 1) / EW_THREADS)),
                       dim3(EW_THREADS), 0, (hipStream_t)stream, n, x, mean, scale, Q, q_div, x_mean, use_clamp, bits);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

extern "C" int cgs_entropy_gaussian_bwd(const float *x, const float *mean, const float *scale, const float *Q,
                                        int64_t n, int64_t q_div, const float *x_mean, int use_clamp,
                                        const float *g_bits, float *g_x, float *g_mean, float *g_scale, float *g_Q,
                                        void *stream) {
    if (n < 0 || q_div < 1) { cgs_set_error("entropy_gaussian_bwd: bad args"); return CGS_ERR_ARG; }
    if (n == 0) return CGS_OK;
    if (!x || !mean || !scale || !Q || !g_bits || !g_x || !g_mean || !g_scale || !g_Q || (use_clamp && !x_mean)) {
        cgs_set_error("entropy_gaussian_bwd: NULL");
        return CGS_ERR_ARG;
    }
    CgsProfScope prof(CGS_PROF_RATE_BWD, (hipStream_t)stream);
    hipLaunchKernelGGL(entropy_gaussian_bwd_kernel, dim3((unsigned)((n + EW_THREADS - 1) / EW_THREADS)),
                       dim3(EW_THREADS), 0, (hipStream_t)stream, n, x, mean, scale, Q, q_div, x_mean, use_clamp,
                       g_bits, g_x, g_mean, g_scale, g_Q);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}
