// Discretised-Gaussian rate terms shared by the element-wise rate kernels (elementwise.hip) and the fused
// per-level rate kernel (ctx.hip).  Reference: utils/entropy_models.py:30-50 (Entropy_gaussian),
// :8-27 (the clamped variant).
#pragma once
#include <hip/hip_runtime.h>

#define INV_SQRT2 0.70710678118654752440f
#define SQRT2 1.4142135623730951f
#define INV_SQRT_PI 0.56418958354775628695f
#define LIK_BOUND 1e-6f

struct RateTerms { float xc, s, inv, zu, zl, diff; bool in_range; };

__device__ __forceinline__ RateTerms rate_terms(float x, float mean, float scale, float q, float x_mean,
                                                int use_clamp) {
    RateTerms t;
    t.in_range = true;
    t.xc = x;
    if (use_clamp) {
        const float lo = x_mean - 15000.f * q, hi = x_mean + 15000.f * q;
        t.in_range = (x >= lo) && (x <= hi);
        t.xc = fminf(fmaxf(x, lo), hi);
    }
    t.s = fmaxf(scale, 1e-9f);
    t.inv = 1.f / t.s;
    // Normal(mean, s).cdf(v) = 0.5 * (1 + erf((v - mean) * (1/s) / sqrt(2)))
    t.zu = ((t.xc + 0.5f * q) - mean) * t.inv / SQRT2;
    t.zl = ((t.xc - 0.5f * q) - mean) * t.inv / SQRT2;
    const float upper = 0.5f * (1.f + erff(t.zu));
    const float lower = 0.5f * (1.f + erff(t.zl));
    t.diff = upper - lower;
    return t;
}


__device__ __forceinline__ float rate_bits(const RateTerms &t) { return -log2f(fmaxf(fabsf(t.diff), LIK_BOUND)); }

struct RateGrads { float gx, gm, gs, gq; };

// gradients of bits = -log2(max(|diff|, bound)) * g_bits w.r.t. (x, mean, scale, q); `sc` is the raw scale input
__device__ __forceinline__ RateGrads rate_grads(const RateTerms &t, float sc, float g_bits) {
    RateGrads g = {0.f, 0.f, 0.f, 0.f};
    const float lik = fabsf(t.diff);
    // Low_bound.backward zeroes the gradient wherever the raw likelihood is below the bound
    if (lik >= LIK_BOUND) {
        const float g_lik = g_bits * (-1.4426950408889634f / lik);
        const float sgn = t.diff > 0.f ? 1.f : (t.diff < 0.f ? -1.f : 0.f);
        const float g_diff = g_lik * sgn;
        const float g_zu = g_diff * INV_SQRT_PI * __expf(-t.zu * t.zu);
        const float g_zl = -g_diff * INV_SQRT_PI * __expf(-t.zl * t.zl);
        const float k = t.inv * INV_SQRT2;
        const float g_xc = (g_zu + g_zl) * k;
        g.gx = t.in_range ? g_xc : 0.f;
        g.gm = -g_xc;
        g.gq = 0.5f * (g_zu - g_zl) * k;
        g.gs = (sc >= 1e-9f) ? -(g_zu * t.zu + g_zl * t.zl) * t.inv : 0.f;
    }
    return g;
}
