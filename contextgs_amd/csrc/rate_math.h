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

// The three divisions of the reference's formula (`scale.reciprocal()`, `/ math.sqrt(2)`, `log2e / likelihood`) as a hardware
// reciprocal / a product + ONE Newton step each — the correctly rounded quotient in all but a vanishing fraction of inputs (then
// 1 ulp off), 3-4 instructions instead of the 16-17 of an IEEE division sequence.  Round 6: with the rate terms inside the
// MFMA kernels of the rate subset (rate_sub.hip) the element maths IS the kernel's VALU time.
__device__ __forceinline__ float rate_rcp(float s) {
    const float r = __builtin_amdgcn_rcpf(s);
    return fmaf(fmaf(-s, r, 1.f), r, r);
}
// erf with the coefficients of the ROCm device library's erff (two polynomials, |x| < 1 and 1 - exp(-(|x| + |x| P(|x|))) beyond),
// branch-free and with the exponential as ONE v_exp_f32 of the product with -log2(e) instead of the library's extended-precision
// range reduction: identical values for |x| < 1, within 2e-8 absolute beyond (erfc(1) = 0.157 times the 1e-7 relative error of
// the exponential) — below the 6e-8 spacing of fp32 numbers near 1 that every likelihood here is a difference of.  19
// instructions against 45 with a divergent branch; 48 evaluations per lane and 16-row tile in rate_sub.hip.
__device__ __forceinline__ float rate_erf(float x) {
    const float a = fabsf(x);
    float p = fmaf(a, 1.699881e-05f, -0.00037867785f);
    p = fmaf(a, p, 0.0038578159f);
    p = fmaf(a, p, -0.024181698f);
    p = fmaf(a, p, 0.10666826f);
    p = fmaf(a, p, 0.6349333f);
    p = fmaf(a, p, 0.12868941f);
    p = fmaf(a, p, a);
    const float big = 1.f - __builtin_amdgcn_exp2f(p * -1.4426950408889634f);
    const float t = a * a;
    float q = fmaf(-0.0005618018f, t, 0.004913816f);
    q = fmaf(t, q, -0.026707515f);
    q = fmaf(t, q, 0.11280011f);
    q = fmaf(t, q, -0.37612295f);
    q = fmaf(t, q, 0.1283791f);
    const float small = fmaf(a, q, a);
    return copysignf(a < 1.f ? small : big, x);
}
__device__ __forceinline__ float rate_div_sqrt2(float x) {
    const float q = x * INV_SQRT2;
    const float c = fmaf(fmaf(-q, SQRT2, x), INV_SQRT2, q);
    return fabsf(q) < 1e30f ? c : q;           // (inf stays inf)
}

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
    t.inv = rate_rcp(t.s);
    // Normal(mean, s).cdf(v) = 0.5 * (1 + erf((v - mean) * (1/s) / sqrt(2)))
    t.zu = rate_div_sqrt2(((t.xc + 0.5f * q) - mean) * t.inv);
    t.zl = rate_div_sqrt2(((t.xc - 0.5f * q) - mean) * t.inv);
    const float upper = 0.5f * (1.f + rate_erf(t.zu));
    const float lower = 0.5f * (1.f + rate_erf(t.zl));
    t.diff = upper - lower;
    return t;
}


// (the argument is >= 1e-6, a normal number: v_log_f32 without log2f's denormal pre-scaling returns the same value)
__device__ __forceinline__ float rate_bits(const RateTerms &t) { return -__log2f(fmaxf(fabsf(t.diff), LIK_BOUND)); }

struct RateGrads { float gx, gm, gs, gq; };

// gradients of bits = -log2(max(|diff|, bound)) * g_bits w.r.t. (x, mean, scale, q); `sc` is the raw scale input
__device__ __forceinline__ RateGrads rate_grads(const RateTerms &t, float sc, float g_bits) {
    RateGrads g = {0.f, 0.f, 0.f, 0.f};
    const float lik = fabsf(t.diff);
    // Low_bound.backward zeroes the gradient wherever the raw likelihood is below the bound
    if (lik >= LIK_BOUND) {
        const float g_lik = g_bits * (-1.4426950408889634f * rate_rcp(lik));
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
