// Counter-based training noise and the step-size map of the context model, shared by the element-wise level kernels
// (ctx.hip) and the fused level kernels (ctx_level.hip).  Reference: scene/gaussian_model.py:1603-1616.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Counter-based noise: u(seed, tensor, element) in [-0.5, 0.5), regenerated (not stored) by the backward.
// 32-bit arithmetic on purpose (two v_mul_lo_u32 per value): the first version was splitmix64, whose three 64-bit
// multiplications (twelve quarter-rate 32-bit multiplies) made the noise kernels compute-bound at ~1.9 TB/s.
// key = lowbias32(seed_lo ^ golden * (tensor + 1)) ^ seed_hi;  u = lowbias32(elem_lo + key + elem_hi * c) >> 8.
// Restated in oracle/context_ref.py:ctx_noise (the fixtures feed the same values to the reference).
__device__ __forceinline__ uint32_t ctx_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t ctx_noise_key(uint64_t seed, uint32_t tensor) {
    return ctx_mix32((uint32_t)seed ^ (0x9E3779B9u * (tensor + 1u))) ^ (uint32_t)(seed >> 32);
}
__device__ __forceinline__ float ctx_noise_k(uint32_t key, uint64_t elem) {
    const uint32_t h = ctx_mix32((uint32_t)elem + key + (uint32_t)(elem >> 32) * 0x632BE5ABu);
    return (float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f;
}
__device__ __forceinline__ float ctx_noise(uint64_t seed, uint32_t tensor, uint64_t elem) {
    return ctx_noise_k(ctx_noise_key(seed, tensor), elem);
}

__device__ __forceinline__ float ctx_step(float q0, float qadj) { return fmaxf(q0 * (1.f + tanhf(qadj)), 1e-9f); }

// accumulator of the three source sums (the rate model's clamp centres): double [CTX_SUM_SLOTS][CTX_SUM_STRIDE]
#define CTX_SUM_SLOTS 64
#define CTX_SUM_STRIDE 16        // doubles: one 128-byte line per slot

__device__ __forceinline__ double ctx_wave_sum(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

