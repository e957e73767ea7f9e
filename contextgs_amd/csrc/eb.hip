// Factorised-prior likelihood of the hyper latents (EntropyBottleneck.forward,
// scene/gaussian_model.py:1556; density maths = utils/entropy_models.py:103-138) as ONE
// fused kernel forward and one backward, instead of ~60 batched [C,3,3]x[C,3,N] matmul /
// softplus / tanh / sigmoid launches per call (rocprof: ~10 ms per training step at 1 M
// anchors).  Per element and per bound the density is a 1 -> 3 -> 3 -> 3 -> 3 -> 1 network
// with tanh gates: ~120 FMA + 24 tanh, register resident.
//
// Mapping: a wave owns ONE channel (its 58 effective parameters are wave-uniform) and walks
// rows 64 at a time; the backward reduces the 58 parameter gradients over the wave on the
// DPP network and keeps them in registers across the whole row loop, so the kernel ends with
// 58 atomics per wave.
#include "cgs_internal.h"

#define EB_P 58          // packed parameters per channel
// layout: [M0 3 | B0 3 | F0 3 | M1 9 | B1 3 | F1 3 | M2 9 | B2 3 | F2 3 | M3 9 | B3 3 | F3 3 | M4 3 | B4 1]
#define OFF_M0 0
#define OFF_B0 3
#define OFF_F0 6
#define OFF_M(k) (9 + 15 * ((k)-1))       // k = 1..3
#define OFF_B(k) (OFF_M(k) + 9)
#define OFF_F(k) (OFF_M(k) + 12)
#define OFF_M4 54
#define OFF_B4 57
#define EB_BOUND 1e-9f

__device__ __forceinline__ float softplusf(float x) { return x > 20.f ? x : log1pf(__expf(x)); }
__device__ __forceinline__ float eb_rcp(float d) {          // 1 / d: hardware reciprocal + one Newton step (the IEEE sequence is 16 instructions)
    const float r = __builtin_amdgcn_rcpf(d);
    return fmaf(fmaf(-d, r, 1.f), r, r);
}
__device__ __forceinline__ float sigmoidf(float x) { return eb_rcp(1.f + __expf(-x)); }
// tanh of the density's hidden units (24 per element and evaluation pair: the library's tanhf, a 38-instruction divergent
// routine, WAS this file's kernels — round 6): 1 - 2 / (1 + exp(2 z)) with one v_exp_f32; absolute error ~1e-7, which is what
// enters y = z + f tanh(z) (the relative error near z = 0 does not: nothing divides by it); |z| clamped where tanh is +-1 in fp32
__device__ __forceinline__ float eb_tanh(float z) {
    const float zc = fminf(fmaxf(z, -20.f), 20.f);
    return 1.f - 2.f * eb_rcp(1.f + __builtin_amdgcn_exp2f(zc * 2.8853900817779268f));
}

struct EbTrace {           // intermediates of one evaluation
    float u, y0[3], t0[3], y[3][3], t[3][3];   // y[k-1], t[k-1] for k = 1..3
    float out;
};

// Effective parameters of channel c: softplus of the matrices, tanh of the factors, biases as they are.  The 58 values are
// the same for every thread of a workgroup (one channel per workgroup), so each is transformed ONCE, by thread i, and
// shared through LDS: as a per-thread loop this was ~2500 instructions of tanhf / softplusf per thread before the first
// element — a third of eb_bits_fwd's time at ~3 elements per lane.  Call from all threads of the workgroup.
__device__ __forceinline__ float eb_transform(int i, float v) {
    const bool is_f = (i >= OFF_F0 && i < OFF_F0 + 3) || (i >= OFF_F(1) && i < OFF_F(1) + 3) ||
                      (i >= OFF_F(2) && i < OFF_F(2) + 3) || (i >= OFF_F(3) && i < OFF_F(3) + 3);
    const bool is_b = (i >= OFF_B0 && i < OFF_B0 + 3) || (i >= OFF_B(1) && i < OFF_B(1) + 3) ||
                      (i >= OFF_B(2) && i < OFF_B(2) + 3) || (i >= OFF_B(3) && i < OFF_B(3) + 3) || i == OFF_B4;
    return is_b ? v : (is_f ? tanhf(v) : softplusf(v));
}

__device__ __forceinline__ void eb_load_params(const float *__restrict__ raw, int c, float p[EB_P]) {
    __shared__ float sp[EB_P];
    if (threadIdx.x < EB_P) sp[threadIdx.x] = eb_transform((int)threadIdx.x, raw[c * EB_P + threadIdx.x]);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < EB_P; ++i) p[i] = sp[i];
}

__device__ __forceinline__ float eb_eval(const float p[EB_P], float u, EbTrace &tr) {
    tr.u = u;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float z = p[OFF_M0 + i] * u + p[OFF_B0 + i];
        tr.t0[i] = eb_tanh(z);
        tr.y0[i] = z + p[OFF_F0 + i] * tr.t0[i];
    }
    float prev[3] = {tr.y0[0], tr.y0[1], tr.y0[2]};
#pragma unroll
    for (int k = 1; k <= 3; ++k) {
        float cur[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float z = p[OFF_M(k) + 3 * i] * prev[0] + p[OFF_M(k) + 3 * i + 1] * prev[1] +
                            p[OFF_M(k) + 3 * i + 2] * prev[2] + p[OFF_B(k) + i];
            tr.t[k - 1][i] = eb_tanh(z);
            cur[i] = z + p[OFF_F(k) + i] * tr.t[k - 1][i];
            tr.y[k - 1][i] = cur[i];
        }
        prev[0] = cur[0]; prev[1] = cur[1]; prev[2] = cur[2];
    }
    tr.out = p[OFF_M4] * prev[0] + p[OFF_M4 + 1] * prev[1] + p[OFF_M4 + 2] * prev[2] + p[OFF_B4];
    return tr.out;
}

// accumulates d(effective params) into gp, returns dL/du
__device__ __forceinline__ float eb_eval_bwd(const float p[EB_P], const EbTrace &tr, float g, float gp[EB_P]) {
    float dy[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        gp[OFF_M4 + j] += g * tr.y[2][j];
        dy[j] = g * p[OFF_M4 + j];
    }
    gp[OFF_B4] += g;
#pragma unroll
    for (int k = 3; k >= 1; --k) {
        float dprev[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float th = tr.t[k - 1][i];
            const float dz = dy[i] * (1.f + p[OFF_F(k) + i] * (1.f - th * th));
            gp[OFF_F(k) + i] += dy[i] * th;
            gp[OFF_B(k) + i] += dz;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float yin = (k == 1) ? tr.y0[j] : tr.y[k - 2][j];
                gp[OFF_M(k) + 3 * i + j] += dz * yin;
                dprev[j] += p[OFF_M(k) + 3 * i + j] * dz;
            }
        }
        dy[0] = dprev[0]; dy[1] = dprev[1]; dy[2] = dprev[2];
    }
    float du = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float th = tr.t0[i];
        const float dz = dy[i] * (1.f + p[OFF_F0 + i] * (1.f - th * th));
        gp[OFF_F0 + i] += dy[i] * th;
        gp[OFF_B0 + i] += dz;
        gp[OFF_M0 + i] += dz * tr.u;
        du += p[OFF_M0 + i] * dz;
    }
    return du;
}

__global__ void __launch_bounds__(256)
    eb_likelihood_fwd_kernel(const float *__restrict__ v, const float *__restrict__ raw, int64_t n, int C,
                             float *__restrict__ lik) {
    // the 4 waves of a workgroup share ONE channel (blockIdx % C): the backward can then combine their parameter
    // gradients in LDS and issue one set of 58 atomics per workgroup
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = (int)(blockIdx.x % C);
    const int64_t w_in_c = (int64_t)(blockIdx.x / C) * 4 + wave, waves_per_c = (int64_t)(gridDim.x / C) * 4;
    float p[EB_P];
    eb_load_params(raw, c, p);
    for (int64_t row = w_in_c * 64 + lane; row < n; row += waves_per_c * 64) {
        const float x = v[row * C + c];
        EbTrace tl, tu;
        const float lo = eb_eval(p, x - 0.5f, tl), up = eb_eval(p, x + 0.5f, tu);
        const float sm = lo + up;
        const float s = sm > 0.f ? -1.f : (sm < 0.f ? 1.f : 0.f);
        lik[row * C + c] = fmaxf(fabsf(sigmoidf(s * up) - sigmoidf(s * lo)), EB_BOUND);
    }
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float eb_dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
    return v + __int_as_float(moved);
}
__device__ __forceinline__ float eb_wave_sum63(float v) {
    v = eb_dpp_add<0xB1, 0xF>(v);
    v = eb_dpp_add<0x4E, 0xF>(v);
    v = eb_dpp_add<0x141, 0xF>(v);
    v = eb_dpp_add<0x140, 0xF>(v);
    v = eb_dpp_add<0x142, 0xA>(v);
    v = eb_dpp_add<0x143, 0xC>(v);
    return v;
}

__global__ void __launch_bounds__(256)
    eb_likelihood_bwd_kernel(const float *__restrict__ v, const float *__restrict__ raw, const float *__restrict__ g_lik,
                             int64_t n, int C, float *__restrict__ g_v, float *__restrict__ g_raw) {
    // the 4 waves of a workgroup share ONE channel (blockIdx % C): the backward can then combine their parameter
    // gradients in LDS and issue one set of 58 atomics per workgroup
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = (int)(blockIdx.x % C);
    const int64_t w_in_c = (int64_t)(blockIdx.x / C) * 4 + wave, waves_per_c = (int64_t)(gridDim.x / C) * 4;
    float p[EB_P], gp[EB_P];
    eb_load_params(raw, c, p);
#pragma unroll
    for (int i = 0; i < EB_P; ++i) gp[i] = 0.f;
    for (int64_t row = w_in_c * 64 + lane; row < n; row += waves_per_c * 64) {
        const float x = v[row * C + c];
        EbTrace tl, tu;
        const float lo = eb_eval(p, x - 0.5f, tl), up = eb_eval(p, x + 0.5f, tu);
        const float sm = lo + up;
        const float s = sm > 0.f ? -1.f : (sm < 0.f ? 1.f : 0.f);
        const float su = sigmoidf(s * up), sl = sigmoidf(s * lo);
        const float d = su - sl;
        const float g = g_lik[row * C + c];
        // LowerBound: pass where the raw likelihood is above the bound or the gradient pushes it up
        const bool pass = (fabsf(d) >= EB_BOUND) || (g < 0.f);
        const float gd = pass ? g * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : 0.f;
        const float g_up = gd * su * (1.f - su) * s;
        const float g_lo = -gd * sl * (1.f - sl) * s;
        float dx = eb_eval_bwd(p, tu, g_up, gp);
        dx += eb_eval_bwd(p, tl, g_lo, gp);
        g_v[row * C + c] = dx;
    }
    // wave reduction of the parameter gradients, workgroup reduction in LDS, then the chain rule through
    // softplus / tanh of the raw parameters and one atomic per parameter and workgroup
    __shared__ float part[4][EB_P];
#pragma unroll
    for (int i = 0; i < EB_P; ++i) {
        const float tot = eb_wave_sum63(gp[i]);
        if (lane == 63) part[wave][i] = tot;
    }
    __syncthreads();
    const int i = threadIdx.x;
    if (i < EB_P) {
        const float tot = (part[0][i] + part[1][i]) + (part[2][i] + part[3][i]);
        const float rv = raw[c * EB_P + i];
        const bool is_f = (i >= OFF_F0 && i < OFF_F0 + 3) || (i >= OFF_F(1) && i < OFF_F(1) + 3) ||
                          (i >= OFF_F(2) && i < OFF_F(2) + 3) || (i >= OFF_F(3) && i < OFF_F(3) + 3);
        const bool is_b = (i >= OFF_B0 && i < OFF_B0 + 3) || (i >= OFF_B(1) && i < OFF_B(1) + 3) ||
                          (i >= OFF_B(2) && i < OFF_B(2) + 3) || (i >= OFF_B(3) && i < OFF_B(3) + 3) || i == OFF_B4;
        const float pv = is_b ? rv : (is_f ? tanhf(rv) : softplusf(rv));
        const float chain = is_b ? 1.f : (is_f ? (1.f - pv * pv) : sigmoidf(rv));
        atomicAdd(&g_raw[c * EB_P + i], tot * chain);
    }
}

static int eb_grid(int64_t n, int C, int blocks_per_cu) {
    // workgroups = multiple of C (4 waves of one channel each), no more than one wave per 64 rows and channel;
    // the backward (2 waves / SIMD by its registers) takes two per CU, which also keeps its per-workgroup
    // parameter-gradient atomics few
    int64_t per_c = (n + 255) / 256;
    int64_t cap = (256 * blocks_per_cu) / C;
    if (cap < 1) cap = 1;
    if (per_c > cap) per_c = cap;
    if (per_c < 1) per_c = 1;
    return (int)(per_c * C);
}

// v, lik, g_lik, g_v: [n, C] row-major; raw, g_raw: [C, 58] packed raw parameters (see layout above)
extern "C" int cgs_eb_likelihood_fwd(const float *v, const float *raw, int64_t n, int C, float *lik, void *stream) {
    if (n < 0 || C < 1) { cgs_set_error("eb_likelihood_fwd: bad args"); return CGS_ERR_ARG; }
    if (n == 0) return CGS_OK;
    if (!v || !raw || !lik) { cgs_set_error("eb_likelihood_fwd: NULL"); return CGS_ERR_ARG; }
    const int grid = eb_grid(n, C, 8);
    hipLaunchKernelGGL(eb_likelihood_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, v, raw, n, C, lik);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

extern "C" int cgs_eb_likelihood_bwd(const float *v, const float *raw, const float *g_lik, int64_t n, int C, float *g_v,
                                     float *g_raw, void *stream) {
    if (n < 0 || C < 1) { cgs_set_error("eb_likelihood_bwd: bad args"); return CGS_ERR_ARG; }
    if (n == 0) return CGS_OK;
    if (!v || !raw || !g_lik || !g_v || !g_raw) { cgs_set_error("eb_likelihood_bwd: NULL"); return CGS_ERR_ARG; }
    const int grid = eb_grid(n, C, 2);
    hipLaunchKernelGGL(eb_likelihood_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, v, raw, g_lik, n, C, g_v,
                       g_raw);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

// ---- training-step forms -----------------------------------------------------------------------------------------------
// The training step consumes the hyper prior in exactly two ways (scene/gaussian_model.py:1556, 1662, 1689):
// the noisy latents as an input of every level MLP, and the SUM of -log2(likelihood) over the rate subset.  These
// entry points produce just that: the noisy latents directly in coding order, and the bit sum of a row subset read
// through an index (no [n_sub, C] likelihood tensor, no log2 / neg / sum launches, no autograd chain through them).

// out[r, c] = hyper[a, c] + u(seed, tensor 3, a * C + c), a = perm[r] (perm NULL: a = r).  The noise is keyed by the
// ORIGINAL anchor index, i.e. it is EntropyBottleneck's x + U(-1/2, 1/2) on the [N, C] latents, then permuted.
__global__ void __launch_bounds__(256)
    hyper_noise_gather_kernel(const float *__restrict__ hyper, const int64_t *__restrict__ perm, int64_t n, int C,
                              uint32_t key, float *__restrict__ out) {
    const int64_t total = n * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / C;
        const int c = (int)(i - r * C);
        const uint64_t e = (uint64_t)(perm ? perm[r] : r) * (uint64_t)C + (uint64_t)c;
        uint32_t h = (uint32_t)e + key + (uint32_t)(e >> 32) * 0x632BE5ABu;
        h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
        out[i] = hyper[e] + ((float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f);
    }
}

// the same with 16-byte accesses (C a multiple of 4, 16-byte aligned bases, n * C / 4 < 2^32)
__global__ void __launch_bounds__(256)
    hyper_noise_gather4_kernel(const float *__restrict__ hyper, const int64_t *__restrict__ perm, int64_t n, int C,
                               uint32_t key, float *__restrict__ out) {
    const uint32_t C4 = (uint32_t)C >> 2, total = (uint32_t)(n * C4);
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const uint32_t r = i / C4, c4 = i - r * C4;
        const uint64_t e = (uint64_t)(perm ? perm[r] : (int64_t)r) * (uint64_t)C + 4 * c4;
        float4 v = *(const float4 *)(hyper + e);
        float u[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint64_t ek = e + k;
            uint32_t h = (uint32_t)ek + key + (uint32_t)(ek >> 32) * 0x632BE5ABu;
            h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
            u[k] = (float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f;
        }
        v.x += u[0]; v.y += u[1]; v.z += u[2]; v.w += u[3];
        ((float4 *)out)[i] = v;
    }
}

static uint32_t eb_noise_key(uint64_t seed, uint32_t tensor) {     // == ctx_noise_key of csrc/ctx.hip
    uint32_t x = (uint32_t)seed ^ (0x9E3779B9u * (tensor + 1u));
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x ^ (uint32_t)(seed >> 32);
}

extern "C" int cgs_hyper_noise_gather(const float *hyper, const int64_t *perm, int64_t n, int C, uint64_t seed, float *out,
                                      void *stream) {
    if (n < 0 || C < 1) { cgs_set_error("hyper_noise_gather: bad args"); return CGS_ERR_ARG; }
    if (n == 0) return CGS_OK;
    if (!hyper || !out) { cgs_set_error("hyper_noise_gather: NULL"); return CGS_ERR_ARG; }
    int64_t blocks = (n * C + 1023) / 1024;
    if (blocks > 4096) blocks = 4096;
    CgsProfScope prof(CGS_PROF_CTX_FWD, (hipStream_t)stream);
    if (!(C & 3) && !(((uintptr_t)hyper | (uintptr_t)out) & 15) && n * (C / 4) < (int64_t)0xFFFF0000)
        hipLaunchKernelGGL(hyper_noise_gather4_kernel, dim3((unsigned)((blocks + 3) / 4)), dim3(256), 0, (hipStream_t)stream, hyper,
                           perm, n, C, eb_noise_key(seed, 3), out);
    else
        hipLaunchKernelGGL(hyper_noise_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, hyper, perm, n, C,
                           eb_noise_key(seed, 3), out);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

#define EB_LN2 0.6931471805599453f

// sum over the rows `rows[0..n)` (NULL: rows 0..n) and all channels of -log2(max(likelihood, 1e-9)); per-block partial
// sums in double, the last block to finish adds them in block order (deterministic) and writes the float result
__global__ void __launch_bounds__(256)
    eb_bits_fwd_kernel(const float *__restrict__ v, const int64_t *__restrict__ rows, const float *__restrict__ raw,
                       int64_t n, int C, double *__restrict__ partial, unsigned int *__restrict__ counter,
                       float *__restrict__ out) {
    __shared__ double sh[4];
    __shared__ bool last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = (int)(blockIdx.x % C);
    const int64_t w_in_c = (int64_t)(blockIdx.x / C) * 4 + wave, waves_per_c = (int64_t)(gridDim.x / C) * 4;
    float p[EB_P];
    eb_load_params(raw, c, p);
    float acc = 0.f;
    // (the row gather of the NEXT element is in flight while this one is evaluated: ~3 waves per SIMD do not hide a dependent
    //  index -> value round trip per element by themselves)
    const int64_t step = waves_per_c * 64;
    int64_t row = w_in_c * 64 + lane;
    float xn = row < n ? v[(rows ? rows[row] : row) * C + c] : 0.f;
    for (; row < n; row += step) {
        const float x = xn;
        const int64_t rnext = row + step;
        if (rnext < n) xn = v[(rows ? rows[rnext] : rnext) * C + c];
        EbTrace tl, tu;
        const float lo = eb_eval(p, x - 0.5f, tl), up = eb_eval(p, x + 0.5f, tu);
        const float sm = lo + up;
        const float s = sm > 0.f ? -1.f : (sm < 0.f ? 1.f : 0.f);
        acc -= __log2f(fmaxf(fabsf(sigmoidf(s * up) - sigmoidf(s * lo)), EB_BOUND));       // (argument >= 1e-9: normal)
    }
    double d = (double)acc;
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) d += __shfl_xor(d, k, 64);
    if (lane == 0) sh[wave] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        cgs_publish(&partial[blockIdx.x], (sh[0] + sh[1]) + (sh[2] + sh[3]));
        last = cgs_ticket_last(counter);
    }
    __syncthreads();
    if (!last) return;
    double t = 0.0;
    for (int j = threadIdx.x; j < (int)gridDim.x; j += 256) t += cgs_published(&partial[j]);
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) t += __shfl_xor(t, k, 64);
    __syncthreads();
    if (lane == 0) sh[wave] = t;
    __syncthreads();
    if (threadIdx.x == 0) *out = (float)((sh[0] + sh[1]) + (sh[2] + sh[3]));
}

// backward of the above for an upstream gradient *g_sum (device scalar): g_v_sub [n, C] (row r <-> rows[r]) and g_raw += .
__global__ void __launch_bounds__(256)
    eb_bits_bwd_kernel(const float *__restrict__ v, const int64_t *__restrict__ rows, const float *__restrict__ raw,
                       const float *__restrict__ g_sum, int64_t n, int C, float *__restrict__ g_v, float *__restrict__ g_raw) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = (int)(blockIdx.x % C);
    const int64_t w_in_c = (int64_t)(blockIdx.x / C) * 4 + wave, waves_per_c = (int64_t)(gridDim.x / C) * 4;
    float p[EB_P], gp[EB_P];
    eb_load_params(raw, c, p);
    const float gs = *g_sum;
#pragma unroll
    for (int i = 0; i < EB_P; ++i) gp[i] = 0.f;
    const int64_t step = waves_per_c * 64;
    int64_t row = w_in_c * 64 + lane;
    float xn = row < n ? v[(rows ? rows[row] : row) * C + c] : 0.f;
    for (; row < n; row += step) {
        const float x = xn;
        const int64_t rnext = row + step;
        if (rnext < n) xn = v[(rows ? rows[rnext] : rnext) * C + c];      // (one wave per SIMD here: nothing else hides the gather)
        EbTrace tl, tu;
        const float lo = eb_eval(p, x - 0.5f, tl), up = eb_eval(p, x + 0.5f, tu);
        const float sm = lo + up;
        const float s = sm > 0.f ? -1.f : (sm < 0.f ? 1.f : 0.f);
        const float su = sigmoidf(s * up), sl = sigmoidf(s * lo);
        const float d = su - sl;
        const float lik = fmaxf(fabsf(d), EB_BOUND);
        const float g = -gs * eb_rcp(lik * EB_LN2);                 // d(-log2 lik)/d lik
        const bool pass = (fabsf(d) >= EB_BOUND) || (g < 0.f);      // LowerBound: as eb_likelihood_bwd_kernel
        const float gd = pass ? g * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : 0.f;
        const float g_up = gd * su * (1.f - su) * s;
        const float g_lo = -gd * sl * (1.f - sl) * s;
        float dx = eb_eval_bwd(p, tu, g_up, gp);
        dx += eb_eval_bwd(p, tl, g_lo, gp);
        g_v[row * C + c] = dx;
    }
    __shared__ float part[4][EB_P];
#pragma unroll
    for (int i = 0; i < EB_P; ++i) {
        const float tot = eb_wave_sum63(gp[i]);
        if (lane == 63) part[wave][i] = tot;
    }
    __syncthreads();
    const int i = threadIdx.x;
    if (i < EB_P) {
        const float tot = (part[0][i] + part[1][i]) + (part[2][i] + part[3][i]);
        const float rv = raw[c * EB_P + i];
        const bool is_f = (i >= OFF_F0 && i < OFF_F0 + 3) || (i >= OFF_F(1) && i < OFF_F(1) + 3) ||
                          (i >= OFF_F(2) && i < OFF_F(2) + 3) || (i >= OFF_F(3) && i < OFF_F(3) + 3);
        const bool is_b = (i >= OFF_B0 && i < OFF_B0 + 3) || (i >= OFF_B(1) && i < OFF_B(1) + 3) ||
                          (i >= OFF_B(2) && i < OFF_B(2) + 3) || (i >= OFF_B(3) && i < OFF_B(3) + 3) || i == OFF_B4;
        const float pv = is_b ? rv : (is_f ? tanhf(rv) : softplusf(rv));
        const float chain = is_b ? 1.f : (is_f ? (1.f - pv * pv) : sigmoidf(rv));
        atomicAdd(&g_raw[c * EB_P + i], tot * chain);
    }
}

#define EB_BITS_MAX_BLOCKS 2048
extern "C" size_t cgs_eb_bits_scratch_bytes(void) { return (size_t)EB_BITS_MAX_BLOCKS * sizeof(double) + CGS_TICKET_BYTES; }

// scratch: cgs_eb_bits_scratch_bytes() bytes whose last CGS_TICKET_BYTES (the arrival ticket, cgs_ticket_last) are ZERO before the
// first use (the kernel leaves them zero); out: float [1]
extern "C" int cgs_eb_bits_fwd(const float *v, const int64_t *rows, const float *raw, int64_t n, int C, void *scratch,
                               size_t scratch_bytes, float *out, void *stream) {
    if (n < 0 || C < 1 || !out || !scratch || scratch_bytes < cgs_eb_bits_scratch_bytes()) { cgs_set_error("eb_bits_fwd: bad args"); return CGS_ERR_ARG; }
    if (n == 0) { CGS_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(float), (hipStream_t)stream)); return CGS_OK; }
    if (!v || !raw) { cgs_set_error("eb_bits_fwd: NULL"); return CGS_ERR_ARG; }
#ifndef EB_BITS_FWD_BPC
#define EB_BITS_FWD_BPC 3        // workgroups per CU.  With the textbook arrival ticket (store, __threadfence, atomic: ~25 ns per workgroup,
#endif                           // serial) 8 per CU = 2 040 workgroups cost 90 us, 3: 50 us, 2: 41 us; fence-free (cgs_publish): 2: 31.0, 3: 30.3, 4 / 8: 34.9 us
    int grid = eb_grid(n, C, EB_BITS_FWD_BPC);
    if (grid > EB_BITS_MAX_BLOCKS) grid = EB_BITS_MAX_BLOCKS / C * C;
    hipLaunchKernelGGL(eb_bits_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, v, rows, raw, n, C, (double *)scratch,
                       (unsigned int *)((char *)scratch + (size_t)EB_BITS_MAX_BLOCKS * sizeof(double)), out);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

extern "C" int cgs_eb_bits_bwd(const float *v, const int64_t *rows, const float *raw, const float *g_sum, int64_t n, int C,
                               float *g_v_sub, float *g_raw, void *stream) {
    if (n < 0 || C < 1) { cgs_set_error("eb_bits_bwd: bad args"); return CGS_ERR_ARG; }
    if (n == 0) return CGS_OK;
    if (!v || !raw || !g_sum || !g_v_sub || !g_raw) { cgs_set_error("eb_bits_bwd: NULL"); return CGS_ERR_ARG; }
    hipLaunchKernelGGL(eb_bits_bwd_kernel, dim3(eb_grid(n, C, 2)), dim3(256), 0, (hipStream_t)stream, v, rows, raw, g_sum, n, C,
                       g_v_sub, g_raw);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}
