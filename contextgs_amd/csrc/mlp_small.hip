// Backward of a fused 2-layer MLP with a tiny output (OUT <= 4: the three step-size outputs of mlp_grid that run
// on EVERY anchor, scene/gaussian_model.py:1603-1608) WITHOUT a saved hidden layer.
//
// For these shapes the hidden activations H [n,100] are ~60 % of the bytes the MLP moves: written by the forward,
// read by the backward, read again by the second layer's weight gradient.  Here the forward stores nothing; the
// backward recomputes H = relu(W1 x + b1) from X on the matrix cores (the same MFMA chain as the forward, so the
// same bits), applies the OUT x HID second layer on the VALU (3 FMAs per hidden unit), and accumulates the second
// layer's weight gradient dW2[o][h] += dY[row,o] * H[row,h] in registers — lane (g,c) already holds H of row c for
// its 28 hidden units, so 84 per-lane accumulators summed over the 16 row lanes at the very end replace a whole
// weight-gradient launch.  Only dZ1 [n,HID] is written (the first layer's weight gradient contracts it with X).
#include "cgs_internal.h"
#include "mlp_frag.h"

template <int IN, int HID, int OUT, int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
    mlp2_bwd_rc_kernel(const float *__restrict__ X, int64_t ldx, const float *__restrict__ W1,
                       const float *__restrict__ b1, const float *__restrict__ W2, const float *__restrict__ dY,
                       int64_t ldy, float *__restrict__ dX, int64_t lddx, int accumulate_dx,
                       float *__restrict__ dZ1, int64_t n, float *__restrict__ partial) {
    constexpr int NTI = (IN + 15) / 16, NT1 = (HID + 15) / 16;
    constexpr int XP = NTI * 16, HP = NT1 * 16;
    constexpr int S1 = frag_pad4mod8(HP), SB = frag_pad4mod8(XP);
    constexpr int E = OUT * HID + OUT;
    __shared__ float W1s[XP * S1];     // [k][j]  forward layout (layer-1 recompute)
    __shared__ float W1n[HP * SB];     // [h][k]  backward layout (dX)
    __shared__ float b1s[HP];
    __shared__ float W2s[OUT * HP];
    __shared__ float red[OUT * HP + OUT];
    const int tid = threadIdx.x, nthr = WAVES * 64;
    frag_stage_transposed<IN, HID, XP, S1>(W1s, W1, tid, nthr);       // W1s[k][j] = W1[j][k]
    frag_stage_loop(W1, HP * SB, tid, nthr, [](int i) { const int h = i / SB, k = i % SB; return (h < HID && k < IN) ? h * IN + k : -1; },
                    [&](int i, float v) { W1n[i] = v; });
    for (int i = tid; i < HP; i += nthr) b1s[i] = i < HID ? b1[i] : 0.f;
    for (int i = tid; i < OUT * HP; i += nthr) W2s[i] = (i % HP) < HID ? W2[(i / HP) * HID + (i % HP)] : 0.f;
    for (int i = tid; i < OUT * HP + OUT; i += nthr) red[i] = 0.f;
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    float accW2[OUT][NT1][4], accb[OUT];
#pragma unroll
    for (int o = 0; o < OUT; ++o) {
        accb[o] = 0.f;
#pragma unroll
        for (int t = 0; t < NT1; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) accW2[o][t][r] = 0.f;
    }
    const int64_t ntiles = (n + 15) / 16;
    for (int64_t tile = (int64_t)blockIdx.x * WAVES + wave; tile < ntiles; tile += (int64_t)gridDim.x * WAVES) {
        asm volatile("" ::: "memory");   // keep the LDS weight reads inside the tile loop (LICM would spill them)
        const int64_t row = tile * 16 + c;
        const bool valid = row < n;
        f32x4 xb[NTI];
#pragma unroll
        for (int q = 0; q < NTI; ++q) xb[q] = frag_load4<IN>(X + row * ldx, q, g, valid);
        float dy[OUT];
#pragma unroll
        for (int o = 0; o < OUT; ++o) dy[o] = valid ? dY[row * ldy + o] : 0.f;
        // H^T = relu(W1 X^T + b1): the forward's MFMA chain
        f32x4 acc1[NT1];
#pragma unroll
        for (int t = 0; t < NT1; ++t) acc1[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < NTI; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (16 * q + j >= IN) continue;
#pragma unroll
                for (int t = 0; t < NT1; ++t)
                    acc1[t] = frag_mfma(W1s[(16 * q + 4 * g + j) * S1 + 16 * t + c], xb[q][j], acc1[t]);
            }
        // second layer on the VALU: dW2 partials, dZ1 = relu'(.) * W2^T dY
#pragma unroll
        for (int t = 0; t < NT1; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int hh = 16 * t + 4 * g + r;
                const float h = fmaxf(acc1[t][r] + b1s[hh], 0.f);
                float dz = 0.f;
#pragma unroll
                for (int o = 0; o < OUT; ++o) {
                    accW2[o][t][r] = fmaf(dy[o], h, accW2[o][t][r]);
                    dz = fmaf(dy[o], W2s[o * HP + hh], dz);
                }
                acc1[t][r] = h > 0.f ? dz : 0.f;
            }
            frag_store4<HID>(dZ1 + row * HID, t, g, valid, acc1[t]);
        }
        if (g == 0) {
#pragma unroll
            for (int o = 0; o < OUT; ++o) accb[o] += dy[o];
        }
        if (dX) {
            f32x4 adx[NTI];
#pragma unroll
            for (int v = 0; v < NTI; ++v) adx[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < NT1; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (16 * t + r >= HID) continue;
#pragma unroll
                    for (int v = 0; v < NTI; ++v)
                        adx[v] = frag_mfma(W1n[(16 * t + 4 * g + r) * SB + 16 * v + c], acc1[t][r], adx[v]);
                }
#pragma unroll
            for (int v = 0; v < NTI; ++v) {
                f32x4 d = adx[v];
                if (accumulate_dx) d += frag_load4<IN>(dX + row * lddx, v, g, valid);
                frag_store4<IN>(dX + row * lddx, v, g, valid, d);
            }
        }
    }
    // dW2 / db2: sum over the 16 row lanes, then over the waves — the waves take turns with plain LDS read-modify-writes
    // (a fixed order: the same bits every run, which LDS float atomics do not give) — one partial image per workgroup
#pragma unroll
    for (int o = 0; o < OUT; ++o) {
#pragma unroll
        for (int t = 0; t < NT1; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = accW2[o][t][r];
                v += __shfl_xor(v, 8);
                v += __shfl_xor(v, 4);
                v += __shfl_xor(v, 2);
                v += __shfl_xor(v, 1);
                accW2[o][t][r] = v;
            }
        float vb = accb[o];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) vb += __shfl_xor(vb, d);
        accb[o] = vb;
    }
    for (int w = 0; w < WAVES; ++w) {
        if (wave == w) {
#pragma unroll
            for (int o = 0; o < OUT; ++o) {
#pragma unroll
                for (int t = 0; t < NT1; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (c == 0) red[o * HP + 16 * t + 4 * g + r] += accW2[o][t][r];
                if (lane == 0) red[OUT * HP + o] += accb[o];
            }
        }
        __syncthreads();
    }
    float *dst = partial + (int64_t)blockIdx.x * E;
    for (int i = tid; i < E; i += nthr) {
        const int o = i / HID, h = i % HID;
        dst[i] = i < OUT * HID ? red[o * HP + h] : red[OUT * HP + (i - OUT * HID)];
    }
}

// one wave per output element: the lanes stride over the workgroup partials (a handful of loads deep instead of a
// serial walk over ~256 partials by two half-empty workgroups)
__global__ void __launch_bounds__(256)
    mlp_small_reduce_kernel(const float *__restrict__ partial, int blocks, int E, int DADB, float *__restrict__ dW,
                            float *__restrict__ db) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= E) return;
    float a = 0.f;
    for (int b = lane; b < blocks; b += 64) a += partial[(int64_t)b * E + e];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d);
    if (lane == 0) {
        if (e < DADB) dW[e] += a;
        else db[e - DADB] += a;
    }
}

template <int IN, int HID, int OUT>
static int launch_rc(const float *X, int64_t ldx, const float *W1, const float *b1, const float *W2, const float *dY,
                     int64_t ldy, float *dX, int64_t lddx, int acc, float *dZ1, float *dW2, float *db2, int64_t n,
                     int num_cus, void *scratch, size_t scratch_bytes, hipStream_t s) {
    constexpr int WAVES = 8, E = OUT * HID + OUT;
    const int64_t tiles = (n + 15) / 16;
    const int64_t want = (tiles + WAVES - 1) / WAVES;
    int64_t grid = want < num_cus ? want : num_cus;
    if ((size_t)grid * E * sizeof(float) > scratch_bytes) { cgs_set_error("mlp2_backward: scratch too small"); return CGS_ERR_WORKSPACE; }
    hipLaunchKernelGGL((mlp2_bwd_rc_kernel<IN, HID, OUT, WAVES>), dim3((unsigned)grid), dim3(WAVES * 64), 0, s, X, ldx, W1, b1,
                       W2, dY, ldy, dX, lddx, acc, dZ1, n, (float *)scratch);
    CGS_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(mlp_small_reduce_kernel, dim3((E + 3) / 4), dim3(256), 0, s, (const float *)scratch, (int)grid, E,
                       OUT * HID, dW2, db2);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

// returns -1 when there is no instance for the shape
int cgs_launch_mlp2_bwd_recompute(int in, int hid, int out, const float *X, int64_t ldx, const float *W1,
                                  const float *b1, const float *W2, const float *dY, int64_t ldy, float *dX,
                                  int64_t lddx, int acc, float *dZ1, float *dW2, float *db2, int64_t n, int num_cus,
                                  void *scratch, size_t scratch_bytes, hipStream_t s) {
    if (in == 71 && hid == 100 && out == 3)
        return launch_rc<71, 100, 3>(X, ldx, W1, b1, W2, dY, ldy, dX, lddx, acc, dZ1, dW2, db2, n, num_cus, scratch, scratch_bytes, s);
    if (in == 15 && hid == 100 && out == 3)
        return launch_rc<15, 100, 3>(X, ldx, W1, b1, W2, dY, ldy, dX, lddx, acc, dZ1, dW2, db2, n, num_cus, scratch, scratch_bytes, s);
    return -1;
}
