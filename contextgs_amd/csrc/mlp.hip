// Fused 2-layer MLPs (Linear -> ReLU -> Linear [-> tanh | sigmoid]) on the gfx950 fp32
// matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, 1024 FMA / instruction).
//
// Why: rocprofv3 of the first end-to-end bench (profiles/r01_rocprof_bench_1m_v0.txt)
// shows the anchor MLPs (gaussian_renderer/__init__.py:112-126, 54 -> 50 -> {10,30,70})
// and the context MLPs (scene/gaussian_model.py:177-188, {71,15} -> 100 -> 175) as THE
// dense-contraction bottleneck: rocBLAS picks poor tiles for these skinny shapes (K = 1 M
// rows for the weight gradients) and torch's bias-gradient reductions cost more than the
// rasterizer.  north_star: "MFMA only if rocprof shows them as a dense contraction
// bottleneck" — it does.
//
// Design: everything is computed TRANSPOSED (weights are the A operand, activations the B
// operand) so that one layer's accumulator registers are directly the next layer's B
// operand: D-layout lane l, reg r holds Z^T[16t + 4(l>>4) + r][row l&15], and a k-step may
// use ANY set of four contraction indices as long as A follows it, so k-step (t, r) uses
// {16t + 4g + r : g = 0..3} — for every contraction, including the ones fed from memory,
// which makes each activation access one 16-byte access per lane (mlp_frag.h).  No LDS
// transposes; LDS holds only the weights, in layouts whose fragment reads are bank-conflict
// free.
//
//   mlp2_fwd      X [n,IN] -> Y [n,OUT] (+ H = relu(.) saved for the backward)
//   mlp2_bwd      dY, Y, H -> dX, dZ1 (= dH masked), dZ2 (= dY * act')
//   weight/bias gradients: mlp_wgrad.hip
#include "cgs_internal.h"
#include "mlp_frag.h"

// wave-tile shape (rows per wave tile = 16 RT, waves per workgroup); overridable for tools/mlp_tiling.sh
#ifndef M2_RT
#define M2_RT 1
#define M2_WAVES 16
#endif

#define ACT_NONE FRAG_ACT_NONE
#define ACT_TANH FRAG_ACT_TANH
#define ACT_SIGMOID FRAG_ACT_SIGMOID

// ------------------------------------------------------------------------------------------
template <int IN, int HID, int OUT, int ACT, int RT, int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
    mlp2_fwd_kernel(const float *__restrict__ X, int64_t ldx, const float *__restrict__ W1,
                    const float *__restrict__ b1, const float *__restrict__ W2, const float *__restrict__ b2,
                    float *__restrict__ Y, int64_t ldy, float *__restrict__ Hsave, int64_t n) {
    constexpr int NTI = (IN + 15) / 16, NT1 = (HID + 15) / 16, NT2 = (OUT + 15) / 16;
    constexpr int XP = NTI * 16, HP = NT1 * 16, OP = NT2 * 16;
    constexpr int S1 = frag_pad4mod8(HP), S2 = frag_pad4mod8(OP);
    __shared__ float W1s[XP * S1];   // [k][j]
    __shared__ float W2s[HP * S2];   // [h][o]
    __shared__ float b1s[HP];
    __shared__ float b2s[OP];
    const int tid = threadIdx.x, nthr = WAVES * 64;
    frag_stage_transposed<IN, HID, XP, S1>(W1s, W1, tid, nthr);       // W1s[k][j] = W1[j][k]
    frag_stage_transposed<HID, OUT, HP, S2>(W2s, W2, tid, nthr);      // W2s[h][o] = W2[o][h]
    for (int i = tid; i < HP; i += nthr) b1s[i] = i < HID ? b1[i] : 0.f;
    for (int i = tid; i < OP; i += nthr) b2s[i] = i < OUT ? b2[i] : 0.f;
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int64_t ntiles = (n + 16 * RT - 1) / (16 * RT);
    // X of the next tile is fetched while this tile is in the matrix pipe (double-buffered in registers)
    f32x4 xb[RT][NTI], xn[RT][NTI];
    bool valid[RT], validn[RT];
    const int64_t tile0 = (int64_t)blockIdx.x * WAVES + wave, tstride = (int64_t)gridDim.x * WAVES;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int64_t row = tile0 * 16 * RT + rt * 16 + c;
        valid[rt] = tile0 < ntiles && row < n;
#pragma unroll
        for (int q = 0; q < NTI; ++q) xb[rt][q] = frag_load4<IN>(X + row * ldx, q, g, valid[rt]);
    }
    for (int64_t tile = tile0; tile < ntiles; tile += tstride) {
        const int64_t row0 = tile * 16 * RT;
        asm volatile("" ::: "memory");   // keep the LDS weight reads inside the tile loop (LICM would spill them)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int64_t row = (tile + tstride) * 16 * RT + rt * 16 + c;
            validn[rt] = row < n;
#pragma unroll
            for (int q = 0; q < NTI; ++q) xn[rt][q] = frag_load4<IN>(X + row * ldx, q, g, validn[rt]);
        }
        f32x4 acc1[NT1][RT];
#pragma unroll
        for (int t = 0; t < NT1; ++t)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc1[t][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < NTI; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (16 * q + j >= IN) continue;   // every contraction index of this step is padding
#pragma unroll
                for (int t = 0; t < NT1; ++t) {
                    const float a = W1s[(16 * q + 4 * g + j) * S1 + 16 * t + c];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc1[t][rt] = frag_mfma(a, xb[rt][q][j], acc1[t][rt]);
                }
            }
        // bias + ReLU; keep H for the backward
#pragma unroll
        for (int t = 0; t < NT1; ++t)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int64_t row = row0 + rt * 16 + c;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc1[t][rt][r] = fmaxf(acc1[t][rt][r] + b1s[16 * t + 4 * g + r], 0.f);
                if (Hsave) frag_store4<HID>(Hsave + row * HID, t, g, valid[rt], acc1[t][rt]);
            }
        f32x4 acc2[NT2][RT];
#pragma unroll
        for (int u = 0; u < NT2; ++u)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc2[u][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NT1; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (16 * t + r >= HID) continue;
#pragma unroll
                for (int u = 0; u < NT2; ++u) {
                    const float a = W2s[(16 * t + 4 * g + r) * S2 + 16 * u + c];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc2[u][rt] = frag_mfma(a, acc1[t][rt][r], acc2[u][rt]);
                }
            }
#pragma unroll
        for (int u = 0; u < NT2; ++u)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int64_t row = row0 + rt * 16 + c;
                f32x4 y;
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = frag_act<ACT>(acc2[u][rt][r] + b2s[16 * u + 4 * g + r]);
                frag_store4<OUT>(Y + row * ldy, u, g, valid[rt], y);
            }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            valid[rt] = validn[rt];
#pragma unroll
            for (int q = 0; q < NTI; ++q) xb[rt][q] = xn[rt][q];
        }
    }
}

// ------------------------------------------------------------------------------------------
template <int IN, int HID, int OUT, int ACT, int RT, int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
    mlp2_bwd_kernel(const float *__restrict__ dY, const float *__restrict__ Y, int64_t ldy,
                    const float *__restrict__ Hsave, const float *__restrict__ W1, const float *__restrict__ W2,
                    float *__restrict__ dX, int64_t lddx, int accumulate_dx, float *__restrict__ dZ2,
                    float *__restrict__ dZ1, int64_t n, const int64_t *__restrict__ dx_rows) {
    // dx_rows (may be NULL): row r of this call's input gradient goes to row dx_rows[r] of dX (distinct rows) — with
    // accumulate_dx this is `dX.index_add_(0, dx_rows, dx)` folded into the store
    constexpr int NT2 = (OUT + 15) / 16, NT1 = (HID + 15) / 16, NTX = (IN + 15) / 16;
    constexpr int OP = NT2 * 16, HP = NT1 * 16, XP = NTX * 16;
    constexpr int SA = frag_pad4mod8(HP), SB = frag_pad4mod8(XP);
    __shared__ float W2n[OP * SA];   // [o][h]
    __shared__ float W1n[HP * SB];   // [h][k]
    const int tid = threadIdx.x, nthr = WAVES * 64;
    frag_stage_loop(W2, OP * SA, tid, nthr, [](int i) { const int o = i / SA, h = i % SA; return (o < OUT && h < HID) ? o * HID + h : -1; },
                    [&](int i, float v) { W2n[i] = v; });
    frag_stage_loop(W1, HP * SB, tid, nthr, [](int i) { const int h = i / SB, k = i % SB; return (h < HID && k < IN) ? h * IN + k : -1; },
                    [&](int i, float v) { W1n[i] = v; });
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int64_t ntiles = (n + 16 * RT - 1) / (16 * RT);
    for (int64_t tile = (int64_t)blockIdx.x * WAVES + wave; tile < ntiles; tile += (int64_t)gridDim.x * WAVES) {
        const int64_t row0 = tile * 16 * RT;
        asm volatile("" ::: "memory");   // keep the LDS weight reads inside the tile loop (LICM would spill them)
        bool valid[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) valid[rt] = row0 + rt * 16 + c < n;
        // dH^T = W2^T dZ2^T
        f32x4 adh[NT1][RT];
#pragma unroll
        for (int t = 0; t < NT1; ++t)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) adh[t][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // Loads run ahead of their use (the scheduler does not move them across the exec-masked blocks by itself):
        // the saved activations, needed last, are issued first; the dY (and Y) fragments LA tiles ahead of the MFMAs.
        f32x4 hv[NT1][RT];
#pragma unroll
        for (int t = 0; t < NT1; ++t)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                hv[t][rt] = frag_load4<HID>(Hsave + (row0 + rt * 16 + c) * HID, t, g, valid[rt]);
        constexpr int LA = NT2 < 3 ? NT2 : 3;
        f32x4 b[NT2][RT], yv[NT2][RT];
#pragma unroll
        for (int u = 0; u < NT2 + LA; ++u) {
            if (u < NT2) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const int64_t row = row0 + rt * 16 + c;
                    b[u][rt] = frag_load4<OUT>(dY + row * ldy, u, g, valid[rt]);
                    if (ACT != ACT_NONE) yv[u][rt] = frag_load4<OUT>(Y + row * ldy, u, g, valid[rt]);
                }
            }
            const int w = u - LA;      // the tile consumed this round
            if (w < 0) continue;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                if (ACT != ACT_NONE) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) b[w][rt][r] *= frag_act_grad<ACT>(yv[w][rt][r]);
                }
                if (dZ2) frag_store4<OUT>(dZ2 + (row0 + rt * 16 + c) * OUT, w, g, valid[rt], b[w][rt]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (16 * w + j >= OUT) continue;
#pragma unroll
                for (int t = 0; t < NT1; ++t) {
                    const float a = W2n[(16 * w + 4 * g + j) * SA + 16 * t + c];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) adh[t][rt] = frag_mfma(a, b[w][rt][j], adh[t][rt]);
                }
            }
        }
        // ReLU mask from the saved activations -> dZ1
#pragma unroll
        for (int t = 0; t < NT1; ++t)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) adh[t][rt][r] = hv[t][rt][r] > 0.f ? adh[t][rt][r] : 0.f;
                frag_store4<HID>(dZ1 + (row0 + rt * 16 + c) * HID, t, g, valid[rt], adh[t][rt]);
            }
        if (dX) {
            f32x4 adx[NTX][RT];
#pragma unroll
            for (int v = 0; v < NTX; ++v)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) adx[v][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < NT1; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (16 * t + r >= HID) continue;
#pragma unroll
                    for (int v = 0; v < NTX; ++v) {
                        const float a = W1n[(16 * t + 4 * g + r) * SB + 16 * v + c];
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) adx[v][rt] = frag_mfma(a, adh[t][rt][r], adx[v][rt]);
                    }
                }
#pragma unroll
            for (int v = 0; v < NTX; ++v)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const int64_t row = row0 + rt * 16 + c;
                    const int64_t drow = dx_rows ? (valid[rt] ? dx_rows[row] : 0) : row;
                    f32x4 d = adx[v][rt];
                    if (accumulate_dx) d += frag_load4<IN>(dX + drow * lddx, v, g, valid[rt]);
                    frag_store4<IN>(dX + drow * lddx, v, g, valid[rt], d);
                }
        }
    }
}

// ------------------------------------------------------------------------------------------

static int num_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

template <int IN, int HID, int OUT, int ACT>
static int launch_fwd(const float *X, int64_t ldx, const float *W1, const float *b1, const float *W2, const float *b2,
                      float *Y, int64_t ldy, float *H, int64_t n, hipStream_t s) {
    // 16-row wave tiles, 16 waves per workgroup: 5-18 % faster than 32 rows / 8 waves on every shape (tools/mlp_tiling.sh)
    constexpr int RT = M2_RT, WAVES = M2_WAVES;
    const int64_t tiles = (n + 16 * RT - 1) / (16 * RT);
    const int64_t want = (tiles + WAVES - 1) / WAVES;
    const int grid = (int)(want < num_cus() ? want : num_cus());
    hipLaunchKernelGGL((mlp2_fwd_kernel<IN, HID, OUT, ACT, RT, WAVES>), dim3(grid), dim3(WAVES * 64), 0, s, X, ldx, W1, b1, W2,
                       b2, Y, ldy, H, n);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

template <int IN, int HID, int OUT, int ACT>
static int launch_bwd(const float *dY, const float *Y, int64_t ldy, const float *H, const float *W1, const float *W2,
                      float *dX, int64_t lddx, int acc, float *dZ2, float *dZ1, int64_t n, const int64_t *dx_rows,
                      hipStream_t s) {
    // 16-row wave tiles, 16 waves per workgroup: 5-18 % faster than 32 rows / 8 waves on every shape (tools/mlp_tiling.sh)
    constexpr int RT = M2_RT, WAVES = M2_WAVES;
    const int64_t tiles = (n + 16 * RT - 1) / (16 * RT);
    const int64_t want = (tiles + WAVES - 1) / WAVES;
    const int grid = (int)(want < num_cus() ? want : num_cus());
    hipLaunchKernelGGL((mlp2_bwd_kernel<IN, HID, OUT, ACT, RT, WAVES>), dim3(grid), dim3(WAVES * 64), 0, s, dY, Y, ldy, H, W1,
                       W2, dX, lddx, acc, dZ2, dZ1, n, dx_rows);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

#define MLP_CONFIGS(X_)           \
    X_(54, 50, 10, ACT_TANH)      \
    X_(54, 50, 30, ACT_SIGMOID)   \
    X_(54, 50, 70, ACT_NONE)      \
    X_(71, 100, 175, ACT_NONE)    \
    X_(15, 100, 175, ACT_NONE)    \
    X_(71, 100, 3, ACT_NONE)      \
    X_(15, 100, 3, ACT_NONE)

// Y = act(W2 relu(W1 x + b1) + b2); H [n,HID] receives relu(.) (may be NULL for inference).
extern "C" int cgs_mlp2_forward(int in, int hid, int out, int act, const float *X, int64_t ldx, const float *W1,
                                const float *b1, const float *W2, const float *b2, float *Y, int64_t ldy, float *H,
                                int64_t n, void *stream) {
    if (n < 0) { cgs_set_error("mlp2_forward: n < 0"); return CGS_ERR_ARG; }
    if (n == 0) return CGS_OK;
    if (!X || !W1 || !b1 || !W2 || !b2 || !Y) { cgs_set_error("mlp2_forward: NULL"); return CGS_ERR_ARG; }
    CgsProfScope prof(CGS_PROF_LMLP_FWD, (hipStream_t)stream);
#define X_(I, Hh, O, A) \
    if (in == I && hid == Hh && out == O && act == A) return launch_fwd<I, Hh, O, A>(X, ldx, W1, b1, W2, b2, Y, ldy, H, n, (hipStream_t)stream);
    MLP_CONFIGS(X_)
#undef X_
    cgs_set_error("mlp2_forward: no kernel instance for %d -> %d -> %d act %d", in, hid, out, act);
    return CGS_ERR_ARG;
}

// dX [n,lddx] (NULL to skip; accumulate_dx adds into it), dZ1 [n,hid], dZ2 [n,out] (may be NULL when act == none:
// then dZ2 == dY).  Weight/bias gradients are ACCUMULATED (atomics) into dW1/db1/dW2/db2: zero or pre-load them.
static int mlp2_backward_impl(int in, int hid, int out, int act, const float *X, int64_t ldx, const float *W1,
                              const float *b1, const float *W2, const float *Y, const float *dY, int64_t ldy, const float *H,
                              float *dX, int64_t lddx, int accumulate_dx, const int64_t *dx_rows, float *dZ1, float *dZ2,
                              float *dW1, float *db1, float *dW2, float *db2, int64_t n, void *scratch,
                              size_t scratch_bytes, void *stream_);

extern "C" int cgs_mlp2_backward(int in, int hid, int out, int act, const float *X, int64_t ldx, const float *W1,
                                 const float *b1, const float *W2, const float *Y, const float *dY, int64_t ldy, const float *H,
                                 float *dX, int64_t lddx, int accumulate_dx, float *dZ1, float *dZ2, float *dW1,
                                 float *db1, float *dW2, float *db2, int64_t n, void *scratch,
                                 size_t scratch_bytes, void *stream_) {
    return mlp2_backward_impl(in, hid, out, act, X, ldx, W1, b1, W2, Y, dY, ldy, H, dX, lddx, accumulate_dx, nullptr, dZ1, dZ2,
                              dW1, db1, dW2, db2, n, scratch, scratch_bytes, stream_);
}

// The same with the input gradient of row r stored to (accumulate_dx: added into) row dx_rows[r] of dX — the rows of a
// SUBSET of a larger batch (distinct indices): `dX.index_add_(0, dx_rows, .)` without the temporary and the launch.
// Needs the saved hidden layer (H != NULL).
extern "C" int cgs_mlp2_backward_rows(int in, int hid, int out, int act, const float *X, int64_t ldx, const float *W1,
                                      const float *b1, const float *W2, const float *Y, const float *dY, int64_t ldy,
                                      const float *H, float *dX, int64_t lddx, int accumulate_dx, const int64_t *dx_rows,
                                      float *dZ1, float *dZ2, float *dW1, float *db1, float *dW2, float *db2, int64_t n,
                                      void *scratch, size_t scratch_bytes, void *stream_) {
    if (dx_rows && !H) { cgs_set_error("mlp2_backward_rows: dx_rows needs the saved hidden layer"); return CGS_ERR_ARG; }
    return mlp2_backward_impl(in, hid, out, act, X, ldx, W1, b1, W2, Y, dY, ldy, H, dX, lddx, accumulate_dx, dx_rows, dZ1, dZ2,
                              dW1, db1, dW2, db2, n, scratch, scratch_bytes, stream_);
}

static int mlp2_backward_impl(int in, int hid, int out, int act, const float *X, int64_t ldx, const float *W1,
                              const float *b1, const float *W2, const float *Y, const float *dY, int64_t ldy, const float *H,
                              float *dX, int64_t lddx, int accumulate_dx, const int64_t *dx_rows, float *dZ1, float *dZ2,
                              float *dW1, float *db1, float *dW2, float *db2, int64_t n, void *scratch,
                              size_t scratch_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0) { cgs_set_error("mlp2_backward: n < 0"); return CGS_ERR_ARG; }
    if (n == 0) return CGS_OK;
    // dW1 == NULL: data gradients only — the weight-gradient products are the caller's to launch later (cgs_mlp2_wgrad)
    // (with H == NULL the recomputing kernel still accumulates the second layer: dW2 / db2 stay required there)
    const bool data_only = !dW1;
    const bool second_inside = data_only && !H;
    bool wg_ok = data_only ? (!db1 && (second_inside ? (dW2 && db2) : (!dW2 && !db2))) : (db1 && dW2 && db2);
    if (!X || !W1 || !W2 || !dY || !dZ1 || (act != ACT_NONE && (!Y || !dZ2)) || !wg_ok) {
        cgs_set_error("mlp2_backward: NULL (or a partial set of weight-gradient pointers)");
        return CGS_ERR_ARG;
    }
    int rc = CGS_ERR_ARG;
    if (!H) {
        // no saved hidden layer: recompute it (instances for the tiny-output shapes only, csrc/mlp_small.hip)
        if (act != ACT_NONE || !b1 || !scratch) { cgs_set_error("mlp2_backward: H == NULL needs act 0, b1 and scratch"); return CGS_ERR_ARG; }
        {
            CgsProfScope prof(CGS_PROF_LMLP_BWD, stream);
            rc = cgs_launch_mlp2_bwd_recompute(in, hid, out, X, ldx, W1, b1, W2, dY, ldy, dX, lddx, accumulate_dx, dZ1, dW2,
                                               db2, n, num_cus(), scratch, scratch_bytes, stream);
        }
        if (rc == -1) { cgs_set_error("mlp2_backward: no recompute instance for %d -> %d -> %d", in, hid, out); return CGS_ERR_ARG; }
        if (rc || data_only) return rc;
        CgsProfScope prof(CGS_PROF_LMLP_WGRAD, stream);
        const CgsWgProduct prod = {dZ1, hid, hid, X, ldx, in, dW1, db1};
        return cgs_launch_wgrad_multi(&prod, 1, n, num_cus(), scratch, scratch_bytes, stream);
    }
    {
        CgsProfScope prof(CGS_PROF_LMLP_BWD, stream);
        bool found = false;
#define X_(I, Hh, O, A)                                                                                              \
    if (!found && in == I && hid == Hh && out == O && act == A) {                                                     \
        found = true;                                                                                                 \
        rc = launch_bwd<I, Hh, O, A>(dY, Y, ldy, H, W1, W2, dX, lddx, accumulate_dx, dZ2, dZ1, n, dx_rows, stream);   \
    }
        MLP_CONFIGS(X_)
#undef X_
        if (!found) { cgs_set_error("mlp2_backward: no kernel instance for %d -> %d -> %d act %d", in, hid, out, act); return CGS_ERR_ARG; }
        if (rc || data_only) return rc;
    }
    CgsProfScope prof(CGS_PROF_LMLP_WGRAD, stream);
    const float *P2 = (act == ACT_NONE || !dZ2) ? dY : dZ2;
    const int64_t ldp2 = (act == ACT_NONE || !dZ2) ? ldy : out;
    const CgsWgProduct prods[2] = {{P2, ldp2, out, H, hid, hid, dW2, db2}, {dZ1, hid, hid, X, ldx, in, dW1, db1}};
    return cgs_launch_wgrad_multi(prods, 2, n, num_cus(), scratch, scratch_bytes, stream);
}

// The weight-gradient products of cgs_mlp2_backward as a call of their own (after a backward with dW1 == NULL):
// dW1 += dZ1^T X, db1 += sum dZ1 and — when H != NULL — dW2 += dZ2^T H, db2 += sum dZ2 (dZ2 [n, lddz2]: the backward's dZ2, or
// dY itself for act == 0).  H == NULL: the recomputing backward has accumulated the second layer already.
extern "C" int cgs_mlp2_wgrad(int in, int hid, int out, const float *X, int64_t ldx, const float *H, const float *dZ2,
                              int64_t lddz2, const float *dZ1, float *dW1, float *db1, float *dW2, float *db2, int64_t n,
                              void *scratch, size_t scratch_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0) { cgs_set_error("mlp2_wgrad: n < 0"); return CGS_ERR_ARG; }
    if (n == 0) return CGS_OK;
    if (!X || !dZ1 || !dW1 || !db1 || (H && (!dZ2 || !dW2 || !db2))) { cgs_set_error("mlp2_wgrad: NULL"); return CGS_ERR_ARG; }
    CgsProfScope prof(CGS_PROF_LMLP_WGRAD, stream);
    if (!H) {
        const CgsWgProduct prod = {dZ1, hid, hid, X, ldx, in, dW1, db1};
        return cgs_launch_wgrad_multi(&prod, 1, n, num_cus(), scratch, scratch_bytes, stream);
    }
    const CgsWgProduct prods[2] = {{dZ2, lddz2, out, H, hid, hid, dW2, db2}, {dZ1, hid, hid, X, ldx, in, dW1, db1}};
    return cgs_launch_wgrad_multi(prods, 2, n, num_cus(), scratch, scratch_bytes, stream);
}

// Workspace for the atomics-free weight-gradient reduction of cgs_mlp2_backward / cgs_anchor_mlp3_backward.
extern "C" size_t cgs_mlp_wgrad_scratch_bytes(void) { return cgs_wgrad_scratch_bytes_for(num_cus()); }
