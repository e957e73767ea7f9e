// One launch per level and direction for the "every row" half of the context model's level loop, training path
// (scene/gaussian_model.py:1594-1616 and its autograd): what round 4 ran as rowcat -> mlp2_fwd<.,100,3> -> noise_quant
// (forward) and noise_quant_bwd -> mlp2_bwd_rc -> wgrad_tail (backward), handing X, qadj, dqadj and dZ1 to each other
// through HBM (~1.9 KB per row forward, ~3.4 KB backward).
//
//   ctxl_fwd_kernel   gathers the level's MLP input row in the operand load ([anchor | coded feat | coded scaling | hyper],
//                     or [anchor * mask | hyper] for the first level), runs IN -> 100 (fp32 MFMA) -> the 3 step-size outputs
//                     (VALU), forms Q = clamp(Q0 (1 + tanh), 1e-9) and writes y = x + U(-1/2, 1/2) Q for the level's rows of
//                     the three parameter tensors (read through the coding permutation) while the row is in registers.
//                     Writes X (the backward's operand), y, Q; nothing else.
//   ctxl_bwd_kernel   recomputes the hidden layer from X (same MFMA chain = same bits), scatters dx = dy (+ the rate subset's
//                     compact gradients) into the parameter gradients, forms d qadj, the hidden-layer gradient, dX, AND the
//                     weight gradients dW1 = dZ1^T [X | 1] (row contraction on the matrix cores, 140 accumulator registers
//                     per lane) and dW2[step rows] = [H | 1]^T d qadj (28 more): one wave per SIMD, the unified 512-register file —
//                     the shape of mlp3_bwd_wg_kernel (mlp3.hip).  dZ1 never exists in memory.
//
// Every global operand goes through RAW BUFFER loads / stores (hardware bounds check): a lane that has nothing to load — a row
// past the end, a piece another lane group owns, a row outside the rate subset — issues the same instruction with an
// out-of-range offset and gets zeros, so there is NO branch around any load and no wait at a branch join: the next tile's
// operands are in flight while this tile is in the matrix pipe (the first version used predicated global loads, whose
// exec-mask joins drained the load queue three times per prefetch; profiles/r05_ctx_level.txt).
//
// The mean / scale outputs of mlp_grid on the ~15 % rate subset and the rate terms stay in their own (small) launches.
#include "cgs_internal.h"
#include "mlp_frag.h"
#include "buf_access.h"
#include "ctx_rows.h"
#include "ctx_noise.h"


// ---- the MLP input row ---------------------------------------------------------------------------------------
// Column order = the reference's cat (:1596-1599): [anchor 0..2 | feat 3..52 | scaling 53..58 | hyper 59..70] (IN = 71),
// [anchor 0..2 | hyper 3..14] (IN = 15).  As B fragments lane (g, c) holds columns 16q + 4g + {0..3} of row c.
struct ClGatherBufs { ClBuf anchor, mask, bf, bs, hyp; bool has_mask; };

// the raw pieces of one row's fragments, as issued (no dependence between them); cl_gather_merge forms the fragments
template <int IN>
struct ClGatherRaw {
    f32x4 f[4];            // IN 71: feature-column pieces of q = 0..3 (lane g == 0: q = 0 unused, see f00)
    f32x4 an, sc, h3, h4;  // anchor (g == 0), scaling piece, hyper pieces
    float f00, m;          // feat[0] (g == 0); mask byte as 0 / 1 (IN 15)
};

template <int IN>
__device__ __forceinline__ void cl_gather_issue(const ClGatherBufs &B, ClGatherRaw<IN> &w, int64_t row, int64_t arow, int64_t pos, int g,
                                                bool valid) {
    const uint32_t hy = (uint32_t)row * (CL_HY * 4), an = (uint32_t)arow * 12;
    w.an = cl_l96(B.anchor, cl_sel(valid && g == 0, an));
    if (IN == 15) {
        w.m = 1.f;
        if (B.has_mask) w.m = (float)__builtin_amdgcn_raw_buffer_load_b8(B.mask, (int)cl_sel(valid && g == 0, (uint32_t)arow), 0, 0);
        // g 0: hyper 0..3 (uses [0]); g 1: 1..4; g 2: 5..8; g 3: 9..12 (uses [0..2])
        w.h3 = cl_l128(B.hyp, cl_sel(valid, hy + (g == 0 ? 0u : (g == 1 ? 4u : (g == 2 ? 20u : 36u)))));
        return;
    }
    const uint32_t bf = (uint32_t)pos * (CL_D * 4), bs = (uint32_t)pos * (CL_S * 4);
    w.f00 = cl_l32(B.bf, cl_sel(valid && g == 0, bf));
    w.f[0] = cl_l128(B.bf, cl_sel(valid && g != 0, bf + (uint32_t)(4 * g - 3) * 4));
    w.f[1] = cl_l128(B.bf, cl_sel(valid, bf + (uint32_t)(16 + 4 * g - 3) * 4));
    w.f[2] = cl_l128(B.bf, cl_sel(valid, bf + (uint32_t)(32 + 4 * g - 3) * 4));
    w.f[3] = cl_l128(B.bf, cl_sel(valid && g < 2, bf + (g == 0 ? 45u : 46u) * 4));       // g 0: feat 45..48; g 1: 46..49 (uses [3])
    w.sc = cl_l96(B.bs, cl_sel(valid && (g == 1 || g == 2), bs + (g == 2 ? 12u : 0u)));     // g 1: scaling 0..2; g 2: 3..5
    w.h3 = cl_l128(B.hyp, cl_sel(valid && g >= 2, hy + (g == 3 ? 4u : 0u)));               // g 2: hyper 0..3 (uses [0]); g 3: 1..4
    w.h4 = cl_l128(B.hyp, cl_sel(valid && g < 2, hy + (g == 0 ? 20u : 36u)));               // g 0: hyper 5..8; g 1: 9..12 (uses [0..2])
}

template <int IN>
__device__ __forceinline__ void cl_gather_merge(const ClGatherRaw<IN> &w, int g, f32x4 (&xb)[ClShape<IN>::NTI]) {
    if (IN == 15) {
        // a product, like the reference's anchor * mask (-0.0 stays -0.0); lanes g > 0 never read w.an
        const f32x4 a0 = (f32x4){w.an[0] * w.m, w.an[1] * w.m, w.an[2] * w.m, w.h3[0]};
        const f32x4 t3 = (f32x4){w.h3[0], w.h3[1], w.h3[2], 0.f};
        xb[0] = g == 0 ? a0 : (g == 3 ? t3 : w.h3);
        return;
    }
    xb[0] = g == 0 ? (f32x4){w.an[0], w.an[1], w.an[2], w.f00} : w.f[0];
    xb[1] = w.f[1];
    xb[2] = w.f[2];
    const f32x4 g1 = (f32x4){w.f[3][3], w.sc[0], w.sc[1], w.sc[2]}, g2 = (f32x4){w.sc[0], w.sc[1], w.sc[2], w.h3[0]};
    xb[3] = g == 0 ? w.f[3] : (g == 1 ? g1 : (g == 2 ? g2 : w.h3));
    if (ClShape<IN>::NTI > 4) {
        const f32x4 t4 = (f32x4){w.h4[0], w.h4[1], w.h4[2], 0.f};
        xb[ClShape<IN>::NTI - 1] = g == 0 ? w.h4 : (g == 1 ? t4 : (f32x4){0.f, 0.f, 0.f, 0.f});
    }
}


// LDS image of the first layer for the transposed chain: W1t[(k * 16 + c) * 8 + t] = W1[16 t + c][k] (zeros in the padding), so
// that the seven A operands of a k-step (hidden tiles t = 0..6 of lane column c) are two 16-byte reads (conflict-free: eight
// consecutive lanes cover 64 consecutive dwords) instead of seven 4-byte ones
#define CL_W1T(XP) ((XP) * 128)
template <int IN>
__device__ __forceinline__ void cl_stage_w1t(float *__restrict__ W1t, const float *__restrict__ W1, int tid, int nthr) {
    frag_stage_loop(W1, CL_W1T(ClShape<IN>::XP), tid, nthr,
                    [](int i) { const int k = i >> 7, c = (i >> 3) & 15, t = i & 7, j = 16 * t + c;
                                return (t < CL_NT1 && j < CL_HID && k < IN) ? j * IN + k : -1; },
                    [&](int i, float v) { W1t[i] = v; });
}

// H^T = relu(W1 X^T + b1): lane (g, c) ends with hidden units 16t + 4g + {0..3} of row c.  Forward and backward call THIS
// chain, so the backward's recomputed hidden layer has the forward's bits.
template <int IN>
__device__ __forceinline__ void cl_layer1(const float *__restrict__ W1t, const float *__restrict__ b1s,
                                          const f32x4 (&xb)[ClShape<IN>::NTI], int g, int c, f32x4 (&acc1)[CL_NT1]) {
#pragma unroll
    for (int t = 0; t < CL_NT1; ++t) acc1[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // (the eight weights of group k + 1 are read before the seven MFMAs of group k are issued, behind scheduling fences: a read
    //  issued just in time leaves the matrix pipe of a one-wave-per-SIMD kernel idle for its round trip; same products, same order)
#if CL_PIPE
    {
        const float *wp0 = W1t + ((4 * g) * 16 + c) * 8;
        f32x4 lo = *(const f32x4 *)wp0, hi = *(const f32x4 *)(wp0 + 4);
#pragma unroll
        for (int q = 0; q < ClShape<IN>::NTI; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (16 * q + j >= IN) continue;
                const int qn = j == 3 ? q + 1 : q, jn = j == 3 ? 0 : j + 1;
                f32x4 lon = lo, hin = hi;
                if (qn < ClShape<IN>::NTI && 16 * qn + jn < IN) {
                    const float *wp = W1t + ((16 * qn + 4 * g + jn) * 16 + c) * 8;
                    lon = *(const f32x4 *)wp; hin = *(const f32x4 *)(wp + 4);
                }
                CL_FENCE();
#pragma unroll
                for (int t = 0; t < CL_NT1; ++t) acc1[t] = frag_mfma(t < 4 ? lo[t] : hi[t - 4], xb[q][j], acc1[t]);
                CL_FENCE();
                lo = lon; hi = hin;
            }
    }
#else
#pragma unroll
    for (int q = 0; q < ClShape<IN>::NTI; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (16 * q + j >= IN) continue;
            const float *wp = W1t + ((16 * q + 4 * g + j) * 16 + c) * 8;
            const f32x4 lo = *(const f32x4 *)wp, hi = *(const f32x4 *)(wp + 4);
#pragma unroll
            for (int t = 0; t < CL_NT1; ++t) acc1[t] = frag_mfma(t < 4 ? lo[t] : hi[t - 4], xb[q][j], acc1[t]);
        }
#endif
#pragma unroll
    for (int t = 0; t < CL_NT1; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc1[t][r] = fmaxf(acc1[t][r] + b1s[16 * t + 4 * g + r], 0.f);
}

// the three step-size outputs of row c, identical in its four lanes: each lane dots its 28 hidden units, the lanes are
// summed pairwise (a + b is commutative: every lane forms the same two partial sums)
__device__ __forceinline__ void cl_qadj(const float *__restrict__ W2qs, const float *__restrict__ b2qs,
                                        const f32x4 (&h)[CL_NT1], int g, float (&qa)[3]) {
    float p0 = 0.f, p1 = 0.f, p2 = 0.f;
#pragma unroll
    for (int t = 0; t < CL_NT1; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int hh = 16 * t + 4 * g + r;
            p0 = fmaf(W2qs[hh], h[t][r], p0);
            p1 = fmaf(W2qs[CL_HP + hh], h[t][r], p1);
            p2 = fmaf(W2qs[2 * CL_HP + hh], h[t][r], p2);
        }
    p0 += __shfl_xor(p0, 16); p1 += __shfl_xor(p1, 16); p2 += __shfl_xor(p2, 16);
    p0 += __shfl_xor(p0, 32); p1 += __shfl_xor(p1, 32); p2 += __shfl_xor(p2, 32);
    qa[0] = p0 + b2qs[0]; qa[1] = p1 + b2qs[1]; qa[2] = p2 + b2qs[2];
}


// the noise of the lane's pieces of level row r (same element -> value map as noise_quant_*_kernel: element r * W + column of
// tensor 0 / 1 / 2); pieces / values the lane does not own get 0
__device__ __forceinline__ void cl_row_noise(ClRow &u, uint32_t kf, uint32_t ks, uint32_t ko, int64_t r, int g) {
    const uint64_t ef = (uint64_t)r * CL_D, es = (uint64_t)r * CL_S, eo = (uint64_t)r * CL_O;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) u.F[i][j] = ctx_noise_k(kf, ef + 16 * i + 4 * g + j);
    const float n48 = ctx_noise_k(kf, ef + 48), n49 = ctx_noise_k(kf, ef + 49);
    u.F[3] = g == 0 ? (f32x4){n48, n49, 0.f, 0.f} : (f32x4){0.f, 0.f, 0.f, 0.f};
    // scaling: g 1 -> columns 0..3, g 2 -> columns 4, 5
    const int s0 = g == 2 ? 4 : 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float v = ctx_noise_k(ks, es + s0 + j);
        u.S[j] = (g == 1 || (g == 2 && j < 2)) ? v : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) u.O[0][j] = ctx_noise_k(ko, eo + 4 * g + j);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float v = ctx_noise_k(ko, eo + 16 + 4 * g + j);
        u.O[1][j] = (g == 3 && j >= 2) ? 0.f : v;
    }
}



// ------------------------------------------------------------------------------------------------------------
#ifndef CLF_WAVES
#define CLF_WAVES 8               // waves per workgroup of the forward
#endif
#ifndef CLF_OCC
#define CLF_OCC 0                 // > 0: force this many waves per SIMD (register budget) — tuning knob of tools/variant_lib.sh
#endif
#ifndef CLF_LDS_STORE
#define CLF_LDS_STORE 1           // the forward's outputs (X, yf / ys / yo, Q: tile images contiguous in memory) leave through a per-wave LDS patch
#endif
#ifndef CLF_GRID
#define CLF_GRID 2                // workgroups per CU in the forward's persistent grid
#endif
#if CLF_OCC > 0
#define CLF_ATTR __attribute__((amdgpu_waves_per_eu(CLF_OCC, CLF_OCC)))
#else
#define CLF_ATTR
#endif

struct ClFwdArgs {
    const float *anchor;          // [n_anchor, 3]
    const int64_t *a_rows;        // [n] anchor row of every level row (context levels: the parent; first level: the row itself)
    const uint8_t *a_mask;        // first level only, may be NULL: the anchor counts as anchor * (mask != 0) (:1758-1759)
    const float *base_f, *base_s; // [n_par, 50] [n_par, 6] coded prefix (context levels)
    const int64_t *pos;           // [n] the parent's position in the prefix
    const float *hyp;             // [n, 12] the level's noisy hyper latents
    const float *W1, *b1, *W2q, *b2q;        // W1 [100, IN], b1 [100], W2q [3, 100] (the step-size rows of the second layer), b2q [3]
    const float *xf, *xs, *xo;               // parameter tensors [n_anchor,50] [n_anchor,6] [n_anchor,30]
    const int64_t *rows;                     // [n] parameter row of every level row (the level's slice of the coding permutation)
    int64_t n, n_anchor, n_par;
    uint64_t seed;
    float q0f, q0s, q0o;
    float *X;                                // [n, IN]
    float *yf, *ys, *yo, *Q;                 // [n,50] [n,6] [n,30] [n,3]
    double *sums;                            // may be NULL (ctx_noise.h)
};

struct ClfIdx { int64_t arow, ppos, srow; };

template <int IN>
__global__ void __launch_bounds__(CLF_WAVES * 64) CLF_ATTR ctxl_fwd_kernel(ClFwdArgs a) {
    constexpr int NTI = ClShape<IN>::NTI, XP = ClShape<IN>::XP;
    __shared__ __attribute__((aligned(16))) float W1s[CL_W1T(XP)];        // cl_stage_w1t
    __shared__ float b1s[CL_HP];
    __shared__ float W2qs[3 * CL_HP];
    __shared__ float b2qs[4];
    __shared__ double part[CLF_WAVES][3];
    constexpr int PATCH = CLF_LDS_STORE ? (16 * IN > CL_ROW_PATCH ? 16 * IN : CL_ROW_PATCH) : 4;       // floats per wave
    static_assert(PATCH % 4 == 0, "16-byte aligned patches");
    __shared__ __attribute__((aligned(16))) float patches[CLF_WAVES * PATCH];
    const int tid = threadIdx.x, nthr = CLF_WAVES * 64;
    float *patch = patches + (tid >> 6) * PATCH;
    cl_stage_w1t<IN>(W1s, a.W1, tid, nthr);
    for (int i = tid; i < CL_HP; i += nthr) b1s[i] = i < CL_HID ? a.b1[i] : 0.f;
    for (int i = tid; i < 3 * CL_HP; i += nthr) W2qs[i] = (i % CL_HP) < CL_HID ? a.W2q[(i / CL_HP) * CL_HID + (i % CL_HP)] : 0.f;
    if (tid < 4) b2qs[tid] = tid < 3 ? a.b2q[tid] : 0.f;
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int64_t n = a.n, ntiles = (n + 15) / 16;
    const uint64_t nb = (uint64_t)n, NA = (uint64_t)a.n_anchor, NP = (uint64_t)a.n_par;
    const ClGatherBufs GB = {cl_buf(a.anchor, NA * 12), cl_buf(a.a_mask, NA), cl_buf(a.base_f, NP * CL_D * 4), cl_buf(a.base_s, NP * CL_S * 4),
                             cl_buf(a.hyp, nb * CL_HY * 4), a.a_mask != nullptr};
    const ClRowBufs XB = {cl_buf(a.xf, NA * CL_D * 4), cl_buf(a.xs, NA * CL_S * 4), cl_buf(a.xo, NA * CL_O * 4)};
    const ClRowBufs YB = {cl_buf(a.yf, nb * CL_D * 4), cl_buf(a.ys, nb * CL_S * 4), cl_buf(a.yo, nb * CL_O * 4)};
    const ClBuf bX = cl_buf(a.X, nb * IN * 4), bQ = cl_buf(a.Q, nb * 12);
    const ClBuf bAr = cl_buf(a.a_rows, nb * 8), bPos = cl_buf(a.pos, nb * 8), bRows = cl_buf(a.rows, nb * 8);
    const uint32_t kf = ctx_noise_key(a.seed, 0), ks = ctx_noise_key(a.seed, 1), ko = ctx_noise_key(a.seed, 2);
    const int64_t tstride = (int64_t)gridDim.x * CLF_WAVES;
    int64_t tile = (int64_t)blockIdx.x * CLF_WAVES + wave;
    float pf = 0.f, ps = 0.f, po = 0.f;

    auto idx_issue = [&](int64_t row) {
        const bool v = row < n;
        ClfIdx ix;
        ix.arow = cl_li64(bAr, cl_sel(v, (uint32_t)row * 8));
        ix.ppos = IN == 15 ? 0 : cl_li64(bPos, cl_sel(v, (uint32_t)row * 8));
        ix.srow = cl_li64(bRows, cl_sel(v, (uint32_t)row * 8));
        return ix;
    };
    // two waves per SIMD: the gathers of tile i + 1 are in flight while tile i is in the matrix pipe, their row indices are
    // fetched one tile before that
    ClGatherRaw<IN> gw;
    ClRow x;
    ClfIdx ix = idx_issue(tile * 16 + c);
    {
        const int64_t row = tile * 16 + c;
        const bool v0 = tile < ntiles && row < n;
        cl_gather_issue<IN>(GB, gw, row, ix.arow, ix.ppos, g, v0);
        cl_row_issue(x, XB, (uint32_t)ix.srow, g, v0);
        ix = idx_issue(row + tstride * 16);
    }
    for (; tile < ntiles; tile += tstride) {
        const int64_t row = tile * 16 + c, rn = row + tstride * 16;
        const bool valid = row < n;
        asm volatile("" ::: "memory");   // keep the LDS weight reads inside the tile loop (LICM would spill them)
        f32x4 xb[NTI];
        cl_gather_merge<IN>(gw, g, xb);
        ClRow xc = x;
        cl_row_mask(xc, g);
        CLB_FENCE();
        // the next tile's operands (their indices arrived during the previous tile) and the indices of the tile after it
        cl_gather_issue<IN>(GB, gw, rn, ix.arow, ix.ppos, g, rn < n);
        cl_row_issue(x, XB, (uint32_t)ix.srow, g, rn < n);
        ix = idx_issue(rn + tstride * 16);
        CLB_FENCE();
#if CLF_LDS_STORE
        cl_xrow_store_tile<IN>(bX, patch, (uint32_t)(tile * 16), g, c, lane, xb);
#else
        cl_xrow_store<IN>(bX, (uint32_t)row * (IN * 4), g, valid, xb);
#endif
        f32x4 acc1[CL_NT1];
        cl_layer1<IN>(W1s, b1s, xb, g, c, acc1);
        float qa[3];
        cl_qadj(W2qs, b2qs, acc1, g, qa);
        const float qf = ctx_step(a.q0f, qa[0]), qs = ctx_step(a.q0s, qa[1]), qo = ctx_step(a.q0o, qa[2]);
#if !CLF_LDS_STORE
        cl_s32(bQ, cl_sel(valid && g < 3, (uint32_t)row * 12 + (uint32_t)g * 4), g == 0 ? qf : (g == 1 ? qs : qo));
#endif
#pragma unroll
        for (int i = 0; i < 4; ++i) pf += cl_sum4(xc.F[i]);
        ps += cl_sum4(xc.S);
        po += cl_sum4(xc.O[0]) + cl_sum4(xc.O[1]);
        ClRow u;
        cl_row_noise(u, kf, ks, ko, row, g);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) xc.F[i][j] = xc.F[i][j] + u.F[i][j] * qf;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xc.S[j] = xc.S[j] + u.S[j] * qs;
            xc.O[0][j] = xc.O[0][j] + u.O[0][j] * qo;
            xc.O[1][j] = xc.O[1][j] + u.O[1][j] * qo;
        }
#if CLF_LDS_STORE
        cl_row_store_tile(xc, (f32x4){qf, qs, qo, 0.f}, YB, bQ, patch, (uint32_t)(tile * 16), g, c, lane);
#else
        cl_row_store(xc, YB, (uint32_t)row, g, valid);
#endif
    }
    if (a.sums) {        // one atomic per block and quantity, spread over CTX_SUM_SLOTS cache lines
        const double sa = ctx_wave_sum((double)pf), sb = ctx_wave_sum((double)ps), sc = ctx_wave_sum((double)po);
        if (lane == 0) { part[wave][0] = sa; part[wave][1] = sb; part[wave][2] = sc; }
        __syncthreads();
        if (tid < 3) {
            double v = 0.0;
            for (int w = 0; w < CLF_WAVES; ++w) v += part[w][tid];
            atomicAdd(&a.sums[(blockIdx.x % CTX_SUM_SLOTS) * CTX_SUM_STRIDE + tid], v);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
#define CLB_WAVES 4
#define CLB_LD 20                 // patch row stride (floats): the F -> N transposes are conflict-free both ways (mlp3.hip)
#define CLB_PATCH (16 * CLB_LD)

struct ClBwdArgs {
    const float *X;                          // [n, IN] (the forward's)
    const float *W1, *b1, *W2q, *b2q;
    const float *dyf, *dys, *dyo;            // [n,50] [n,6] [n,30] upstream gradients of y (each may be NULL = zeros)
    const float *dQ_ext;                     // [n,3] upstream gradient of Q, may be NULL
    const int64_t *rows;                     // [n]
    float *dxf, *dxs, *dxo;                  // FULL-size parameter gradients [n_anchor, .]: rows rows[r] are written
    const int32_t *side_map;                 // may be NULL; >= 0: the row is in the rate subset, its compact gradients are row side_map[r] of
    const float *side_f, *side_s, *side_o, *side_Q;     //   these [m,50] [m,6] [m,30] [m,3] arrays
    const float *dx_sub;                     //   and its extra input gradient (the mean / scale branch of the MLP) is row side_map[r] of this [m, IN]; may be NULL
    float *dX;                               // [n, IN] input-row gradient
    float *dhyp;                             // may be NULL; [n_anchor, 12]: row rows[r] receives the hyper columns of dX[r] (the last 12)
    float *partial;                          // [gridDim.x][E] weight-gradient images, E = 100 IN + 100 + 300 + 3
    int64_t n, n_anchor, m;
    uint64_t seed;
    float q0f, q0s, q0o;
};

__device__ __forceinline__ void clb_put(float *patch, f32x4 v, int g, int c) { *(f32x4 *)(patch + c * CLB_LD + 4 * g) = v; }
__device__ __forceinline__ f32x4 clb_get(const float *patch, int g, int c) {
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = patch[(4 * g + r) * CLB_LD + c];
    return o;
}

template <int IN>
struct ClbOps {            // the global operands of one 16-row tile, fetched one tile ahead; nothing here is branched on at issue
    f32x4 xb[ClShape<IN>::NTI];
    ClRow dy;
    float dq_ext;          // lane g < 3: dQ_ext[row, g]
    int64_t srow;
    int sm;                // -1: not in the rate subset (side_map == NULL reads as 0 and is ignored through has_side)
};

// dx = dy (+ the rate side) of the lane's pieces: stored to the parameter gradients and multiplied with the regenerated noise,
// piece by piece (nothing but the three sums stays live); the sums are then added over the row's 4 lanes
__device__ __forceinline__ void clb_dx_and_sums(const ClRow &dy, const ClRow &sd, const ClRowBufs &B, uint32_t srow, uint32_t kf,
                                                uint32_t ks, uint32_t ko, int64_t r, int g, bool on, float &af, float &as, float &ao) {
    const uint32_t of = srow * (CL_D * 4), os = srow * (CL_S * 4), oo = srow * (CL_O * 4);
    const uint64_t ef = (uint64_t)r * CL_D, es = (uint64_t)r * CL_S, eo = (uint64_t)r * CL_O;
    af = as = ao = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const f32x4 v = dy.F[i] + sd.F[i];
        cl_s128(B.f, cl_sel(on, of + (uint32_t)(16 * i + 4 * g) * 4), v);
#pragma unroll
        for (int j = 0; j < 4; ++j) af = fmaf(v[j], ctx_noise_k(kf, ef + 16 * i + 4 * g + j), af);
    }
    {       // columns 48, 49: lane g == 0 only (the other lanes' pieces are zeros, their store goes nowhere)
        const f32x4 v = dy.F[3] + sd.F[3];
        cl_s64(B.f, cl_sel(on && g == 0, of + 48 * 4), v);
        af = fmaf(v[0], ctx_noise_k(kf, ef + 48), af);
        af = fmaf(v[1], ctx_noise_k(kf, ef + 49), af);
    }
    {       // scaling: g 1 -> columns 0..3, g 2 -> columns 4, 5 ([2], [3] of that piece belong to the next row)
        f32x4 v = dy.S + sd.S;
        if (g == 2) { v[2] = 0.f; v[3] = 0.f; }
        cl_s128(B.s, cl_sel(on && g == 1, os), v);
        cl_s64(B.s, cl_sel(on && g == 2, os + 16), v);
        const int s0 = g == 2 ? 4 : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) as = fmaf(v[j], ctx_noise_k(ks, es + s0 + j), as);      // lanes g 0, 3: v = 0
    }
    {
        const f32x4 v = dy.O[0] + sd.O[0];
        cl_s128(B.o, cl_sel(on, oo + (uint32_t)(4 * g) * 4), v);
#pragma unroll
        for (int j = 0; j < 4; ++j) ao = fmaf(v[j], ctx_noise_k(ko, eo + 4 * g + j), ao);
    }
    {
        f32x4 v = dy.O[1] + sd.O[1];
        if (g == 3) { v[2] = 0.f; v[3] = 0.f; }
        cl_s128(B.o, cl_sel(on && g != 3, oo + (uint32_t)(16 + 4 * g) * 4), v);
        cl_s64(B.o, cl_sel(on && g == 3, oo + 28 * 4), v);
#pragma unroll
        for (int j = 0; j < 4; ++j) ao = fmaf(v[j], ctx_noise_k(ko, eo + 16 + 4 * g + j), ao);
    }
    af += __shfl_xor(af, 16); as += __shfl_xor(as, 16); ao += __shfl_xor(ao, 16);
    af += __shfl_xor(af, 32); as += __shfl_xor(as, 32); ao += __shfl_xor(ao, 32);
}

template <int IN>
__global__ void __launch_bounds__(CLB_WAVES * 64) __attribute__((amdgpu_waves_per_eu(1, 1))) ctxl_bwd_kernel(ClBwdArgs a) {
    constexpr int NTI = ClShape<IN>::NTI, XP = ClShape<IN>::XP;
    constexpr int NPATCH = CL_NT1 + NTI;                     // per wave: H, then dZ1 0..6, X 7..
    // LDS: W1t (layer 1, cl_stage_w1t) | W1n4 [h][c][4]: W1[h][16 v + c], v < 4 | W1n1 [h][c]: W1[h][64 + c] | b1 | W2q | b2q
    constexpr int N4 = CL_HP * 64, N1 = NTI > 4 ? CL_HP * 16 : 0;
    constexpr int WF = CL_W1T(XP) + N4 + N1 + CL_HP + 3 * CL_HP + 4;
    constexpr int PF = CLB_WAVES * (NPATCH * CLB_PATCH + 64);      // + the wave's 16 x 4 patch of d qadj
    constexpr int E = CL_HID * IN + CL_HID + 3 * CL_HID + 3;
    static_assert(WF + PF >= E, "the LDS image of the weight gradients reuses the weight region");
    static_assert(WF % 4 == 0, "patches are written with 16-byte stores");
    __shared__ __attribute__((aligned(16))) float lds[WF + PF];
    float *W1s = lds, *W1n4 = W1s + CL_W1T(XP), *W1n1 = W1n4 + N4, *b1s = W1n1 + N1, *W2qs = b1s + CL_HP, *b2qs = W2qs + 3 * CL_HP;
    const int tid = threadIdx.x, nthr = CLB_WAVES * 64;
    cl_stage_w1t<IN>(W1s, a.W1, tid, nthr);
    frag_stage_loop(a.W1, N4, tid, nthr,
                    [](int i) { const int h = i >> 6, c_ = (i >> 2) & 15, v = i & 3, k = 16 * v + c_; return (h < CL_HID && k < IN) ? h * IN + k : -1; },
                    [&](int i, float v) { W1n4[i] = v; });
    frag_stage_loop(a.W1, N1, tid, nthr, [](int i) { const int h = i >> 4, k = 64 + (i & 15); return (h < CL_HID && k < IN) ? h * IN + k : -1; },
                    [&](int i, float v) { W1n1[i] = v; });
    for (int i = tid; i < CL_HP; i += nthr) b1s[i] = i < CL_HID ? a.b1[i] : 0.f;
    for (int i = tid; i < 3 * CL_HP; i += nthr) W2qs[i] = (i % CL_HP) < CL_HID ? a.W2q[(i / CL_HP) * CL_HID + (i % CL_HP)] : 0.f;
    if (tid < 4) b2qs[tid] = tid < 3 ? a.b2q[tid] : 0.f;
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    float *patches = lds + WF + wave * (NPATCH * CLB_PATCH + 64);
    float *dqp = patches + NPATCH * CLB_PATCH;               // [16 rows][4]: d qadj of the tile's rows (column 3 = 0)
    const int64_t n = a.n, ntiles = (n + 15) / 16;
    const uint64_t nb = (uint64_t)n, NA = (uint64_t)a.n_anchor, M = (uint64_t)a.m;
    const bool has_side = a.side_map != nullptr;
    const ClBuf bX = cl_buf(a.X, nb * IN * 4), bdX = cl_buf(a.dX, nb * IN * 4), bSub = cl_buf(a.dx_sub, M * IN * 4);
    const ClRowBufs DYB = {cl_buf(a.dyf, nb * CL_D * 4), cl_buf(a.dys, nb * CL_S * 4), cl_buf(a.dyo, nb * CL_O * 4)};
    const ClRowBufs DXB = {cl_buf(a.dxf, NA * CL_D * 4), cl_buf(a.dxs, NA * CL_S * 4), cl_buf(a.dxo, NA * CL_O * 4)};
    const ClRowBufs SDB = {cl_buf(a.side_f, M * CL_D * 4), cl_buf(a.side_s, M * CL_S * 4), cl_buf(a.side_o, M * CL_O * 4)};
    const ClBuf bSQ = cl_buf(a.side_Q, M * 12), bQe = cl_buf(a.dQ_ext, nb * 12), bRows = cl_buf(a.rows, nb * 8),
                bMap = cl_buf(a.side_map, nb * 4), bDH = cl_buf(a.dhyp, NA * 48);
    const uint32_t kf = ctx_noise_key(a.seed, 0), ks = ctx_noise_key(a.seed, 1), ko = ctx_noise_key(a.seed, 2);
    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
    // accumulators held for the whole launch (matrix-core outputs: AGPRs)
    f32x4 aw1[CL_NT1][NTI];                  // reg r of lane (g, c) = dW1[16t + 4g + r][16v + c]; column IN = db1
    f32x4 aw2[CL_NT1];                       // reg r of lane (g, c < 3) = dW2q[c][16t + 4g + r]; hidden index 100 = db2q[c]
#pragma unroll
    for (int t = 0; t < CL_NT1; ++t) {
        aw2[t] = zero;
#pragma unroll
        for (int v = 0; v < NTI; ++v) aw1[t][v] = zero;
    }

    // prefetch of the next tile's operands in two stages (register pressure): X and the scalars mid-tile, dy before the dW1
    // products (its registers are those the dX chain has just released)
    auto op_issue_x = [&](ClbOps<IN> &op, int64_t row) {
        const bool v = row < n;
        cl_xrow_load<IN>(bX, (uint32_t)row * (IN * 4), g, v, op.xb);
        op.srow = cl_li64(bRows, cl_sel(v, (uint32_t)row * 8));
        op.sm = __builtin_amdgcn_raw_buffer_load_b32(bMap, (int)cl_sel(v, (uint32_t)row * 4), 0, 0);
        op.dq_ext = cl_l32(bQe, cl_sel(v && g < 3, (uint32_t)row * 12 + (uint32_t)g * 4));
    };
    auto op_issue_dy = [&](ClbOps<IN> &op, int64_t row) { cl_row_issue(op.dy, DYB, (uint32_t)row, g, row < n); };

    const int64_t tstride = (int64_t)gridDim.x * CLB_WAVES;
    int64_t tile = (int64_t)blockIdx.x * CLB_WAVES + wave;
    ClbOps<IN> op;
    op_issue_x(op, tile < ntiles ? tile * 16 + c : n);
    op_issue_dy(op, tile < ntiles ? tile * 16 + c : n);
    for (; tile < ntiles; tile += tstride) {
        const int64_t row0 = tile * 16, row = row0 + c;
        const bool valid = row < n;
        asm volatile("" ::: "memory");   // keep the LDS weight reads inside the tile loop (LICM would spill them)
        const bool chosen = has_side && valid && op.sm >= 0;
        const uint32_t sm = chosen ? (uint32_t)op.sm : 0u;
        // the rate subset's compact gradients of this row (needed after the first block of MFMAs)
        ClRow sd;
        cl_row_issue(sd, SDB, sm, g, chosen);
        const float sq = cl_l32(bSQ, cl_sel(chosen && g < 3, sm * 12 + (uint32_t)g * 4));
        // [X | 1] towards layout N; H^T = relu(W1 X^T + b1) (the forward's chain, the forward's bits), H towards layout N
        cl_xrow_mask<IN>(g, op.xb);
#pragma unroll
        for (int q = 0; q < NTI; ++q) clb_put(patches + (CL_NT1 + q) * CLB_PATCH, op.xb[q], g, c);
        f32x4 acc1[CL_NT1];
        cl_layer1<IN>(W1s, b1s, op.xb, g, c, acc1);
#pragma unroll
        for (int t = 0; t < CL_NT1; ++t) clb_put(patches + t * CLB_PATCH, acc1[t], g, c);
        float qa[3];
        cl_qadj(W2qs, b2qs, acc1, g, qa);
        // dx = dy (+ rate side), scattered to the parameter rows; sum_c dx u per tensor
        float af, as, ao;
        clb_dx_and_sums(op.dy, sd, DXB, (uint32_t)op.srow, kf, ks, ko, row, g, valid, af, as, ao);
        // lane g < 3 holds the external + side gradient of Q[row, g]: bring all three to every lane of the row
        const float ext = op.dq_ext + sq;
        const float e0 = __shfl(ext, c, 64), e1 = __shfl(ext, 16 + c, 64), e2 = __shfl(ext, 32 + c, 64);
        float dq[3];
        {
            const float t0 = tanhf(qa[0]), t1 = tanhf(qa[1]), t2 = tanhf(qa[2]);
            dq[0] = (valid && a.q0f * (1.f + t0) >= 1e-9f) ? (af + e0) * a.q0f * (1.f - t0 * t0) : 0.f;
            dq[1] = (valid && a.q0s * (1.f + t1) >= 1e-9f) ? (as + e1) * a.q0s * (1.f - t1 * t1) : 0.f;
            dq[2] = (valid && a.q0o * (1.f + t2) >= 1e-9f) ? (ao + e2) * a.q0o * (1.f - t2 * t2) : 0.f;
        }
        if (g == 0) *(f32x4 *)(dqp + 4 * c) = (f32x4){dq[0], dq[1], dq[2], 0.f};
        // this tile's global operands are consumed: the next tile's are fetched behind ~300 MFMAs
        CLB_FENCE();
        f32x4 dxs[NTI];
        cl_xrow_load<IN>(bSub, sm * (IN * 4), g, chosen, dxs);
        const uint32_t hrow = (uint32_t)op.srow * 48u;             // this tile's row of dhyp (op.srow is the next tile's from here on)
        op_issue_x(op, row + tstride * 16);
        CLB_FENCE();
        // rows 4g + r of this tile that exist (the ones columns of [X | 1] and [H | 1]); d qadj in layout N
        f32x4 ones, dqn;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ones[r] = row0 + 4 * g + r < n ? 1.f : 0.f;
            dqn[r] = dqp[(4 * g + r) * 4 + (c & 3)];
        }
        if (c >= 4) dqn = zero;
        // per hidden tile: dW2q^T += [H | 1]^T dq (row contraction), then dZ1 = relu'(.) * W2q^T dq over the same patch
        f32x4 hn_next = clb_get(patches, g, c);
#pragma unroll
        for (int t = 0; t < CL_NT1; ++t) {
            f32x4 hn = hn_next;
            if (CL_PIPE && t + 1 < CL_NT1) {          // the next tile of H^T is in flight during this tile's products
                hn_next = clb_get(patches + (t + 1) * CLB_PATCH, g, c);
                CLB_FENCE();
            }
            if (t == CL_NT1 - 1 && c == CL_HID - 16 * (CL_NT1 - 1)) hn = ones;           // hidden index 100 of [H | 1]
#pragma unroll
            for (int r = 0; r < 4; ++r) aw2[t] = frag_mfma(hn[r], dqn[r], aw2[t]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int hh = 16 * t + 4 * g + r;
                float dz = 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k) dz = fmaf(dq[k], W2qs[k * CL_HP + hh], dz);
                acc1[t][r] = acc1[t][r] > 0.f ? dz : 0.f;
            }
            clb_put(patches + t * CLB_PATCH, acc1[t], g, c);
            if (!CL_PIPE && t + 1 < CL_NT1) hn_next = clb_get(patches + (t + 1) * CLB_PATCH, g, c);
        }
        // dX = W1^T dZ1 (+ the subset's mean / scale branch)
        f32x4 adx[NTI];
#pragma unroll
        for (int v = 0; v < NTI; ++v) adx[v] = zero;
        {
            f32x4 w4 = *(const f32x4 *)(W1n4 + ((4 * g) * 16 + c) * 4);
            float w1 = NTI > 4 ? W1n1[(4 * g) * 16 + c] : 0.f;
#pragma unroll
            for (int t = 0; t < CL_NT1; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (16 * t + r >= CL_HID) continue;
                    const int tn = r == 3 ? t + 1 : t, rn = r == 3 ? 0 : r + 1;
                    f32x4 w4n = w4;
                    float w1n = w1;
                    if (tn < CL_NT1 && 16 * tn + rn < CL_HID) {
                        const int hcn = (16 * tn + 4 * g + rn) * 16 + c;
                        w4n = *(const f32x4 *)(W1n4 + hcn * 4);
                        if (NTI > 4) w1n = W1n1[hcn];
                    }
                    if (CL_PIPE) CLB_FENCE();
#pragma unroll
                    for (int v = 0; v < NTI && v < 4; ++v) adx[v] = frag_mfma(w4[v], acc1[t][r], adx[v]);
                    if (NTI > 4) adx[NTI - 1] = frag_mfma(w1, acc1[t][r], adx[NTI - 1]);
                    if (CL_PIPE) CLB_FENCE();
                    w4 = w4n; w1 = w1n;
                }
        }
#pragma unroll
        for (int v = 0; v < NTI; ++v) adx[v] += dxs[v];          // (a dx_sub piece that runs past its row end is cut by the store)
        cl_xrow_store<IN>(bdX, (uint32_t)row * (IN * 4), g, valid, adx);
        // the hyper latents' columns (the last 12 of the row) also go straight to the latents' gradient row, in parameter order: their
        // consumer (the hyper prior's backward) then needs no pass through the inverse coding permutation.  A NULL dhyp drops the stores.
        {
            constexpr int H0 = IN - 12;
#pragma unroll
            for (int v = 0; v < NTI; ++v) {
                if (16 * v + 15 < H0) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int h = 16 * v + 4 * g + j - H0;
                    cl_s32(bDH, cl_sel(valid && h >= 0 && h < 12, hrow + (uint32_t)h * 4u), adx[v][j]);
                }
            }
        }
        // dW1 += dZ1^T [X | 1]
        f32x4 xn[NTI];
#pragma unroll
        for (int q = 0; q < NTI; ++q) xn[q] = clb_get(patches + (CL_NT1 + q) * CLB_PATCH, g, c);
        if (c == IN - 16 * (NTI - 1)) xn[NTI - 1] = ones;                 // column IN of [X | 1]
        CLB_FENCE();
        op_issue_dy(op, row + tstride * 16);
        CLB_FENCE();
        f32x4 dn_next = clb_get(patches, g, c);
#pragma unroll
        for (int t = 0; t < CL_NT1; ++t) {
            const f32x4 dn = dn_next;
            if (t + 1 < CL_NT1) dn_next = clb_get(patches + (t + 1) * CLB_PATCH, g, c);
            if (CL_PIPE) CLB_FENCE();
#pragma unroll
            for (int v = 0; v < NTI; ++v)
#pragma unroll
                for (int r = 0; r < 4; ++r) aw1[t][v] = frag_mfma(dn[r], xn[v][r], aw1[t][v]);
            if (CL_PIPE) CLB_FENCE();
        }
    }
    // ---- the workgroup's image [dW1 100 x IN | db1 100 | dW2q 3 x 100 | db2q 3]: the waves take turns (fixed order) ----
    __syncthreads();
    for (int i = tid; i < E; i += nthr) lds[i] = 0.f;
    __syncthreads();
    float *img1 = lds, *imgb1 = img1 + CL_HID * IN, *img2 = imgb1 + CL_HID, *imgb2 = img2 + 3 * CL_HID;
    for (int w = 0; w < CLB_WAVES; ++w) {
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < CL_NT1; ++t) {
#pragma unroll
                for (int v = 0; v < NTI; ++v)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = 16 * t + 4 * g + r, k = 16 * v + c;
                        if (m < CL_HID) {
                            if (k < IN) img1[m * IN + k] += aw1[t][v][r];
                            else if (k == IN) imgb1[m] += aw1[t][v][r];
                        }
                    }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int hh = 16 * t + 4 * g + r;
                    if (c < 3) {
                        if (hh < CL_HID) img2[c * CL_HID + hh] += aw2[t][r];
                        else if (hh == CL_HID) imgb2[c] += aw2[t][r];
                    }
                }
            }
        }
        __syncthreads();
    }
    float *dst = a.partial + (int64_t)blockIdx.x * E;
    for (int i = tid; i < E; i += nthr) dst[i] = lds[i];
}

// ------------------------------------------------------------------------------------------------------------
static int cl_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

static bool cl_fits(int64_t rows, int64_t row_bytes) { return rows >= 0 && (uint64_t)rows * (uint64_t)row_bytes < CL_MAX_BYTES; }

// Forward of one level, every row (see the head of this file).  in_dim 71: context level — X[r] = [anchor[a_rows[r]] |
// base_f[pos[r]] | base_s[pos[r]] | hyp[r]]; in_dim 15: first level — X[r] = [anchor[a_rows[r]] (* a_mask) | hyp[r]].
// W2q / b2q: the LAST three rows of mlp_grid's second layer (the step-size adjustments, :1603-1608).
extern "C" int cgs_ctx_level_fwd(int in_dim, const float *anchor, int64_t n_anchor, const int64_t *a_rows, const uint8_t *a_mask,
                                 const float *base_f, const float *base_s, int64_t n_par, const int64_t *pos, const float *hyp,
                                 int64_t n, const float *W1, const float *b1, const float *W2q, const float *b2q, const float *xf,
                                 const float *xs, const float *xo, const int64_t *rows, uint64_t seed, float q0f, float q0s,
                                 float q0o, float *X, float *yf, float *ys, float *yo, float *Q, double *sums3, void *stream) {
    if (n < 0 || n_anchor < 0 || n_par < 0 || (in_dim != 71 && in_dim != 15)) { cgs_set_error("ctx_level_fwd: bad args (in_dim 71 or 15)"); return CGS_ERR_ARG; }
    if (n == 0) return CGS_OK;
    if (!anchor || !a_rows || !hyp || !W1 || !b1 || !W2q || !b2q || !xf || !xs || !xo || !rows || !X || !yf || !ys || !yo || !Q ||
        (in_dim == 71 && (!base_f || !base_s || !pos))) {
        cgs_set_error("ctx_level_fwd: NULL");
        return CGS_ERR_ARG;
    }
    // 32-bit buffer offsets (raw buffer instructions): every operand must stay below 4 GB
    if (!cl_fits(n, in_dim * 4) || !cl_fits(n, CL_D * 4) || !cl_fits(n_anchor, CL_D * 4) || !cl_fits(n_par, CL_D * 4)) {
        cgs_set_error("ctx_level_fwd: an operand exceeds 4 GB (n %lld, n_anchor %lld)", (long long)n, (long long)n_anchor);
        return CGS_ERR_ARG;
    }
    ClFwdArgs a;
    a.anchor = anchor; a.a_rows = a_rows; a.a_mask = in_dim == 15 ? a_mask : nullptr; a.base_f = base_f; a.base_s = base_s; a.pos = pos;
    a.hyp = hyp; a.W1 = W1; a.b1 = b1; a.W2q = W2q; a.b2q = b2q;
    a.xf = xf; a.xs = xs; a.xo = xo; a.rows = rows; a.n = n; a.n_anchor = n_anchor; a.n_par = n_par; a.seed = seed;
    a.q0f = q0f; a.q0s = q0s; a.q0o = q0o;
    a.X = X; a.yf = yf; a.ys = ys; a.yo = yo; a.Q = Q; a.sums = sums3;
    const int64_t tiles = (n + 15) / 16, want = (tiles + CLF_WAVES - 1) / CLF_WAVES;
    const int64_t cap = (int64_t)CLF_GRID * cl_cus();
    const unsigned grid = (unsigned)(want < cap ? want : cap);
    CgsProfScope prof(CGS_PROF_CTX_FWD, (hipStream_t)stream);
    if (in_dim == 71) hipLaunchKernelGGL((ctxl_fwd_kernel<71>), dim3(grid), dim3(CLF_WAVES * 64), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((ctxl_fwd_kernel<15>), dim3(grid), dim3(CLF_WAVES * 64), 0, (hipStream_t)stream, a);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

extern "C" size_t cgs_ctx_level_bwd_scratch_bytes(void) { return (size_t)cl_cus() * (CL_HID * 71 + CL_HID + 3 * CL_HID + 3) * sizeof(float); }

// Backward of cgs_ctx_level_fwd.  dxf / dxs / dxo: the FULL-size parameter gradients [n_anchor, .], rows rows[r] are
// overwritten with dy[r] (+ the rate subset's compact gradient row side_map[r] when >= 0; the side arrays have m_side rows).
// dX [n, in_dim] receives the input-row gradient (+ row side_map[r] of dx_sub).  dW1 / db1 / dW2q / db2q are ACCUMULATED into
// (zero or pre-load them): atomics-free, the same bits every run.  scratch >= cgs_ctx_level_bwd_scratch_bytes().
// d_hyp_rows (cgs_ctx_level_bwd2; may be NULL): [n_anchor, 12] — row rows[r] also receives the last 12 columns of dX[r] (the gradient
// of the level's hyper latents, in the latents' own row order).
extern "C" int cgs_ctx_level_bwd2(int in_dim, const float *X, const float *W1, const float *b1, const float *W2q,
                                  const float *b2q, const float *dyf, const float *dys, const float *dyo, const float *dQ_ext,
                                  int64_t n, uint64_t seed, float q0f, float q0s, float q0o, const int64_t *rows, int64_t n_anchor,
                                  float *dxf, float *dxs, float *dxo, const int32_t *side_map, int64_t m_side, const float *side_f,
                                  const float *side_s, const float *side_o, const float *side_Q, const float *dx_sub, float *dX,
                                  float *d_hyp_rows, float *dW1, float *db1, float *dW2q, float *db2q, void *scratch,
                                  size_t scratch_bytes, void *stream);
extern "C" int cgs_ctx_level_bwd(int in_dim, const float *X, const float *W1, const float *b1, const float *W2q,
                                 const float *b2q, const float *dyf, const float *dys, const float *dyo, const float *dQ_ext,
                                 int64_t n, uint64_t seed, float q0f, float q0s, float q0o, const int64_t *rows, int64_t n_anchor,
                                 float *dxf, float *dxs, float *dxo, const int32_t *side_map, int64_t m_side, const float *side_f,
                                 const float *side_s, const float *side_o, const float *side_Q, const float *dx_sub, float *dX,
                                 float *dW1, float *db1, float *dW2q, float *db2q, void *scratch, size_t scratch_bytes, void *stream) {
    return cgs_ctx_level_bwd2(in_dim, X, W1, b1, W2q, b2q, dyf, dys, dyo, dQ_ext, n, seed, q0f, q0s, q0o, rows, n_anchor, dxf, dxs, dxo,
                              side_map, m_side, side_f, side_s, side_o, side_Q, dx_sub, dX, nullptr, dW1, db1, dW2q, db2q, scratch,
                              scratch_bytes, stream);
}

extern "C" int cgs_ctx_level_bwd2(int in_dim, const float *X, const float *W1, const float *b1, const float *W2q,
                                  const float *b2q, const float *dyf, const float *dys, const float *dyo, const float *dQ_ext,
                                  int64_t n, uint64_t seed, float q0f, float q0s, float q0o, const int64_t *rows, int64_t n_anchor,
                                  float *dxf, float *dxs, float *dxo, const int32_t *side_map, int64_t m_side, const float *side_f,
                                  const float *side_s, const float *side_o, const float *side_Q, const float *dx_sub, float *dX,
                                  float *d_hyp_rows, float *dW1, float *db1, float *dW2q, float *db2q, void *scratch,
                                  size_t scratch_bytes, void *stream) {
    if (n < 0 || n_anchor < 0 || m_side < 0 || (in_dim != 71 && in_dim != 15)) { cgs_set_error("ctx_level_bwd: bad args (in_dim 71 or 15)"); return CGS_ERR_ARG; }
    if (n == 0) return CGS_OK;
    if (!X || !W1 || !b1 || !W2q || !b2q || !rows || !dxf || !dxs || !dxo || !dX || !dW1 || !db1 || !dW2q || !db2q || !scratch) {
        cgs_set_error("ctx_level_bwd: NULL");
        return CGS_ERR_ARG;
    }
    if (side_map && m_side > 0 && (!side_f || !side_s || !side_o || !side_Q)) { cgs_set_error("ctx_level_bwd: side_map needs the four side arrays"); return CGS_ERR_ARG; }
    if (!cl_fits(n, in_dim * 4) || !cl_fits(n, CL_D * 4) || !cl_fits(n_anchor, CL_D * 4) || !cl_fits(m_side, in_dim * 4)) {
        cgs_set_error("ctx_level_bwd: an operand exceeds 4 GB (n %lld, n_anchor %lld)", (long long)n, (long long)n_anchor);
        return CGS_ERR_ARG;
    }
    const int E = CL_HID * in_dim + CL_HID + 3 * CL_HID + 3;
    const int64_t tiles = (n + 15) / 16, want = (tiles + CLB_WAVES - 1) / CLB_WAVES;
    int64_t grid = want < cl_cus() ? want : cl_cus();
    if ((size_t)grid * E * sizeof(float) > scratch_bytes) { cgs_set_error("ctx_level_bwd: scratch too small"); return CGS_ERR_WORKSPACE; }
    ClBwdArgs a;
    const bool side = side_map != nullptr && m_side > 0;
    a.X = X; a.W1 = W1; a.b1 = b1; a.W2q = W2q; a.b2q = b2q; a.dyf = dyf; a.dys = dys; a.dyo = dyo; a.dQ_ext = dQ_ext;
    a.rows = rows; a.dxf = dxf; a.dxs = dxs; a.dxo = dxo; a.side_map = side ? side_map : nullptr; a.side_f = side ? side_f : nullptr;
    a.side_s = side ? side_s : nullptr; a.side_o = side ? side_o : nullptr; a.side_Q = side ? side_Q : nullptr;
    a.dx_sub = side ? dx_sub : nullptr; a.dX = dX; a.dhyp = d_hyp_rows; a.partial = (float *)scratch;
    a.n = n; a.n_anchor = n_anchor; a.m = side ? m_side : 0; a.seed = seed; a.q0f = q0f; a.q0s = q0s; a.q0o = q0o;
    {
        CgsProfScope prof(CGS_PROF_CTX_BWD, (hipStream_t)stream);
        if (in_dim == 71) hipLaunchKernelGGL((ctxl_bwd_kernel<71>), dim3((unsigned)grid), dim3(CLB_WAVES * 64), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((ctxl_bwd_kernel<15>), dim3((unsigned)grid), dim3(CLB_WAVES * 64), 0, (hipStream_t)stream, a);
        CGS_CHECK_HIP(hipGetLastError());
    }
    CgsProfScope prof(CGS_PROF_LMLP_WGRAD, (hipStream_t)stream);
    const CgsWgProduct prods[2] = {{nullptr, 0, CL_HID, nullptr, 0, in_dim, dW1, db1}, {nullptr, 0, 3, nullptr, 0, CL_HID, dW2q, db2q}};
    return cgs_launch_wgrad_reduce((const float *)scratch, (int)grid, prods, 2, (hipStream_t)stream);
}
