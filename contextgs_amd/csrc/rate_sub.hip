// The rate half of the context model's level loop on the ~15 % rate subset, fused (round 6): what rounds 3-5 ran as
//   forward   row gather X[loc] -> mlp2_fwd<IN,100,175> (pred [m,175] + H [m,100] to HBM) -> level_rate_fwd
//   backward  level_rate_bwd (d_pred [m,175] to HBM) -> mlp2_bwd (dZ1, dX to HBM) -> wgrad_multi + 2 reductions
// i.e. ~8 launches per level and direction that hand ~3.5 KB per chosen row to each other through HBM
// (scene/gaussian_model.py:1600-1608 on the chosen rows, :1658-1694, utils/entropy_models.py:30-50).
//
//   rs_main_kernel<IN, false>   forward: X rows through `loc`, hidden layer, the 172 mean / scale outputs of mlp_grid, the three
//                               Entropy_gaussian terms of every element against the level's noisy values, the three bit sums.
//                               NOTHING is written but the sums: the [m,175] prediction never exists in memory.
//   rs_main_kernel<IN, true>    backward, data half: the forward is recomputed (same MFMA chains = same bits), d bits / d(mean,
//                               scale, x, Q, mask weight) formed in registers, dH = W2^T dZ2, dZ1, dX = W1^T dZ1 on the matrix
//                               cores; writes the compact side arrays (gradients of the noisy values / step sizes / input row that
//                               cgs_ctx_level_bwd adds in) and the operands of the weight gradients as 16-row tiles in COLUMN-major
//                               order ([tile][column][16 rows]): the layout whose fragments are one 16-byte load per lane.
//   rs_wgrad_kernel<IN>         dW2 = dZ2^T [H | 1], dW1 = dZ1^T [X | 1] with the ROW index as the MFMA contraction: the 119 output
//                               tiles are split over the eight waves of a workgroup (60-80 accumulator registers each, two waves per
//                               SIMD), every operand fragment is ONE coalesced 1 KB wave load, no LDS, no atomics; per-workgroup
//                               images are summed in block order by wgrad_multi_reduce_kernel (bit-reproducible).
//
// Output tiles of the second layer are PERMUTED so that the mean and the scale of an element sit in the same lane and register
// slot of two neighbouring tiles, and so that a lane's four elements are one 16-byte piece of the level's noisy values in the
// ClRow layout (ctx_rows.h): block b in 0..5 = [feat 0..15 | 16..31 | 32..47 | feat 48,49 + scaling 0..5 | offsets 0..15 | 16..29],
// tile 2b = its means, tile 2b + 1 = its scales.  The permutation lives in the LDS image of W2 and in the dump of dW2 only.
#include "cgs_internal.h"
#include "mlp_frag.h"
#include "buf_access.h"
#include "ctx_rows.h"
#include "rate_math.h"

#define RS_NB 6                   // element blocks
#define RS_NT2 12                 // permuted output tiles
#define RS_OP 192
#define RS_SA 116                 // row stride of the [o'][h] image of W2: b32 reads at (4g + j) * SA + c are conflict-free
#define RS_OUT 175                // rows of W2 (172 mean / scale rows + the 3 step-size rows cgs_ctx_level_* own)
#define RS_K 10                   // mask weights per anchor
#ifndef RS_WAVES_F
#define RS_WAVES_F 8              // waves per workgroup, forward (one workgroup per CU: the weights fill 128 KB of LDS)
#endif
#ifndef RS_WAVES_B
#define RS_WAVES_B 8              // backward: two waves per SIMD, 256 registers each
#endif
#ifndef RS_STAGGER
#define RS_STAGGER 0              // > 0: the second wave of every SIMD starts RS_STAGGER x 4096 cycles late (MFMA phase of one wave
#endif                            //      over the element maths of the other)

// permuted output row o' = 16 (2 b + is_scale) + 4 g + r  ->  row of mlp_grid's second layer, -1 = padding
__host__ __device__ __forceinline__ int rs_w2row(int op) {
    const int u = op >> 4, b = u >> 1, sc = u & 1, l = op & 15;
    if (b < 3) { const int e = 16 * b + l; return sc ? 50 + e : e; }
    if (b == 3) {
        if (l < 2) return sc ? 98 + l : 48 + l;                         // feat 48, 49 (g 0)
        if (l >= 4 && l < 8) return (sc ? 106 : 100) + (l - 4);         // scaling 0..3 (g 1)
        if (l >= 8 && l < 10) return (sc ? 110 : 104) + (l - 8);        // scaling 4, 5 (g 2)
        return -1;
    }
    const int e = (b == 4 ? 0 : 16) + l;
    if (e >= 30) return -1;
    return (sc ? 142 : 112) + e;
}

// the inverse: row of the second layer (< 172) -> permuted output row
__device__ __forceinline__ int rs_w2perm(int row) {
    int sc = 0, e = row, base;
    if (row < 100) { sc = row >= 50; e = row - 50 * sc; base = e < 48 ? 32 * (e >> 4) + (e & 15) : 96 + (e - 48); }
    else if (row < 112) { sc = row >= 106; e = row - 100 - 6 * sc; base = e < 4 ? 100 + e : 104 + (e - 4); }
    else { sc = row >= 142; e = row - 112 - 30 * sc; base = e < 16 ? 128 + e : 160 + (e - 16); }
    return base + 16 * sc;
}

struct RsArgs {
    const float *X;                          // [n, IN] the level's input rows (written by cgs_ctx_level_fwd)
    const int64_t *loc;                      // [m] level rows of the rate subset
    const float *W1, *b1, *W2, *b2;          // [100, IN] [100] [175, 100] [175]
    const float *yf, *ys, *yo, *Q;           // [n,50] [n,6] [n,30] [n,3] the level's noisy values and step sizes
    const float *masks;                      // [m, 10] mask weights of the chosen rows (may be NULL = ones)
    const float *x_means;                    // [3] clamp centres (use_clamp)
    int64_t n, m;
    int use_clamp;
    float *sums;                             // forward: [3] accumulated into
    const float *g_sums;                     // backward: [3] upstream gradient of the three sums
    float *side_f, *side_s, *side_o, *side_Q;        // [m,50] [m,6] [m,30] [m,3]
    float *dx_sub;                           // [m, IN]
    float *d_masks;                          // [m, 10] (may be NULL)
    float *dZ2t, *Ht, *dZ1t, *Xt;            // [ntiles][192 | 112 | 112 | XP][16]
};

template <int IN>
struct RsOps {             // the global operands of one 16-row tile, fetched one tile ahead
    f32x4 xb[ClShape<IN>::NTI];
    ClRow y;
    f32x4 q3, mk4, mk5;
};

template <int IN, bool BWD>
__global__ void __launch_bounds__((BWD ? RS_WAVES_B : RS_WAVES_F) * 64) rs_main_kernel(RsArgs a) {
    constexpr int RS_WAVES = BWD ? RS_WAVES_B : RS_WAVES_F;
    constexpr int NTI = ClShape<IN>::NTI, XP = ClShape<IN>::XP, SB = ClShape<IN>::SB;
    __shared__ __attribute__((aligned(16))) float W2n[RS_OP * RS_SA];      // [o'][h]
    __shared__ __attribute__((aligned(16))) float W1n[CL_HP * SB];         // [h][k]
    __shared__ float b1s[CL_HP];
    __shared__ float b2p[RS_OP];
    __shared__ float part[RS_WAVES][3];
    const int tid = threadIdx.x, nthr = RS_WAVES * 64;
    // The weights are read in THEIR order (coalesced 16-byte loads, nothing between a load and the next: all of a thread's loads
    // are in flight together) and scattered into the LDS images; walking the images and gathering cost ~15 us per launch — 43
    // dependent round trips per thread — which is half the kernel on the small levels.
    for (int i = tid; i < (RS_OP * RS_SA + CL_HP * SB) / 4; i += nthr) {
        float *dst = i < RS_OP * RS_SA / 4 ? W2n + 4 * i : W1n + 4 * (i - RS_OP * RS_SA / 4);
        *(f32x4 *)dst = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (int i = tid; i < CL_HP; i += nthr) b1s[i] = i < CL_HID ? a.b1[i] : 0.f;
    for (int i = tid; i < RS_OP; i += nthr) {
        const int row = rs_w2row(i);
        b2p[i] = row >= 0 ? a.b2[row] : 0.f;
    }
    __syncthreads();
    {
        constexpr int N2 = (RS_OUT - 3) * CL_HID / 4;        // the 172 mean / scale rows as float4s (rows are 400 bytes)
        constexpr int PER = (N2 + RS_WAVES * 64 - 1) / (RS_WAVES * 64);
        f32x4 v[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = tid + k * nthr;
            v[k] = i < N2 ? *(const f32x4 *)(a.W2 + 4 * i) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = tid + k * nthr;
            if (i < N2) {
                const int row = (4 * i) / CL_HID, h = (4 * i) % CL_HID;
                *(f32x4 *)(W2n + rs_w2perm(row) * RS_SA + h) = v[k];
            }
        }
        constexpr int N1 = CL_HID * IN;
        constexpr int PER1 = (N1 + RS_WAVES * 64 - 1) / (RS_WAVES * 64);
        float w[PER1];
#pragma unroll
        for (int k = 0; k < PER1; ++k) {
            const int i = tid + k * nthr;
            w[k] = i < N1 ? a.W1[i] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < PER1; ++k) {
            const int i = tid + k * nthr;
            if (i < N1) W1n[(i / IN) * SB + i % IN] = w[k];
        }
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int64_t m = a.m, ntiles = (m + 15) / 16;
    const uint64_t nb = (uint64_t)a.n, mb = (uint64_t)m, tb = (uint64_t)ntiles * 64;
    const ClBuf bX = cl_buf(a.X, nb * IN * 4), bLoc = cl_buf(a.loc, mb * 8), bQ = cl_buf(a.Q, nb * 12),
                bM = cl_buf(a.masks, mb * RS_K * 4);
    const ClRowBufs YB = {cl_buf(a.yf, nb * CL_D * 4), cl_buf(a.ys, nb * CL_S * 4), cl_buf(a.yo, nb * CL_O * 4)};
    const bool has_masks = a.masks != nullptr;
    float xm[3] = {0.f, 0.f, 0.f}, gsum[3] = {0.f, 0.f, 0.f};
    if (a.use_clamp) { xm[0] = a.x_means[0]; xm[1] = a.x_means[1]; xm[2] = a.x_means[2]; }
    if (BWD) { gsum[0] = a.g_sums[0]; gsum[1] = a.g_sums[1]; gsum[2] = a.g_sums[2]; }
    const int use_clamp = a.use_clamp;
    // first mask index of the lane's two offset pieces (offsets 4g.. and 16 + 4g..: three offsets share a weight, :1664)
    const int lo4 = (4 * g) / 3, lo5 = (16 + 4 * g) / 3;

    auto idx_issue = [&](int64_t s) { return cl_li64(bLoc, cl_sel(s < m, (uint32_t)s * 8)); };
    auto op_issue = [&](RsOps<IN> &op, int64_t s, int64_t r) {
        const bool v = s < m;
        cl_xrow_load<IN>(bX, (uint32_t)r * (IN * 4), g, v, op.xb);
        cl_row_issue(op.y, YB, (uint32_t)r, g, v);
        op.q3 = cl_l96(bQ, cl_sel(v, (uint32_t)r * 12));
        op.mk4 = cl_l64(bM, cl_sel(v && has_masks, (uint32_t)s * (RS_K * 4) + (uint32_t)lo4 * 4));
        op.mk5 = cl_l64(bM, cl_sel(v && has_masks, (uint32_t)s * (RS_K * 4) + (uint32_t)lo5 * 4));
    };

    const int64_t tstride = (int64_t)gridDim.x * RS_WAVES;
    int64_t tile = (int64_t)blockIdx.x * RS_WAVES + wave;
#if RS_STAGGER > 0
    if ((wave >> 2) & 1)
        for (int i = 0; i < RS_STAGGER; ++i) __builtin_amdgcn_s_sleep(64);
#endif
    float accb[3] = {0.f, 0.f, 0.f};
    RsOps<IN> opn;
    int64_t rn;
    {
        const int64_t s0 = tile < ntiles ? tile * 16 + c : m;
        const int64_t r0 = idx_issue(s0);
        op_issue(opn, s0, r0);
        rn = idx_issue(s0 + tstride * 16);
    }
    for (; tile < ntiles; tile += tstride) {
        const int64_t s = tile * 16 + c, sn = s + tstride * 16;
        const bool valid = s < m;
        asm volatile("" ::: "memory");   // keep the LDS weight reads inside the tile loop (LICM would spill them)
        RsOps<IN> op = opn;
        cl_xrow_mask<IN>(g, op.xb);
        cl_row_mask(op.y, g);
        CLB_FENCE();
        op_issue(opn, sn, rn);
        rn = idx_issue(sn + tstride * 16);
        CLB_FENCE();

        // ---- H^T = relu(W1 X^T + b1): lane (g, c) ends with hidden units 16t + 4g + {0..3} of row c ----
        // (every MFMA group's LDS operands are read one group ahead, behind scheduling fences: a just-in-time ds_read leaves the
        //  matrix pipe idle for its round trip — with two waves per SIMD nothing else covers it; groups keep >= 2 independent
        //  accumulators between dependent MFMAs)
        f32x4 acc1[CL_NT1];
#pragma unroll
        for (int t = 0; t < CL_NT1; ++t) acc1[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            // groups (q, half): hidden tiles 0..3 / 4..6 of input tile q
            f32x4 wc[4], wn[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) wc[i] = *(const f32x4 *)(W1n + (16 * i + c) * SB + 4 * g);
#pragma unroll
            for (int grp = 0; grp < 2 * NTI; ++grp) {
                const int q = grp >> 1, t0 = (grp & 1) * 4, nt = (grp & 1) ? 3 : 4;
                if (grp + 1 < 2 * NTI) {
                    const int qn = (grp + 1) >> 1, tn = ((grp + 1) & 1) * 4;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (tn + i < CL_NT1) wn[i] = *(const f32x4 *)(W1n + (16 * (tn + i) + c) * SB + 16 * qn + 4 * g);
                }
                CLB_FENCE();
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (i < nt) acc1[t0 + i] = frag_mfma(wc[i][j], op.xb[q][j], acc1[t0 + i]);
                CLB_FENCE();
#pragma unroll
                for (int i = 0; i < 4; ++i) wc[i] = wn[i];
            }
        }
#pragma unroll
        for (int t = 0; t < CL_NT1; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc1[t][r] = fmaxf(acc1[t][r] + b1s[16 * t + 4 * g + r], 0.f);
        uint32_t hpos = 0;
        if (BWD) {
            // operands of the weight gradients, [tile][column][16 rows]: [X | 1] and [H | 1] (rows past the end: zeros)
            const ClBuf bXt = cl_buf(a.Xt, tb * XP * 4), bHt = cl_buf(a.Ht, tb * CL_HP * 4);
#pragma unroll
            for (int q = 0; q < NTI; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = 16 * q + 4 * g + r;
                    const float v = col == IN ? (valid ? 1.f : 0.f) : op.xb[q][r];
                    cl_s32(bXt, (((uint32_t)tile * XP + (uint32_t)col) * 16 + (uint32_t)c) * 4, v);
                }
#pragma unroll
            for (int t = 0; t < CL_NT1; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = 16 * t + 4 * g + r;
                    const float v = valid ? (col == CL_HID ? 1.f : acc1[t][r]) : 0.f;
                    cl_s32(bHt, (((uint32_t)tile * CL_HP + (uint32_t)col) * 16 + (uint32_t)c) * 4, v);
                    hpos |= acc1[t][r] > 0.f ? (1u << (4 * t + r)) : 0u;
                }
        }
        // ---- the 172 mean / scale outputs in the permuted tile order: lane (g, c) ends with o' = 16u + 4g + {0..3} of row c ----
        f32x4 acc2[RS_NT2];
#pragma unroll
        for (int u = 0; u < RS_NT2; ++u) acc2[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            f32x4 wa = *(const f32x4 *)(W2n + c * RS_SA + 4 * g), wb = *(const f32x4 *)(W2n + (16 + c) * RS_SA + 4 * g);
#pragma unroll
            for (int grp = 0; grp < CL_NT1 * (RS_NT2 / 2); ++grp) {
                const int t = grp / (RS_NT2 / 2), u = 2 * (grp % (RS_NT2 / 2));
                f32x4 na = wa, nb = wb;
                if (grp + 1 < CL_NT1 * (RS_NT2 / 2)) {
                    const int tn = (grp + 1) / (RS_NT2 / 2), un = 2 * ((grp + 1) % (RS_NT2 / 2));
                    na = *(const f32x4 *)(W2n + (16 * un + c) * RS_SA + 16 * tn + 4 * g);
                    nb = *(const f32x4 *)(W2n + (16 * (un + 1) + c) * RS_SA + 16 * tn + 4 * g);
                }
                CLB_FENCE();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc2[u] = frag_mfma(wa[r], acc1[t][r], acc2[u]);
                    acc2[u + 1] = frag_mfma(wb[r], acc1[t][r], acc2[u + 1]);
                }
                CLB_FENCE();
                wa = na; wb = nb;
            }
        }
        // ---- the rate terms of the lane's 24 element slots (utils/entropy_models.py:30-50) ----
        ClRow sd;
        float gq[3] = {0.f, 0.f, 0.f};
        float pm[4] = {0.f, 0.f, 0.f, 0.f};          // mask-weight gradients: pieces (offsets 4g.., 16 + 4g..) x (first, second weight)
#pragma unroll
        for (int b = 0; b < RS_NB; ++b) {
            const f32x4 yv = b < 3 ? op.y.F[b] : (b == 3 ? (g == 0 ? op.y.F[3] : op.y.S) : op.y.O[b - 4]);
            f32x4 gxv = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // which slots of this block hold an element (ctx_rows.h ClRow), and of which tensor
                bool slot = true;
                if (b == 3) slot = g == 1 || ((g == 0 || g == 2) && r < 2);
                if (b == 5) slot = g != 3 || r < 2;
                const int kind = b < 3 ? 0 : (b == 3 ? (g == 0 ? 0 : 1) : 2);
                const bool ok = valid && slot;
                const float mean = acc2[2 * b][r] + b2p[32 * b + 4 * g + r];
                const float scale = acc2[2 * b + 1][r] + b2p[32 * b + 16 + 4 * g + r];
                const float q = kind == 0 ? op.q3[0] : (kind == 1 ? op.q3[1] : op.q3[2]);
                const float xmean = kind == 0 ? xm[0] : (kind == 1 ? xm[1] : xm[2]);
                float w = 1.f;
                int wi = 0;
                if (b >= 4) {
                    wi = (b == 4 ? (4 * g + r) / 3 - lo4 : (16 + 4 * g + r) / 3 - lo5);        // 0 or 1
                    const f32x4 mk = b == 4 ? op.mk4 : op.mk5;
                    if (has_masks) w = wi ? mk[1] : mk[0];
                }
                const RateTerms rt = rate_terms(yv[r], mean, scale, q, xmean, use_clamp);
                if (!BWD) {
                    const float bits = rate_bits(rt) * w;
                    if (kind == 0) accb[0] += ok ? bits : 0.f;
                    else if (kind == 1) accb[1] += ok ? bits : 0.f;
                    else accb[2] += ok ? bits : 0.f;
                } else {
                    const float gb = (kind == 0 ? gsum[0] : (kind == 1 ? gsum[1] : gsum[2])) * w;
                    const RateGrads G = rate_grads(rt, scale, gb);
                    acc2[2 * b][r] = ok ? G.gm : 0.f;
                    acc2[2 * b + 1][r] = ok ? G.gs : 0.f;
                    gxv[r] = ok ? G.gx : 0.f;
                    const float gqv = ok ? G.gq : 0.f;
                    if (kind == 0) gq[0] += gqv;
                    else if (kind == 1) gq[1] += gqv;
                    else gq[2] += gqv;
                    if (b >= 4) {            // d (bits * w) / d w = bits (:1664)
                        const float bw = ok ? gsum[2] * rate_bits(rt) : 0.f;
                        const int pi = 2 * (b - 4);
                        pm[pi] += wi ? 0.f : bw;
                        pm[pi + 1] += wi ? bw : 0.f;
                    }
                }
            }
            if (BWD) {
                if (b < 3) sd.F[b] = gxv;
                else if (b == 3) { sd.F[3] = gxv; sd.S = gxv; }
                else sd.O[b - 4] = gxv;
            }
        }
        if (!BWD) continue;

        // ---- compact side arrays: gradients of the noisy values (row s), of the three step sizes, of the mask weights ----
        const ClRowBufs SDB = {cl_buf(a.side_f, mb * CL_D * 4), cl_buf(a.side_s, mb * CL_S * 4), cl_buf(a.side_o, mb * CL_O * 4)};
        cl_row_store(sd, SDB, (uint32_t)s, g, valid);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            gq[k] += __shfl_xor(gq[k], 16);
            gq[k] += __shfl_xor(gq[k], 32);
        }
        cl_s32(cl_buf(a.side_Q, mb * 12), cl_sel(valid && g < 3, (uint32_t)s * 12 + (uint32_t)g * 4), g == 0 ? gq[0] : (g == 1 ? gq[1] : gq[2]));
        if (a.d_masks) {
            // weight k collects the pieces whose (first index + slot) is k, from the row's four lanes, in a fixed order;
            // lane g < 3 forms weights 3g .. 3g + 2, lane 3 weight 9
            float P[4][4];
#pragma unroll
            for (int gg = 0; gg < 4; ++gg)
#pragma unroll
                for (int i = 0; i < 4; ++i) P[gg][i] = __shfl(pm[i], gg * 16 + c, 64);
            float mw[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int k = g < 3 ? 3 * g + j : 9;
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    const int l4 = (4 * gg) / 3, l5 = (16 + 4 * gg) / 3;
                    mw[j] += (l4 == k ? P[gg][0] : 0.f) + (l4 + 1 == k ? P[gg][1] : 0.f);
                    mw[j] += (l5 == k ? P[gg][2] : 0.f) + (l5 + 1 == k ? P[gg][3] : 0.f);
                }
            }
            const ClBuf bDm = cl_buf(a.d_masks, mb * RS_K * 4);
            const uint32_t o = (uint32_t)s * (RS_K * 4) + (uint32_t)g * 12;
            cl_s96(bDm, cl_sel(valid && g < 3, o), (f32x4){mw[0], mw[1], mw[2], 0.f});
            cl_s32(bDm, cl_sel(valid && g == 3, o), mw[0]);
        }
        // ---- dZ2 (permuted order) for the weight gradients; dH^T = W2^T dZ2^T; dZ1; dX = W1^T dZ1 ----
        {
            const ClBuf bZ2 = cl_buf(a.dZ2t, tb * RS_OP * 4);
#pragma unroll
            for (int u = 0; u < RS_NT2; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    cl_s32(bZ2, (((uint32_t)tile * RS_OP + (uint32_t)(16 * u + 4 * g + r)) * 16 + (uint32_t)c) * 4, acc2[u][r]);
        }
        f32x4 adh[CL_NT1];
#pragma unroll
        for (int t = 0; t < CL_NT1; ++t) adh[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            float wc[CL_NT1], wn[CL_NT1];
#pragma unroll
            for (int t = 0; t < CL_NT1; ++t) wc[t] = W2n[(4 * g) * RS_SA + 16 * t + c];
#pragma unroll
            for (int st = 0; st < 4 * RS_NT2; ++st) {
                const int u = st >> 2, j = st & 3;
                if (st + 1 < 4 * RS_NT2) {
                    const int un = (st + 1) >> 2, jn = (st + 1) & 3;
#pragma unroll
                    for (int t = 0; t < CL_NT1; ++t) wn[t] = W2n[(16 * un + 4 * g + jn) * RS_SA + 16 * t + c];
                }
                CLB_FENCE();
#pragma unroll
                for (int t = 0; t < CL_NT1; ++t) adh[t] = frag_mfma(wc[t], acc2[u][j], adh[t]);
                CLB_FENCE();
#pragma unroll
                for (int t = 0; t < CL_NT1; ++t) wc[t] = wn[t];
            }
        }
        {
            const ClBuf bZ1 = cl_buf(a.dZ1t, tb * CL_HP * 4);
#pragma unroll
            for (int t = 0; t < CL_NT1; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    adh[t][r] = (hpos >> (4 * t + r)) & 1u ? adh[t][r] : 0.f;
                    cl_s32(bZ1, (((uint32_t)tile * CL_HP + (uint32_t)(16 * t + 4 * g + r)) * 16 + (uint32_t)c) * 4, adh[t][r]);
                }
        }
        f32x4 adx[NTI];
#pragma unroll
        for (int v = 0; v < NTI; ++v) adx[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            constexpr int NST = 4 * CL_NT1;          // k-steps (t, r): hidden units 16t + 4g' + r (those >= 100 carry zeros)
            float wc[NTI], wn[NTI];
#pragma unroll
            for (int v = 0; v < NTI; ++v) wc[v] = W1n[(4 * g) * SB + 16 * v + c];
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                const int t = st >> 2, r = st & 3;
                if (st + 1 < NST) {
                    const int tn = (st + 1) >> 2, rn = (st + 1) & 3;
#pragma unroll
                    for (int v = 0; v < NTI; ++v) wn[v] = W1n[(16 * tn + 4 * g + rn) * SB + 16 * v + c];
                }
                CLB_FENCE();
#pragma unroll
                for (int v = 0; v < NTI; ++v) adx[v] = frag_mfma(wc[v], adh[t][r], adx[v]);
                CLB_FENCE();
#pragma unroll
                for (int v = 0; v < NTI; ++v) wc[v] = wn[v];
            }
        }
        cl_xrow_store<IN>(cl_buf(a.dx_sub, mb * IN * 4), (uint32_t)s * (IN * 4), g, valid, adx);
    }
    if (!BWD) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float v = accb[k];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
            if (lane == 0) part[wave][k] = v;
        }
        __syncthreads();
        if (tid < 3) {
            float v = 0.f;
            for (int w = 0; w < RS_WAVES; ++w) v += part[w][tid];
            atomicAdd(&a.sums[tid], v);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Weight gradients from the column-major tiles.  out[m][n] = sum over rows A[row][m] B[row][n]: lane (g, c) feeds the MFMA of
// k-step r with A[row 4g + r][16 i + c] and B[row 4g + r][16 j + c] — the four rows 4g..4g+3 of a column are the 16 contiguous
// bytes one b128 load returns, a wave's load is 1 KB contiguous.
#define RW_WAVES 8
struct RwArgs {
    const float *dZ2t, *Ht, *dZ1t, *Xt;
    int64_t ntiles, tiles_per_block;
    float *partial;                          // [gridDim.x][E], E = 100 IN + 100 + 175 * 100 + 175
};

template <int NA, int NB>
__device__ __forceinline__ void rw_accumulate(ClBuf A, int colsA, int a0, ClBuf B, int colsB, int64_t t0, int64_t t1, int g, int c,
                                              f32x4 (&acc)[NA][NB]) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // operands TWO tiles ahead (a tile is ~0.75 us of MFMAs, a load from HBM / the other XCDs' L2 takes longer under load)
    f32x4 an[2][NA], bn[2][NB];
    auto issue = [&](int64_t tile, f32x4 (&pa)[NA], f32x4 (&pb)[NB]) {
        const bool on = tile < t1;
#pragma unroll
        for (int i = 0; i < NA; ++i)
            pa[i] = cl_l128(A, cl_sel(on, (((uint32_t)tile * (uint32_t)colsA + (uint32_t)(16 * (a0 + i) + c)) * 16 + 4 * (uint32_t)g) * 4));
#pragma unroll
        for (int j = 0; j < NB; ++j)
            pb[j] = cl_l128(B, cl_sel(on, (((uint32_t)tile * (uint32_t)colsB + (uint32_t)(16 * j + c)) * 16 + 4 * (uint32_t)g) * 4));
    };
    auto products = [&](const f32x4 (&av)[NA], const f32x4 (&bv)[NB]) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[i][j] = frag_mfma(av[i][r], bv[j][r], acc[i][j]);
    };
    issue(t0, an[0], bn[0]);
    issue(t0 + 1, an[1], bn[1]);
    for (int64_t tile = t0; tile < t1; tile += 2) {          // (two tiles per trip: the two register stages keep their names)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            f32x4 av[NA], bv[NB];
#pragma unroll
            for (int i = 0; i < NA; ++i) av[i] = an[st][i];
#pragma unroll
            for (int j = 0; j < NB; ++j) bv[j] = bn[st][j];
            CLB_FENCE();
            issue(tile + st + 2, an[st], bn[st]);
            CLB_FENCE();
            products(av, bv);              // (a tile past the end was loaded as zeros: adds nothing)
        }
    }
}

template <int IN>
__global__ void __launch_bounds__(RW_WAVES * 64) rs_wgrad_kernel(RwArgs a) {
    constexpr int NTI = ClShape<IN>::NTI, XP = ClShape<IN>::XP;
    constexpr int E1 = CL_HID * IN + CL_HID, E = E1 + RS_OUT * CL_HID + RS_OUT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int64_t t0 = (int64_t)blockIdx.x * a.tiles_per_block;
    const int64_t t1 = t0 + a.tiles_per_block < a.ntiles ? t0 + a.tiles_per_block : a.ntiles;
    const uint64_t tb = (uint64_t)a.ntiles * 64;
    float *img = a.partial + (int64_t)blockIdx.x * E;
    float *img1 = img, *imgb1 = img + CL_HID * IN, *img2 = img + E1, *imgb2 = img2 + RS_OUT * CL_HID;
    if (wave < 6) {
        // dW2[o'][h] for the permuted tiles u = 2 wave, 2 wave + 1 (one element block: its means and its scales)
        f32x4 acc[2][CL_NT1];
        rw_accumulate<2, CL_NT1>(cl_buf(a.dZ2t, tb * RS_OP * 4), RS_OP, 2 * wave, cl_buf(a.Ht, tb * CL_HP * 4), CL_HP, t0, t1, g, c, acc);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rs_w2row(16 * (2 * wave + i) + 4 * g + r);
                if (row < 0) continue;
#pragma unroll
                for (int j = 0; j < CL_NT1; ++j) {
                    const int h = 16 * j + c;
                    if (h < CL_HID) img2[row * CL_HID + h] = acc[i][j][r];
                    else if (h == CL_HID) imgb2[row] = acc[i][j][r];
                }
            }
        if (wave == 0) {        // the three step-size rows belong to cgs_ctx_level_bwd: zeros here
            for (int i = lane; i < 3 * CL_HID; i += 64) img2[(RS_OUT - 3) * CL_HID + i] = 0.f;
            if (lane < 3) imgb2[RS_OUT - 3 + lane] = 0.f;
        }
    } else {
        // dW1[hid][k] for hidden tiles 0..3 (wave 6) / 4..6 (wave 7)
        const ClBuf bA = cl_buf(a.dZ1t, tb * CL_HP * 4), bB = cl_buf(a.Xt, tb * XP * 4);
        auto dump = [&](int t, const f32x4 (&row)[NTI]) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int hid = 16 * t + 4 * g + r;
                if (hid >= CL_HID) continue;
#pragma unroll
                for (int j = 0; j < NTI; ++j) {
                    const int k = 16 * j + c;
                    if (k < IN) img1[hid * IN + k] = row[j][r];
                    else if (k == IN) imgb1[hid] = row[j][r];
                }
            }
        };
        if (wave == 6) {
            f32x4 acc[4][NTI];
            rw_accumulate<4, NTI>(bA, CL_HP, 0, bB, XP, t0, t1, g, c, acc);
#pragma unroll
            for (int i = 0; i < 4; ++i) dump(i, acc[i]);
        } else {
            f32x4 acc[3][NTI];
            rw_accumulate<3, NTI>(bA, CL_HP, 4, bB, XP, t0, t1, g, c, acc);
#pragma unroll
            for (int i = 0; i < 3; ++i) dump(4 + i, acc[i]);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
static int rs_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

static bool rs_fits(int64_t rows, int64_t row_bytes) { return rows >= 0 && (uint64_t)rows * (uint64_t)row_bytes < CL_MAX_BYTES; }

static int rs_check(const char *what, int in_dim, const float *X, int64_t n, const int64_t *loc, int64_t m, const float *W1,
                    const float *b1, const float *W2, const float *b2, const float *yf, const float *ys, const float *yo,
                    const float *Q, const float *x_means, int use_clamp) {
    if (n < 0 || m < 0 || (in_dim != 71 && in_dim != 15)) { cgs_set_error("%s: bad args (in_dim 71 or 15)", what); return CGS_ERR_ARG; }
    if (m == 0) return CGS_OK;
    if (!X || !loc || !W1 || !b1 || !W2 || !b2 || !yf || !ys || !yo || !Q || (use_clamp && !x_means)) { cgs_set_error("%s: NULL", what); return CGS_ERR_ARG; }
    const int64_t ntiles = (m + 15) / 16;
    if (!rs_fits(n, in_dim * 4) || !rs_fits(n, CL_D * 4) || !rs_fits(m, CL_D * 4) || !rs_fits(ntiles * 16, RS_OP * 4)) {
        cgs_set_error("%s: an operand exceeds 4 GB (n %lld, m %lld)", what, (long long)n, (long long)m);
        return CGS_ERR_ARG;
    }
    return CGS_OK;
}

static int64_t rs_grid(int64_t m, int waves) {
    const int64_t tiles = (m + 15) / 16, want = (tiles + waves - 1) / waves;
    return want < rs_cus() ? want : rs_cus();
}

// Bits of the rate subset of one level (scene/gaussian_model.py:1658-1669 on rows loc[0..m) of the level): sums3 [3] +=
// (feat, scaling, offsets * mask weight) bits.  X [n, in_dim]: the level's input rows as cgs_ctx_level_fwd wrote them; W1 / b1 /
// W2 / b2: mlp_grid[level] ([100, in_dim], [100], [175, 100], [175]); yf / ys / yo / Q: the level's noisy values and step sizes
// [n, .]; masks [m, 10]: the mask weights of the chosen rows (NULL = ones); x_means [3]: clamp centres when use_clamp.
extern "C" int cgs_rate_sub_fwd(int in_dim, const float *X, int64_t n, const int64_t *loc, int64_t m, const float *W1,
                                const float *b1, const float *W2, const float *b2, const float *yf, const float *ys,
                                const float *yo, const float *Q, const float *masks, const float *x_means, int use_clamp,
                                float *sums3, void *stream) {
    int rc = rs_check("rate_sub_fwd", in_dim, X, n, loc, m, W1, b1, W2, b2, yf, ys, yo, Q, x_means, use_clamp);
    if (rc || m == 0) return rc;
    if (!sums3) { cgs_set_error("rate_sub_fwd: NULL"); return CGS_ERR_ARG; }
    RsArgs a = {};
    a.X = X; a.loc = loc; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.yf = yf; a.ys = ys; a.yo = yo; a.Q = Q; a.masks = masks;
    a.x_means = use_clamp ? x_means : nullptr; a.n = n; a.m = m; a.use_clamp = use_clamp ? 1 : 0; a.sums = sums3;
    const unsigned grid = (unsigned)rs_grid(m, RS_WAVES_F);
    CgsProfScope prof(CGS_PROF_RATE_FWD, (hipStream_t)stream);
    if (in_dim == 71) hipLaunchKernelGGL((rs_main_kernel<71, false>), dim3(grid), dim3(RS_WAVES_F * 64), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((rs_main_kernel<15, false>), dim3(grid), dim3(RS_WAVES_F * 64), 0, (hipStream_t)stream, a);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

static int64_t rw_blocks(int64_t ntiles) {
    // a workgroup writes (and the reduction reads) a ~100 KB image whatever its rows: at least 8 tiles each
    int64_t blocks = (ntiles + 7) / 8;
    if (blocks > rs_cus()) blocks = rs_cus();
    return blocks < 1 ? 1 : blocks;
}

extern "C" size_t cgs_rate_sub_bwd_scratch_bytes(int in_dim, int64_t m) {
    if (m <= 0 || (in_dim != 71 && in_dim != 15)) return 0;
    const int64_t ntiles = (m + 15) / 16, xp = in_dim == 71 ? 80 : 16;
    const int64_t E = (int64_t)CL_HID * in_dim + CL_HID + RS_OUT * CL_HID + RS_OUT;
    return (size_t)(ntiles * 16 * (RS_OP + 2 * CL_HP + xp) + rw_blocks(ntiles) * E) * sizeof(float);
}

// Backward of cgs_rate_sub_fwd.  g_sums3 [3] (device): upstream gradient of the three sums.  Writes the compact side arrays
// side_f [m,50] / side_s [m,6] / side_o [m,30] / side_Q [m,3] (row s = gradient of the noisy values / step sizes of level row
// loc[s]: the side arrays of cgs_ctx_level_bwd), dx_sub [m, in_dim] (gradient of the input row through the mean / scale branch),
// d_masks [m,10] (may be NULL; every row written), and ASSIGNS dW1 [100, in_dim], db1 [100], dW2 [175, 100], db2 [175] (the
// three step-size rows of dW2 / db2 are zeros: cgs_ctx_level_bwd accumulates them).  scratch >= cgs_rate_sub_bwd_scratch_bytes().
extern "C" int cgs_rate_sub_bwd(int in_dim, const float *X, int64_t n, const int64_t *loc, int64_t m, const float *W1,
                                const float *b1, const float *W2, const float *b2, const float *yf, const float *ys,
                                const float *yo, const float *Q, const float *masks, const float *x_means, int use_clamp,
                                const float *g_sums3, float *side_f, float *side_s, float *side_o, float *side_Q, float *dx_sub,
                                float *d_masks, float *dW1, float *db1, float *dW2, float *db2, void *scratch, size_t scratch_bytes,
                                void *stream) {
    int rc = rs_check("rate_sub_bwd", in_dim, X, n, loc, m, W1, b1, W2, b2, yf, ys, yo, Q, x_means, use_clamp);
    if (rc) return rc;
    if (m == 0) { cgs_set_error("rate_sub_bwd: m == 0 (the caller zero-fills the weight gradients of an empty subset)"); return CGS_ERR_ARG; }
    if (!g_sums3 || !side_f || !side_s || !side_o || !side_Q || !dx_sub || !dW1 || !db1 || !dW2 || !db2 || !scratch) {
        cgs_set_error("rate_sub_bwd: NULL");
        return CGS_ERR_ARG;
    }
    if (scratch_bytes < cgs_rate_sub_bwd_scratch_bytes(in_dim, m)) { cgs_set_error("rate_sub_bwd: scratch too small"); return CGS_ERR_WORKSPACE; }
    const int64_t ntiles = (m + 15) / 16, xp = in_dim == 71 ? 80 : 16;
    float *ws = (float *)scratch;
    RsArgs a = {};
    a.X = X; a.loc = loc; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.yf = yf; a.ys = ys; a.yo = yo; a.Q = Q; a.masks = masks;
    a.x_means = use_clamp ? x_means : nullptr; a.n = n; a.m = m; a.use_clamp = use_clamp ? 1 : 0; a.g_sums = g_sums3;
    a.side_f = side_f; a.side_s = side_s; a.side_o = side_o; a.side_Q = side_Q; a.dx_sub = dx_sub; a.d_masks = masks ? d_masks : nullptr;
    a.dZ2t = ws; a.Ht = a.dZ2t + ntiles * 16 * RS_OP; a.dZ1t = a.Ht + ntiles * 16 * CL_HP; a.Xt = a.dZ1t + ntiles * 16 * CL_HP;
    float *partial = a.Xt + ntiles * 16 * xp;
    const unsigned grid = (unsigned)rs_grid(m, RS_WAVES_B);
    {
        CgsProfScope prof(CGS_PROF_RATE_BWD, (hipStream_t)stream);
        if (in_dim == 71) hipLaunchKernelGGL((rs_main_kernel<71, true>), dim3(grid), dim3(RS_WAVES_B * 64), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((rs_main_kernel<15, true>), dim3(grid), dim3(RS_WAVES_B * 64), 0, (hipStream_t)stream, a);
        CGS_CHECK_HIP(hipGetLastError());
    }
    CgsProfScope prof(CGS_PROF_LMLP_WGRAD, (hipStream_t)stream);
    RwArgs w;
    w.dZ2t = a.dZ2t; w.Ht = a.Ht; w.dZ1t = a.dZ1t; w.Xt = a.Xt; w.ntiles = ntiles; w.partial = partial;
    const int64_t blocks = rw_blocks(ntiles);
    w.tiles_per_block = (ntiles + blocks - 1) / blocks;
    const unsigned wgrid = (unsigned)((ntiles + w.tiles_per_block - 1) / w.tiles_per_block);
    if (in_dim == 71) hipLaunchKernelGGL((rs_wgrad_kernel<71>), dim3(wgrid), dim3(RW_WAVES * 64), 0, (hipStream_t)stream, w);
    else hipLaunchKernelGGL((rs_wgrad_kernel<15>), dim3(wgrid), dim3(RW_WAVES * 64), 0, (hipStream_t)stream, w);
    CGS_CHECK_HIP(hipGetLastError());
    const CgsWgProduct prods[2] = {{nullptr, 0, CL_HID, nullptr, 0, in_dim, dW1, db1}, {nullptr, 0, RS_OUT, nullptr, 0, CL_HID, dW2, db2}};
    return cgs_launch_wgrad_reduce_assign(partial, (int)wgrid, prods, 2, (hipStream_t)stream);
}
