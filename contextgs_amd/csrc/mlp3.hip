// The three anchor MLPs of generate_neural_gaussians (gaussian_renderer/__init__.py:112,122,126:
// mlp_opacity 54->50->10 tanh, mlp_color 54->50->30 sigmoid, mlp_cov 54->50->70) share their
// input row, so they run as ONE fused fp32-MFMA launch forward and one backward: the X
// fragments are loaded once, the three hidden activations land in one [n,150] buffer (which
// makes the three first-layer weight gradients a single [150 x 54] contraction), and the three
// dX contributions are summed in the accumulator registers.  Same transposed-chaining design
// as mlp.hip (weights = A operand from LDS, activations = B operand in registers, one 16-byte
// access per lane for every activation read/write — mlp_frag.h).
#include "cgs_internal.h"
#include "mlp_frag.h"
#include "buf_access.h"

#define M3_IN 54
#define M3_HID 50
#define M3_NTI 4             // ceil(54/16)
#define M3_XP 64
#define M3_NT1 4             // ceil(50/16)
#define M3_HP 64
#define M3_HCAT 150          // valid hidden columns of the three heads
// Row layouts of what the forward and the backward hand over in memory.  A wave instruction moves a 64-byte chunk (16
// features) per row, and a chunk that straddles two 64-byte sectors is two write requests (tools/micro/bf16_split.hip:
// +46 % on a store-heavy MFMA kernel at EQUAL bytes).  Padding a row to whole sectors costs bytes, though, and the two were
// measured against each other per buffer on one box (tools/ab_m3_layout.sh, profiles/r03_row_alignment.txt):
//   Hcat   [n][150], X_out [n][54]  packed: the forward is bound by the bytes it stores (padded: +5 %)
//   dZ1cat [n][3][64]               padded (a head's 50 columns on a 256-byte boundary, 768-byte rows, pad columns zero):
//                                   the backward gains 6 % from it
#define M3_HLD 150           // Hcat row stride, heads 50 columns apart
#define M3_HPITCH 50
#define M3_GLD 192           // dZ1cat row stride, heads 64 columns apart (dW1cat [192, 54] / db1cat [192] use the same rows)
#define M3_GPITCH 64
#define M3_XLD 54
// wave-tile shapes (rows per wave tile = 16 RT, waves per workgroup); overridable for tools/mlp_tiling.sh
#ifndef M3_FWD_RT
#define M3_FWD_RT 1
#define M3_FWD_WAVES 16
#ifndef M3_FWD_LDS_STORE
#define M3_FWD_LDS_STORE 1     // the forward's row-major outputs (Y_*, X_out) leave through a per-wave LDS patch as contiguous 1 KB stores
#endif
#ifndef M3W_ROUTER
#define M3W_ROUTER 1          // fused backward: weight-gradient MFMAs of a group interleaved over its accumulators
#endif
#ifndef M3_TAILMAP
#define M3_TAILMAP 1           // ROWS forward: inputs 48..53 in tail order (two MFMA k-steps instead of four per hidden tile)
#endif
#define M3_FWD_PATCH (16 * 70)  // floats per wave: the widest tile image (Y_cov)
#endif
#ifndef M3_BWD_RT
#define M3_BWD_RT 2
#define M3_BWD_WAVES 8
#endif

// M3_FUSED_WGRAD: the backward computes the weight gradients in the same launch (mlp3_bwd_wg_kernel); 0 = the round-3 pair
// mlp3_bwd_kernel + wgrad_multi (kept for the data-only entry of the deferred weight-gradient mode and for A/B builds)
#ifndef M3_FUSED_WGRAD
#define M3_FUSED_WGRAD 1
#endif
// M3_PIPE: explicit register double-buffering of the LDS weight fragments in the forward kernel
#ifndef M3_PIPE
#define M3_PIPE 0
#endif
#define M3_FENCE()                         \
    do {                                   \
        asm volatile("" ::: "memory");     \
        __builtin_amdgcn_sched_barrier(0); \
    } while (0)

struct M3Head {
    const float *W1, *b1, *W2, *b2;
    float *Y;            // forward output [n, OUT] (backward: saved forward output, read-only use)
    const float *dY;     // backward input  [n, OUT]
    float *dZ2;          // backward scratch [n, OUT] (only for heads with an activation)
};

// ---- LDS images ------------------------------------------------------------------------------
template <int OUT>
struct M3FwdLds {
    static constexpr int NT2 = (OUT + 15) / 16, OP = NT2 * 16;
    static constexpr int S1 = frag_pad4mod8(M3_HP), S2 = frag_pad4mod8(OP);
    static constexpr int FLOATS = M3_XP * S1 + M3_HP * S2 + M3_HP + OP;
};

template <int OUT>
__device__ __forceinline__ void m3_stage_fwd(float *lds, const M3Head &h, int tid, int nthr) {
    using L = M3FwdLds<OUT>;
    float *W1s = lds, *W2s = W1s + M3_XP * L::S1, *b1s = W2s + M3_HP * L::S2, *b2s = b1s + M3_HP;
    frag_stage_transposed<M3_IN, M3_HID, M3_XP, L::S1>(W1s, h.W1, tid, nthr);      // W1s[k][j] = W1[j][k]
    frag_stage_transposed<M3_HID, OUT, M3_HP, L::S2>(W2s, h.W2, tid, nthr);        // W2s[h][o] = W2[o][h]
    for (int i = tid; i < M3_HP; i += nthr) b1s[i] = i < M3_HID ? h.b1[i] : 0.f;
    for (int i = tid; i < L::OP; i += nthr) b2s[i] = i < OUT ? h.b2[i] : 0.f;
}

// (round 5) the forward's stores and loads go through raw buffer instructions (buf_access.h): the predicated global accesses
// of rounds 1-4 — 59 of them behind 88 branches — had a `s_waitcnt vmcnt(0)` at 25 exec-mask joins per tile; the tile loop now
// waits with partial counts only.  Measured on one box: 517 us either way at 1 M anchors (profiles/r05_mlp3_fwd_buffer_ab.txt) — with 16 waves
// per CU the other waves covered those drains; the kernel stays bound by its 1.26 KB of stores per anchor in 64-byte chunks
// that straddle sectors (profiles/r03_row_alignment.txt).  Kept for the bounds-checked tails and the shared helpers.
#define M3_MAX_ROWS (4ll << 20)          // x 600-byte Hcat rows = 2.5 GB: every operand of a launch stays below CL_MAX_BYTES
struct M3FwdBufs { ClBuf H, Y[3], Xo, X, src, feat, anc; };

template <int OUT, int ACT, int RT, bool TILED, bool TM>
__device__ __forceinline__ void m3_head_fwd(const float *lds, const M3Head &h, int head, const f32x4 (&xb)[RT][M3_NTI],
                                            const bool (&valid)[RT], int64_t row0, int g, int c,
                                            float *__restrict__ Hcat, const M3FwdBufs &B, float *patch) {
    using L = M3FwdLds<OUT>;
    const float *W1s = lds, *W2s = W1s + M3_XP * L::S1, *b1s = W2s + M3_HP * L::S2, *b2s = b1s + M3_HP;
    f32x4 acc1[M3_NT1][RT];
#pragma unroll
    for (int t = 0; t < M3_NT1; ++t)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc1[t][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#if M3_PIPE
    // A fragments of k-group q + 1 are read from LDS before the 16 MFMAs of group q are issued (two register
    // buffers): the compiler's own schedule reads two values, waits for them and issues two MFMAs, i.e. an exposed LDS
    // round trip per pair (profiles/r03_mlp_pipeline.txt)
    float a1[2][4][M3_NT1];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < M3_NT1; ++t) a1[0][j][t] = W1s[(4 * g + j) * L::S1 + 16 * t + c];
    // input row of W1 for (q, j): 16 q + 4 g + j — in the last piece of a TM operand 48 + g + 4 j (m3_load_x_rows_b)
    auto krow = [&](int q, int j) { return (TM && q == M3_NTI - 1) ? 16 * q + g + 4 * j : 16 * q + 4 * g + j; };
#pragma unroll
    for (int q = 0; q < M3_NTI; ++q) {
        if (q + 1 < M3_NTI) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < M3_NT1; ++t)
                    a1[(q + 1) & 1][j][t] = W1s[krow(q + 1, j) * L::S1 + 16 * t + c];
        }
        M3_FENCE();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (16 * q + j >= M3_IN) continue;
            if (TM && q == M3_NTI - 1 && 16 * q + 4 * j >= M3_IN) continue;          // tail order: inputs 48 + g + 4 j
#pragma unroll
            for (int t = 0; t < M3_NT1; ++t)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc1[t][rt] = frag_mfma(a1[q & 1][j][t], xb[rt][q][j], acc1[t][rt]);
        }
        M3_FENCE();
    }
#else
#pragma unroll
    for (int q = 0; q < M3_NTI; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (16 * q + j >= M3_IN) continue;
            if (TM && q == M3_NTI - 1 && 16 * q + 4 * j >= M3_IN) continue;
#pragma unroll
            for (int t = 0; t < M3_NT1; ++t) {
                const float a = W1s[((TM && q == M3_NTI - 1) ? 16 * q + g + 4 * j : 16 * q + 4 * g + j) * L::S1 + 16 * t + c];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc1[t][rt] = frag_mfma(a, xb[rt][q][j], acc1[t][rt]);
            }
        }
#endif
#pragma unroll
    for (int t = 0; t < M3_NT1; ++t)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int64_t row = row0 + rt * 16 + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc1[t][rt][r] = fmaxf(acc1[t][rt][r] + b1s[16 * t + 4 * g + r], 0.f);
            if (TILED)
                frag_tstore4<M3_HID>(B.H, (uint32_t)((row0 + rt * 16) >> 4) * (16 * M3_HLD * 4) + (uint32_t)head * (16 * M3_HID * 4), t, g, c,
                                     valid[rt], acc1[t][rt]);
            else
                frag_bstore4<M3_HID>(B.H, (uint32_t)row * (M3_HLD * 4) + (uint32_t)(M3_HPITCH * head) * 4, t, g, valid[rt], acc1[t][rt]);
        }
    f32x4 acc2[L::NT2][RT];
#pragma unroll
    for (int u = 0; u < L::NT2; ++u)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc2[u][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#if M3_PIPE
    float a2[2][4][L::NT2];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int u = 0; u < L::NT2; ++u) a2[0][r][u] = W2s[(4 * g + r) * L::S2 + 16 * u + c];
#pragma unroll
    for (int t = 0; t < M3_NT1; ++t) {
        if (t + 1 < M3_NT1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (16 * (t + 1) + r >= M3_HID) continue;
#pragma unroll
                for (int u = 0; u < L::NT2; ++u) a2[(t + 1) & 1][r][u] = W2s[(16 * (t + 1) + 4 * g + r) * L::S2 + 16 * u + c];
            }
        }
        M3_FENCE();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (16 * t + r >= M3_HID) continue;
#pragma unroll
            for (int u = 0; u < L::NT2; ++u)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc2[u][rt] = frag_mfma(a2[t & 1][r][u], acc1[t][rt][r], acc2[u][rt]);
        }
        M3_FENCE();
    }
#else
#pragma unroll
    for (int t = 0; t < M3_NT1; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (16 * t + r >= M3_HID) continue;
#pragma unroll
            for (int u = 0; u < L::NT2; ++u) {
                const float a = W2s[(16 * t + 4 * g + r) * L::S2 + 16 * u + c];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc2[u][rt] = frag_mfma(a, acc1[t][rt][r], acc2[u][rt]);
            }
        }
#endif
#if M3_FWD_LDS_STORE
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        f32x4 yv[L::NT2];
#pragma unroll
        for (int u = 0; u < L::NT2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) yv[u][r] = frag_act<ACT>(acc2[u][rt][r] + b2s[16 * u + 4 * g + r]);
        frag_tile_store<OUT, L::NT2>(patch, B.Y[head], (uint32_t)(row0 + rt * 16) * (OUT * 4), yv, g, c, 16 * g + c);
    }
#else
#pragma unroll
    for (int u = 0; u < L::NT2; ++u)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int64_t row = row0 + rt * 16 + c;
            f32x4 y;
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = frag_act<ACT>(acc2[u][rt][r] + b2s[16 * u + 4 * g + r]);
            frag_bstore4<OUT>(B.Y[head], (uint32_t)row * (OUT * 4), u, g, valid[rt], y);
        }
#endif
}

// Input row assembled on the fly (ROWS variant): [ feat_src[src_row[r], 0:50] | (a - cam)/|a - cam| | |a - cam| ] with a =
// anchor[r] — gaussian_renderer/__init__.py:106-110 fused into the MLP's operand load, so the [n,54] input is written
// once (for the weight-gradient kernel) instead of written by a gather kernel and read back here.
struct M3Rows {
    const float *feat_src;      // [*, 50] (the context model's output in coding order)
    const int64_t *src_row;     // [n]
    const float *anchor;        // [n, 3] visible anchors
    const float *cam;           // [3] on the device
    float *X_out;               // [n, 54] side output
    float *d_feat_src;          // backward: rows src_row[r] of this [*, 50] buffer receive dX[:, 0:50]
    float *d_anchor;            // backward: [n, 3]
};

// through buffer loads: `srow` = src_row[row] (fetched a tile earlier by the caller), no branch anywhere
// TM (the forward's own operand): the last piece in TAIL ORDER — lane g, slot j holds input 48 + g + 4 j instead of 48 + 4 g + j, so
// that the six inputs 48..53 fill the k-slots of TWO MFMA steps (j = 0: 48..51, j = 1: 52, 53) instead of being spread over four;
// the matching rows of W1 are read in m3_head_fwd.  Free here: the tail is assembled in registers.
template <bool TM = false>
__device__ __forceinline__ f32x4 m3_load_x_rows_b(const M3FwdBufs &B, float cam0, float cam1, float cam2, int64_t row, int64_t srow,
                                                  int q, int g, bool valid) {
    const uint32_t fo = (uint32_t)srow * (M3_HID * 4);
    if (q < 3) return cl_l128(B.feat, cl_sel(valid, fo + (uint32_t)(16 * q + 4 * g) * 4));
    if (TM) {
        const f32x4 f2 = cl_l64(B.feat, cl_sel(valid && g < 2, fo + 48 * 4));                 // features 48, 49
        const f32x4 a = cl_l96(B.anc, cl_sel(valid, (uint32_t)row * 12));
        const float ux = a[0] - cam0, uy = a[1] - cam1, uz = a[2] - cam2;
        const float dist = sqrtf(ux * ux + uy * uy + uz * uz);
        const float s0 = g == 0 ? f2[0] : (g == 1 ? f2[1] : (g == 2 ? ux / dist : uy / dist));
        const float s1 = g == 0 ? uz / dist : (g == 1 ? dist : 0.f);
        return valid ? (f32x4){s0, s1, 0.f, 0.f} : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const f32x4 f2 = cl_l64(B.feat, cl_sel(valid && g == 0, fo + 48 * 4));                    // features 48, 49
    const f32x4 a = cl_l96(B.anc, cl_sel(valid && g < 2, (uint32_t)row * 12));
    const float ux = a[0] - cam0, uy = a[1] - cam1, uz = a[2] - cam2;
    const float dist = sqrtf(ux * ux + uy * uy + uz * uz);
    const f32x4 v0 = (f32x4){f2[0], f2[1], ux / dist, uy / dist}, v1 = (f32x4){uz / dist, dist, 0.f, 0.f};
    const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
    return !valid ? z : (g == 0 ? v0 : (g == 1 ? v1 : z));
}

// TILED (ROWS only): Hcat — a buffer only the fused backward reads, 600 of the 1256 bytes per anchor this kernel stores — leaves
// in fragment-major form (buf_access.h frag_tstore4): every store instruction one contiguous KB instead of sixteen 64-byte chunks
// 600 bytes apart (the kernel is bound by its stores, and the L2 has to merge those chunks into lines: with non-temporal stores,
// i.e. without the merging, it takes 1.41 ms instead of 0.51).  456 -> 421 us at 1 M anchors; X_out the same way gave the forward
// another 12 us and cost the one-wave backward 25 (profiles/r06_mlp3_tiled.txt): X_out stays row-major.
template <int O0, int A0, int O1, int A1, int O2, int A2, int RT, int WAVES, bool ROWS, bool TILED = false>
__global__ void __launch_bounds__(WAVES * 64)
    mlp3_fwd_kernel(const float *__restrict__ X, int64_t ldx, M3Head h0, M3Head h1, M3Head h2,
                    float *__restrict__ Hcat, int64_t n, M3Rows R) {
    constexpr int WFL = M3FwdLds<O0>::FLOATS + M3FwdLds<O1>::FLOATS + M3FwdLds<O2>::FLOATS;
    static_assert(WFL % 4 == 0 && O0 <= 70 && O1 <= 70 && O2 <= 70 && M3_IN <= 70, "patch: 16-byte aligned, the widest image fits");
    __shared__ __attribute__((aligned(16))) float lds[WFL + (M3_FWD_LDS_STORE ? WAVES * M3_FWD_PATCH : 0)];
    float *l0 = lds, *l1 = l0 + M3FwdLds<O0>::FLOATS, *l2 = l1 + M3FwdLds<O1>::FLOATS;
    const int tid = threadIdx.x, nthr = WAVES * 64;
    float *patch = lds + WFL + (M3_FWD_LDS_STORE ? (tid >> 6) * M3_FWD_PATCH : 0);
    m3_stage_fwd<O0>(l0, h0, tid, nthr);
    m3_stage_fwd<O1>(l1, h1, tid, nthr);
    m3_stage_fwd<O2>(l2, h2, tid, nthr);
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int64_t ntiles = (n + 16 * RT - 1) / (16 * RT);
    // X of the next tile is fetched while this tile is in the matrix pipe (double-buffered in registers)
    f32x4 xb[RT][M3_NTI], xn[RT][M3_NTI];
    bool valid[RT], validn[RT];
    const int64_t tile0 = (int64_t)blockIdx.x * WAVES + wave, tstride = (int64_t)gridDim.x * WAVES;
    M3FwdBufs B;
    {
        const uint64_t nb = (uint64_t)n;
        const uint64_t nbt = (nb + 15) / 16 * 16;
        B.H = cl_buf(Hcat, (TILED ? nbt : nb) * (M3_HLD * 4));
        B.Y[0] = cl_buf(h0.Y, nb * (O0 * 4)); B.Y[1] = cl_buf(h1.Y, nb * (O1 * 4)); B.Y[2] = cl_buf(h2.Y, nb * (O2 * 4));
        B.Xo = cl_buf(ROWS ? R.X_out : nullptr, nb * (M3_XLD * 4));
        B.X = cl_buf(ROWS ? nullptr : X, nb > 0 ? ((nb - 1) * (uint64_t)ldx + M3_IN) * 4 : 0);
        B.src = cl_buf(ROWS ? R.src_row : nullptr, nb * 8);
        B.feat = cl_buf(ROWS ? R.feat_src : nullptr, CL_MAX_BYTES);       // (its row count is not an argument: bounded by src_row)
        B.anc = cl_buf(ROWS ? R.anchor : nullptr, nb * 12);
    }
    const float cam0 = ROWS ? R.cam[0] : 0.f, cam1 = ROWS ? R.cam[1] : 0.f, cam2 = ROWS ? R.cam[2] : 0.f;
    const uint32_t ldx4 = (uint32_t)ldx * 4;
    // src_row of the tile AFTER the next one is in flight while the next tile's rows are gathered through src_row of the next
    int64_t srow_n[RT], srow_nn[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int64_t row = tile0 * 16 * RT + rt * 16 + c, rown = row + tstride * 16 * RT;
        valid[rt] = tile0 < ntiles && row < n;
        const int64_t srow = ROWS ? cl_li64(B.src, cl_sel(valid[rt], (uint32_t)row * 8)) : 0;
        srow_n[rt] = ROWS ? cl_li64(B.src, cl_sel(rown < n, (uint32_t)rown * 8)) : 0;
#pragma unroll
        for (int q = 0; q < M3_NTI; ++q)
            xb[rt][q] = ROWS ? m3_load_x_rows_b<M3_TAILMAP>(B, cam0, cam1, cam2, row, srow, q, g, valid[rt])
                             : frag_bmask4<M3_IN>(frag_bload4<M3_IN>(B.X, (uint32_t)row * ldx4, q, g, valid[rt]), q, g);
    }
    for (int64_t tile = tile0; tile < ntiles; tile += tstride) {
        const int64_t row0 = tile * 16 * RT;
        asm volatile("" ::: "memory");   // keep the LDS weight reads inside the tile loop (LICM would spill them)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int64_t row = (tile + tstride) * 16 * RT + rt * 16 + c, rownn = row + tstride * 16 * RT;
            validn[rt] = row < n;
            srow_nn[rt] = ROWS ? cl_li64(B.src, cl_sel(rownn < n, (uint32_t)rownn * 8)) : 0;
#pragma unroll
            for (int q = 0; q < M3_NTI; ++q)
                xn[rt][q] = ROWS ? m3_load_x_rows_b<M3_TAILMAP>(B, cam0, cam1, cam2, row, srow_n[rt], q, g, validn[rt])
                                 : frag_bmask4<M3_IN>(frag_bload4<M3_IN>(B.X, (uint32_t)row * ldx4, q, g, validn[rt]), q, g);
        }
        if (ROWS) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#if M3_FWD_LDS_STORE
                if (R.X_out) {
                    if (M3_TAILMAP) {       // (the last piece is in tail order: values 48 + g and 52 + g of the row)
#pragma unroll
                        for (int q = 0; q < M3_NTI - 1; ++q) cl_patch_put4(patch + c * M3_IN + 16 * q + 4 * g, xb[rt][q], 4);
                        patch[c * M3_IN + 48 + g] = xb[rt][M3_NTI - 1][0];
                        if (g < 2) patch[c * M3_IN + 52 + g] = xb[rt][M3_NTI - 1][1];
                        cl_patch_flush<64 * M3_IN>(patch, B.Xo, (uint32_t)(row0 + rt * 16) * (M3_XLD * 4), lane);
                    } else {
                        frag_tile_store<M3_IN, M3_NTI>(patch, B.Xo, (uint32_t)(row0 + rt * 16) * (M3_XLD * 4), xb[rt], g, c, lane);
                    }
                }
#else
#pragma unroll
                for (int q = 0; q < M3_NTI; ++q)
                    frag_bstore4<M3_IN>(B.Xo, (uint32_t)(row0 + rt * 16 + c) * (M3_XLD * 4), q, g, valid[rt], xb[rt][q]);
#endif
        }
        m3_head_fwd<O0, A0, RT, TILED, ROWS && M3_TAILMAP>(l0, h0, 0, xb, valid, row0, g, c, Hcat, B, patch);
        m3_head_fwd<O1, A1, RT, TILED, ROWS && M3_TAILMAP>(l1, h1, 1, xb, valid, row0, g, c, Hcat, B, patch);
        m3_head_fwd<O2, A2, RT, TILED, ROWS && M3_TAILMAP>(l2, h2, 2, xb, valid, row0, g, c, Hcat, B, patch);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            valid[rt] = validn[rt];
            srow_n[rt] = srow_nn[rt];
#pragma unroll
            for (int q = 0; q < M3_NTI; ++q) xb[rt][q] = xn[rt][q];
        }
    }
}

// ---- backward ------------------------------------------------------------------------------------
template <int OUT>
struct M3BwdLds {
    static constexpr int NT2 = (OUT + 15) / 16, OP = NT2 * 16;
    static constexpr int SA = frag_pad4mod8(M3_HP), SB = frag_pad4mod8(M3_XP);
    static constexpr int FLOATS = OP * SA + M3_HP * SB;
};

template <int OUT>
__device__ __forceinline__ void m3_stage_bwd(float *lds, const M3Head &h, int tid, int nthr) {
    using L = M3BwdLds<OUT>;
    float *W2n = lds, *W1n = W2n + L::OP * L::SA;
    frag_stage_loop(h.W2, L::OP * L::SA, tid, nthr,
                    [](int i) { const int o = i / L::SA, hh = i % L::SA; return (o < OUT && hh < M3_HID) ? o * M3_HID + hh : -1; },
                    [&](int i, float v) { W2n[i] = v; });
    frag_stage_loop(h.W1, M3_HP * L::SB, tid, nthr,
                    [](int i) { const int hh = i / L::SB, k = i % L::SB; return (hh < M3_HID && k < M3_IN) ? hh * M3_IN + k : -1; },
                    [&](int i, float v) { W1n[i] = v; });
}

template <int OUT, int ACT, int RT>
__device__ __forceinline__ void m3_head_bwd(const float *lds, const M3Head &h, int head, const bool (&valid)[RT],
                                            int64_t row0, int g, int c, const float *__restrict__ Hcat,
                                            float *__restrict__ dZ1cat, f32x4 (&adx)[M3_NTI][RT]) {
    using L = M3BwdLds<OUT>;
    const float *W2n = lds, *W1n = W2n + L::OP * L::SA;
    f32x4 adh[M3_NT1][RT];
#pragma unroll
    for (int t = 0; t < M3_NT1; ++t)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) adh[t][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // Every load of the head is issued before its first store: the head pointers may alias as far as the compiler
    // knows, so a store in between would pin the later loads behind it (and their latency in front of the MFMAs).
    f32x4 b[L::NT2][RT], yv[L::NT2][RT], hv[M3_NT1][RT];
#pragma unroll
    for (int u = 0; u < L::NT2; ++u)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int64_t row = row0 + rt * 16 + c;
            b[u][rt] = frag_load4<OUT>(h.dY + row * OUT, u, g, valid[rt]);
            if (ACT != FRAG_ACT_NONE) yv[u][rt] = frag_load4<OUT>(h.Y + row * OUT, u, g, valid[rt]);
        }
#pragma unroll
    for (int t = 0; t < M3_NT1; ++t)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
            hv[t][rt] = frag_load4<M3_HID>(Hcat + (row0 + rt * 16 + c) * M3_HLD + M3_HPITCH * head, t, g, valid[rt]);
    if (ACT != FRAG_ACT_NONE) {
#pragma unroll
        for (int u = 0; u < L::NT2; ++u)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) b[u][rt][r] *= frag_act_grad<ACT>(yv[u][rt][r]);
                frag_store4<OUT>(h.dZ2 + (row0 + rt * 16 + c) * OUT, u, g, valid[rt], b[u][rt]);
            }
    }
#pragma unroll
    for (int u = 0; u < L::NT2; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (16 * u + j >= OUT) continue;
#pragma unroll
            for (int t = 0; t < M3_NT1; ++t) {
                const float a = W2n[(16 * u + 4 * g + j) * L::SA + 16 * t + c];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) adh[t][rt] = frag_mfma(a, b[u][rt][j], adh[t][rt]);
            }
        }
#pragma unroll
    for (int t = 0; t < M3_NT1; ++t)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int64_t at = (row0 + rt * 16 + c) * M3_GLD + M3_GPITCH * head;
#pragma unroll
            for (int r = 0; r < 4; ++r) adh[t][rt][r] = hv[t][rt][r] > 0.f ? adh[t][rt][r] : 0.f;
            frag_store4<M3_HP>(dZ1cat + at, t, g, valid[rt], adh[t][rt]);
        }
#pragma unroll
    for (int t = 0; t < M3_NT1; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (16 * t + r >= M3_HID) continue;
#pragma unroll
            for (int v = 0; v < M3_NTI; ++v) {
                const float a = W1n[(16 * t + 4 * g + r) * L::SB + 16 * v + c];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) adx[v][rt] = frag_mfma(a, adh[t][rt][r], adx[v][rt]);
            }
        }
}

template <int O0, int A0, int O1, int A1, int O2, int A2, int RT, int WAVES, bool ROWS>
__global__ void __launch_bounds__(WAVES * 64)
    mlp3_bwd_kernel(M3Head h0, M3Head h1, M3Head h2, const float *__restrict__ Hcat, float *__restrict__ dZ1cat,
                    float *__restrict__ dX, int64_t lddx, int64_t n, M3Rows R) {
    __shared__ float lds[M3BwdLds<O0>::FLOATS + M3BwdLds<O1>::FLOATS + M3BwdLds<O2>::FLOATS];
    float *l0 = lds, *l1 = l0 + M3BwdLds<O0>::FLOATS, *l2 = l1 + M3BwdLds<O1>::FLOATS;
    const int tid = threadIdx.x, nthr = WAVES * 64;
    m3_stage_bwd<O0>(l0, h0, tid, nthr);
    m3_stage_bwd<O1>(l1, h1, tid, nthr);
    m3_stage_bwd<O2>(l2, h2, tid, nthr);
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int64_t ntiles = (n + 16 * RT - 1) / (16 * RT);
    for (int64_t tile = (int64_t)blockIdx.x * WAVES + wave; tile < ntiles; tile += (int64_t)gridDim.x * WAVES) {
        const int64_t row0 = tile * 16 * RT;
        asm volatile("" ::: "memory");   // keep the LDS weight reads inside the tile loop (LICM would spill them)
        bool valid[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) valid[rt] = row0 + rt * 16 + c < n;
        f32x4 adx[M3_NTI][RT];
#pragma unroll
        for (int v = 0; v < M3_NTI; ++v)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) adx[v][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        m3_head_bwd<O0, A0, RT>(l0, h0, 0, valid, row0, g, c, Hcat, dZ1cat, adx);
        m3_head_bwd<O1, A1, RT>(l1, h1, 1, valid, row0, g, c, Hcat, dZ1cat, adx);
        m3_head_bwd<O2, A2, RT>(l2, h2, 2, valid, row0, g, c, Hcat, dZ1cat, adx);
        if (ROWS) {
            // dX[:, 0:50] goes straight to the rows of the source's gradient (distinct rows: plain stores), the four view
            // columns are pulled back to the anchor: u = a - cam, v = u/|u|:  da = (dv - v (v.dv)) / |u| + v d|u|
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int64_t row = row0 + rt * 16 + c;
                const int64_t srow = valid[rt] ? R.src_row[row] : 0;
                float *dst = R.d_feat_src + srow * M3_HID;
#pragma unroll
                for (int v = 0; v < 3; ++v)
                    if (valid[rt]) *(f32x4_a4 *)(dst + 16 * v + 4 * g) = adx[v][rt];
                // columns 48..55 sit in the q = 3 fragments of lanes g = 0 (48..51) and g = 1 (52..55) of this row
                const float z52 = __shfl(adx[3][rt][0], 16 + c, 64), z53 = __shfl(adx[3][rt][1], 16 + c, 64);
                if (g == 0 && valid[rt]) {
                    dst[48] = adx[3][rt][0];
                    dst[49] = adx[3][rt][1];
                    const float dvx = adx[3][rt][2], dvy = adx[3][rt][3], dvz = z52, dd = z53;
                    const float ux = R.anchor[3 * row] - R.cam[0], uy = R.anchor[3 * row + 1] - R.cam[1],
                                uz = R.anchor[3 * row + 2] - R.cam[2];
                    const float dist = sqrtf(ux * ux + uy * uy + uz * uz), inv = 1.f / dist;
                    const float vx = ux * inv, vy = uy * inv, vz = uz * inv;
                    const float dot = vx * dvx + vy * dvy + vz * dvz;
                    R.d_anchor[3 * row] = (dvx - vx * dot) * inv + vx * dd;
                    R.d_anchor[3 * row + 1] = (dvy - vy * dot) * inv + vy * dd;
                    R.d_anchor[3 * row + 2] = (dvz - vz * dot) * inv + vz * dd;
                }
            }
        } else if (dX) {
#pragma unroll
            for (int v = 0; v < M3_NTI; ++v)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    frag_store4<M3_IN>(dX + (row0 + rt * 16 + c) * lddx, v, g, valid[rt], adx[v][rt]);
        }
    }
}


// ---- backward with the weight gradients inside (round 4) ---------------------------------------------------------
// mlp3_bwd_kernel hands dZ1cat / dZ2 (928 B per row) to a second launch (wgrad_multi) that re-reads them together with X and
// Hcat (1.9 KB per row) only to contract them over the rows.  Here the same wave that forms dZ2 / dZ1 for its 16 rows also
// accumulates the weight gradients of those rows, with the ROW index as the MFMA contraction and the 80 output tiles
// (dW1cat [192 x 64] = 48 tiles, dW2 [(16 + 32 + 80) x 64] = 32 tiles; the bias gradients ride in the padding column of X
// (col 54 = 1) and of H (col 50 = 1)) held in 320 accumulator registers per lane for the whole launch: one wave per SIMD
// (4-wave workgroups, one per CU), the unified 512-entry VGPR + AGPR file.
// Layouts: the data-gradient chain keeps layout F (lane (g, c): row c, features 16q + 4g + {0..3} = the D layout of the
// transposed chain).  A row contraction needs both operands as lane (g, c): rows 4g + {0..3}, feature 16q + c (layout N =
// what the D registers of a NON-transposed product would hold): MFMA k-step r then contracts rows {4g + r}.  F -> N is a
// 16 x 16 transpose through a wave-private LDS patch (one ds_write_b128, four ds_read_b32, stride 20 floats: conflict-free
// both ways); 36 of them per 16-row tile.
// At the end the waves of a workgroup add their tiles into one LDS image [dW | db] per product (wave by wave: a fixed order),
// the image goes to scratch and cgs_launch_wgrad_reduce sums the workgroups' images in block order: bit-reproducible.
#define M3W_LD 20
#ifndef M3W_XPREFETCH
#define M3W_XPREFETCH 1      // X of the next tile is fetched during the last head of this one
#endif
#define M3W_PATCH (16 * M3W_LD)
#ifndef M3W_PREFETCH_AT
#define M3W_PREFETCH_AT 2    // where in a head the next head's global operands are requested: 0 after the transposes are written, 1 after dH, 2 after dW2
#endif
#ifndef M3W_EARLY
#define M3W_EARLY 0         // a loop's first LDS operands are requested one loop ahead
#endif
#define M3W_NPATCH 13          // per wave: H 0..3, dZ2 4..8, dZ1 9..12 (X uses 0..3 before the first head)
#define M3W_WAVES 4
#define M3W_E (M3_GLD * (M3_IN + 1) + (10 + 30 + 70) * (M3_HID + 1))

__device__ __forceinline__ f32x4 m3w_f2n(float *patch, f32x4 v, int g, int c) {
    *(f32x4 *)(patch + c * M3W_LD + 4 * g) = v;
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = patch[(4 * g + r) * M3W_LD + c];
    return o;
}

// operands of one head for one 16-row tile (layout F), fetched one head ahead of their use — through raw buffer loads (round 5,
// buf_access.h): no branch around a load, so no exec-mask join that drains the load queue in the middle of the prefetch (the
// predicated global loads of round 4 cost 26 `s_waitcnt vmcnt(0)` per tile); the pieces that run past a row end are zeroed by
// mask() when the head consumes them
struct M3wBufs { ClBuf dY[3], Y[3], H, X; };
template <int OUT, int ACT>
struct M3wOps {
    static constexpr int NT2 = (OUT + 15) / 16;
    f32x4 dy[NT2], y[ACT != FRAG_ACT_NONE ? NT2 : 1], h[M3_NT1];
    template <bool TILED = false>
    __device__ __forceinline__ void load(const M3wBufs &B, int head, int64_t row, int g, bool valid) {
        const uint32_t ro = (uint32_t)row * (OUT * 4), rh = (uint32_t)row * (M3_HLD * 4) + (uint32_t)(M3_HPITCH * head) * 4;
        const uint32_t th = (uint32_t)(row >> 4) * (16 * M3_HLD * 4) + (uint32_t)head * (16 * M3_HID * 4);
#pragma unroll
        for (int u = 0; u < NT2; ++u) {
            dy[u] = frag_bload4<OUT>(B.dY[head], ro, u, g, valid);
            if (ACT != FRAG_ACT_NONE) y[u] = frag_bload4<OUT>(B.Y[head], ro, u, g, valid);
        }
#pragma unroll
        for (int t = 0; t < M3_NT1; ++t)
            h[t] = TILED ? frag_tload4<M3_HID>(B.H, th, t, g, (int)(row & 15), valid) : frag_bload4<M3_HID>(B.H, rh, t, g, valid);
    }
    __device__ __forceinline__ void mask(int g) {
#pragma unroll
        for (int u = 0; u < NT2; ++u) {
            dy[u] = frag_bmask4<OUT>(dy[u], u, g);
            if (ACT != FRAG_ACT_NONE) y[u] = frag_bmask4<OUT>(y[u], u, g);
        }
#pragma unroll
        for (int t = 0; t < M3_NT1; ++t) h[t] = frag_bmask4<M3_HID>(h[t], t, g);
    }
};

__device__ __forceinline__ void m3w_put(float *patch, f32x4 v, int g, int c) { *(f32x4 *)(patch + c * M3W_LD + 4 * g) = v; }
__device__ __forceinline__ f32x4 m3w_get(const float *patch, int g, int c) {
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = patch[(4 * g + r) * M3W_LD + c];
    return o;
}

// One head of one tile.  Order of issue (one wave per SIMD: nothing else hides a latency): the F -> N transposes of H and
// dZ2 are written and read back BEFORE the dH products, those of dZ1 before the dW2 products, so every LDS round trip has
// a block of MFMAs in front of its first use; `prefetch` (the next head's global loads) is issued after the first batch of
// transposes.
template <int OUT, int ACT, class Prefetch>
__device__ __forceinline__ void m3w_head(const float *lds, float *patches, M3wOps<OUT, ACT> &op, f32x4 ones, int g, int c,
                                         const f32x4 (&xn)[M3_NTI], f32x4 (&adx)[M3_NTI],
                                         f32x4 (&aw2)[(OUT + 15) / 16][M3_NT1], f32x4 (&aw1)[M3_NT1][M3_NTI], Prefetch prefetch) {
    using L = M3BwdLds<OUT>;
    const float *W2n = lds, *W1n = W2n + L::OP * L::SA;
    op.mask(g);
    f32x4 b[L::NT2];
#pragma unroll
    for (int u = 0; u < L::NT2; ++u) {
        b[u] = op.dy[u];
        if (ACT != FRAG_ACT_NONE) {
#pragma unroll
            for (int r = 0; r < 4; ++r) b[u][r] *= frag_act_grad<ACT>(op.y[u][r]);
        }
    }
    // patches 0..3: H, 4..8: dZ2
#pragma unroll
    for (int t = 0; t < M3_NT1; ++t) m3w_put(patches + t * M3W_PATCH, op.h[t], g, c);
#pragma unroll
    for (int u = 0; u < L::NT2; ++u) m3w_put(patches + (4 + u) * M3W_PATCH, b[u], g, c);
    if (M3W_PREFETCH_AT == 0) prefetch();
    // (round 5) Every MFMA group's LDS operands are read one group ahead, behind scheduling fences, and each loop's FIRST
    // operands one loop ahead (M3W_EARLY): with one wave per SIMD a read issued just in time leaves the matrix pipe idle for
    // its whole round trip — 375 of the 616 MFMAs of a tile had an lgkmcnt wait right in front of them (1023 -> 880 us).
    auto ldw2 = [&](int u, int j) {
        f32x4 w;
#pragma unroll
        for (int t = 0; t < M3_NT1; ++t) w[t] = W2n[(16 * u + 4 * g + j) * L::SA + 16 * t + c];
        return w;
    };
    auto ldw1 = [&](int t, int r) {
        f32x4 w;
#pragma unroll
        for (int v = 0; v < M3_NTI; ++v) w[v] = W1n[(16 * t + 4 * g + r) * L::SB + 16 * v + c];
        return w;
    };
    // dH = W2^T dZ2 (transposed chain, layout F)
    f32x4 adh[M3_NT1], hn[M3_NT1], zc;
#pragma unroll
    for (int t = 0; t < M3_NT1; ++t) adh[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        f32x4 wc = ldw2(0, 0);
        if (M3W_EARLY) {                        // the N forms of H and of the first dZ2 tile: needed by the dW2 products
#pragma unroll
            for (int t = 0; t < M3_NT1; ++t) hn[t] = m3w_get(patches + t * M3W_PATCH, g, c);
            zc = m3w_get(patches + 4 * M3W_PATCH, g, c);
        }
#pragma unroll
        for (int u = 0; u < L::NT2; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (16 * u + j >= OUT) continue;
                const int un = j == 3 ? u + 1 : u, jn = j == 3 ? 0 : j + 1;
                f32x4 wn = wc;
                if (un < L::NT2 && 16 * un + jn < OUT) wn = ldw2(un, jn);
                M3_FENCE();
#pragma unroll
                for (int t = 0; t < M3_NT1; ++t) adh[t] = frag_mfma(wc[t], b[u][j], adh[t]);
                M3_FENCE();
                wc = wn;
            }
    }
    if (M3W_PREFETCH_AT == 1) prefetch();
#pragma unroll
    for (int t = 0; t < M3_NT1; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) adh[t][r] = op.h[t][r] > 0.f ? adh[t][r] : 0.f;
        m3w_put(patches + (9 + t) * M3W_PATCH, adh[t], g, c);          // patches 9..12: dZ1
    }
    // dW2 += dZ2^T [H | 1] (row contraction, layout N)
    if (!M3W_EARLY) {
#pragma unroll
        for (int t = 0; t < M3_NT1; ++t) hn[t] = m3w_get(patches + t * M3W_PATCH, g, c);
        zc = m3w_get(patches + 4 * M3W_PATCH, g, c);
    }
    if (c == M3_HID - 48) hn[3] = ones;                         // column 50 of [H | 1]
    f32x4 wx, dc;
    if (M3W_EARLY) {                            // first operands of the dX and dW1 loops, in flight during the dW2 products
        wx = ldw1(0, 0);
        dc = m3w_get(patches + 9 * M3W_PATCH, g, c);
    }
#pragma unroll
    for (int u = 0; u < L::NT2; ++u) {
        f32x4 zn = zc;
        if (u + 1 < L::NT2) zn = m3w_get(patches + (4 + u + 1) * M3W_PATCH, g, c);
        M3_FENCE();
#if M3W_ROUTER      // r outermost: consecutive MFMAs go to DIFFERENT accumulators (same additions per accumulator, same order)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < M3_NT1; ++t) aw2[u][t] = frag_mfma(zc[r], hn[t][r], aw2[u][t]);
#else
#pragma unroll
        for (int t = 0; t < M3_NT1; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) aw2[u][t] = frag_mfma(zc[r], hn[t][r], aw2[u][t]);
#endif
        M3_FENCE();
        zc = zn;
    }
    if (M3W_PREFETCH_AT == 2) prefetch();
    // dX += W1^T dZ1
    if (!M3W_EARLY) wx = ldw1(0, 0);
#pragma unroll
    for (int t = 0; t < M3_NT1; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (16 * t + r >= M3_HID) continue;
            const int tn = r == 3 ? t + 1 : t, rn = r == 3 ? 0 : r + 1;
            f32x4 wn = wx;
            if (tn < M3_NT1 && 16 * tn + rn < M3_HID) wn = ldw1(tn, rn);
            M3_FENCE();
#pragma unroll
            for (int v = 0; v < M3_NTI; ++v) adx[v] = frag_mfma(wx[v], adh[t][r], adx[v]);
            M3_FENCE();
            wx = wn;
        }
    // dW1 += dZ1^T [X | 1]
    if (!M3W_EARLY) dc = m3w_get(patches + 9 * M3W_PATCH, g, c);
#pragma unroll
    for (int t = 0; t < M3_NT1; ++t) {
        f32x4 dn = dc;
        if (t + 1 < M3_NT1) dn = m3w_get(patches + (9 + t + 1) * M3W_PATCH, g, c);
        M3_FENCE();
#if M3W_ROUTER
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int v = 0; v < M3_NTI; ++v) aw1[t][v] = frag_mfma(dc[r], xn[v][r], aw1[t][v]);
#else
#pragma unroll
        for (int v = 0; v < M3_NTI; ++v)
#pragma unroll
            for (int r = 0; r < 4; ++r) aw1[t][v] = frag_mfma(dc[r], xn[v][r], aw1[t][v]);
#endif
        M3_FENCE();
        dc = dn;
    }
}

template <bool ROWS, bool TL = false>         // TL: Hcat is the TILED forward's (fragment-major: one contiguous KB per load)
__global__ void __launch_bounds__(M3W_WAVES * 64) __attribute__((amdgpu_waves_per_eu(1, 1)))
    mlp3_bwd_wg_kernel(M3Head h0, M3Head h1, M3Head h2, const float *__restrict__ X, int64_t ldx,
                       const float *__restrict__ Hcat, float *__restrict__ dX, int64_t lddx, int64_t n, M3Rows R,
                       float *__restrict__ partial) {
    constexpr int WF = M3BwdLds<10>::FLOATS + M3BwdLds<30>::FLOATS + M3BwdLds<70>::FLOATS;
    constexpr int PF = M3W_WAVES * M3W_NPATCH * M3W_PATCH;
    static_assert(WF + PF >= M3W_E, "the LDS image of the weight gradients reuses the weight region");
    __shared__ __attribute__((aligned(16))) float lds[WF + PF];
    float *l0 = lds, *l1 = l0 + M3BwdLds<10>::FLOATS, *l2 = l1 + M3BwdLds<30>::FLOATS;
    const int tid = threadIdx.x, nthr = M3W_WAVES * 64;
    m3_stage_bwd<10>(l0, h0, tid, nthr);
    m3_stage_bwd<30>(l1, h1, tid, nthr);
    m3_stage_bwd<70>(l2, h2, tid, nthr);
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    float *patches = lds + WF + wave * M3W_NPATCH * M3W_PATCH;
    const int64_t ntiles = (n + 15) / 16;
    f32x4 a2_0[1][M3_NT1], a2_1[2][M3_NT1], a2_2[5][M3_NT1], a1_0[M3_NT1][M3_NTI], a1_1[M3_NT1][M3_NTI], a1_2[M3_NT1][M3_NTI];
    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < M3_NT1; ++t) {
        a2_0[0][t] = zero;
#pragma unroll
        for (int u = 0; u < 2; ++u) a2_1[u][t] = zero;
#pragma unroll
        for (int u = 0; u < 5; ++u) a2_2[u][t] = zero;
#pragma unroll
        for (int v = 0; v < M3_NTI; ++v) { a1_0[t][v] = zero; a1_1[t][v] = zero; a1_2[t][v] = zero; }
    }
    const int64_t tstride = (int64_t)gridDim.x * M3W_WAVES;
    int64_t tile = (int64_t)blockIdx.x * M3W_WAVES + wave;
    const uint64_t nb = (uint64_t)n;
    M3wBufs B;
    B.dY[0] = cl_buf(h0.dY, nb * 40); B.dY[1] = cl_buf(h1.dY, nb * 120); B.dY[2] = cl_buf(h2.dY, nb * 280);
    B.Y[0] = cl_buf(h0.Y, nb * 40); B.Y[1] = cl_buf(h1.Y, nb * 120); B.Y[2] = cl_buf(h2.Y, nb * 280);
    static_assert(ROWS || !TL, "the tiled hand-over belongs to the ROWS pair");
    B.H = cl_buf(Hcat, (TL ? (nb + 15) / 16 * 16 : nb) * (M3_HLD * 4));
    B.X = cl_buf(X, nb > 0 ? ((nb - 1) * (uint64_t)ldx + M3_IN) * 4 : 0);
    const ClBuf bSrc = cl_buf(ROWS ? R.src_row : nullptr, nb * 8), bAnc = cl_buf(ROWS ? R.anchor : nullptr, nb * 12);
    const float cam0 = ROWS ? R.cam[0] : 0.f, cam1 = ROWS ? R.cam[1] : 0.f, cam2 = ROWS ? R.cam[2] : 0.f;
    const uint32_t ldx4 = (uint32_t)ldx * 4;
    // (round 6) X == NULL (ROWS): the forward did not keep its assembled input rows (216 of the 1256 bytes per anchor it
    // stores: the forward is bound by exactly those stores) — the rows are assembled again here, by the forward's own loader
    // from the same operands: the same bits
    const bool regather = ROWS && X == nullptr;
    M3FwdBufs FB;
    FB.feat = cl_buf(regather ? R.feat_src : nullptr, CL_MAX_BYTES);
    FB.anc = bAnc;
    auto load_x = [&](f32x4 (&x)[M3_NTI], int64_t row, int64_t srow, bool v) {
#pragma unroll
        for (int q = 0; q < M3_NTI; ++q)
            x[q] = regather ? m3_load_x_rows_b(FB, cam0, cam1, cam2, row, srow, q, g, v)
                            : frag_bload4<M3_IN>(B.X, (uint32_t)row * ldx4, q, g, v);
    };
    M3wOps<10, 1> op0;
    M3wOps<30, 2> op1;
    M3wOps<70, 0> op2;
    f32x4 xf[M3_NTI];
    int64_t srow_next = 0;                  // (regather) source row of the NEXT tile's row c, fetched a tile ahead
    {
        const int64_t row = tile * 16 + c;
        const bool v0 = tile < ntiles && row < n;
        op0.template load<TL>(B, 0, row, g, v0);
        const int64_t s0 = regather ? cl_li64(bSrc, cl_sel(v0, (uint32_t)row * 8)) : 0;
        load_x(xf, row, s0, v0);
    }
    for (; tile < ntiles; tile += tstride) {
        const int64_t row0 = tile * 16;
        asm volatile("" ::: "memory");   // keep the LDS weight reads inside the tile loop (LICM would spill them)
        const bool valid = row0 + c < n;
        const int64_t rown = row0 + tstride * 16 + c;
        const bool validn = rown < n;
        // [X | 1] in layout N, and the ones column of [H | 1]: rows 4g + r of this tile that exist
        f32x4 ones, xn[M3_NTI];
#pragma unroll
        for (int r = 0; r < 4; ++r) ones[r] = row0 + 4 * g + r < n ? 1.f : 0.f;
        // (ROWS) what the end of the tile needs from memory, issued here: the row's source index and its anchor
        const int64_t srow_pre = ROWS ? cl_li64(bSrc, cl_sel(valid, (uint32_t)(row0 + c) * 8)) : 0;
#if !M3W_XPREFETCH
        load_x(xf, row0 + c, srow_pre, valid);
#endif
        if (regather) srow_next = cl_li64(bSrc, cl_sel(validn, (uint32_t)rown * 8));
        const f32x4 anc_pre = ROWS ? cl_l96(bAnc, cl_sel(valid && g == 0, (uint32_t)(row0 + c) * 12)) : zero;
#pragma unroll
        for (int q = 0; q < M3_NTI; ++q) m3w_put(patches + q * M3W_PATCH, frag_bmask4<M3_IN>(xf[q], q, g), g, c);
#pragma unroll
        for (int q = 0; q < M3_NTI; ++q) xn[q] = m3w_get(patches + q * M3W_PATCH, g, c);
        if (c == M3_IN - 48) xn[3] = ones;                      // column 54 of [X | 1]
        f32x4 adx[M3_NTI];
#pragma unroll
        for (int v = 0; v < M3_NTI; ++v) adx[v] = zero;
        m3w_head<10, 1>(l0, patches, op0, ones, g, c, xn, adx, a2_0, a1_0, [&]() { op1.template load<TL>(B, 1, row0 + c, g, valid); });
        m3w_head<30, 2>(l1, patches, op1, ones, g, c, xn, adx, a2_1, a1_1, [&]() { op2.template load<TL>(B, 2, row0 + c, g, valid); });
        m3w_head<70, 0>(l2, patches, op2, ones, g, c, xn, adx, a2_2, a1_2, [&]() {
            op0.template load<TL>(B, 0, rown, g, validn);
#if M3W_XPREFETCH
            load_x(xf, rown, srow_next, validn);
#endif
        });
        if (ROWS) {
            const int64_t row = row0 + c;
            const int64_t srow = valid ? srow_pre : 0;
            float *dst = R.d_feat_src + srow * M3_HID;
#pragma unroll
            for (int v = 0; v < 3; ++v)
                if (valid) *(f32x4_a4 *)(dst + 16 * v + 4 * g) = adx[v];
            const float z52 = __shfl(adx[3][0], 16 + c, 64), z53 = __shfl(adx[3][1], 16 + c, 64);
            if (g == 0 && valid) {
                dst[48] = adx[3][0];
                dst[49] = adx[3][1];
                const float dvx = adx[3][2], dvy = adx[3][3], dvz = z52, dd = z53;
                const float ux = anc_pre[0] - cam0, uy = anc_pre[1] - cam1, uz = anc_pre[2] - cam2;
                const float dist = sqrtf(ux * ux + uy * uy + uz * uz), inv = 1.f / dist;
                const float vx = ux * inv, vy = uy * inv, vz = uz * inv;
                const float dot = vx * dvx + vy * dvy + vz * dvz;
                R.d_anchor[3 * row] = (dvx - vx * dot) * inv + vx * dd;
                R.d_anchor[3 * row + 1] = (dvy - vy * dot) * inv + vy * dd;
                R.d_anchor[3 * row + 2] = (dvz - vz * dot) * inv + vz * dd;
            }
        } else if (dX) {
#pragma unroll
            for (int v = 0; v < M3_NTI; ++v) frag_store4<M3_IN>(dX + (row0 + c) * lddx, v, g, valid, adx[v]);
        }
    }
    // ---- the workgroup's image [dW1cat 192 x 54 | db1cat 192 | dW2_h OUT_h x 50 | db2_h OUT_h ...] ----
    __syncthreads();
    for (int i = tid; i < M3W_E; i += nthr) lds[i] = 0.f;
    __syncthreads();
    float *img1 = lds, *imgb1 = img1 + M3_GLD * M3_IN;
    for (int w = 0; w < M3W_WAVES; ++w) {
        if (wave == w) {
#pragma unroll
            for (int hd = 0; hd < 3; ++hd) {
                const f32x4 (&a1)[M3_NT1][M3_NTI] = hd == 0 ? a1_0 : (hd == 1 ? a1_1 : a1_2);
#pragma unroll
                for (int t = 0; t < M3_NT1; ++t)
#pragma unroll
                    for (int v = 0; v < M3_NTI; ++v)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int m = M3_GPITCH * hd + 16 * t + 4 * g + r, k = 16 * v + c;
                            if (k < M3_IN) img1[m * M3_IN + k] += a1[t][v][r];
                            else if (k == M3_IN) imgb1[m] += a1[t][v][r];
                        }
            }
            float *img2 = imgb1 + M3_GLD;
#define M3W_IMG2(ACC, NT2_, OUT_)                                                          \
            {                                                                               \
                _Pragma("unroll") for (int u = 0; u < NT2_; ++u)                            \
                _Pragma("unroll") for (int t = 0; t < M3_NT1; ++t)                          \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) {                             \
                    const int o = 16 * u + 4 * g + r, hh = 16 * t + c;                      \
                    if (o < OUT_) {                                                         \
                        if (hh < M3_HID) img2[o * M3_HID + hh] += ACC[u][t][r];             \
                        else if (hh == M3_HID) img2[OUT_ * M3_HID + o] += ACC[u][t][r];     \
                    }                                                                       \
                }                                                                           \
                img2 += OUT_ * (M3_HID + 1);                                                \
            }
            M3W_IMG2(a2_0, 1, 10)
            M3W_IMG2(a2_1, 2, 30)
            M3W_IMG2(a2_2, 5, 70)
#undef M3W_IMG2
        }
        __syncthreads();
    }
    float *dstp = partial + (int64_t)blockIdx.x * M3W_E;
    for (int i = tid; i < M3W_E; i += nthr) dstp[i] = lds[i];
}


static int m3_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

// Row strides (floats) of the buffers the forward / backward hand to each other: out4 = {Hcat row stride (150: head h in
// columns 50 h ..), X_out row stride of the ROWS variant (54), dZ1cat row stride (192), dZ1cat column pitch of a head (64:
// head h in columns 64 h .. 64 h + 49; dW1cat [192, 54] and db1cat [192] use the same row numbering)}.
extern "C" int cgs_anchor_mlp3_layout(int *out4) {
    if (!out4) { cgs_set_error("cgs_anchor_mlp3_layout: NULL"); return CGS_ERR_ARG; }
    out4[0] = M3_HLD; out4[1] = M3_XLD; out4[2] = M3_GLD; out4[3] = M3_GPITCH;
    return CGS_OK;
}

// X [n, ldx]; W1* [50,54], b1* [50], W2* [OUT,50], b2* [OUT]; outputs Y_op [n,10] (tanh), Y_color [n,30] (sigmoid),
// Y_cov [n,70]; Hcat [n,150] = relu hidden of the three MLPs (NULL for inference).
extern "C" int cgs_anchor_mlp3_forward(const float *X, int64_t ldx, const float *const *W1, const float *const *b1,
                                       const float *const *W2, const float *const *b2, float *Y_op, float *Y_color,
                                       float *Y_cov, float *Hcat, int64_t n, void *stream) {
    if (n < 0) { cgs_set_error("anchor_mlp3_forward: n < 0"); return CGS_ERR_ARG; }
    if (n == 0) return CGS_OK;
    if (!X || !W1 || !b1 || !W2 || !b2 || !Y_op || !Y_color || !Y_cov) { cgs_set_error("anchor_mlp3_forward: NULL"); return CGS_ERR_ARG; }
    // 16-row wave tiles, 16 waves per workgroup (4 per SIMD, 111-118 VGPRs): 548 -> 510 us at 1 M anchors against the
    // 32-row / 8-wave shape (tools/mlp_tiling.sh); the backward measured the same either way and keeps 32 / 8
    constexpr int RT = M3_FWD_RT, WAVES = M3_FWD_WAVES;
    M3Head h[3];
    float *ys[3] = {Y_op, Y_color, Y_cov};
    for (int i = 0; i < 3; ++i) h[i] = M3Head{W1[i], b1[i], W2[i], b2[i], ys[i], nullptr, nullptr};
    if (ldx < M3_IN || ldx > 4096) { cgs_set_error("anchor_mlp3_forward: ldx out of range"); return CGS_ERR_ARG; }
    CgsProfScope prof(CGS_PROF_MLP_FWD, (hipStream_t)stream);
    // the kernel addresses its operands through 32-bit byte offsets (raw buffers): M3_MAX_ROWS rows per launch
    const int64_t step = ldx <= 256 ? M3_MAX_ROWS : M3_MAX_ROWS / 16;
    for (int64_t r0 = 0; r0 < n; r0 += step) {
        const int64_t m = n - r0 < step ? n - r0 : step;
        for (int i = 0; i < 3; ++i) h[i].Y = ys[i] + r0 * (i == 0 ? 10 : (i == 1 ? 30 : 70));
        const int64_t tiles = (m + 16 * RT - 1) / (16 * RT);
        const int64_t want = (tiles + WAVES - 1) / WAVES;
        const int grid = (int)(want < m3_cus() ? want : m3_cus());
        hipLaunchKernelGGL((mlp3_fwd_kernel<10, 1, 30, 2, 70, 0, RT, WAVES, false>), dim3(grid), dim3(WAVES * 64), 0, (hipStream_t)stream,
                           X + r0 * ldx, ldx, h[0], h[1], h[2], Hcat ? Hcat + r0 * M3_HLD : nullptr, m, M3Rows{});
    }
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

// The same with the input row assembled on the fly: X[r] = [feat_src[src_row[r], 0:50] | view direction (3) | distance (1)]
// of anchor_vis[r] as seen from cam3 (device float[3]); X_out [n,54] receives the assembled rows (the weight-gradient
// pass of the backward reads them).
extern "C" int cgs_anchor_mlp3_forward_rows_t(const float *feat_src, const int64_t *src_row, const float *anchor_vis,
                                              const float *cam3, float *X_out, const float *const *W1,
                                              const float *const *b1, const float *const *W2, const float *const *b2,
                                              float *Y_op, float *Y_color, float *Y_cov, float *Hcat, int64_t n, int tiled,
                                              void *stream);
extern "C" int cgs_anchor_mlp3_forward_rows(const float *feat_src, const int64_t *src_row, const float *anchor_vis,
                                            const float *cam3, float *X_out, const float *const *W1,
                                            const float *const *b1, const float *const *W2, const float *const *b2,
                                            float *Y_op, float *Y_color, float *Y_cov, float *Hcat, int64_t n, void *stream) {
    return cgs_anchor_mlp3_forward_rows_t(feat_src, src_row, anchor_vis, cam3, X_out, W1, b1, W2, b2, Y_op, Y_color, Y_cov, Hcat, n, 0,
                                          stream);
}
// tiled != 0: Hcat (ceil(n / 16) * 16 rows) leaves in fragment-major form (mlp3_fwd_kernel, TILED) — only
// cgs_anchor_mlp3_backward_rows_t(tiled = 1) reads that form.  One launch (n <= M3_MAX_ROWS) in that case.
extern "C" int cgs_anchor_mlp3_forward_rows_t(const float *feat_src, const int64_t *src_row, const float *anchor_vis,
                                              const float *cam3, float *X_out, const float *const *W1,
                                              const float *const *b1, const float *const *W2, const float *const *b2,
                                              float *Y_op, float *Y_color, float *Y_cov, float *Hcat, int64_t n, int tiled,
                                              void *stream) {
    if (tiled && n > M3_MAX_ROWS) { cgs_set_error("anchor_mlp3_forward_rows: tiled hand-over needs <= %lld rows", (long long)M3_MAX_ROWS); return CGS_ERR_ARG; }
    if (n < 0) { cgs_set_error("anchor_mlp3_forward_rows: n < 0"); return CGS_ERR_ARG; }
    if (n == 0) return CGS_OK;
    if (!feat_src || !src_row || !anchor_vis || !cam3 || !W1 || !b1 || !W2 || !b2 || !Y_op || !Y_color || !Y_cov) {
        cgs_set_error("anchor_mlp3_forward_rows: NULL");
        return CGS_ERR_ARG;
    }
    // 16-row wave tiles, 16 waves per workgroup (4 per SIMD, 111-118 VGPRs): 548 -> 510 us at 1 M anchors against the
    // 32-row / 8-wave shape (tools/mlp_tiling.sh); the backward measured the same either way and keeps 32 / 8
    constexpr int RT = M3_FWD_RT, WAVES = M3_FWD_WAVES;
    M3Head h[3];
    float *ys[3] = {Y_op, Y_color, Y_cov};
    for (int i = 0; i < 3; ++i) h[i] = M3Head{W1[i], b1[i], W2[i], b2[i], ys[i], nullptr, nullptr};
    CgsProfScope prof(CGS_PROF_MLP_FWD, (hipStream_t)stream);
    // 32-bit byte offsets inside the kernel (raw buffers): M3_MAX_ROWS rows per launch; feat_src itself must stay below 4 GB
    // (21 M source rows)
    for (int64_t r0 = 0; r0 < n; r0 += M3_MAX_ROWS) {
        const int64_t m = n - r0 < M3_MAX_ROWS ? n - r0 : M3_MAX_ROWS;
        for (int i = 0; i < 3; ++i) h[i].Y = ys[i] + r0 * (i == 0 ? 10 : (i == 1 ? 30 : 70));
        const int64_t tiles = (m + 16 * RT - 1) / (16 * RT);
        const int64_t want = (tiles + WAVES - 1) / WAVES;
        const int grid = (int)(want < m3_cus() ? want : m3_cus());
        M3Rows R{feat_src, src_row + r0, anchor_vis + 3 * r0, cam3, X_out ? X_out + r0 * M3_XLD : nullptr, nullptr, nullptr};
        if (tiled)
            hipLaunchKernelGGL((mlp3_fwd_kernel<10, 1, 30, 2, 70, 0, RT, WAVES, true, true>), dim3(grid), dim3(WAVES * 64), 0, (hipStream_t)stream,
                               nullptr, 0, h[0], h[1], h[2], Hcat, m, R);
        else
            hipLaunchKernelGGL((mlp3_fwd_kernel<10, 1, 30, 2, 70, 0, RT, WAVES, true>), dim3(grid), dim3(WAVES * 64), 0, (hipStream_t)stream,
                               nullptr, 0, h[0], h[1], h[2], Hcat ? Hcat + r0 * M3_HLD : nullptr, m, R);
    }
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

// dY_* are the gradients of the three outputs; Y_op / Y_color the saved forward outputs (activation derivatives).
// Scratch: dZ1cat [n,150], dZ2_op [n,10], dZ2_color [n,30].  dX [n, lddx] may be NULL.  Weight / bias gradients are
// ACCUMULATED (atomics): dW1cat [150,54], db1cat [150], dW2[i] [OUT_i,50], db2[i] [OUT_i].
static int m3_backward(const float *X, int64_t ldx, const float *const *W1, const float *const *W2, const float *Y_op,
                       const float *Y_color, const float *dY_op, const float *dY_color, const float *dY_cov,
                       const float *Hcat, float *dX, int64_t lddx, float *dZ1cat, float *dZ2_op, float *dZ2_color,
                       float *dW1cat, float *db1cat, float *const *dW2, float *const *db2, int64_t n, void *scratch,
                       size_t scratch_bytes, const M3Rows *rows, void *stream_, int tiled = 0);
extern "C" int cgs_anchor_mlp3_wgrad(const float *X, int64_t ldx, const float *Hcat, const float *dZ1cat,
                                     const float *dZ2_op, const float *dZ2_color, const float *dY_cov, float *dW1cat,
                                     float *db1cat, float *const *dW2, float *const *db2, int64_t n, void *scratch,
                                     size_t scratch_bytes, void *stream_);

extern "C" int cgs_anchor_mlp3_backward(const float *X, int64_t ldx, const float *const *W1, const float *const *W2,
                                        const float *Y_op, const float *Y_color, const float *dY_op,
                                        const float *dY_color, const float *dY_cov, const float *Hcat, float *dX,
                                        int64_t lddx, float *dZ1cat, float *dZ2_op, float *dZ2_color, float *dW1cat,
                                        float *db1cat, float *const *dW2, float *const *db2, int64_t n, void *scratch,
                                        size_t scratch_bytes, void *stream_) {
    return m3_backward(X, ldx, W1, W2, Y_op, Y_color, dY_op, dY_color, dY_cov, Hcat, dX, lddx, dZ1cat, dZ2_op, dZ2_color,
                       dW1cat, db1cat, dW2, db2, n, scratch, scratch_bytes, nullptr, stream_);
}

// Backward of cgs_anchor_mlp3_forward_rows: X is the [n,54] side output of the forward; instead of a dense dX the
// feature columns are stored into rows src_row[r] of d_feat_src [*,50] (distinct rows; rows no visible anchor reads
// are the caller's to zero) and the view columns are pulled back to d_anchor_vis [n,3].
extern "C" int cgs_anchor_mlp3_backward_rows_t(const float *X, const float *feat_src, const int64_t *src_row, const float *anchor_vis,
                                               const float *cam3, const float *const *W1, const float *const *W2,
                                               const float *Y_op, const float *Y_color, const float *dY_op,
                                               const float *dY_color, const float *dY_cov, const float *Hcat,
                                               float *d_feat_src, float *d_anchor_vis, float *dZ1cat, float *dZ2_op,
                                               float *dZ2_color, float *dW1cat, float *db1cat, float *const *dW2,
                                               float *const *db2, int64_t n, int tiled, void *scratch, size_t scratch_bytes,
                                               void *stream_);
extern "C" int cgs_anchor_mlp3_backward_rows(const float *X, const float *feat_src, const int64_t *src_row, const float *anchor_vis,
                                             const float *cam3, const float *const *W1, const float *const *W2,
                                             const float *Y_op, const float *Y_color, const float *dY_op,
                                             const float *dY_color, const float *dY_cov, const float *Hcat,
                                             float *d_feat_src, float *d_anchor_vis, float *dZ1cat, float *dZ2_op,
                                             float *dZ2_color, float *dW1cat, float *db1cat, float *const *dW2,
                                             float *const *db2, int64_t n, void *scratch, size_t scratch_bytes,
                                             void *stream_) {
    return cgs_anchor_mlp3_backward_rows_t(X, feat_src, src_row, anchor_vis, cam3, W1, W2, Y_op, Y_color, dY_op, dY_color, dY_cov, Hcat,
                                           d_feat_src, d_anchor_vis, dZ1cat, dZ2_op, dZ2_color, dW1cat, db1cat, dW2, db2, n, 0, scratch,
                                           scratch_bytes, stream_);
}
// tiled != 0: Hcat is the fragment-major buffer of cgs_anchor_mlp3_forward_rows_t(tiled = 1) (X stays row-major); the weight
// gradients must be asked for in the same call (the fused form is the only reader of that layout).
extern "C" int cgs_anchor_mlp3_backward_rows_t(const float *X, const float *feat_src, const int64_t *src_row, const float *anchor_vis,
                                               const float *cam3, const float *const *W1, const float *const *W2,
                                               const float *Y_op, const float *Y_color, const float *dY_op,
                                               const float *dY_color, const float *dY_cov, const float *Hcat,
                                               float *d_feat_src, float *d_anchor_vis, float *dZ1cat, float *dZ2_op,
                                               float *dZ2_color, float *dW1cat, float *db1cat, float *const *dW2,
                                               float *const *db2, int64_t n, int tiled, void *scratch, size_t scratch_bytes,
                                               void *stream_) {
    if (n == 0) return CGS_OK;           // no visible anchor: nothing to do (empty tensors arrive as NULL pointers)
    if (!src_row || !anchor_vis || !cam3 || !d_feat_src || !d_anchor_vis) { cgs_set_error("anchor_mlp3_backward_rows: NULL"); return CGS_ERR_ARG; }
    if (!X && (!feat_src || !dW1cat || n > M3_MAX_ROWS)) {
        cgs_set_error("anchor_mlp3_backward_rows: X == NULL needs feat_src, the fused weight-gradient form and <= %lld rows", (long long)M3_MAX_ROWS);
        return CGS_ERR_ARG;
    }
    M3Rows R{X ? nullptr : feat_src, src_row, anchor_vis, cam3, nullptr, d_feat_src, d_anchor_vis};
    return m3_backward(X, M3_XLD, W1, W2, Y_op, Y_color, dY_op, dY_color, dY_cov, Hcat, nullptr, 0, dZ1cat, dZ2_op, dZ2_color,
                       dW1cat, db1cat, dW2, db2, n, scratch, scratch_bytes, &R, stream_, tiled);
}

static int m3_backward(const float *X, int64_t ldx, const float *const *W1, const float *const *W2, const float *Y_op,
                       const float *Y_color, const float *dY_op, const float *dY_color, const float *dY_cov,
                       const float *Hcat, float *dX, int64_t lddx, float *dZ1cat, float *dZ2_op, float *dZ2_color,
                       float *dW1cat, float *db1cat, float *const *dW2, float *const *db2, int64_t n, void *scratch,
                       size_t scratch_bytes, const M3Rows *rows, void *stream_, int tiled) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0) { cgs_set_error("anchor_mlp3_backward: n < 0"); return CGS_ERR_ARG; }
    if (n == 0) return CGS_OK;
    // dW1cat == NULL: data gradients only — the weight-gradient launch is the caller's to make later (cgs_anchor_mlp3_wgrad)
    const bool data_only = !dW1cat;
    const bool no_x = !X && rows && rows->feat_src;          // the rows are assembled again by the fused kernel
    if ((!X && !no_x) || !W1 || !W2 || !Y_op || !Y_color || !dY_op || !dY_color || !dY_cov || !Hcat || (!no_x && (!dZ1cat || !dZ2_op ||
        !dZ2_color)) || (data_only ? (db1cat || dW2 || db2) : (!db1cat || !dW2 || !db2))) {
        cgs_set_error("anchor_mlp3_backward: NULL (or a partial set of weight-gradient pointers)");
        return CGS_ERR_ARG;
    }
    constexpr int RT = M3_BWD_RT, WAVES = M3_BWD_WAVES;
    M3Head h[3];
    h[0] = M3Head{W1[0], nullptr, W2[0], nullptr, const_cast<float *>(Y_op), dY_op, dZ2_op};
    h[1] = M3Head{W1[1], nullptr, W2[1], nullptr, const_cast<float *>(Y_color), dY_color, dZ2_color};
    h[2] = M3Head{W1[2], nullptr, W2[2], nullptr, nullptr, dY_cov, nullptr};
#if M3_FUSED_WGRAD
    // (the fused kernel addresses its operands through 32-bit byte offsets: beyond M3_MAX_ROWS rows the two-launch form runs)
    if (!data_only && n <= M3_MAX_ROWS && ldx <= 256) {
        // data AND weight gradients in one launch (mlp3_bwd_wg_kernel): dZ1cat / dZ2_* stay untouched
        const int64_t tiles16 = (n + 15) / 16;
        const int64_t wantw = (tiles16 + M3W_WAVES - 1) / M3W_WAVES;
        const int gridw = (int)(wantw < m3_cus() ? wantw : m3_cus());
        if (scratch && scratch_bytes >= (size_t)gridw * M3W_E * sizeof(float)) {
            {
                CgsProfScope prof(CGS_PROF_MLP_BWD, stream);
                if (rows && tiled)
                    hipLaunchKernelGGL((mlp3_bwd_wg_kernel<true, true>), dim3(gridw), dim3(M3W_WAVES * 64), 0, stream, h[0], h[1], h[2], X,
                                       ldx, Hcat, nullptr, 0, n, *rows, (float *)scratch);
                else if (rows)
                    hipLaunchKernelGGL((mlp3_bwd_wg_kernel<true>), dim3(gridw), dim3(M3W_WAVES * 64), 0, stream, h[0], h[1], h[2], X,
                                       ldx, Hcat, nullptr, 0, n, *rows, (float *)scratch);
                else
                    hipLaunchKernelGGL((mlp3_bwd_wg_kernel<false>), dim3(gridw), dim3(M3W_WAVES * 64), 0, stream, h[0], h[1], h[2], X,
                                       ldx, Hcat, dX, lddx, n, M3Rows{}, (float *)scratch);
                CGS_CHECK_HIP(hipGetLastError());
            }
            CgsProfScope prof(CGS_PROF_MLP_WGRAD, stream);
            const CgsWgProduct prods[4] = {{nullptr, 0, M3_GLD, nullptr, 0, M3_IN, dW1cat, db1cat},
                                           {nullptr, 0, 10, nullptr, 0, M3_HID, dW2[0], db2[0]},
                                           {nullptr, 0, 30, nullptr, 0, M3_HID, dW2[1], db2[1]},
                                           {nullptr, 0, 70, nullptr, 0, M3_HID, dW2[2], db2[2]}};
            return cgs_launch_wgrad_reduce((const float *)scratch, gridw, prods, 4, stream);
        }
    }
#endif
    if (no_x) { cgs_set_error("anchor_mlp3_backward: X == NULL but the fused weight-gradient form could not run (scratch)"); return CGS_ERR_WORKSPACE; }
    if (tiled) {
        cgs_set_error("anchor_mlp3_backward: the tiled hand-over is read by the fused weight-gradient form only (needs the weight-gradient pointers, "
                      "<= %lld rows and its scratch)", (long long)M3_MAX_ROWS);
        return CGS_ERR_ARG;
    }
    const int64_t tiles = (n + 16 * RT - 1) / (16 * RT);
    const int64_t want = (tiles + WAVES - 1) / WAVES;
    const int grid = (int)(want < m3_cus() ? want : m3_cus());
    {
        CgsProfScope prof(CGS_PROF_MLP_BWD, stream);
        if (rows)
            hipLaunchKernelGGL((mlp3_bwd_kernel<10, 1, 30, 2, 70, 0, RT, WAVES, true>), dim3(grid), dim3(WAVES * 64), 0, stream,
                               h[0], h[1], h[2], Hcat, dZ1cat, nullptr, 0, n, *rows);
        else
            hipLaunchKernelGGL((mlp3_bwd_kernel<10, 1, 30, 2, 70, 0, RT, WAVES, false>), dim3(grid), dim3(WAVES * 64), 0, stream,
                               h[0], h[1], h[2], Hcat, dZ1cat, dX, lddx, n, M3Rows{});
        CGS_CHECK_HIP(hipGetLastError());
    }
    if (data_only) return CGS_OK;
    return cgs_anchor_mlp3_wgrad(X, ldx, Hcat, dZ1cat, dZ2_op, dZ2_color, dY_cov, dW1cat, db1cat, dW2, db2, n, scratch,
                                 scratch_bytes, stream_);
}

// The weight-gradient launch of cgs_anchor_mlp3_backward as a call of its own (after a backward with dW1cat == NULL):
// X [n, ldx] (the forward's X_out for the _rows pair, ldx = cgs_anchor_mlp3_layout()[1]), dZ1cat / dZ2_op / dZ2_color as the
// backward left them, dY_cov the incoming gradient of the covariance head (its second layer has no activation).
extern "C" int cgs_anchor_mlp3_wgrad(const float *X, int64_t ldx, const float *Hcat, const float *dZ1cat,
                                     const float *dZ2_op, const float *dZ2_color, const float *dY_cov, float *dW1cat,
                                     float *db1cat, float *const *dW2, float *const *db2, int64_t n, void *scratch,
                                     size_t scratch_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0) { cgs_set_error("anchor_mlp3_wgrad: n < 0"); return CGS_ERR_ARG; }
    if (n == 0) return CGS_OK;
    if (!X || !Hcat || !dZ1cat || !dZ2_op || !dZ2_color || !dY_cov || !dW1cat || !db1cat || !dW2 || !db2) {
        cgs_set_error("anchor_mlp3_wgrad: NULL");
        return CGS_ERR_ARG;
    }
    CgsProfScope prof(CGS_PROF_MLP_WGRAD, stream);
    // the three first layers as one [150 x 54] product and the three second layers, all in ONE launch over the
    // same rows (7 wave tasks): an Hcat / dZ1cat row is pulled from HBM once
    const CgsWgProduct prods[4] = {
        {dZ1cat, M3_GLD, M3_GLD, X, ldx, M3_IN, dW1cat, db1cat},       // dW1cat [192, 54]: head h = rows 64 h .. 64 h + 49
        {dZ2_op, 10, 10, Hcat, M3_HLD, M3_HID, dW2[0], db2[0]},
        {dZ2_color, 30, 30, Hcat + M3_HPITCH, M3_HLD, M3_HID, dW2[1], db2[1]},
        {dY_cov, 70, 70, Hcat + 2 * M3_HPITCH, M3_HLD, M3_HID, dW2[2], db2[2]}};
    return cgs_launch_wgrad_multi(prods, 4, n, m3_cus(), scratch, scratch_bytes, stream);
}
