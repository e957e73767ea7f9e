// Row / fragment helpers shared by the fused level kernels (ctx_level.hip) and the fused rate-subset kernels (rate_sub.hip).
#pragma once
#include "cgs_internal.h"
#include "mlp_frag.h"
#include "buf_access.h"

#define CL_D 50
#define CL_S 6
#define CL_O 30
#define CL_HY 12
#define CL_HID 100
#define CL_NT1 7
#define CL_HP 112
#define CL_S1 116                 // frag_pad4mod8(112)

template <int IN>
struct ClShape {
    static constexpr int NTI = (IN + 15) / 16, XP = NTI * 16;
    static constexpr int SB = frag_pad4mod8(XP);
};

// no IR-level motion of the loads (memory clobber) and no machine-scheduler motion (sched_barrier) across
#ifndef CL_PIPE
#define CL_PIPE 1             // LDS operands of the next MFMA group are read before the current group is issued
#endif
#define CL_FENCE() CLB_FENCE()
#define CLB_FENCE()                        \
    do {                                   \
        asm volatile("" ::: "memory");     \
        __builtin_amdgcn_sched_barrier(0); \
    } while (0)

// a row of [n, IN] (X, dX, dx_sub) as fragments: q < NTI - 1 (or IN 15) full pieces, the last tile's pieces end at column IN
template <int IN>
__device__ __forceinline__ void cl_xrow_load(ClBuf b, uint32_t rowoff, int g, bool on, f32x4 (&x)[ClShape<IN>::NTI]) {
    constexpr int NTI = ClShape<IN>::NTI;
#pragma unroll
    for (int q = 0; q < NTI; ++q) {
        const int col0 = 16 * q + 4 * g;                  // (lane dependent through g)
        x[q] = cl_l128(b, cl_sel(on && col0 < IN, rowoff + (uint32_t)col0 * 4));
    }
}
// a 16-byte piece that starts inside the row but ends behind it carries the next row's first values: zero them at use
template <int IN>
__device__ __forceinline__ void cl_xrow_mask(int g, f32x4 (&x)[ClShape<IN>::NTI]) {
    constexpr int q = ClShape<IN>::NTI - 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) x[q][j] = 16 * q + 4 * g + j >= IN ? 0.f : x[q][j];      // (selects: a branch here puts the row in scratch)
}
template <int IN>
__device__ __forceinline__ void cl_xrow_store(ClBuf b, uint32_t rowoff, int g, bool on, const f32x4 (&x)[ClShape<IN>::NTI]) {
    constexpr int NTI = ClShape<IN>::NTI, R = IN - 16 * (NTI - 1);       // columns of the last tile: 7 (IN 71) or 15 (IN 15)
#pragma unroll
    for (int q = 0; q < NTI - 1; ++q) cl_s128(b, cl_sel(on, rowoff + (uint32_t)(16 * q + 4 * g) * 4), x[q]);
    const uint32_t o = rowoff + (uint32_t)(16 * (NTI - 1) + 4 * g) * 4;
    // lanes whose piece lies wholly inside the row store 16 bytes; the lane holding the row's last 3 columns stores 12
    cl_s128(b, cl_sel(on && 4 * g + 3 < R, o), x[NTI - 1]);
    cl_s96(b, cl_sel(on && 4 * g + 3 == R, o), x[NTI - 1]);
}

// ---- a lane's share of the 86 parameter values of a row (features 50, scaling 6, offsets 30) -------------------
// 16-byte pieces, so that a wave instruction touches 16 rows x 64 contiguous bytes:
//   F[i], i < 3: feature columns 16 i + 4 g + {0..3};  F[3]: columns 48, 49 (lane g == 0 only)
//   S: scaling columns 0..3 (g == 1) / 4, 5 (g == 2);  O[i]: offset columns 16 i + 4 g + {0..3} (O[1] of g == 3: 28, 29)
struct ClRow { f32x4 F[4], S, O[2]; };
struct ClRowBufs { ClBuf f, s, o; };

// loads as issued: the pieces of lanes that own only two of their four values carry the next row's first values in [2], [3]
__device__ __forceinline__ void cl_row_issue(ClRow &v, const ClRowBufs &B, uint32_t row, int g, bool on) {
    const uint32_t of = row * (CL_D * 4), os = row * (CL_S * 4), oo = row * (CL_O * 4);
#pragma unroll
    for (int i = 0; i < 3; ++i) v.F[i] = cl_l128(B.f, cl_sel(on, of + (uint32_t)(16 * i + 4 * g) * 4));
    v.F[3] = cl_l64(B.f, cl_sel(on && g == 0, of + 48 * 4));
    v.S = cl_l128(B.s, cl_sel(on && (g == 1 || g == 2), os + (g == 2 ? 16u : 0u)));
    v.O[0] = cl_l128(B.o, cl_sel(on, oo + (uint32_t)(4 * g) * 4));
    v.O[1] = cl_l128(B.o, cl_sel(on, oo + (uint32_t)(16 + 4 * g) * 4));
}
__device__ __forceinline__ void cl_row_mask(ClRow &v, int g) {
    v.S[2] = g == 2 ? 0.f : v.S[2];
    v.S[3] = g == 2 ? 0.f : v.S[3];
    v.O[1][2] = g == 3 ? 0.f : v.O[1][2];
    v.O[1][3] = g == 3 ? 0.f : v.O[1][3];
}
__device__ __forceinline__ void cl_row_store(const ClRow &v, const ClRowBufs &B, uint32_t row, int g, bool on) {
    const uint32_t of = row * (CL_D * 4), os = row * (CL_S * 4), oo = row * (CL_O * 4);
#pragma unroll
    for (int i = 0; i < 3; ++i) cl_s128(B.f, cl_sel(on, of + (uint32_t)(16 * i + 4 * g) * 4), v.F[i]);
    cl_s64(B.f, cl_sel(on && g == 0, of + 48 * 4), v.F[3]);
    cl_s128(B.s, cl_sel(on && g == 1, os), v.S);
    cl_s64(B.s, cl_sel(on && g == 2, os + 16), v.S);
    cl_s128(B.o, cl_sel(on, oo + (uint32_t)(4 * g) * 4), v.O[0]);
    cl_s128(B.o, cl_sel(on && g != 3, oo + (uint32_t)(16 + 4 * g) * 4), v.O[1]);
    cl_s64(B.o, cl_sel(on && g == 3, oo + 28 * 4), v.O[1]);
}

// The same rows (and Q) of a whole 16-row tile through a per-wave LDS patch: the three tile images — 16 x 50, 16 x 6, 16 x 30
// floats, each contiguous in its array — are assembled in the patch and leave as contiguous 16-byte-per-lane stores
// (buf_access.h frag_tile_store; rows behind the end of an array are dropped by its bounds check).  c = the lane's row of the tile.
#define CL_ROW_PATCH (16 * CL_D)          // floats: the feature image; scaling + offsets + Q (16 x (6 + 30 + 4)) reuse it afterwards
__device__ __forceinline__ void cl_row_store_tile(const ClRow &v, f32x4 q3, const ClRowBufs &B, ClBuf bQ, float *patch, uint32_t row0,
                                                  int g, int c, int lane) {
    float *pf = patch;
#pragma unroll
    for (int i = 0; i < 3; ++i) cl_patch_put4(pf + c * CL_D + 16 * i + 4 * g, v.F[i], 4);
    if (g == 0) cl_patch_put4(pf + c * CL_D + 48, v.F[3], 2);
    cl_patch_flush<64 * CL_D>(pf, B.f, row0 * (CL_D * 4), lane);
    float *ps = patch, *po = ps + 16 * CL_S, *pq = po + 16 * CL_O;          // (LDS operations of a wave execute in order)
    if (g == 1) cl_patch_put4(ps + c * CL_S, v.S, 4);
    if (g == 2) cl_patch_put4(ps + c * CL_S + 4, v.S, 2);
    cl_patch_put4(po + c * CL_O + 4 * g, v.O[0], 4);
    cl_patch_put4(po + c * CL_O + 16 + 4 * g, v.O[1], g == 3 ? 2 : 4);
    if (g == 0) cl_patch_put4(pq + c * 3, q3, 3);
    cl_patch_flush<64 * CL_S>(ps, B.s, row0 * (CL_S * 4), lane);
    cl_patch_flush<64 * CL_O>(po, B.o, row0 * (CL_O * 4), lane);
    cl_patch_flush<64 * 3>(pq, bQ, row0 * 12, lane);
}
// a 16-row tile of [n, IN] (X) the same way: 64 * IN contiguous bytes
template <int IN>
__device__ __forceinline__ void cl_xrow_store_tile(ClBuf b, float *patch, uint32_t row0, int g, int c, int lane,
                                                   const f32x4 (&x)[ClShape<IN>::NTI]) {
    constexpr int NTI = ClShape<IN>::NTI;
#pragma unroll
    for (int q = 0; q < NTI; ++q) {
        const int col0 = 16 * q + 4 * g;
        if (q < NTI - 1) cl_patch_put4(patch + c * IN + col0, x[q], 4);
        else if (col0 < IN) cl_patch_put4(patch + c * IN + col0, x[q], IN - col0 < 4 ? IN - col0 : 4);
    }
    cl_patch_flush<64 * IN>(patch, b, row0 * (IN * 4), lane);
}

__device__ __forceinline__ float cl_sum4(f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }
