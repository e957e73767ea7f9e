// Fragment helpers shared by the fused-MLP kernels (mlp.hip, mlp3.hip).
//
// Every contraction in these kernels walks its index in the order (tile q, element j):
// MFMA k-step (q, j) contracts the four indices {16q + 4g + j : g = 0..3}.  That is the
// accumulator (D) layout of v_mfma_f32_16x16x4_f32 — lane (g, c) holds features
// 16q + 4g + {0..3} of row c — so (1) one layer's accumulators are the next layer's B operand
// and (2) every activation access is ONE 16-byte access per lane: a wave instruction touches
// 16 rows x 64 contiguous bytes.  (The first version used k = 4s + g for the layers fed from
// memory: 16 rows x 16 B per instruction; with ~0.7 MB of activations in flight per CU the
// partially used lines were evicted before their neighbours were read and the kernels ran at
// 4-8x their HBM bound — profiles/r01_rocprof_bench_1m_v1_mfma_mlp.txt.)
//
// Row strides here are 3..175 floats, so the 16-byte accesses are only 4-byte aligned; gfx950
// under HSA runs in unaligned-access mode and the compiler emits global_load/store_dwordx4 for
// the align(4) vector type below.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef f32x4 f32x4_a4 __attribute__((aligned(4)));

__device__ __forceinline__ f32x4 frag_mfma(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// columns [col0, col0+4) of a row with DIM valid columns; col0 = 16q + 4g
template <int DIM>
__device__ __forceinline__ f32x4 frag_load4(const float *__restrict__ rowp, int q, int g, bool valid) {
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int col0 = 16 * q + 4 * g;
    if (valid) {
        if (16 * q + 15 < DIM || col0 + 3 < DIM) {
            v = *(const f32x4_a4 *)(rowp + col0);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (col0 + j < DIM) v[j] = rowp[col0 + j];
        }
    }
    return v;
}

template <int DIM>
__device__ __forceinline__ void frag_store4(float *__restrict__ rowp, int q, int g, bool valid, f32x4 v) {
    const int col0 = 16 * q + 4 * g;
    if (valid) {
        if (16 * q + 15 < DIM || col0 + 3 < DIM) {
            *(f32x4_a4 *)(rowp + col0) = v;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (col0 + j < DIM) rowp[col0 + j] = v[j];
        }
    }
}

// smallest s >= x with s % 8 == 4: A-fragment rows 16q + 4g + j are 4 apart, so g = 0,1 (one
// 32-lane LDS pass) land 16 banks apart
constexpr int frag_pad4mod8(int x) { int s = 4; while (s < x) s += 8; return s; }

// LDS staging of a weight matrix in TRANSPOSED form: dst [RP][S] (row stride S), dst[r][c] = src[c][r] for r < ROWS, c < COLS
// (src row-major [COLS][ROWS], e.g. nn.Linear's weight [out][in] -> [in][out]), zeros in the padding.  The source is read in
// ITS order — consecutive threads, consecutive addresses — and scattered into LDS; reading it in the destination's order is a
// 4-byte gather with a row-length stride (64 cache lines per wave load).  Measured (round 4, same-box A/B of the headline
// step, FRAG_STAGE_COALESCED=0 = the old loop): 129 vs 130-131 us per forward launch — the weights are L2-resident and the
// gather was ~1 us of a launch, not the ~10 us it was suspected of.
#ifndef FRAG_STAGE_COALESCED
#define FRAG_STAGE_COALESCED 1
#endif
// A staging loop  for (i = tid; i < n; i += nthr) put(i, ok(i) ? src[index(i)] : 0)  with EIGHT of a thread's loads in flight.
// As a plain loop the compiler emits load -> s_waitcnt vmcnt(0) -> LDS store per round (it does not unroll a loop of unknown trip
// count, and a predicated load sits behind an exec-mask branch whose join drains the load queue): one exposed round trip per
// round — 80 rounds = ~50 us at the head of mlp3_bwd_wg_kernel, as much again over the level and rate kernels of a step (round 6;
// rate_sub.hip found it first).  Here the load is UNCONDITIONAL (index 0 where the value is padding) and selected afterwards.
//   at(i) -> source index, or -1 for padding (value 0)
template <typename AT, typename PUT>
__device__ __forceinline__ void frag_stage_loop(const float *__restrict__ src, int n, int tid, int nthr, AT &&at, PUT &&put) {
    constexpr int UN = 8;
    for (int i = tid; i < n; i += UN * nthr) {          // (a short tail rides in the same batch: its slots beyond n load index 0 and put nothing)
        float v[UN];
        int ix[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int j = i + u * nthr;
            ix[u] = j < n ? at(j) : -1;
            v[u] = src[ix[u] < 0 ? 0 : ix[u]];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int j = i + u * nthr;
            if (j < n) put(j, ix[u] < 0 ? 0.f : v[u]);
        }
    }
}

template <int ROWS, int COLS, int RP, int S>
__device__ __forceinline__ void frag_stage_transposed(float *__restrict__ dst, const float *__restrict__ src, int tid, int nthr) {
#if FRAG_STAGE_COALESCED
    for (int i = tid; i < RP * S; i += nthr) {
        const int r = i / S, c = i % S;
        if (r >= ROWS || c >= COLS) dst[i] = 0.f;
    }
    frag_stage_loop(src, COLS * ROWS, tid, nthr, [](int i) { return i; },
                    [&](int i, float v) { const int c = i / ROWS, r = i % ROWS; dst[r * S + c] = v; });
#else
    for (int i = tid; i < RP * S; i += nthr) {
        const int r = i / S, c = i % S;
        dst[i] = (r < ROWS && c < COLS) ? src[c * ROWS + r] : 0.f;
    }
#endif
}


#define FRAG_ACT_NONE 0
#define FRAG_ACT_TANH 1
#define FRAG_ACT_SIGMOID 2

template <int ACT>
__device__ __forceinline__ float frag_act(float z) {
    if (ACT == FRAG_ACT_TANH) return tanhf(z);
    if (ACT == FRAG_ACT_SIGMOID) return 1.f / (1.f + __expf(-z));
    return z;
}
template <int ACT>
__device__ __forceinline__ float frag_act_grad(float y) {
    if (ACT == FRAG_ACT_TANH) return 1.f - y * y;
    if (ACT == FRAG_ACT_SIGMOID) return y * (1.f - y);
    return 1.f;
}
