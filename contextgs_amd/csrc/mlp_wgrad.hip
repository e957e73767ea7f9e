// Weight / bias gradients of the fused MLPs:  dW[a][b] += sum_rows P[row][a] Q[row][b],
// db[a] += sum_rows P[row][a]   (P = dZ [n, DA], Q = layer input [n, DB], both row-major in HBM).
//
// fp32 MFMA 16x16x4 with the ROW index as the contraction dimension: one instruction consumes 4
// rows (lane (g, c) supplies row 4q + g).  The output-tile indices m and n are free to permute,
// so a wave works on a 64-feature GROUP of P against a 64-feature group of Q as 4 x 4 tiles in
// which tile j's index c stands for feature 64*group + 4c + j: one 16-byte load per lane (a wave
// instruction = 4 rows x 256 contiguous bytes) feeds four tiles, i.e. 2 loads per 16 MFMAs.
// The next 4*UNR rows are prefetched into a second register set while the current ones are in
// the matrix pipe.  (History: v1 scattered tiles over waves, 2 dword loads per MFMA, L1-bound;
// v2 register-blocked UA x UB tiles with dword loads and no prefetch ran latency-bound at
// ~4x its HBM time — profiles/r01_rocprof_bench_1m_v1_mfma_mlp.txt.)
//
// The 8 waves of a workgroup take (group pair, row sub-range); rows are split over workgroups;
// each wave ends with one fp32 atomic per owned output element.  The bias gradient is a plain
// per-lane running sum of the P fragments, reduced over g at the end.
#include "cgs_internal.h"
#include "mlp_frag.h"

typedef int i32x4 __attribute__((ext_vector_type(4)));

// One operand stream of a workgroup: a raw buffer descriptor over rows [r_begin, r_end) that ends exactly at the
// last valid element, so rows past the range (tail iterations, the prefetch) and the overhang of the last row read
// as 0 from the hardware bounds check — no branches around the loads, which is what lets the prefetch overlap.
struct WgStream {
    __amdgpu_buffer_rsrc_t rsrc;
    int ld4;            // row stride in bytes
    int col4;           // this lane's first column, in bytes
    bool m[4];          // column col0 + j < dim  (a 16-byte access that crosses the row end picks up the next row)
    bool partial;       // wave-uniform: this 64-feature group crosses the row end
    bool narrow;        // wave-uniform: the group holds <= 16 features — lane c carries feature 64 group + c alone (ONE tile)
};

__device__ __forceinline__ WgStream wg_stream(const float *base, int64_t ld, int dim, int group, int c, int64_t r_begin,
                                              int64_t r_end, bool narrow = false) {
    WgStream st;
    st.narrow = narrow;
    const float *p = base + r_begin * ld;
    const int64_t rows = r_end - r_begin;
    const uint32_t bytes = rows > 0 ? (uint32_t)(((rows - 1) * ld + dim) * 4) : 0u;
    st.rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, bytes, 0x00020000);
    st.ld4 = (int)ld * 4;
    const int col0 = narrow ? 64 * group + c : 64 * group + 4 * c;
    st.col4 = col0 * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) st.m[j] = (!narrow || j == 0) && col0 + j < dim;
    st.partial = narrow || 64 * group + 63 >= dim;
    return st;
}

__device__ __forceinline__ f32x4 wg_load4(const WgStream &st, int rel_row) {
    if (st.narrow) {      // one feature per lane: a 4-byte access (components 1..3 are masked off at consume time)
        const int raw = __builtin_amdgcn_raw_buffer_load_b32(st.rsrc, rel_row * st.ld4 + st.col4, 0, 0);
        return (f32x4){__builtin_bit_cast(float, raw), 0.f, 0.f, 0.f};
    }
    const i32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(st.rsrc, rel_row * st.ld4 + st.col4, 0, 0);
    return __builtin_bit_cast(f32x4, raw);
}

// applied at consume time: a use next to the load would put the wait there and defeat the prefetch
__device__ __forceinline__ f32x4 wg_mask(const WgStream &st, f32x4 v) {
    if (st.partial) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = st.m[j] ? v[j] : 0.f;
    }
    return v;
}

template <int UNR>
struct WgFrag {
    f32x4 a[UNR], b[UNR];
};

template <int UNR>
__device__ __forceinline__ void wg_fetch(WgFrag<UNR> &f, const WgStream &sa, const WgStream &sb, int rel_row0, int g) {
#pragma unroll
    for (int q = 0; q < UNR; ++q) {
        f.a[q] = wg_load4(sa, rel_row0 + 4 * q + g);
        f.b[q] = wg_load4(sb, rel_row0 + 4 * q + g);
    }
}

template <int UNR>
__device__ __forceinline__ void wg_consume(const WgFrag<UNR> &f, const WgStream &sa, const WgStream &sb,
                                           f32x4 (&acc)[4][4], f32x4 &bsum) {
#pragma unroll
    for (int q = 0; q < UNR; ++q) {
        const f32x4 a = wg_mask(sa, f.a[q]), b = wg_mask(sb, f.b[q]);
        // a narrow operand fills tile index 0 only: 4 (or 1) MFMAs instead of 16 for a <= 16-feature group
#pragma unroll
        for (int ja = 0; ja < 4; ++ja) {
            if (ja > 0 && sa.narrow) break;
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) {
                if (jb > 0 && sb.narrow) break;
                acc[ja][jb] = frag_mfma(a[ja], b[jb], acc[ja][jb]);
            }
        }
        bsum += a;
    }
}

// no IR-level motion of the loads (memory clobber) and no machine-scheduler motion (sched_barrier) across
#define WG_FENCE()                        \
    do {                                  \
        asm volatile("" ::: "memory");    \
        __builtin_amdgcn_sched_barrier(0); \
    } while (0)

#define WG_MAX_E (175 * 100 + 175)   // largest dW + db image (mlp_grid second layer)

// partial == NULL: every wave adds its tiles to dW/db with global fp32 atomics.
// partial != NULL: the waves of a workgroup combine in an LDS image of [dW | db] (ds_add_f32), the
// workgroup stores it to partial[blockIdx.x][E] and wgrad_reduce_kernel sums the images — no global
// atomics (same-line atomics from the 8 XCDs' L2s were the dominant cost of the atomic variant) and a
// run-to-run deterministic summation order.
template <int UNR>
__global__ void __launch_bounds__(512)
    wgrad4_kernel(const float *__restrict__ P, int64_t ldp, int DA, const float *__restrict__ Q, int64_t ldq, int DB,
                  float *__restrict__ dW, float *__restrict__ db, int64_t n, int64_t rows_per_block, int GB,
                  int npairs, int nsub, float *__restrict__ partial) {
    __shared__ float img[WG_MAX_E];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: scalar loop control
    const int pair = wave % npairs, sub = wave / npairs;
    const bool active = sub < nsub;
    const int E = DA * DB + DA;
    if (partial) {
        for (int i = tid; i < E; i += 512) img[i] = 0.f;
        __syncthreads();
    }
    const int ga = pair / GB, gb = pair % GB;
    const int acol0 = 64 * ga + 4 * c, bcol0 = 64 * gb + 4 * c;
    if (active) {
        const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
        const int64_t r_end = min(n, r_begin + rows_per_block);
        f32x4 acc[4][4], bsum = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ja = 0; ja < 4; ++ja)
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) acc[ja][jb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const WgStream sa = wg_stream(P, ldp, DA, ga, c, r_begin, r_end);
        const WgStream sb = wg_stream(Q, ldq, DB, gb, c, r_begin, r_end);
        const int rows = (int)(r_end - r_begin), stride = nsub * 4 * UNR;
        int row0 = sub * 4 * UNR;
        WgFrag<UNR> f0, f1;
        wg_fetch<UNR>(f0, sa, sb, row0, g);
        // Straight-line double-buffered body (no mid-loop exit: rows past the range load as zeros, so a surplus
        // half-iteration is harmless) with fences, or the prefetch gets sunk next to its first use.
        for (; row0 < rows; row0 += 2 * stride) {
            wg_fetch<UNR>(f1, sa, sb, row0 + stride, g);
            WG_FENCE();
            wg_consume<UNR>(f0, sa, sb, acc, bsum);
            WG_FENCE();
            wg_fetch<UNR>(f0, sa, sb, row0 + 2 * stride, g);
            WG_FENCE();
            wg_consume<UNR>(f1, sa, sb, acc, bsum);
            WG_FENCE();
        }
        float *const outW = partial ? img : dW;
        float *const outb = partial ? img + DA * DB : db;
        // D layout of tile (ja, jb): reg r of lane (g, c) <-> a = 64 ga + 4 (4g + r) + ja, b = 64 gb + 4c + jb
#pragma unroll
        for (int ja = 0; ja < 4; ++ja)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a = 64 * ga + 4 * (4 * g + r) + ja;
                if (a >= DA) continue;
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) {
                    const int b = bcol0 + jb;
                    if (b < DB) atomicAdd(&outW[a * DB + b], acc[ja][jb][r]);
                }
            }
        if (gb == 0 && (partial || db != nullptr)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = bsum[j];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                if (g == 0 && acol0 + j < DA) atomicAdd(&outb[acol0 + j], v);
            }
        }
    }
    if (partial) {
        __syncthreads();
        float *dst = partial + (int64_t)blockIdx.x * E;
        for (int i = tid; i < E; i += 512) dst[i] = img[i];
    }
}

// dW[e] += sum_b partial[b][e]  (e < DA*DB), db[e - DA*DB] += ... ; 64 elements x 4 block-chunks per workgroup
__global__ void __launch_bounds__(256)
    wgrad_reduce_kernel(const float *__restrict__ partial, int blocks, int E, int DADB, float *__restrict__ dW,
                        float *__restrict__ db) {
    __shared__ float s[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + tx;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (e < E) {
        int b = ty;
        for (; b + 12 < blocks; b += 16) {
            a0 += partial[(int64_t)b * E + e];
            a1 += partial[(int64_t)(b + 4) * E + e];
            a2 += partial[(int64_t)(b + 8) * E + e];
            a3 += partial[(int64_t)(b + 12) * E + e];
        }
        for (; b < blocks; b += 4) a0 += partial[(int64_t)b * E + e];
    }
    s[ty][tx] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (ty == 0 && e < E) {
        const float t = (s[0][tx] + s[1][tx]) + (s[2][tx] + s[3][tx]);
        if (e < DADB) dW[e] += t;
        else if (db) db[e - DADB] += t;
    }
}

static int wgrad_blocks_per_cu() {
    static int v = 0;
    if (!v) {
        const char *e = nullptr;
        v = e ? atoi(e) : 1;
        if (v < 1) v = 1;
        if (v > 2) v = 2;
    }
    return v;
}

size_t cgs_wgrad_scratch_bytes_for(int num_cus) { return (size_t)2 * num_cus * 25000 * sizeof(float); }

// scratch (>= cgs_mlp_wgrad_scratch_bytes()) selects the atomics-free two-pass path; NULL the atomic one.
int cgs_launch_wgrad2(const float *P, int64_t ldp, int DA, const float *Q, int64_t ldq, int DB, float *dW, float *db,
                      int64_t n, int num_cus, void *scratch, size_t scratch_bytes, hipStream_t s) {
    if (n <= 0) return CGS_OK;
    constexpr int UNR = 2;
    const int GA = (DA + 63) / 64, GB = (DB + 63) / 64, npairs = GA * GB;
    const int E = DA * DB + DA;
    if (npairs > 8 || E > WG_MAX_E) { cgs_set_error("wgrad: %d x %d not supported", DA, DB); return CGS_ERR_ARG; }
    const int nsub = 8 / npairs;
    int64_t blocks = (n + 255) / 256;
    int64_t cap = (int64_t)(scratch ? wgrad_blocks_per_cu() : 1) * num_cus;
    if (scratch) {
        const int64_t fit = (int64_t)(scratch_bytes / ((size_t)E * sizeof(float)));
        if (fit < 1) { cgs_set_error("wgrad: scratch too small"); return CGS_ERR_ARG; }
        if (cap > fit) cap = fit;
    }
    if (blocks > cap) blocks = cap;
    int64_t rpb = (n + blocks - 1) / blocks;
    const int64_t quantum = (int64_t)nsub * 4 * UNR;
    rpb = (rpb + quantum - 1) / quantum * quantum;
    blocks = (n + rpb - 1) / rpb;
#define WG_LAUNCH(U)                                                                                                  \
    hipLaunchKernelGGL((wgrad4_kernel<U>), dim3((unsigned)blocks), dim3(512), 0, s, P, ldp, DA, Q, ldq, DB, dW, db, n, rpb, \
                       GB, npairs, nsub, (float *)scratch)
    if (UNR == 4) WG_LAUNCH(4);
    else if (UNR == 1) WG_LAUNCH(1);
    else WG_LAUNCH(2);
#undef WG_LAUNCH
    CGS_CHECK_HIP(hipGetLastError());
    if (scratch) {
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((E + 63) / 64)), dim3(256), 0, s, (const float *)scratch,
                           (int)blocks, E, DA * DB, dW, db);
        CGS_CHECK_HIP(hipGetLastError());
    }
    return CGS_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Several weight-gradient products of ONE backward pass (the two layers of an MLP; the first layer and the three
// second layers of the anchor MLPs) over the same rows in ONE launch: every (64-feature group of P) x (64-feature
// group of Q) pair of every product is a task for one wave, all waves of a workgroup walk the same row range, so a
// row's cache lines (e.g. the three 200-byte slices of an Hcat row) are pulled from HBM once, and 2-4 launches +
// reductions become one of each.  Same inner loop and LDS-image / two-pass reduction as wgrad4_kernel.
#ifndef WGM_NARROW
#define WGM_NARROW 1         // one-tile tasks for <= 16-feature groups (tools/variant_lib.sh ... -DWGM_NARROW=0 for the A/B)
#endif
#define WGM_MAX_TASKS 16
#define WGM_MAX_PROD 4
#define WGM_MAX_E 25000      // floats of LDS image: 175x100+175 + 100x71+100 (both layers of mlp_grid) = 24875
struct WgmTask {
    const float *P, *Q;
    int ldp, ldq, DA, DB, ga, gb, img_off;
    int narrow;          // bit 0: the P group holds <= 16 features, bit 1: the Q group does
};
struct WgmArgs {
    WgmTask t[WGM_MAX_TASKS];
    int ntask, nsub, E;
};
struct WgmProducts {
    float *dW[WGM_MAX_PROD], *db[WGM_MAX_PROD];
    int off[WGM_MAX_PROD + 1], dadb[WGM_MAX_PROD];
    int nprod;
};

template <int UNR>
__global__ void __launch_bounds__(1024)
    wgrad_multi_kernel(WgmArgs a, int64_t n, int64_t rows_per_block, float *__restrict__ partial) {
    __shared__ float img[WGM_MAX_E];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < a.E; i += 1024) img[i] = 0.f;
    // wave -> (task, sub-range): wave % ntask, so that a SIMD (waves go to the four SIMDs round-robin) hosts the sub-waves
    // of few tasks — task k in waves k*nsub .. k*nsub + nsub - 1 measured 2-20 % slower.  Every task has the same number of
    // waves: all waves of the workgroup move through the row range at the same pace, which is what lets the tasks that
    // read the same rows share them through the caches.  (Scalar copies: dynamic indexing of the by-value argument
    // struct would go through scratch.)
    const int ti = wave % a.ntask, sub = wave / a.ntask;
    const int nsub = sub < a.nsub ? a.nsub : 0;
    const float *P = nullptr, *Q = nullptr;
    int ldp = 0, ldq = 0, DA = 0, DB = 0, ga = 0, gb = 0, img_off = 0, narrow = 0;
#pragma unroll
    for (int k = 0; k < WGM_MAX_TASKS; ++k)
        if (k == ti) {
            P = a.t[k].P; Q = a.t[k].Q; ldp = a.t[k].ldp; ldq = a.t[k].ldq; DA = a.t[k].DA; DB = a.t[k].DB;
            ga = a.t[k].ga; gb = a.t[k].gb; img_off = a.t[k].img_off; narrow = a.t[k].narrow;
        }
    const bool nA = narrow & 1, nB = narrow & 2;
    f32x4 acc[4][4], bsum = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ja = 0; ja < 4; ++ja)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) acc[ja][jb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (nsub > 0) {
        const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
        const int64_t r_end = min(n, r_begin + rows_per_block);
        const WgStream sa = wg_stream(P, ldp, DA, ga, c, r_begin, r_end, nA);
        const WgStream sb = wg_stream(Q, ldq, DB, gb, c, r_begin, r_end, nB);
        const int rows = (int)(r_end - r_begin), stride = nsub * 4 * UNR;
        int row0 = sub * 4 * UNR;
        WgFrag<UNR> f0, f1;
        wg_fetch<UNR>(f0, sa, sb, row0, g);
        for (; row0 < rows; row0 += 2 * stride) {
            wg_fetch<UNR>(f1, sa, sb, row0 + stride, g);
            WG_FENCE();
            wg_consume<UNR>(f0, sa, sb, acc, bsum);
            WG_FENCE();
            wg_fetch<UNR>(f0, sa, sb, row0 + 2 * stride, g);
            WG_FENCE();
            wg_consume<UNR>(f1, sa, sb, acc, bsum);
            WG_FENCE();
        }
    }
    __syncthreads();      // image zeroed
    // Image update without LDS float atomics (ds_add_f32 retires ~0.6 lanes per clock on gfx950: 16 waves x 68 wave-wide
    // atomics were a fixed ~40 us of every launch): tasks own disjoint parts of the image, and the nsub waves of one task
    // take turns, separated by workgroup barriers, with plain read-add-write.
    for (int s = 0; s < a.nsub; ++s) {            // a.nsub = the largest wave count of a task
        if (nsub > 0 && sub == s) {
            float *const outW = img + img_off;
            float *const outb = outW + DA * DB;
            // D layout of tile (ja, jb): reg r of lane (g, c) <-> P feature 64 ga + 4 (4g + r) + ja (narrow: 64 ga + 4g + r,
            // ja = 0 only), Q feature 64 gb + 4c + jb (narrow: 64 gb + c, jb = 0 only)
#pragma unroll
            for (int ja = 0; ja < 4; ++ja) {
                if (ja > 0 && nA) break;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ar = nA ? 64 * ga + 4 * g + r : 64 * ga + 4 * (4 * g + r) + ja;
                    if (ar >= DA) continue;
#pragma unroll
                    for (int jb = 0; jb < 4; ++jb) {
                        if (jb > 0 && nB) break;
                        const int b = nB ? 64 * gb + c : 64 * gb + 4 * c + jb;
                        if (b < DB) outW[ar * DB + b] += acc[ja][jb][r];
                    }
                }
            }
            if (gb == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j > 0 && nA) break;
                    float v = bsum[j];
                    v += __shfl_xor(v, 16);
                    v += __shfl_xor(v, 32);
                    const int ac = nA ? 64 * ga + c : 64 * ga + 4 * c + j;
                    if (g == 0 && ac < DA) outb[ac] += v;
                }
            }
        }
        __syncthreads();
    }
    float *dst = partial + (int64_t)blockIdx.x * a.E;
    for (int i = tid; i < a.E; i += 1024) dst[i] = img[i];
}

__global__ void __launch_bounds__(256)
    wgrad_multi_reduce_kernel(const float *__restrict__ partial, int blocks, int E, WgmProducts pr, int assign) {
    __shared__ float s[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + tx;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (e < E) {
        int b = ty;
        for (; b + 12 < blocks; b += 16) {
            a0 += partial[(int64_t)b * E + e];
            a1 += partial[(int64_t)(b + 4) * E + e];
            a2 += partial[(int64_t)(b + 8) * E + e];
            a3 += partial[(int64_t)(b + 12) * E + e];
        }
        for (; b < blocks; b += 4) a0 += partial[(int64_t)b * E + e];
    }
    s[ty][tx] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (ty == 0 && e < E) {
        const float t = (s[0][tx] + s[1][tx]) + (s[2][tx] + s[3][tx]);
        int k = 0;
#pragma unroll
        for (int q = 1; q < WGM_MAX_PROD; ++q)
            if (q < pr.nprod && e >= pr.off[q]) k = q;
        float *dW = nullptr, *db = nullptr;
        int off = 0, dadb = 0;
#pragma unroll
        for (int q = 0; q < WGM_MAX_PROD; ++q)
            if (q == k) { dW = pr.dW[q]; db = pr.db[q]; off = pr.off[q]; dadb = pr.dadb[q]; }
        const int le = e - off;
        // assign: the images cover every entry, the destination needs no zero fill
        if (le < dadb) dW[le] = assign ? t : dW[le] + t;
        else if (db) db[le - dadb] = assign ? t : db[le - dadb] + t;
    }
}

// ------------------------------------------------------------------------------------------------------------
// One product whose Q is one full 64-feature group plus a NARROW tail (64 < DB <= 80): the first-layer gradient of the level
// MLPs' step-size branch, dW1 [100 x 71] += dZ1^T X over every row of a level (round 4).  wgrad_multi_kernel gives the 7-feature
// tail a task of its own that issues the same 16 MFMAs per 4 rows as a full 64 x 64 pair (the launch was MFMA-bound on
// padding: 64 MFMAs per 4 rows for 7 100 useful of 16 384 outputs, 248 us at 800 k rows).  Here a task is an A group against
// ALL of Q: the wave that has A's fragment in registers also multiplies it with the tail tile (lane c carries feature 64 + c,
// one 4-byte load), 20 MFMAs per task and 4 rows, 40 for the product instead of 64; the tasks of a launch stay of equal
// weight (what the multi kernel's lock-step walk needs).  512 threads (8 waves: GA tasks x nsub sub-ranges), same LDS-image /
// two-pass reduction and summation order rules as the other weight-gradient kernels (bit-reproducible).
#ifndef WG_TAIL_MIN_WAVES
#define WG_TAIL_MIN_WAVES 1      // 140 VGPRs, three waves per SIMD, no spills; 4 (128 VGPRs, 40 B of scratch) measured 1 % slower
#endif
template <int UNR>
__global__ void __launch_bounds__(512, WG_TAIL_MIN_WAVES)
    wgrad_tail_kernel(const float *__restrict__ P, int64_t ldp, int DA, const float *__restrict__ Q, int64_t ldq, int DB,
                      int64_t n, int64_t rows_per_block, int GA, int nsub, float *__restrict__ partial) {
    __shared__ float img[128 * 80 + 128];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int E = DA * DB + DA;
    for (int i = tid; i < E; i += 512) img[i] = 0.f;
    const int ga = wave % GA, sub = wave / GA;
    const bool active = sub < nsub;
    f32x4 acc[4][4], accT[4], bsum = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ja = 0; ja < 4; ++ja) {
        accT[ja] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) acc[ja][jb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if (active) {
        const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
        const int64_t r_end = min(n, r_begin + rows_per_block);
        const WgStream sa = wg_stream(P, ldp, DA, ga, c, r_begin, r_end);
        const WgStream sb = wg_stream(Q, ldq, DB, 0, c, r_begin, r_end);
        const WgStream st = wg_stream(Q, ldq, DB, 1, c, r_begin, r_end, true);
        const int rows = (int)(r_end - r_begin), stride = nsub * 4 * UNR;
        int row0 = sub * 4 * UNR;
        WgFrag<UNR> f0, f1;
        float t0[UNR], t1[UNR];
        auto fetch = [&](WgFrag<UNR> &f, float (&t)[UNR], int rel) {
            wg_fetch<UNR>(f, sa, sb, rel, g);
#pragma unroll
            for (int q = 0; q < UNR; ++q) t[q] = wg_load4(st, rel + 4 * q + g)[0];
        };
        auto consume = [&](const WgFrag<UNR> &f, const float (&t)[UNR]) {
            wg_consume<UNR>(f, sa, sb, acc, bsum);
#pragma unroll
            for (int q = 0; q < UNR; ++q) {
                const f32x4 a = wg_mask(sa, f.a[q]);
                const float tv = st.m[0] ? t[q] : 0.f;
#pragma unroll
                for (int ja = 0; ja < 4; ++ja) accT[ja] = frag_mfma(a[ja], tv, accT[ja]);
            }
        };
        fetch(f0, t0, row0);
        for (; row0 < rows; row0 += 2 * stride) {
            fetch(f1, t1, row0 + stride);
            WG_FENCE();
            consume(f0, t0);
            WG_FENCE();
            fetch(f0, t0, row0 + 2 * stride);
            WG_FENCE();
            consume(f1, t1);
            WG_FENCE();
        }
    }
    __syncthreads();      // image zeroed
    for (int s = 0; s < nsub; ++s) {              // the sub-waves of a task take turns (plain read-add-write, fixed order)
        if (active && sub == s) {
            float *const outW = img, *const outb = img + DA * DB;
#pragma unroll
            for (int ja = 0; ja < 4; ++ja)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ar = 64 * ga + 4 * (4 * g + r) + ja;
                    if (ar >= DA) continue;
#pragma unroll
                    for (int jb = 0; jb < 4; ++jb) outW[ar * DB + 4 * c + jb] += acc[ja][jb][r];     // 4c + jb < 64 < DB
                    if (64 + c < DB) outW[ar * DB + 64 + c] += accT[ja][r];
                }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = bsum[j];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                const int ac = 64 * ga + 4 * c + j;
                if (g == 0 && ac < DA) outb[ac] += v;
            }
        }
        __syncthreads();
    }
    float *dst = partial + (int64_t)blockIdx.x * E;
    for (int i = tid; i < E; i += 512) dst[i] = img[i];
}

static int launch_wgrad_tail(const CgsWgProduct &p, int64_t n, int num_cus, void *scratch, size_t scratch_bytes, hipStream_t s) {
    constexpr int UNR = 2;
    const int GA = (p.DA + 63) / 64, nsub = 8 / GA, E = p.DA * p.DB + p.DA;
    int64_t blocks = (n + 255) / 256;
    int64_t cap = (int64_t)num_cus * 2;                       // 29 KB of LDS image, 512 threads: two workgroups per CU
    const int64_t fit_blocks = (int64_t)(scratch_bytes / ((size_t)E * sizeof(float)));
    if (cap > fit_blocks) cap = fit_blocks;
    if (blocks > cap) blocks = cap;
    int64_t rpb = (n + blocks - 1) / blocks;
    const int64_t quantum = (int64_t)nsub * 4 * UNR;
    rpb = (rpb + quantum - 1) / quantum * quantum;
    blocks = (n + rpb - 1) / rpb;
    hipLaunchKernelGGL((wgrad_tail_kernel<UNR>), dim3((unsigned)blocks), dim3(512), 0, s, p.P, p.ldp, p.DA, p.Q, p.ldq, p.DB, n, rpb,
                       GA, nsub, (float *)scratch);
    CGS_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((E + 63) / 64)), dim3(256), 0, s, (const float *)scratch, (int)blocks, E,
                       p.DA * p.DB, p.dW, p.db);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

// prods: nprod (<= 4) products over the same n rows.  Returns CGS_ERR_WORKSPACE-free: falls back to one launch per
// product (cgs_launch_wgrad2) when the combination does not fit one workgroup (tasks > 16, LDS image, no scratch).
int cgs_launch_wgrad_multi(const CgsWgProduct *prods, int nprod, int64_t n, int num_cus, void *scratch,
                           size_t scratch_bytes, hipStream_t s) {
    if (n <= 0 || nprod <= 0) return CGS_OK;
#ifndef WG_NO_TAIL_KERNEL
    if (nprod == 1 && scratch && prods[0].DB > 64 && prods[0].DB <= 80 && prods[0].DA <= 128 && prods[0].db &&
        (size_t)(prods[0].DA * prods[0].DB + prods[0].DA) * sizeof(float) <= scratch_bytes)
        return launch_wgrad_tail(prods[0], n, num_cus, scratch, scratch_bytes, s);
#endif
    constexpr bool disabled = false;
    WgmArgs a;
    WgmProducts pr;
    int ntask = 0, E = 0;
    bool fits = scratch != nullptr && nprod <= WGM_MAX_PROD && !disabled;
    for (int k = 0; k < nprod && fits; ++k) {
        const CgsWgProduct &p = prods[k];
        const int GA = (p.DA + 63) / 64, GB = (p.DB + 63) / 64;
        pr.dW[k] = p.dW; pr.db[k] = p.db; pr.off[k] = E; pr.dadb[k] = p.DA * p.DB;
        for (int ga = 0; ga < GA && fits; ++ga)
            for (int gb = 0; gb < GB; ++gb) {
                if (ntask >= WGM_MAX_TASKS) { fits = false; break; }
                const int narrow = (WGM_NARROW && p.DA - 64 * ga <= 16 ? 1 : 0) | (WGM_NARROW && p.DB - 64 * gb <= 16 ? 2 : 0);
                a.t[ntask++] = WgmTask{p.P, p.Q, (int)p.ldp, (int)p.ldq, p.DA, p.DB, ga, gb, E, narrow};
            }
        E += p.DA * p.DB + p.DA;
    }
    if (fits && E > WGM_MAX_E) fits = false;
    if (fits && (size_t)E * sizeof(float) > scratch_bytes) fits = false;
    if (!fits) {
        for (int k = 0; k < nprod; ++k) {
            const CgsWgProduct &p = prods[k];
            int rc = cgs_launch_wgrad2(p.P, p.ldp, p.DA, p.Q, p.ldq, p.DB, p.dW, p.db, n, num_cus, scratch, scratch_bytes, s);
            if (rc) return rc;
        }
        return CGS_OK;
    }
    for (int k = ntask; k < WGM_MAX_TASKS; ++k) a.t[k] = WgmTask{nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0, 0};
    // One-tile (narrow) tasks only when EVERY task of the launch is narrow in the same way: the tasks of a workgroup share
    // their rows through the caches because all waves move through the row range at the same pace; tasks of unequal
    // weight lose that (measured: a narrow minority racing ahead, or wave counts in proportion to the work, made the
    // mixed launches 10-40 % SLOWER), while a launch of equal narrow tasks (mlp_grid[0]'s 100 x 15 product) halves.
    {
        bool uniform = ntask > 0;
        for (int k = 1; k < ntask; ++k) uniform = uniform && a.t[k].narrow == a.t[0].narrow;
        for (int k = 0; k < ntask; ++k)
            if (!uniform) a.t[k].narrow = 0;
        a.nsub = 16 / ntask;
    }
    for (int k = nprod; k < WGM_MAX_PROD; ++k) { pr.dW[k] = nullptr; pr.db[k] = nullptr; pr.off[k] = E; pr.dadb[k] = 0; }
    pr.off[WGM_MAX_PROD] = E;
    pr.nprod = nprod;
#ifndef WGM_UNR
#define WGM_UNR 2
#endif
    constexpr int UNR = WGM_UNR;
    a.ntask = ntask;
    a.E = E;
#ifndef WGM_MIN_ROWS
#define WGM_MIN_ROWS 256         // rows per workgroup at least: a workgroup zeroes, combines and writes a ~100 KB image whatever its rows
#endif
    int64_t blocks = (n + WGM_MIN_ROWS - 1) / WGM_MIN_ROWS;
    int64_t cap = num_cus;
    const int64_t fit_blocks = (int64_t)(scratch_bytes / ((size_t)E * sizeof(float)));
    if (cap > fit_blocks) cap = fit_blocks;
    if (blocks > cap) blocks = cap;
    int64_t rpb = (n + blocks - 1) / blocks;
    const int64_t quantum = (int64_t)a.nsub * 4 * UNR;
    rpb = (rpb + quantum - 1) / quantum * quantum;
    blocks = (n + rpb - 1) / rpb;
    hipLaunchKernelGGL((wgrad_multi_kernel<UNR>), dim3((unsigned)blocks), dim3(1024), 0, s, a, n, rpb, (float *)scratch);
    CGS_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(wgrad_multi_reduce_kernel, dim3((unsigned)((E + 63) / 64)), dim3(256), 0, s, (const float *)scratch,
                       (int)blocks, E, pr, 0);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

// Sum `blocks` per-workgroup images laid out [dW | db] per product (the layout of WgmProducts) into the products'
// gradient buffers: the reduction half of the two-pass scheme, for kernels that build the images themselves
// (anchor_gen.hip).
static int wgrad_reduce_launch(const float *partial, int blocks, const CgsWgProduct *prods, int nprod, int assign, hipStream_t s);
int cgs_launch_wgrad_reduce(const float *partial, int blocks, const CgsWgProduct *prods, int nprod, hipStream_t s) {
    return wgrad_reduce_launch(partial, blocks, prods, nprod, 0, s);
}
// the same, ASSIGNING the sums (the images cover every entry of the products: no zero fill of the destination)
int cgs_launch_wgrad_reduce_assign(const float *partial, int blocks, const CgsWgProduct *prods, int nprod, hipStream_t s) {
    return wgrad_reduce_launch(partial, blocks, prods, nprod, 1, s);
}
static int wgrad_reduce_launch(const float *partial, int blocks, const CgsWgProduct *prods, int nprod, int assign, hipStream_t s) {
    if (nprod <= 0 || nprod > WGM_MAX_PROD || blocks <= 0) { cgs_set_error("wgrad_reduce: bad args"); return CGS_ERR_ARG; }
    WgmProducts pr;
    int E = 0;
    for (int k = 0; k < nprod; ++k) {
        pr.dW[k] = prods[k].dW; pr.db[k] = prods[k].db; pr.off[k] = E; pr.dadb[k] = prods[k].DA * prods[k].DB;
        E += prods[k].DA * prods[k].DB + prods[k].DA;
    }
    for (int k = nprod; k < WGM_MAX_PROD; ++k) { pr.dW[k] = nullptr; pr.db[k] = nullptr; pr.off[k] = E; pr.dadb[k] = 0; }
    pr.off[WGM_MAX_PROD] = E;
    pr.nprod = nprod;
    hipLaunchKernelGGL(wgrad_multi_reduce_kernel, dim3((unsigned)((E + 63) / 64)), dim3(256), 0, s, partial, blocks, E, pr, assign);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}
