// Fused element-wise stages of the per-level context model (scene/gaussian_model.py:1556-1707), training path.
// The level loop of the reference is ~120 small torch kernels per level and direction (gathers, cats, uniform_,
// tanh/clamp chains, masked sums); rocprof of the 1 M-anchor step shows them as ~8 ms of 5-25 us launches
// (profiles/r01_rocprof_bench_1m_v1_mfma_mlp.txt).  Three HBM-bound kernels per direction replace them:
//
//   rowcat        out[r] = [src0[idx0[r]] | src1[idx1[r]] | ...]        the MLP input of a level (:1594-1600,
//                                                                       :1650-1651, :1711-1724)
//   noise_quant   Q = clamp(Q0 (1 + tanh(q)), 1e-9); y = x + U(-.5,.5) Q  (:1603-1616) for feat/scaling/offsets
//   level_rate    sum of the discretised-Gaussian bits of the chosen rows (:1658-1669 + utils/entropy_models.py:30-50)
//
// All three are pure streaming kernels: their roofline is HBM bytes (rows x widths x 4 B).
#include <initializer_list>
#include <type_traits>
#include "cgs_internal.h"
#include "rate_math.h"
#include "ctx_noise.h"

// ------------------------------------------------------------------------------------------------------------
#define RC_MAX_SRC 4
#ifndef RC_LDS
#define RC_LDS 1          // LDS-staged multi-source rowcat forward (tools/variant_lib.sh ... -DRC_LDS=0 for the A/B)
#endif
struct RowcatArgs {
    const float *src[RC_MAX_SRC];
    float *dsrc[RC_MAX_SRC];
    const int64_t *idx[RC_MAX_SRC];
    const uint8_t *rmask[RC_MAX_SRC];      // optional: source row r counts as src[r] * (rmask[r] != 0)
    int width[RC_MAX_SRC], ld[RC_MAX_SRC], mode[RC_MAX_SRC], begin[RC_MAX_SRC];
    int nsrc, W;
};

__global__ void __launch_bounds__(256) rowcat_fwd_kernel(RowcatArgs a, int64_t n, float *__restrict__ out) {
    const int64_t total = n * a.W;
    const int64_t stride = (int64_t)gridDim.x * 256;
    // four elements per trip: their row-index loads, then their (dependent) source loads, are issued together
    for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < total; i0 += 4 * stride) {
        const float *p[4];
        const uint8_t *mk[4];
        int64_t row[4], mrow[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + u * stride;
            ok[u] = i < total;
            const int64_t r = ok[u] ? i / a.W : 0;
            const int c = ok[u] ? (int)(i - r * a.W) : 0;
            int s = 0;
#pragma unroll
            for (int k = 1; k < RC_MAX_SRC; ++k)
                if (k < a.nsrc && c >= a.begin[k]) s = k;
            row[u] = (ok[u] && a.idx[s]) ? a.idx[s][r] : r;
            p[u] = a.src[s] + (c - a.begin[s]);
            mk[u] = a.rmask[s];
            mrow[u] = row[u];
            row[u] *= a.ld[s];
        }
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v[u] = ok[u] ? p[u][row[u]] : 0.f;
            if (ok[u] && mk[u]) v[u] *= mk[u][mrow[u]] ? 1.f : 0.f;       // a product, like the reference's anchor * mask (-0.0 stays -0.0)
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (ok[u]) out[i0 + u * stride] = v[u];
    }
}

// The same through LDS: a workgroup assembles RC_ROWS output rows in shared memory (sources whose rows are 8-byte
// aligned and even-sized are pulled as float2 pairs) and streams the finished block out as float4 — the output rows are
// W floats wide with W odd in practice (71 = 3 + 50 + 6 + 12), so their own alignment is 4 bytes, but a block of 32 rows
// starts 16-byte aligned.  55 wide memory instructions per 71-float row instead of 142 scalar ones.
#define RC_ROWS 32
__global__ void __launch_bounds__(256) rowcat_fwd_lds_kernel(RowcatArgs a, int64_t n, float *__restrict__ out, unsigned pair_mask) {
    extern __shared__ __align__(16) float rc_lds[];
    const int W = a.W, tid = threadIdx.x;
    for (int64_t r0 = (int64_t)blockIdx.x * RC_ROWS; r0 < n; r0 += (int64_t)gridDim.x * RC_ROWS) {
        const int R = (int)((n - r0) < RC_ROWS ? (n - r0) : RC_ROWS);
#pragma unroll
        for (int s = 0; s < RC_MAX_SRC; ++s) {
            if (s >= a.nsrc) break;
            const int w = a.width[s], ld = a.ld[s], b = a.begin[s];
            const float *src = a.src[s];
            const int64_t *idx = a.idx[s];
            const uint8_t *mk = a.rmask[s];
            if (pair_mask >> s & 1) {
                const int w2 = w >> 1, items = R * w2;
                const float inv = 1.f / (float)w2;
                for (int it = tid; it < items; it += 256) {
                    const int rl = (int)(((float)it + 0.5f) * inv), c2 = it - rl * w2;
                    const int64_t row = idx ? idx[r0 + rl] : r0 + rl;
                    float2 v = ((const float2 *)(src + row * ld))[c2];
                    if (mk) { const float m = mk[row] ? 1.f : 0.f; v.x *= m; v.y *= m; }
                    rc_lds[rl * W + b + 2 * c2] = v.x;
                    rc_lds[rl * W + b + 2 * c2 + 1] = v.y;
                }
            } else {
                const int items = R * w;
                const float inv = 1.f / (float)w;
                for (int it = tid; it < items; it += 256) {
                    const int rl = (int)(((float)it + 0.5f) * inv), c = it - rl * w;
                    const int64_t row = idx ? idx[r0 + rl] : r0 + rl;
                    float v = src[row * ld + c];
                    if (mk) v *= mk[row] ? 1.f : 0.f;
                    rc_lds[rl * W + b + c] = v;
                }
            }
        }
        __syncthreads();
        const int total = R * W;
        float *dst = out + r0 * W;                          // 16-byte aligned: r0 is a multiple of 32
        for (int q = tid; q < (total >> 2); q += 256) ((float4 *)dst)[q] = ((const float4 *)rc_lds)[q];
        for (int q = (total & ~3) + tid; q < total; q += 256) dst[q] = rc_lds[q];
        __syncthreads();
    }
}

// One source, rows of whole float4s (e.g. the [N,12] hyper latents' gradient back through the coding permutation): out[r] =
// src[idx[r]] with 16-byte accesses.
__global__ void __launch_bounds__(256) rowgather4_kernel(const float4 *__restrict__ src, const int64_t *__restrict__ idx, int64_t n,
                                                         int w4, int ld4, float4 *__restrict__ out) {
    const int64_t total = n * w4;
    const bool small = total < (1ll << 32);                      // 32-bit division (an emulated 64-bit one is ~5x the instructions)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = small ? (int64_t)((uint32_t)i / (uint32_t)w4) : i / w4;
        const int c4 = (int)(i - r * w4);
        out[i] = src[idx[r] * ld4 + c4];
    }
}

static unsigned stream_grid(int64_t total, int per_block);

// out[a, 0:w] = row idx[a] (NULL: a) of the VIRTUAL concatenation of up to RC_MAX_SRC row blocks: block k holds rows
// begin[k] .. begin[k+1] - 1 as src[k] + (r - begin[k]) * ld[k] (ld in floats; src[k] == NULL: zeros).  The gradient of the
// hyper latents arrives as one block per level — two of them strided column slices of the levels' input-row gradients — and
// leaves through the inverse coding permutation: one pass instead of cat + copy + gather.
struct SegGatherArgs {
    const float *src[RC_MAX_SRC];
    int64_t ld[RC_MAX_SRC], begin[RC_MAX_SRC + 1];
    int nseg;
};

__global__ void __launch_bounds__(256) gather_rows_segmented_kernel(SegGatherArgs a, const int64_t *__restrict__ idx, int64_t n, int w,
                                                                    float *__restrict__ out) {
    const int64_t total = n * w;
    const bool small = total < (1ll << 32);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = small ? (int64_t)((uint32_t)i / (uint32_t)w) : i / w;
        const int c = (int)(i - row * w);
        const int64_t r = idx ? idx[row] : row;
        int k = 0;
#pragma unroll
        for (int q = 1; q < RC_MAX_SRC; ++q)
            if (q < a.nseg && r >= a.begin[q]) k = q;
        const float *base = nullptr;
        int64_t ld = 0, b0 = 0;
#pragma unroll
        for (int q = 0; q < RC_MAX_SRC; ++q)
            if (q == k) { base = a.src[q]; ld = a.ld[q]; b0 = a.begin[q]; }
        out[i] = base ? base[(r - b0) * ld + c] : 0.f;
    }
}

extern "C" int cgs_gather_rows_segmented(int nseg, const float *const *src, const int64_t *ld, const int64_t *begin,
                                         const int64_t *idx, int64_t n, int w, float *out, void *stream) {
    if (nseg < 1 || nseg > RC_MAX_SRC || !src || !ld || !begin || n < 0 || w < 1) { cgs_set_error("gather_rows_segmented: bad args"); return CGS_ERR_ARG; }
    if (n == 0) return CGS_OK;
    if (!out) { cgs_set_error("gather_rows_segmented: NULL out"); return CGS_ERR_ARG; }
    SegGatherArgs a;
    a.nseg = nseg;
    for (int k = 0; k < RC_MAX_SRC; ++k) {
        a.src[k] = k < nseg ? src[k] : nullptr;
        a.ld[k] = k < nseg ? ld[k] : 0;
        a.begin[k] = k < nseg ? begin[k] : begin[nseg];
        if (k < nseg && (begin[k + 1] < begin[k] || (src[k] && ld[k] < w))) { cgs_set_error("gather_rows_segmented: bad block %d", k); return CGS_ERR_ARG; }
    }
    a.begin[RC_MAX_SRC] = begin[nseg];
    CgsProfScope prof(CGS_PROF_CTX_BWD, (hipStream_t)stream);
    hipLaunchKernelGGL(gather_rows_segmented_kernel, dim3(stream_grid(n * w, 256 * 4)), dim3(256), 0, (hipStream_t)stream, a, idx, n, w, out);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

// mode 0: no gradient; 1: store (rows of this source are distinct, or identity); 2: atomic add (rows repeat)
__global__ void __launch_bounds__(256) rowcat_bwd_kernel(RowcatArgs a, int64_t n, const float *__restrict__ dout) {
    const int64_t total = n * a.W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / a.W;
        const int c = (int)(i - r * a.W);
        int s = 0;
#pragma unroll
        for (int k = 1; k < RC_MAX_SRC; ++k)
            if (k < a.nsrc && c >= a.begin[k]) s = k;
        if (a.mode[s] == 0) continue;
        const int64_t row = a.idx[s] ? a.idx[s][r] : r;
        float *p = a.dsrc[s] + row * a.ld[s] + (c - a.begin[s]);
        float gv = dout[i];
        if (a.rmask[s]) gv *= a.rmask[s][row] ? 1.f : 0.f;
        if (a.mode[s] == 1) *p = gv;
        else atomicAdd(p, gv);
    }
}

static int rowcat_fill(RowcatArgs &a, int nsrc, const void *const *data, const int64_t *const *idx, const int *width,
                       const int *ld, const int *mode, bool bwd, const uint8_t *const *rmask = nullptr) {
    if (nsrc < 1 || nsrc > RC_MAX_SRC || !data || !width || !ld) { cgs_set_error("rowcat: bad source list"); return CGS_ERR_ARG; }
    a.nsrc = nsrc;
    int W = 0;
    for (int s = 0; s < RC_MAX_SRC; ++s) {
        const bool on = s < nsrc;
        a.src[s] = on && !bwd ? (const float *)data[s] : nullptr;
        a.dsrc[s] = on && bwd ? (float *)data[s] : nullptr;
        a.idx[s] = on && idx ? idx[s] : nullptr;
        a.rmask[s] = on && rmask ? rmask[s] : nullptr;
        a.width[s] = on ? width[s] : 0;
        a.ld[s] = on ? ld[s] : 0;
        a.mode[s] = on && mode ? mode[s] : 0;
        a.begin[s] = W;
        if (on) {
            if (width[s] < 1 || ld[s] < width[s]) { cgs_set_error("rowcat: bad width/ld of source %d", s); return CGS_ERR_ARG; }
            if (!data[s] && (!bwd || (mode && mode[s]))) { cgs_set_error("rowcat: NULL source %d", s); return CGS_ERR_ARG; }
            W += width[s];
        }
    }
    a.W = W;
    return CGS_OK;
}

static unsigned stream_grid(int64_t total, int per_block) {
    int64_t b = (total + per_block - 1) / per_block;
    const int64_t cap = 256 * 16;
    return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

extern "C" int cgs_rowcat_fwd_masked(int nsrc, const void *const *data, const int64_t *const *idx, const uint8_t *const *rowmask,
                                     const int *width, const int *ld, int64_t n, float *out, void *stream);
extern "C" int cgs_rowcat_fwd(int nsrc, const void *const *data, const int64_t *const *idx, const int *width,
                              const int *ld, int64_t n, float *out, void *stream) {
    return cgs_rowcat_fwd_masked(nsrc, data, idx, nullptr, width, ld, n, out, stream);
}

extern "C" int cgs_rowcat_fwd_masked(int nsrc, const void *const *data, const int64_t *const *idx, const uint8_t *const *rowmask,
                                     const int *width, const int *ld, int64_t n, float *out, void *stream) {
    if (n < 0) { cgs_set_error("rowcat_fwd: n < 0"); return CGS_ERR_ARG; }
    RowcatArgs a;
    int rc = rowcat_fill(a, nsrc, data, idx, width, ld, nullptr, false, rowmask);
    if (rc) return rc;
    if (n == 0) return CGS_OK;
    if (!out) { cgs_set_error("rowcat_fwd: NULL out"); return CGS_ERR_ARG; }
    CgsProfScope prof(CGS_PROF_CTX_FWD, (hipStream_t)stream);
    if (nsrc == 1 && a.idx[0] && !a.rmask[0] && !(a.W & 3) && !(a.ld[0] & 3) && !(((uintptr_t)a.src[0] | (uintptr_t)out) & 15) &&
        n * (a.W / 4) < (1ll << 40)) {
        hipLaunchKernelGGL(rowgather4_kernel, dim3(stream_grid(n * (a.W / 4), 256 * 2)), dim3(256), 0, (hipStream_t)stream,
                           (const float4 *)a.src[0], a.idx[0], n, a.W / 4, a.ld[0] / 4, (float4 *)out);
        CGS_CHECK_HIP(hipGetLastError());
        return CGS_OK;
    }
    // wide rows of several sources (the context rows of a level): staged through LDS; single narrow gathers: element-wise
    if (RC_LDS && nsrc >= 2 && a.W >= 16 && a.W <= 256 && n >= 4 * RC_ROWS && !((uintptr_t)out & 15)) {
        unsigned pair_mask = 0;
        for (int s = 0; s < nsrc; ++s)
            if (!(a.width[s] & 1) && !(a.ld[s] & 1) && !((uintptr_t)a.src[s] & 7)) pair_mask |= 1u << s;
        const int64_t blocks = (n + RC_ROWS - 1) / RC_ROWS;
        hipLaunchKernelGGL(rowcat_fwd_lds_kernel, dim3((unsigned)(blocks < 256 * 8 ? blocks : 256 * 8)), dim3(256),
                           (size_t)RC_ROWS * a.W * sizeof(float), (hipStream_t)stream, a, n, out, pair_mask);
    } else {
        hipLaunchKernelGGL(rowcat_fwd_kernel, dim3(stream_grid(n * a.W, 256 * 4)), dim3(256), 0, (hipStream_t)stream, a, n, out);
    }
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

extern "C" int cgs_rowcat_bwd_masked(int nsrc, void *const *ddata, const int64_t *const *idx, const uint8_t *const *rowmask,
                                     const int *width, const int *ld, const int *mode, int64_t n, const float *dout, void *stream);
extern "C" int cgs_rowcat_bwd(int nsrc, void *const *ddata, const int64_t *const *idx, const int *width, const int *ld,
                              const int *mode, int64_t n, const float *dout, void *stream) {
    return cgs_rowcat_bwd_masked(nsrc, ddata, idx, nullptr, width, ld, mode, n, dout, stream);
}

extern "C" int cgs_rowcat_bwd_masked(int nsrc, void *const *ddata, const int64_t *const *idx, const uint8_t *const *rowmask,
                                     const int *width, const int *ld, const int *mode, int64_t n, const float *dout, void *stream) {
    if (n < 0 || !mode) { cgs_set_error("rowcat_bwd: bad args"); return CGS_ERR_ARG; }
    RowcatArgs a;
    int rc = rowcat_fill(a, nsrc, (const void *const *)ddata, idx, width, ld, mode, true, rowmask);
    if (rc) return rc;
    if (n == 0) return CGS_OK;
    if (!dout) { cgs_set_error("rowcat_bwd: NULL dout"); return CGS_ERR_ARG; }
    CgsProfScope prof(CGS_PROF_CTX_BWD, (hipStream_t)stream);
    hipLaunchKernelGGL(rowcat_bwd_kernel, dim3(stream_grid(n * a.W, 256 * 4)), dim3(256), 0, (hipStream_t)stream, a, n, dout);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

// out[idx[i]] = g[i] for ASCENDING distinct row indices, every other row of out = 0, in one pass: the thread that writes
// element c of row idx[i] also clears element c of the rows between idx[i-1] and idx[i] (and, for the last index, of the
// rows behind it).  The backward of x[visible rows] (gaussian_renderer/__init__.py:44-50): a zero fill of [N, w] plus an
// index_copy_ become one launch that writes every output line once.  Meant for dense index sets (the visible anchors are
// ~all anchors); the caller falls back to fill + scatter when fewer than 1/8 of the rows are listed.
__global__ void __launch_bounds__(256)
    scatter_rows_sorted_kernel(const float *__restrict__ g, const int64_t *__restrict__ idx, int64_t n, int64_t N, int w,
                               float *__restrict__ out) {
    // a workgroup takes 256 / w consecutive list entries per trip (32-bit index arithmetic; w <= 256)
    const int rpb = 256 / w;
    const int li = (int)threadIdx.x / w, c = (int)threadIdx.x - li * w;
    if (li >= rpb) return;
    for (int64_t i = (int64_t)blockIdx.x * rpb + li; i < n; i += (int64_t)gridDim.x * rpb) {
        const int64_t r = idx[i];
        const int64_t prev = i > 0 ? idx[i - 1] : -1;
        for (int64_t z = prev + 1; z < r; ++z) out[z * w + c] = 0.f;
        out[r * w + c] = g[i * w + c];
        if (i == n - 1)
            for (int64_t z = r + 1; z < N; ++z) out[z * w + c] = 0.f;
    }
}

// The same pass with wider lanes (the one-float-per-lane form spends its time on index loads and 64-bit address
// arithmetic, two index loads per float): V = 2 moves float2 pairs of even-width rows (8-byte aligned: w even), ROW = a lane
// owns a whole row of w <= 4 floats.
template <int V>
__global__ void __launch_bounds__(256)
    scatter_rows_sorted_vec_kernel(const float *__restrict__ g, const int64_t *__restrict__ idx, int64_t n, int64_t N, int wv,
                                   float *__restrict__ out) {
    typedef typename std::conditional<V == 2, float2, float4>::type T;
    const int rpb = 256 / wv;
    const int li = (int)threadIdx.x / wv, c = (int)threadIdx.x - li * wv;
    if (li >= rpb) return;
    const T zero = {};
    for (int64_t i = (int64_t)blockIdx.x * rpb + li; i < n; i += (int64_t)gridDim.x * rpb) {
        const int64_t r = idx[i];
        const int64_t prev = i > 0 ? idx[i - 1] : -1;
        for (int64_t z = prev + 1; z < r; ++z) ((T *)out)[z * wv + c] = zero;
        ((T *)out)[r * wv + c] = ((const T *)g)[i * wv + c];
        if (i == n - 1)
            for (int64_t z = r + 1; z < N; ++z) ((T *)out)[z * wv + c] = zero;
    }
}

template <int W>
__global__ void __launch_bounds__(256)
    scatter_rows_sorted_row_kernel(const float *__restrict__ g, const int64_t *__restrict__ idx, int64_t n, int64_t N,
                                   float *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = idx[i];
        const int64_t prev = i > 0 ? idx[i - 1] : -1;
        float v[W];
#pragma unroll
        for (int c = 0; c < W; ++c) v[c] = g[i * W + c];
        for (int64_t z = prev + 1; z < r; ++z)
#pragma unroll
            for (int c = 0; c < W; ++c) out[z * W + c] = 0.f;
#pragma unroll
        for (int c = 0; c < W; ++c) out[r * W + c] = v[c];
        if (i == n - 1)
            for (int64_t z = r + 1; z < N; ++z)
#pragma unroll
                for (int c = 0; c < W; ++c) out[z * W + c] = 0.f;
    }
}

// out[idx[i], :] += g[i, :] for DISTINCT rows idx (any order): one lane per element, plain read-modify-write — the rows are
// distinct, so no two lanes meet (torch's index_add_ takes float atomics per element: 14 us for 150 k rows of 10-12 floats)
__global__ void __launch_bounds__(256)
    add_rows_kernel(const float *__restrict__ g, const int64_t *__restrict__ idx, int64_t n, int w, float *__restrict__ out) {
    const int64_t total = n * w;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / w;
        const int c = (int)(e - r * w);
        float *q = out + idx[r] * w + c;
        *q += g[e];
    }
}

extern "C" int cgs_add_rows(const float *g, const int64_t *idx, int64_t n, int64_t N, int w, float *out, void *stream) {
    if (n < 0 || N < 0 || w < 1 || w > 256) { cgs_set_error("add_rows: bad sizes (1 <= w <= 256)"); return CGS_ERR_ARG; }
    if (n == 0) return CGS_OK;
    if (!g || !idx || !out) { cgs_set_error("add_rows: NULL"); return CGS_ERR_ARG; }
    CgsProfScope prof(CGS_PROF_CTX_BWD, (hipStream_t)stream);
    hipLaunchKernelGGL(add_rows_kernel, dim3(stream_grid(n * w, 256 * 4)), dim3(256), 0, (hipStream_t)stream, g, idx, n, w, out);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

extern "C" int cgs_scatter_rows_sorted(const float *g, const int64_t *idx, int64_t n, int64_t N, int w, float *out,
                                       void *stream) {
    if (n < 0 || N < n || w < 1 || w > 256) { cgs_set_error("scatter_rows_sorted: bad sizes (1 <= w <= 256)"); return CGS_ERR_ARG; }
    if (N == 0) return CGS_OK;
    if (!out) { cgs_set_error("scatter_rows_sorted: NULL out"); return CGS_ERR_ARG; }
    if (n == 0) {
        CGS_CHECK_HIP(hipMemsetAsync(out, 0, (size_t)N * w * sizeof(float), (hipStream_t)stream));
        return CGS_OK;
    }
    if (!g || !idx) { cgs_set_error("scatter_rows_sorted: NULL input"); return CGS_ERR_ARG; }
    CgsProfScope prof(CGS_PROF_CTX_BWD, (hipStream_t)stream);
    const bool al8 = (((uintptr_t)g | (uintptr_t)out) & 7) == 0, al16 = (((uintptr_t)g | (uintptr_t)out) & 15) == 0;
    if (w == 3)
        hipLaunchKernelGGL(scatter_rows_sorted_row_kernel<3>, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, g,
                           idx, n, N, out);
    else if (w == 1)
        hipLaunchKernelGGL(scatter_rows_sorted_row_kernel<1>, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, g,
                           idx, n, N, out);
    else if (w % 4 == 0 && al16)
        hipLaunchKernelGGL(scatter_rows_sorted_vec_kernel<4>, dim3(stream_grid(n, (256 / (w / 4)) * 2)), dim3(256), 0,
                           (hipStream_t)stream, g, idx, n, N, w / 4, out);
    else if (w % 2 == 0 && al8)
        hipLaunchKernelGGL(scatter_rows_sorted_vec_kernel<2>, dim3(stream_grid(n, (256 / (w / 2)) * 2)), dim3(256), 0,
                           (hipStream_t)stream, g, idx, n, N, w / 2, out);
    else
        hipLaunchKernelGGL(scatter_rows_sorted_kernel, dim3(stream_grid(n, (256 / w) * 4)), dim3(256), 0, (hipStream_t)stream, g,
                           idx, n, N, w, out);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

// out[i] = x[idx[i]] for narrow rows (w <= 256 floats): the forward of the unique-row gathers of the step (visible anchors'
// positions [N,3], masks [N,10], the rate subset's rows; gaussian_renderer/__init__.py:44-50) — torch's index_select takes its
// generic gather kernel for these shapes (29 us per call at 1 M rows, one 64-bit index load and one address computation per
// FLOAT); a lane per row for w = 1 / 3, float2 / float4 lanes for even widths (the forms of cgs_scatter_rows_sorted).
template <int W>
__global__ void __launch_bounds__(256)
    gather_rows_row_kernel(const float *__restrict__ x, const int64_t *__restrict__ idx, int64_t n, float *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = idx[i];
        float v[W];
#pragma unroll
        for (int c = 0; c < W; ++c) v[c] = x[r * W + c];
#pragma unroll
        for (int c = 0; c < W; ++c) out[i * W + c] = v[c];
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
    gather_rows_vec_kernel(const T *__restrict__ x, const int64_t *__restrict__ idx, int64_t n, int wv, T *__restrict__ out) {
    const int rpb = 256 / wv;
    const int li = (int)threadIdx.x / wv, c = (int)threadIdx.x - li * wv;
    if (li >= rpb) return;
    for (int64_t i = (int64_t)blockIdx.x * rpb + li; i < n; i += (int64_t)gridDim.x * rpb) out[i * wv + c] = x[idx[i] * wv + c];
}

extern "C" int cgs_gather_rows(const float *x, const int64_t *idx, int64_t n, int w, float *out, void *stream) {
    if (n < 0 || w < 1 || w > 256) { cgs_set_error("gather_rows: bad sizes (1 <= w <= 256)"); return CGS_ERR_ARG; }
    if (n == 0) return CGS_OK;
    if (!x || !idx || !out) { cgs_set_error("gather_rows: NULL"); return CGS_ERR_ARG; }
    CgsProfScope prof(CGS_PROF_CTX_FWD, (hipStream_t)stream);
    const bool al8 = (((uintptr_t)x | (uintptr_t)out) & 7) == 0, al16 = (((uintptr_t)x | (uintptr_t)out) & 15) == 0;
    if (w == 3)
        hipLaunchKernelGGL(gather_rows_row_kernel<3>, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, x, idx, n, out);
    else if (w == 1)
        hipLaunchKernelGGL(gather_rows_row_kernel<1>, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, x, idx, n, out);
    else if (w % 4 == 0 && al16)
        hipLaunchKernelGGL(gather_rows_vec_kernel<float4>, dim3(stream_grid(n, (256 / (w / 4)) * 2)), dim3(256), 0, (hipStream_t)stream,
                           (const float4 *)x, idx, n, w / 4, (float4 *)out);
    else if (w % 2 == 0 && al8)
        hipLaunchKernelGGL(gather_rows_vec_kernel<float2>, dim3(stream_grid(n, (256 / (w / 2)) * 2)), dim3(256), 0, (hipStream_t)stream,
                           (const float2 *)x, idx, n, w / 2, (float2 *)out);
    else
        hipLaunchKernelGGL(gather_rows_vec_kernel<float>, dim3(stream_grid(n, (256 / w) * 4)), dim3(256), 0, (hipStream_t)stream, x, idx,
                           n, w, out);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Rows NOT in a list get zeros.  A backward that writes only the rows a view touched (rows idx[0..n) of an [n_full, w]
// gradient buffer, distinct) used to start from torch.zeros: a full-buffer fill (144 MB + 200 MB per training view at 1 M
// anchors for the 99.5 % of rows that are overwritten right after).  Instead: every listed row gets the call's generation
// number in a persistent uint32 stamp array (cgs_mark_rows; no clearing between calls: generations only grow, the caller
// restarts at a zeroed array after 2^32 - 1), and cgs_zero_unmarked_rows writes zeros to the rows whose stamp is older.
__global__ void __launch_bounds__(256)
    mark_rows_kernel(const int64_t *__restrict__ idx, int64_t n, int64_t n_full, uint32_t gen, uint32_t *__restrict__ stamp) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = idx[i];
        if (r >= 0 && r < n_full) stamp[r] = gen;          // (an index outside the array is ignored, not written through)
    }
}

struct ZeroRowsArgs { float *dst[4]; int w[4]; int narr; };

__global__ void __launch_bounds__(256)
    zero_unmarked_rows_kernel(const uint32_t *__restrict__ stamp, uint32_t gen, int64_t n_full, ZeroRowsArgs a) {
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n_full; r += (int64_t)gridDim.x * 256) {
        if (stamp[r] == gen) continue;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < a.narr) {
                float *q = a.dst[k] + r * a.w[k];
                for (int c = 0; c < a.w[k]; ++c) q[c] = 0.f;
            }
    }
}

extern "C" int cgs_mark_rows(const int64_t *idx, int64_t n, int64_t n_full, uint32_t gen, uint32_t *stamp, void *stream) {
    if (n < 0 || n_full < 0 || gen == 0) { cgs_set_error("mark_rows: bad args (gen must be > 0)"); return CGS_ERR_ARG; }
    if (n == 0) return CGS_OK;
    if (!idx || !stamp) { cgs_set_error("mark_rows: NULL"); return CGS_ERR_ARG; }
    hipLaunchKernelGGL(mark_rows_kernel, dim3(stream_grid(n, 256 * 4)), dim3(256), 0, (hipStream_t)stream, idx, n, n_full, gen, stamp);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

extern "C" int cgs_zero_unmarked_rows(const uint32_t *stamp, uint32_t gen, int64_t n_full, int narr, float *const *dst,
                                      const int *width, void *stream) {
    if (n_full < 0 || narr < 1 || narr > 4 || !dst || !width) { cgs_set_error("zero_unmarked_rows: bad args (1..4 arrays)"); return CGS_ERR_ARG; }
    if (n_full == 0) return CGS_OK;
    if (!stamp) { cgs_set_error("zero_unmarked_rows: NULL"); return CGS_ERR_ARG; }
    ZeroRowsArgs a;
    a.narr = narr;
    for (int k = 0; k < 4; ++k) {
        a.dst[k] = k < narr ? dst[k] : nullptr;
        a.w[k] = k < narr ? width[k] : 0;
        if (k < narr && (!dst[k] || width[k] < 1)) { cgs_set_error("zero_unmarked_rows: array %d", k); return CGS_ERR_ARG; }
    }
    hipLaunchKernelGGL(zero_unmarked_rows_kernel, dim3(stream_grid(n_full, 256 * 4)), dim3(256), 0, (hipStream_t)stream, stamp, gen,
                       n_full, a);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

// ------------------------------------------------------------------------------------------------------------
#ifndef NQ_VEC2
#define NQ_VEC2 1         // float2 forms of the noise_quant kernels (tools/variant_lib.sh ... -DNQ_VEC2=0 for the A/B)
#endif
// 16 lanes per row: a wave instruction touches 4 rows x 64 contiguous bytes of each tensor.
// sums (may be NULL): double [3], += the sums of the SOURCE values read (features, scaling, offsets): the levels of a
// step together read every row of the three parameter tensors exactly once, which makes these the numerators of the
// rate model's three clamp centres (scene/gaussian_model.py:1664-1668) without a separate pass over 344 B per anchor.
__global__ void __launch_bounds__(256)
    noise_quant_fwd_kernel(const float *__restrict__ xf, const float *__restrict__ xs, const float *__restrict__ xo,
                           const float *__restrict__ qadj, const int64_t *__restrict__ rows, int64_t n, int D, int S,
                           int O, uint64_t seed, float q0f, float q0s, float q0o, float *__restrict__ yf,
                           float *__restrict__ ys, float *__restrict__ yo, float *__restrict__ Q,
                           double *__restrict__ sums) {
    const int l = threadIdx.x & 15;
    const uint32_t kf = ctx_noise_key(seed, 0), ks = ctx_noise_key(seed, 1), ko = ctx_noise_key(seed, 2);
    float pf = 0.f, ps = 0.f, po = 0.f;
    for (int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4; r < n; r += ((int64_t)gridDim.x * 256) >> 4) {
        // (every lane evaluates the three step sizes itself: lanes 0..2 computing them once and a ds_bpermute broadcast to the
        //  16 lanes of the row measured SLOWER, 44 vs 41 us per ctx_fwd launch — the tanhf is not what these kernels wait for)
        const float qf = ctx_step(q0f, qadj[r * 3 + 0]), qs = ctx_step(q0s, qadj[r * 3 + 1]),
                    qo = ctx_step(q0o, qadj[r * 3 + 2]);
        if (l < 3) Q[r * 3 + l] = l == 0 ? qf : (l == 1 ? qs : qo);
        const int64_t sr = rows ? rows[r] : r;        // source row: the level's slice of the coding-order permutation
        for (int c = l; c < D; c += 16) { const float v = xf[sr * D + c]; pf += v; yf[r * D + c] = v + ctx_noise_k(kf, (uint64_t)r * D + c) * qf; }
        for (int c = l; c < S; c += 16) { const float v = xs[sr * S + c]; ps += v; ys[r * S + c] = v + ctx_noise_k(ks, (uint64_t)r * S + c) * qs; }
        for (int c = l; c < O; c += 16) { const float v = xo[sr * O + c]; po += v; yo[r * O + c] = v + ctx_noise_k(ko, (uint64_t)r * O + c) * qo; }
    }
    if (sums) {        // one atomic per block and quantity, spread over CTX_SUM_SLOTS cache lines (same-address atomics serialise)
        __shared__ double part[4][3];
        const double a = ctx_wave_sum((double)pf), b = ctx_wave_sum((double)ps), c = ctx_wave_sum((double)po);
        if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6][0] = a; part[threadIdx.x >> 6][1] = b; part[threadIdx.x >> 6][2] = c; }
        __syncthreads();
        if (threadIdx.x < 3) {
            const int k = threadIdx.x;
            atomicAdd(&sums[(blockIdx.x % CTX_SUM_SLOTS) * CTX_SUM_STRIDE + k], part[0][k] + part[1][k] + part[2][k] + part[3][k]);
        }
    }
}

// The same with 8-byte accesses (even D, S, O with D <= 64, S <= 32, O <= 32 — the reference's 50 / 6 / 30 — and 8-byte
// aligned bases): a lane moves float2 pairs, 4 loads + 4 stores per row pass instead of 7 + 7, all loads issued before the
// first store.  Same element -> noise mapping, so the same bits as the scalar kernel.
__global__ void __launch_bounds__(256)
    noise_quant_fwd2_kernel(const float *__restrict__ xf, const float *__restrict__ xs, const float *__restrict__ xo,
                            const float *__restrict__ qadj, const int64_t *__restrict__ rows, int64_t n, int D, int S,
                            int O, uint64_t seed, float q0f, float q0s, float q0o, float *__restrict__ yf,
                            float *__restrict__ ys, float *__restrict__ yo, float *__restrict__ Q,
                            double *__restrict__ sums) {
    const int l = threadIdx.x & 15;
    const int D2 = D >> 1, S2 = S >> 1, O2 = O >> 1;
    const uint32_t kf = ctx_noise_key(seed, 0), ks = ctx_noise_key(seed, 1), ko = ctx_noise_key(seed, 2);
    float pf = 0.f, ps = 0.f, po = 0.f;
    for (int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4; r < n; r += ((int64_t)gridDim.x * 256) >> 4) {
        const int64_t sr = rows ? rows[r] : r;        // source row: the level's slice of the coding-order permutation
        const float2 *sf = (const float2 *)(xf + sr * D), *ss = (const float2 *)(xs + sr * S), *so = (const float2 *)(xo + sr * O);
        const bool f1 = l + 16 < D2, hs = l < S2, ho = l < O2;
        const float2 z = make_float2(0.f, 0.f);
        const float2 a0 = l < D2 ? sf[l] : z, a1 = f1 ? sf[l + 16] : z, b = hs ? ss[l] : z, c = ho ? so[l] : z;
        // (every lane evaluates the three step sizes itself: lanes 0..2 computing them once and a ds_bpermute broadcast to the
        //  16 lanes of the row measured SLOWER, 44 vs 41 us per ctx_fwd launch — the tanhf is not what these kernels wait for)
        const float qf = ctx_step(q0f, qadj[r * 3 + 0]), qs = ctx_step(q0s, qadj[r * 3 + 1]),
                    qo = ctx_step(q0o, qadj[r * 3 + 2]);
        if (l < 3) Q[r * 3 + l] = l == 0 ? qf : (l == 1 ? qs : qo);
        pf += (a0.x + a0.y) + (a1.x + a1.y);
        ps += b.x + b.y;
        po += c.x + c.y;
        const uint64_t ef = (uint64_t)r * D + 2 * l, es = (uint64_t)r * S + 2 * l, eo = (uint64_t)r * O + 2 * l;
        float2 *df = (float2 *)(yf + r * D), *ds = (float2 *)(ys + r * S), *dd = (float2 *)(yo + r * O);
        if (l < D2) df[l] = make_float2(a0.x + ctx_noise_k(kf, ef) * qf, a0.y + ctx_noise_k(kf, ef + 1) * qf);
        if (f1) df[l + 16] = make_float2(a1.x + ctx_noise_k(kf, ef + 32) * qf, a1.y + ctx_noise_k(kf, ef + 33) * qf);
        if (hs) ds[l] = make_float2(b.x + ctx_noise_k(ks, es) * qs, b.y + ctx_noise_k(ks, es + 1) * qs);
        if (ho) dd[l] = make_float2(c.x + ctx_noise_k(ko, eo) * qo, c.y + ctx_noise_k(ko, eo + 1) * qo);
    }
    if (sums) {
        __shared__ double part[4][3];
        const double a = ctx_wave_sum((double)pf), b = ctx_wave_sum((double)ps), c = ctx_wave_sum((double)po);
        if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6][0] = a; part[threadIdx.x >> 6][1] = b; part[threadIdx.x >> 6][2] = c; }
        __syncthreads();
        if (threadIdx.x < 3) {
            const int k = threadIdx.x;
            atomicAdd(&sums[(blockIdx.x % CTX_SUM_SLOTS) * CTX_SUM_STRIDE + k], part[0][k] + part[1][k] + part[2][k] + part[3][k]);
        }
    }
}

// out3 = (sum over the slots) / counts (float), then the slots are zeroed for the next step's accumulation
__global__ void means_finalize_kernel(double *__restrict__ sums, double ia, double ib, double ic, float *__restrict__ out) {
    const int k = threadIdx.x;
    if (k < 3) {
        double v = 0.0;
        for (int s = 0; s < CTX_SUM_SLOTS; ++s) { v += sums[s * CTX_SUM_STRIDE + k]; sums[s * CTX_SUM_STRIDE + k] = 0.0; }
        out[k] = (float)(v * (k == 0 ? ia : (k == 1 ? ib : ic)));
    }
}

extern "C" size_t cgs_means_accum_doubles(void) { return (size_t)CTX_SUM_SLOTS * CTX_SUM_STRIDE; }

extern "C" int cgs_means_finalize(double *sums3, int64_t na, int64_t nb, int64_t nc, float *out3, void *stream) {
    if (!sums3 || !out3 || na < 0 || nb < 0 || nc < 0) { cgs_set_error("means_finalize: bad args"); return CGS_ERR_ARG; }
    hipLaunchKernelGGL(means_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums3, na ? 1.0 / (double)na : 0.0,
                       nb ? 1.0 / (double)nb : 0.0, nc ? 1.0 / (double)nc : 0.0, out3);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

__device__ __forceinline__ float sum16(float v) {
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    return v;
}

// d_x = d_y: identity, left to the caller — or, when the forward read its rows through `rows`, scattered here into the
// FULL-size gradients dxf/dxs/dxo at those rows (a missing d_y scatters zeros);
// d_qadj[r,k] = (sum_c d_y[r,c] u[r,c] + dQ_ext[r,k]) * dQ/dqadj
// FAST (D <= 64, S <= 16, O <= 32 — the reference's 50 / 6 / 30): the column loops have fixed trip counts (4 / 1 / 2 per
// lane, predicated), so the upstream and side gradients of a row are all loaded before the first store (378 -> 297 us
// per step; the same treatment made the FORWARD slower, 317 -> 337 us, and was not kept there).
template <bool FAST>
__global__ void __launch_bounds__(256)
    noise_quant_bwd_kernel(const float *__restrict__ dyf, const float *__restrict__ dys, const float *__restrict__ dyo,
                           const float *__restrict__ dQ_ext, const float *__restrict__ qadj, int64_t n, int D, int S,
                           int O, uint64_t seed, float q0f, float q0s, float q0o, float *__restrict__ dqadj,
                           const int64_t *__restrict__ rows, float *__restrict__ dxf, float *__restrict__ dxs,
                           float *__restrict__ dxo, const int32_t *__restrict__ side_map,
                           const float *__restrict__ sf, const float *__restrict__ ss, const float *__restrict__ so,
                           const float *__restrict__ sQ) {
    const int l = threadIdx.x & 15;
    const uint32_t kf = ctx_noise_key(seed, 0), ks = ctx_noise_key(seed, 1), ko = ctx_noise_key(seed, 2);
    for (int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4; r < n; r += ((int64_t)gridDim.x * 256) >> 4) {
        float af = 0.f, as = 0.f, ao = 0.f, side_q = 0.f;
        if (rows) {
            const int64_t sr = rows[r];
            // sm >= 0: this row is in the rate subset and its rate gradients sit in row sm of the compact side arrays
            const int64_t sm = side_map ? (int64_t)side_map[r] : -1;
            if (FAST) {
                float gf[4], gs, go[2];
#pragma unroll
                for (int k = 0; k < 4; ++k) { const int c = l + 16 * k; gf[k] = (dyf && c < D) ? dyf[r * D + c] : 0.f; }
                gs = (dys && l < S) ? dys[r * S + l] : 0.f;
#pragma unroll
                for (int k = 0; k < 2; ++k) { const int c = l + 16 * k; go[k] = (dyo && c < O) ? dyo[r * O + c] : 0.f; }
                if (sm >= 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { const int c = l + 16 * k; if (c < D) gf[k] += sf[sm * D + c]; }
                    if (l < S) gs += ss[sm * S + l];
#pragma unroll
                    for (int k = 0; k < 2; ++k) { const int c = l + 16 * k; if (c < O) go[k] += so[sm * O + c]; }
                    if (l < 3) side_q = sQ[sm * 3 + l];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = l + 16 * k;
                    if (c < D) { dxf[sr * D + c] = gf[k]; af += gf[k] * ctx_noise_k(kf, (uint64_t)r * D + c); }
                }
                if (l < S) { dxs[sr * S + l] = gs; as += gs * ctx_noise_k(ks, (uint64_t)r * S + l); }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int c = l + 16 * k;
                    if (c < O) { dxo[sr * O + c] = go[k]; ao += go[k] * ctx_noise_k(ko, (uint64_t)r * O + c); }
                }
            } else {
            for (int c = l; c < D; c += 16) { float g = dyf ? dyf[r * D + c] : 0.f; if (sm >= 0) g += sf[sm * D + c]; dxf[sr * D + c] = g; af += g * ctx_noise_k(kf, (uint64_t)r * D + c); }
            for (int c = l; c < S; c += 16) { float g = dys ? dys[r * S + c] : 0.f; if (sm >= 0) g += ss[sm * S + c]; dxs[sr * S + c] = g; as += g * ctx_noise_k(ks, (uint64_t)r * S + c); }
            for (int c = l; c < O; c += 16) { float g = dyo ? dyo[r * O + c] : 0.f; if (sm >= 0) g += so[sm * O + c]; dxo[sr * O + c] = g; ao += g * ctx_noise_k(ko, (uint64_t)r * O + c); }
            if (sm >= 0 && l < 3) side_q = sQ[sm * 3 + l];
            }
        } else {
            if (dyf) for (int c = l; c < D; c += 16) af += dyf[r * D + c] * ctx_noise_k(kf, (uint64_t)r * D + c);
            if (dys) for (int c = l; c < S; c += 16) as += dys[r * S + c] * ctx_noise_k(ks, (uint64_t)r * S + c);
            if (dyo) for (int c = l; c < O; c += 16) ao += dyo[r * O + c] * ctx_noise_k(ko, (uint64_t)r * O + c);
        }
        af = sum16(af);
        as = sum16(as);
        ao = sum16(ao);
        if (l < 3) {
            const float q0 = l == 0 ? q0f : (l == 1 ? q0s : q0o);
            float g = l == 0 ? af : (l == 1 ? as : ao);
            if (dQ_ext) g += dQ_ext[r * 3 + l];
            g += side_q;
            const float t = tanhf(qadj[r * 3 + l]);
            dqadj[r * 3 + l] = (q0 * (1.f + t) >= 1e-9f) ? g * q0 * (1.f - t * t) : 0.f;
        }
    }
}

static bool nq_vec2_ok(int D, int S, int O, std::initializer_list<const void *> ptrs) {
    if ((D | S | O) & 1 || D > 64 || S > 32 || O > 32) return false;
    for (const void *p : ptrs)
        if ((uintptr_t)p & 7) return false;
    return true;
}

extern "C" int cgs_noise_quant_fwd(const float *xf, const float *xs, const float *xo, const float *qadj,
                                   const int64_t *rows, int64_t n, int D, int S, int O, uint64_t seed, float q0f,
                                   float q0s, float q0o, float *yf, float *ys, float *yo, float *Q, double *sums3,
                                   void *stream) {
    if (n < 0 || D < 1 || S < 1 || O < 1) { cgs_set_error("noise_quant_fwd: bad args"); return CGS_ERR_ARG; }
    if (n == 0) return CGS_OK;
    if (!xf || !xs || !xo || !qadj || !yf || !ys || !yo || !Q) { cgs_set_error("noise_quant_fwd: NULL"); return CGS_ERR_ARG; }
    CgsProfScope prof(CGS_PROF_CTX_FWD, (hipStream_t)stream);
    if (NQ_VEC2 && nq_vec2_ok(D, S, O, {xf, xs, xo, yf, ys, yo}))
        hipLaunchKernelGGL(noise_quant_fwd2_kernel, dim3(stream_grid(n * 16, 256 * 4)), dim3(256), 0, (hipStream_t)stream, xf, xs,
                           xo, qadj, rows, n, D, S, O, seed, q0f, q0s, q0o, yf, ys, yo, Q, sums3);
    else
        hipLaunchKernelGGL(noise_quant_fwd_kernel, dim3(stream_grid(n * 16, 256 * 4)), dim3(256), 0, (hipStream_t)stream, xf, xs,
                           xo, qadj, rows, n, D, S, O, seed, q0f, q0s, q0o, yf, ys, yo, Q, sums3);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

extern "C" int cgs_noise_quant_bwd(const float *dyf, const float *dys, const float *dyo, const float *dQ_ext,
                                   const float *qadj, int64_t n, int D, int S, int O, uint64_t seed, float q0f,
                                   float q0s, float q0o, float *dqadj, const int64_t *rows, float *dxf, float *dxs,
                                   float *dxo, const int32_t *side_map, const float *side_f, const float *side_s,
                                   const float *side_o, const float *side_Q, void *stream) {
    if (n < 0 || D < 1 || S < 1 || O < 1) { cgs_set_error("noise_quant_bwd: bad args"); return CGS_ERR_ARG; }
    if (n == 0) return CGS_OK;
    if (!qadj || !dqadj) { cgs_set_error("noise_quant_bwd: NULL"); return CGS_ERR_ARG; }
    if (rows && (!dxf || !dxs || !dxo)) { cgs_set_error("noise_quant_bwd: rows without dxf/dxs/dxo"); return CGS_ERR_ARG; }
    if (side_map && (!rows || !side_f || !side_s || !side_o || !side_Q)) { cgs_set_error("noise_quant_bwd: side_map needs rows and the four side arrays"); return CGS_ERR_ARG; }
    CgsProfScope prof(CGS_PROF_CTX_BWD, (hipStream_t)stream);
    // (a float2 form of this kernel, like noise_quant_fwd2_kernel, measured 7 % SLOWER: 96 -> 104 us per launch)
    if (D <= 64 && S <= 16 && O <= 32)
        hipLaunchKernelGGL(noise_quant_bwd_kernel<true>, dim3(stream_grid(n * 16, 256 * 4)), dim3(256), 0, (hipStream_t)stream, dyf,
                           dys, dyo, dQ_ext, qadj, n, D, S, O, seed, q0f, q0s, q0o, dqadj, rows, dxf, dxs, dxo, side_map, side_f, side_s, side_o, side_Q);
    else
        hipLaunchKernelGGL(noise_quant_bwd_kernel<false>, dim3(stream_grid(n * 16, 256 * 4)), dim3(256), 0, (hipStream_t)stream, dyf,
                           dys, dyo, dQ_ext, qadj, n, D, S, O, seed, q0f, q0s, q0o, dqadj, rows, dxf, dxs, dxo, side_map, side_f, side_s, side_o, side_Q);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

// ------------------------------------------------------------------------------------------------------------
// One wave per chosen row s (level row r = loc[s], anchor row grows[s]); element e of the row's D + 6 + 3K values:
//   e <  D        feat     x = yf[r,e]        mean = pred[s, e]             scale = pred[s, D + e]          q = Q[r,0]
//   e <  D + 6    scaling  x = ys[r,e-D]      mean = pred[s, 2D + .]        scale = pred[s, 2D + 6 + .]     q = Q[r,1]
//   else          offsets  x = yo[r,.]        mean = pred[s, 2D + 12 + .]   scale = pred[s, 2D + 12 + 3K + .] q = Q[r,2]
//                 weighted by masks[grows[s], ./3]   (binary_grid_masks.repeat(1,1,3), :1664)
#ifndef RATE_LANES
#define RATE_LANES 32        // lanes per chosen row in level_rate_* (64 = round 2's wave per row; -DRATE_LANES=64 for the A/B)
#endif
struct RateElem { float x, mean, scale, q, w, xm; int kind, mcol, scol; int64_t xoff, moff; };

__device__ __forceinline__ RateElem rate_elem(int e, int64_t s, int64_t r, int64_t grow, int D, int K, int64_t ldp,
                                              const float *__restrict__ yf, const float *__restrict__ ys,
                                              const float *__restrict__ yo, const float *__restrict__ Q,
                                              const float *__restrict__ pred, const float *__restrict__ masks,
                                              const float *__restrict__ x_means) {
    RateElem t;
    const int O = 3 * K;
    const float *xsrc;
    int c;
    if (e < D) { t.kind = 0; c = e; xsrc = yf; t.xoff = r * D + c; t.mcol = c; t.scol = D + c; }
    else if (e < D + 6) { t.kind = 1; c = e - D; xsrc = ys; t.xoff = r * 6 + c; t.mcol = 2 * D + c; t.scol = 2 * D + 6 + c; }
    else { t.kind = 2; c = e - D - 6; xsrc = yo; t.xoff = r * O + c; t.mcol = 2 * D + 12 + c; t.scol = 2 * D + 12 + O + c; }
    t.x = xsrc[t.xoff];
    t.mean = pred[s * ldp + t.mcol];
    t.scale = pred[s * ldp + t.scol];
    t.q = Q[r * 3 + t.kind];
    t.moff = grow * K + c / 3;
    t.w = (t.kind == 2 && masks) ? masks[t.moff] : 1.f;
    t.xm = x_means ? x_means[t.kind] : 0.f;
    return t;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ void __launch_bounds__(256)
    level_rate_fwd_kernel(const float *__restrict__ yf, const float *__restrict__ ys, const float *__restrict__ yo,
                          const float *__restrict__ Q, const int64_t *__restrict__ loc, const float *__restrict__ pred,
                          const float *__restrict__ masks, const int64_t *__restrict__ grows,
                          const float *__restrict__ x_means, int use_clamp, int64_t n_sub, int D, int K,
                          int64_t ldp, float *__restrict__ sums) {
    __shared__ float part[4][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int E = D + 6 + 3 * K;
    float acc[3] = {0.f, 0.f, 0.f};
    // HALF a wave per row: the reference's 86 elements are three passes of 32 lanes (90 % of the lanes busy) where a
    // whole wave needed two passes of 64 (67 %) — the kernel is bound by the erf / log arithmetic, not by its loads
    const int l = lane & (RATE_LANES - 1);
    for (int64_t s = ((int64_t)blockIdx.x * 4 + wave) * (64 / RATE_LANES) + lane / RATE_LANES; s < n_sub;
         s += (int64_t)gridDim.x * 4 * (64 / RATE_LANES)) {
        const int64_t r = loc ? loc[s] : s, grow = grows ? grows[s] : s;
        for (int e = l; e < E; e += RATE_LANES) {
            const RateElem t = rate_elem(e, s, r, grow, D, K, ldp, yf, ys, yo, Q, pred, masks, x_means);
            const float b = rate_bits(rate_terms(t.x, t.mean, t.scale, t.q, t.xm, use_clamp)) * t.w;
            acc[0] += t.kind == 0 ? b : 0.f;
            acc[1] += t.kind == 1 ? b : 0.f;
            acc[2] += t.kind == 2 ? b : 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float v = wave_sum(acc[k]);
        if (lane == 0) part[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicAdd(&sums[threadIdx.x], (part[0][threadIdx.x] + part[1][threadIdx.x]) +
                                                       (part[2][threadIdx.x] + part[3][threadIdx.x]));
}

// d_pred [n_sub, P] is fully written; d_yf/d_ys/d_yo rows loc[s] are written (other rows: caller zero-fills);
// dQ [n_l, 3] rows loc[s] are written.
__global__ void __launch_bounds__(256)
    level_rate_bwd_kernel(const float *__restrict__ yf, const float *__restrict__ ys, const float *__restrict__ yo,
                          const float *__restrict__ Q, const int64_t *__restrict__ loc, const float *__restrict__ pred,
                          const float *__restrict__ masks, const int64_t *__restrict__ grows,
                          const float *__restrict__ x_means, int use_clamp, int64_t n_sub, int D, int K,
                          int64_t ldp, const float *__restrict__ g_sums, float *__restrict__ d_pred, float *__restrict__ d_yf,
                          float *__restrict__ d_ys, float *__restrict__ d_yo, float *__restrict__ dQ,
                          float *__restrict__ d_masks, int compact) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int E = D + 6 + 3 * K, P = 2 * E;
    const float g0 = g_sums[0], g1 = g_sums[1], g2 = g_sums[2];
    const int l = lane & (RATE_LANES - 1);          // half a wave per row, as in the forward
    for (int64_t s = ((int64_t)blockIdx.x * 4 + wave) * (64 / RATE_LANES) + lane / RATE_LANES; s < n_sub;
         s += (int64_t)gridDim.x * 4 * (64 / RATE_LANES)) {
        const int64_t r = loc ? loc[s] : s, grow = grows ? grows[s] : s;
        const int64_t ro = compact ? s : r;         // output row of d_yf / d_ys / d_yo / dQ
        float gq[3] = {0.f, 0.f, 0.f};
        for (int c = l; c < ldp - P; c += RATE_LANES) d_pred[s * ldp + P + c] = 0.f;   // outputs beyond the mean/scale block (the step sizes)
        for (int e = l; e < E; e += RATE_LANES) {
            const RateElem t = rate_elem(e, s, r, grow, D, K, ldp, yf, ys, yo, Q, pred, masks, x_means);
            const RateTerms rt = rate_terms(t.x, t.mean, t.scale, t.q, t.xm, use_clamp);
            const float gb = (t.kind == 0 ? g0 : (t.kind == 1 ? g1 : g2)) * t.w;
            const RateGrads g = rate_grads(rt, t.scale, gb);
            d_pred[s * ldp + t.mcol] = g.gm;
            d_pred[s * ldp + t.scol] = g.gs;
            float *dx = t.kind == 0 ? d_yf : (t.kind == 1 ? d_ys : d_yo);
            dx[t.xoff + (ro - r) * (t.kind == 0 ? D : (t.kind == 1 ? 6 : 3 * K))] = g.gx;
            // the offsets' bits are weighted by the (straight-through) binary mask: d bits*w / d w = bits (:1664)
            if (d_masks && t.kind == 2) atomicAdd(&d_masks[t.moff], g2 * rate_bits(rt));
            gq[0] += t.kind == 0 ? g.gq : 0.f;
            gq[1] += t.kind == 1 ? g.gq : 0.f;
            gq[2] += t.kind == 2 ? g.gq : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float v = gq[k];
#pragma unroll
            for (int o = RATE_LANES / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o);
            if (l == 0) dQ[ro * 3 + k] = v;
        }
    }
}

static int level_rate_check(const float *yf, const float *ys, const float *yo, const float *Q, const float *pred,
                            int64_t n_sub, int D, int K, int64_t ldpred, const char *what) {
    if (n_sub < 0 || D < 1 || K < 1 || ldpred < 2 * (D + 6 + 3 * K) || ldpred > 2 * (D + 6 + 3 * K) + 64) { cgs_set_error("%s: bad args", what); return CGS_ERR_ARG; }
    if (n_sub > 0 && (!yf || !ys || !yo || !Q || !pred)) { cgs_set_error("%s: NULL", what); return CGS_ERR_ARG; }
    return CGS_OK;
}

extern "C" int cgs_level_rate_fwd(const float *yf, const float *ys, const float *yo, const float *Q, const int64_t *loc,
                                  const float *pred, const float *masks, const int64_t *grows, const float *x_means,
                                  int use_clamp, int64_t n_sub, int D, int K, int64_t ldpred, float *sums,
                                  void *stream) {
    int rc = level_rate_check(yf, ys, yo, Q, pred, n_sub, D, K, ldpred, "level_rate_fwd");
    if (rc) return rc;
    if (n_sub == 0) return CGS_OK;
    if (!sums || (use_clamp && !x_means)) { cgs_set_error("level_rate_fwd: NULL"); return CGS_ERR_ARG; }
    CgsProfScope prof(CGS_PROF_RATE_FWD, (hipStream_t)stream);
    hipLaunchKernelGGL(level_rate_fwd_kernel, dim3(stream_grid(n_sub, 4 * 4)), dim3(256), 0, (hipStream_t)stream, yf, ys, yo,
                       Q, loc, pred, masks, grows, use_clamp ? x_means : nullptr, use_clamp, n_sub, D, K, ldpred, sums);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

extern "C" int cgs_level_rate_bwd(const float *yf, const float *ys, const float *yo, const float *Q, const int64_t *loc,
                                  const float *pred, const float *masks, const int64_t *grows, const float *x_means,
                                  int use_clamp, int64_t n_sub, int D, int K, int64_t ldpred, const float *g_sums,
                                  float *d_pred, float *d_yf, float *d_ys, float *d_yo, float *dQ, float *d_masks,
                                  int compact, void *stream) {
    int rc = level_rate_check(yf, ys, yo, Q, pred, n_sub, D, K, ldpred, "level_rate_bwd");
    if (rc) return rc;
    if (n_sub == 0) return CGS_OK;
    if (!g_sums || !d_pred || !d_yf || !d_ys || !d_yo || !dQ || (use_clamp && !x_means)) {
        cgs_set_error("level_rate_bwd: NULL");
        return CGS_ERR_ARG;
    }
    CgsProfScope prof(CGS_PROF_RATE_BWD, (hipStream_t)stream);
    hipLaunchKernelGGL(level_rate_bwd_kernel, dim3(stream_grid(n_sub, 4 * 4)), dim3(256), 0, (hipStream_t)stream, yf, ys, yo,
                       Q, loc, pred, masks, grows, use_clamp ? x_means : nullptr, use_clamp, n_sub, D, K, ldpred, g_sums, d_pred,
                       d_yf, d_ys, d_yo, dQ, masks ? d_masks : nullptr, compact);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Backward of the context gather (the parent rows read by the children of the next level) WITHOUT atomics: the
// children of every parent are known from the level plan (CSR: offs[p] .. offs[p+1] into `order`), so one wave
// per parent sums its children's gradient rows (lane = column, each child row is one coalesced read) and writes
// the parent's gradient once.  Deterministic, no zero fill for d_f / d_s, and ~4x faster than the 47 M fp32
// atomics of the scatter-add version at 800 k children.
// dout [n_children, ldo]: columns [0, wa) -> d_anchor[parent_row[p]], [wa, wa+DF) -> d_f[p], [wa+DF, wa+DF+DS) -> d_s[p]
__global__ void __launch_bounds__(256)
    ctx_gather_bwd_kernel(const float *__restrict__ dout, int64_t ldo, int64_t n_parents,
                          const int64_t *__restrict__ offs, const int64_t *__restrict__ order,
                          const int64_t *__restrict__ parent_row, float *__restrict__ d_anchor,
                          float *__restrict__ d_f, float *__restrict__ d_s, int wa, int DF, int DS, int acc_anchor) {
    const int lane = threadIdx.x & 63;
    const int W = wa + DF + DS;
    for (int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); p < n_parents; p += (int64_t)gridDim.x * 4) {
        const int64_t b = offs[p], e = offs[p + 1];
        float acc0 = 0.f, acc1 = 0.f;                     // columns lane and lane + 64
        // The children's row indices come in with ONE coalesced load per 64 children and are broadcast from a register,
        // and four row loads are in flight before the first is added: the plain loop (index load -> row load -> add,
        // per child) exposed two dependent memory latencies per child with ~5 children per parent.  Same additions in
        // the same order.
        for (int64_t base = b; base < e; base += 64) {
            const int m = (int)((e - base) < 64 ? (e - base) : 64);
            const int64_t mine = lane < m ? order[base + lane] : 0;
            for (int j = 0; j < m; j += 4) {
                float v0[4], v1[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t child = __shfl(mine, (j + u) & 63, 64);      // wave-uniform source lane
                    const float *row = dout + child * ldo;
                    const bool ok = j + u < m;
                    v0[u] = (ok && lane < W) ? row[lane] : 0.f;
                    v1[u] = (ok && lane + 64 < W) ? row[lane + 64] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (j + u < m) { acc0 += v0[u]; acc1 += v1[u]; }
                }
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = lane + 64 * h;
            const float v = h ? acc1 : acc0;
            if (c >= W) continue;
            if (c < wa) {       // (acc_anchor: the rows also carry what an earlier launch left there — distinct rows within a launch)
                if (d_anchor && e > b) { float *q = d_anchor + parent_row[p] * wa + c; *q = (acc_anchor & 1) ? *q + v : v; }
            }
            // (bits 1 / 2: d_f / d_s already hold the other gradient of the parents' rows — the level outputs' gradient buffer —
            //  and receive the children's sums on top: no separate buffer, no add launch behind this one)
            else if (c < wa + DF) { if (d_f) { float *q = d_f + p * DF + (c - wa); *q = (acc_anchor & 2) ? *q + v : v; } }
            else if (d_s) { float *q = d_s + p * DS + (c - wa - DF); *q = (acc_anchor & 4) ? *q + v : v; }
        }
    }
}

// The same sums with FOUR parents per wave (round 6): a 16-lane group owns a parent, a lane four consecutive columns (one 16-byte
// load per child row: 15 lanes cover the 59 columns of the reference's shapes, wa 3 + DF 50 + DS 6), and four children's rows are
// requested before the first is added.  The one-parent-per-wave kernel above is a chain of three dependent round trips per parent
// (list bounds, child indices, rows) with ~5 children each: 1.4 TB/s; four independent chains per wave and a quarter of the
// instructions per byte.  Same additions in the same (child) order per column.
__global__ void __launch_bounds__(256)
    ctx_gather_bwd4_kernel(const float *__restrict__ dout, int64_t ldo, int64_t n_parents,
                           const int64_t *__restrict__ offs, const int64_t *__restrict__ order,
                           const int64_t *__restrict__ parent_row, float *__restrict__ d_anchor,
                           float *__restrict__ d_f, float *__restrict__ d_s, int acc_anchor) {
    constexpr int WA = 3, DF = 50, DS = 6, W = WA + DF + DS;
    const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
    const int64_t stride = (int64_t)gridDim.x * 16;
    for (int64_t p0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4; p0 < n_parents; p0 += stride) {
        const int64_t p = p0 + grp;
        const bool live = p < n_parents;
        const int64_t b = live ? offs[p] : 0, e = live ? offs[p + 1] : 0;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool col_ok = 4 * sub < W;                     // lanes 0..14
        for (int64_t k = b; __builtin_amdgcn_ballot_w64(k < e) != 0ull; k += 4) {
            int64_t child[4];
            float4 v[4];
            // UNCONDITIONAL loads from clamped indices (a load behind `k + u < e` is an exec-mask branch whose join drains the load
            // queue: the four index loads and the four row loads were eight exposed round trips); selected afterwards.
            // (e > b here for some lane; a lane without children reads entry 0 / row order[0]: valid memory, value dropped)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t kk = k + u < e ? k + u : (e > b ? e - 1 : 0);
                child[u] = order[kk];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                // (16-byte load at a 4-byte-aligned address: rows are ldo floats apart; lane 15 of a group reads columns 60..63
                //  of the row — inside it, ldo >= 64 is checked by the launcher — and drops them)
                const float4 t = *(const float4 *)(dout + child[u] * ldo + 4 * sub);
                const bool ok = col_ok && (k + u < e);
                v[u] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (k + u < e) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
            }
        }
        if (!live || !col_ok) continue;
        const float a4[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = 4 * sub + j;
            if (c >= W) continue;
            const float vv = a4[j];
            if (c < WA) {
                if (d_anchor && e > b) { float *q = d_anchor + parent_row[p] * WA + c; *q = (acc_anchor & 1) ? *q + vv : vv; }
            } else if (c < WA + DF) {
                if (d_f) { float *q = d_f + p * DF + (c - WA); *q = (acc_anchor & 2) ? *q + vv : vv; }
            } else if (d_s) {
                float *q = d_s + p * DS + (c - WA - DF);
                *q = (acc_anchor & 4) ? *q + vv : vv;
            }
        }
    }
}

extern "C" int cgs_ctx_gather_bwd_acc(const float *dout, int64_t ldo, int64_t n_parents, const int64_t *offs,
                                      const int64_t *order, const int64_t *parent_row, float *d_anchor, float *d_f,
                                      float *d_s, int wa, int DF, int DS, int accumulate_anchor, void *stream);
extern "C" int cgs_ctx_gather_bwd(const float *dout, int64_t ldo, int64_t n_parents, const int64_t *offs,
                                  const int64_t *order, const int64_t *parent_row, float *d_anchor, float *d_f,
                                  float *d_s, int wa, int DF, int DS, void *stream) {
    return cgs_ctx_gather_bwd_acc(dout, ldo, n_parents, offs, order, parent_row, d_anchor, d_f, d_s, wa, DF, DS, 0, stream);
}
static const bool g_gather1 = getenv("CGS_CTX_GATHER1") != nullptr;       // A/B knob: the one-parent-per-wave kernel
// accumulate_anchor: bit 0 = d_anchor rows are ADDED to (the levels of one backward share one anchor-gradient buffer);
// bit 1 / bit 2 = d_f / d_s rows are added to (they are the first rows of the gradient buffer of the level outputs)
extern "C" int cgs_ctx_gather_bwd_acc(const float *dout, int64_t ldo, int64_t n_parents, const int64_t *offs,
                                      const int64_t *order, const int64_t *parent_row, float *d_anchor, float *d_f,
                                      float *d_s, int wa, int DF, int DS, int accumulate_anchor, void *stream) {
    if (n_parents < 0 || wa < 0 || DF < 0 || DS < 0 || wa + DF + DS < 1 || wa + DF + DS > 128 || ldo < wa + DF + DS) {
        cgs_set_error("ctx_gather_bwd: bad args");
        return CGS_ERR_ARG;
    }
    if (n_parents == 0) return CGS_OK;
    if (!dout || !offs || !order || (d_anchor && !parent_row)) { cgs_set_error("ctx_gather_bwd: NULL"); return CGS_ERR_ARG; }
    CgsProfScope prof(CGS_PROF_CTX_BWD, (hipStream_t)stream);
    if (wa == 3 && DF == 50 && DS == 6 && ldo >= 64 && !g_gather1) {      // (ldo >= 64: every lane's 16-byte load stays inside its row)
        hipLaunchKernelGGL(ctx_gather_bwd4_kernel, dim3(stream_grid(n_parents, 16 * 2)), dim3(256), 0, (hipStream_t)stream, dout, ldo,
                           n_parents, offs, order, parent_row, d_anchor, d_f, d_s, accumulate_anchor & 7);
        CGS_CHECK_HIP(hipGetLastError());
        return CGS_OK;
    }
    hipLaunchKernelGGL(ctx_gather_bwd_kernel, dim3(stream_grid(n_parents, 4 * 8)), dim3(256), 0, (hipStream_t)stream, dout,
                       ldo, n_parents, offs, order, parent_row, d_anchor, d_f, d_s, wa, DF, DS, accumulate_anchor & 7);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

// ------------------------------------------------------------------------------------------------------------
// The scalar bookkeeping at the end of the rate model (scene/gaussian_model.py:1687-1694) as one launch each way:
//   tot_k = rate * sum_l S[l,k];  out = [ (tot_f + tot_s + tot_o + rate * h) / n_tot,  tot_f / n_f,  tot_s / n_s,  tot_o / n_o ]
//   raw   = [ 1 - live fraction, rate * h, S[0,:].sum(), S[1,:].sum(), ... ]   (the per-level report, :1697-1705)
// As torch ops on one-element tensors this was ~25 launches forward and ~20 backward per step.
struct RateFinishArgs { float rate, inv_nf, inv_ns, inv_no, inv_ntot, dead_frac; int L; };

__global__ void rate_finish_fwd_kernel(const float *__restrict__ S, const float *__restrict__ hsum, RateFinishArgs a,
                                       float *__restrict__ out4, float *__restrict__ raw) {
    if (threadIdx.x != 0) return;
    float tf = 0.f, ts = 0.f, to = 0.f;
    for (int l = 0; l < a.L; ++l) {
        tf += S[3 * l]; ts += S[3 * l + 1]; to += S[3 * l + 2];
        raw[2 + l] = S[3 * l] + S[3 * l + 1] + S[3 * l + 2];
    }
    tf *= a.rate; ts *= a.rate; to *= a.rate;
    const float sh = hsum[0] * a.rate;
    out4[0] = (tf + ts + to + sh) * a.inv_ntot;
    out4[1] = tf * a.inv_nf;
    out4[2] = ts * a.inv_ns;
    out4[3] = to * a.inv_no;
    raw[0] = a.dead_frac;
    raw[1] = sh;
}

__global__ void rate_finish_bwd_kernel(const float *__restrict__ ga, const float *__restrict__ gb, const float *__restrict__ gc,
                                       const float *__restrict__ gd, RateFinishArgs a, float *__restrict__ dS,
                                       float *__restrict__ dh) {
    // (one pointer per output: an output the loss does not read has no gradient tensor at all)
    const int i = threadIdx.x;
    const float g0 = (ga ? ga[0] : 0.f) * a.rate * a.inv_ntot;
    if (i < 3 * a.L) {
        const int k = i % 3;
        const float *gp = k == 0 ? gb : (k == 1 ? gc : gd);
        dS[i] = g0 + (gp ? gp[0] : 0.f) * a.rate * (k == 0 ? a.inv_nf : (k == 1 ? a.inv_ns : a.inv_no));
    }
    if (i == 0) dh[0] = g0;
}

extern "C" int cgs_rate_finish_fwd(const float *S, int L, const float *hsum, float rate, double n_feat, double n_scaling,
                                   double n_offsets, float dead_frac, float *out4, float *raw, void *stream) {
    if (L < 0 || L > 16 || !S || !hsum || !out4 || !raw) { cgs_set_error("rate_finish_fwd: bad args"); return CGS_ERR_ARG; }
    const double nt = n_feat + n_scaling + n_offsets;
    RateFinishArgs a{rate, (float)(1.0 / (n_feat > 1 ? n_feat : 1)), (float)(1.0 / (n_scaling > 1 ? n_scaling : 1)),
                     (float)(1.0 / (n_offsets > 1 ? n_offsets : 1)), (float)(1.0 / (nt > 1 ? nt : 1)), dead_frac, L};
    hipLaunchKernelGGL(rate_finish_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, S, hsum, a, out4, raw);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

extern "C" int cgs_rate_finish_bwd4(const float *g_all, const float *g_feat, const float *g_scaling, const float *g_offsets, int L,
                                    float rate, double n_feat, double n_scaling, double n_offsets, float *dS, float *dh,
                                    void *stream) {
    if (L < 0 || L > 16 || !dS || !dh) { cgs_set_error("rate_finish_bwd: bad args"); return CGS_ERR_ARG; }
    const double nt = n_feat + n_scaling + n_offsets;
    RateFinishArgs a{rate, (float)(1.0 / (n_feat > 1 ? n_feat : 1)), (float)(1.0 / (n_scaling > 1 ? n_scaling : 1)),
                     (float)(1.0 / (n_offsets > 1 ? n_offsets : 1)), (float)(1.0 / (nt > 1 ? nt : 1)), 0.f, L};
    hipLaunchKernelGGL(rate_finish_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, g_all, g_feat, g_scaling, g_offsets, a, dS,
                       dh);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

extern "C" int cgs_rate_finish_bwd(const float *g4, int L, float rate, double n_feat, double n_scaling, double n_offsets,
                                   float *dS, float *dh, void *stream) {
    if (!g4) { cgs_set_error("rate_finish_bwd: bad args"); return CGS_ERR_ARG; }
    return cgs_rate_finish_bwd4(g4, g4 + 1, g4 + 2, g4 + 3, L, rate, n_feat, n_scaling, n_offsets, dS, dh, stream);
}
