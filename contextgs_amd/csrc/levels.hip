// Level division of the context model: `torch_unique_with_indices` of
// utils/multi_level.py:3-31 on the integer voxel keys round(anchor / voxel_size / level_scale)
// (scene/gaussian_model.py:1751-1765) as a handful of launches: key range -> packed keys -> stable radix sort of
// (key, index) -> run heads -> scan -> (inverse, first occurrence, counts, unique rows).
// Contract: unique rows in ascending lexicographic order (col 0, then 1, then 2; -0.0 merged with 0.0),
// `inverse[i]` = unique row of input row i, `first[g]` = SMALLEST input index of group g (stable sort: the first
// element of a run), `counts[g]` = group size.  The torch composition this replaces was ~15 element-wise / index
// launches around three sorts (one per column).
#include "cgs_internal.h"

int cgs_scan_exclusive_u32_total(const uint32_t *in, uint32_t *out, int64_t n, void *scratch, size_t scratch_bytes,
                                 uint32_t *grand_total, hipStream_t stream);

// out7 (device floats): min of the three columns, max of the three columns, 1.0 if some value is not an integer.
// Ordered-int atomics on the float bits (values are finite voxel coordinates).
__device__ __forceinline__ int lv_f2o(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7FFFFFFF; }
__device__ __forceinline__ float lv_o2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

__global__ void __launch_bounds__(256) level_range_init_kernel(int *__restrict__ o) {
    const int k = threadIdx.x;
    if (k < 3) o[k] = 0x7FFFFFFF;                  // min
    else if (k < 6) o[k] = (int)0x80000000;        // max
    else if (k == 6) o[k] = 0;
}

__global__ void __launch_bounds__(256) level_range_kernel(const float *__restrict__ keys, int64_t n, int *__restrict__ o) {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    int bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = keys[3 * i + c] + 0.0f;
            mn[c] = fminf(mn[c], v);
            mx[c] = fmaxf(mx[c], v);
            bad |= (rintf(v) != v) ? 1 : 0;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            mn[c] = fminf(mn[c], __shfl_xor(mn[c], d, 64));
            mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], d, 64));
        }
        bad |= __shfl_xor(bad, d, 64);
    }
    // one set of atomics per WORKGROUP (the four waves combine in LDS): same-address global atomics serialise
    __shared__ float smn[4][3], smx[4][3];
    __shared__ int sbad[4];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { smn[wave][c] = mn[c]; smx[wave][c] = mx[c]; }
        sbad[wave] = bad;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        atomicMin(&o[c], lv_f2o(fminf(fminf(smn[0][c], smn[1][c]), fminf(smn[2][c], smn[3][c]))));
        atomicMax(&o[3 + c], lv_f2o(fmaxf(fmaxf(smx[0][c], smx[1][c]), fmaxf(smx[2][c], smx[3][c]))));
    }
    if (threadIdx.x == 3 && (sbad[0] | sbad[1] | sbad[2] | sbad[3])) atomicOr(&o[6], 1);
}

__global__ void level_range_finish_kernel(const int *__restrict__ o, float *__restrict__ out7) {
    const int k = threadIdx.x;
    if (k < 6) out7[k] = lv_o2f(o[k]);
    else if (k == 6) out7[k] = o[k] ? 1.f : 0.f;
}

extern "C" int cgs_level_key_range(const float *keys, int64_t n, float *out7, void *scratch8, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n <= 0 || !keys || !out7 || !scratch8) { cgs_set_error("level_key_range: bad args"); return CGS_ERR_ARG; }
    int *o = (int *)scratch8;
    hipLaunchKernelGGL(level_range_init_kernel, dim3(1), dim3(64), 0, stream, o);
    const int64_t want = (n + 255) / 256;
    hipLaunchKernelGGL(level_range_kernel, dim3((unsigned)(want < 512 ? want : 512)), dim3(256), 0, stream, keys, n, o);
    hipLaunchKernelGGL(level_range_finish_kernel, dim3(1), dim3(64), 0, stream, (const int *)o, out7);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

struct LvPack { int lo[3]; int b1, b2; };          // packed key = ((x - lo0) << (b1 + b2)) | ((y - lo1) << b2) | (z - lo2)

__device__ __forceinline__ uint64_t lv_key(const float *__restrict__ keys, int64_t i, const LvPack &p) {
    const uint64_t x = (uint64_t)((int)(keys[3 * i] + 0.0f) - p.lo[0]);
    const uint64_t y = (uint64_t)((int)(keys[3 * i + 1] + 0.0f) - p.lo[1]);
    const uint64_t z = (uint64_t)((int)(keys[3 * i + 2] + 0.0f) - p.lo[2]);
    return (x << (p.b1 + p.b2)) | (y << p.b2) | z;
}

__global__ void __launch_bounds__(256)
    level_pack_kernel(const float *__restrict__ keys, int64_t n, LvPack p, uint32_t *__restrict__ w0, uint32_t *__restrict__ idx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    w0[i] = (uint32_t)lv_key(keys, i, p);
    idx[i] = (uint32_t)i;
}

__global__ void __launch_bounds__(256)
    level_hiword_kernel(const float *__restrict__ keys, int64_t n, LvPack p, const uint32_t *__restrict__ order,
                        uint32_t *__restrict__ w1) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    w1[i] = (uint32_t)(lv_key(keys, order[i], p) >> 32);
}

__global__ void __launch_bounds__(256)
    level_heads_kernel(const float *__restrict__ keys, int64_t n, LvPack p, const uint32_t *__restrict__ order,
                       uint32_t *__restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    flag[i] = (i == 0 || lv_key(keys, order[i], p) != lv_key(keys, order[i - 1], p)) ? 1u : 0u;
}

__global__ void __launch_bounds__(256)
    level_finish_kernel(const float *__restrict__ keys, int64_t n, const uint32_t *__restrict__ order,
                        const uint32_t *__restrict__ flag, const uint32_t *__restrict__ excl, int64_t *__restrict__ inverse,
                        int64_t *__restrict__ first, uint32_t *__restrict__ start, float *__restrict__ unique) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t f = flag[i], g = excl[i] + f - 1u, o = order[i];
    inverse[o] = (int64_t)g;
    if (f) {
        first[g] = (int64_t)o;
        start[g] = (uint32_t)i;
#pragma unroll
        for (int c = 0; c < 3; ++c) unique[3 * (int64_t)g + c] = keys[3 * (int64_t)o + c] + 0.0f;
    }
}

__global__ void __launch_bounds__(256)
    level_counts_kernel(int64_t n, const uint32_t *__restrict__ start, const uint32_t *__restrict__ total,
                        int64_t *__restrict__ counts) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t m = *total;
    if (g >= (int64_t)m) return;
    counts[g] = (int64_t)((g + 1 < (int64_t)m ? start[g + 1] : (uint32_t)n) - start[g]);
}

extern "C" size_t cgs_level_unique_scratch_bytes(int64_t n) {
    if (n < 1) n = 1;
    // 8 uint32 arrays of n (two key words, order x2, ping-pong x2, flags, scan / starts) + sort / scan scratch
    return (size_t)9 * cgs_align_up((size_t)n * 4, 256) + cgs_sort_scratch_bytes(n) + cgs_scan_scratch_bytes(n) + 1024;
}

// keys [n,3]: integer-valued floats (checked by the caller through cgs_level_key_range); lo / bits: HOST arrays of the
// three column minima and key widths (sum <= 62, each <= 31).  Outputs: inverse [n], first [n] / counts [n] /
// unique [n,3] of which the first *n_unique_host entries are valid (ONE stream synchronisation for that count).
extern "C" int cgs_level_unique(const float *keys, int64_t n, const int32_t *lo, const int32_t *bits, int64_t *inverse,
                                int64_t *first, int64_t *counts, float *unique, int64_t *n_unique_host, void *scratch,
                                size_t scratch_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!n_unique_host) { cgs_set_error("level_unique: NULL n_unique_host"); return CGS_ERR_ARG; }
    *n_unique_host = 0;
    if (n < 0 || n >= (1ll << 31)) { cgs_set_error("level_unique: bad n"); return CGS_ERR_ARG; }
    if (n == 0) return CGS_OK;
    if (!keys || !lo || !bits || !inverse || !first || !counts || !unique || !scratch) { cgs_set_error("level_unique: NULL"); return CGS_ERR_ARG; }
    const int total_bits = bits[0] + bits[1] + bits[2];
    if (bits[0] < 1 || bits[1] < 1 || bits[2] < 1 || bits[0] > 31 || bits[1] > 31 || bits[2] > 31 || total_bits > 62) {
        cgs_set_error("level_unique: key widths %d+%d+%d not packable", bits[0], bits[1], bits[2]);
        return CGS_ERR_ARG;
    }
    if (scratch_bytes < cgs_level_unique_scratch_bytes(n)) { cgs_set_error("level_unique: scratch too small"); return CGS_ERR_WORKSPACE; }
    CgsCarver cv(scratch, scratch_bytes);
    uint32_t *w0 = cv.take<uint32_t>(n), *w1 = cv.take<uint32_t>(n), *ord_a = cv.take<uint32_t>(n), *ord_b = cv.take<uint32_t>(n);
    uint32_t *kt = cv.take<uint32_t>(n), *vt = cv.take<uint32_t>(n), *ko = cv.take<uint32_t>(n), *flag = cv.take<uint32_t>(n);
    uint32_t *excl = cv.take<uint32_t>(n);
    const size_t sort_bytes = cgs_sort_scratch_bytes(n), scan_bytes = cgs_scan_scratch_bytes(n);
    void *sort_scratch = cv.take<char>(sort_bytes);
    char *scan_scratch = cv.take<char>(scan_bytes + 256);
    if (!cv.ok) { cgs_set_error("level_unique: scratch too small"); return CGS_ERR_WORKSPACE; }
    uint32_t *total = (uint32_t *)(scan_scratch + scan_bytes);
    LvPack p{{lo[0], lo[1], lo[2]}, bits[1], bits[2]};
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipLaunchKernelGGL(level_pack_kernel, grid, block, 0, stream, keys, n, p, w0, ord_a);
    CGS_CHECK_HIP(hipGetLastError());
    // stable LSD: low word (all its bits), then — keys wider than 32 bits — the high word gathered through the order
    int rc = cgs_sort_pairs_u32(w0, ord_a, ko, ord_b, kt, vt, n, 0, total_bits < 32 ? total_bits : 32, sort_scratch, sort_bytes, stream);
    if (rc) return rc;
    uint32_t *order = ord_b;
    if (total_bits > 32) {
        hipLaunchKernelGGL(level_hiword_kernel, grid, block, 0, stream, keys, n, p, (const uint32_t *)ord_b, w1);
        CGS_CHECK_HIP(hipGetLastError());
        rc = cgs_sort_pairs_u32(w1, ord_b, ko, ord_a, kt, vt, n, 0, total_bits - 32, sort_scratch, sort_bytes, stream);
        if (rc) return rc;
        order = ord_a;
    }
    hipLaunchKernelGGL(level_heads_kernel, grid, block, 0, stream, keys, n, p, (const uint32_t *)order, flag);
    CGS_CHECK_HIP(hipGetLastError());
    rc = cgs_scan_exclusive_u32_total(flag, excl, n, scan_scratch, scan_bytes, total, stream);
    if (rc) return rc;
    uint32_t *start = w0;                            // the packed low words are no longer needed
    hipLaunchKernelGGL(level_finish_kernel, grid, block, 0, stream, keys, n, (const uint32_t *)order, (const uint32_t *)flag,
                       (const uint32_t *)excl, inverse, first, start, unique);
    hipLaunchKernelGGL(level_counts_kernel, grid, block, 0, stream, n, (const uint32_t *)start, (const uint32_t *)total, counts);
    CGS_CHECK_HIP(hipGetLastError());
    static thread_local uint32_t *pinned = nullptr;
    if (!pinned) CGS_CHECK_HIP(hipHostMalloc((void **)&pinned, 64, hipHostMallocDefault));
    CGS_CHECK_HIP(hipMemcpyAsync(pinned, total, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    CGS_CHECK_HIP(hipStreamSynchronize(stream));
    *n_unique_host = pinned[0];
    return CGS_OK;
}
