// Anchor initialisation kNN: `distCUDA2` of the simple_knn wheel the reference calls in
// create_from_pcd (scene/gaussian_model.py:389,407; the wheel is NOT in the mount — restated from its published
// behaviour): for every point the MEAN of the squared distances to its 3 nearest OTHER points (exact, fp32
// ((dx*dx + dy*dy) + dz*dz), duplicates count as neighbours at distance 0, only the point's own index is excluded).
//
// MI355X design: exact search over a two-level box hierarchy built from one sort.
//   1. bounding box (ordered-int atomics), 30-bit Morton key per point, stable radix sort (prims.hip);
//   2. points gathered into curve order as float4 (w = original index);
//   3. LEAF = 64 consecutive points = one wave; SUPER box = 64 consecutive leaves.  Equal-count leaves follow
//      the density of the cloud, so clustered scans behave like uniform ones;
//   4. query: one wave per leaf, one lane per query point.  The wave scans its own leaf, then walks the super
//      boxes (wave-uniform scalar loads); a box is opened only if it is closer than SOME lane's current 3rd-best
//      distance (ballot), and an opened leaf's 64 points are broadcast lane by lane (v_readlane) into a
//      branch-free 3-element insertion (5 min/max).  Lanes of a wave are neighbours on the curve, so the boxes
//      one lane needs are mostly the boxes all lanes need.
// Box distances use the same monotone fp32 operations as point distances, so pruning never drops a true neighbour.
#include <cfloat>
#include "cgs_internal.h"

#define KNN_LEAF 64
#define KNN_FAN 64

__device__ __forceinline__ uint32_t knn_ord(float f) {          // order-preserving float -> uint
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float knn_unord(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

__global__ void __launch_bounds__(256) knn_bbox_init_kernel(uint32_t *bbox) {
    if (threadIdx.x < 3) bbox[threadIdx.x] = 0xFFFFFFFFu;
    else if (threadIdx.x < 6) bbox[threadIdx.x] = 0u;
}

__global__ void __launch_bounds__(256) knn_bbox_kernel(const float *__restrict__ pts, int64_t n, uint32_t *bbox) {
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = pts[3 * i + a];
            lo[a] = fminf(lo[a], v);
            hi[a] = fmaxf(hi[a], v);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], o));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o));
        }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            atomicMin(&bbox[a], knn_ord(lo[a]));
            atomicMax(&bbox[3 + a], knn_ord(hi[a]));
        }
    }
}

__device__ __forceinline__ uint32_t knn_spread3(uint32_t v) {     // 10 bits -> every third bit
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ void __launch_bounds__(256)
knn_morton_kernel(const float *__restrict__ pts, int64_t n, const uint32_t *__restrict__ bbox, uint32_t *keys,
                  uint32_t *vals) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t code = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float lo = knn_unord(bbox[a]), hi = knn_unord(bbox[3 + a]);
        const float ext = hi - lo;
        const float t = ext > 0.f ? (pts[3 * i + a] - lo) / ext : 0.f;
        const uint32_t c = (uint32_t)fminf(fmaxf(t * 1024.f, 0.f), 1023.f);
        code |= knn_spread3(c) << a;
    }
    keys[i] = code;
    vals[i] = (uint32_t)i;
}

__global__ void __launch_bounds__(256)
knn_gather_kernel(const float *__restrict__ pts, const uint32_t *__restrict__ order, int64_t n, float4 *sorted) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = order[i];
    sorted[i] = make_float4(pts[3 * (int64_t)j], pts[3 * (int64_t)j + 1], pts[3 * (int64_t)j + 2], __uint_as_float(j));
}

// box layout: 8 floats = lo.xyz, pad, hi.xyz, pad
__device__ __forceinline__ void knn_wave_box(float lo[3], float hi[3], float *dst, int lane) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], o));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o));
        }
    }
    if (lane == 0) {
        dst[0] = lo[0]; dst[1] = lo[1]; dst[2] = lo[2]; dst[3] = 0.f;
        dst[4] = hi[0]; dst[5] = hi[1]; dst[6] = hi[2]; dst[7] = 0.f;
    }
}

__global__ void __launch_bounds__(64) knn_leaf_box_kernel(const float4 *__restrict__ sorted, int64_t n, float *leaf_box) {
    const int64_t leaf = blockIdx.x;
    const int lane = threadIdx.x;
    const int64_t i = leaf * KNN_LEAF + lane;
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    if (i < n) {
        const float4 p = sorted[i];
        lo[0] = hi[0] = p.x; lo[1] = hi[1] = p.y; lo[2] = hi[2] = p.z;
    }
    knn_wave_box(lo, hi, leaf_box + 8 * leaf, lane);
}

__global__ void __launch_bounds__(64)
knn_super_box_kernel(const float *__restrict__ leaf_box, int64_t n_leaf, float *super_box) {
    const int64_t sb = blockIdx.x;
    const int lane = threadIdx.x;
    const int64_t l = sb * KNN_FAN + lane;
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    if (l < n_leaf) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = leaf_box[8 * l + a]; hi[a] = leaf_box[8 * l + 4 + a]; }
    }
    knn_wave_box(lo, hi, super_box + 8 * sb, lane);
}

__device__ __forceinline__ float knn_box_dist2(float px, float py, float pz, const float *__restrict__ box) {
    const float dx = fmaxf(fmaxf(box[0] - px, px - box[4]), 0.f);
    const float dy = fmaxf(fmaxf(box[1] - py, py - box[5]), 0.f);
    const float dz = fmaxf(fmaxf(box[2] - pz, pz - box[6]), 0.f);
    return (dx * dx + dy * dy) + dz * dz;
}

struct KnnBest { float b0, b1, b2; };

__device__ __forceinline__ void knn_insert(KnnBest &k, float d) {
    const float m0 = fmaxf(k.b0, d);
    k.b0 = fminf(k.b0, d);
    const float m1 = fmaxf(k.b1, m0);
    k.b1 = fminf(k.b1, m0);
    k.b2 = fminf(k.b2, m1);
}

// all 64 lanes compare their query against the (up to 64) points of leaf `l`; `skip_self`: the lane's own slot
__device__ __forceinline__ void knn_scan_leaf(const float4 *__restrict__ sorted, int64_t n, int64_t l, int lane,
                                              float px, float py, float pz, bool skip_self, KnnBest &k) {
    const int64_t base = l * KNN_LEAF;
    const int m = (int)min((int64_t)KNN_LEAF, n - base);
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < m) q = sorted[base + lane];
    for (int t = 0; t < m; ++t) {
        const float qx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.x), t));
        const float qy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.y), t));
        const float qz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.z), t));
        const float dx = px - qx, dy = py - qy, dz = pz - qz;
        float d = (dx * dx + dy * dy) + dz * dz;
        if (skip_self && t == lane) d = FLT_MAX;
        knn_insert(k, d);
    }
}

__global__ void __launch_bounds__(64)
knn_query_kernel(const float4 *__restrict__ sorted, int64_t n, const float *__restrict__ leaf_box, int64_t n_leaf,
                 const float *__restrict__ super_box, int64_t n_super, float *__restrict__ out) {
    const int64_t leaf = blockIdx.x;
    const int lane = threadIdx.x;
    const int64_t i = leaf * KNN_LEAF + lane;
    const bool valid = i < n;
    const float4 p = sorted[valid ? i : n - 1];
    KnnBest k;
    // padding lanes start "full" (-1 is below every distance): they never open a box and never change
    k.b0 = k.b1 = k.b2 = valid ? FLT_MAX : -1.f;
    knn_scan_leaf(sorted, n, leaf, lane, p.x, p.y, p.z, true, k);
    for (int64_t sb = 0; sb < n_super; ++sb) {
        if (__ballot(knn_box_dist2(p.x, p.y, p.z, super_box + 8 * sb) < k.b2) == 0ull) continue;
        const int64_t l_end = min(n_leaf, (sb + 1) * KNN_FAN);
        for (int64_t l = sb * KNN_FAN; l < l_end; ++l) {
            if (l == leaf) continue;
            if (__ballot(knn_box_dist2(p.x, p.y, p.z, leaf_box + 8 * l) < k.b2) == 0ull) continue;
            knn_scan_leaf(sorted, n, l, lane, p.x, p.y, p.z, false, k);
        }
    }
    if (valid) out[__float_as_uint(p.w)] = ((k.b0 + k.b1) + k.b2) / 3.f;
}

static size_t knn_carve(int64_t n, size_t *off_keys, size_t *off_sorted, size_t *off_leaf, size_t *off_super,
                        size_t *off_bbox, size_t *off_sort) {
    const int64_t n_leaf = (n + KNN_LEAF - 1) / KNN_LEAF, n_super = (n_leaf + KNN_FAN - 1) / KNN_FAN;
    size_t o = 0;
    *off_keys = o;   o += cgs_align_up((size_t)n * 4 * 6, 256);        // keys, vals, out x2, tmp x2
    *off_sorted = o; o += cgs_align_up((size_t)n * 16, 256);
    *off_leaf = o;   o += cgs_align_up((size_t)n_leaf * 32, 256);
    *off_super = o;  o += cgs_align_up((size_t)n_super * 32, 256);
    *off_bbox = o;   o += 256;
    *off_sort = o;   o += cgs_sort_scratch_bytes(n);
    return o + 256;
}

extern "C" size_t cgs_knn_scratch_bytes(int64_t n) {
    size_t a, b, c, d, e, f;
    return knn_carve(n > 0 ? n : 1, &a, &b, &c, &d, &e, &f);
}

extern "C" int cgs_knn_mean_dist2(const float *points, int64_t n, float *mean_dist2, void *scratch,
                                  size_t scratch_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0 || (n > 0 && (!points || !mean_dist2 || !scratch))) { cgs_set_error("knn: bad arguments"); return CGS_ERR_ARG; }
    if (n == 0) return CGS_OK;
    if (n >= (1ll << 31)) { cgs_set_error("knn: n must fit in int32"); return CGS_ERR_ARG; }
    size_t o_keys, o_sorted, o_leaf, o_super, o_bbox, o_sort;
    const size_t need = knn_carve(n, &o_keys, &o_sorted, &o_leaf, &o_super, &o_bbox, &o_sort);
    if (scratch_bytes < need) { cgs_set_error("knn: scratch too small (%zu < %zu)", scratch_bytes, need); return CGS_ERR_WORKSPACE; }
    char *ws = (char *)scratch;
    uint32_t *keys = (uint32_t *)(ws + o_keys), *vals = keys + n, *keys_out = vals + n, *vals_out = keys_out + n,
             *keys_tmp = vals_out + n, *vals_tmp = keys_tmp + n;
    float4 *sorted = (float4 *)(ws + o_sorted);
    float *leaf_box = (float *)(ws + o_leaf), *super_box = (float *)(ws + o_super);
    uint32_t *bbox = (uint32_t *)(ws + o_bbox);
    const int64_t n_leaf = (n + KNN_LEAF - 1) / KNN_LEAF, n_super = (n_leaf + KNN_FAN - 1) / KNN_FAN;
    const unsigned nb = (unsigned)((n + 255) / 256);

    hipLaunchKernelGGL(knn_bbox_init_kernel, dim3(1), dim3(256), 0, stream, bbox);
    hipLaunchKernelGGL(knn_bbox_kernel, dim3(nb < 2048u ? nb : 2048u), dim3(256), 0, stream, points, n, bbox);
    hipLaunchKernelGGL(knn_morton_kernel, dim3(nb), dim3(256), 0, stream, points, n, (const uint32_t *)bbox, keys, vals);
    CGS_CHECK_HIP(hipGetLastError());
    const int rc = cgs_sort_pairs_u32(keys, vals, keys_out, vals_out, keys_tmp, vals_tmp, n, 0, 30, ws + o_sort,
                                      scratch_bytes - o_sort, stream_);
    if (rc) return rc;
    hipLaunchKernelGGL(knn_gather_kernel, dim3(nb), dim3(256), 0, stream, points, (const uint32_t *)vals_out, n, sorted);
    hipLaunchKernelGGL(knn_leaf_box_kernel, dim3((unsigned)n_leaf), dim3(64), 0, stream, (const float4 *)sorted, n, leaf_box);
    hipLaunchKernelGGL(knn_super_box_kernel, dim3((unsigned)n_super), dim3(64), 0, stream, (const float *)leaf_box, n_leaf,
                       super_box);
    hipLaunchKernelGGL(knn_query_kernel, dim3((unsigned)n_leaf), dim3(64), 0, stream, (const float4 *)sorted, n,
                       (const float *)leaf_box, n_leaf, (const float *)super_box, n_super, mean_dist2);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}
