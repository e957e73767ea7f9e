// Analysis hooks (experiment builds only, not part of include/cgs.h) and one test hook.  First: how many wave iterations the blend kernels need with the current
// mapping (one 8x8 quadrant per wave, one Gaussian per iteration) versus a mapping where the four 16-lane rows of
// a wave own one 4x4 pixel block each and walk their own Gaussian lists (four Gaussians per iteration).
// Uses only the bounding boxes of the alpha >= 1/255 ellipses, like the blend kernels' own quadrant masks.
#include "cgs_internal.h"

#ifdef CGS_EXPERIMENTS   // counting kernels behind tools/blend_occupancy.py: experiment builds only
__global__ void __launch_bounds__(256)
    blend_occupancy_kernel(int tiles_x, const uint2 *__restrict__ ranges, const uint32_t *__restrict__ gid_sorted,
                           const float4 *__restrict__ rec, const uint32_t *__restrict__ tile_last,
                           unsigned long long *__restrict__ out) {
    __shared__ unsigned int cnt[4][5];     // per quadrant: [0] quadrant visits, [1..4] visits of its four 4x4 blocks
    const int tile = blockIdx.x, tid = threadIdx.x;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const uint32_t tlast = tile_last[tile];
    const uint2 range = ranges[tile];
    unsigned long long q_iters = 0, b_iters = 0, b_visits = 0;
    __shared__ unsigned int exact_cnt, oct_cnt, pix_cnt;
    unsigned long long e_visits = 0, o_visits = 0, p_hits = 0;
    for (uint32_t base = 0; base < tlast; base += 256) {
        if (tid < 20) cnt[tid / 5][tid % 5] = 0;
        if (tid == 0) { exact_cnt = 0; oct_cnt = 0; pix_cnt = 0; }
        __syncthreads();
        const uint32_t pos = base + tid;
        if (pos < tlast) {
            const uint32_t g = gid_sorted[range.x + pos];
            const float4 r0 = rec[3 * (size_t)g], r1 = rec[3 * (size_t)g + 1], r2 = rec[3 * (size_t)g + 2];
            // alpha >= 1/255  <=>  A dx^2 + B dx dy + C dy^2 >= thr (log2 domain, A/B/C pre-scaled and negative)
            const float cA = r0.z, cB = r0.w, cC = r1.x, thr = -log2f(255.f * r1.y);
            // diagonal half extents of the same ellipse: for q(d) = -(A dx^2 + B dx dy + C dy^2) <= -thr the extent
            // along u = x + y is sqrt(-thr * (iA + iC + 2 iB)) with the inverse form [[a, b/2],[b/2, c]]^-1
            const float a_ = -cA, b_ = -cB, c_ = -cC, det = a_ * c_ - 0.25f * b_ * b_;
            const float iA = c_ / det, iC = a_ / det, iB = -0.5f * b_ / det;
            const float hu = sqrtf(fmaxf(-thr * (iA + iC + 2.f * iB), 0.f)) * 1.002f + 0.03f;
            const float hv = sqrtf(fmaxf(-thr * (iA + iC - 2.f * iB), 0.f)) * 1.002f + 0.03f;
            const float gx = r0.x, gy = r0.y, hx = r2.y, hy = r2.z;
            for (int q = 0; q < 4; ++q) {
                const float x0 = (float)(tx * 16 + (q & 1) * 8), y0 = (float)(ty * 16 + (q >> 1) * 8);
                if ((gx - hx <= x0 + 7.f) && (gx + hx >= x0) && (gy - hy <= y0 + 7.f) && (gy + hy >= y0)) {
                    atomicAdd(&cnt[q][0], 1u);
                    for (int r = 0; r < 4; ++r) {
                        const float bx = x0 + (r & 1) * 4, by = y0 + (r >> 1) * 4;
                        if ((gx - hx <= bx + 3.f) && (gx + hx >= bx) && (gy - hy <= by + 3.f) && (gy + hy >= by)) {
                            atomicAdd(&cnt[q][1 + r], 1u);
                            bool any = false;
                            unsigned int hits = 0;
                            for (int py = 0; py < 4; ++py)
                                for (int px = 0; px < 4; ++px) {
                                    const float dx = gx - (bx + px), dy = gy - (by + py);
                                    const bool h = (cA * dx * dx + cB * dx * dy + cC * dy * dy) >= thr;
                                    any |= h;
                                    hits += h ? 1u : 0u;
                                }
                            if (any) atomicAdd(&exact_cnt, 1u);
                            atomicAdd(&pix_cnt, hits);
                            // octagon: the block's range of (x + y) and (x - y) against the diagonal extents
                            const float uc = gx + gy, vc = gx - gy;
                            const bool oct = (uc - hu <= bx + by + 6.f) && (uc + hu >= bx + by) && (vc - hv <= bx + 3.f - by) && (vc + hv >= bx - by - 3.f);
                            if (oct) atomicAdd(&oct_cnt, 1u);
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (tid == 0)
            for (int q = 0; q < 4; ++q) {
                q_iters += cnt[q][0];
                unsigned int mx = 0;
                for (int r = 0; r < 4; ++r) { mx = max(mx, cnt[q][1 + r]); b_visits += cnt[q][1 + r]; }
                b_iters += mx;
            }
        if (tid == 0) { e_visits += exact_cnt; o_visits += oct_cnt; p_hits += pix_cnt; }
        __syncthreads();
    }
    if (tid == 0) { atomicAdd(&out[0], q_iters); atomicAdd(&out[1], b_iters); atomicAdd(&out[2], b_visits); atomicAdd(&out[3], e_visits); atomicAdd(&out[4], o_visits); atomicAdd(&out[5], p_hits); }
}

extern "C" int cgs_debug_blend_occupancy(const cgs_raster_cfg *cfg, int64_t P, int64_t R, void *geom_ws, size_t geom_bytes,
                                         void *bin_ws, size_t bin_bytes, void *img_ws, size_t img_bytes, int64_t *out3,
                                         void *stream) {
    CgsGeom g;
    CgsBin b;
    CgsImg im;
    if (!cgs_img_carve(&im, img_ws, img_bytes, cfg->image_height, cfg->image_width) || !cgs_geom_carve(&g, geom_ws, geom_bytes, P) ||
        !cgs_bin_carve(&b, bin_ws, bin_bytes, P, R)) {
        cgs_set_error("debug_blend_occupancy: workspace");
        return CGS_ERR_WORKSPACE;
    }
    CGS_CHECK_HIP(hipMemsetAsync(out3, 0, 6 * sizeof(int64_t), (hipStream_t)stream));
    const int tx = cgs_tiles_x(cfg), ty = cgs_tiles_y(cfg);
    hipLaunchKernelGGL(blend_occupancy_kernel, dim3((unsigned)(tx * ty)), dim3(256), 0, (hipStream_t)stream, tx,
                       (const uint2 *)im.ranges, (const uint32_t *)b.gid_sorted, (const float4 *)g.rec,
                       (const uint32_t *)im.tile_last, (unsigned long long *)out3);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

// ---- splat-parallel backward mapping, counted instead of timed ----
// Mapping under study: lane = Gaussian of a bucket of 64 consecutive list entries, the wave walks the tile's pixels and
// carries each pixel's transmittance / colour recurrence across the bucket with DPP scans; a lane's nine gradient sums
// stay in its registers (no 16-lane reduction, no LDS atomics).  Its cost is the number of (pixel, bucket) visits times
// a wave-instruction count per visit, and its efficiency the fraction of the 64 lanes whose Gaussian actually touches
// the pixel.  out[0] = (pixel, bucket) pairs up to the pixel's own last contributor, out[1] = those with >= 1 lane hit,
// out[2] = lane hits (alpha >= 1/255) in them, out[3] = (4x4 block, bucket) pairs with >= 1 hit (a wave could skip whole
// blocks with one ballot), out[4] = buckets.
__global__ void __launch_bounds__(256)
    blend_splat_occupancy_kernel(int W, int H, int tiles_x, const uint2 *__restrict__ ranges,
                                 const uint32_t *__restrict__ gid_sorted, const float4 *__restrict__ rec,
                                 const uint32_t *__restrict__ tile_last, const uint32_t *__restrict__ n_contrib,
                                 unsigned long long *__restrict__ out) {
    __shared__ float4 s0[64], s1[64];
    __shared__ unsigned int blk_any[16];
    const int tile = blockIdx.x, tid = threadIdx.x;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const uint32_t tlast = tile_last[tile];
    const uint2 range = ranges[tile];
    const int px = tx * 16 + (tid & 15), py = ty * 16 + (tid >> 4);
    const int blk = (tid >> 6) * 4 + ((tid & 15) >> 2);
    const uint32_t my_last = (px < W && py < H) ? n_contrib[(size_t)py * W + px] : 0u;
    unsigned long long pairs = 0, visits = 0, hits = 0, blk_visits = 0, buckets = 0;
    for (uint32_t base = 0; base < tlast; base += 64) {
        __syncthreads();
        if (tid < 64) {
            const uint32_t pos = base + tid;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
            if (pos < tlast) {
                const uint32_t g = gid_sorted[range.x + pos];
                a = rec[3 * (size_t)g];
                b = rec[3 * (size_t)g + 1];
            } else {
                b.y = 0.f;      // opacity 0: never hits
            }
            s0[tid] = a; s1[tid] = b;
        }
        if (tid < 16) blk_any[tid] = 0;
        __syncthreads();
        unsigned int h = 0;
        if (base < my_last) {
            const uint32_t n = min(64u, my_last - base);
            for (uint32_t j = 0; j < n; ++j) {
                const float4 r0 = s0[j], r1 = s1[j];
                const float dx = r0.x - (float)px, dy = r0.y - (float)py;
                const float p2 = fmaf(r0.z * dx, dx, fmaf(r1.x * dy, dy, (r0.w * dx) * dy));
                const float alpha = fminf(0.99f, r1.y * __builtin_amdgcn_exp2f(p2));
                h += (p2 <= 0.f && alpha >= (1.0f / 255.0f)) ? 1u : 0u;
            }
            pairs += 1;
            visits += h ? 1 : 0;
            hits += h;
            if (h) atomicOr(&blk_any[blk], 1u);
        }
        __syncthreads();
        if (tid < 16) blk_visits += blk_any[tid];
        if (tid == 0) buckets += 1;
    }
    // block-level sums
    __shared__ unsigned long long red[5];
    if (tid < 5) red[tid] = 0;
    __syncthreads();
    atomicAdd(&red[0], pairs); atomicAdd(&red[1], visits); atomicAdd(&red[2], hits); atomicAdd(&red[3], blk_visits); atomicAdd(&red[4], buckets);
    __syncthreads();
    if (tid < 5) atomicAdd(&out[tid], red[tid]);
}

extern "C" int cgs_debug_blend_splat_occupancy(const cgs_raster_cfg *cfg, int64_t P, int64_t R, void *geom_ws, size_t geom_bytes,
                                               void *bin_ws, size_t bin_bytes, void *img_ws, size_t img_bytes, int64_t *out5,
                                               void *stream) {
    CgsGeom g;
    CgsBin b;
    CgsImg im;
    if (!cgs_img_carve(&im, img_ws, img_bytes, cfg->image_height, cfg->image_width) || !cgs_geom_carve(&g, geom_ws, geom_bytes, P) ||
        !cgs_bin_carve(&b, bin_ws, bin_bytes, P, R)) {
        cgs_set_error("debug_blend_splat_occupancy: workspace");
        return CGS_ERR_WORKSPACE;
    }
    CGS_CHECK_HIP(hipMemsetAsync(out5, 0, 5 * sizeof(int64_t), (hipStream_t)stream));
    const int tx = cgs_tiles_x(cfg), ty = cgs_tiles_y(cfg);
    hipLaunchKernelGGL(blend_splat_occupancy_kernel, dim3((unsigned)(tx * ty)), dim3(256), 0, (hipStream_t)stream,
                       cfg->image_width, cfg->image_height, tx, (const uint2 *)im.ranges, (const uint32_t *)b.gid_sorted,
                       (const float4 *)g.rec, (const uint32_t *)im.tile_last, (const uint32_t *)im.n_contrib,
                       (unsigned long long *)out5);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}
#endif  // CGS_EXPERIMENTS

// ---- test hook: the per-tile lists of csrc/tile_bin.hip against the round-1 binning (emit_pairs + 32-bit pair sort) ----
// Re-bins the geometry of the last forward into a SECOND binning workspace with the round-1 path and counts the entries
// of gid_sorted and of the tile ranges that differ from what the forward left in bin_ws / img_ws (out2[0], out2[1]).
// tests/test_raster_gpu.py::test_tile_lists_equal_the_pair_sort; declared in include/cgs.h as a test hook.
__global__ void __launch_bounds__(256)
    dbg_count_diff_kernel(int64_t n, const uint32_t *__restrict__ a, const uint32_t *__restrict__ b,
                          unsigned long long *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && a[i] != b[i]) atomicAdd(out, 1ull);
}

extern "C" int cgs_debug_bin_compare(const cgs_raster_cfg *cfg, int64_t P, int64_t R, int64_t R_ws /* the count bin_ws was carved with */,
                                     void *geom_ws, size_t geom_bytes,
                                     void *bin_ws, size_t bin_bytes, void *img_ws, size_t img_bytes, void *bin_ws2,
                                     size_t bin_bytes2, void *ranges2 /* [tiles] uint2 */, int64_t *out2, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CgsGeom g;
    CgsBin b, b2;
    CgsImg im, im2;
    if (!cgs_img_carve(&im, img_ws, img_bytes, cfg->image_height, cfg->image_width) || !cgs_geom_carve(&g, geom_ws, geom_bytes, P) ||
        !cgs_bin_carve(&b, bin_ws, bin_bytes, P, R_ws) || !cgs_bin_carve(&b2, bin_ws2, bin_bytes2, P, R)) {
        cgs_set_error("debug_bin_compare: workspace");
        return CGS_ERR_WORKSPACE;
    }
    CGS_CHECK_HIP(hipMemsetAsync(out2, 0, 2 * sizeof(int64_t), stream));
    if (R == 0) return CGS_OK;
    int rc;
    if ((rc = cgs_launch_emit_pairs(cfg, P, g, b2, stream))) return rc;
    int bits = 0;
    while ((1u << bits) < (uint32_t)(cgs_tiles_x(cfg) * cgs_tiles_y(cfg))) ++bits;
    if ((rc = cgs_sort_pairs_u32(b2.tile_key_a, b2.gid_a, b2.tile_key_c, b2.gid_sorted, b2.tile_key_b, b2.gid_b, R, 0, bits,
                                 b2.scratch, b2.scratch_bytes, stream)))
        return rc;
    im2 = im;
    im2.ranges = (uint2 *)ranges2;
    if ((rc = cgs_launch_ranges(cfg, R, b2, im2, stream))) return rc;
    hipLaunchKernelGGL(dbg_count_diff_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, stream, R,
                       (const uint32_t *)b.gid_sorted, (const uint32_t *)b2.gid_sorted, (unsigned long long *)out2);
    const int64_t nr = 2ll * cgs_tiles_x(cfg) * cgs_tiles_y(cfg);
    hipLaunchKernelGGL(dbg_count_diff_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, stream, nr,
                       (const uint32_t *)im.ranges, (const uint32_t *)ranges2, (unsigned long long *)out2 + 1);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

#ifdef CGS_EXPERIMENTS
// ---- analysis hook: wave iterations of the row mapping (four 4x4 blocks per wave) against an eight-group mapping (eight 4x2
// half-blocks per wave, 8-lane groups), both with the octagon test of raster_blend_rows.hip and the backward's per-group bound
// (entries behind the last contribution of the group's own pixels are dropped).  out4 = {iterations 4x4, iterations 4x2,
// (group, Gaussian) visits 4x4, visits 4x2, iterations 4x4 with rows that do not wait for each other at the 32-entry segment boundaries}.  tools/blend_occupancy.py; not part of include/cgs.h.
#include <hip/hip_fp16.h>
__device__ __forceinline__ bool dbg_oct_hits(float rx, float ry, float hx, float hy, float hu, float hv, float x0, float x1,
                                             float y0, float y1) {
    // pixel-centre ranges [x0, x1] x [y0, y1] (tile-relative) against box and diagonals
    return (rx - hx <= x1) && (rx + hx >= x0) && (ry - hy <= y1) && (ry + hy >= y0) && (rx + ry - hu <= x1 + y1) &&
           (rx + ry + hu >= x0 + y0) && (rx - ry - hv <= x1 - y0) && (rx - ry + hv >= x0 - y1);
}

__global__ void __launch_bounds__(256)
    blend_group_occupancy_kernel(int W, int H, int tiles_x, const uint2 *__restrict__ ranges,
                                 const uint32_t *__restrict__ gid_sorted, const float4 *__restrict__ rec,
                                 const uint32_t *__restrict__ tile_last, const uint32_t *__restrict__ n_contrib,
                                 unsigned long long *__restrict__ out) {
    __shared__ uint32_t sm16[256], sm32[256], snc[256], last16[16], last32[32];
    __shared__ uint32_t c16s[4][4][8];      // [wave][row][segment] visit counts of the current batch
    __shared__ unsigned long long red[5];
    const int tile = blockIdx.x, tid = threadIdx.x;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const uint32_t tlast = tile_last[tile];
    const uint2 range = ranges[tile];
    {
        const int px = tx * 16 + (tid & 15), py = ty * 16 + (tid >> 4);
        snc[tid] = (px < W && py < H) ? n_contrib[(size_t)py * W + px] : 0u;
    }
    if (tid < 5) red[tid] = 0;
    __syncthreads();
    if (tid < 16) {       // 4x4 block (bx, by)
        const int bx = tid & 3, by = tid >> 2;
        uint32_t m = 0;
        for (int y = 0; y < 4; ++y) for (int x = 0; x < 4; ++x) m = max(m, snc[(by * 4 + y) * 16 + bx * 4 + x]);
        last16[tid] = m;
    }
    if (tid < 32) {       // 4x2 half-block (bx, hy)
        const int bx = tid & 3, hy = tid >> 2;
        uint32_t m = 0;
        for (int y = 0; y < 2; ++y) for (int x = 0; x < 4; ++x) m = max(m, snc[(hy * 2 + y) * 16 + bx * 4 + x]);
        last32[tid] = m;
    }
    unsigned long long it16 = 0, it32 = 0, v16 = 0, v32 = 0, it16_free = 0;
    for (uint32_t base = 0; base < tlast; base += 256) {
        __syncthreads();
        uint32_t m16 = 0, m32 = 0;
        const uint32_t pos = base + tid;
        if (pos < tlast) {
            const uint32_t g = gid_sorted[range.x + pos];
            const float4 r0 = rec[3 * (size_t)g], r2 = rec[3 * (size_t)g + 2];
            const uint32_t db = __float_as_uint(r2.w);
            const float hu = __half2float(__ushort_as_half((unsigned short)(db & 0xFFFFu)));
            const float hv = __half2float(__ushort_as_half((unsigned short)(db >> 16)));
            const float rx = r0.x - (float)(tx * 16), ry = r0.y - (float)(ty * 16);
            for (int k = 0; k < 16; ++k) {
                const float x0 = (float)(4 * (k & 3)), y0 = (float)(4 * (k >> 2));
                if (dbg_oct_hits(rx, ry, r2.y, r2.z, hu, hv, x0, x0 + 3.f, y0, y0 + 3.f)) m16 |= 1u << k;
            }
            for (int k = 0; k < 32; ++k) {
                const float x0 = (float)(4 * (k & 3)), y0 = (float)(2 * (k >> 2));
                if (dbg_oct_hits(rx, ry, r2.y, r2.z, hu, hv, x0, x0 + 3.f, y0, y0 + 1.f)) m32 |= 1u << k;
            }
        }
        sm16[tid] = m16; sm32[tid] = m32;
        __syncthreads();
        if (tid < 32) {                   // (segment s, wave w): the wave's quadrant = blocks (2 qx + {0,1}, 2 qy + {0,1})
            const int s = tid >> 2, w = tid & 3, qx = w & 1, qy = w >> 1;
            uint32_t c16[4] = {0, 0, 0, 0}, c32[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int j = 0; j < 32; ++j) {
                const uint32_t position = base + (uint32_t)(s * 32 + j) + 1u;
                const uint32_t a = sm16[s * 32 + j], b = sm32[s * 32 + j];
                for (int k = 0; k < 4; ++k) {
                    const int blk = (2 * qy + (k >> 1)) * 4 + 2 * qx + (k & 1);
                    if (((a >> blk) & 1u) && position <= last16[blk]) ++c16[k];
                }
                for (int k = 0; k < 8; ++k) {
                    const int hb = (4 * qy + (k >> 1)) * 4 + 2 * qx + (k & 1);
                    if (((b >> hb) & 1u) && position <= last32[hb]) ++c32[k];
                }
            }
            uint32_t mx16 = 0, mx32 = 0;
            for (int k = 0; k < 4; ++k) { mx16 = max(mx16, c16[k]); v16 += c16[k]; c16s[w][k][s] = c16[k]; }
            for (int k = 0; k < 8; ++k) { mx32 = max(mx32, c32[k]); v32 += c32[k]; }
            it16 += mx16; it32 += mx32;
        }
        __syncthreads();
        if (tid < 4) {                    // rows that advance through the batch's segments independently: max over rows of the SUM
            uint32_t mx = 0;
            for (int k = 0; k < 4; ++k) {
                uint32_t t = 0;
                for (int s = 0; s < 8; ++s) t += c16s[tid][k][s];
                mx = max(mx, t);
            }
            it16_free += mx;
        }
    }
    atomicAdd(&red[4], it16_free);
    atomicAdd(&red[0], it16); atomicAdd(&red[1], it32); atomicAdd(&red[2], v16); atomicAdd(&red[3], v32);
    __syncthreads();
    if (tid < 5) atomicAdd(&out[tid], red[tid]);
}

extern "C" int cgs_debug_blend_group_occupancy(const cgs_raster_cfg *cfg, int64_t P, int64_t R, void *geom_ws, size_t geom_bytes,
                                               void *bin_ws, size_t bin_bytes, void *img_ws, size_t img_bytes, int64_t *out4,
                                               void *stream) {
    CgsGeom g;
    CgsBin b;
    CgsImg im;
    if (!cgs_img_carve(&im, img_ws, img_bytes, cfg->image_height, cfg->image_width) || !cgs_geom_carve(&g, geom_ws, geom_bytes, P) ||
        !cgs_bin_carve(&b, bin_ws, bin_bytes, P, R)) {
        cgs_set_error("debug_blend_group_occupancy: workspace");
        return CGS_ERR_WORKSPACE;
    }
    CGS_CHECK_HIP(hipMemsetAsync(out4, 0, 5 * sizeof(int64_t), (hipStream_t)stream));
    const int tx = cgs_tiles_x(cfg), ty = cgs_tiles_y(cfg);
    hipLaunchKernelGGL(blend_group_occupancy_kernel, dim3((unsigned)(tx * ty)), dim3(256), 0, (hipStream_t)stream,
                       cfg->image_width, cfg->image_height, tx, (const uint2 *)im.ranges, (const uint32_t *)b.gid_sorted,
                       (const float4 *)g.rec, (const uint32_t *)im.tile_last, (const uint32_t *)im.n_contrib,
                       (unsigned long long *)out4);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}
#endif  // CGS_EXPERIMENTS
