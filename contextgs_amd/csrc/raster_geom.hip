// Per-Gaussian stages of the tile rasterizer (R1/R2/R4/R6 of SURVEY §2.1):
// preprocess (project, cov3D -> conic, radius, tile rect), pair emission in
// depth order, tile ranges.  One lane per Gaussian, all arrays streamed once,
// coalesced; the camera matrices are wave-uniform and live in SGPRs.
//
// The reference's CUDA rasterizer is not in the mount; semantics follow the
// public 3DGS algorithm as recorded in SURVEY.md Appendix A and the call-site
// contract gaussian_renderer/__init__.py:179-205,250-285.
#include <hip/hip_fp16.h>
#include "cgs_internal.h"
#include "raster_math.h"
#include "raster_pre.h"

#define PRE_THREADS 256

template <bool FILTER_ONLY>
__global__ void __launch_bounds__(PRE_THREADS)
    preprocess_kernel(int64_t P, int W, int H, float tanfovx, float tanfovy, float scale_modifier,
                      const float *__restrict__ viewmatrix, const float *__restrict__ projmatrix,
                      const float *__restrict__ means3D, const float *__restrict__ colors,
                      const float *__restrict__ opacities, const float *__restrict__ scales,
                      const float *__restrict__ rotations, float4 *__restrict__ rec,
                      uint32_t *__restrict__ depth_key, uint32_t *__restrict__ tiles,
                      uint2 *__restrict__ rect, int32_t *__restrict__ radii) {
    const int64_t i = (int64_t)blockIdx.x * PRE_THREADS + threadIdx.x;
    if (i >= P) return;

    float V[16], Pm[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { V[k] = viewmatrix[k]; Pm[k] = projmatrix[k]; }

    const float3 p = make_float3(means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]);
    const float3 s = make_float3(scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]);
    const float4 q = make_float4(rotations[4 * i], rotations[4 * i + 1], rotations[4 * i + 2],
                                 rotations[4 * i + 3]);

    cgs_pre_fwd_one<FILTER_ONLY>(i, p, s, q, FILTER_ONLY ? 0.f : opacities[i], FILTER_ONLY ? 0.f : colors[3 * i],
                                 FILTER_ONLY ? 0.f : colors[3 * i + 1], FILTER_ONLY ? 0.f : colors[3 * i + 2], V, Pm, W, H, tanfovx,
                                 tanfovy, scale_modifier, rec, depth_key, tiles, rect, radii);
}

// prefilter_voxel (gaussian_renderer/__init__.py:232-287) in one launch: the reference evaluates get_scaling /
// get_rotation on all N rows and then reads three scale columns and rotation row 0 (:262-266, :283); here the kernel
// reads the raw scaling rows (exp applied to the three columns it uses unless the model is decoded), takes the ONE
// normalised rotation every anchor shares, and writes `radii_pure > 0` as a bool byte.  Same cgs_project as the
// rasterizer's preprocess, so the same anchors are visible.
__global__ void __launch_bounds__(PRE_THREADS)
    filter_voxel_kernel(int64_t N, int W, int H, float tanfovx, float tanfovy, float scale_modifier,
                        const float *__restrict__ viewmatrix, const float *__restrict__ projmatrix,
                        const float *__restrict__ means3D, const float *__restrict__ scaling, int64_t ld, int scales_are_log,
                        const float *__restrict__ rot1, uint8_t *__restrict__ visible) {
    const int64_t i = (int64_t)blockIdx.x * PRE_THREADS + threadIdx.x;
    if (i >= N) return;
    float V[16], Pm[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { V[k] = viewmatrix[k]; Pm[k] = projmatrix[k]; }
    const float3 p = make_float3(means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]);
    float sx = scaling[i * ld], sy = scaling[i * ld + 1], sz = scaling[i * ld + 2];
    if (scales_are_log) { sx = expf(sx); sy = expf(sy); sz = expf(sz); }
    const float4 q = make_float4(rot1[0], rot1[1], rot1[2], rot1[3]);
    CgsProj pr;
    const bool ok = cgs_project<float>(p, make_float3(sx, sy, sz), q, V, Pm, W, H, tanfovx, tanfovy, scale_modifier, pr);
    int32_t radius = 0;
    if (ok) {
        const int gx = (W + CGS_TILE - 1) / CGS_TILE, gy = (H + CGS_TILE - 1) / CGS_TILE;
        const float r = pr.radius;
        const int x0 = min(gx, max(0, (int)((pr.px - r) / (float)CGS_TILE)));
        const int y0 = min(gy, max(0, (int)((pr.py - r) / (float)CGS_TILE)));
        const int x1 = min(gx, max(0, (int)((pr.px + r + (float)(CGS_TILE - 1)) / (float)CGS_TILE)));
        const int y1 = min(gy, max(0, (int)((pr.py + r + (float)(CGS_TILE - 1)) / (float)CGS_TILE)));
        if ((x1 - x0) * (y1 - y0) > 0) radius = (int32_t)r;
    }
    visible[i] = radius > 0 ? 1 : 0;
}

int cgs_launch_filter_voxel(const cgs_raster_cfg *cfg, int64_t N, const float *means3D, const float *scaling, int64_t ld,
                            int scales_are_log, const float *rot1, uint8_t *visible, hipStream_t stream) {
    if (N == 0) return CGS_OK;
    CgsProfScope prof(CGS_PROF_FILTER, stream);
    hipLaunchKernelGGL(filter_voxel_kernel, dim3((unsigned)((N + PRE_THREADS - 1) / PRE_THREADS)), dim3(PRE_THREADS), 0, stream, N,
                       cfg->image_width, cfg->image_height, cfg->tanfovx, cfg->tanfovy, cfg->scale_modifier, cfg->viewmatrix,
                       cfg->projmatrix, means3D, scaling, ld, scales_are_log, rot1, visible);
    CGS_CHECK_LAUNCH(stream, cfg->debug);
    return CGS_OK;
}

int cgs_launch_preprocess(const cgs_raster_cfg *cfg, int64_t P, const float *means3D, const float *colors,
                          const float *opacities, const float *scales, const float *rotations, CgsGeom &g,
                          int32_t *radii, bool filter_only, hipStream_t stream) {
    if (P == 0) return CGS_OK;
    const unsigned nb = (unsigned)((P + PRE_THREADS - 1) / PRE_THREADS);
    CgsProfScope prof(filter_only ? CGS_PROF_FILTER : CGS_PROF_PREPROCESS, stream);
    if (filter_only) {
        hipLaunchKernelGGL(preprocess_kernel<true>, dim3(nb), dim3(PRE_THREADS), 0, stream, P,
                           cfg->image_width, cfg->image_height, cfg->tanfovx, cfg->tanfovy,
                           cfg->scale_modifier, cfg->viewmatrix, cfg->projmatrix, means3D,
                           (const float *)nullptr, (const float *)nullptr, scales, rotations,
                           (float4 *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint2 *)nullptr, radii);
    } else {
        hipLaunchKernelGGL(preprocess_kernel<false>, dim3(nb), dim3(PRE_THREADS), 0, stream, P,
                           cfg->image_width, cfg->image_height, cfg->tanfovx, cfg->tanfovy,
                           cfg->scale_modifier, cfg->viewmatrix, cfg->projmatrix, means3D, colors, opacities,
                           scales, rotations, g.rec, g.depth_key, g.tiles, g.rect, radii);
    }
    CGS_CHECK_LAUNCH(stream, cfg->debug);
    return CGS_OK;
}

// ---- offsets: tiles[] gathered into depth order (input of the scan) --------
__global__ void __launch_bounds__(256) gather_tiles_kernel(int64_t P, const uint32_t *__restrict__ order,
                                                           const uint32_t *__restrict__ tiles,
                                                           uint32_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < P) out[i] = tiles[order[i]];
}

__global__ void __launch_bounds__(256) iota_kernel(int64_t n, uint32_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (uint32_t)i;
}

int cgs_launch_iota(int64_t n, uint32_t *out, hipStream_t stream) {
    if (n == 0) return CGS_OK;
    hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, out);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

int cgs_launch_gather_tiles(int64_t P, const uint32_t *order, const uint32_t *tiles, uint32_t *out,
                            hipStream_t stream) {
    if (P == 0) return CGS_OK;
    hipLaunchKernelGGL(gather_tiles_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, stream, P, order,
                       tiles, out);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

// ---- pair emission (R4): walk Gaussians in depth order, write (tile, id) ---
__global__ void __launch_bounds__(256)
    emit_pairs_kernel(int64_t P, int tiles_x, const uint32_t *__restrict__ order,
                      const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ tiles,
                      const uint2 *__restrict__ rect, uint32_t *__restrict__ tile_key,
                      uint32_t *__restrict__ gid) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const uint32_t g = order[i];
    const uint2 rc = rect[g];          // (0,0) for a Gaussian that touches no tile: the loops below do not run
                                       // (one random gather per Gaussian instead of two: tiles[g] is not needed)
    const int x0 = rc.x & 0xFFFF, y0 = rc.x >> 16, x1 = rc.y & 0xFFFF, y1 = rc.y >> 16;
    uint32_t off = offsets[i];
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) {
            tile_key[off] = (uint32_t)(y * tiles_x + x);
            gid[off] = g;
            ++off;
        }
}

int cgs_launch_emit_pairs(const cgs_raster_cfg *cfg, int64_t P, CgsGeom &g, CgsBin &b, hipStream_t stream) {
    if (P == 0) return CGS_OK;
    CgsProfScope prof(CGS_PROF_EMIT_PAIRS, stream);
    hipLaunchKernelGGL(emit_pairs_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, stream, P,
                       cgs_tiles_x(cfg), g.order, g.offsets, g.tiles, g.rect, b.tile_key_a, b.gid_a);
    CGS_CHECK_LAUNCH(stream, cfg->debug);
    return CGS_OK;
}

// ---- tile ranges (R6) -------------------------------------------------------
__global__ void __launch_bounds__(256) ranges_kernel(int64_t R, const uint32_t *__restrict__ tile_key,
                                                     uint2 *__restrict__ ranges) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= R) return;
    const uint32_t t = tile_key[i];
    if (i == 0) ranges[t].x = 0;
    else {
        const uint32_t prev = tile_key[i - 1];
        if (prev != t) {
            ranges[prev].y = (uint32_t)i;
            ranges[t].x = (uint32_t)i;
        }
    }
    if (i == R - 1) ranges[t].y = (uint32_t)R;
}

int cgs_launch_ranges(const cgs_raster_cfg *cfg, int64_t R, CgsBin &b, CgsImg &im, hipStream_t stream) {
    const size_t nt = (size_t)cgs_tiles_x(cfg) * cgs_tiles_y(cfg);
    CGS_CHECK_HIP(hipMemsetAsync(im.ranges, 0, nt * sizeof(uint2), stream));
    if (R == 0) return CGS_OK;
    CgsProfScope prof(CGS_PROF_RANGES, stream);
    hipLaunchKernelGGL(ranges_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, stream, R,
                       (const uint32_t *)b.tile_key_c, im.ranges);
    CGS_CHECK_LAUNCH(stream, cfg->debug);
    return CGS_OK;
}
