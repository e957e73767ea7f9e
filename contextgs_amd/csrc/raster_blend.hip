// Tile alpha-blend forward (R7) and backward (R8, blend half) for gfx950.
//
// Mapping: one 256-thread workgroup per 16x16 tile; each of its 4 waves owns
// one 8x8 pixel quadrant (lane -> (lane&7, lane>>3)), so a whole wave
// terminates as soon as its 64 pixels are saturated and so that a Gaussian
// whose alpha>=1/255 footprint misses the quadrant is never evaluated by that
// wave.  Per batch of 256 list entries every thread gathers one 48-byte
// record into LDS and classifies it against the 4 quadrants; a 64-bit ballot
// per (source wave, quadrant) turns that into scalar bit masks, and each wave
// then walks only its own set bits with s_ff1 — no per-entry divergent test.
// All lanes read the same LDS record (broadcast ds_read_b128 x3).
//
// Semantics: SURVEY.md Appendix A (the CUDA source is not in the mount).
#include <stdlib.h>
#include "cgs_internal.h"

#define BLEND_THREADS 256
#define ALPHA_MIN (1.0f / 255.0f)
#define T_EPS 0.0001f
#define INV_LOG2E 0.6931471805599453f

struct BlendEval {
    float dx, dy, g, alpha;
    bool hit;
};

// One (pixel, Gaussian) evaluation; shared verbatim by forward and backward so
// both passes take bit-identical skip decisions.
__device__ __forceinline__ BlendEval blend_eval(const float4 r0, const float4 r1, float pxf, float pyf) {
    BlendEval e;
    e.dx = r0.x - pxf;
    e.dy = r0.y - pyf;
    const float p2 = fmaf(r0.z * e.dx, e.dx, fmaf(r1.x * e.dy, e.dy, (r0.w * e.dx) * e.dy));
    e.g = __builtin_amdgcn_exp2f(p2);
    e.alpha = fminf(0.99f, r1.y * e.g);
    e.hit = (p2 <= 0.f) && (e.alpha >= ALPHA_MIN);
    return e;
}

__device__ __forceinline__ uint32_t quadrant_mask(float gx, float gy, float hx, float hy, int tile_px0,
                                                  int tile_py0) {
    uint32_t m = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float x0 = (float)(tile_px0 + (q & 1) * 8), y0 = (float)(tile_py0 + (q >> 1) * 8);
        const bool ov = (gx - hx <= x0 + 7.f) && (gx + hx >= x0) && (gy - hy <= y0 + 7.f) && (gy + hy >= y0);
        m |= (ov ? 1u : 0u) << q;
    }
    return m;
}

__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// The product library has ONE blend path: the row-mapped kernels of raster_blend_rows.hip.  The quadrant-mapped kernels
// of this file (round 1's mapping) and every timing ablation exist only in -DCGS_EXPERIMENTS builds (tools/), where
// CGS_BLEND_ROWS=0 selects them.

int cgs_launch_blend_fwd(const cgs_raster_cfg *cfg, CgsGeom &g, CgsBin &b, CgsImg &im, float *out_color,
                         hipStream_t stream) {
    const int tx = cgs_tiles_x(cfg), ty = cgs_tiles_y(cfg);
    CgsProfScope prof(CGS_PROF_BLEND_FWD, stream);
    (void)tx; (void)ty;
    return cgs_launch_blend_fwd_rows(cfg, g, b, im, out_color, stream);
}

// ---------------------------------------------------------------------------
// Backward
// ---------------------------------------------------------------------------

int cgs_launch_blend_bwd(const cgs_raster_cfg *cfg, CgsGeom &g, CgsBin &b, CgsImg &im, const float *dL_dout,
                         float *dL_dmean2D_px, float *dL_dconic, float *dL_dopacity, float *dL_dcolors,
                         hipStream_t stream) {
    const int tx = cgs_tiles_x(cfg), ty = cgs_tiles_y(cfg);
    CgsProfScope prof(CGS_PROF_BLEND_BWD, stream);
    (void)tx; (void)ty;
    return cgs_launch_blend_bwd_rows(cfg, g, b, im, dL_dout, dL_dmean2D_px, dL_dconic, dL_dopacity, dL_dcolors, stream);
}

// R_eff / non-empty tile statistics for the roofline accounting.
__global__ void __launch_bounds__(256) raster_stats_kernel(int ntiles, const uint32_t *__restrict__ tile_last,
                                                           unsigned long long *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    unsigned long long v = 0, nz = 0;
    if (i < ntiles) { v = tile_last[i]; nz = v ? 1ull : 0ull; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        v += __shfl_xor(v, d, 64);
        nz += __shfl_xor(nz, d, 64);
    }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], v); atomicAdd(&out[1], nz); }
}

int cgs_launch_stats(const cgs_raster_cfg *cfg, CgsImg &im, int64_t *stats_out, hipStream_t stream) {
    const int nt = cgs_tiles_x(cfg) * cgs_tiles_y(cfg);
    CGS_CHECK_HIP(hipMemsetAsync(stats_out, 0, 2 * sizeof(int64_t), stream));
    hipLaunchKernelGGL(raster_stats_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, stream, nt,
                       (const uint32_t *)im.tile_last, (unsigned long long *)stats_out);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}
