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

#ifdef CGS_EXPERIMENTS   // round 1's quadrant-mapped kernels: experiment builds only (tools/)
__global__ void __launch_bounds__(BLEND_THREADS)
    blend_fwd_kernel(int W, int H, int tiles_x, const uint2 *__restrict__ ranges,
                     const uint32_t *__restrict__ gid_sorted, const float4 *__restrict__ rec,
                     const float *__restrict__ bg, float *__restrict__ out_color,
                     float *__restrict__ final_T, uint32_t *__restrict__ n_contrib,
                     uint32_t *__restrict__ tile_last) {
    __shared__ float4 srec[BLEND_THREADS * 3];
    __shared__ uint64_t qmask[4][4];   // [quadrant][source wave]
    __shared__ uint32_t wave_last[4];

    const int tile = blockIdx.x;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int px = tx * CGS_TILE + (wave & 1) * 8 + (lane & 7);
    const int py = ty * CGS_TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = ranges[tile];

    float T = 1.f, cr = 0.f, cg = 0.f, cb = 0.f;
    uint32_t last = 0;
    bool done = !inside;

    for (uint32_t start = range.x; start < range.y; start += BLEND_THREADS) {
        if (__syncthreads_count(done) == BLEND_THREADS) break;
        const uint32_t i = start + tid;
        uint32_t m4 = 0;
        if (i < range.y) {
            const uint32_t g = gid_sorted[i];
            const float4 r0 = rec[3 * (size_t)g], r1 = rec[3 * (size_t)g + 1], r2 = rec[3 * (size_t)g + 2];
            srec[tid * 3] = r0;
            srec[tid * 3 + 1] = r1;
            srec[tid * 3 + 2] = r2;
            m4 = quadrant_mask(r0.x, r0.y, r2.y, r2.z, tx * CGS_TILE, ty * CGS_TILE);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint64_t b = __ballot((m4 >> q) & 1u);
            if (lane == 0) qmask[q][wave] = b;
        }
        __syncthreads();
        const uint32_t base_pos = start - range.x;
        if (!__all(done)) {
            for (int s = 0; s < 4; ++s) {
                uint64_t m = uniform_u64(qmask[wave][s]);
                while (m) {
                    const int j = __builtin_ctzll(m);
                    m &= m - 1;
                    const int e = s * 64 + j;
                    const float4 r0 = srec[e * 3], r1 = srec[e * 3 + 1];
                    const float blue = srec[e * 3 + 2].x;
                    const BlendEval ev = blend_eval(r0, r1, pxf, pyf);
                    if (!done && ev.hit) {
                        const float test_T = T * (1.f - ev.alpha);
                        if (test_T < T_EPS) {
                            done = true;
                        } else {
                            const float w = ev.alpha * T;
                            cr = fmaf(r1.z, w, cr);
                            cg = fmaf(r1.w, w, cg);
                            cb = fmaf(blue, w, cb);
                            T = test_T;
                            last = base_pos + (uint32_t)e + 1u;
                        }
                    }
                }
                if (__all(done)) break;
            }
        }
    }

    if (inside) {
        const size_t pix = (size_t)py * W + px;
        const size_t hw = (size_t)H * W;
        final_T[pix] = T;
        n_contrib[pix] = last;
        out_color[pix] = fmaf(T, bg[0], cr);
        out_color[hw + pix] = fmaf(T, bg[1], cg);
        out_color[2 * hw + pix] = fmaf(T, bg[2], cb);
    }
    // deepest contributor of the tile: where the backward walk starts, and R_eff.
    uint32_t wl = last;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wl = max(wl, (uint32_t)__shfl_xor((int)wl, d, 64));
    if (lane == 0) wave_last[wave] = wl;
    __syncthreads();
    if (tid == 0) tile_last[tile] = max(max(wave_last[0], wave_last[1]), max(wave_last[2], wave_last[3]));
}

#endif  // CGS_EXPERIMENTS
// The product library has ONE blend path: the row-mapped kernels of raster_blend_rows.hip.  The quadrant-mapped kernels
// of this file (round 1's mapping) and every timing ablation exist only in -DCGS_EXPERIMENTS builds (tools/), where
// CGS_BLEND_ROWS=0 selects them.
#ifdef CGS_EXPERIMENTS
static bool blend_rows_enabled() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("CGS_BLEND_ROWS"); v = (e && e[0] == '0') ? 0 : 1; }
    return v != 0;
}
#endif

int cgs_launch_blend_fwd(const cgs_raster_cfg *cfg, CgsGeom &g, CgsBin &b, CgsImg &im, float *out_color,
                         hipStream_t stream) {
    const int tx = cgs_tiles_x(cfg), ty = cgs_tiles_y(cfg);
    CgsProfScope prof(CGS_PROF_BLEND_FWD, stream);
#ifndef CGS_EXPERIMENTS
    (void)tx; (void)ty;
    return cgs_launch_blend_fwd_rows(cfg, g, b, im, out_color, stream);
#else
    if (blend_rows_enabled()) return cgs_launch_blend_fwd_rows(cfg, g, b, im, out_color, stream);
    hipLaunchKernelGGL(blend_fwd_kernel, dim3((unsigned)(tx * ty)), dim3(BLEND_THREADS), 0, stream,
                       cfg->image_width, cfg->image_height, tx, (const uint2 *)im.ranges,
                       (const uint32_t *)b.gid_sorted, (const float4 *)g.rec, cfg->bg, out_color, im.final_T,
                       im.n_contrib, im.tile_last);
    CGS_CHECK_LAUNCH(stream, cfg->debug);
    return CGS_OK;
#endif
}

// ---------------------------------------------------------------------------
// Backward
// ---------------------------------------------------------------------------
#ifdef CGS_EXPERIMENTS
// Full-wave sum on the DPP network (no LDS crossbar): after the 6 steps lane 63
// holds the total of all 64 lanes.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
    return v + __int_as_float(moved);
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v = dpp_add<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xF>(v);   // row_half_mirror
    v = dpp_add<0x140, 0xF>(v);   // row_mirror
    v = dpp_add<0x142, 0xA>(v);   // row_bcast:15 -> rows 1,3
    v = dpp_add<0x143, 0xC>(v);   // row_bcast:31 -> rows 2,3
    return v;
}

template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}

#define NGRAD 9   // Mx, My, Sa, Sb, Sc, dop, dr, dg, db

template <int ABLATE>
__global__ void __launch_bounds__(BLEND_THREADS)
    blend_bwd_kernel(int W, int H, int tiles_x, const uint2 *__restrict__ ranges,
                     const uint32_t *__restrict__ gid_sorted, const float4 *__restrict__ rec,
                     const float *__restrict__ bg, const float *__restrict__ final_T,
                     const uint32_t *__restrict__ n_contrib, const uint32_t *__restrict__ tile_last,
                     const float *__restrict__ dL_dout, float *__restrict__ dL_dmean2D_px,
                     float *__restrict__ dL_dconic, float *__restrict__ dL_dopacity,
                     float *__restrict__ dL_dcolors) {
    __shared__ float4 srec[BLEND_THREADS * 3];
    __shared__ uint32_t sgid[BLEND_THREADS];
    __shared__ float sacc[BLEND_THREADS][NGRAD];
    __shared__ uint64_t qmask[4][4];

    const int tile = blockIdx.x;
    const uint32_t tlast = tile_last[tile];
    if (tlast == 0) return;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int px = tx * CGS_TILE + (wave & 1) * 8 + (lane & 7);
    const int py = ty * CGS_TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = ranges[tile];
    const size_t pix = (size_t)py * W + px, hw = (size_t)H * W;

    const float T_final = inside ? final_T[pix] : 0.f;
    const uint32_t my_last = inside ? n_contrib[pix] : 0u;
    float T = T_final;
    float gr = 0.f, gg = 0.f, gb = 0.f;
    if (inside) { gr = dL_dout[pix]; gg = dL_dout[hw + pix]; gb = dL_dout[2 * hw + pix]; }
    const float bg_dot = bg[0] * gr + bg[1] * gg + bg[2] * gb;
    const float neg_bg_T = -T_final * bg_dot;     // background term of dL/dalpha, per pixel constant
    // The published backward keeps accum_rec[3] (colour accumulated behind the current Gaussian) and the last
    // colour; only their dot products with the pixel's dL/dout enter dL/dalpha, so the recurrence is carried as ONE
    // scalar: acc_dot' = last_alpha * last_cdot + (1 - last_alpha) * acc_dot  (7 -> 3 instructions, and the three
    // colour differences fold into one subtraction)
    float acc_dot = 0.f, last_cdot = 0.f, last_alpha = 0.f;

    const int nbatch = (int)((tlast + BLEND_THREADS - 1) / BLEND_THREADS);
    for (int bi = nbatch - 1; bi >= 0; --bi) {
        const uint32_t base_pos = (uint32_t)bi * BLEND_THREADS;     // 0-based position of entry 0
        const uint32_t pos = base_pos + tid;
        uint32_t m4 = 0;
        __syncthreads();   // previous batch fully flushed before LDS is reused
        if (pos < tlast) {
            const uint32_t g = gid_sorted[range.x + pos];
            const float4 r0 = rec[3 * (size_t)g], r1 = rec[3 * (size_t)g + 1], r2 = rec[3 * (size_t)g + 2];
            srec[tid * 3] = r0;
            srec[tid * 3 + 1] = r1;
            srec[tid * 3 + 2] = r2;
            sgid[tid] = g;
            m4 = quadrant_mask(r0.x, r0.y, r2.y, r2.z, tx * CGS_TILE, ty * CGS_TILE);
        }
#pragma unroll
        for (int k = 0; k < NGRAD; ++k) sacc[tid][k] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint64_t b = __ballot((m4 >> q) & 1u);
            if (lane == 0) qmask[q][wave] = b;
        }
        __syncthreads();

        for (int s = 3; s >= 0; --s) {
            uint64_t m = uniform_u64(qmask[wave][s]);
            while (m) {
                const int j = 63 - __builtin_clzll(m);
                m &= ~(1ull << j);
                const int e = s * 64 + j;
                const uint32_t position = base_pos + (uint32_t)e + 1u;   // 1-based
                const float4 r0 = srec[e * 3], r1 = srec[e * 3 + 1];
                const float blue = srec[e * 3 + 2].x;
                const BlendEval ev = blend_eval(r0, r1, pxf, pyf);
                const bool act = (position <= my_last) && ev.hit;
                if (__ballot(act) == 0ull) continue;
                float v[NGRAD];
#pragma unroll
                for (int k = 0; k < NGRAD; ++k) v[k] = 0.f;
                if (act) {
                    const float om = 1.f - ev.alpha;
                    const float inv_om = 1.f / om;      // one IEEE division shared by the two quotients below
                    T = T * inv_om;
                    const float w = ev.alpha * T;
                    acc_dot = fmaf(last_alpha, last_cdot, (1.f - last_alpha) * acc_dot);
                    last_cdot = fmaf(r1.z, gr, fmaf(r1.w, gg, blue * gb));
                    float dL_dalpha = (last_cdot - acc_dot) * T;
                    last_alpha = ev.alpha;
                    dL_dalpha = fmaf(neg_bg_T, inv_om, dL_dalpha);
                    // dL/dG = opacity * dL/dalpha and d power2 / d (gx, gy) = (2A dx + B dy, 2C dy + B dx) have
                    // per-GAUSSIAN coefficients: the pixels sum g dL/dalpha times 1, dx, dy, dx^2, dx dy, dy^2 and
                    // the opacity factor and the 2x2 map are applied once at the flush
                    const float gG = ev.g * dL_dalpha;
                    const float gx = gG * ev.dx, gy = gG * ev.dy;
                    v[0] = gx;
                    v[1] = gy;
                    v[2] = gx * ev.dx;
                    v[3] = gx * ev.dy;
                    v[4] = gy * ev.dy;
                    v[5] = gG;
                    v[6] = w * gr;
                    v[7] = w * gg;
                    v[8] = w * gb;
                }
                if (ABLATE < 3) {
                    // Transposing reduction: instead of 9 independent 6-step wave sums (54 DPP adds) and 9 LDS
                    // atomics from one lane, two butterfly stages fold the 8 values v0..v7 onto the lanes of
                    // each quad (lane l ends up owning value 4j + (l&3)), two row rotations finish the sum over
                    // the 16-lane row, and the four rows add their partials with ONE ds_add_f32 (4-way same-
                    // address conflicts only).  26 VALU + 1 LDS instruction per (wave, Gaussian).
                    const bool b0 = lane & 1, b1 = lane & 2;
                    float a4[4], b2[2];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float keep = b0 ? v[2 * j + 1] : v[2 * j], send = b0 ? v[2 * j] : v[2 * j + 1];
                        a4[j] = keep + dpp_move<0xB1>(send);          // quad_perm [1,0,3,2]
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float keep = b1 ? a4[2 * j + 1] : a4[2 * j], send = b1 ? a4[2 * j] : a4[2 * j + 1];
                        b2[j] = keep + dpp_move<0x4E>(send);          // quad_perm [2,3,0,1]
                    }
                    float c8 = v[8];
                    c8 += dpp_move<0xB1>(c8);
                    c8 += dpp_move<0x4E>(c8);
                    b2[0] += dpp_move<0x124>(b2[0]); b2[0] += dpp_move<0x128>(b2[0]);   // row_ror:4, row_ror:8
                    b2[1] += dpp_move<0x124>(b2[1]); b2[1] += dpp_move<0x128>(b2[1]);
                    c8 += dpp_move<0x124>(c8); c8 += dpp_move<0x128>(c8);
                    const int sub = lane & 15;
                    if (ABLATE < 2) {
                        if (sub < NGRAD) atomicAdd(&sacc[e][sub], sub < 4 ? b2[0] : (sub < 8 ? b2[1] : c8));
                    } else {
                        asm volatile("" ::"v"(b2[0]), "v"(b2[1]), "v"(c8));
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < NGRAD; ++k) asm volatile("" ::"v"(v[k]));
                }
            }
        }
        __syncthreads();
        if (pos < tlast && ABLATE < 1) {
            const uint32_t g = sgid[tid];
            const float a0 = sacc[tid][0], a1 = sacc[tid][1], a2 = sacc[tid][2], a3 = sacc[tid][3],
                        a4 = sacc[tid][4], a5 = sacc[tid][5], a6 = sacc[tid][6], a7 = sacc[tid][7],
                        a8 = sacc[tid][8];
            if (a0 != 0.f || a1 != 0.f || a2 != 0.f || a3 != 0.f || a4 != 0.f || a5 != 0.f || a6 != 0.f ||
                a7 != 0.f || a8 != 0.f) {
                // power = power2 / log2(e); conic = (-2A, -B, -2C) / log2(e); (a0, a1) = sum gG (dx, dy) -> the mean
                // gradient through the Gaussian's own (pre-scaled) conic A, B, C = rec.z, rec.w, rec'.x
                const float4 q0 = srec[tid * 3], q1 = srec[tid * 3 + 1];
                const float cC = q1.x, op = q1.y;                 // a0..a4 carry dL/dalpha * g: times the opacity = dL/dG
                atomicAdd(&dL_dmean2D_px[2 * (size_t)g], op * fmaf(2.f * q0.z, a0, q0.w * a1) * INV_LOG2E);
                atomicAdd(&dL_dmean2D_px[2 * (size_t)g + 1], op * fmaf(2.f * cC, a1, q0.w * a0) * INV_LOG2E);
                atomicAdd(&dL_dconic[3 * (size_t)g], -0.5f * op * a2);
                atomicAdd(&dL_dconic[3 * (size_t)g + 1], -op * a3);
                atomicAdd(&dL_dconic[3 * (size_t)g + 2], -0.5f * op * a4);
                atomicAdd(&dL_dopacity[g], a5);
                atomicAdd(&dL_dcolors[3 * (size_t)g], a6);
                atomicAdd(&dL_dcolors[3 * (size_t)g + 1], a7);
                atomicAdd(&dL_dcolors[3 * (size_t)g + 2], a8);
            }
        }
    }
}

#endif  // CGS_EXPERIMENTS

int cgs_launch_blend_bwd(const cgs_raster_cfg *cfg, CgsGeom &g, CgsBin &b, CgsImg &im, const float *dL_dout,
                         float *dL_dmean2D_px, float *dL_dconic, float *dL_dopacity, float *dL_dcolors,
                         hipStream_t stream) {
    const int tx = cgs_tiles_x(cfg), ty = cgs_tiles_y(cfg);
    CgsProfScope prof(CGS_PROF_BLEND_BWD, stream);
#ifndef CGS_EXPERIMENTS
    (void)tx; (void)ty;
    return cgs_launch_blend_bwd_rows(cfg, g, b, im, dL_dout, dL_dmean2D_px, dL_dconic, dL_dopacity, dL_dcolors, stream);
#else
    if (blend_rows_enabled() && !getenv("CGS_BWD_ABLATE")) {
        return cgs_launch_blend_bwd_rows(cfg, g, b, im, dL_dout, dL_dmean2D_px, dL_dconic, dL_dopacity, dL_dcolors, stream);
    }
    static int ablate = -1;     // CGS_BWD_ABLATE=1..3: timing experiments only (wrong results)
    if (ablate < 0) { const char *e = getenv("CGS_BWD_ABLATE"); ablate = e ? atoi(e) : 0; }
#define BWD_LAUNCH(A)                                                                                               \
    hipLaunchKernelGGL(blend_bwd_kernel<A>, dim3((unsigned)(tx * ty)), dim3(BLEND_THREADS), 0, stream,             \
                       cfg->image_width, cfg->image_height, tx, (const uint2 *)im.ranges,                          \
                       (const uint32_t *)b.gid_sorted, (const float4 *)g.rec, cfg->bg, (const float *)im.final_T,   \
                       (const uint32_t *)im.n_contrib, (const uint32_t *)im.tile_last, dL_dout, dL_dmean2D_px,      \
                       dL_dconic, dL_dopacity, dL_dcolors)
    switch (ablate) { case 1: BWD_LAUNCH(1); break; case 2: BWD_LAUNCH(2); break; case 3: BWD_LAUNCH(3); break; default: BWD_LAUNCH(0); }
#undef BWD_LAUNCH
    CGS_CHECK_LAUNCH(stream, cfg->debug);
    return CGS_OK;
#endif
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
