// Tile alpha-blend forward / backward, ROW-MAPPED variant: the four 16-lane DPP rows of a wave own one 4x4 pixel
// block each (together the wave's 8x8 quadrant) and walk their OWN Gaussian lists, so one wave iteration processes
// up to four different Gaussians.
//
// Why: the blend kernels are VALU-issue bound (~95 wave instructions per (quadrant, Gaussian)
// visit, all 64 lanes issue them however few pixels contribute).  In the dense-anchor scenes of the headline
// benchmark a Gaussian's alpha >= 1/255 footprint overlaps on average only 2.1 of the four 4x4 blocks of a quadrant
// it touches (tools/blend_occupancy.py: 11.9 M quadrant visits per 1080p view, 7.6 M row-mapped iterations), so
// letting the rows advance independently removes a third of the iterations.  Everything the reduction needs already
// happens inside a 16-lane row (two quad butterflies + two row rotations), and each row adds into the LDS
// accumulator of ITS Gaussian (fewer same-address conflicts than four rows adding into one).
//
// Same arithmetic per (pixel, Gaussian) as raster_blend.hip (blend_eval is identical), same list order per pixel, so
// results are bit-identical to the quadrant-mapped kernels for the forward and equal up to the summation order of
// the per-Gaussian partial sums for the backward.
#include <stdlib.h>
#include <hip/hip_fp16.h>
#include "cgs_internal.h"

#define RB_THREADS 256
#define RB_ALPHA_MIN (1.0f / 255.0f)
#define RB_T_EPS 0.0001f
#define RB_INV_LOG2E 0.6931471805599453f
#define RB_NGRAD 9

namespace {

struct RbEval { float dx, dy, g, alpha; bool hit; };

__device__ __forceinline__ RbEval rb_eval(const float4 r0, const float4 r1, float pxf, float pyf) {
    RbEval e;
    e.dx = r0.x - pxf;
    e.dy = r0.y - pyf;
    const float p2 = fmaf(r0.z * e.dx, e.dx, fmaf(r1.x * e.dy, e.dy, (r0.w * e.dx) * e.dy));
    e.g = __builtin_amdgcn_exp2f(p2);
    e.alpha = fminf(0.99f, r1.y * e.g);
    e.hit = (p2 <= 0.f) && (e.alpha >= RB_ALPHA_MIN);
    return e;
}

// 16-bit mask of the 4x4-pixel blocks (row-major 4x4 grid of the 16x16 tile) the alpha >= 1/255 ellipse can reach: its
// bounding box (hx, hy) AND its extents along the two diagonals (`diag` = two fp16 halves: half extent of x + y, of x - y,
// from the preprocess kernel) — an octagon around the ellipse.  Output-invariant: a culled block has no pixel the blend loop
// would not skip anyway; on the bench scene the diagonals drop 10 % of the box test's block visits
// (tools/blend_occupancy.py: 25.4 M -> 22.8 M, 21.8 M have a contributing pixel).
__device__ __forceinline__ uint32_t rb_block_mask(float gx, float gy, float hx, float hy, float diag, int tile_px0,
                                                  int tile_py0) {
    // everything relative to the tile's first pixel: the block bounds are literals (no per-tile registers kept live
    // across the batch loop — 28 of them cost the forward kernel half its occupancy when they were hoisted)
    const float rx = gx - (float)tile_px0, ry = gy - (float)tile_py0;
    const float xl = rx - hx, xh = rx + hx, yl = ry - hy, yh = ry + hy;
    uint32_t xm = 0, ym = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        xm |= ((xl <= (float)(4 * k + 3)) && (xh >= (float)(4 * k))) ? (1u << k) : 0u;
        ym |= ((yl <= (float)(4 * k + 3)) && (yh >= (float)(4 * k))) ? (1u << k) : 0u;
    }
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) m |= ((ym >> k) & 1u) ? (xm << (4 * k)) : 0u;
    const uint32_t db = __float_as_uint(diag);
    const float hu = __half2float(__ushort_as_half((unsigned short)(db & 0xFFFFu)));
    const float hv = __half2float(__ushort_as_half((unsigned short)(db >> 16)));
    const float ul = rx + ry - hu, uh = rx + ry + hu, vl = rx - ry - hv, vh = rx - ry + hv;
    // blocks with bx + by = s share the range [4 s, 4 s + 6] of x + y; blocks with bx - by = d the range
    // [4 d - 3, 4 d + 3] of x - y (bit index of a block = 4 by + bx)
    const uint32_t DU[7] = {0x0001u, 0x0012u, 0x0124u, 0x1248u, 0x2480u, 0x4800u, 0x8000u};
    const uint32_t DV[7] = {0x1000u, 0x2100u, 0x4210u, 0x8421u, 0x0842u, 0x0084u, 0x0008u};
    uint32_t um = 0, vm = 0;
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        um |= ((ul <= (float)(4 * s + 6)) && (uh >= (float)(4 * s))) ? DU[s] : 0u;
        vm |= ((vl <= (float)(4 * (s - 3) + 3)) && (vh >= (float)(4 * (s - 3) - 3))) ? DV[s] : 0u;
    }
    return m & um & vm;
}

template <int CTRL>
__device__ __forceinline__ float rb_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}

struct RbLane {
    int px, py, blk;
};

// lane -> pixel: wave = quadrant, row (lane >> 4) = 4x4 block of the quadrant, lane & 15 = pixel of the block
__device__ __forceinline__ RbLane rb_lane(int tx, int ty, int wave, int lane) {
    const int row = lane >> 4, i = lane & 15;
    const int bx = (wave & 1) * 2 + (row & 1), by = (wave >> 1) * 2 + (row >> 1);
    RbLane l;
    l.px = tx * CGS_TILE + bx * 4 + (i & 3);
    l.py = ty * CGS_TILE + by * 4 + (i >> 2);
    l.blk = by * 4 + bx;
    return l;
}

}  // namespace

__global__ void __launch_bounds__(RB_THREADS)
    blend_fwd_rows_kernel(int W, int H, int tiles_x, const uint2 *__restrict__ ranges,
                          const uint32_t *__restrict__ gid_sorted, const float4 *__restrict__ rec,
                          const float *__restrict__ bg, float *__restrict__ out_color, float *__restrict__ final_T,
                          uint32_t *__restrict__ n_contrib, uint32_t *__restrict__ tile_last) {
    __shared__ float4 srec[RB_THREADS * 3];
    __shared__ uint32_t bmask[16][8];     // [block][32-entry segment]: 32-bit masks keep the per-lane bit walk cheap
    __shared__ uint32_t wave_last[4];

    const int tile = blockIdx.x;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const RbLane L = rb_lane(tx, ty, wave, lane);
    const bool inside = L.px < W && L.py < H;
    const float pxf = (float)L.px, pyf = (float)L.py;
    const uint2 range = ranges[tile];

    float T = 1.f, cr = 0.f, cg = 0.f, cb = 0.f;
    uint32_t last = 0;
    bool done = !inside;

    // The batch after the one being walked is fetched into registers BEFORE the walk starts (gid -> three dependent
    // 16-byte gathers, ~2 us of latency per batch) and handed to LDS at the top of the next round.
    float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), p1 = p0, p2 = p0;
    bool pvalid = range.x + tid < range.y;
    if (pvalid) {
        const uint32_t g = gid_sorted[range.x + tid];
        p0 = rec[3 * (size_t)g]; p1 = rec[3 * (size_t)g + 1]; p2 = rec[3 * (size_t)g + 2];
    }
    for (uint32_t start = range.x; start < range.y; start += RB_THREADS) {
        if (__syncthreads_count(done) == RB_THREADS) break;
        uint32_t m16 = 0;
        if (pvalid) {
            srec[tid * 3] = p0;
            srec[tid * 3 + 1] = p1;
            srec[tid * 3 + 2] = p2;
            m16 = rb_block_mask(p0.x, p0.y, p2.y, p2.z, p2.w, tx * CGS_TILE, ty * CGS_TILE);
        }
        {
            const uint32_t i = start + RB_THREADS + tid;
            pvalid = i < range.y;
            if (pvalid) {
                const uint32_t g = gid_sorted[i];
                p0 = rec[3 * (size_t)g]; p1 = rec[3 * (size_t)g + 1]; p2 = rec[3 * (size_t)g + 2];
            }
        }
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            const uint64_t bal = __ballot((m16 >> b) & 1u);
            if (lane == 0) { bmask[b][2 * wave] = (uint32_t)bal; bmask[b][2 * wave + 1] = (uint32_t)(bal >> 32); }
        }
        __syncthreads();
        const uint32_t base_pos = start - range.x;
        if (!__all(done)) {
            for (int s = 0; s < 8; ++s) {
                uint32_t m = done ? 0u : bmask[L.blk][s];
                while (__ballot(m != 0u) != 0ull) {
                    const bool has = m != 0u;
                    const int j = has ? __builtin_ctz(m) : 0;
                    m &= m - 1u;                                     // 0 stays 0
                    const int e = s * 32 + j;
                    const float4 r0 = srec[e * 3], r1 = srec[e * 3 + 1];
                    const float blue = srec[e * 3 + 2].x;
                    const RbEval ev = rb_eval(r0, r1, pxf, pyf);
                    // branch-free (as in the backward): a lane that takes no contribution blends weight 0, an exact no-op
                    const bool act = has && !done && ev.hit;
                    const float alpha = act ? ev.alpha : 0.f;
                    const float test_T = T * (1.f - alpha);
                    const bool stop = act && test_T < RB_T_EPS;
                    const bool upd = act && !stop;
                    const float w = upd ? alpha * T : 0.f;
                    cr = fmaf(r1.z, w, cr);
                    cg = fmaf(r1.w, w, cg);
                    cb = fmaf(blue, w, cb);
                    T = upd ? test_T : T;
                    last = upd ? base_pos + (uint32_t)e + 1u : last;
                    done = done || stop;
                }
                if (__all(done)) break;
            }
        }
    }

    if (inside) {
        const size_t pix = (size_t)L.py * W + L.px;
        const size_t hw = (size_t)H * W;
        final_T[pix] = T;
        n_contrib[pix] = last;
        out_color[pix] = fmaf(T, bg[0], cr);
        out_color[hw + pix] = fmaf(T, bg[1], cg);
        out_color[2 * hw + pix] = fmaf(T, bg[2], cb);
    }
    uint32_t wl = last;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wl = max(wl, (uint32_t)__shfl_xor((int)wl, d, 64));
    if (lane == 0) wave_last[wave] = wl;
    __syncthreads();
    if (tid == 0) tile_last[tile] = max(max(wave_last[0], wave_last[1]), max(wave_last[2], wave_last[3]));
}

template <int ABL>      // ABL != 0: timing ablations (WRONG results), see cgs_launch_blend_bwd_rows
__global__ void __launch_bounds__(RB_THREADS)
    blend_bwd_rows_kernel(int W, int H, int tiles_x, const uint2 *__restrict__ ranges,
                          const uint32_t *__restrict__ gid_sorted, const float4 *__restrict__ rec,
                          const float *__restrict__ bg, const float *__restrict__ final_T,
                          const uint32_t *__restrict__ n_contrib, const uint32_t *__restrict__ tile_last,
                          const float *__restrict__ dL_dout, float *__restrict__ dL_dmean2D_px,
                          float *__restrict__ dL_dconic, float *__restrict__ dL_dopacity,
                          float *__restrict__ dL_dcolors) {
    __shared__ float4 srec[RB_THREADS * 3];
    __shared__ uint32_t sgid[RB_THREADS];
    __shared__ float sacc[RB_THREADS][RB_NGRAD];
    __shared__ uint32_t bmask[16][8];     // [block][32-entry segment]: 32-bit masks keep the per-lane bit walk cheap

    const int tile = blockIdx.x;
    const uint32_t tlast = tile_last[tile];
    if (tlast == 0) return;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const RbLane L = rb_lane(tx, ty, wave, lane);
    const bool inside = L.px < W && L.py < H;
    const float pxf = (float)L.px, pyf = (float)L.py;
    const uint2 range = ranges[tile];
    const size_t pix = (size_t)L.py * W + L.px, hw = (size_t)H * W;

    const float T_final = inside ? final_T[pix] : 0.f;
    const uint32_t my_last = inside ? n_contrib[pix] : 0u;
    uint32_t blk_last = my_last;       // maximum over the 16 lanes (pixels) of the row
    blk_last = max(blk_last, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)blk_last, 0xB1, 0xF, 0xF, false));
    blk_last = max(blk_last, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)blk_last, 0x4E, 0xF, 0xF, false));
    blk_last = max(blk_last, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)blk_last, 0x124, 0xF, 0xF, false));
    blk_last = max(blk_last, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)blk_last, 0x128, 0xF, 0xF, false));
    float T = T_final;
    float gr = 0.f, gg = 0.f, gb = 0.f;
    if (inside) { gr = dL_dout[pix]; gg = dL_dout[hw + pix]; gb = dL_dout[2 * hw + pix]; }
    const float bg_dot = bg[0] * gr + bg[1] * gg + bg[2] * gb;
    const float neg_bg_T = -T_final * bg_dot;
    float acc_dot = 0.f, last_cdot = 0.f, last_alpha = 0.f;       // scalar colour recurrence (see raster_blend.hip)

    const int nbatch = (int)((tlast + RB_THREADS - 1) / RB_THREADS);
    // the batch after the one being walked is fetched into registers before the walk starts (see the forward)
    float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), p1 = p0, p2 = p0;
    uint32_t pg = 0;
    {
        const uint32_t pos0 = (uint32_t)(nbatch - 1) * RB_THREADS + tid;
        if (pos0 < tlast) {
            pg = gid_sorted[range.x + pos0];
            p0 = rec[3 * (size_t)pg]; p1 = rec[3 * (size_t)pg + 1]; p2 = rec[3 * (size_t)pg + 2];
        }
    }
    for (int bi = nbatch - 1; bi >= 0; --bi) {
        const uint32_t base_pos = (uint32_t)bi * RB_THREADS;
        const uint32_t pos = base_pos + tid;
        uint32_t m16 = 0;
        __syncthreads();   // previous batch fully flushed before LDS is reused
        if (pos < tlast) {
            srec[tid * 3] = p0;
            srec[tid * 3 + 1] = p1;
            srec[tid * 3 + 2] = p2;
            sgid[tid] = pg;
            m16 = rb_block_mask(p0.x, p0.y, p2.y, p2.z, p2.w, tx * CGS_TILE, ty * CGS_TILE);
        }
        if (bi > 0) {      // every position of an earlier batch is < tlast
            pg = gid_sorted[range.x + pos - RB_THREADS];
            p0 = rec[3 * (size_t)pg]; p1 = rec[3 * (size_t)pg + 1]; p2 = rec[3 * (size_t)pg + 2];
        }
#pragma unroll
        for (int k = 0; k < RB_NGRAD; ++k) sacc[tid][k] = 0.f;
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            const uint64_t bal = __ballot((m16 >> b) & 1u);
            if (lane == 0) { bmask[b][2 * wave] = (uint32_t)bal; bmask[b][2 * wave + 1] = (uint32_t)(bal >> 32); }
        }
        __syncthreads();

        for (int s = 7; s >= 0; --s) {
            // entries behind the LAST contribution of every pixel of this 4x4 block (n_contrib: where the forward stopped)
            // cannot contribute to it: the row drops them from its list (the tile-wide bound `tlast` is the maximum over
            // 256 pixels, a block's own bound over 16)
            const int lim = (int)blk_last - (int)base_pos - s * 32 - 1;       // highest admissible bit of this segment
            uint32_t m = bmask[L.blk][s];
            m = lim < 0 ? 0u : (lim >= 31 ? m : (m & ((2u << lim) - 1u)));
            while (__ballot(m != 0u) != 0ull) {
                const bool has = m != 0u;
                const int j = 31 - __builtin_clz(m | 1u);            // m == 0: j = 0, has = false
                m &= ~(1u << j);
                const int e = s * 32 + j;
                const uint32_t position = base_pos + (uint32_t)e + 1u;   // 1-based
                const float4 r0 = srec[e * 3], r1 = srec[e * 3 + 1];
                const float blue = srec[e * 3 + 2].x;
                if (ABL == 4) { if (has && r0.x == 12345.f) atomicAdd(&sacc[e][0], r1.x + blue); continue; }
                const RbEval ev = rb_eval(r0, r1, pxf, pyf);
                const bool act = has && (position <= my_last) && ev.hit;
                if ((ABL != 7) && __ballot(act) == 0ull) continue;
                if (ABL == 3) { if (act && ev.alpha == 12345.f) atomicAdd(&sacc[e][0], ev.g + blue); continue; }
                // Branch-free: a lane whose pixel takes no contribution runs the same updates on alpha = 0, G = 0, for which
                // every one of them is an exact no-op (T / 1 = T, w = 0, the colour recurrence with alpha = 0 hands on
                // the value the next contributing step would have computed) — two selects instead of a divergent block,
                // nine zero-initialisations and the moves that merge its results (the kernel is VALU-issue bound).
                const float alpha = act ? ev.alpha : 0.f, Gm = act ? ev.g : 0.f;
                const float om = 1.f - alpha;
                // 1/(1 - alpha), alpha <= 0.99: v_rcp_f32 + one Newton step (3 instructions, <= 1 ulp) instead of the
                // IEEE division sequence (10)
                float inv_om = __builtin_amdgcn_rcpf(om);
                inv_om = inv_om * fmaf(-om, inv_om, 2.f);
                T = T * inv_om;
                const float w = alpha * T;
                acc_dot = fmaf(last_alpha, last_cdot, (1.f - last_alpha) * acc_dot);
                last_cdot = fmaf(r1.z, gr, fmaf(r1.w, gg, blue * gb));
                float dL_dalpha = (last_cdot - acc_dot) * T;
                last_alpha = alpha;
                dL_dalpha = fmaf(neg_bg_T, inv_om, dL_dalpha);
                const float gG = Gm * dL_dalpha;
                const float gx = gG * ev.dx, gy = gG * ev.dy;
                float v[RB_NGRAD];
                v[0] = gx;
                v[1] = gy;
                v[2] = gx * ev.dx;
                v[3] = gx * ev.dy;
                v[4] = gy * ev.dy;
                v[5] = gG;
                v[6] = w * gr;
                v[7] = w * gg;
                v[8] = w * gb;
                if (ABL == 2) {
                    const float sm = v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7] + v[8];
                    if (sm == 12345.f) atomicAdd(&sacc[e][0], sm);
                    continue;
                }
                // transposing reduction inside each 16-lane row (identical to raster_blend.hip); every row then adds
                // into the accumulator of ITS OWN Gaussian
                const bool b0 = lane & 1, b1 = lane & 2;
                float a4[4], b2[2];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float keep = b0 ? v[2 * q + 1] : v[2 * q], send = b0 ? v[2 * q] : v[2 * q + 1];
                    a4[q] = keep + rb_dpp<0xB1>(send);
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float keep = b1 ? a4[2 * q + 1] : a4[2 * q], send = b1 ? a4[2 * q] : a4[2 * q + 1];
                    b2[q] = keep + rb_dpp<0x4E>(send);
                }
                float c8 = v[8];
                c8 += rb_dpp<0xB1>(c8);
                c8 += rb_dpp<0x4E>(c8);
                b2[0] += rb_dpp<0x124>(b2[0]); b2[0] += rb_dpp<0x128>(b2[0]);
                b2[1] += rb_dpp<0x124>(b2[1]); b2[1] += rb_dpp<0x128>(b2[1]);
                c8 += rb_dpp<0x124>(c8); c8 += rb_dpp<0x128>(c8);
                // keep the last three DPP additions in front of the predicated store: sunk into its exec-masked block
                // they split into a full-exec v_mov_dpp plus an add each (and a zero for the mov's `old` operand)
                asm volatile("" : "+v"(b2[0]), "+v"(b2[1]), "+v"(c8));
                const int sub = lane & 15;
                const float red = sub < 4 ? b2[0] : (sub < 8 ? b2[1] : c8);
                if (ABL == 1) { if (has && red == 12345.f) atomicAdd(&sacc[e][sub], red); continue; }
                if (ABL == 6) { if (has && sub < RB_NGRAD) sacc[e][sub] = red; continue; }
                if (has && sub < RB_NGRAD) atomicAdd(&sacc[e][sub], red);
            }
        }
        __syncthreads();
        if (ABL == 5 && tlast != 0x7fffffffu) continue;
        if (pos < tlast) {
            const uint32_t g = sgid[tid];
            const float a0 = sacc[tid][0], a1 = sacc[tid][1], a2 = sacc[tid][2], a3 = sacc[tid][3],
                        a4 = sacc[tid][4], a5 = sacc[tid][5], a6 = sacc[tid][6], a7 = sacc[tid][7],
                        a8 = sacc[tid][8];
            if (a0 != 0.f || a1 != 0.f || a2 != 0.f || a3 != 0.f || a4 != 0.f || a5 != 0.f || a6 != 0.f ||
                a7 != 0.f || a8 != 0.f) {
                const float4 q0 = srec[tid * 3], q1 = srec[tid * 3 + 1];
                const float cC = q1.x, op = q1.y;
                atomicAdd(&dL_dmean2D_px[2 * (size_t)g], op * fmaf(2.f * q0.z, a0, q0.w * a1) * RB_INV_LOG2E);
                atomicAdd(&dL_dmean2D_px[2 * (size_t)g + 1], op * fmaf(2.f * cC, a1, q0.w * a0) * RB_INV_LOG2E);
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

int cgs_launch_blend_fwd_rows(const cgs_raster_cfg *cfg, CgsGeom &g, CgsBin &b, CgsImg &im, float *out_color,
                              hipStream_t stream) {
    const int tx = cgs_tiles_x(cfg), ty = cgs_tiles_y(cfg);
    hipLaunchKernelGGL(blend_fwd_rows_kernel, dim3((unsigned)(tx * ty)), dim3(RB_THREADS), 0, stream, cfg->image_width,
                       cfg->image_height, tx, (const uint2 *)im.ranges, (const uint32_t *)b.gid_sorted,
                       (const float4 *)g.rec, cfg->bg, out_color, im.final_T, im.n_contrib, im.tile_last);
    CGS_CHECK_LAUNCH(stream, cfg->debug);
    return CGS_OK;
}

int cgs_launch_blend_bwd_rows(const cgs_raster_cfg *cfg, CgsGeom &g, CgsBin &b, CgsImg &im, const float *dL_dout,
                              float *dL_dmean2D_px, float *dL_dconic, float *dL_dopacity, float *dL_dcolors,
                              hipStream_t stream) {
    const int tx = cgs_tiles_x(cfg), ty = cgs_tiles_y(cfg);
#define RB_BWD(A)                                                                                                     \
    hipLaunchKernelGGL(blend_bwd_rows_kernel<A>, dim3((unsigned)(tx * ty)), dim3(RB_THREADS), 0, stream,              \
                       cfg->image_width, cfg->image_height, tx, (const uint2 *)im.ranges,                             \
                       (const uint32_t *)b.gid_sorted, (const float4 *)g.rec, cfg->bg, (const float *)im.final_T,      \
                       (const uint32_t *)im.n_contrib, (const uint32_t *)im.tile_last, dL_dout, dL_dmean2D_px,         \
                       dL_dconic, dL_dopacity, dL_dcolors)
#ifndef CGS_EXPERIMENTS
    RB_BWD(0);
#else
    static int abl = -1;        // CGS_ROWS_ABL=1..7: timing experiments only (wrong gradients), tools/rows_ablate.sh
    if (abl < 0) { const char *e = getenv("CGS_ROWS_ABL"); abl = e ? atoi(e) : 0; }
    switch (abl) {
        case 1: RB_BWD(1); break;
        case 2: RB_BWD(2); break;
        case 3: RB_BWD(3); break;
        case 4: RB_BWD(4); break;
        case 5: RB_BWD(5); break;
        case 6: RB_BWD(6); break;
        case 7: RB_BWD(7); break;
        default: RB_BWD(0);
    }
#endif
#undef RB_BWD
    CGS_CHECK_LAUNCH(stream, cfg->debug);
    return CGS_OK;
}
