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
// Round 3: a row's Gaussians are a LIST of batch indices (rb_list_build: built by the row itself from the staging threads'
// 16-bit block masks, one DPP scan, no atomics) that the row walks with its own cursor - rows no longer wait for each
// other at 32-entry segment boundaries as with the bit masks of round 2 (-18..20 % wave iterations, tools/rb_iters.py);
// blocks are culled against the octagon (box + diagonals) of the alpha >= 1/255 ellipse (rb_block_mask).
// profiles/r03_blend_lists.txt has the measurements, including what the backward's time is made of now.
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
#ifndef RB_TILE_ORDER
#define RB_TILE_ORDER 1      // 0: tile = workgroup id (raster order)
#endif
#ifndef RB_XCD_RUN
#define RB_XCD_RUN 0         // > 0: runs of RB_XCD_RUN consecutive entries of the tile order share an XCD (rb_slot)
#endif


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

// ballot of a bool (HIP's __ballot takes an int: a v_cndmask + v_cmp_ne pair per call)
__device__ __forceinline__ uint64_t rb_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

template <int CTRL>
__device__ __forceinline__ float rb_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}

struct RbLane {
    int px, py, blk;
};

// Workgroup -> entry of the tile order.  Workgroups are dealt to the eight XCDs round robin (workgroup b -> XCD b % 8, each with
// its own L2) and neighbouring tiles read the same Gaussians' records: with entry = b, eight neighbours fetch them into eight
// L2s.  Inside every group of 8 R consecutive entries (started together: the longest-first order survives at that granularity)
// XCD x takes entries R x .. R x + R - 1.
__device__ __forceinline__ uint32_t rb_slot(uint32_t b, uint32_t n) {
#if RB_XCD_RUN > 0
    const uint32_t G = 8u * RB_XCD_RUN, q = b / G, r = b % G;
    return (q + 1u) * G <= n ? q * G + (r & 7u) * RB_XCD_RUN + (r >> 3) : b;
#else
    return b;
#endif
}

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

// Per-block entry lists of one staged batch.  Every staging thread owns one entry of the batch and publishes a 16-bit mask
// of the 4x4 blocks its octagon reaches (smask).  After the staging barrier every ROW builds the list of ITS OWN block: lane
// j of the row takes entries 16 j .. 16 j + 15 (two 16-byte LDS reads), extracts its block's bit from each mask, an
// exclusive scan over the row's sixteen counts (four DPP steps) gives it the slot of its first entry, and it stores the
// batch indices of its entries in ascending order.  The list is written and read by the same wave: no workgroup barrier, no
// atomics.  A row then walks its list with its own cursor — no bit scan, and no waiting for the other rows of the wave at
// 32-entry segment boundaries as the mask walk had (tools/rb_iters.py: 6.45 M -> 5.28 M backward wave iterations per view).
struct RbLists {
    uint8_t list[16][RB_THREADS];       // [block][k] = batch index of the block's k-th entry (ascending)
    uint16_t smask[RB_THREADS];         // block mask of every entry of the batch
};

template <int CTRL>
__device__ __forceinline__ uint32_t rb_dpp_u(uint32_t v) {      // bound_ctrl: lanes shifted in from outside the row read 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}

// returns the number of entries of the row's block; `lim` = highest admissible batch index for this block (>= 255: all)
__device__ __forceinline__ uint32_t rb_list_build(RbLists &S, int blk, int lane, int lim) {
    const int sub = lane & 15;
    const uint4 w0 = ((const uint4 *)S.smask)[sub * 2], w1 = ((const uint4 *)S.smask)[sub * 2 + 1];
    const uint32_t w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    uint32_t bits = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) bits |= ((w[k >> 1] >> (blk + 16 * (k & 1))) & 1u) << k;
    const int l = lim - 16 * sub;                                 // highest admissible k of this lane
    bits = l < 0 ? 0u : (l >= 15 ? bits : (bits & ((2u << l) - 1u)));
    const uint32_t cnt = (uint32_t)__builtin_popcount(bits);
    uint32_t inc = cnt;
    inc += rb_dpp_u<0x111>(inc);      // row_shr:1
    inc += rb_dpp_u<0x112>(inc);      // row_shr:2
    inc += rb_dpp_u<0x114>(inc);      // row_shr:4
    inc += rb_dpp_u<0x118>(inc);      // row_shr:8
    uint32_t slot = inc - cnt;
    while (bits) {
        const int k = __builtin_ctz(bits);
        bits &= bits - 1u;
        S.list[blk][slot++] = (uint8_t)(16 * sub + k);
    }
    const uint32_t total = (uint32_t)__shfl((int)inc, lane | 15, 64);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // the row's own stores are visible to its loads below
    __builtin_amdgcn_wave_barrier();
    return total;
}

__global__ void __launch_bounds__(RB_THREADS)
    blend_fwd_rows_kernel(int W, int H, int tiles_x, const uint2 *__restrict__ ranges,
                          const uint32_t *__restrict__ gid_sorted, const float4 *__restrict__ rec,
                          const float *__restrict__ bg, float *__restrict__ out_color, float *__restrict__ final_T,
                          uint32_t *__restrict__ n_contrib, uint32_t *__restrict__ tile_last,
                          const uint32_t *__restrict__ tile_order) {
    __shared__ float4 srec[RB_THREADS * 3];
    __shared__ RbLists S;
    __shared__ uint32_t wave_last[4];

    const int tile = RB_TILE_ORDER ? (int)tile_order[rb_slot(blockIdx.x, gridDim.x)] : (int)rb_slot(blockIdx.x, gridDim.x);        // longest lists first (tile_order_kernel)
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const RbLane L = rb_lane(tx, ty, wave, lane);
    const bool inside = L.px < W && L.py < H;
    float pxf = (float)L.px, pyf = (float)L.py;
    asm volatile("" : "+v"(pxf), "+v"(pyf));      // keep the converted coordinates in registers (the compiler re-converts per iteration)
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
        if (__syncthreads_count(done) == RB_THREADS) break;      // (also: the previous walk is over, LDS may be rewritten)
        uint32_t m16 = 0;
        if (pvalid) {
            srec[tid * 3] = p0;
            srec[tid * 3 + 1] = p1;
            srec[tid * 3 + 2] = p2;
            m16 = rb_block_mask(p0.x, p0.y, p2.y, p2.z, p2.w, tx * CGS_TILE, ty * CGS_TILE);
        } else {
            // a lane without an entry blends record `e` of a stale list slot with weight 0: the record must be finite
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            srec[tid * 3] = z; srec[tid * 3 + 1] = z; srec[tid * 3 + 2] = z;
        }
        {
            const uint32_t i = start + RB_THREADS + tid;
            pvalid = i < range.y;
            if (pvalid) {
                const uint32_t g = gid_sorted[i];
                p0 = rec[3 * (size_t)g]; p1 = rec[3 * (size_t)g + 1]; p2 = rec[3 * (size_t)g + 2];
            }
        }
        S.smask[tid] = (uint16_t)m16;
        __syncthreads();
        const uint32_t base_pos = start - range.x;
        if (rb_ballot(!done) != 0ull) {
            const uint32_t cnt = rb_list_build(S, L.blk, lane, RB_THREADS);
            uint32_t i = 0, lastb = 0;             // lastb: batch index + 1 of the pixel's last contribution in this batch
            uint32_t e_next = S.list[L.blk][0];
            // every lane of a row advances i together (also the lanes whose pixel is finished); the wave leaves when no
            // unfinished pixel has entries left
            while (rb_ballot(!done && i < cnt) != 0ull) {
                const bool has = i < cnt;
                const uint32_t e = e_next;
                i += has ? 1u : 0u;
                e_next = S.list[L.blk][i & (RB_THREADS - 1)];       // next entry's index: in flight during this iteration
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
                lastb = upd ? e + 1u : lastb;
                done = done || stop;
            }
            last = lastb ? base_pos + lastb : last;
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

__global__ void __launch_bounds__(RB_THREADS)
    blend_bwd_rows_kernel(int W, int H, int tiles_x, const uint2 *__restrict__ ranges,
                          const uint32_t *__restrict__ gid_sorted, const float4 *__restrict__ rec,
                          const float *__restrict__ bg, const float *__restrict__ final_T,
                          const uint32_t *__restrict__ n_contrib, const uint32_t *__restrict__ tile_last,
                          const float *__restrict__ dL_dout, float *__restrict__ dL_dmean2D_px,
                          float *__restrict__ dL_dconic, float *__restrict__ dL_dopacity,
                          float *__restrict__ dL_dcolors, const uint32_t *__restrict__ tile_order) {
    // 22.6 KB of LDS per workgroup = seven workgroups per CU: two float4 per record plus its blue component (the third
    // float4 only carries cull extents the staging thread has in registers), no copy of the Gaussian ids (the flush reads
    // gid_sorted again)
    __shared__ float4 srec[RB_THREADS * 2];
    __shared__ float sblue[RB_THREADS];
    __shared__ float sacc[RB_THREADS][RB_NGRAD];
    __shared__ RbLists S;

    const int tile = RB_TILE_ORDER ? (int)tile_order[rb_slot(blockIdx.x, gridDim.x)] : (int)rb_slot(blockIdx.x, gridDim.x);        // longest lists first (tile_order_kernel)
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
            srec[tid * 2] = p0;
            srec[tid * 2 + 1] = p1;
            sblue[tid] = p2.x;
            m16 = rb_block_mask(p0.x, p0.y, p2.y, p2.z, p2.w, tx * CGS_TILE, ty * CGS_TILE);
        } else {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            srec[tid * 2] = z; srec[tid * 2 + 1] = z; sblue[tid] = 0.f;
        }
        if (bi > 0) {      // every position of an earlier batch is < tlast
            pg = gid_sorted[range.x + pos - RB_THREADS];
            p0 = rec[3 * (size_t)pg]; p1 = rec[3 * (size_t)pg + 1]; p2 = rec[3 * (size_t)pg + 2];
        }
#pragma unroll
        for (int k = 0; k < RB_NGRAD; ++k) sacc[tid][k] = 0.f;
        S.smask[tid] = (uint16_t)m16;
        __syncthreads();

        {
            // entries behind the LAST contribution of every pixel of this 4x4 block (n_contrib: where the forward stopped)
            // cannot contribute to it: they never enter the block's list (the tile-wide bound `tlast` is the maximum over
            // 256 pixels, a block's own bound over 16)
            int i = (int)rb_list_build(S, L.blk, lane, (int)blk_last - (int)base_pos - 1) - 1;
            uint32_t e_next = S.list[L.blk][max(i, 0)];
            while (rb_ballot(i >= 0) != 0ull) {
                const bool has = i >= 0;
                const uint32_t e = e_next;
                i -= has ? 1 : 0;
                e_next = S.list[L.blk][max(i, 0)];                   // next entry's index: in flight during this iteration
                const uint32_t position = base_pos + e + 1u;         // 1-based
                const float4 r0 = srec[e * 2], r1 = srec[e * 2 + 1];
                const float blue = sblue[e];
                const RbEval ev = rb_eval(r0, r1, pxf, pyf);
                const bool act = has && (position <= my_last) && ev.hit;
                if (rb_ballot(act) == 0ull) continue;
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
                // (red != 0: a row whose 16 pixels took nothing from its entry — the wave goes on while ANY row has a contribution —
                //  would add nine zeros through the LDS float-atomic unit, the kernel's second bound: -6 %, same sums bit for bit;
                //  profiles/r05_blend_bwd_ablations.txt)
                if (has && sub < RB_NGRAD && red != 0.f) atomicAdd(&sacc[e][sub], red);
            }
        }
        __syncthreads();
        if (pos < tlast) {
            const float a0 = sacc[tid][0], a1 = sacc[tid][1], a2 = sacc[tid][2], a3 = sacc[tid][3],
                        a4 = sacc[tid][4], a5 = sacc[tid][5], a6 = sacc[tid][6], a7 = sacc[tid][7],
                        a8 = sacc[tid][8];
            if (a0 != 0.f || a1 != 0.f || a2 != 0.f || a3 != 0.f || a4 != 0.f || a5 != 0.f || a6 != 0.f ||
                a7 != 0.f || a8 != 0.f) {
                const uint32_t g = gid_sorted[range.x + pos];
                const float4 q0 = srec[tid * 2], q1 = srec[tid * 2 + 1];
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



// Workgroup -> tile map of the blend kernels: tiles in descending order of their list length CLASS (octaves), raster order
// inside a class.  A view's tiles differ by two orders of magnitude in work and the hardware starts workgroups in index order:
// with tile = workgroup id the last wave of workgroups holds whatever tiles the raster order put last, and the kernel ends when
// the longest of them does; longest first, the tail is made of the shortest lists (LPT scheduling): blend forward 452 -> 401 us,
// backward 958 -> 901 us on the headline scene (same box, profiles/r06_tile_order.txt).  Raster order inside a class keeps
// neighbouring tiles — which read the same Gaussians' records — running together: with an arbitrary order inside
// quarter-octave classes the heavy-pair scene LOST 2 % of its step.  One workgroup: a stable counting sort, thread t owns
// a contiguous run of tiles and one counter per class ([class][thread] in LDS, scanned in that order).  Deterministic.
#define TO_CLASSES 16
#define TO_THREADS 1024
#ifndef TO_DENSE
#define TO_DENSE 4096u      // mean list length above which a view keeps the raster order
#endif
#ifndef TO_TOP
#define TO_TOP 13            // lists of >= 2^(TO_TOP - 1) = 4096 entries share the top class (see below)
#endif
// 0: empty; class c >= 1: [2^(c-1), 2^c) entries; everything from 4096 entries on is ONE class: such lists saturate their pixels
// long before their end, their length says nothing about their work, and sorting them by it only takes neighbouring tiles apart
// (the heavy-pair scene — every list 16 k entries, a few hundred walked — lost 12-18 % of its blend kernels to that).
__device__ __forceinline__ int to_class(uint32_t len) {
    if (len == 0) return 0;
    const int c = 32 - __clz(len);
    return c < TO_TOP ? c : TO_TOP;
}
__global__ void __launch_bounds__(TO_THREADS) tile_order_kernel(int nt, const uint2 *__restrict__ ranges, uint32_t *__restrict__ order) {
    __shared__ uint32_t cnt[TO_CLASSES][TO_THREADS];       // 64 KB
    __shared__ uint32_t wsum[TO_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (nt + TO_THREADS - 1) / TO_THREADS;
    const int t0 = tid * per, t1 = min(nt, t0 + per);
#pragma unroll
    for (int c = 0; c < TO_CLASSES; ++c) cnt[c][tid] = 0;
    uint32_t total = 0;
    for (int t = t0; t < t1; ++t) {
        const uint2 r = ranges[t];
        const uint32_t len = r.y > r.x ? r.y - r.x : 0u;
        total += len >> 4;                                        // (in units of 16 entries: no overflow at 2^32 pairs)
        cnt[to_class(len)][tid] += 1;
    }
    // A view of LONG lists (mean above TO_DENSE entries per tile: the heavy-pair scene has 16 k) keeps the raster order: its
    // tiles saturate after a few hundred entries whatever their length, there is no tail to cut, and running the densest tiles
    // together costs the blend kernels 12-18 % (they stream their long lists through the same L2s at the same time, where the
    // raster order mixes them with light tiles) — measured on both scenes, profiles/r06_tile_order.txt.
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) total += __shfl_xor(total, o, 64);
    if (lane == 0) wsum[wave] = total;
    __syncthreads();
    total = 0;
    for (int w = 0; w < TO_THREADS / 64; ++w) total += wsum[w];
    __syncthreads();
    if ((uint64_t)total * 16u > (uint64_t)nt * TO_DENSE) {
        for (int t = t0; t < t1; ++t) order[t] = (uint32_t)t;
        return;
    }
    // exclusive scan of the flat array in (class DESCENDING, thread ascending) order: thread j owns flat entries 16 j .. 16 j + 15
    uint32_t v[TO_CLASSES], mine = 0;
#pragma unroll
    for (int k = 0; k < TO_CLASSES; ++k) {
        const int f = tid * TO_CLASSES + k;                    // flat index: class = 15 - f / 1024, thread = f % 1024
        v[k] = cnt[TO_CLASSES - 1 - f / TO_THREADS][f % TO_THREADS];
        mine += v[k];
    }
    uint32_t inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t run = inc - mine;
    for (int w = 0; w < wave; ++w) run += wsum[w];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TO_CLASSES; ++k) {
        const int f = tid * TO_CLASSES + k;
        cnt[TO_CLASSES - 1 - f / TO_THREADS][f % TO_THREADS] = run;
        run += v[k];
    }
    __syncthreads();
    for (int t = t0; t < t1; ++t) {
        const uint2 r = ranges[t];
        const int c = to_class(r.y > r.x ? r.y - r.x : 0u);
        order[cnt[c][tid]++] = (uint32_t)t;
    }
}

int cgs_launch_tile_order(const cgs_raster_cfg *cfg, CgsImg &im, hipStream_t stream) {
    const int nt = cgs_tiles_x(cfg) * cgs_tiles_y(cfg);
    hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(TO_THREADS), 0, stream, nt, (const uint2 *)im.ranges, im.tile_order);
    CGS_CHECK_LAUNCH(stream, cfg->debug);
    return CGS_OK;
}

int cgs_launch_blend_fwd_rows(const cgs_raster_cfg *cfg, CgsGeom &g, CgsBin &b, CgsImg &im, float *out_color,
                              hipStream_t stream) {
    const int tx = cgs_tiles_x(cfg), ty = cgs_tiles_y(cfg);
    hipLaunchKernelGGL(blend_fwd_rows_kernel, dim3((unsigned)(tx * ty)), dim3(RB_THREADS), 0, stream, cfg->image_width,
                       cfg->image_height, tx, (const uint2 *)im.ranges, (const uint32_t *)b.gid_sorted,
                       (const float4 *)g.rec, cfg->bg, out_color, im.final_T, im.n_contrib, im.tile_last,
                       (const uint32_t *)im.tile_order);
    CGS_CHECK_LAUNCH(stream, cfg->debug);
    return CGS_OK;
}

int cgs_launch_blend_bwd_rows(const cgs_raster_cfg *cfg, CgsGeom &g, CgsBin &b, CgsImg &im, const float *dL_dout,
                              float *dL_dmean2D_px, float *dL_dconic, float *dL_dopacity, float *dL_dcolors,
                              hipStream_t stream) {
    const int tx = cgs_tiles_x(cfg), ty = cgs_tiles_y(cfg);
    const float4 *rec = (const float4 *)g.rec;
    hipLaunchKernelGGL(blend_bwd_rows_kernel, dim3((unsigned)(tx * ty)), dim3(RB_THREADS), 0, stream, cfg->image_width,
                       cfg->image_height, tx, (const uint2 *)im.ranges, (const uint32_t *)b.gid_sorted, rec,
                       cfg->bg, (const float *)im.final_T, (const uint32_t *)im.n_contrib, (const uint32_t *)im.tile_last,
                       dL_dout, dL_dmean2D_px, dL_dconic, dL_dopacity, dL_dcolors, (const uint32_t *)im.tile_order);
    CGS_CHECK_LAUNCH(stream, cfg->debug);
    return CGS_OK;
}
