// Tile binning (R4 + R5 of SURVEY section 2.1) without a materialised unsorted pair list.
//
// The reference duplicates every Gaussian into (tile | depth, id) keys and radix-sorts 64-bit keys
// (diff-gaussian-rasterization's duplicateWithKeys + cub::DeviceRadixSort::SortPairs).  Here the Gaussians are
// already in depth order (raster_geom / prims: stable 32-bit sort of P keys), so the per-tile lists are a STABLE
// sort of the (tile, id) pairs by tile alone, and the unsorted pair stream is a pure function of three per-Gaussian
// arrays in depth order (tile rectangle, exclusive pair offset, id).  The first radix pass therefore never reads
// pairs: its histogram and scatter kernels GENERATE the 4096 pairs of their block from those arrays
// (block -> first Gaussian by a binary search done once per block in block_first_kernel; pair -> Gaussian through
// start markers in LDS and a max-scan; pair -> tile by one division), and write them already partitioned by the low
// digit as (16-bit tile key, 32-bit id).  The second pass is an ordinary pass over 6-byte pairs.
//
// The last pass writes ids only: the end of every tile's list is an atomicMax of the positions of the key runs a block
// sees in its staged (sorted) items, the starts are an exclusive prefix maximum over the tiles (ranges_fix_kernel).
//
//   bytes per pair:  pass A writes 6; pass B reads 2 (histogram) + 6 and writes 4     = 18
//   before        :  emit 8; two passes of 4 + 8 + 8; tile ranges 4                   = 52
//
// Tile keys are 16 bits: taken for grids of at most 65536 tiles (4096 x 4096 pixels and beyond); larger grids keep the
// round-1 path (emit_pairs + cgs_sort_pairs_u32).  Ranking inside a block is the wave64 ballot match of prims.hip.
#include "cgs_internal.h"
#include "buf_access.h"

#define TB_THREADS 256
#define TB_ITEMS 16
#define TB_TILE (TB_THREADS * TB_ITEMS)
#define TB_WAVES (TB_THREADS / CGS_WAVE)
#define TB_MAXR 256
#ifndef TB_XCD_MAP
#define TB_XCD_MAP 0      // 1: columns dealt to the XCDs in contiguous eighths (cgs_xcd_item) — what wins 10 % in the depth sort LOSES
                          // 9-10 us per pass here (same-box A/B, profiles/r06_depth_sort.txt); 0: column = workgroup id
#endif

int cgs_scan_exclusive_u32_total(const uint32_t *in, uint32_t *out, int64_t n, void *scratch, size_t scratch_bytes,
                                 uint32_t *grand_total, hipStream_t stream);

namespace {

// rect_lo / rect_hi[i] = packed tile rectangle of the i-th Gaussian in depth order, cnt[i] = its tile count.
// GR_ITEMS dependent (order -> rectangle) chains per thread, requested together from clamped indices.  Same-box A/B at 5.8 M
// Gaussians: 1 / 2 / 4 / 8 chains = 90 / 92 / 92 / 83 us — the kernel is NOT latency-bound: 5.8 M gathers of 8 bytes in depth order
// fetch a whole line each (~0.75 GB through the fabric for 46 MB of rectangles), which is what its time is
#ifndef GR_ITEMS
#define GR_ITEMS 2
#endif
#ifndef GR_XCD
#define GR_XCD 1
#endif
__global__ void __launch_bounds__(256)
    gather_rects_kernel(int64_t P, const uint32_t *__restrict__ order, const uint2 *__restrict__ rect,
                        uint32_t *__restrict__ rect_lo, uint32_t *__restrict__ rect_hi, uint32_t *__restrict__ cnt) {
#if GR_XCD
    const int64_t blk = cgs_xcd_item((P + 256 * GR_ITEMS - 1) / (256 * GR_ITEMS));
    if (blk < 0) return;
#else
    const int64_t blk = blockIdx.x;
#endif
    const int64_t i0 = blk * (256 * GR_ITEMS) + threadIdx.x;
    uint32_t o[GR_ITEMS];
    uint2 rc[GR_ITEMS];
#pragma unroll
    for (int k = 0; k < GR_ITEMS; ++k) o[k] = order[min(i0 + 256 * k, P - 1)];
#pragma unroll
    for (int k = 0; k < GR_ITEMS; ++k) rc[k] = rect[o[k]];
#pragma unroll
    for (int k = 0; k < GR_ITEMS; ++k) {
        const int64_t i = i0 + 256 * k;
        if (i < P) {
            rect_lo[i] = rc[k].x;
            rect_hi[i] = rc[k].y;
            const uint32_t w = (rc[k].y & 0xFFFFu) - (rc[k].x & 0xFFFFu), h = (rc[k].y >> 16) - (rc[k].x >> 16);
            cnt[i] = w * h;
        }
    }
}

// bf[b] = index (depth order) of the Gaussian that owns pair b * TB_TILE, b < nb;  bf[nb] = P - 1.  Also clears the tile
// ranges (the last pass accumulates into them).
__global__ void __launch_bounds__(256)
    block_first_kernel(int64_t nb, int64_t P, const uint32_t *__restrict__ offsets, uint32_t *__restrict__ bf, int nt,
                       uint2 *__restrict__ ranges, const uint32_t *__restrict__ R_dev, uint32_t R_cap) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b < nt) ranges[b] = make_uint2(0u, 0u);
    if (R_dev) nb = ((int64_t)min(*R_dev, R_cap) + TB_TILE - 1) / TB_TILE;      // speculative launch: the pair count is on the device
    if (b > nb) return;
    if (b == nb) { bf[b] = (uint32_t)(P - 1); return; }
    const uint32_t target = (uint32_t)(b * TB_TILE);
    int64_t lo = 0, hi = P;                 // first index whose offset is > target
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (offsets[mid] <= target) lo = mid + 1; else hi = mid;
    }
    bf[b] = (uint32_t)(lo - 1);             // offsets[0] = 0 <= target: lo >= 1
}

// Per block: sidx[p] = (index of the Gaussian that owns pair base + p) - i_lo, for p < TB_TILE.
// Gaussians with at least one tile whose first pair lies inside the block put their relative index at that pair's
// slot; an inclusive max-scan hands it on to the pairs behind it (slot 0 belongs to i_lo itself = relative 0).
__device__ __forceinline__ void build_owner_index(uint32_t *sidx, uint32_t *swave /*[TB_WAVES]*/, uint32_t base, uint32_t i_lo,
                                                  uint32_t i_hi, int64_t P, uint32_t R,
                                                  const uint32_t *__restrict__ offsets) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < TB_ITEMS; ++k) sidx[k * TB_THREADS + tid] = 0u;
    __syncthreads();
    for (uint32_t i = i_lo + 1u + (uint32_t)tid; i <= i_hi; i += TB_THREADS) {
        const uint32_t o = offsets[i];
        const uint32_t on = (int64_t)i + 1 < P ? offsets[i + 1] : R;
        if (on > o && o > base && o - base < (uint32_t)TB_TILE) sidx[o - base] = i - i_lo;
    }
    __syncthreads();
    uint32_t v[TB_ITEMS];
    uint32_t run = 0;
#pragma unroll
    for (int k = 0; k < TB_ITEMS; ++k) {
        v[k] = sidx[tid * TB_ITEMS + k];
        run = max(run, v[k]);
    }
    uint32_t inc = run;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64);
        if (lane >= d) inc = max(inc, o);
    }
    if (lane == 63) swave[wave] = inc;
    __syncthreads();
    uint32_t pre = (uint32_t)__shfl_up((int)inc, 1, 64);
    if (lane == 0) pre = 0;
#pragma unroll
    for (int w = 0; w < TB_WAVES; ++w) pre = (w < wave) ? max(pre, swave[w]) : pre;
    run = pre;
#pragma unroll
    for (int k = 0; k < TB_ITEMS; ++k) {
        run = max(run, v[k]);
        sidx[tid * TB_ITEMS + k] = run;
    }
    __syncthreads();
}

struct TbPair { uint32_t key, i, mask; };

// Buckets of the two-level binning (second half of this file): BK_W x BK_H tiles, one bit of a 32-bit mask per tile
#define BK_W 8
#define BK_H 4
#define BK_TILES (BK_W * BK_H)

// bucket rectangle of a tile rectangle (exclusive maxima); an empty tile rectangle covers no bucket
__device__ __forceinline__ void bucket_rect(uint32_t lo, uint32_t hi, uint32_t &cx0, uint32_t &cy0, uint32_t &cw, uint32_t &ch) {
    const uint32_t x0 = lo & 0xFFFFu, y0 = lo >> 16, x1 = hi & 0xFFFFu, y1 = hi >> 16;
    const bool some = x1 > x0 && y1 > y0;
    cx0 = x0 / BK_W;
    cy0 = y0 / BK_H;
    cw = some ? (x1 + BK_W - 1) / BK_W - cx0 : 0u;
    ch = some ? (y1 + BK_H - 1) / BK_H - cy0 : 0u;
}

// pair j (global index) -> (tile key, index of its Gaussian in depth order).  COARSE: pairs are (Gaussian, bucket), `offsets`
// the scan of the bucket counts, tiles_x the number of buckets per row, and .mask holds the bucket's tiles the Gaussian covers
template <bool COARSE>
__device__ __forceinline__ TbPair gen_pair(uint32_t j, uint32_t base, uint32_t i_lo, const uint32_t *sidx,
                                           const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ rect_lo,
                                           const uint32_t *__restrict__ rect_hi, uint32_t tiles_x) {
    TbPair p;
    p.i = i_lo + sidx[j - base];
    p.mask = 0u;
    const uint32_t local = j - offsets[p.i];
    const uint32_t lo = rect_lo[p.i], hi = rect_hi[p.i];
    uint32_t x0 = lo & 0xFFFFu, y0 = lo >> 16, w = (hi & 0xFFFFu) - x0, h = 0;
    if (COARSE) bucket_rect(lo, hi, x0, y0, w, h);
    // local / w for local < 2^24, w < 2^16: float quotient, one correction step either way
    uint32_t q = (uint32_t)((float)local * __builtin_amdgcn_rcpf((float)w));
    int32_t r = (int32_t)(local - q * w);
    if (r < 0) { --q; r += (int32_t)w; }
    if (r >= (int32_t)w) { ++q; r -= (int32_t)w; }
    p.key = (y0 + q) * tiles_x + x0 + (uint32_t)r;
    if (COARSE) {
        const int32_t tx0 = (int32_t)((x0 + (uint32_t)r) * BK_W), ty0 = (int32_t)((y0 + q) * BK_H);
        const int32_t c0 = max((int32_t)(lo & 0xFFFFu) - tx0, 0), c1 = min((int32_t)(hi & 0xFFFFu) - tx0, BK_W);
        const int32_t r0 = max((int32_t)(lo >> 16) - ty0, 0), r1 = min((int32_t)(hi >> 16) - ty0, BK_H);
        const uint32_t cols = ((1u << c1) - 1u) & ~((1u << c0) - 1u), rows = ((1u << r1) - 1u) & ~((1u << r0) - 1u);
        p.mask = cols * ((rows & 1u) | ((rows & 2u) << 7) | ((rows & 4u) << 14) | ((rows & 8u) << 21));
    }
    return p;
}

template <bool GEN, bool COARSE = false>
__global__ void __launch_bounds__(TB_THREADS)
    tb_hist_kernel(const uint16_t *__restrict__ keys, const uint32_t *__restrict__ offsets,
                   const uint32_t *__restrict__ rect_lo, const uint32_t *__restrict__ rect_hi,
                   const uint32_t *__restrict__ bf, int64_t P, uint32_t R, uint32_t tiles_x,
                   uint32_t *__restrict__ hist /*[rows][ncol]*/, int shift, int nbits, const uint32_t *__restrict__ R_dev,
                   uint32_t ncol, int xcd) {
    // column of this workgroup: its id, or (xcd) the XCD-aware assignment of cgs_xcd_item() — neighbouring columns, whose runs are
    // adjacent in the output, on the same XCD's L2 (prims.hip: sort_tile_of_block)
    const int64_t col_ = xcd ? cgs_xcd_item((int64_t)ncol) : (int64_t)blockIdx.x;
    if (col_ < 0) return;
    const uint32_t col = (uint32_t)col_;
    if (R_dev) R = min(*R_dev, R);         // speculative launch: R (argument) is the capacity, the count is on the device
    __shared__ uint32_t h[TB_MAXR];
    __shared__ uint32_t sidx[GEN ? TB_TILE : 1];
    __shared__ uint32_t swave[TB_WAVES];
    const int tid = threadIdx.x;
    const uint32_t mask = (1u << nbits) - 1u;
    h[tid] = 0;
    // A block owns a COLUMN of the histogram = `per` consecutive blocks of TB_TILE pairs (per = 1 when the grid has a block per
    // TB_TILE pairs: the fine passes; the bucket pass runs a fixed grid because its pair count is known on the device only).
    const uint32_t nblk = (R + (uint32_t)TB_TILE - 1u) / (uint32_t)TB_TILE, per = max(1u, (nblk + ncol - 1u) / ncol);
    const uint32_t b0 = col * per, b1 = min(nblk, b0 + per);
    __syncthreads();
    for (uint32_t blk = b0; blk < b1; ++blk) {
        const uint32_t base = blk * (uint32_t)TB_TILE;
        uint32_t i_lo = 0;
        if (GEN) {
            i_lo = bf[blk];
            build_owner_index(sidx, swave, base, i_lo, bf[blk + 1], P, R, offsets);
        }
#pragma unroll 4
        for (int k = 0; k < TB_ITEMS; ++k) {
            const uint32_t j = base + (uint32_t)(k * TB_THREADS + tid);
            if (j < R) {
                const uint32_t key = GEN ? gen_pair<COARSE>(j, base, i_lo, sidx, offsets, rect_lo, rect_hi, tiles_x).key
                                         : (uint32_t)keys[j];
                atomicAdd(&h[(key >> shift) & mask], 1u);
            }
        }
        __syncthreads();                    // (the owner index is rebuilt by the next block of the column)
    }
    if (tid < (1 << nbits)) hist[(int64_t)tid * ncol + col] = h[tid];     // (a column past the pairs writes zeros)
}

// One stable radix pass.  GEN: the block's pairs come from gen_pair (first pass); otherwise from (keys_in, vals_in).
// COARSE (with GEN and FINAL): (Gaussian, bucket) pairs; mask_out receives every pair's tile mask next to its id.
template <bool GEN, bool FINAL, int NBITS, bool COARSE = false>
__global__ void __launch_bounds__(TB_THREADS)
    tb_scatter_kernel(const uint16_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                      const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ rect_lo,
                      const uint32_t *__restrict__ rect_hi, const uint32_t *__restrict__ order,
                      const uint32_t *__restrict__ bf, int64_t P, uint32_t R, uint32_t tiles_x,
                      uint16_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, uint2 *__restrict__ ranges,
                      const uint32_t *__restrict__ hist_scanned, int shift, const uint32_t *__restrict__ R_dev,
                      uint32_t ncol, int xcd, const uint32_t *__restrict__ digit_totals, uint32_t *__restrict__ mask_out = nullptr) {
    const int64_t col_ = xcd ? cgs_xcd_item((int64_t)ncol) : (int64_t)blockIdx.x;      // (see tb_hist_kernel)
    if (col_ < 0) return;
    const uint32_t col = (uint32_t)col_;
    if (R_dev) R = min(*R_dev, R);
    constexpr int nbits = NBITS;
    const uint32_t nblk = (R + (uint32_t)TB_TILE - 1u) / (uint32_t)TB_TILE, per = max(1u, (nblk + ncol - 1u) / ncol);
    const uint32_t b0 = col * per, b1 = min(nblk, b0 + per);   // this block's column (see tb_hist_kernel)
    if (b0 >= b1) return;
    __shared__ uint32_t wcnt[TB_WAVES][TB_MAXR];
    __shared__ uint32_t lstart[TB_MAXR], gbase[TB_MAXR];
    __shared__ uint32_t wsum[TB_WAVES];
    __shared__ uint32_t stage[TB_TILE + TB_TILE / 2];      // sval[TB_TILE] | skey (u16)[TB_TILE]; the owner index before
    uint32_t *sval = stage;
    uint16_t *skey = (uint16_t *)(stage + TB_TILE);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t mask = (1u << nbits) - 1u;
    // thread d: where the column's next item of digit d goes (advanced by every block of the column)
    uint32_t gcur = tid < (1 << nbits) ? hist_scanned[(int64_t)tid * ncol + col] : 0u;
    if (digit_totals) {
        // hist_scanned holds per-digit exclusive scans over the columns (cgs_launch_digit_scan: one launch instead of the generic
        // scan's two): the digit's first global position = the totals of the digits below it, summed here
        const uint32_t gt = tid < (1 << nbits) ? digit_totals[tid] : 0u;
        uint32_t ginc = gt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(ginc, o, 64);
            if (lane >= o) ginc += up;
        }
        if (lane == 63) wsum[wave] = ginc;
        __syncthreads();
        uint32_t goff = 0;
#pragma unroll
        for (int w = 0; w < TB_WAVES; ++w) goff += (w < wave) ? wsum[w] : 0u;
        gcur += goff + ginc - gt;
    }
    for (uint32_t blk = b0; blk < b1; ++blk) {
    __syncthreads();                                        // the previous block's staging is drained
#pragma unroll
    for (int w = 0; w < TB_WAVES; ++w) wcnt[w][tid] = 0;
    const uint32_t base = blk * (uint32_t)TB_TILE;
    uint32_t i_lo = 0;
    if (GEN) {
        i_lo = bf[blk];
        build_owner_index(stage, wsum, base, i_lo, bf[blk + 1], P, R, offsets);
    } else {
        __syncthreads();
    }
    const uint32_t wbase = base + (uint32_t)wave * (TB_TILE / TB_WAVES);
    uint32_t key[TB_ITEMS], val[TB_ITEMS], rank[TB_ITEMS], tmask[COARSE ? TB_ITEMS : 1];
    volatile uint32_t *my = wcnt[wave];
    const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < TB_ITEMS; ++r) {
        const uint32_t j = wbase + (uint32_t)(r * 64 + lane);
        const bool valid = j < R;
        key[r] = 0xFFFFu;
        val[r] = 0u;
        if (valid) {
            if (GEN) {
                const TbPair p = gen_pair<COARSE>(j, base, i_lo, stage, offsets, rect_lo, rect_hi, tiles_x);
                key[r] = p.key;
                val[r] = order[p.i];
                if (COARSE) tmask[r] = p.mask;
            } else {
                key[r] = keys_in[j];
                val[r] = vals_in[j];
            }
        }
        const uint32_t d = (key[r] >> shift) & mask;
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < nbits; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        uint32_t old = 0;
        rank[r] = 0;
        if (valid) {
            const int leader = __builtin_ctzll(peers);
            if (lane == leader) {
                old = my[d];
                my[d] = old + (uint32_t)__builtin_popcountll(peers);
            }
            old = __shfl(old, leader, 64);
            rank[r] = old + (uint32_t)__builtin_popcountll(peers & lt_mask);
        }
    }
    __syncthreads();      // ranks done; the owner index in `stage` is dead from here on
    {
        const int d = tid;
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < TB_WAVES; ++w) tot += wcnt[w][d];
        uint32_t inc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(inc, o, 64);
            if (lane >= o) inc += up;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t woff = 0;
#pragma unroll
        for (int w = 0; w < TB_WAVES; ++w) woff += (w < wave) ? wsum[w] : 0u;
        uint32_t run = woff + inc - tot;
        lstart[d] = run;
        // first global position of this block's items of the digit MINUS their first staging slot: position = gbase + slot
        gbase[d] = gcur - run;
        gcur += tot;
#pragma unroll
        for (int w = 0; w < TB_WAVES; ++w) {
            const uint32_t c = wcnt[w][d];
            wcnt[w][d] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < TB_ITEMS; ++r) {
        const uint32_t j = wbase + (uint32_t)(r * 64 + lane);
        if (j < R) {
            const uint32_t d = (key[r] >> shift) & mask;
            const uint32_t slot = wcnt[wave][d] + rank[r];
            skey[slot] = (uint16_t)key[r];
            sval[slot] = val[r];
            // (the masks go out from the registers: staging them too costs 16 KB of LDS = two workgroups per CU less)
            if (COARSE) mask_out[gbase[d] + slot] = tmask[r];
        }
    }
    __syncthreads();
    const int count = (int)min((uint32_t)TB_TILE, R - base);
    for (int j = tid; j < count; j += TB_THREADS) {
        const uint32_t k = skey[j];
        const uint32_t d = (k >> shift) & mask;
        const uint32_t pos = gbase[d] + (uint32_t)j;
        vals_out[pos] = sval[j];
        if (FINAL && !COARSE) {           // (the bucket pass is ONE pass: its runs start where the scanned histogram says)
            // equal keys are adjacent in the staged order (same digit; inside a digit the arrival order is the order
            // the previous pass left, ascending in the low digit): the last item of a run bounds the tile's list
            if (j + 1 == count || (uint32_t)skey[j + 1] != k) atomicMax(&ranges[k].y, pos + 1u);
        } else if (!FINAL) {
            keys_out[pos] = (uint16_t)k;
        }
    }
    }
}

// ranges[t].x = end of the closest non-empty tile before t (exclusive prefix maximum of .y); empty tiles stay (0, 0)
__global__ void __launch_bounds__(1024) ranges_fix_kernel(int nt, uint2 *__restrict__ ranges) {
    __shared__ uint32_t wmax[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (nt + 1023) / 1024;
    const int t0 = tid * per, t1 = min(nt, t0 + per);
    uint32_t run = 0;
    uint32_t y8[8];                       // up to 8192 tiles (1080p: 8100): one round of independent loads, kept in registers
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        y8[k] = (per <= 8 && t0 + k < t1) ? ranges[t0 + k].y : 0u;
        run = max(run, y8[k]);
    }
    if (per > 8)
        for (int t = t0; t < t1; ++t) run = max(run, ranges[t].y);
    uint32_t inc = run;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64);
        if (lane >= d) inc = max(inc, o);
    }
    if (lane == 63) wmax[wave] = inc;
    __syncthreads();
    uint32_t pre = (uint32_t)__shfl_up((int)inc, 1, 64);
    if (lane == 0) pre = 0;
    for (int w = 0; w < wave; ++w) pre = max(pre, wmax[w]);
    if (per <= 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (y8[k]) { ranges[t0 + k].x = pre; pre = y8[k]; }
    } else {
        for (int t = t0; t < t1; ++t) {
            const uint32_t y = ranges[t].y;
            if (y) { ranges[t].x = pre; pre = y; }
        }
    }
}


// ---- two-level binning (round 5): buckets of BK_W x BK_H tiles, then tile lists inside a bucket -----------------------------
//
// A view whose Gaussians cover MANY tiles each (136 M pairs at 5.8 M Gaussians: bench.py's heavy-pair scene) makes the radix
// passes above move 18 bytes per (tile, Gaussian) pair and rank every pair by a wave-wide digit match.  The per-tile lists are
// a stable partition of the depth-ordered Gaussians, so they can be cut in two steps that touch the fine pairs ONCE, as a
// 4-byte store:
//   1. one stable radix pass over (Gaussian, BUCKET) pairs (a bucket = 8 x 4 tiles; <= 256 buckets = one 8-bit digit; the pass
//      is the pair-generating kernel above with COARSE), which leaves per bucket the ids in depth order and, next to every id,
//      the 32-bit mask of the bucket's tiles that Gaussian covers;
//   2. bucket lists cut into chunks of BK_CHUNK entries: a counting kernel (per chunk and tile: popcounts of mask-bit ballots),
//      ONE exclusive scan over the counts laid out [tile][chunk of the tile's bucket] — which is both the start of every tile's
//      list (tiles in ascending order: the same gid_sorted, entry for entry, as the radix path) and every chunk's write
//      position in it — and a fill kernel that re-reads (id, mask), ranks by ballot + mbcnt (depth order is lane order), stages
//      a round's entries per tile in LDS and writes them as runs.
//   bytes per fine pair: 4 (the store);  per bucket pair (6.5 x fewer in the heavy scene): 8 written, 4 + 8 read.
#define BK_CHUNK 1024
#define BK_THREADS 256
#define BK_WAVES (BK_THREADS / CGS_WAVE)
#define BK_MAXB 256
// bk_tab words: bucket ranges (uint2[256]) | chunk prefix [257] | number of chunks | first count slot of every tile [8192]
#define BK_TAB_CR 0
#define BK_TAB_CS 512
#define BK_TAB_NCH 770
#define BK_TAB_TS 1024
#define BK_TAB_WORDS (BK_TAB_TS + BK_MAXB * BK_TILES)

__global__ void __launch_bounds__(256)
    bk_coarse_count_kernel(int64_t P, const uint32_t *__restrict__ rect_lo, const uint32_t *__restrict__ rect_hi,
                           uint32_t *__restrict__ ccnt) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    uint32_t cx0, cy0, cw, ch;
    bucket_rect(rect_lo[i], rect_hi[i], cx0, cy0, cw, ch);
    ccnt[i] = cw * ch;
}

__device__ __forceinline__ uint32_t bk_wave_excl_sum(uint32_t v, int lane, uint32_t &total) {
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64);
        if (lane >= d) inc += o;
    }
    total = (uint32_t)__shfl((int)inc, 63, 64);
    return inc - v;
}

// one workgroup: bucket ranges (the scanned histogram of the bucket pass: row d, column 0 = where bucket d's run starts), chunk
// table, first count slot of every tile, tile ranges cleared
__global__ void __launch_bounds__(256)
    bk_table_kernel(uint32_t *__restrict__ tab, int nbk, uint32_t bx, uint32_t tiles_x, int nt, uint2 *__restrict__ ranges,
                    const uint32_t *__restrict__ hist_scanned, int64_t ncol, int nrow, const uint32_t *__restrict__ Rc_dev,
                    uint32_t cap) {
    __shared__ uint32_t nbl[BK_MAXB];
    __shared__ uint32_t wtmp[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint2 *cr = (uint2 *)(tab + BK_TAB_CR);
    const uint32_t Rc = min(*Rc_dev, cap);
    const uint32_t start = tid < nrow ? hist_scanned[(int64_t)tid * ncol] : Rc;
    const uint32_t end = tid + 1 < nrow ? hist_scanned[(int64_t)(tid + 1) * ncol] : Rc;
    const uint32_t len = tid < nbk ? end - start : 0u;
    if (tid < nbk) cr[tid] = make_uint2(start, end);
    const uint32_t n = (len + BK_CHUNK - 1) / BK_CHUNK;
    nbl[tid] = n;
    __syncthreads();
    uint32_t wt;
    uint32_t ex = bk_wave_excl_sum(n, lane, wt);
    if (lane == 0) wtmp[wave] = wt;
    __syncthreads();
    for (int w = 0; w < wave; ++w) ex += wtmp[w];
    tab[BK_TAB_CS + tid] = ex;
    if (tid == 255) { tab[BK_TAB_CS + 256] = ex + n; tab[BK_TAB_NCH] = ex + n; }
    __syncthreads();
    uint32_t sum = 0;
    for (int u = 0; u < BK_TILES; ++u) {
        const int t = tid * BK_TILES + u;
        if (t < nt) {
            const uint32_t ty = (uint32_t)t / tiles_x, tx = (uint32_t)t - ty * tiles_x;
            sum += nbl[(ty / BK_H) * bx + tx / BK_W];
        }
    }
    uint32_t run = bk_wave_excl_sum(sum, lane, wt);
    if (lane == 0) wtmp[wave] = wt;
    __syncthreads();
    for (int w = 0; w < wave; ++w) run += wtmp[w];
    for (int u = 0; u < BK_TILES; ++u) {
        const int t = tid * BK_TILES + u;
        if (t < nt) {
            const uint32_t ty = (uint32_t)t / tiles_x, tx = (uint32_t)t - ty * tiles_x;
            tab[BK_TAB_TS + t] = run;
            run += nbl[(ty / BK_H) * bx + tx / BK_W];
            ranges[t] = make_uint2(0u, 0u);
        }
    }
}

struct BkChunk { uint32_t b, k, s, e; };

// chunk c -> (bucket, index inside the bucket, coarse entries [s, e)); cs = the chunk prefix in LDS
__device__ __forceinline__ BkChunk bk_chunk(uint32_t c, const uint32_t *cs, const uint32_t *__restrict__ tab) {
    uint32_t lo = 0, hi = BK_MAXB;                   // largest b with cs[b] <= c (buckets without chunks repeat the value: take the last)
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const uint32_t mid = (lo + hi) >> 1;
        if (cs[mid] <= c) lo = mid; else hi = mid;
    }
    BkChunk q;
    q.b = lo;
    q.k = c - cs[lo];
    const uint2 r = ((const uint2 *)(tab + BK_TAB_CR))[lo];
    q.s = r.x + q.k * BK_CHUNK;
    q.e = min(r.y, q.s + BK_CHUNK);
    // (a chunk belongs to a wave: say so, or buffer resources built from q need a waterfall loop around every access)
    q.b = (uint32_t)__builtin_amdgcn_readfirstlane((int)q.b);
    q.k = (uint32_t)__builtin_amdgcn_readfirstlane((int)q.k);
    q.s = (uint32_t)__builtin_amdgcn_readfirstlane((int)q.s);
    q.e = (uint32_t)__builtin_amdgcn_readfirstlane((int)q.e);
    return q;
}

// tile j of bucket b -> tile index, or -1 outside the grid
__device__ __forceinline__ int bk_tile(uint32_t b, int j, uint32_t bx, uint32_t tiles_x, uint32_t tiles_y) {
    const uint32_t tx = (b % bx) * BK_W + (uint32_t)(j & (BK_W - 1)), ty = (b / bx) * BK_H + (uint32_t)(j / BK_W);
    return tx < tiles_x && ty < tiles_y ? (int)(ty * tiles_x + tx) : -1;
}

// A chunk belongs to ONE WAVE (no workgroup barriers, no LDS staging): the wave's 64 lanes read 64 consecutive bucket entries
// per round; ballot j = the entries that cover tile j, in depth order, so entry l of the round goes to the tile's write
// position + mbcnt(ballot j) — the lanes of a ballot store a contiguous run.  The per-tile positions / counts are wave-uniform
// and live in scalar registers.
__global__ void __launch_bounds__(BK_THREADS)
    bk_count_kernel(const uint32_t *__restrict__ masks, const uint32_t *__restrict__ tab, uint32_t tiles_x, uint32_t tiles_y,
                    uint32_t bx, uint32_t *__restrict__ counts) {
    __shared__ uint32_t cs[BK_MAXB + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i <= BK_MAXB; i += BK_THREADS) cs[i] = tab[BK_TAB_CS + i];
    const uint32_t nch = tab[BK_TAB_NCH];
    __syncthreads();
    for (uint32_t c = blockIdx.x * BK_WAVES + (uint32_t)wave; c < nch; c += gridDim.x * BK_WAVES) {
        const BkChunk q = bk_chunk(c, cs, tab);
        uint32_t acc[BK_TILES];
#pragma unroll
        for (int j = 0; j < BK_TILES; ++j) acc[j] = 0;
        const ClBuf bm = cl_buf(masks + q.s, (uint64_t)(q.e - q.s) * 4u);      // a lane past the chunk's end reads zero
        const uint32_t n = q.e - q.s;
        uint32_t m_next = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(bm, lane * 4, 0, 0);
        for (uint32_t p0 = 0; p0 < n; p0 += CGS_WAVE) {
            const uint32_t m = m_next;
            m_next = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(bm, (int)((p0 + CGS_WAVE + (uint32_t)lane) * 4u), 0, 0);
#pragma unroll
            for (int j = 0; j < BK_TILES; ++j) acc[j] += (uint32_t)__builtin_popcountll(__ballot(m & (1u << j)));
        }
        uint32_t mine = 0;
#pragma unroll
        for (int j = 0; j < BK_TILES; ++j) mine = lane == j ? acc[j] : mine;
        if (lane < BK_TILES) {
            const int t = bk_tile(q.b, lane, bx, tiles_x, tiles_y);
            if (t >= 0) counts[tab[BK_TAB_TS + t] + q.k] = mine;
        }
    }
}

// (raw buffer accesses: a lane without an entry / without the bit issues the same instruction with an out-of-range offset — no
//  exec-mask branch, hence no join at which the compiler waits for everything in flight; the store's bounds check against the
//  workspace capacity also covers a speculative launch whose capacity was too small.  csrc/buf_access.h)
__global__ void __launch_bounds__(BK_THREADS)
    bk_fill_kernel(const uint32_t *__restrict__ ids, const uint32_t *__restrict__ masks, const uint32_t *__restrict__ tab,
                   uint32_t tiles_x, uint32_t tiles_y, uint32_t bx, const uint32_t *__restrict__ counts,
                   const uint32_t *__restrict__ S, uint32_t cap, uint32_t *__restrict__ out, uint2 *__restrict__ ranges) {
    __shared__ uint32_t cs[BK_MAXB + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i <= BK_MAXB; i += BK_THREADS) cs[i] = tab[BK_TAB_CS + i];
    const uint32_t nch = tab[BK_TAB_NCH];
    const ClBuf bout = cl_buf(out, (uint64_t)cap * 4u);
    __syncthreads();
    for (uint32_t c = blockIdx.x * BK_WAVES + (uint32_t)wave; c < nch; c += gridDim.x * BK_WAVES) {
        const BkChunk q = bk_chunk(c, cs, tab);
        uint32_t first = 0;                                        // lane j < 32: where the chunk's first entry of tile j goes
        if (lane < BK_TILES) {
            const int t = bk_tile(q.b, lane, bx, tiles_x, tiles_y);
            if (t >= 0) {
                const uint32_t slot = tab[BK_TAB_TS + t];
                first = S[slot + q.k];
                if (q.k == 0) {                                    // the bucket's first chunk also publishes its tiles' ranges
                    const uint32_t last = slot + (cs[q.b + 1] - cs[q.b]) - 1u;
                    // (min with cap: a speculative launch whose capacity was too small must still leave ranges inside the
                    //  workspace — the blend behind it reads them before the caller renders again)
                    const uint32_t end = min(S[last] + counts[last], cap);
                    if (end > first) ranges[t] = make_uint2(first, end);
                }
            }
        }
        uint32_t gb[BK_TILES];                                     // byte offsets, wave-uniform: scalar registers
#pragma unroll
        for (int j = 0; j < BK_TILES; ++j) gb[j] = (uint32_t)__builtin_amdgcn_readlane((int)first, j) * 4u;
        // the chunk's entries [q.s, q.e) as buffers of their own: a lane past the end reads zeros
        const ClBuf bm = cl_buf(masks + q.s, (uint64_t)(q.e - q.s) * 4u), bi = cl_buf(ids + q.s, (uint64_t)(q.e - q.s) * 4u);
        const uint32_t n = q.e - q.s;
        uint32_t m_next = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(bm, lane * 4, 0, 0);
        uint32_t id_next = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(bi, lane * 4, 0, 0);
        // (the first round's entries are waited for HERE: a wait for them at the loop head would also make every later round
        //  wait for the previous round's 32 stores)
        asm volatile("" : "+v"(m_next), "+v"(id_next));
        for (uint32_t p0 = 0; p0 < n; p0 += CGS_WAVE) {
            const uint32_t m = m_next, id = id_next;
            const int on = (int)((p0 + CGS_WAVE + (uint32_t)lane) * 4u);      // the next round's entries are in flight during this one
            m_next = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(bm, on, 0, 0);
            id_next = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(bi, on, 0, 0);
#pragma unroll
            for (int j = 0; j < BK_TILES; ++j) {
                const bool bit = m & (1u << j);
                const uint64_t bal = __ballot(bit);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                __builtin_amdgcn_raw_buffer_store_b32((int)id, bout, (int)cl_sel(bit, gb[j] + rank * 4u), 0, 0);
                gb[j] += (uint32_t)__builtin_popcountll(bal) * 4u;
            }
        }
    }
}

}  // namespace

// after the depth sort: rectangles and tile counts in depth order (g.sort_b / g.sort_d / g.sort_a)
int cgs_launch_gather_rects(int64_t P, CgsGeom &g, hipStream_t stream) {
    if (P == 0) return CGS_OK;
    hipLaunchKernelGGL(gather_rects_kernel, dim3(GR_XCD ? cgs_xcd_grid((P + 256 * GR_ITEMS - 1) / (256 * GR_ITEMS)) : (unsigned)((P + 256 * GR_ITEMS - 1) / (256 * GR_ITEMS))), dim3(256), 0, stream, P,
                       (const uint32_t *)g.order, (const uint2 *)g.rect, g.sort_b, g.sort_d, g.sort_a);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

bool cgs_tile_bin16_ok(int tile_bits) { return tile_bits <= 16; }

// per-tile lists of Gaussian ids (b.gid_sorted) and their tile keys ((uint16_t *)b.tile_key_c), depth order inside a tile
// R_dev != nullptr: speculative launch — R is a CAPACITY (the binning workspace holds R pairs), the pair count is read on
// the device from *R_dev by every kernel; a count above the capacity leaves garbage inside the workspace and the caller
// renders again (cgs_raster_render_spec).
int cgs_launch_tile_bin16(const cgs_raster_cfg *cfg, int64_t P, int64_t R, int tile_bits, CgsGeom &g, CgsBin &b, CgsImg &im,
                          hipStream_t stream, const uint32_t *R_dev) {
    const int nt = cgs_tiles_x(cfg) * cgs_tiles_y(cfg);
    if (R == 0 || P == 0) {
        CGS_CHECK_HIP(hipMemsetAsync(im.ranges, 0, (size_t)nt * sizeof(uint2), stream));
        return CGS_OK;
    }
    if (R >= (1ll << 32) - TB_TILE) { cgs_set_error("tile binning: pair count out of range"); return CGS_ERR_ARG; }
    const int64_t nb = (R + TB_TILE - 1) / TB_TILE;
    const int bits_b = tile_bits > 8 ? tile_bits / 2 : 0;          // second pass (high digit); 0: one pass is enough
    const int bits_a = tile_bits - bits_b;
    uint32_t *bf = b.tile_key_b;                                    // [nb + 1], the round-1 path's ping-pong buffer
    uint16_t *key_a = (uint16_t *)b.tile_key_a;
    uint32_t *hist = (uint32_t *)b.scratch;
    const size_t hist_bytes = cgs_align_up((size_t)TB_MAXR * nb * sizeof(uint32_t), 256);
    if (b.scratch_bytes < hist_bytes + cgs_scan_scratch_bytes((int64_t)TB_MAXR * nb)) {
        cgs_set_error("tile binning: scratch too small");
        return CGS_ERR_WORKSPACE;
    }
    char *scan_scratch = (char *)b.scratch + hist_bytes;
    const size_t scan_bytes = b.scratch_bytes - hist_bytes;
    const uint32_t tiles_x = (uint32_t)cgs_tiles_x(cfg);
    const uint32_t *offsets = g.offsets, *rlo = g.sort_b, *rhi = g.sort_d, *order = g.order;
    int rc;
    // the digit width is a template parameter: the ballot match unrolls to exactly that many steps
    // (columns XCD-aware, per-digit column scans + digit totals: cgs_xcd_item / cgs_launch_digit_scan, as in the depth sort)
    uint32_t *totals = (uint32_t *)scan_scratch;                    // [TB_MAXR]
    (void)scan_bytes;
    const unsigned grid = TB_XCD_MAP ? cgs_xcd_grid(nb) : (unsigned)nb;
#define TB_SCATTER_N(GEN, FINAL, N, KIN, VIN, KOUT, VOUT, SHIFT)                                                          \
    hipLaunchKernelGGL((tb_scatter_kernel<GEN, FINAL, N>), dim3(grid), dim3(TB_THREADS), 0, stream, KIN, VIN,             \
                       offsets, rlo, rhi, order, (const uint32_t *)bf, P, (uint32_t)R, tiles_x, KOUT, VOUT, im.ranges,    \
                       (const uint32_t *)hist, SHIFT, R_dev, (uint32_t)nb, TB_XCD_MAP, (const uint32_t *)totals)
#define TB_SCATTER(GEN, FINAL, NB, KIN, VIN, KOUT, VOUT, SHIFT)                                                           \
    do { switch (NB) {                                                                                                    \
        case 1: TB_SCATTER_N(GEN, FINAL, 1, KIN, VIN, KOUT, VOUT, SHIFT); break;                                          \
        case 2: TB_SCATTER_N(GEN, FINAL, 2, KIN, VIN, KOUT, VOUT, SHIFT); break;                                          \
        case 3: TB_SCATTER_N(GEN, FINAL, 3, KIN, VIN, KOUT, VOUT, SHIFT); break;                                          \
        case 4: TB_SCATTER_N(GEN, FINAL, 4, KIN, VIN, KOUT, VOUT, SHIFT); break;                                          \
        case 5: TB_SCATTER_N(GEN, FINAL, 5, KIN, VIN, KOUT, VOUT, SHIFT); break;                                          \
        case 6: TB_SCATTER_N(GEN, FINAL, 6, KIN, VIN, KOUT, VOUT, SHIFT); break;                                          \
        case 7: TB_SCATTER_N(GEN, FINAL, 7, KIN, VIN, KOUT, VOUT, SHIFT); break;                                          \
        default: TB_SCATTER_N(GEN, FINAL, 8, KIN, VIN, KOUT, VOUT, SHIFT); break;                                         \
    } } while (0)
    {
        // first pass: pairs generated from the per-Gaussian arrays, written partitioned by the low digit
        CgsProfScope prof(CGS_PROF_EMIT_PAIRS, stream);
        const int64_t nthr = nb + 1 > nt ? nb + 1 : nt;
        hipLaunchKernelGGL(block_first_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, stream, nb, P, offsets, bf,
                           nt, im.ranges, R_dev, (uint32_t)R);
        hipLaunchKernelGGL(tb_hist_kernel<true>, dim3(grid), dim3(TB_THREADS), 0, stream,
                           (const uint16_t *)nullptr, offsets, rlo, rhi, (const uint32_t *)bf, P, (uint32_t)R, tiles_x,
                           hist, 0, bits_a, R_dev, (uint32_t)nb, TB_XCD_MAP);
        CGS_CHECK_HIP(hipGetLastError());
        if ((rc = cgs_launch_digit_scan(hist, totals, 1 << bits_a, nb, stream))) return rc;
        if (bits_b)
            TB_SCATTER(true, false, bits_a, (const uint16_t *)nullptr, (const uint32_t *)nullptr, key_a, b.gid_a, 0);
        else
            TB_SCATTER(true, true, bits_a, (const uint16_t *)nullptr, (const uint32_t *)nullptr, (uint16_t *)nullptr,
                       b.gid_sorted, 0);
        CGS_CHECK_LAUNCH(stream, cfg->debug);
    }
    if (bits_b) {
        CgsProfScope prof(CGS_PROF_TILE_SORT, stream);
        hipLaunchKernelGGL(tb_hist_kernel<false>, dim3(grid), dim3(TB_THREADS), 0, stream,
                           (const uint16_t *)key_a, offsets, rlo, rhi, (const uint32_t *)bf, P, (uint32_t)R, tiles_x, hist,
                           bits_a, bits_b, R_dev, (uint32_t)nb, TB_XCD_MAP);
        CGS_CHECK_HIP(hipGetLastError());
        if ((rc = cgs_launch_digit_scan(hist, totals, 1 << bits_b, nb, stream))) return rc;
        TB_SCATTER(false, true, bits_b, (const uint16_t *)key_a, (const uint32_t *)b.gid_a, (uint16_t *)nullptr,
                   b.gid_sorted, bits_a);
        CGS_CHECK_LAUNCH(stream, cfg->debug);
    }
    {
        CgsProfScope prof(CGS_PROF_RANGES, stream);
        hipLaunchKernelGGL(ranges_fix_kernel, dim3(1), dim3(1024), 0, stream, nt, im.ranges);
        CGS_CHECK_LAUNCH(stream, cfg->debug);
    }
    return CGS_OK;
}

// ---- launcher of the two-level binning ------------------------------------------------------------------------------------
static void bk_dims(const cgs_raster_cfg *cfg, uint32_t &bx, uint32_t &by) {
    bx = (uint32_t)(cgs_tiles_x(cfg) + BK_W - 1) / BK_W;
    by = (uint32_t)(cgs_tiles_y(cfg) + BK_H - 1) / BK_H;
}

bool cgs_tile_bin_buckets_ok(const cgs_raster_cfg *cfg) {
    uint32_t bx, by;
    bk_dims(cfg, bx, by);
    return bx * by <= BK_MAXB;
}
// (the fill kernel addresses the lists through 32-bit byte offsets)
bool cgs_tile_bin_buckets_fits(int64_t R) { return R < (1ll << 30) - TB_TILE; }

// count slots of the [tile][chunk] table for a workspace of R pairs (the number of chunks itself is known on the device only)
int64_t cgs_bucket_count_slots(int64_t R) { return (int64_t)BK_TILES * ((R + BK_CHUNK - 1) / BK_CHUNK + BK_MAXB); }
size_t cgs_bucket_tab_words(void) { return BK_TAB_WORDS; }

// Same contract as cgs_launch_tile_bin16 (R_dev: speculative launch, R = capacity).  Needs cgs_tile_bin_buckets_ok(cfg).
int cgs_launch_tile_bin_buckets(const cgs_raster_cfg *cfg, int64_t P, int64_t R, CgsGeom &g, CgsBin &b, CgsImg &im,
                                hipStream_t stream, const uint32_t *R_dev) {
    (void)R_dev;                                      // the bucket-pair count is always read on the device (g.total[1])
    const int nt = cgs_tiles_x(cfg) * cgs_tiles_y(cfg);
    if (R == 0 || P == 0) {
        CGS_CHECK_HIP(hipMemsetAsync(im.ranges, 0, (size_t)nt * sizeof(uint2), stream));
        return CGS_OK;
    }
    if (!cgs_tile_bin_buckets_fits(R)) { cgs_set_error("tile binning (buckets): pair count out of range"); return CGS_ERR_ARG; }
    uint32_t bx, by;
    bk_dims(cfg, bx, by);
    const int nbk = (int)(bx * by);
    int bits = 1;
    while ((1 << bits) < nbk) ++bits;
    const int64_t nb = (R + TB_TILE - 1) / TB_TILE;  // the bucket pairs are at most the tile pairs
    const int64_t nbc = nb < 2048 ? nb : 2048;       // columns of the pass (the kernels split the blocks the device counts over them)
    uint32_t *bf = b.tile_key_b;
    uint32_t *hist = (uint32_t *)b.scratch;
    const size_t hist_bytes = cgs_align_up((size_t)TB_MAXR * nbc * sizeof(uint32_t), 256);
    if (b.scratch_bytes < hist_bytes + cgs_scan_scratch_bytes((int64_t)TB_MAXR * nbc) || !b.bk_tab || !g.sort_c) {
        cgs_set_error("tile binning: scratch too small");
        return CGS_ERR_WORKSPACE;
    }
    char *scan_scratch = (char *)b.scratch + hist_bytes;
    const size_t scan_bytes = b.scratch_bytes - hist_bytes;
    const uint32_t tiles_x = (uint32_t)cgs_tiles_x(cfg), tiles_y = (uint32_t)cgs_tiles_y(cfg);
    const uint32_t *rlo = g.sort_b, *rhi = g.sort_d, *order = g.order;
    uint32_t *coff = g.sort_c, *Rc_dev = g.total + 1;
    uint32_t *tab = b.bk_tab;
    uint2 *cranges = (uint2 *)(tab + BK_TAB_CR);
    uint32_t *cid = b.gid_a, *cmask = b.tile_key_a;
    int rc;
    static const bool tr = getenv("CGS_BK_TRACE") != nullptr;      // debugging: synchronise and report after every launch
#define BK_TR(name) do { if (tr) { hipError_t e_ = hipStreamSynchronize(stream); fprintf(stderr, "[bk] %s: %s\n", name, hipGetErrorString(e_)); } } while (0)
    {
        CgsProfScope prof(CGS_PROF_EMIT_PAIRS, stream);
        hipLaunchKernelGGL(bk_coarse_count_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, stream, P, rlo, rhi, coff);
        CGS_CHECK_HIP(hipGetLastError());
        BK_TR("coarse_count");
        if ((rc = cgs_scan_exclusive_u32_total(coff, coff, P, g.scratch, g.scratch_bytes, Rc_dev, stream))) return rc;
        BK_TR("coarse scan");
        if (tr) { uint32_t h2[2]; hipMemcpy(h2, g.total, 8, hipMemcpyDeviceToHost); fprintf(stderr, "[bk] P %lld, tile pairs %u, bucket pairs %u\n", (long long)P, h2[0], h2[1]); }
        const int64_t nthr = nb + 1 > BK_MAXB ? nb + 1 : BK_MAXB;
        hipLaunchKernelGGL(block_first_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, stream, nb, P,
                           (const uint32_t *)coff, bf, BK_MAXB, cranges, (const uint32_t *)Rc_dev, (uint32_t)R);
        BK_TR("block_first");
        hipLaunchKernelGGL((tb_hist_kernel<true, true>), dim3((unsigned)nbc), dim3(TB_THREADS), 0, stream,
                           (const uint16_t *)nullptr, (const uint32_t *)coff, rlo, rhi, (const uint32_t *)bf, P, (uint32_t)R, bx,
                           hist, 0, bits, (const uint32_t *)Rc_dev, (uint32_t)nbc, 0);
        CGS_CHECK_HIP(hipGetLastError());
        BK_TR("hist");
        if ((rc = cgs_scan_exclusive_u32_total(hist, hist, (int64_t)(1 << bits) * nbc, scan_scratch, scan_bytes, nullptr, stream)))
            return rc;
        BK_TR("hist scan");
#define BK_SCATTER_N(N)                                                                                                   \
    hipLaunchKernelGGL((tb_scatter_kernel<true, true, N, true>), dim3((unsigned)nbc), dim3(TB_THREADS), 0, stream,        \
                       (const uint16_t *)nullptr, (const uint32_t *)nullptr, (const uint32_t *)coff, rlo, rhi, order,     \
                       (const uint32_t *)bf, P, (uint32_t)R, bx, (uint16_t *)nullptr, cid, cranges, (const uint32_t *)hist, 0, \
                       (const uint32_t *)Rc_dev, (uint32_t)nbc, 0, (const uint32_t *)nullptr, cmask)
        switch (bits) {
            case 1: BK_SCATTER_N(1); break;
            case 2: BK_SCATTER_N(2); break;
            case 3: BK_SCATTER_N(3); break;
            case 4: BK_SCATTER_N(4); break;
            case 5: BK_SCATTER_N(5); break;
            case 6: BK_SCATTER_N(6); break;
            case 7: BK_SCATTER_N(7); break;
            default: BK_SCATTER_N(8); break;
        }
        BK_TR("scatter");
        CGS_CHECK_LAUNCH(stream, cfg->debug);
    }
    {
        CgsProfScope prof(CGS_PROF_TILE_SORT, stream);
        hipLaunchKernelGGL(bk_table_kernel, dim3(1), dim3(256), 0, stream, tab, nbk, bx, tiles_x, nt, im.ranges,
                           (const uint32_t *)hist, nbc, 1 << bits, (const uint32_t *)Rc_dev, (uint32_t)R);
        BK_TR("table");
        hipLaunchKernelGGL(bk_count_kernel, dim3(2048), dim3(BK_THREADS), 0, stream, (const uint32_t *)cmask,
                           (const uint32_t *)tab, tiles_x, tiles_y, bx, b.bk_counts);
        CGS_CHECK_HIP(hipGetLastError());
        BK_TR("count");
        if ((rc = cgs_scan_exclusive_u32_total(b.bk_counts, b.bk_scan, cgs_bucket_count_slots(R), b.bk_scan_scratch,
                                               b.bk_scan_scratch_bytes, nullptr, stream)))
            return rc;
        BK_TR("count scan");
        hipLaunchKernelGGL(bk_fill_kernel, dim3(2048), dim3(BK_THREADS), 0, stream, (const uint32_t *)cid, (const uint32_t *)cmask,
                           (const uint32_t *)tab, tiles_x, tiles_y, bx, (const uint32_t *)b.bk_counts,
                           (const uint32_t *)b.bk_scan, (uint32_t)R, b.gid_sorted, im.ranges);
        BK_TR("fill");
        CGS_CHECK_LAUNCH(stream, cfg->debug);
    }
    return CGS_OK;
}
