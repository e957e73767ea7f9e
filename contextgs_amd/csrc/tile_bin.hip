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

#define TB_THREADS 256
#define TB_ITEMS 16
#define TB_TILE (TB_THREADS * TB_ITEMS)
#define TB_WAVES (TB_THREADS / CGS_WAVE)
#define TB_MAXR 256

int cgs_scan_exclusive_u32_total(const uint32_t *in, uint32_t *out, int64_t n, void *scratch, size_t scratch_bytes,
                                 uint32_t *grand_total, hipStream_t stream);

namespace {

// rect_lo / rect_hi[i] = packed tile rectangle of the i-th Gaussian in depth order, cnt[i] = its tile count
__global__ void __launch_bounds__(256)
    gather_rects_kernel(int64_t P, const uint32_t *__restrict__ order, const uint2 *__restrict__ rect,
                        uint32_t *__restrict__ rect_lo, uint32_t *__restrict__ rect_hi, uint32_t *__restrict__ cnt) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const uint2 rc = rect[order[i]];
    rect_lo[i] = rc.x;
    rect_hi[i] = rc.y;
    const uint32_t w = (rc.y & 0xFFFFu) - (rc.x & 0xFFFFu), h = (rc.y >> 16) - (rc.x >> 16);
    cnt[i] = w * h;
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

struct TbPair { uint32_t key, i; };

// pair j (global index) -> (tile key, index of its Gaussian in depth order)
__device__ __forceinline__ TbPair gen_pair(uint32_t j, uint32_t base, uint32_t i_lo, const uint32_t *sidx,
                                           const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ rect_lo,
                                           const uint32_t *__restrict__ rect_hi, uint32_t tiles_x) {
    TbPair p;
    p.i = i_lo + sidx[j - base];
    const uint32_t local = j - offsets[p.i];
    const uint32_t lo = rect_lo[p.i], hi = rect_hi[p.i];
    const uint32_t x0 = lo & 0xFFFFu, y0 = lo >> 16, w = (hi & 0xFFFFu) - x0;
    // local / w for local < 2^24, w < 2^16: float quotient, one correction step either way
    uint32_t q = (uint32_t)((float)local * __builtin_amdgcn_rcpf((float)w));
    int32_t r = (int32_t)(local - q * w);
    if (r < 0) { --q; r += (int32_t)w; }
    if (r >= (int32_t)w) { ++q; r -= (int32_t)w; }
    p.key = (y0 + q) * tiles_x + x0 + (uint32_t)r;
    return p;
}

template <bool GEN>
__global__ void __launch_bounds__(TB_THREADS)
    tb_hist_kernel(const uint16_t *__restrict__ keys, const uint32_t *__restrict__ offsets,
                   const uint32_t *__restrict__ rect_lo, const uint32_t *__restrict__ rect_hi,
                   const uint32_t *__restrict__ bf, int64_t P, uint32_t R, uint32_t tiles_x,
                   uint32_t *__restrict__ hist /*[rows][nblocks]*/, int shift, int nbits, const uint32_t *__restrict__ R_dev) {
    if (R_dev) R = min(*R_dev, R);         // speculative launch: R (argument) is the capacity, the count is on the device
    __shared__ uint32_t h[TB_MAXR];
    __shared__ uint32_t sidx[GEN ? TB_TILE : 1];
    __shared__ uint32_t swave[TB_WAVES];
    const int tid = threadIdx.x;
    const uint32_t mask = (1u << nbits) - 1u;
    h[tid] = 0;
    const uint32_t base = blockIdx.x * (uint32_t)TB_TILE;
    uint32_t i_lo = 0;
    if (GEN && base < R) {                 // (base >= R: a block past the pairs of a speculative launch writes zeros)
        i_lo = bf[blockIdx.x];
        build_owner_index(sidx, swave, base, i_lo, bf[blockIdx.x + 1], P, R, offsets);
    } else {
        __syncthreads();
    }
#pragma unroll 4
    for (int k = 0; k < TB_ITEMS; ++k) {
        const uint32_t j = base + (uint32_t)(k * TB_THREADS + tid);
        if (j < R) {
            const uint32_t key = GEN ? gen_pair(j, base, i_lo, sidx, offsets, rect_lo, rect_hi, tiles_x).key
                                     : (uint32_t)keys[j];
            atomicAdd(&h[(key >> shift) & mask], 1u);
        }
    }
    __syncthreads();
    if (tid < (1 << nbits)) hist[(int64_t)tid * gridDim.x + blockIdx.x] = h[tid];
}

// One stable radix pass.  GEN: the block's pairs come from gen_pair (first pass); otherwise from (keys_in, vals_in).
template <bool GEN, bool FINAL, int NBITS>
__global__ void __launch_bounds__(TB_THREADS)
    tb_scatter_kernel(const uint16_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                      const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ rect_lo,
                      const uint32_t *__restrict__ rect_hi, const uint32_t *__restrict__ order,
                      const uint32_t *__restrict__ bf, int64_t P, uint32_t R, uint32_t tiles_x,
                      uint16_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, uint2 *__restrict__ ranges,
                      const uint32_t *__restrict__ hist_scanned, int shift, const uint32_t *__restrict__ R_dev) {
    if (R_dev) R = min(*R_dev, R);
    if (blockIdx.x * (uint32_t)TB_TILE >= R) return;
    constexpr int nbits = NBITS;
    __shared__ uint32_t wcnt[TB_WAVES][TB_MAXR];
    __shared__ uint32_t lstart[TB_MAXR], gbase[TB_MAXR];
    __shared__ uint32_t wsum[TB_WAVES];
    __shared__ uint32_t stage[TB_TILE + TB_TILE / 2];      // sval[TB_TILE] | skey (u16)[TB_TILE]; the owner index before
    uint32_t *sval = stage;
    uint16_t *skey = (uint16_t *)(stage + TB_TILE);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t mask = (1u << nbits) - 1u;
#pragma unroll
    for (int w = 0; w < TB_WAVES; ++w) wcnt[w][tid] = 0;
    const uint32_t base = blockIdx.x * (uint32_t)TB_TILE;
    uint32_t i_lo = 0;
    if (GEN) {
        i_lo = bf[blockIdx.x];
        build_owner_index(stage, wsum, base, i_lo, bf[blockIdx.x + 1], P, R, offsets);
    } else {
        __syncthreads();
    }
    const uint32_t wbase = base + (uint32_t)wave * (TB_TILE / TB_WAVES);
    uint32_t key[TB_ITEMS], val[TB_ITEMS], rank[TB_ITEMS];
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
                const TbPair p = gen_pair(j, base, i_lo, stage, offsets, rect_lo, rect_hi, tiles_x);
                key[r] = p.key;
                val[r] = order[p.i];
            } else {
                key[r] = keys_in[j];
                val[r] = vals_in[j];
            }
        }
        const uint32_t d = (key[r] >> shift) & mask;
        uint64_t peers = __ballot(valid);
#if defined(CGS_EXPERIMENTS) && defined(TB_ABL) && TB_ABL == 1      // timing only: no digit match, identity placement
        rank[r] = (uint32_t)(wave * (TB_TILE / TB_WAVES) + r * 64 + lane);
        continue;
#endif
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
        gbase[d] = (d < (1 << nbits) ? hist_scanned[(int64_t)d * gridDim.x + blockIdx.x] : 0u) - run;
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
#if defined(CGS_EXPERIMENTS) && defined(TB_ABL) && TB_ABL == 1
            const uint32_t slot = rank[r];
#else
            const uint32_t slot = wcnt[wave][d] + rank[r];
#endif
            skey[slot] = (uint16_t)key[r];
            sval[slot] = val[r];
        }
    }
    __syncthreads();
    const int count = (int)min((uint32_t)TB_TILE, R - base);
#if defined(CGS_EXPERIMENTS) && defined(TB_ABL) && TB_ABL == 2      // timing only: no global stores
    if (R != 0xFFFFFFFFu) return;
#endif
    for (int j = tid; j < count; j += TB_THREADS) {
        const uint32_t k = skey[j];
        const uint32_t d = (k >> shift) & mask;
#if defined(CGS_EXPERIMENTS) && defined(TB_ABL) && TB_ABL == 1
        const uint32_t pos = base + (uint32_t)j + (gbase[d] & 0u) + (lstart[d] & 0u);
#else
        const uint32_t pos = gbase[d] + (uint32_t)j;
#endif
        vals_out[pos] = sval[j];
        if (FINAL) {
            // equal keys are adjacent in the staged order (same digit; inside a digit the arrival order is the order
            // the previous pass left, ascending in the low digit): the last item of a run bounds the tile's list
            if (j + 1 == count || (uint32_t)skey[j + 1] != k) atomicMax(&ranges[k].y, pos + 1u);
        } else {
            keys_out[pos] = (uint16_t)k;
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

}  // namespace

// after the depth sort: rectangles and tile counts in depth order (g.sort_b / g.sort_d / g.sort_a)
int cgs_launch_gather_rects(int64_t P, CgsGeom &g, hipStream_t stream) {
    if (P == 0) return CGS_OK;
    hipLaunchKernelGGL(gather_rects_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, stream, P,
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
#define TB_SCATTER_N(GEN, FINAL, N, KIN, VIN, KOUT, VOUT, SHIFT)                                                          \
    hipLaunchKernelGGL((tb_scatter_kernel<GEN, FINAL, N>), dim3((unsigned)nb), dim3(TB_THREADS), 0, stream, KIN, VIN,     \
                       offsets, rlo, rhi, order, (const uint32_t *)bf, P, (uint32_t)R, tiles_x, KOUT, VOUT, im.ranges,    \
                       (const uint32_t *)hist, SHIFT, R_dev)
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
        hipLaunchKernelGGL(tb_hist_kernel<true>, dim3((unsigned)nb), dim3(TB_THREADS), 0, stream,
                           (const uint16_t *)nullptr, offsets, rlo, rhi, (const uint32_t *)bf, P, (uint32_t)R, tiles_x,
                           hist, 0, bits_a, R_dev);
        CGS_CHECK_HIP(hipGetLastError());
        if ((rc = cgs_scan_exclusive_u32_total(hist, hist, (int64_t)(1 << bits_a) * nb, scan_scratch, scan_bytes, nullptr,
                                               stream)))
            return rc;
        if (bits_b)
            TB_SCATTER(true, false, bits_a, (const uint16_t *)nullptr, (const uint32_t *)nullptr, key_a, b.gid_a, 0);
        else
            TB_SCATTER(true, true, bits_a, (const uint16_t *)nullptr, (const uint32_t *)nullptr, (uint16_t *)nullptr,
                       b.gid_sorted, 0);
        CGS_CHECK_LAUNCH(stream, cfg->debug);
    }
    if (bits_b) {
        CgsProfScope prof(CGS_PROF_TILE_SORT, stream);
        hipLaunchKernelGGL(tb_hist_kernel<false>, dim3((unsigned)nb), dim3(TB_THREADS), 0, stream,
                           (const uint16_t *)key_a, offsets, rlo, rhi, (const uint32_t *)bf, P, (uint32_t)R, tiles_x, hist,
                           bits_a, bits_b, R_dev);
        CGS_CHECK_HIP(hipGetLastError());
        if ((rc = cgs_scan_exclusive_u32_total(hist, hist, (int64_t)(1 << bits_b) * nb, scan_scratch, scan_bytes, nullptr,
                                               stream)))
            return rc;
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
