// Device-wide primitives for the binning stage: exclusive scan (R3 in
// SURVEY §2.1) and a stable LSD radix sort of (u32 key, u32 value) pairs
// (R5).  Written for wave64: ranks inside a wave come from 64-bit ballots,
// cross-wave composition goes through LDS, nothing uses 32-lane idioms.
#include "cgs_internal.h"

#define SCAN_THREADS 256
#define SCAN_ITEMS 8
#define SCAN_TILE (SCAN_THREADS * SCAN_ITEMS)

#define SORT_THREADS 256
#ifndef SORT_ITEMS
#define SORT_ITEMS 16
#endif
#define SORT_TILE (SORT_THREADS * SORT_ITEMS)
#define SORT_WAVES (SORT_THREADS / CGS_WAVE)
#define RADIX_BITS 8
#define RADIX 256

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// Block-wide exclusive scan of one value per thread (256 threads); returns the
// exclusive prefix, *total gets the block sum.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *total,
                                                         uint32_t *lds_wave /*[4]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = wave_inclusive_scan(v, lane);
    if (lane == 63) lds_wave[wave] = inc;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; ++w) {
        uint32_t s = lds_wave[w];
        if (w < wave) wbase += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return wbase + inc - v;
}

// A thread's SCAN_ITEMS (8) consecutive items as two 16-byte accesses when the run lies inside [0, n) and on a 16-byte boundary
// (it does for every full tile of a 16-byte-aligned array); element by element otherwise.  (Round 6: the element-wise form made
// every load instruction of a wave touch all of the wave's 64 32-byte segments for 4 bytes each.)
__device__ __forceinline__ void scan_load8(const uint32_t *__restrict__ in, int64_t base, int64_t n, uint32_t (&v)[SCAN_ITEMS]) {
    static_assert(SCAN_ITEMS == 8, "scan_load8");
    if (base + SCAN_ITEMS <= n && (((uintptr_t)(in + base)) & 15) == 0) {
        const uint4 a = *(const uint4 *)(in + base), b = *(const uint4 *)(in + base + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; ++i) v[i] = base + i < n ? in[base + i] : 0u;
    }
}
// (uint8 input — flags, one byte per item, ANY non-zero byte counts as 1: a thread's eight bytes are one 8-byte load)
__device__ __forceinline__ void scan_load8(const uint8_t *__restrict__ in, int64_t base, int64_t n, uint32_t (&v)[SCAN_ITEMS]) {
    if (base + SCAN_ITEMS <= n && (((uintptr_t)(in + base)) & 7) == 0) {
        const uint2 a = *(const uint2 *)(in + base);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = ((a.x >> (8 * i)) & 0xFFu) ? 1u : 0u; v[4 + i] = ((a.y >> (8 * i)) & 0xFFu) ? 1u : 0u; }
    } else {
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; ++i) v[i] = (base + i < n && in[base + i]) ? 1u : 0u;
    }
}
__device__ __forceinline__ void scan_store8(uint32_t *__restrict__ out, int64_t base, int64_t n, const uint32_t (&v)[SCAN_ITEMS]) {
    if (base + SCAN_ITEMS <= n && (((uintptr_t)(out + base)) & 15) == 0) {
        *(uint4 *)(out + base) = make_uint4(v[0], v[1], v[2], v[3]);
        *(uint4 *)(out + base + 4) = make_uint4(v[4], v[5], v[6], v[7]);
    } else {
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; ++i)
            if (base + i < n) out[base + i] = v[i];
    }
}

template <typename IN>
__global__ void __launch_bounds__(SCAN_THREADS) scan_reduce_kernel(const IN *__restrict__ in,
                                                                   uint32_t *__restrict__ sums,
                                                                   int64_t n) {
    __shared__ uint32_t lds_wave[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS], acc = 0;
    scan_load8(in, base, n, v);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) acc += v[i];
    uint32_t tot;
    block_exclusive_scan(acc, &tot, lds_wave);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// Scans one tile per block.  block_offsets == nullptr => single-tile call.
template <typename IN>
__global__ void __launch_bounds__(SCAN_THREADS)
    scan_apply_kernel(const IN *__restrict__ in, uint32_t *__restrict__ out,
                      const uint32_t *__restrict__ block_offsets, int64_t n,
                      uint32_t *__restrict__ grand_total) {
    __shared__ uint32_t lds_wave[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t acc = 0;
    scan_load8(in, base, n, v);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) acc += v[i];
    uint32_t tot;
    uint32_t excl = block_exclusive_scan(acc, &tot, lds_wave);
    uint32_t run = excl + (block_offsets ? block_offsets[blockIdx.x] : 0u);
    uint32_t o[SCAN_ITEMS];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) { o[i] = run; run += v[i]; }
    scan_store8(out, base, n, o);
    if (grand_total && blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_THREADS - 1)
        *grand_total = run;
}

// The same, but block_totals holds the per-block TOTALS (scan_reduce_kernel's output, not yet scanned): every block adds
// up the totals of the blocks before it itself (<= SCAN_SELF_PREFIX_MAX values, L2-resident), which removes the
// recursive scan of the block sums: two launches per scan instead of three to five.
#define SCAN_SELF_PREFIX_MAX 16384
template <typename IN>
__global__ void __launch_bounds__(SCAN_THREADS)
    scan_apply_selfprefix_kernel(const IN *__restrict__ in, uint32_t *__restrict__ out,
                                 const uint32_t *__restrict__ block_totals, int64_t n, uint32_t *__restrict__ grand_total) {
    __shared__ uint32_t lds_wave[4];
    __shared__ uint32_t lds_wave2[4];
    uint32_t before = 0;
    for (int j = threadIdx.x; j < (int)blockIdx.x; j += SCAN_THREADS) before += block_totals[j];
    uint32_t block_base;
    block_exclusive_scan(before, &block_base, lds_wave2);
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t acc = 0;
    scan_load8(in, base, n, v);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) acc += v[i];
    uint32_t tot;
    uint32_t excl = block_exclusive_scan(acc, &tot, lds_wave);
    uint32_t run = excl + block_base;
    uint32_t o[SCAN_ITEMS];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) { o[i] = run; run += v[i]; }
    scan_store8(out, base, n, o);
    if (grand_total && blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_THREADS - 1)
        *grand_total = run;
}

static int64_t scan_blocks(int64_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE; }

extern "C" size_t cgs_scan_scratch_bytes(int64_t n) {
    size_t bytes = 0;
    int64_t m = scan_blocks(n);
    while (m > 1) {
        bytes += cgs_align_up((size_t)m * sizeof(uint32_t), 256);
        m = scan_blocks(m);
    }
    return bytes + 256;
}

// Recursive helper: exclusive scan of `in` into `out`, optional grand total.  IN: uint32 or uint8 items (the block sums of the
// recursion are always uint32).
template <typename IN>
static int scan_rec(const IN *in, uint32_t *out, int64_t n, char *scratch, size_t scratch_bytes,
                    uint32_t *grand_total, hipStream_t stream) {
    if (n <= 0) return CGS_OK;
    int64_t nb = scan_blocks(n);
    if (nb == 1) {
        hipLaunchKernelGGL(scan_apply_kernel<IN>, dim3(1), dim3(SCAN_THREADS), 0, stream, in, out,
                           (const uint32_t *)nullptr, n, grand_total);
        CGS_CHECK_HIP(hipGetLastError());
        return CGS_OK;
    }
    size_t need = cgs_align_up((size_t)nb * sizeof(uint32_t), 256);
    if (need > scratch_bytes) {
        cgs_set_error("scan: scratch too small (%zu < %zu)", scratch_bytes, need);
        return CGS_ERR_WORKSPACE;
    }
    uint32_t *sums = (uint32_t *)scratch;
    hipLaunchKernelGGL(scan_reduce_kernel<IN>, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, stream, in, sums, n);
    CGS_CHECK_HIP(hipGetLastError());
    if (nb <= SCAN_SELF_PREFIX_MAX) {
        hipLaunchKernelGGL(scan_apply_selfprefix_kernel<IN>, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, stream, in, out,
                           (const uint32_t *)sums, n, grand_total);
        CGS_CHECK_HIP(hipGetLastError());
        return CGS_OK;
    }
    int rc = scan_rec<uint32_t>(sums, sums, nb, scratch + need, scratch_bytes - need, nullptr, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(scan_apply_kernel<IN>, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, stream, in, out,
                       (const uint32_t *)sums, n, grand_total);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

// exclusive scan of BYTES (0 / 1 flags) into uint32 positions + the grand total (the expansion's survivor flags: expand.hip)
int cgs_scan_exclusive_u8_total(const uint8_t *in, uint32_t *out, int64_t n, void *scratch, size_t scratch_bytes,
                                uint32_t *grand_total, hipStream_t stream) {
    return scan_rec<uint8_t>(in, out, n, (char *)scratch, scratch_bytes, grand_total, stream);
}

int cgs_scan_exclusive_u32_total(const uint32_t *in, uint32_t *out, int64_t n, void *scratch,
                                 size_t scratch_bytes, uint32_t *grand_total, hipStream_t stream) {
    return scan_rec<uint32_t>(in, out, n, (char *)scratch, scratch_bytes, grand_total, stream);
}

extern "C" int cgs_scan_exclusive_u32(const uint32_t *in, uint32_t *out, int64_t n, void *scratch,
                                      size_t scratch_bytes, void *stream) {
    if (n < 0) { cgs_set_error("scan: negative n"); return CGS_ERR_ARG; }
    return scan_rec<uint32_t>(in, out, n, (char *)scratch, scratch_bytes, nullptr, (hipStream_t)stream);
}

// ----------------------------------------------------------------------------
// Radix sort
// ----------------------------------------------------------------------------
// Tile layout shared by the histogram and scatter kernels: block b owns items
// [b*SORT_TILE, (b+1)*SORT_TILE); inside it wave w owns a contiguous quarter
// and walks it in rounds of 64 so that (wave, round, lane) is ascending input
// order — that is what makes the per-digit ranks stable.
//
// BITS = digit width (8: 256 digits, one per thread; 9: 512 digits, two consecutive ones per thread).  XF: the keys read
// from keys_in are DEPTH keys (float bits of a view depth > 0.2, or 0xFFFFFFFF for a culled Gaussian) and are sorted as
// cgs_depth_key27(): 27 bits = three 9-bit passes instead of four 8-bit ones (see cgs_sort_depth_keys).

// float bits of the near plane (raster_math.h: t.z <= 0.2f is culled): every live depth key lies above it
#define CGS_DEPTH_KEY_BASE 0x3E4CCCCDu
#define CGS_DEPTH_KEY_MAX 0x07FFFFFFu        // 2^27 - 1: culled Gaussians (their position in the order is never read: 0 tiles)

// Order-preserving on (0.2, ~13107): bits(z) - bits(0.2).  A live key outside that range saturates and reports itself
// (*overflow = epoch): the caller sorts that view again on the full 32 bits.
__device__ __forceinline__ uint32_t cgs_depth_key27(uint32_t k, uint32_t *overflow, uint32_t epoch) {
    if (k == 0xFFFFFFFFu) return CGS_DEPTH_KEY_MAX;
    const uint32_t d = k - CGS_DEPTH_KEY_BASE;          // (k below the base wraps to a huge value: reported like a far one)
    if (d >= CGS_DEPTH_KEY_MAX) {
        *overflow = epoch;                              // (same value from every reporting lane: a benign race)
        return CGS_DEPTH_KEY_MAX;
    }
    return d;
}

// Hardware workgroup -> tile: cgs_xcd_item() (cgs_internal.h).  The runs two NEIGHBOURING tiles write for a digit are adjacent in
// the output (a tile's run starts where its predecessor's ends); with tile = workgroup id neighbours always sit on different
// XCDs (workgroups go to the eight XCDs round robin) and every run is a partial-line write of its own, with the XCD-aware
// assignment they share an L2 that merges them: three 9-bit passes 192 -> 172 us, four 8-bit passes 214 -> 208 us at 5.8 M keys
// (tools/sort_micro.py, profiles/r06_depth_sort.txt).  SORT_XCD_MAP 0: tile = workgroup id.
#ifndef SORT_XCD_MAP
#define SORT_XCD_MAP 1
#endif
__device__ __forceinline__ int64_t sort_tile_of_block(int64_t nb) {
#if SORT_XCD_MAP
    return cgs_xcd_item(nb);
#else
    return (int64_t)blockIdx.x < nb ? (int64_t)blockIdx.x : -1;
#endif
}
static unsigned sort_grid(int64_t nb) { return cgs_xcd_grid(nb); }

template <int BITS, bool XF>
__global__ void __launch_bounds__(SORT_THREADS)
    radix_hist_kernel(const uint32_t *__restrict__ keys, uint32_t *__restrict__ hist /*[1 << BITS][nblocks]*/,
                      int64_t n, int64_t nb, int shift, uint32_t digit_mask, uint32_t *overflow, uint32_t epoch) {
    constexpr int NR = 1 << BITS;
    __shared__ uint32_t h[NR];
    const int64_t tile = sort_tile_of_block(nb);
    if (tile < 0) return;
    for (int d = threadIdx.x; d < NR; d += SORT_THREADS) h[d] = 0;
    __syncthreads();
    const int64_t base = tile * SORT_TILE;
    // (all of the thread's keys requested first, from clamped indices: a load inside `if (idx < n)` sits behind an exec-mask
    //  branch whose join drains the load queue — sixteen exposed round trips per thread)
    uint32_t kk[SORT_ITEMS];
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const int64_t idx = base + (int64_t)i * SORT_THREADS + threadIdx.x;
        kk[i] = keys[idx < n ? idx : n - 1];
    }
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const int64_t idx = base + (int64_t)i * SORT_THREADS + threadIdx.x;
        if (idx < n) {
            uint32_t k = kk[i];
            if (XF) k = cgs_depth_key27(k, overflow, epoch);
            atomicAdd(&h[(k >> shift) & digit_mask], 1u);
        }
    }
    __syncthreads();
    for (int d = threadIdx.x; d < NR; d += SORT_THREADS) hist[(int64_t)d * nb + tile] = h[d];
}

// Exclusive scan of every digit's row of the [digits][nblocks] table in place (workgroup d owns digit d: nblocks entries, a few
// per thread) + the digit's total.  The scatter kernel adds the digits' exclusive prefix itself (a scan over the digits it
// already has the code for): ONE launch between histogram and scatter instead of the generic scan's two (reduce + apply over
// the 256 x nblocks table, ~17 us per pass at 5.8 M keys).
__global__ void __launch_bounds__(SORT_THREADS)
    radix_digit_scan_kernel(uint32_t *__restrict__ hist /*[gridDim.x][nblocks]*/, uint32_t *__restrict__ totals /*[gridDim.x]*/,
                            int64_t nblocks) {
    __shared__ uint32_t wsum[SORT_WAVES];
    __shared__ uint32_t carry_s;
    uint32_t *row = hist + (int64_t)blockIdx.x * nblocks;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < nblocks; base += SORT_THREADS * 4) {
        const int64_t i0 = base + (int64_t)threadIdx.x * 4;
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (i0 + k < nblocks) ? row[i0 + k] : 0u;
        const uint32_t mine = v[0] + v[1] + v[2] + v[3];
        uint32_t inc = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(inc, o, 64);
            if (lane >= o) inc += up;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t run = carry_s;
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) run += (w < wave) ? wsum[w] : 0u;
        run += inc - mine;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i0 + k < nblocks) row[i0 + k] = run;
            run += v[k];
        }
        __syncthreads();
        if (threadIdx.x == SORT_THREADS - 1) carry_s = run;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry_s;
}

template <int BITS, bool XF>
__global__ void __launch_bounds__(SORT_THREADS)
    radix_scatter_kernel(const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                         uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
                         const uint32_t *__restrict__ hist_scanned, const uint32_t *__restrict__ digit_totals,
                         int64_t n, int64_t nb, int shift, uint32_t digit_mask) {
    constexpr int NR = 1 << BITS;
    constexpr int DPT = NR / SORT_THREADS;         // digits per thread: thread t owns digits t * DPT .. t * DPT + DPT - 1
    const int64_t tile = sort_tile_of_block(nb);
    if (tile < 0) return;
    // (16-bit counters and one global-minus-local base per digit: 38 KB of LDS per workgroup at 512 digits = four workgroups
    //  per CU; with 32-bit counters and separate bases it was 44 KB = three, and the pass 40 % slower than the 256-digit one)
    __shared__ uint16_t wcnt[SORT_WAVES][NR];      // running per-wave digit counts -> bases (a tile has 4096 items)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // (the table entries of the thread's digits are requested now, the ranking below hides the latency)
    uint32_t gtot[DPT], hrow[DPT];
#pragma unroll
    for (int j = 0; j < DPT; ++j) {
        const int d = threadIdx.x * DPT + j;
        gtot[j] = digit_totals[d];
        hrow[j] = hist_scanned[(int64_t)d * nb + tile];
    }
#pragma unroll
    for (int w = 0; w < SORT_WAVES; ++w)
        for (int d = threadIdx.x; d < NR; d += SORT_THREADS) wcnt[w][d] = 0;
    __syncthreads();

    const int64_t wbase = tile * SORT_TILE + (int64_t)wave * (SORT_TILE / SORT_WAVES);
    uint32_t key[SORT_ITEMS], val[SORT_ITEMS], rank[SORT_ITEMS];
    volatile uint16_t *my = wcnt[wave];
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    uint32_t dummy_overflow;
    // (keys and values of all rounds requested before the first is ranked, from clamped indices: see radix_hist_kernel)
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const int64_t idx = wbase + (int64_t)r * 64 + lane;
        const int64_t ci = idx < n ? idx : n - 1;
        key[r] = keys_in[ci];
        val[r] = vals_in ? vals_in[ci] : (uint32_t)idx;
    }
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const int64_t idx = wbase + (int64_t)r * 64 + lane;
        const bool valid = idx < n;
        key[r] = valid ? key[r] : 0xFFFFFFFFu;
        if (XF) key[r] = cgs_depth_key27(key[r], &dummy_overflow, 0u);      // (the histogram kernel of this pass reported overflows)
        val[r] = valid ? val[r] : 0u;                                       // vals_in == nullptr: values = positions
        const uint32_t d = (key[r] >> shift) & digit_mask;
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < BITS; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        uint32_t old = 0;
        if (valid) {
            const int leader = __builtin_ctzll(peers);
            if (lane == leader) {
                old = my[d];
                my[d] = (uint16_t)(old + (uint32_t)__builtin_popcountll(peers));
            }
            old = __shfl(old, leader, 64);
            rank[r] = old + (uint32_t)__builtin_popcountll(peers & lt_mask);
        } else {
            rank[r] = 0;
        }
    }
    __syncthreads();
    // Thread t owns DPT consecutive digits.  Items are first placed in LDS in (digit, wave, round, lane) order — the order
    // they must have in the output — and then streamed out: consecutive LDS slots of one digit go to consecutive global
    // addresses, so a wave writes a few contiguous runs (avg. SORT_TILE / digits items each) instead of 64 isolated
    // 4-byte stores per instruction.
    __shared__ uint32_t gdelta[NR];        // (first global position of this block's items of the digit) - (its first LDS slot)
    __shared__ uint32_t wsum[SORT_WAVES], gsum[SORT_WAVES];
    __shared__ uint32_t skey[SORT_TILE], sval[SORT_TILE];
    {
        uint32_t tot[DPT], mine = 0, gmine = 0;
#pragma unroll
        for (int j = 0; j < DPT; ++j) {
            tot[j] = 0;
#pragma unroll
            for (int w = 0; w < SORT_WAVES; ++w) tot[j] += wcnt[w][threadIdx.x * DPT + j];
            mine += tot[j];
            gmine += gtot[j];
        }
        // exclusive scans over the digits of (this block's counts, the global digit totals): inclusive wave scan + wave offsets
        // (hist_scanned rows are per-digit exclusive scans over the blocks — radix_digit_scan_kernel — so the first global
        //  position of digit d = sum of the totals of the digits below it + this block's entry of d's row)
        uint32_t inc = mine, ginc = gmine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(inc, o, 64), gup = __shfl_up(ginc, o, 64);
            if (lane >= o) { inc += up; ginc += gup; }
        }
        if (lane == 63) { wsum[wave] = inc; gsum[wave] = ginc; }
        __syncthreads();
        uint32_t woff = 0, goff = 0;
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) { woff += (w < wave) ? wsum[w] : 0u; goff += (w < wave) ? gsum[w] : 0u; }
        uint32_t run = woff + inc - mine, grun = goff + ginc - gmine;
#pragma unroll
        for (int j = 0; j < DPT; ++j) {
            const int d = threadIdx.x * DPT + j;
            gdelta[d] = grun + hrow[j] - run;
            grun += gtot[j];
#pragma unroll
            for (int w = 0; w < SORT_WAVES; ++w) {
                const uint32_t c = wcnt[w][d];
                wcnt[w][d] = (uint16_t)run;    // LDS slot of wave w's first item with digit d
                run += c;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const int64_t idx = wbase + (int64_t)r * 64 + lane;
        if (idx < n) {
            const uint32_t d = (key[r] >> shift) & digit_mask;
            const uint32_t slot = (uint32_t)wcnt[wave][d] + rank[r];
            skey[slot] = key[r];
            sval[slot] = val[r];
        }
    }
    __syncthreads();
    const int64_t tile_base = tile * SORT_TILE;
    const int count = (int)((n - tile_base) < (int64_t)SORT_TILE ? (n - tile_base) : (int64_t)SORT_TILE);
    for (int j = threadIdx.x; j < count; j += SORT_THREADS) {
        const uint32_t k = skey[j];
        const uint32_t d = (k >> shift) & digit_mask;
        const uint32_t pos = gdelta[d] + (uint32_t)j;
        keys_out[pos] = k;
        vals_out[pos] = sval[j];
    }
}

int cgs_launch_iota(int64_t n, uint32_t *out, hipStream_t stream);      // raster_geom.hip

static int64_t sort_blocks(int64_t n) { return (n + SORT_TILE - 1) / SORT_TILE; }
#define SORT_MAX_RADIX 512

extern "C" size_t cgs_sort_scratch_bytes(int64_t n) {
    int64_t nb = sort_blocks(n > 0 ? n : 1);
    size_t hist = cgs_align_up((size_t)SORT_MAX_RADIX * nb * sizeof(uint32_t), 256);
    return hist + 256 + SORT_MAX_RADIX * sizeof(uint32_t);
}

// (A one-sweep variant — global digit counts of all passes from one launch, per-(block, digit) status words chained by a
//  decoupled look-back inside the scatter kernel: 6 launches instead of 16 — was built and measured in round 4: 315 us
//  against 262 us for this three-kernel pass structure at 5.8 M keys.  ~1000 co-resident blocks publish their counts at
//  the same moment and every thread then walks hundreds of predecessors' words one L2 round trip at a time;
//  profiles/r04_onesweep_sort.txt.)
// (the tile binning's passes use the same one-launch column scan: tile_bin.hip)
int cgs_launch_digit_scan(uint32_t *hist, uint32_t *totals, int digits, int64_t ncols, hipStream_t stream) {
    hipLaunchKernelGGL(radix_digit_scan_kernel, dim3((unsigned)digits), dim3(SORT_THREADS), 0, stream, hist, totals, ncols);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

template <int BITS, bool XF>
static int sort_pass(const uint32_t *src_k, const uint32_t *src_v, uint32_t *dst_k, uint32_t *dst_v, uint32_t *hist,
                     uint32_t *totals, int64_t n, int64_t nb, int shift, uint32_t mask, uint32_t *overflow, uint32_t epoch,
                     hipStream_t stream) {
    hipLaunchKernelGGL((radix_hist_kernel<BITS, XF>), dim3(sort_grid(nb)), dim3(SORT_THREADS), 0, stream, src_k, hist, n, nb, shift,
                       mask, overflow, epoch);
    CGS_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(radix_digit_scan_kernel, dim3(1u << BITS), dim3(SORT_THREADS), 0, stream, hist, totals, nb);
    CGS_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL((radix_scatter_kernel<BITS, XF>), dim3(sort_grid(nb)), dim3(SORT_THREADS), 0, stream, src_k, src_v, dst_k,
                       dst_v, (const uint32_t *)hist, (const uint32_t *)totals, n, nb, shift, mask);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

// depth27: keys_in are depth keys, sorted as cgs_depth_key27() in three 9-bit passes (keys_out then holds the 27-bit keys)
static int sort_classic(const uint32_t *keys_in, const uint32_t *vals_in, uint32_t *keys_out, uint32_t *vals_out,
                        uint32_t *keys_tmp, uint32_t *vals_tmp, int64_t n, int bit_lo, int bit_hi, int passes, void *scratch,
                        size_t scratch_bytes, hipStream_t stream, bool depth27 = false, uint32_t *overflow = nullptr,
                        uint32_t epoch = 0) {
    (void)scratch_bytes;
    const int64_t nb = sort_blocks(n);
    const int bits_per = depth27 ? 9 : RADIX_BITS;
    const size_t hist_bytes = cgs_align_up((size_t)(1u << bits_per) * nb * sizeof(uint32_t), 256);
    uint32_t *hist = (uint32_t *)scratch;
    uint32_t *totals = (uint32_t *)((char *)scratch + hist_bytes);      // [digits] totals of the pass

    // Choose the ping-pong start so that the last pass lands in *_out.
    const uint32_t *src_k = keys_in, *src_v = vals_in;
    uint32_t *dst_k = (passes & 1) ? keys_out : keys_tmp;
    uint32_t *dst_v = (passes & 1) ? vals_out : vals_tmp;
    for (int p = 0; p < passes; ++p) {
        const int shift = bit_lo + p * bits_per;
        const int nb_bits = (bit_hi - shift) < bits_per ? (bit_hi - shift) : bits_per;
        const uint32_t mask = (1u << nb_bits) - 1u;
        int rc;
        if (!depth27) rc = sort_pass<RADIX_BITS, false>(src_k, src_v, dst_k, dst_v, hist, totals, n, nb, shift, mask, nullptr, 0, stream);
        else if (p == 0) rc = sort_pass<9, true>(src_k, src_v, dst_k, dst_v, hist, totals, n, nb, shift, mask, overflow, epoch, stream);
        else rc = sort_pass<9, false>(src_k, src_v, dst_k, dst_v, hist, totals, n, nb, shift, mask, nullptr, 0, stream);
        if (rc) return rc;
        src_k = dst_k;
        src_v = dst_v;
        dst_k = (dst_k == keys_out) ? keys_tmp : keys_out;
        dst_v = (dst_v == vals_out) ? vals_tmp : vals_out;
    }
    return CGS_OK;
}

extern "C" int cgs_sort_pairs_u32(const uint32_t *keys_in, const uint32_t *vals_in, uint32_t *keys_out,
                                  uint32_t *vals_out, uint32_t *keys_tmp, uint32_t *vals_tmp, int64_t n,
                                  int bit_lo, int bit_hi, void *scratch, size_t scratch_bytes,
                                  void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0 || bit_lo < 0 || bit_hi > 32 || bit_hi < bit_lo) {
        cgs_set_error("sort: bad arguments");
        return CGS_ERR_ARG;
    }
    if (n >= (1ll << 32)) { cgs_set_error("sort: n must fit in uint32"); return CGS_ERR_ARG; }
    const int bits = bit_hi - bit_lo;
    const int passes = (bits + RADIX_BITS - 1) / RADIX_BITS;
    if (n == 0) return CGS_OK;
    if (passes == 0) {
        if (keys_out != keys_in)
            CGS_CHECK_HIP(hipMemcpyAsync(keys_out, keys_in, n * 4, hipMemcpyDeviceToDevice, stream));
        if (vals_in == nullptr) {
            int rc = cgs_launch_iota(n, vals_out, stream);
            if (rc) return rc;
        } else if (vals_out != vals_in) {
            CGS_CHECK_HIP(hipMemcpyAsync(vals_out, vals_in, n * 4, hipMemcpyDeviceToDevice, stream));
        }
        return CGS_OK;
    }
    if (scratch_bytes < cgs_sort_scratch_bytes(n)) {
        cgs_set_error("sort: scratch too small (%zu < %zu)", scratch_bytes, cgs_sort_scratch_bytes(n));
        return CGS_ERR_WORKSPACE;
    }
    return sort_classic(keys_in, vals_in, keys_out, vals_out, keys_tmp, vals_tmp, n, bit_lo, bit_hi, passes, scratch,
                        scratch_bytes, stream);
}

// Stable sort of DEPTH keys (float bits of view depths above the 0.2 near plane; 0xFFFFFFFF = culled, never read back) with
// values = positions: the order of cgs_sort_pairs_u32(keys, NULL, ..., 0, 32) for every live key whenever no live depth
// reaches ~13107 (bits(z) - bits(0.2) < 2^27 - 1) — three 9-bit passes over 27-bit keys instead of four 8-bit passes.
// A live key outside the range makes the first pass write `epoch` to *overflow (device memory, otherwise untouched): the
// order is then NOT valid and the caller sorts again with cgs_sort_pairs_u32.  keys_out receives the 27-bit keys.
extern "C" int cgs_sort_depth_keys(const uint32_t *keys_in, uint32_t *keys_out, uint32_t *vals_out, uint32_t *keys_tmp,
                                   uint32_t *vals_tmp, int64_t n, void *scratch, size_t scratch_bytes, uint32_t *overflow,
                                   uint32_t epoch, void *stream_) {
    if (n < 0 || n >= (1ll << 32) || !overflow) { cgs_set_error("sort_depth_keys: bad arguments"); return CGS_ERR_ARG; }
    if (n == 0) return CGS_OK;
    if (scratch_bytes < cgs_sort_scratch_bytes(n)) {
        cgs_set_error("sort: scratch too small (%zu < %zu)", scratch_bytes, cgs_sort_scratch_bytes(n));
        return CGS_ERR_WORKSPACE;
    }
    return sort_classic(keys_in, nullptr, keys_out, vals_out, keys_tmp, vals_tmp, n, 0, 27, 3, scratch, scratch_bytes,
                        (hipStream_t)stream_, true, overflow, epoch);
}

// ----------------------------------------------------------------------------
// Indices of the non-zero bytes of a mask, ascending (torch.nonzero of a bool [n] — the visible-anchor list of
// gaussian_renderer/__init__.py:44-50), in two halves: _launch enqueues flags -> scan -> scatter and the 4-byte copy of
// the count behind them, _wait blocks on that copy's event only.  torch.nonzero drains the stream to size its result;
// here the kernels the caller enqueues between the halves (the context model's accessors, the step's bookkeeping) keep
// the device busy while the host learns the count.  idx_out has room for n entries; the first *count are valid.
// ----------------------------------------------------------------------------
// (round 6: the scan runs over the mask's bytes themselves — no uint32 flag array, no launch to make one)
__global__ void __launch_bounds__(256)
    nz_scatter_kernel(const uint8_t *__restrict__ f, const uint32_t *__restrict__ pos, int64_t n, int64_t *__restrict__ idx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && f[i]) idx[pos[i]] = i;
}

struct NzSlot { uint32_t *pinned; hipEvent_t ev; bool pending; uint64_t ticket; };
static thread_local NzSlot g_nz_slot = {nullptr, nullptr, false, 0};

extern "C" size_t cgs_nonzero_scratch_bytes(int64_t n) {
    if (n < 1) n = 1;
    return 2 * cgs_align_up((size_t)n * 4, 256) + cgs_scan_scratch_bytes(n) + 512;
}

extern "C" int cgs_nonzero_launch(const uint8_t *mask, int64_t n, int64_t *idx_out, void *scratch, size_t scratch_bytes,
                                  void *stream_, uint64_t *ticket) {
    hipStream_t stream = (hipStream_t)stream_;
    NzSlot &sl = g_nz_slot;
    if (!ticket) { cgs_set_error("nonzero_launch: NULL ticket"); return CGS_ERR_ARG; }
    *ticket = 0;
    if (n < 0 || n >= (1ll << 31)) { cgs_set_error("nonzero: bad n"); return CGS_ERR_ARG; }
    if (!sl.pinned) {
        CGS_CHECK_HIP(hipHostMalloc((void **)&sl.pinned, 64, hipHostMallocDefault));
        CGS_CHECK_HIP(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
    }
    sl.pending = false;
    sl.pinned[0] = 0;
    *ticket = sl.ticket = cgs_new_ticket(3);
    if (n == 0) return CGS_OK;
    if (!mask || !idx_out || !scratch) { cgs_set_error("nonzero: NULL"); return CGS_ERR_ARG; }
    if (scratch_bytes < cgs_nonzero_scratch_bytes(n)) { cgs_set_error("nonzero: scratch too small"); return CGS_ERR_WORKSPACE; }
    CgsCarver cv(scratch, scratch_bytes);
    uint32_t *f = cv.take<uint32_t>(n), *pos = cv.take<uint32_t>(n);
    (void)f;      // (the uint32 flag array of rounds 1-5: still carved so that cgs_nonzero_scratch_bytes keeps its meaning)
    const size_t scan_bytes = cgs_scan_scratch_bytes(n);
    char *scan_scratch = cv.take<char>(scan_bytes + 256);
    uint32_t *total = (uint32_t *)(scan_scratch + scan_bytes);
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    int rc = cgs_scan_exclusive_u8_total(mask, pos, n, scan_scratch, scan_bytes, total, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(nz_scatter_kernel, grid, block, 0, stream, mask, (const uint32_t *)pos, n, idx_out);
    CGS_CHECK_HIP(hipGetLastError());
    CGS_CHECK_HIP(hipMemcpyAsync(sl.pinned, total, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    CGS_CHECK_HIP(hipEventRecord(sl.ev, stream));
    sl.pending = true;
    return CGS_OK;
}

extern "C" int cgs_nonzero_wait(uint64_t ticket, int64_t *count_host) {
    NzSlot &sl = g_nz_slot;
    if (!count_host) { cgs_set_error("nonzero_wait: NULL"); return CGS_ERR_ARG; }
    *count_host = 0;
    if (!sl.pinned) { cgs_set_error("nonzero_wait: no launch on this thread"); return CGS_ERR_ARG; }
    if (ticket == 0 || ticket != sl.ticket) {
        cgs_set_error("nonzero_wait: stale ticket (another nonzero launch was issued on this thread since)");
        return CGS_ERR_ARG;
    }
    if (sl.pending) {
        CGS_CHECK_HIP(hipEventSynchronize(sl.ev));
        sl.pending = false;
    }
    *count_host = sl.pinned[0];
    return CGS_OK;
}
