// Per-step bookkeeping of the training context model as two launches (scene/gaussian_model.py:1658-1661 and the
// plan cache of context_model.py): which anchors enter the rate estimate, where they sit in coding order, and
// whether the cached level plan is still valid.
//
// The torch composition this replaces was ~35 launches per step: rand_like, <=, &, two (==).all() reductions,
// index_select by the permutation, cumsum, cat, index_select of the level bounds, one read-back, nonzero_static
// (four kernels), three offset subtractions and a gather — all over N-element vectors, all bound by launch latency.
//
//   choose_flags    r in coding order, a = perm[r]:  flag[r] = (u(seed, a) <= thresh  or  given[a]) and mask[a];
//                   per-4096-row block counts; per-level counts, the number of live anchors and the plan-validity
//                   flag (anchor / mask still equal to the copies the plan was built from) into `meta`.
//   choose_compact  ordered compaction of the flagged rows: coding-order position, original index and level-local
//                   position of every chosen anchor (the block bases are the prefix of the block counts, summed
//                   by the block itself: at most N / 4096 values); optionally the inverse, for every row of a level its
//                   index in the level's chosen list or -1 (what the rate backward scatters through).
// The host reads `meta` (ONE synchronisation, the sizes of the level subsets) between the two.
#include "cgs_internal.h"

#define CP_THREADS 256
#define CP_PER_THREAD 4           // (16 was tried in round 6 for fewer closing atomics: flags 36 -> 41 us, compact 11 -> 25 us: worse)
#define CP_CHUNK (CP_THREADS * CP_PER_THREAD)
#define CP_MAX_LEVELS 8
#define CP_SLOT_INTS 32           // ints between two slots of the spread `meta` (cgs_ctx_choose_flags_slots): a 128-byte line each

namespace {

__device__ __forceinline__ uint32_t cp_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

struct CpBounds { int64_t b[CP_MAX_LEVELS + 1]; int n; };

__device__ __forceinline__ int cp_level(const CpBounds &B, int64_t r) {
    int l = 0;
#pragma unroll
    for (int k = 1; k < CP_MAX_LEVELS; ++k)
        if (k < B.n && r >= B.b[k]) l = k;
    return l;
}

}  // namespace

__global__ void __launch_bounds__(CP_THREADS)
    ctx_choose_flags_kernel(const int64_t *__restrict__ perm, int64_t n, const uint8_t *__restrict__ mask,
                            const uint8_t *__restrict__ given, uint64_t seed, float thresh,
                            const float *__restrict__ anchor, const float *__restrict__ anchor_ref,
                            const uint8_t *__restrict__ mask_ref, CpBounds B, uint8_t *__restrict__ flags,
                            uint32_t *__restrict__ block_counts, int32_t *__restrict__ meta_all, int nslots) {
    // (a block's closing atomics go to slot blockIdx.x % nslots, CP_SLOT_INTS ints = one 128-byte line per slot: ~1000 blocks adding
    //  to ONE live counter serialise on that address — most of the kernel's 34 us with a single slot)
    int32_t *__restrict__ meta = meta_all + (size_t)(blockIdx.x % (unsigned)nslots) * CP_SLOT_INTS;
    // per-thread tallies, reduced once per block: [0] chosen, [1] live, [2] stale, [3 + l] chosen of level l
    __shared__ uint32_t tally[3 + CP_MAX_LEVELS];
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t key = cp_mix32((uint32_t)seed ^ 0x9E3779B9u) ^ (uint32_t)(seed >> 32);
    const int64_t base = (int64_t)blockIdx.x * CP_CHUNK;
    if (tid < 3 + CP_MAX_LEVELS) tally[tid] = 0;
    __syncthreads();
    uint32_t mine = 0, live_n = 0, stale_n = 0, lvl_n[CP_MAX_LEVELS];
#pragma unroll
    for (int l = 0; l < CP_MAX_LEVELS; ++l) lvl_n[l] = 0;
    // every load of the thread's four rows is requested before the first is used (the dependent pair perm[r] -> mask[perm[r]] was one
    // exposed round trip per row with the rows handled one after the other: 36 us for 35 MB)
    int64_t rr[CP_PER_THREAD], aa[CP_PER_THREAD];
    bool inb[CP_PER_THREAD];
#pragma unroll
    for (int k = 0; k < CP_PER_THREAD; ++k) {
        rr[k] = base + (int64_t)k * CP_THREADS + tid;
        inb[k] = rr[k] < n;
        aa[k] = inb[k] ? (perm ? perm[rr[k]] : rr[k]) : 0;
    }
    float an[CP_PER_THREAD][3], ar[CP_PER_THREAD][3];
    uint8_t mr[CP_PER_THREAD], mrr[CP_PER_THREAD];
#pragma unroll
    for (int k = 0; k < CP_PER_THREAD; ++k) {
        const int64_t r = inb[k] ? rr[k] : 0;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            an[k][j] = anchor_ref ? anchor[3 * r + j] : 0.f;
            ar[k][j] = anchor_ref ? anchor_ref[3 * r + j] : 0.f;
        }
        mr[k] = mask ? mask[r] : (uint8_t)1;
        mrr[k] = mask_ref ? mask_ref[r] : (uint8_t)0;
    }
    uint8_t ma[CP_PER_THREAD], ga[CP_PER_THREAD];
#pragma unroll
    for (int k = 0; k < CP_PER_THREAD; ++k) {
        ma[k] = mask ? mask[aa[k]] : (uint8_t)1;
        ga[k] = given ? given[aa[k]] : (uint8_t)0;
    }
#pragma unroll
    for (int k = 0; k < CP_PER_THREAD; ++k) {
        if (!inb[k]) continue;
        const int64_t r = rr[k], a = aa[k];
        const bool live = ma[k] != 0;
        bool f;
        if (given) f = ga[k] != 0;
        else f = (float)(cp_mix32((uint32_t)a + key + (uint32_t)((uint64_t)a >> 32) * 0x632BE5ABu) >> 8) * (1.0f / 16777216.0f) <= thresh;
        f = f && live;
        flags[r] = f ? 1 : 0;
        // the staleness test and the live count are sums over ALL anchors: taken over anchor r instead of perm[r], so
        // that the two [n,3] tensors and the reference mask stream in order instead of being gathered through perm
        bool stale = false;
        if (anchor_ref) stale = an[k][0] != ar[k][0] || an[k][1] != ar[k][1] || an[k][2] != ar[k][2];
        const bool live_r = mr[k] != 0;
        if (mask_ref) stale = stale || (live_r != (mrr[k] != 0));
        mine += f ? 1u : 0u;
        live_n += live_r ? 1u : 0u;
        stale_n += stale ? 1u : 0u;
        const int lvl = cp_level(B, r);
#pragma unroll
        for (int l = 0; l < CP_MAX_LEVELS; ++l) lvl_n[l] += (f && lvl == l) ? 1u : 0u;
    }
    auto wave_sum = [](uint32_t v) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d, 64);
        return v;
    };
    mine = wave_sum(mine);
    live_n = wave_sum(live_n);
    stale_n = wave_sum(stale_n);
#pragma unroll
    for (int l = 0; l < CP_MAX_LEVELS; ++l) lvl_n[l] = wave_sum(lvl_n[l]);
    if (lane == 0) {
        atomicAdd(&tally[0], mine);
        atomicAdd(&tally[1], live_n);
        atomicAdd(&tally[2], stale_n);
#pragma unroll
        for (int l = 0; l < CP_MAX_LEVELS; ++l)
            if (l < B.n) atomicAdd(&tally[3 + l], lvl_n[l]);
    }
    __syncthreads();
    if (tid == 0) {
        block_counts[blockIdx.x] = tally[0];
        if (tally[1]) atomicAdd(&meta[1], (int32_t)tally[1]);
        if (tally[2]) atomicOr(&meta[0], 1);
    }
    if (tid >= 1 && tid <= B.n && tally[3 + tid - 1]) atomicAdd(&meta[2 + tid - 1], (int32_t)tally[3 + tid - 1]);
}

__global__ void __launch_bounds__(CP_THREADS)
    ctx_choose_compact_kernel(const uint8_t *__restrict__ flags, const uint32_t *__restrict__ block_counts,
                              const int64_t *__restrict__ perm, int64_t n, CpBounds B, CpBounds CB,
                              int64_t *__restrict__ nz, int64_t *__restrict__ rows, int64_t *__restrict__ loc,
                              int32_t *__restrict__ sub_map) {
    __shared__ uint32_t part[CP_THREADS];
    __shared__ uint32_t wave_cnt[CP_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // base = number of chosen rows in the blocks before this one
    uint32_t s = 0;
    for (int j = tid; j < (int)blockIdx.x; j += CP_THREADS) s += block_counts[j];
    part[tid] = s;
    __syncthreads();
    for (int d = CP_THREADS / 2; d >= 1; d >>= 1) {
        if (tid < d) part[tid] += part[tid + d];
        __syncthreads();
    }
    uint32_t running = part[0];
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * CP_CHUNK;
    const uint64_t lt = (1ull << lane) - 1ull;
    for (int k = 0; k < CP_PER_THREAD; ++k) {
        const int64_t r = base + (int64_t)k * CP_THREADS + tid;
        const bool f = r < n && flags[r] != 0;
        const uint64_t bal = __ballot(f);
        if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t before = (uint32_t)__popcll(bal & lt);
        uint32_t total = 0;
#pragma unroll
        for (int w = 0; w < CP_THREADS / 64; ++w) {
            const uint32_t c = wave_cnt[w];
            before += w < wave ? c : 0u;
            total += c;
        }
        if (f) {
            const uint32_t p = running + before;
            const int lvl = cp_level(B, r);
            nz[p] = r;
            rows[p] = perm ? perm[r] : r;
            loc[p] = r - B.b[lvl];
            if (sub_map) sub_map[r] = (int32_t)((int64_t)p - CB.b[lvl]);
        } else if (sub_map && r < n) {
            sub_map[r] = -1;
        }
        running += total;
        __syncthreads();
    }
}

extern "C" size_t cgs_ctx_choose_blocks(int64_t n) { return (size_t)((n + CP_CHUNK - 1) / CP_CHUNK); }

// meta: int32 [2 + nlevels], ZEROED by this call before the kernel: [0] != 0 <=> anchor / mask differ from the
// reference copies, [1] = number of live (mask != 0) anchors, [2 + l] = chosen rows of level l (coding order).
// bounds_host: int64 [nlevels + 1], coding-order level boundaries (bounds[0] = 0, bounds[nlevels] = n).
// meta_slots: int32 [nslots][32] (cgs_ctx_choose_flags_slots; ZEROED by this call): slot s, entries 0 .. 1 + nlevels = the partial
// `meta` of the blocks s, s + nslots, ...: the caller adds the slots up ([0]: ORs them).  nslots == 1 with a [2 + nlevels] array
// is cgs_ctx_choose_flags.
static int cp_choose_flags(const int64_t *perm, int64_t n, const uint8_t *mask, const uint8_t *given, uint64_t seed, float thresh,
                           const float *anchor, const float *anchor_ref, const uint8_t *mask_ref, const int64_t *bounds_host,
                           int nlevels, uint8_t *flags, uint32_t *block_counts, int32_t *meta, int nslots, void *stream) {
    if (n < 0 || nlevels < 1 || nlevels > CP_MAX_LEVELS || !bounds_host || !flags || !block_counts || !meta || nslots < 1 || nslots > 256) {
        cgs_set_error("ctx_choose_flags: bad args");
        return CGS_ERR_ARG;
    }
    if ((anchor_ref && !anchor) || (mask_ref && !mask)) { cgs_set_error("ctx_choose_flags: reference without live tensor"); return CGS_ERR_ARG; }
    const size_t meta_bytes = nslots == 1 ? (size_t)(2 + nlevels) * sizeof(int32_t) : (size_t)nslots * CP_SLOT_INTS * sizeof(int32_t);
    CGS_CHECK_HIP(hipMemsetAsync(meta, 0, meta_bytes, (hipStream_t)stream));
    if (n == 0) return CGS_OK;
    CpBounds B;
    B.n = nlevels;
    for (int l = 0; l <= nlevels; ++l) B.b[l] = bounds_host[l];
    CgsProfScope prof(CGS_PROF_CTX_FWD, (hipStream_t)stream);
    hipLaunchKernelGGL(ctx_choose_flags_kernel, dim3((unsigned)cgs_ctx_choose_blocks(n)), dim3(CP_THREADS), 0,
                       (hipStream_t)stream, perm, n, mask, given, seed, thresh, anchor, anchor_ref, mask_ref, B, flags,
                       block_counts, meta, nslots);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

extern "C" int cgs_ctx_choose_flags(const int64_t *perm, int64_t n, const uint8_t *mask, const uint8_t *given,
                                    uint64_t seed, float thresh, const float *anchor, const float *anchor_ref,
                                    const uint8_t *mask_ref, const int64_t *bounds_host, int nlevels, uint8_t *flags,
                                    uint32_t *block_counts, int32_t *meta, void *stream) {
    return cp_choose_flags(perm, n, mask, given, seed, thresh, anchor, anchor_ref, mask_ref, bounds_host, nlevels, flags, block_counts,
                           meta, 1, stream);
}

extern "C" int cgs_ctx_choose_slot_ints(void) { return CP_SLOT_INTS; }

extern "C" int cgs_ctx_choose_flags_slots(const int64_t *perm, int64_t n, const uint8_t *mask, const uint8_t *given,
                                          uint64_t seed, float thresh, const float *anchor, const float *anchor_ref,
                                          const uint8_t *mask_ref, const int64_t *bounds_host, int nlevels, uint8_t *flags,
                                          uint32_t *block_counts, int32_t *meta_slots, int nslots, void *stream) {
    if (nslots < 2) { cgs_set_error("ctx_choose_flags_slots: nslots >= 2 (one slot: cgs_ctx_choose_flags)"); return CGS_ERR_ARG; }
    return cp_choose_flags(perm, n, mask, given, seed, thresh, anchor, anchor_ref, mask_ref, bounds_host, nlevels, flags, block_counts,
                           meta_slots, nslots, stream);
}

// nz / rows / loc: int64 [>= number of chosen rows]; entries appear in coding order.  sub_map (may be NULL): int32 [n],
// row r of level l -> its index in level l's part of the chosen list, -1 when not chosen; needs chosen_counts_host
// (int64 [nlevels], the per-level counts read back from `meta` of cgs_ctx_choose_flags).
extern "C" int cgs_ctx_choose_compact(const uint8_t *flags, const uint32_t *block_counts, const int64_t *perm, int64_t n,
                                      const int64_t *bounds_host, int nlevels, int64_t *nz, int64_t *rows, int64_t *loc,
                                      int32_t *sub_map, const int64_t *chosen_counts_host, void *stream) {
    if (n < 0 || nlevels < 1 || nlevels > CP_MAX_LEVELS || !bounds_host || !flags || !block_counts || !nz || !rows || !loc) {
        cgs_set_error("ctx_choose_compact: bad args");
        return CGS_ERR_ARG;
    }
    if (sub_map && !chosen_counts_host) { cgs_set_error("ctx_choose_compact: sub_map needs the per-level chosen counts"); return CGS_ERR_ARG; }
    if (n == 0) return CGS_OK;
    CpBounds B, CB;
    B.n = CB.n = nlevels;
    for (int l = 0; l <= nlevels; ++l) B.b[l] = bounds_host[l];
    CB.b[0] = 0;
    for (int l = 0; l < nlevels; ++l) CB.b[l + 1] = CB.b[l] + (chosen_counts_host ? chosen_counts_host[l] : 0);
    CgsProfScope prof(CGS_PROF_CTX_FWD, (hipStream_t)stream);
    hipLaunchKernelGGL(ctx_choose_compact_kernel, dim3((unsigned)cgs_ctx_choose_blocks(n)), dim3(CP_THREADS), 0,
                       (hipStream_t)stream, flags, block_counts, perm, n, B, CB, nz, rows, loc, sub_map);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}
