// Internal helpers shared by the HIP translation units of libcgs_hip.so.
// gfx950 only: wave = 64 lanes, no portability shims.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/cgs.h"

#define CGS_TILE 16          // tile edge in pixels (reference: 16x16 tiles)
// 1 (experiment builds, profiles/r04_blend_bwd_global_acc.txt): the blend backward adds every block visit's nine terms straight
// to the gradient arrays (no-return global atomics) and hands preprocess_bwd RAW sums in dL_dmean2D_px / dL_dconic
// (raster_blend_rows.hip, blend_bwd_rows_ga_kernel) — parity-green, 2.5 x slower (global float atomics retire at ~80 G/s
// chip-wide); 0 (product): the LDS-accumulated kernel with the per-Gaussian factors applied at its flush
#ifndef CGS_BLEND_BWD_RAW
#define CGS_BLEND_BWD_RAW 0
#endif
#define CGS_WAVE 64

void cgs_set_error(const char *fmt, ...);
uint64_t cgs_new_ticket(int kind);      // api.hip: tickets of the *_launch / *_wait pairs (kind: 1 raster, 2 expand, 3 nonzero)

#define CGS_CHECK_HIP(expr)                                                   \
    do {                                                                      \
        hipError_t e__ = (expr);                                              \
        if (e__ != hipSuccess) {                                              \
            cgs_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,       \
                          hipGetErrorString(e__));                            \
            return CGS_ERR_HIP;                                               \
        }                                                                     \
    } while (0)

// After a launch: always pick up launch-configuration errors; in debug mode
// also synchronise so that the failing kernel is the one reported.
#define CGS_CHECK_LAUNCH(stream, debug)                                       \
    do {                                                                      \
        CGS_CHECK_HIP(hipGetLastError());                                     \
        if (debug) CGS_CHECK_HIP(hipStreamSynchronize(stream));               \
    } while (0)

static inline size_t cgs_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- optional per-kernel timing (bench.py's roofline leg) -------------------
// HIP events recorded on the launch stream around a kernel (or a short fixed
// sequence of kernels); off by default, zero cost when off.
enum CgsProfId {
    CGS_PROF_FILTER = 0, CGS_PROF_PREPROCESS, CGS_PROF_DEPTH_SORT, CGS_PROF_OFFSETS_SCAN, CGS_PROF_EMIT_PAIRS,
    CGS_PROF_TILE_SORT, CGS_PROF_RANGES, CGS_PROF_BLEND_FWD, CGS_PROF_BLEND_BWD, CGS_PROF_PREPROCESS_BWD,
    CGS_PROF_EXPAND_FWD, CGS_PROF_EXPAND_BWD, CGS_PROF_RATE_FWD, CGS_PROF_RATE_BWD, CGS_PROF_MLP_FWD,
    CGS_PROF_MLP_BWD, CGS_PROF_MLP_WGRAD, CGS_PROF_CTX_FWD, CGS_PROF_CTX_BWD, CGS_PROF_LOSS_FWD, CGS_PROF_LOSS_BWD,
    CGS_PROF_LMLP_FWD, CGS_PROF_LMLP_BWD, CGS_PROF_LMLP_WGRAD,      // the level MLPs of the context model (cgs_mlp2_*, cgs_ctx_level_bwd's reduction)
    CGS_PROF_COUNT
};
extern int g_cgs_prof_on;
void cgs_prof_begin(int id, hipStream_t stream);
void cgs_prof_end(int id, hipStream_t stream);
struct CgsProfScope {
    int id; hipStream_t s;
    CgsProfScope(int id_, hipStream_t s_) : id(id_), s(s_) { if (g_cgs_prof_on) cgs_prof_begin(id, s); }
    ~CgsProfScope() { if (g_cgs_prof_on) cgs_prof_end(id, s); }
};

// Bump allocator over a caller-owned workspace.
struct CgsCarver {
    char *base;
    size_t off;
    size_t cap;
    bool ok;
    CgsCarver(void *p, size_t n) : base((char *)p), off(0), cap(n), ok(true) {}
    template <typename T> T *take(size_t count) {
        off = cgs_align_up(off, 256);
        size_t bytes = count * sizeof(T);
        T *r = (T *)(base ? base + off : nullptr);
        off += bytes;
        if (base && off > cap) ok = false;
        return r;
    }
    size_t used() const { return cgs_align_up(off, 256); }
};

// ---- geometry workspace layout (per-Gaussian state of one view) ----------
// rec: 3 x float4 per Gaussian, AoS so that one gather touches one 48-byte
// span:  r0 = (px, py, A, B)   A,B,C = conic pre-scaled for exp2:
//        r1 = (C, opacity, red, green)   power2 = A dx^2 + C dy^2 + B dx dy
//        r2 = (blue, rcut2, conic_a_raw?, unused) -- see raster_preprocess.hip
struct CgsGeom {
    float4 *rec;            // [3P]
    uint32_t *depth_key;    // [P] float bits of view depth (positive => order-preserving)
    uint32_t *tiles;        // [P] tiles touched (0 => culled)
    uint2 *rect;            // [P] packed tile rect: .x = x0 | y0<<16, .y = x1 | y1<<16 (exclusive max)
    uint32_t *order;        // [P] Gaussian ids in ascending (depth, id) order
    uint32_t *offsets;      // [P] exclusive scan of tiles[order[i]]
    uint32_t *sort_a;       // [P] ping-pong scratch (keys)
    uint32_t *sort_b;       // [P]
    uint32_t *sort_c;       // [P] bucket-pair offsets of the two-level binning (tile_bin.hip)
    uint32_t *sort_d;       // [P]
    uint32_t *total;        // [2] device: num_rendered
    void *scratch;          // scan/sort scratch
    size_t scratch_bytes;
};

struct CgsBin {
    uint32_t *tile_key_a;   // [R]
    uint32_t *tile_key_b;   // [R]
    uint32_t *gid_a;        // [R]
    uint32_t *gid_b;        // [R]
    uint32_t *tile_key_c;   // [R] sorted keys
    uint32_t *gid_sorted;   // [R] final per-tile lists (depth order inside each tile)
    void *scratch;
    size_t scratch_bytes;
    // two-level binning (tile_bin.hip): bucket / chunk tables, per-(tile, chunk) counts and their scan
    uint32_t *bk_tab;       // [cgs_bucket_tab_words()]
    uint32_t *bk_counts;    // [cgs_bucket_count_slots(R)]
    uint32_t *bk_scan;      // [cgs_bucket_count_slots(R)]
    void *bk_scan_scratch;
    size_t bk_scan_scratch_bytes;
};

struct CgsImg {
    uint2 *ranges;          // [tiles] (start, end) into gid_sorted
    float *final_T;         // [H*W]
    uint32_t *n_contrib;    // [H*W] 1-based position (within the tile list) of the last contributor
    uint32_t *tile_last;    // [tiles] max over the tile's pixels of n_contrib
    uint32_t *tile_order;   // [tiles] tile ids, longest lists first (the blend kernels' workgroup -> tile map)
};

size_t cgs_geom_carve(CgsGeom *g, void *ws, size_t bytes, int64_t P);
size_t cgs_bin_carve(CgsBin *b, void *ws, size_t bytes, int64_t P, int64_t R);
size_t cgs_img_carve(CgsImg *im, void *ws, size_t bytes, int32_t H, int32_t W);

// internal launchers (raster_*.hip)
int cgs_launch_preprocess(const cgs_raster_cfg *cfg, int64_t P, const float *means3D,
                          const float *colors, const float *opacities, const float *scales,
                          const float *rotations, CgsGeom &g, int32_t *radii, bool filter_only,
                          hipStream_t stream);
int cgs_launch_emit_pairs(const cgs_raster_cfg *cfg, int64_t P, CgsGeom &g, CgsBin &b,
                          hipStream_t stream);
int cgs_launch_ranges(const cgs_raster_cfg *cfg, int64_t R, CgsBin &b, CgsImg &im,
                      hipStream_t stream);
// tile_bin.hip: binning with a pair-generating first radix pass and 16-bit tile keys (grids of <= 65536 tiles)
int cgs_launch_gather_rects(int64_t P, CgsGeom &g, hipStream_t stream);
bool cgs_tile_bin16_ok(int tile_bits);
bool cgs_tile_bin_buckets_ok(const cgs_raster_cfg *cfg);
bool cgs_tile_bin_buckets_fits(int64_t R);
int64_t cgs_bucket_count_slots(int64_t R);
size_t cgs_bucket_tab_words(void);
int cgs_launch_tile_bin_buckets(const cgs_raster_cfg *cfg, int64_t P, int64_t R, CgsGeom &g, CgsBin &b, CgsImg &im,
                                hipStream_t stream, const uint32_t *R_dev);
int cgs_launch_tile_bin16(const cgs_raster_cfg *cfg, int64_t P, int64_t R, int tile_bits, CgsGeom &g, CgsBin &b, CgsImg &im,
                          hipStream_t stream, const uint32_t *R_dev = nullptr);     // per-tile lists AND im.ranges
// raster_blend_rows.hip: row-mapped variants (four 4x4 blocks per wave), selected by cgs_blend_rows_enabled()
int cgs_launch_blend_fwd_rows(const cgs_raster_cfg *cfg, CgsGeom &g, CgsBin &b, CgsImg &im, float *out_color,
                              hipStream_t stream);
int cgs_launch_blend_bwd_rows(const cgs_raster_cfg *cfg, CgsGeom &g, CgsBin &b, CgsImg &im, const float *dL_dout,
                              float *dL_dmean2D_px, float *dL_dconic, float *dL_dopacity, float *dL_dcolors,
                              hipStream_t stream);
int cgs_launch_tile_order(const cgs_raster_cfg *cfg, CgsImg &im, hipStream_t stream);      // raster_blend_rows.hip
int cgs_launch_blend_fwd(const cgs_raster_cfg *cfg, CgsGeom &g, CgsBin &b, CgsImg &im,
                         float *out_color, hipStream_t stream);
int cgs_launch_blend_bwd(const cgs_raster_cfg *cfg, CgsGeom &g, CgsBin &b, CgsImg &im,
                         const float *dL_dout, float *dL_dmean2D_px, float *dL_dconic,
                         float *dL_dopacity, float *dL_dcolors, hipStream_t stream);
int cgs_launch_preprocess_bwd(const cgs_raster_cfg *cfg, int64_t P, const float4 *rec_raw, const float *means3D,
                              const float *scales, const float *rotations,
                              const int32_t *radii, const float *dL_dmean2D_px,
                              const float *dL_dconic, float *dL_dmeans3D, float *dL_dmeans2D,
                              float *dL_dscales, float *dL_drotations, hipStream_t stream);

// expand_raster.hip: the anchor expansion fused with the rasterizer's preprocess stage
struct CgsExpandSrc {
    int64_t n_anchor;
    int K;
    const uint8_t *flags;      // survivor flags, one byte per slot (cgs_expand_count's mask_out)
    const uint32_t *pos;
    const float *anchor, *gscaling, *offsets, *neural_opacity, *color_in, *cov_in;
    const int64_t *src_row;
};
int cgs_launch_expand_preprocess(const cgs_raster_cfg *cfg, const CgsExpandSrc &x, float *scaling_out, float *xyz_out,
                                 float *rot_out, CgsGeom &g, int32_t *radii, hipStream_t stream);

// XCD-aware work assignment for kernels whose NEIGHBOURING work items write adjacent memory (radix passes: a tile's run for a
// digit starts where its predecessor's ends).  Workgroups are dealt to the eight XCDs round robin (workgroup i -> XCD i % 8,
// each with its own L2): item = workgroup id puts neighbours on different XCDs, every short run then leaves its L2 as a partial
// line of its own.  Here XCD x takes the x-th eighth of the items, so neighbours meet in one L2.  Launch cgs_xcd_grid(n)
// workgroups; cgs_xcd_item(n) is the workgroup's item or -1 for a padding workgroup.
#ifdef __HIPCC__
__device__ __forceinline__ int64_t cgs_xcd_item(int64_t n_items) {
    const int64_t chunk = (n_items + 7) / 8;
    const int64_t t = (int64_t)(blockIdx.x & 7u) * chunk + (int64_t)(blockIdx.x >> 3);
    return ((int64_t)(blockIdx.x >> 3) < chunk && t < n_items) ? t : -1;
}
#endif
static inline unsigned cgs_xcd_grid(int64_t n_items) { return (unsigned)(8 * ((n_items + 7) / 8)); }
// Arrival ticket of a launch whose LAST workgroup finishes the job (adds up the per-workgroup partial sums in workgroup order).
// The textbook form — store the partial, __threadfence(), atomicAdd on a counter — costs ~25 ns PER WORKGROUP, serial, on this
// chip: the device-scope release of the fence (eight L2s), not the atomic (profiles/r06_same_address_atomics.txt: eb_bits_fwd
// 42.7 us with it, 32.0 us without, at 504 workgroups; 90 -> 36 us at 2040).  Here the partial is PUBLISHED BY AN ATOMIC EXCHANGE at
// device scope (performed at the coherence point; its return value is waited for, so it is there before the ticket is taken) and the
// ticket is a relaxed atomic: no fence in any workgroup but the last, which fences once (acquire) and reads the partials with
// device-scope atomic loads.  Two levels (workgroup b counts in slot b % CGS_TICKET_SLOTS, a 128-byte line each; the last arrival
// of a slot counts at the root) keep the same-address atomics short.  t: CGS_TICKET_BYTES bytes, zero before the first launch (the
// last workgroup leaves them zero).  Call from ONE thread per workgroup; true for exactly one workgroup of the launch.
#define CGS_TICKET_SLOTS 8
#define CGS_TICKET_BYTES ((CGS_TICKET_SLOTS + 1) * 128)
#ifdef __HIPCC__
__device__ __forceinline__ void cgs_publish(double *slot, double v) {
    const unsigned long long old = atomicExch((unsigned long long *)slot, (unsigned long long)__double_as_longlong(v));
    asm volatile("" ::"v"(old) : "memory");          // the exchange has returned: performed, before anything below is issued
}
__device__ __forceinline__ double cgs_published(const double *slot) {
    return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ bool cgs_ticket_last(unsigned int *t) {
    const unsigned nb = gridDim.x, S = CGS_TICKET_SLOTS, slot = blockIdx.x % S;
    const unsigned in_slot = (nb - slot + S - 1) / S;
    if (atomicAdd(&t[32 * (1 + slot)], 1u) != in_slot - 1u) return false;
    t[32 * (1 + slot)] = 0u;
    if (atomicAdd(&t[0], 1u) != (nb < S ? nb : S) - 1u) return false;
    t[0] = 0u;
    __threadfence();
    return true;
}
#endif
// prims.hip: per-digit exclusive scans over the columns of a [digits][ncols] table in place + the digits' totals (one launch)
int cgs_launch_digit_scan(uint32_t *hist, uint32_t *totals, int digits, int64_t ncols, hipStream_t stream);

static inline int cgs_tiles_x(const cgs_raster_cfg *c) { return (c->image_width + CGS_TILE - 1) / CGS_TILE; }
static inline int cgs_tiles_y(const cgs_raster_cfg *c) { return (c->image_height + CGS_TILE - 1) / CGS_TILE; }

// mlp_wgrad.hip
int cgs_launch_wgrad2(const float *P, int64_t ldp, int DA, const float *Q, int64_t ldq, int DB, float *dW, float *db,
                      int64_t n, int num_cus, void *scratch, size_t scratch_bytes, hipStream_t s);
size_t cgs_wgrad_scratch_bytes_for(int num_cus);
// mlp_small.hip: -1 = no instance
int cgs_launch_mlp2_bwd_recompute(int in, int hid, int out, const float *X, int64_t ldx, const float *W1,
                                  const float *b1, const float *W2, const float *dY, int64_t ldy, float *dX,
                                  int64_t lddx, int acc, float *dZ1, float *dW2, float *db2, int64_t n, int num_cus,
                                  void *scratch, size_t scratch_bytes, hipStream_t s);
struct CgsWgProduct { const float *P; int64_t ldp; int DA; const float *Q; int64_t ldq; int DB; float *dW; float *db; };
int cgs_launch_wgrad_multi(const CgsWgProduct *prods, int nprod, int64_t n, int num_cus, void *scratch,
                           size_t scratch_bytes, hipStream_t s);
int cgs_launch_wgrad_reduce(const float *partial, int blocks, const CgsWgProduct *prods, int nprod, hipStream_t s);
int cgs_launch_wgrad_reduce_assign(const float *partial, int blocks, const CgsWgProduct *prods, int nprod, hipStream_t s);
