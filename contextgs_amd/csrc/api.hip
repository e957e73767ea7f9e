// extern "C" entry points of libcgs_hip.so (see include/cgs.h) for the
// rasterizer, plus error plumbing and workspace carving.
#include <stdarg.h>
#include <string.h>
#include "cgs_internal.h"

static thread_local char g_err[512] = "";

void cgs_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *cgs_last_error(void) { return g_err; }
extern "C" int cgs_version(void) { return CGS_VERSION; }
#ifndef CGS_SOURCE_DIGEST
#define CGS_SOURCE_DIGEST "unknown"
#endif
#ifndef CGS_BUILD_FLAGS
#define CGS_BUILD_FLAGS ""
#endif
extern "C" const char *cgs_build_info(void) { return CGS_SOURCE_DIGEST "|" CGS_BUILD_FLAGS; }

int cgs_scan_exclusive_u32_total(const uint32_t *in, uint32_t *out, int64_t n, void *scratch,
                                 size_t scratch_bytes, uint32_t *grand_total, hipStream_t stream);
int cgs_launch_iota(int64_t n, uint32_t *out, hipStream_t stream);
int cgs_launch_gather_tiles(int64_t P, const uint32_t *order, const uint32_t *tiles, uint32_t *out,
                            hipStream_t stream);
int cgs_launch_stats(const cgs_raster_cfg *cfg, CgsImg &im, int64_t *stats_out, hipStream_t stream);

// ---- workspace carving ---------------------------------------------------------
size_t cgs_geom_carve(CgsGeom *g, void *ws, size_t bytes, int64_t P) {
    CgsCarver c(ws, bytes);
    const size_t n = (size_t)(P > 0 ? P : 1);
    g->rec = c.take<float4>(3 * n);
    g->depth_key = c.take<uint32_t>(n);
    g->tiles = c.take<uint32_t>(n);
    g->rect = c.take<uint2>(n);
    g->order = c.take<uint32_t>(n);
    g->offsets = c.take<uint32_t>(n);
    g->sort_a = c.take<uint32_t>(n);
    g->sort_b = c.take<uint32_t>(n);
    g->sort_c = c.take<uint32_t>(n);
    g->sort_d = c.take<uint32_t>(n);
    g->total = c.take<uint32_t>(4);        // pair count | bucket-pair count | depth-range report of the depth sort | -
    size_t sb = cgs_sort_scratch_bytes((int64_t)n);
    size_t sc = cgs_scan_scratch_bytes((int64_t)n);
    g->scratch_bytes = sb > sc ? sb : sc;
    g->scratch = c.take<char>(g->scratch_bytes);
    return c.ok ? c.used() : 0;
}

size_t cgs_bin_carve(CgsBin *b, void *ws, size_t bytes, int64_t P, int64_t R) {
    (void)P;
    CgsCarver c(ws, bytes);
    const size_t n = (size_t)(R > 0 ? R : 1);
    b->tile_key_a = c.take<uint32_t>(n);
    b->tile_key_b = c.take<uint32_t>(n);
    b->gid_a = c.take<uint32_t>(n);
    b->gid_b = c.take<uint32_t>(n);
    b->tile_key_c = c.take<uint32_t>(n);
    b->gid_sorted = c.take<uint32_t>(n);
    b->scratch_bytes = cgs_sort_scratch_bytes((int64_t)n);
    b->scratch = c.take<char>(b->scratch_bytes);
    const int64_t slots = cgs_bucket_count_slots((int64_t)n);
    b->bk_tab = c.take<uint32_t>(cgs_bucket_tab_words());
    b->bk_counts = c.take<uint32_t>((size_t)slots);
    b->bk_scan = c.take<uint32_t>((size_t)slots);
    b->bk_scan_scratch_bytes = cgs_scan_scratch_bytes(slots);
    b->bk_scan_scratch = c.take<char>(b->bk_scan_scratch_bytes);
    return c.ok ? c.used() : 0;
}

size_t cgs_img_carve(CgsImg *im, void *ws, size_t bytes, int32_t H, int32_t W) {
    CgsCarver c(ws, bytes);
    const size_t tiles = (size_t)((W + CGS_TILE - 1) / CGS_TILE) * ((H + CGS_TILE - 1) / CGS_TILE);
    const size_t hw = (size_t)H * W;
    im->ranges = c.take<uint2>(tiles);
    im->final_T = c.take<float>(hw);
    im->n_contrib = c.take<uint32_t>(hw);
    im->tile_last = c.take<uint32_t>(tiles);
    im->tile_order = c.take<uint32_t>(tiles);
    return c.ok ? c.used() : 0;
}

extern "C" size_t cgs_raster_geom_bytes(int64_t P) {
    CgsGeom g;
    CgsCarver probe(nullptr, 0);
    (void)probe;
    return cgs_geom_carve(&g, nullptr, 0, P);
}
extern "C" size_t cgs_raster_bin_bytes(int64_t P, int64_t R) {
    CgsBin b;
    return cgs_bin_carve(&b, nullptr, 0, P, R);
}
extern "C" size_t cgs_raster_img_bytes(int32_t H, int32_t W) {
    CgsImg im;
    return cgs_img_carve(&im, nullptr, 0, H, W);
}
extern "C" size_t cgs_raster_bwd_scratch_bytes(int64_t P) {
    const size_t n = (size_t)(P > 0 ? P : 1);
    return cgs_align_up(2 * n * sizeof(float), 256) + cgs_align_up(3 * n * sizeof(float), 256);
}

static int check_cfg(const cgs_raster_cfg *cfg) {
    if (!cfg) { cgs_set_error("cfg is NULL"); return CGS_ERR_ARG; }
    if (cfg->image_height <= 0 || cfg->image_width <= 0) {
        cgs_set_error("bad image size %dx%d", cfg->image_width, cfg->image_height);
        return CGS_ERR_ARG;
    }
    if (cfg->image_height > 65535 * CGS_TILE / 16 * 16 || cfg->image_width > 65535 * 16) {
        cgs_set_error("image too large for 16-bit tile coordinates");
        return CGS_ERR_ARG;
    }
    if (!cfg->viewmatrix || !cfg->projmatrix) { cgs_set_error("matrices are NULL"); return CGS_ERR_ARG; }
    return CGS_OK;
}

// ---- visible_filter ----------------------------------------------------------------
extern "C" int cgs_filter(const cgs_raster_cfg *cfg, int64_t N, const float *means3D, const float *scales,
                          const float *rotations, int32_t *radii, void *stream) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (N < 0) { cgs_set_error("N < 0"); return CGS_ERR_ARG; }
    if (N > 0 && (!means3D || !scales || !rotations || !radii)) {
        cgs_set_error("cgs_filter: NULL input");
        return CGS_ERR_ARG;
    }
    CgsGeom g;
    memset(&g, 0, sizeof(g));
    return cgs_launch_preprocess(cfg, N, means3D, nullptr, nullptr, scales, rotations, g, radii, true,
                                 (hipStream_t)stream);
}

int cgs_launch_filter_voxel(const cgs_raster_cfg *cfg, int64_t N, const float *means3D, const float *scaling, int64_t ld,
                            int scales_are_log, const float *rot1, uint8_t *visible, hipStream_t stream);

// prefilter_voxel in one launch: scaling [N, ld] raw rows (columns 0..2 used; exp applied when scales_are_log), rot1 [4]
// the normalised rotation shared by every anchor, visible [N] receives (radii > 0) as bool bytes.
extern "C" int cgs_filter_voxel(const cgs_raster_cfg *cfg, int64_t N, const float *means3D, const float *scaling,
                                int64_t ld, int scales_are_log, const float *rot1, uint8_t *visible, void *stream) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (N < 0 || ld < 3) { cgs_set_error("cgs_filter_voxel: bad N / ld"); return CGS_ERR_ARG; }
    if (N > 0 && (!means3D || !scaling || !rot1 || !visible)) { cgs_set_error("cgs_filter_voxel: NULL input"); return CGS_ERR_ARG; }
    return cgs_launch_filter_voxel(cfg, N, means3D, scaling, ld, scales_are_log, rot1, visible, (hipStream_t)stream);
}

// ---- forward stage 1 -----------------------------------------------------------------
// cgs_raster_preprocess_launch enqueues projection, the depth sort and the pair-offset scan, and the 4-byte copy of the
// pair count behind them; cgs_raster_preprocess_wait blocks on THAT copy's event only.  What the caller enqueues in between
// (cgs_raster_render_spec) keeps the device busy while the host learns the count.  cgs_raster_preprocess = both.
struct RasterCountSlot {
    uint32_t *pinned; hipEvent_t ev; bool pending; uint64_t ticket;
    // what a second depth sort of the launch needs (see raster_count_tail): valid from _launch to _wait
    int64_t P; void *geom_ws; size_t geom_bytes; hipStream_t stream; uint32_t epoch; bool ranged; bool spec_between; bool resorted;
};
static thread_local RasterCountSlot g_raster_slot = {nullptr, nullptr, false, 0, 0, nullptr, 0, nullptr, 0, false, false, false};
// Depth keys of a view sort on 27 bits (three passes) while every live depth stays below ~13107 (cgs_sort_depth_keys); the first
// view that reports a depth beyond that is sorted again on the full 32 bits (four passes) and so is every later view of the thread.
static thread_local bool g_depth_keys_full = false;
extern "C" int cgs_sort_depth_keys(const uint32_t *keys_in, uint32_t *keys_out, uint32_t *vals_out, uint32_t *keys_tmp,
                                   uint32_t *vals_tmp, int64_t n, void *scratch, size_t scratch_bytes, uint32_t *overflow,
                                   uint32_t epoch, void *stream);
extern "C" int cgs_debug_set_depth_keys_full(int on) { const int was = g_depth_keys_full; g_depth_keys_full = on != 0; return was; }

// Tickets of the *_launch / *_wait pairs: a slot holds ONE count per kind and host thread, so a second launch of the same kind
// overwrites what an earlier launch's wait would have read.  Every launch hands out a ticket (kind in the top byte, a
// per-thread serial below); the wait takes it back and refuses a ticket that is not the slot's current one instead of
// returning another launch's count.
uint64_t cgs_new_ticket(int kind) {
    static thread_local uint64_t serial = 0;
    return ((uint64_t)kind << 56) | (++serial & 0x00FFFFFFFFFFFFFFull);
}
static int raster_count_tail(int64_t P, CgsGeom &g, RasterCountSlot &sl, hipStream_t stream, bool full_keys = false);

extern "C" int cgs_raster_preprocess_launch(const cgs_raster_cfg *cfg, int64_t P, const float *means3D,
                                            const float *colors, const float *opacities, const float *scales,
                                            const float *rotations, void *geom_ws, size_t geom_bytes, int32_t *radii,
                                            void *stream_, uint64_t *ticket) {
    hipStream_t stream = (hipStream_t)stream_;
    RasterCountSlot &sl = g_raster_slot;
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (!ticket) { cgs_set_error("cgs_raster_preprocess_launch: NULL ticket"); return CGS_ERR_ARG; }
    *ticket = 0;
    if (P < 0 || P >= (1ll << 31)) { cgs_set_error("P out of range"); return CGS_ERR_ARG; }
    if (!sl.pinned) {
        CGS_CHECK_HIP(hipHostMalloc((void **)&sl.pinned, 64, hipHostMallocDefault));
        CGS_CHECK_HIP(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
    }
    sl.pending = false;
    sl.pinned[0] = 0;
    sl.ranged = sl.spec_between = sl.resorted = false;
    *ticket = sl.ticket = cgs_new_ticket(1);
    if (P == 0) return CGS_OK;
    if (!means3D || !colors || !opacities || !scales || !rotations || !radii || !geom_ws) {
        cgs_set_error("cgs_raster_preprocess: NULL input");
        return CGS_ERR_ARG;
    }
    CgsGeom g;
    if (!cgs_geom_carve(&g, geom_ws, geom_bytes, P)) {
        cgs_set_error("geometry workspace too small: %zu < %zu", geom_bytes, cgs_raster_geom_bytes(P));
        return CGS_ERR_WORKSPACE;
    }
    if ((rc = cgs_launch_preprocess(cfg, P, means3D, colors, opacities, scales, rotations, g, radii, false,
                                    stream)))
        return rc;
    sl.P = P; sl.geom_ws = geom_ws; sl.geom_bytes = geom_bytes; sl.stream = stream;
    return raster_count_tail(P, g, sl, stream);
}

// depth sort, tile rectangles in depth order, pair-offset scan and the copy of the pair count behind a preprocess launch
static int raster_count_tail(int64_t P, CgsGeom &g, RasterCountSlot &sl, hipStream_t stream, bool full_keys) {
    int rc;
    // depth order (stable: ties keep ascending Gaussian id)
    {
        CgsProfScope prof(CGS_PROF_DEPTH_SORT, stream);
        // (values = positions: the sort's first pass generates them, no iota launch)
        sl.ranged = !(full_keys || g_depth_keys_full);
        if (sl.ranged) {
            // 27-bit keys, three passes; a live depth beyond the range writes this launch's epoch to g.total[2], which travels
            // to the host with the pair count: cgs_raster_preprocess_wait then sorts again on 32 bits
            sl.epoch = (uint32_t)(sl.ticket & 0x7FFFFFFFu) + 1u;
            rc = cgs_sort_depth_keys(g.depth_key, g.sort_a, g.order, g.sort_b, g.sort_d, P, g.scratch, g.scratch_bytes,
                                     g.total + 2, sl.epoch, stream);
        } else {
            rc = cgs_sort_pairs_u32(g.depth_key, nullptr, g.sort_a, g.order, g.sort_b, g.sort_d, P, 0, 32, g.scratch,
                                    g.scratch_bytes, stream);
        }
        if (rc) return rc;
    }
    {
        CgsProfScope prof(CGS_PROF_OFFSETS_SCAN, stream);
        // tile rectangles and tile counts in depth order (sort_b / sort_d / sort_a are free after the sort)
        if ((rc = cgs_launch_gather_rects(P, g, stream))) return rc;
        if ((rc = cgs_scan_exclusive_u32_total(g.sort_a, g.offsets, P, g.scratch, g.scratch_bytes, g.total,
                                               stream)))
            return rc;
    }
    CGS_CHECK_HIP(hipMemcpyAsync(sl.pinned, g.total, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    CGS_CHECK_HIP(hipEventRecord(sl.ev, stream));
    sl.pending = true;
    return CGS_OK;
}

// cgs_raster_preprocess_launch with the Gaussians taken straight from the anchor expansion (csrc/expand_raster.hip): slot i
// of the n_anchor * K slots with flags[i] != 0 is Gaussian pos[i] (flags / pos / neural_opacity from cgs_expand_count_launch,
// P = the count it returned); scaling_out [P,3] receives the Gaussians' scales (the one per-Gaussian tensor the training
// loss reads), xyz_out [P,3] / rot_out [P,4] (both or neither) the positions and rotations for a later cgs_raster_backward.
// Everything downstream (cgs_raster_render*, _wait, cgs_raster_backward + cgs_expand_backward) is unchanged.
extern "C" int cgs_raster_preprocess_expand_launch(const cgs_raster_cfg *cfg, int64_t n_anchor, int K, const uint8_t *flags,
                                                   const uint32_t *pos, const float *anchor, const float *gscaling,
                                                   const float *offsets, const float *neural_opacity, const float *color_in,
                                                   const float *cov_in, const int64_t *src_row, int64_t P, float *scaling_out,
                                                   float *xyz_out, float *rot_out, void *geom_ws, size_t geom_bytes,
                                                   int32_t *radii, void *stream_, uint64_t *ticket) {
    hipStream_t stream = (hipStream_t)stream_;
    RasterCountSlot &sl = g_raster_slot;
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (!ticket) { cgs_set_error("cgs_raster_preprocess_expand_launch: NULL ticket"); return CGS_ERR_ARG; }
    *ticket = 0;
    if (P < 0 || P >= (1ll << 31) || n_anchor < 0 || K < 1 || n_anchor * (int64_t)K >= (1ll << 31) || P > n_anchor * (int64_t)K) {
        cgs_set_error("cgs_raster_preprocess_expand: sizes out of range");
        return CGS_ERR_ARG;
    }
    if (!sl.pinned) {
        CGS_CHECK_HIP(hipHostMalloc((void **)&sl.pinned, 64, hipHostMallocDefault));
        CGS_CHECK_HIP(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
    }
    sl.pending = false;
    sl.pinned[0] = 0;
    sl.ranged = sl.spec_between = sl.resorted = false;
    *ticket = sl.ticket = cgs_new_ticket(1);
    if (P == 0) return CGS_OK;
    if (!flags || !pos || !anchor || !gscaling || !offsets || !neural_opacity || !color_in || !cov_in || !scaling_out || !radii ||
        !geom_ws) {
        cgs_set_error("cgs_raster_preprocess_expand: NULL input");
        return CGS_ERR_ARG;
    }
    CgsGeom g;
    if (!cgs_geom_carve(&g, geom_ws, geom_bytes, P)) {
        cgs_set_error("geometry workspace too small: %zu < %zu", geom_bytes, cgs_raster_geom_bytes(P));
        return CGS_ERR_WORKSPACE;
    }
    const CgsExpandSrc x{n_anchor, K, flags, pos, anchor, gscaling, offsets, neural_opacity, color_in, cov_in, src_row};
    if ((xyz_out == nullptr) != (rot_out == nullptr)) { cgs_set_error("cgs_raster_preprocess_expand: xyz_out and rot_out go together"); return CGS_ERR_ARG; }
    if ((rc = cgs_launch_expand_preprocess(cfg, x, scaling_out, xyz_out, rot_out, g, radii, stream))) return rc;
    sl.P = P; sl.geom_ws = geom_ws; sl.geom_bytes = geom_bytes; sl.stream = stream;
    return raster_count_tail(P, g, sl, stream);
}

extern "C" int cgs_raster_preprocess_wait2(uint64_t ticket, int64_t *num_rendered_host, int *order_changed) {
    RasterCountSlot &sl = g_raster_slot;
    if (order_changed) *order_changed = 0;
    if (!num_rendered_host) { cgs_set_error("num_rendered_host is NULL"); return CGS_ERR_ARG; }
    *num_rendered_host = 0;
    if (!sl.pinned) { cgs_set_error("cgs_raster_preprocess_wait: no launch on this thread"); return CGS_ERR_ARG; }
    if (ticket == 0 || ticket != sl.ticket) {
        cgs_set_error("cgs_raster_preprocess_wait: stale ticket (another preprocess launch was issued on this thread since)");
        return CGS_ERR_ARG;
    }
    if (sl.pending) {
        CGS_CHECK_HIP(hipEventSynchronize(sl.ev));
        sl.pending = false;
        if (sl.ranged && sl.pinned[2] == sl.epoch) {
            // a live depth beyond the 27-bit key range (~13107): the order in the workspace is not the depth order.  Sort this
            // view again on the full 32 bits — the depth keys are intact — and keep to that for the thread's later views.
            g_depth_keys_full = true;
            CgsGeom g;
            if (!cgs_geom_carve(&g, sl.geom_ws, sl.geom_bytes, sl.P)) { cgs_set_error("geometry workspace too small"); return CGS_ERR_WORKSPACE; }
            int rc = raster_count_tail(sl.P, g, sl, sl.stream, true);
            if (rc) return rc;
            CGS_CHECK_HIP(hipEventSynchronize(sl.ev));
            sl.pending = false;
            sl.resorted = true;
        }
    }
    if (sl.resorted && order_changed) *order_changed = 1;
    *num_rendered_host = (int64_t)sl.pinned[0];
    return CGS_OK;
}

extern "C" int cgs_raster_preprocess_wait(uint64_t ticket, int64_t *num_rendered_host) {
    int changed = 0;
    int rc = cgs_raster_preprocess_wait2(ticket, num_rendered_host, &changed);
    if (rc) return rc;
    if (changed && g_raster_slot.spec_between) {
        cgs_set_error("cgs_raster_preprocess_wait: the view was sorted again on 32-bit depth keys after cgs_raster_render_spec ran on the "
                      "first order; call cgs_raster_render with the returned count (or use cgs_raster_preprocess_wait2)");
        return CGS_ERR_RESPEC;
    }
    return CGS_OK;
}

extern "C" int cgs_raster_preprocess(const cgs_raster_cfg *cfg, int64_t P, const float *means3D,
                                     const float *colors, const float *opacities, const float *scales,
                                     const float *rotations, void *geom_ws, size_t geom_bytes, int32_t *radii,
                                     int64_t *num_rendered_host, void *stream_) {
    if (!num_rendered_host) { cgs_set_error("num_rendered_host is NULL"); return CGS_ERR_ARG; }
    *num_rendered_host = 0;
    uint64_t ticket = 0;
    int rc = cgs_raster_preprocess_launch(cfg, P, means3D, colors, opacities, scales, rotations, geom_ws, geom_bytes, radii,
                                          stream_, &ticket);
    if (rc) return rc;
    return cgs_raster_preprocess_wait(ticket, num_rendered_host);
}

// ---- forward stage 2 -----------------------------------------------------------------
static int tile_bits(const cgs_raster_cfg *cfg) {
    const uint32_t nt = (uint32_t)(cgs_tiles_x(cfg) * cgs_tiles_y(cfg));
    int bits = 0;
    while ((1u << bits) < nt) ++bits;
    return bits;
}

// Which binning: 0 = by the pair count per Gaussian (the two-level path from CGS_BUCKET_MIN_RATIO tiles per Gaussian on),
// 1 = radix passes over (tile, Gaussian) pairs, 2 = two-level wherever the grid allows it.  cgs_debug_set_bin_mode is the
// test / measurement hook (tests/test_raster_gpu.py runs the list comparisons under both).
static int g_bin_mode = 0;
#ifndef CGS_BUCKET_MIN_RATIO
#define CGS_BUCKET_MIN_RATIO 6
#endif
extern "C" int cgs_debug_set_bin_mode(int mode) {
    if (mode < 0 || mode > 2) { cgs_set_error("cgs_debug_set_bin_mode: 0 (auto), 1 (radix) or 2 (buckets)"); return CGS_ERR_ARG; }
    g_bin_mode = mode;
    return CGS_OK;
}
static bool use_buckets(const cgs_raster_cfg *cfg, int64_t P, int64_t R) {
    if (g_bin_mode == 1 || !cgs_tile_bin_buckets_ok(cfg) || !cgs_tile_bin_buckets_fits(R)) return false;
    return g_bin_mode == 2 || R >= (int64_t)CGS_BUCKET_MIN_RATIO * P;
}

static int raster_render_impl(const cgs_raster_cfg *cfg, int64_t P, int64_t R, bool spec, void *geom_ws,
                              size_t geom_bytes, void *bin_ws, size_t bin_bytes, void *img_ws,
                              size_t img_bytes, float *out_color, hipStream_t stream) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (!cfg->bg || !out_color || !img_ws) { cgs_set_error("cgs_raster_render: NULL input"); return CGS_ERR_ARG; }
    CgsGeom g;
    CgsBin b;
    CgsImg im;
    memset(&g, 0, sizeof(g));
    memset(&b, 0, sizeof(b));
    const bool bin16 = cgs_tile_bin16_ok(tile_bits(cfg));
    if (spec && (!bin16 || P <= 0 || R <= 0)) {
        cgs_set_error("cgs_raster_render_spec: needs a grid of <= 65536 tiles, P > 0 and a positive capacity");
        return CGS_ERR_ARG;
    }
    if (!cgs_img_carve(&im, img_ws, img_bytes, cfg->image_height, cfg->image_width)) {
        cgs_set_error("image workspace too small");
        return CGS_ERR_WORKSPACE;
    }
    if (P > 0 && (!geom_ws || !cgs_geom_carve(&g, geom_ws, geom_bytes, P))) {
        cgs_set_error("geometry workspace missing or too small");
        return CGS_ERR_WORKSPACE;
    }
    if (R > 0) {
        if (!bin_ws || !cgs_bin_carve(&b, bin_ws, bin_bytes, P, R)) {
            cgs_set_error("binning workspace too small: %zu < %zu", bin_bytes, cgs_raster_bin_bytes(P, R));
            return CGS_ERR_WORKSPACE;
        }
        if (bin16 && use_buckets(cfg, P, R)) {
            // csrc/tile_bin.hip, two-level: bucket lists by one radix pass, tile lists by count + scan + fill
            if ((rc = cgs_launch_tile_bin_buckets(cfg, P, R, g, b, im, stream, spec ? g.total : nullptr))) return rc;
        } else if (bin16) {
            // csrc/tile_bin.hip: the first radix pass generates its pairs, 16-bit tile keys
            if ((rc = cgs_launch_tile_bin16(cfg, P, R, tile_bits(cfg), g, b, im, stream, spec ? g.total : nullptr))) return rc;
        } else {
            if ((rc = cgs_launch_emit_pairs(cfg, P, g, b, stream))) return rc;
            CgsProfScope prof(CGS_PROF_TILE_SORT, stream);
            if ((rc = cgs_sort_pairs_u32(b.tile_key_a, b.gid_a, b.tile_key_c, b.gid_sorted, b.tile_key_b, b.gid_b,
                                         R, 0, tile_bits(cfg), b.scratch, b.scratch_bytes, stream)))
                return rc;
        }
    }
    if (!(bin16 && R > 0) && (rc = cgs_launch_ranges(cfg, R, b, im, stream))) return rc;
    if ((rc = cgs_launch_tile_order(cfg, im, stream))) return rc;
    return cgs_launch_blend_fwd(cfg, g, b, im, out_color, stream);
}

extern "C" int cgs_raster_render(const cgs_raster_cfg *cfg, int64_t P, int64_t R, void *geom_ws,
                                 size_t geom_bytes, void *bin_ws, size_t bin_bytes, void *img_ws,
                                 size_t img_bytes, float *out_color, void *stream_) {
    return raster_render_impl(cfg, P, R, false, geom_ws, geom_bytes, bin_ws, bin_bytes, img_ws, img_bytes, out_color,
                              (hipStream_t)stream_);
}

// Speculative render between cgs_raster_preprocess_launch and _wait: R_cap is the capacity of the binning workspace
// (cgs_raster_bin_bytes(P, R_cap)), the pair count itself stays on the device.  Valid when the count _wait returns is
// <= R_cap (the per-tile lists are then exactly those of cgs_raster_render, and the backward takes R_cap as its R);
// otherwise the caller renders again with cgs_raster_render and the true count.  Grids of more than 65536 tiles: CGS_ERR_ARG.
extern "C" int cgs_raster_render_spec(const cgs_raster_cfg *cfg, int64_t P, int64_t R_cap, void *geom_ws,
                                      size_t geom_bytes, void *bin_ws, size_t bin_bytes, void *img_ws,
                                      size_t img_bytes, float *out_color, void *stream_) {
    g_raster_slot.spec_between = true;
    return raster_render_impl(cfg, P, R_cap, true, geom_ws, geom_bytes, bin_ws, bin_bytes, img_ws, img_bytes, out_color,
                              (hipStream_t)stream_);
}

// ---- backward -----------------------------------------------------------------------------
extern "C" int cgs_raster_backward(const cgs_raster_cfg *cfg, int64_t P, int64_t R, const float *means3D,
                                   const float *colors, const float *opacities, const float *scales,
                                   const float *rotations, const int32_t *radii, void *geom_ws,
                                   size_t geom_bytes, void *bin_ws, size_t bin_bytes, void *img_ws,
                                   size_t img_bytes, const float *dL_dout, float *dL_dmeans3D,
                                   float *dL_dmeans2D, float *dL_dcolors, float *dL_dopacities,
                                   float *dL_dscales, float *dL_drotations, void *scratch,
                                   size_t scratch_bytes, void *stream_) {
    (void)colors;
    (void)opacities;
    hipStream_t stream = (hipStream_t)stream_;
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (P == 0) return CGS_OK;
    if (!dL_dout || !dL_dmeans3D || !dL_dmeans2D || !dL_dcolors || !dL_dopacities || !dL_dscales ||
        !dL_drotations || !scratch || !radii) {
        cgs_set_error("cgs_raster_backward: NULL input");
        return CGS_ERR_ARG;
    }
    if (scratch_bytes < cgs_raster_bwd_scratch_bytes(P)) {
        cgs_set_error("backward scratch too small");
        return CGS_ERR_WORKSPACE;
    }
    CgsGeom g;
    CgsBin b;
    CgsImg im;
    memset(&b, 0, sizeof(b));
    if (!geom_ws || !img_ws || !cgs_geom_carve(&g, geom_ws, geom_bytes, P) ||
        !cgs_img_carve(&im, img_ws, img_bytes, cfg->image_height, cfg->image_width)) {
        cgs_set_error("workspace too small");
        return CGS_ERR_WORKSPACE;
    }
    float *d_mean_px = (float *)scratch;
    float *d_conic = (float *)((char *)scratch + cgs_align_up(2 * (size_t)P * sizeof(float), 256));
    CGS_CHECK_HIP(hipMemsetAsync(scratch, 0, cgs_raster_bwd_scratch_bytes(P), stream));
    if (R > 0) {
        if (!bin_ws || !cgs_bin_carve(&b, bin_ws, bin_bytes, P, R)) {
            cgs_set_error("binning workspace missing or too small");
            return CGS_ERR_WORKSPACE;
        }
        if ((rc = cgs_launch_blend_bwd(cfg, g, b, im, dL_dout, d_mean_px, d_conic, dL_dopacities, dL_dcolors,
                                       stream)))
            return rc;
    }
    return cgs_launch_preprocess_bwd(cfg, P, CGS_BLEND_BWD_RAW ? (const float4 *)g.rec : nullptr, means3D, scales, rotations, radii,
                                     d_mean_px, d_conic, dL_dmeans3D, dL_dmeans2D, dL_dscales, dL_drotations, stream);
}

extern "C" int cgs_raster_stats(const cgs_raster_cfg *cfg, void *img_ws, size_t img_bytes, int64_t *stats_out,
                                void *stream) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    CgsImg im;
    if (!img_ws || !stats_out || !cgs_img_carve(&im, img_ws, img_bytes, cfg->image_height, cfg->image_width)) {
        cgs_set_error("image workspace missing or too small");
        return CGS_ERR_WORKSPACE;
    }
    return cgs_launch_stats(cfg, im, stats_out, (hipStream_t)stream);
}
