/*
 * cgs.h — C-ABI of libcgs_hip.so, the MI355X (gfx950) hot-path library of
 * contextgs_amd.
 *
 * Every entry point takes raw DEVICE pointers (unless the name says host),
 * sizes, and the hipStream_t (as void*) the work must be enqueued on.  The
 * caller owns every buffer; the library never allocates or frees device
 * memory and never synchronises the device, except for the single 4-byte
 * read-back of `num_rendered` in cgs_raster_preprocess (the reference
 * rasterizer has the same one; cgs_raster_preprocess_launch / _wait + cgs_raster_render_spec hide it behind the render).  Every function returns 0 on success and a
 * non-zero code otherwise; cgs_last_error() gives the thread-local message.
 *
 * What each group replaces in the reference (wyf0912/ContextGS, paths
 * relative to the reference checkout):
 *
 *   cgs_filter                 GaussianRasterizer.visible_filter
 *                              (call site gaussian_renderer/__init__.py:280-285;
 *                              the CUDA extension itself is NOT in the mount)
 *   cgs_raster_*               GaussianRasterizer.__call__ forward + autograd
 *                              backward (call sites
 *                              gaussian_renderer/__init__.py:179-205, train.py:211)
 *   cgs_expand_*               generate_neural_gaussians' elementwise chain
 *                              (gaussian_renderer/__init__.py:106-145)
 *   cgs_entropy_gaussian_*     utils/entropy_models.py:30-50,141-156
 *   cgs_ste_multistep,
 *   cgs_quantize_anchor        utils/encodings.py:203-231
 *   cgs_level_key_range,
 *   cgs_level_unique           utils/multi_level.py:3-31 (torch.unique(dim=0) with
 *                              indices on the voxel keys of
 *                              scene/gaussian_model.py:1751-1765): packed keys,
 *                              cgs_sort_pairs_u32, run heads, scan
 *   cgs_anchor_gen_*           generate_neural_gaussians' anchor MLPs + mask +
 *                              compaction + per-Gaussian tail as one fused
 *                              kernel family (gaussian_renderer/__init__.py:106-145)
 *   cgs_ac_*                   torchac.encode_float_cdf / decode_float_cdf as
 *                              called from utils/encodings.py:108,138,157,178
 *   cgs_gaussian_ac_*,
 *   cgs_gaussian_stream_minmax encoder_gaussian / decoder_gaussian
 *                              (utils/encodings.py:83-144) without the
 *                              [n_sym, L] float table
 *   cgs_bernoulli_ac_*         encoder / decoder of the offset masks
 *                              (utils/encodings.py:147-180) as chunk streams
 *                              on the device (container version 2)
 *   cgs_pread_ranges,
 *   cgs_pwrite_ranges          the container's file reads / writes
 *                              (scene/gaussian_model.py:1235-1238, 1455-1481)
 *   cgs_mark_rows,
 *   cgs_zero_unmarked_rows     the zero rows autograd gives `x[visible_mask]`
 *                              outside the mask (gaussian_renderer/__init__.py:73-81)
 */
#ifndef CGS_H
#define CGS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CGS_VERSION 100 /* 0.1.0 */

#define CGS_OK 0
#define CGS_ERR_ARG 1
#define CGS_ERR_HIP 2
#define CGS_ERR_WORKSPACE 3
#define CGS_ERR_BOUNDS 4
#define CGS_ERR_RESPEC 5 /* cgs_raster_preprocess_wait: render again with cgs_raster_render (see there) */

int cgs_version(void);
const char *cgs_last_error(void);

/* ------------------------------------------------------------------ */
/* Rasterizer                                                          */
/* ------------------------------------------------------------------ */

/* Mirrors GaussianRasterizationSettings (the 12 keyword fields built at
 * gaussian_renderer/__init__.py:179-192).  Matrices are the tensors the
 * reference passes: ROW-vector convention, p_view = [x y z 1] @ viewmatrix,
 * row-major 4x4 fp32 on the device.  sh_degree / campos are accepted for
 * API parity; colours are always precomputed on this path (shs=None). */
typedef struct cgs_raster_cfg {
    int32_t image_height;
    int32_t image_width;
    float tanfovx;
    float tanfovy;
    float scale_modifier;
    int32_t prefiltered;
    int32_t debug;              /* !=0: synchronise + check after each kernel */
    const float *viewmatrix;    /* device, 16 floats */
    const float *projmatrix;    /* device, 16 floats */
    const float *campos;        /* device, 3 floats (unused: colours precomputed) */
    const float *bg;            /* device, 3 floats */
} cgs_raster_cfg;

/* visible_filter: preprocess-only; radii[i] > 0 iff anchor i survives the
 * near cull and touches at least one tile.  scales [N,3], rotations [N,4]
 * (r,x,y,z; used as given, not renormalised). */
int cgs_filter(const cgs_raster_cfg *cfg, int64_t N, const float *means3D,
               const float *scales, const float *rotations, int32_t *radii,
               void *stream);
/* prefilter_voxel (gaussian_renderer/__init__.py:232-287) in one launch: scaling [N, ld] = the model's RAW scaling rows
 * (columns 0..2 are read; exp is applied when scales_are_log != 0, i.e. the model is not a decoded one), rot1 [4] the
 * normalised rotation row every anchor shares (:283 repeats row 0), visible [N] receives `radii_pure > 0` as bool bytes. */
int cgs_filter_voxel(const cgs_raster_cfg *cfg, int64_t N, const float *means3D,
                     const float *scaling, int64_t ld, int scales_are_log, const float *rot1,
                     uint8_t *visible, void *stream);

/* Workspace sizes in bytes (host-side helpers, no device work). */
size_t cgs_raster_geom_bytes(int64_t P);
size_t cgs_raster_bin_bytes(int64_t P, int64_t num_rendered);
size_t cgs_raster_img_bytes(int32_t height, int32_t width);

/* Forward, stage 1: per-Gaussian preprocess, depth sort, offsets scan.
 * Writes radii[P] and the geometry workspace; returns the number of
 * tile/Gaussian pairs in *num_rendered_host (stream is synchronised once for
 * this 8-byte read-back). */
int cgs_raster_preprocess(const cgs_raster_cfg *cfg, int64_t P,
                          const float *means3D, const float *colors,
                          const float *opacities, const float *scales,
                          const float *rotations, void *geom_ws,
                          size_t geom_bytes, int32_t *radii,
                          int64_t *num_rendered_host, void *stream);

/* The same in two halves: _launch enqueues everything of stage 1 plus the 4-byte copy of the pair count and returns;
 * _wait blocks on THAT copy's event only (no stream drain) and returns the count.  Between the halves the caller may enqueue
 * cgs_raster_render_spec (below), so that the device works on the binning and the blend while the host learns the count.
 * TICKETS (all three *_launch / *_wait pairs of this header): the library keeps one pinned count per kind and host thread, so
 * a later launch of the same kind replaces what an earlier one's wait would read.  Every _launch writes a non-zero *ticket;
 * _wait(ticket, .) on the same thread returns the count of THAT launch and fails (CGS_ERR_ARG, "stale ticket") when another
 * launch of the kind was issued in between — it never returns another launch's count. */
int cgs_raster_preprocess_launch(const cgs_raster_cfg *cfg, int64_t P,
                                 const float *means3D, const float *colors,
                                 const float *opacities, const float *scales,
                                 const float *rotations, void *geom_ws,
                                 size_t geom_bytes, int32_t *radii, void *stream,
                                 uint64_t *ticket);
int cgs_raster_preprocess_wait(uint64_t ticket, int64_t *num_rendered_host);
/* DEPTH ORDER on 27-bit keys.  The depth sort of a view (the reference sorts 64-bit tile|depth keys per tile pair,
 * gaussian_renderer/__init__.py:197-205 -> its CUDA extension) runs on bits(z) - bits(0.2): three 9-bit passes instead of
 * four 8-bit ones, exact while every live depth is below ~13107 (cgs_sort_depth_keys below).  The first pass reports a live
 * depth beyond that to the host together with the pair count; _wait then sorts the view AGAIN on the full 32 bits before it
 * returns (same result as before, one sort later; the thread's later views go straight to 32 bits).  What a caller must know:
 * a cgs_raster_render_spec enqueued between _launch and _wait ran on the first order.  cgs_raster_preprocess_wait2 reports
 * that in *order_changed (the caller renders again with cgs_raster_render); cgs_raster_preprocess_wait returns
 * CGS_ERR_RESPEC in that case (count valid, same remedy) and CGS_OK otherwise. */
int cgs_raster_preprocess_wait2(uint64_t ticket, int64_t *num_rendered_host, int *order_changed);
/* test hook: force (1) / release (0) the 32-bit depth sort for this host thread; returns the previous setting */
int cgs_debug_set_depth_keys_full(int on);

/* Forward, stage 2: per-tile lists (stable by depth inside a tile), tile ranges, alpha blend.
 * out_color is [3, H, W]. */
int cgs_raster_render(const cgs_raster_cfg *cfg, int64_t P,
                      int64_t num_rendered, void *geom_ws, size_t geom_bytes,
                      void *bin_ws, size_t bin_bytes, void *img_ws,
                      size_t img_bytes, float *out_color, void *stream);

/* Stage 2 enqueued BEFORE the pair count is known on the host: `capacity` pairs fit the binning workspace
 * (cgs_raster_bin_bytes(P, capacity)), the kernels read the count on the device.  The result is the one of
 * cgs_raster_render when the count cgs_raster_preprocess_wait returns is <= capacity (pass `capacity` as num_rendered to
 * cgs_raster_backward then); otherwise it is garbage inside the buffers and the caller calls cgs_raster_render with the
 * true count.  Grids of more than 65536 tiles, P == 0 or capacity <= 0: error, nothing enqueued. */
int cgs_raster_render_spec(const cgs_raster_cfg *cfg, int64_t P,
                           int64_t capacity, void *geom_ws, size_t geom_bytes,
                           void *bin_ws, size_t bin_bytes, void *img_ws,
                           size_t img_bytes, float *out_color, void *stream);

/* Backward of the two stages above.  dL_dout is [3,H,W].  dL_dcolors and
 * dL_dopacities must be zero-initialised by the caller (the blend pass
 * accumulates into them); the other four are written for every Gaussian
 * (zeros for culled ones) and may arrive uninitialised.  dL_dmeans2D is [P,3] (x,y in the NDC-scaled convention the
 * densification threshold assumes; z = 0), the others match their inputs. */
int cgs_raster_backward(const cgs_raster_cfg *cfg, int64_t P,
                        int64_t num_rendered, const float *means3D,
                        const float *colors, const float *opacities,
                        const float *scales, const float *rotations,
                        const int32_t *radii, void *geom_ws, size_t geom_bytes,
                        void *bin_ws, size_t bin_bytes, void *img_ws,
                        size_t img_bytes, const float *dL_dout,
                        float *dL_dmeans3D, float *dL_dmeans2D,
                        float *dL_dcolors, float *dL_dopacities,
                        float *dL_dscales, float *dL_drotations,
                        void *scratch, size_t scratch_bytes, void *stream);

/* ---- the anchor expansion fused with the rasterizer's preprocess stage (csrc/expand_raster.hip) ----
 * Training path of render(): gaussian_renderer/__init__.py:130-145 (generate_neural_gaussians' tail) feeding :179-205.
 * cgs_raster_preprocess_expand_launch = cgs_raster_preprocess_launch whose Gaussians are the surviving slots of
 * cgs_expand_count_launch (flags / pos / neural_opacity [n_anchor*K], P = its count): slot i with flags[i] != 0 is Gaussian
 * pos[i], computed from anchor [n_anchor,3], gscaling [*,6] / offsets [*,K,3] (row src_row[n] of anchor n when src_row != NULL,
 * else row n), color_in [n_anchor*K,3], cov_in [n_anchor*K,7] exactly as cgs_expand_write would, and handed to the preprocess
 * stage in registers.  scaling_out [P,3] receives the scales; xyz_out [P,3] and rot_out [P,4] (both or neither) the positions
 * and normalised rotations — what cgs_raster_backward needs besides the workspaces; colours and opacities are not stored.
 * cgs_raster_preprocess_wait / cgs_raster_render* / cgs_raster_backward / cgs_expand_backward follow as usual.  Records, radii
 * and scales are bit-identical to cgs_expand_write + cgs_raster_preprocess_launch (same device functions, same values). */
int cgs_raster_preprocess_expand_launch(const cgs_raster_cfg *cfg, int64_t n_anchor, int K,
                                        const uint8_t *flags, const uint32_t *pos, const float *anchor,
                                        const float *gscaling, const float *offsets,
                                        const float *neural_opacity, const float *color_in,
                                        const float *cov_in, const int64_t *src_row, int64_t P,
                                        float *scaling_out, float *xyz_out, float *rot_out, void *geom_ws,
                                        size_t geom_bytes, int32_t *radii, void *stream, uint64_t *ticket);
size_t cgs_raster_bwd_scratch_bytes(int64_t P);

/* Statistics of the last render held in img_ws (device reads; async):
 * stats_out[0] = R_eff = sum over tiles of the deepest contributor position
 * (pairs the blend pass had to read), stats_out[1] = number of non-empty
 * tiles. int64 x 2 on the device. */
int cgs_raster_stats(const cgs_raster_cfg *cfg, void *img_ws, size_t img_bytes,
                     int64_t *stats_out, void *stream);
/* test hook (tests/test_raster_gpu.py::test_tile_lists_equal_the_pair_sort): re-bins
 * the geometry of the last forward into a second binning workspace with the
 * classic (tile, depth)-pair sort and counts the entries of the per-tile lists
 * (out2[0]) and of the tile ranges (out2[1]) that differ from what the forward
 * left in bin_ws / img_ws.  R_ws = the pair count bin_ws was carved with. */
int cgs_debug_bin_compare(const cgs_raster_cfg *cfg, int64_t P, int64_t R,
                          int64_t R_ws, void *geom_ws, size_t geom_bytes,
                          void *bin_ws, size_t bin_bytes, void *img_ws,
                          size_t img_bytes, void *bin_ws2, size_t bin_bytes2,
                          void *ranges2, int64_t *out2, void *stream);
/* Test / measurement hook: which tile binning cgs_raster_render* uses.  0 = chosen per
 * view by the pair count per Gaussian (default), 1 = radix passes over (tile, Gaussian)
 * pairs, 2 = the two-level binning (buckets of 8 x 4 tiles) wherever the grid has
 * <= 256 buckets.  Both leave the same lists and ranges (R3-R6 of SURVEY section 2.1;
 * the reference's duplicateWithKeys + SortPairs + identifyTileRanges, whose sources
 * are not in the mount). */
int cgs_debug_set_bin_mode(int mode);
/* "<sha256 of the sources the library was built from>|<compiler flags>"
 * (contextgs_amd/build.py); the loader refuses a CGS_LIB_PATH library whose
 * digest is not that of the sources next to it. */
const char *cgs_build_info(void);

/* ------------------------------------------------------------------ */
/* Generic device primitives used by the path (exposed for tests)      */
/* ------------------------------------------------------------------ */

/* Exclusive prefix sum of uint32, n elements; scratch from
 * cgs_scan_scratch_bytes(n). in == out allowed. */
size_t cgs_scan_scratch_bytes(int64_t n);
int cgs_scan_exclusive_u32(const uint32_t *in, uint32_t *out, int64_t n,
                           void *scratch, size_t scratch_bytes, void *stream);

/* Stable LSD radix sort of (key,value) uint32 pairs on key bits
 * [bit_lo, bit_hi).  Result lands in keys_out/vals_out; keys_tmp/vals_tmp
 * are ping-pong buffers of the same size. scratch from
 * cgs_sort_scratch_bytes(n).  vals_in == NULL: the values are the input
 * positions 0..n-1 (result = the sorting permutation). */
size_t cgs_sort_scratch_bytes(int64_t n);
int cgs_sort_pairs_u32(const uint32_t *keys_in, const uint32_t *vals_in,
                       uint32_t *keys_out, uint32_t *vals_out,
                       uint32_t *keys_tmp, uint32_t *vals_tmp, int64_t n,
                       int bit_lo, int bit_hi, void *scratch,
                       size_t scratch_bytes, void *stream);
/* Stable sort of DEPTH keys — float bits of view depths above the 0.2 near plane, 0xFFFFFFFF for a culled Gaussian (whose
 * place in the order is never read) — with values = positions: for the live keys the order of
 * cgs_sort_pairs_u32(keys_in, NULL, ..., 0, 32, ...) whenever bits(z) - bits(0.2) < 2^27 - 1 for all of them (z < ~13107),
 * in three 9-bit passes over those 27-bit keys (keys_out receives them).  A live key outside the range makes the first
 * pass write `epoch` to *overflow (device memory, not touched otherwise): the order is then not valid and the caller sorts
 * again with cgs_sort_pairs_u32; keys_in is never modified. */
int cgs_sort_depth_keys(const uint32_t *keys_in, uint32_t *keys_out, uint32_t *vals_out,
                        uint32_t *keys_tmp, uint32_t *vals_tmp, int64_t n, void *scratch,
                        size_t scratch_bytes, uint32_t *overflow, uint32_t epoch, void *stream);

/* ------------------------------------------------------------------ */
/* Per-kernel timing (used by bench.py's roofline leg)                  */
/* ------------------------------------------------------------------ */
/* When enabled, every hot kernel launch is bracketed by HIP events on its own
 * launch stream.  cgs_prof_enable resets the counters; cgs_prof_read waits for
 * the pending events of kernel `id` and returns total milliseconds and launch
 * count since the last reset.  Off by default; zero cost when off. */
int cgs_prof_enable(int on);
int cgs_prof_count(void);
const char *cgs_prof_name(int id);
int cgs_prof_read(int id, double *total_ms, int64_t *launches);

/* ------------------------------------------------------------------ */
/* Quantisers and rate model (utils/encodings.py, utils/entropy_models.py) */
/* ------------------------------------------------------------------ */

/* Quantize_anchor.forward (utils/encodings.py:219-227): anchors [N,3],
 * min_v/max_v [3] on the device; writes anchors_q [N,3] and the integer-valued
 * quantized [N,3].  round_digits = anchor_round_digits (16). */
int cgs_quantize_anchor(const float *anchors, const float *min_v,
                        const float *max_v, int64_t N, int round_digits,
                        float *anchors_q, float *quantized, void *stream);

/* STE_multistep.forward (utils/encodings.py:205-213): out = round(clamp(x)/Q)*Q.
 * x has n elements; element i uses Q[i / q_div] (q_div = row length for one Q
 * per row, 1 for elementwise Q). */
int cgs_ste_multistep(const float *x, const float *Q, int64_t n, int64_t q_div,
                      int use_clamp, float *out, void *stream);

/* Entropy_gaussian.forward (utils/entropy_models.py:34-50): bits =
 * -log2(max(|Phi(x+Q/2) - Phi(x-Q/2)|, 1e-6)) with the +-15000 Q clamp about
 * *x_mean (a device scalar).  Backward reproduces autograd through the
 * reference's graph including Low_bound.backward (:149-156), fused, without
 * the reference's host round trip. g_Q is elementwise; the caller reduces it
 * over the broadcast dimension. */
int cgs_entropy_gaussian_fwd(const float *x, const float *mean,
                             const float *scale, const float *Q, int64_t n,
                             int64_t q_div, const float *x_mean, int use_clamp,
                             float *bits, void *stream);
int cgs_entropy_gaussian_bwd(const float *x, const float *mean,
                             const float *scale, const float *Q, int64_t n,
                             int64_t q_div, const float *x_mean, int use_clamp,
                             const float *g_bits, float *g_x, float *g_mean,
                             float *g_scale, float *g_Q, void *stream);

/* ------------------------------------------------------------------ */
/* Fused 2-layer MLPs on the fp32 matrix cores                            */
/* ------------------------------------------------------------------ */
/* Y = act(W2 relu(W1 x + b1) + b2) for the nn.Sequential(Linear, ReLU, Linear
 * [, Tanh | Sigmoid]) modules of the path: mlp_opacity / mlp_color / mlp_cov
 * (scene/gaussian_model.py:153-174, 54 -> 50 -> {10 tanh, 30 sigmoid, 70}) and
 * mlp_grid[l] (:177-188, {71, 15} -> 100 -> 175).  act: 0 none, 1 tanh,
 * 2 sigmoid.  Weights in nn.Linear layout (W1 [hid,in], W2 [out,hid]).
 * X [n, ldx], Y [n, ldy]; H [n, hid] receives relu(.) for the backward (NULL
 * for inference).  Only the (in, hid, out, act) combinations above are
 * instantiated; others return CGS_ERR_ARG. */
int cgs_mlp2_forward(int in, int hid, int out, int act, const float *X,
                     int64_t ldx, const float *W1, const float *b1,
                     const float *W2, const float *b2, float *Y, int64_t ldy,
                     float *H, int64_t n, void *stream);
/* Backward: dX [n, lddx] (NULL to skip; accumulate_dx != 0 adds into it), dZ1
 * [n, hid] and dZ2 [n, out] are scratch outputs (dZ2 may be NULL when act == 0).
 * dW1/db1/dW2/db2 are ACCUMULATED into: zero or pre-load them.  scratch (device,
 * >= cgs_mlp_wgrad_scratch_bytes()) holds per-workgroup partial weight gradients
 * that a second kernel sums (deterministic, no global atomics); with scratch ==
 * NULL the partials are combined with fp32 atomics instead.
 * H == NULL (forward called with H == NULL; instances for {71,15} -> 100 -> 3, act 0): the
 * hidden layer is recomputed from X, W1, b1 instead of being stored and re-read, and the
 * second layer's weight gradient is accumulated inside the same kernel; needs scratch.
 * b1 is only read in that mode.
 * dW1 == NULL (then db1 == NULL too, and dW2 == db2 == NULL unless H == NULL): data gradients only — the
 * weight-gradient products are left to a later cgs_mlp2_wgrad on the dZ1 / dZ2 this call wrote (a data-parallel
 * step launches them behind the per-anchor gradient all-reduces, contextgs_amd/mlp.py defer_weight_gradients). */
size_t cgs_mlp_wgrad_scratch_bytes(void);
int cgs_mlp2_backward(int in, int hid, int out, int act, const float *X,
                      int64_t ldx, const float *W1, const float *b1, const float *W2,
                      const float *Y, const float *dY, int64_t ldy,
                      const float *H, float *dX, int64_t lddx,
                      int accumulate_dx, float *dZ1, float *dZ2, float *dW1,
                      float *db1, float *dW2, float *db2, int64_t n,
                      void *scratch, size_t scratch_bytes, void *stream);
/* cgs_mlp2_backward with the input gradient of row r stored to (accumulate_dx != 0: added into) row dx_rows[r] of dX:
 * the rows of a SUBSET of a larger batch (distinct indices; scene/gaussian_model.py:1658-1669 evaluates mlp_grid on
 * the rate subset) — `dX.index_add_(0, dx_rows, .)` folded into the store.  Needs the saved hidden layer (H != NULL). */
int cgs_mlp2_backward_rows(int in, int hid, int out, int act, const float *X, int64_t ldx,
                           const float *W1, const float *b1, const float *W2, const float *Y,
                           const float *dY, int64_t ldy, const float *H, float *dX, int64_t lddx,
                           int accumulate_dx, const int64_t *dx_rows, float *dZ1, float *dZ2,
                           float *dW1, float *db1, float *dW2, float *db2, int64_t n,
                           void *scratch, size_t scratch_bytes, void *stream);
/* The weight-gradient products of cgs_mlp2_backward on their own: dW1 += dZ1^T X, db1 += sum_rows dZ1 and, when
 * H != NULL, dW2 += dZ2^T H, db2 += sum_rows dZ2 (dZ2 [n, lddz2] = the backward's dZ2, or dY itself for act == 0).
 * H == NULL: the recomputing backward has accumulated the second layer already.  scratch as above. */
int cgs_mlp2_wgrad(int in, int hid, int out, const float *X, int64_t ldx, const float *H,
                   const float *dZ2, int64_t lddz2, const float *dZ1, float *dW1, float *db1,
                   float *dW2, float *db2, int64_t n, void *scratch, size_t scratch_bytes,
                   void *stream);

/* ---- fused element-wise stages of the per-level context model (training path) ----
 * Reference: scene/gaussian_model.py:1556-1707 (multi_scale_generating).
 *
 * cgs_rowcat_fwd: out[r] = [src_0[idx_0[r]] | src_1[idx_1[r]] | ...] — the input row of
 * mlp_grid[level] (:1594-1600: level anchor + hyper prior; :1650-1651 / :1711-1724: parent
 * anchor, parent feat, parent scaling + hyper prior).  data/idx/width/ld are HOST arrays
 * of nsrc (<= 4) entries; idx[s] == NULL means row r of source s; out is [n, sum(width)].
 * cgs_rowcat_bwd scatters dout back: mode[s] 0 = no gradient, 1 = store (distinct rows;
 * rows never referenced must be pre-zeroed by the caller), 2 = atomic add (repeating rows;
 * pre-zeroed by the caller). */
int cgs_rowcat_fwd(int nsrc, const void *const *data, const int64_t *const *idx,
                   const int *width, const int *ld, int64_t n, float *out,
                   void *stream);
int cgs_rowcat_bwd(int nsrc, void *const *ddata, const int64_t *const *idx,
                   const int *width, const int *ld, const int *mode, int64_t n,
                   const float *dout, void *stream);
/* The same with an optional row mask per source (rowmask[s] == NULL: none): source row r counts as
 * src_s[r] * (rowmask[s][r] != 0) — `anchor * mask_anchor.unsqueeze(1)` of scene/gaussian_model.py:1758-1759 (masked anchors
 * move to the origin from level 1 up) folded into the gather; the backward multiplies the scattered gradient likewise. */
int cgs_rowcat_fwd_masked(int nsrc, const void *const *data, const int64_t *const *idx,
                          const uint8_t *const *rowmask, const int *width, const int *ld,
                          int64_t n, float *out, void *stream);
int cgs_rowcat_bwd_masked(int nsrc, void *const *ddata, const int64_t *const *idx,
                          const uint8_t *const *rowmask, const int *width, const int *ld,
                          const int *mode, int64_t n, const float *dout, void *stream);

/* out [N, w] = 0 everywhere except out[idx[i]] = g[i], for ASCENDING distinct idx [n]: the backward of a row gather by the
 * visible-anchor list (gaussian_renderer/__init__.py:44-50) in one launch instead of a zero fill + a scatter. */
int cgs_scatter_rows_sorted(const float *g, const int64_t *idx, int64_t n, int64_t N, int w,
                            float *out, void *stream);
/* out[idx[i], 0:w] += g[i, 0:w] for DISTINCT row indices idx [n] (any order; rows < N): the second reader of a tensor adding
 * its rows' gradient into the first reader's buffer (the rate subset's rows of the mask weights and of the hyper latents,
 * scene/gaussian_model.py:1662-1669) without float atomics. */
int cgs_add_rows(const float *g, const int64_t *idx, int64_t n, int64_t N, int w, float *out,
                 void *stream);
/* out[i, 0:w] = x[idx[i], 0:w] (int64 row indices, 1 <= w <= 256, dense fp32 rows):
 * the forward of the same gathers (x[visible rows], gaussian_renderer/__init__.py:44-50). */
int cgs_gather_rows(const float *x, const int64_t *idx, int64_t n, int w,
                    float *out, void *stream);
/* Zeros for the rows a row list does NOT name.  The backward of a view writes
 * only the rows idx[0..n) (distinct) of an [n_full, w] gradient buffer
 * (gaussian_renderer/__init__.py:73-81: `[visible_mask]` indexing; autograd
 * zero-fills the rest): cgs_mark_rows stamps the listed rows with the call's
 * generation number `gen` (> 0, strictly increasing per stamp array; the
 * array starts zeroed and is never cleared), cgs_zero_unmarked_rows writes
 * zeros to every row of up to four arrays dst[k] [n_full, width[k]] whose
 * stamp differs from `gen`.  Replaces a full-buffer fill. */
int cgs_mark_rows(const int64_t *idx, int64_t n, int64_t n_full, uint32_t gen,
                  uint32_t *stamp, void *stream);
int cgs_zero_unmarked_rows(const uint32_t *stamp, uint32_t gen, int64_t n_full,
                           int narr, float *const *dst, const int *width,
                           void *stream);
/* Atomics-free backward of a context assembly whose first three sources are gathered parent rows
 * (anchor position [N,wa] by original row, coded features [n_parents,DF] and scaling [n_parents,DS]
 * by position in the coded prefix): the children of parent p are order[offs[p] .. offs[p+1])
 * (CSR from the level plan); their dout rows [n_children, ldo] are summed per parent.  d_f / d_s
 * are fully written; d_anchor [N,wa] (pre-zeroed) gets row parent_row[p] for parents with children.
 * Any of d_anchor / d_f / d_s may be NULL. */
int cgs_ctx_gather_bwd(const float *dout, int64_t ldo, int64_t n_parents,
                       const int64_t *offs, const int64_t *order,
                       const int64_t *parent_row, float *d_anchor, float *d_f,
                       float *d_s, int wa, int DF, int DS, void *stream);
/* the same; accumulate_anchor is a bit set: bit 0 = the d_anchor rows are added to (one anchor-gradient buffer shared by the
 * levels of a backward), bit 1 / bit 2 = the d_f / d_s rows are added to (they already hold the parents' other gradient) */
int cgs_ctx_gather_bwd_acc(const float *dout, int64_t ldo, int64_t n_parents,
                       const int64_t *offs, const int64_t *order,
                       const int64_t *parent_row, float *d_anchor, float *d_f,
                       float *d_s, int wa, int DF, int DS, int accumulate_anchor, void *stream);
/* Adaptive step sizes + training noise (:1603-1616):
 *   Q[r,k] = max(q0_k * (1 + tanh(qadj[r,k])), 1e-9),  k = feat, scaling, offsets
 *   yf = xf + u * Q[r,0], ys = xs + u * Q[r,1], yo = xo + u * Q[r,2],  u ~ U[-0.5, 0.5)
 * xf [n,D], xs [n,S], xo [n,O], qadj/Q [n,3].  u is a counter-based hash of (seed, tensor,
 * element) that the backward regenerates; the reference draws it with torch's Philox
 * uniform_, so streams differ while the distribution is the same.
 * rows (may be NULL): int64 [n], the forward reads row rows[r] of xf/xs/xo instead of row r — the
 * level's slice of the coding-order permutation, so the permuted copies of the parameter tensors
 * are never materialised; noise and outputs stay indexed by r.
 * The backward returns d_qadj; d_x == d_y is left to the caller unless rows is given, in which case
 * the rows of the FULL-size gradients dxf/dxs/dxo named by rows are written here (distinct rows; a
 * NULL dy* scatters zeros).  dy* may be NULL (no gradient); dQ_ext [n,3] is the gradient that
 * reaches Q from elsewhere (the rate term), may be NULL.
 * side_map (may be NULL; needs rows): int32 [n], side_map[r] = s >= 0 says row r is in the rate
 * subset and its rate gradients are row s of the COMPACT arrays side_f [n_sub,D], side_s [n_sub,S],
 * side_o [n_sub,O], side_Q [n_sub,3] (cgs_level_rate_bwd with compact != 0); they are added to
 * dy* / dQ_ext on the fly, so no N-row buffer is filled, scattered into and added.
 * sums3 (forward, may be NULL): double [cgs_means_accum_doubles()] on the device, zero before the
 * first use (slotted accumulator: same-address atomics serialise), += the sums of the SOURCE values
 * read from xf / xs / xo.  Over the levels of one step these are the numerators of the rate model's three
 * clamp centres (_anchor_feat.mean(), get_scaling.mean(), _offset.mean(); scene/gaussian_model.py:
 * 1664-1668); cgs_means_finalize turns them into float [3] means and zeroes the accumulator. */
int cgs_noise_quant_fwd(const float *xf, const float *xs, const float *xo,
                        const float *qadj, const int64_t *rows, int64_t n, int D,
                        int S, int O, uint64_t seed, float q0f, float q0s, float q0o,
                        float *yf, float *ys, float *yo, float *Q, double *sums3,
                        void *stream);
/* Training-step forms of the hyper prior (csrc/eb.hip; scene/gaussian_model.py:1556, 1662, 1689).
 * hyper_noise_gather: out[r,:] = hyper[a,:] + U(-1/2,1/2) with a = perm[r] (NULL: r), the noise being the
 * counter-based u(seed, tensor 3, a*C + c) — EntropyBottleneck's training quantisation, delivered in coding
 * order.  eb_bits: out[0] = sum over rows[0..n) (NULL: 0..n) and channels of -log2(max(likelihood, 1e-9))
 * of v [.,C] under the packed prior raw [C,58]; deterministic (block partials added in block order).
 * scratch: cgs_eb_bits_scratch_bytes() bytes, ZERO before the first use (the kernel leaves its arrival ticket zero).  eb_bits_bwd: for the
 * upstream gradient *g_sum (device scalar) g_v_sub [n,C] (row r <-> rows[r]) and g_raw [C,58] += . */
int cgs_hyper_noise_gather(const float *hyper, const int64_t *perm, int64_t n, int C,
                           uint64_t seed, float *out, void *stream);
size_t cgs_eb_bits_scratch_bytes(void);
int cgs_eb_bits_fwd(const float *v, const int64_t *rows, const float *raw, int64_t n, int C,
                    void *scratch, size_t scratch_bytes, float *out, void *stream);
int cgs_eb_bits_bwd(const float *v, const int64_t *rows, const float *raw, const float *g_sum,
                    int64_t n, int C, float *g_v_sub, float *g_raw, void *stream);
/* End of the rate model (scene/gaussian_model.py:1687-1705) in one launch each way.  S [L,3] = per level the
 * summed bits of (features, scaling, offsets) of the rate subset, hsum [1] the hyper bits, rate = live
 * fraction of the anchors, n_* the element counts of the three sums.  out4 = (bit_per_param,
 * bit_per_feat_param, bit_per_scaling_param, bit_per_offsets_param); raw [2+L] = (dead_frac, rate*hsum,
 * per-level bit sums) for the per-level report.  bwd: dS [L,3], dh [1] from g4. */
int cgs_rate_finish_fwd(const float *S, int L, const float *hsum, float rate, double n_feat,
                        double n_scaling, double n_offsets, float dead_frac, float *out4,
                        float *raw, void *stream);
int cgs_rate_finish_bwd(const float *g4, int L, float rate, double n_feat, double n_scaling,
                        double n_offsets, float *dS, float *dh, void *stream);
/* the same with one (nullable) one-element gradient per entry of out4: an entry the loss does not read (train.py:207 reads
 * bit_per_param only) has no gradient tensor, and no zero-filled [4] buffer has to be built for the others */
int cgs_rate_finish_bwd4(const float *g_all, const float *g_feat, const float *g_scaling,
                         const float *g_offsets, int L, float rate, double n_feat, double n_scaling,
                         double n_offsets, float *dS, float *dh, void *stream);
/* Per-step bookkeeping of the training context model (csrc/ctx_plan.hip; scene/gaussian_model.py:1658-1661
 * `choose_mask`, restricted per level).  choose_flags: for coding-order position r (anchor a = perm[r], or r
 * when perm is NULL) flag[r] = (u(seed, a) <= thresh, or given[a] when given != NULL) && mask[a]; writes
 * block counts (cgs_ctx_choose_blocks(n) uint32) and meta = int32 [2 + nlevels] (zeroed by the call):
 * [0] != 0 <=> anchor [N,3] / mask differ from anchor_ref / mask_ref (either may be NULL: not checked),
 * [1] = live anchors, [2 + l] = chosen rows of level l.  bounds_host: int64 [nlevels + 1] level boundaries in
 * coding order, on the HOST.  choose_compact: coding-order position (nz), original index (rows) and
 * level-local position (loc) of every chosen row, in coding order. */
size_t cgs_ctx_choose_blocks(int64_t n);
int cgs_ctx_choose_flags(const int64_t *perm, int64_t n, const uint8_t *mask,
                         const uint8_t *given, uint64_t seed, float thresh,
                         const float *anchor, const float *anchor_ref,
                         const uint8_t *mask_ref, const int64_t *bounds_host, int nlevels,
                         uint8_t *flags, uint32_t *block_counts, int32_t *meta, void *stream);
/* The same with `meta` spread over nslots (2 .. 256) slots of cgs_ctx_choose_slot_ints() ints each (one 128-byte line; zeroed by the
 * call): slot s holds the partial meta of the blocks s, s + nslots, ...; the caller adds entries [1 ..] over the slots and ORs [0].
 * (~1000 workgroups adding to one counter serialise on its address: 34 -> ~15 us at 1 M anchors.) */
int cgs_ctx_choose_slot_ints(void);
int cgs_ctx_choose_flags_slots(const int64_t *perm, int64_t n, const uint8_t *mask,
                               const uint8_t *given, uint64_t seed, float thresh,
                               const float *anchor, const float *anchor_ref,
                               const uint8_t *mask_ref, const int64_t *bounds_host, int nlevels,
                               uint8_t *flags, uint32_t *block_counts, int32_t *meta_slots, int nslots, void *stream);
int cgs_ctx_choose_compact(const uint8_t *flags, const uint32_t *block_counts,
                           const int64_t *perm, int64_t n, const int64_t *bounds_host,
                           int nlevels, int64_t *nz, int64_t *rows, int64_t *loc,
                           int32_t *sub_map, const int64_t *chosen_counts_host, void *stream);
size_t cgs_means_accum_doubles(void);
int cgs_means_finalize(double *sums3, int64_t na, int64_t nb, int64_t nc, float *out3,
                       void *stream);
int cgs_noise_quant_bwd(const float *dyf, const float *dys, const float *dyo,
                        const float *dQ_ext, const float *qadj, int64_t n, int D,
                        int S, int O, uint64_t seed, float q0f, float q0s, float q0o,
                        float *dqadj, const int64_t *rows, float *dxf, float *dxs,
                        float *dxo, const int32_t *side_map, const float *side_f,
                        const float *side_s, const float *side_o,
                        const float *side_Q, void *stream);
/* Bits of the chosen rows of one level (:1658-1669 with utils/entropy_models.py:30-50):
 * for s < n_sub, r = loc[s] (row inside the level; NULL = s):
 *   sums[0] += bits(yf[r], mean_f, scale_f, Q[r,0])      sums[1] += bits(ys[r], ..., Q[r,1])
 *   sums[2] += bits(yo[r], ..., Q[r,2]) * masks[grows[s], k]   (masks [N,K], may be NULL;
 *                                                               grows == NULL: row s of masks)
 * pred [n_sub, ldpred >= 2(D+6+3K)] = [mean_f D | scale_f D | mean_s 6 | scale_s 6 | mean_o 3K |
 * scale_o 3K] (the first 2(D+6+3K) outputs of mlp_grid).  x_means [3] are the clamp
 * centres when use_clamp != 0.  sums [3] is ACCUMULATED into.  The backward takes
 * g_sums [3] (device) and writes d_pred (all of it), rows loc[s] of d_yf/d_ys/d_yo and
 * of dQ [n_level,3] (the caller zero-fills the other rows); d_masks [N,K] (may be NULL,
 * pre-zeroed) receives the gradient of the mask weights (+= g_sums[2] * bits).
 * compact != 0: d_yf/d_ys/d_yo/dQ are [n_sub, .] arrays and row s (not loc[s]) is written —
 * the side arrays of cgs_noise_quant_bwd. */
int cgs_level_rate_fwd(const float *yf, const float *ys, const float *yo,
                       const float *Q, const int64_t *loc, const float *pred,
                       const float *masks, const int64_t *grows,
                       const float *x_means, int use_clamp, int64_t n_sub, int D,
                       int K, int64_t ldpred, float *sums, void *stream);
int cgs_level_rate_bwd(const float *yf, const float *ys, const float *yo,
                       const float *Q, const int64_t *loc, const float *pred,
                       const float *masks, const int64_t *grows,
                       const float *x_means, int use_clamp, int64_t n_sub, int D,
                       int K, int64_t ldpred, const float *g_sums, float *d_pred, float *d_yf,
                       float *d_ys, float *d_yo, float *dQ, float *d_masks,
                       int compact, void *stream);

/* The rate of one level's chosen rows with mlp_grid's mean / scale branch inside (round 6,
 * csrc/rate_sub.hip): scene/gaussian_model.py:1600-1608 restricted to the rows loc[0..m) of
 * the level + :1658-1669 + utils/entropy_models.py:30-50, ONE launch forward — the [m,175]
 * prediction never exists in memory — and the backward with its weight gradients in three.
 * in_dim 71 / 15; X [n,in_dim]: the level's input rows as cgs_ctx_level_fwd wrote them;
 * W1 [100,in_dim], b1 [100], W2 [175,100], b2 [175]: mlp_grid[level]; yf / ys / yo / Q: the
 * level's noisy values [n,50] [n,6] [n,30] and step sizes [n,3]; masks [m,10]: the mask
 * weights of the chosen rows (NULL = ones); x_means [3] (use_clamp).  sums3 [3] is
 * ACCUMULATED into.  The backward writes the compact side arrays of cgs_ctx_level_bwd
 * (side_f [m,50], side_s [m,6], side_o [m,30], side_Q [m,3], dx_sub [m,in_dim]), d_masks
 * [m,10] (may be NULL; every row written) and ASSIGNS dW1 / db1 / dW2 [175,100] / db2 [175]
 * (the three step-size rows: zeros — cgs_ctx_level_bwd accumulates them).
 * scratch >= cgs_rate_sub_bwd_scratch_bytes(in_dim, m). */
int cgs_rate_sub_fwd(int in_dim, const float *X, int64_t n, const int64_t *loc, int64_t m,
                     const float *W1, const float *b1, const float *W2, const float *b2,
                     const float *yf, const float *ys, const float *yo, const float *Q,
                     const float *masks, const float *x_means, int use_clamp, float *sums3,
                     void *stream);
size_t cgs_rate_sub_bwd_scratch_bytes(int in_dim, int64_t m);
int cgs_rate_sub_bwd(int in_dim, const float *X, int64_t n, const int64_t *loc, int64_t m,
                     const float *W1, const float *b1, const float *W2, const float *b2,
                     const float *yf, const float *ys, const float *yo, const float *Q,
                     const float *masks, const float *x_means, int use_clamp,
                     const float *g_sums3, float *side_f, float *side_s, float *side_o,
                     float *side_Q, float *dx_sub, float *d_masks, float *dW1, float *db1,
                     float *dW2, float *db2, void *scratch, size_t scratch_bytes, void *stream);

/* One launch per level and direction for the every-row half of the level loop, training path (round 5, csrc/ctx_level.hip;
 * scene/gaussian_model.py:1594-1616 and its autograd) — replaces cgs_rowcat_fwd + cgs_mlp2_forward(., 100, 3) +
 * cgs_noise_quant_fwd, and cgs_noise_quant_bwd + cgs_mlp2_backward(., 100, 3) + the first layer's cgs_mlp2_wgrad.
 * in_dim 71 (context level): X[r] = [anchor[a_rows[r]] | base_f[pos[r]] | base_s[pos[r]] | hyp[r]] with base_f [., 50] /
 * base_s [., 6] the coded prefix; in_dim 15 (first level): X[r] = [anchor[a_rows[r]] (* a_mask[a_rows[r]], uint8, may be
 * NULL) | hyp[r]], hyp [n, 12].  W1 [100, in_dim], b1 [100]; W2q [3, 100] / b2q [3] = the LAST three rows of mlp_grid's second
 * layer (the step-size adjustments).  Outputs: X [n, in_dim] (kept for the backward and the rate subset), Q [n,3] =
 * clamp(q0 (1 + tanh(.)), 1e-9), y* = x*[rows[r]] + U(-1/2,1/2) Q (the counter-based noise of cgs_noise_quant_fwd: same
 * element -> value map); xf [N,50], xs [N,6], xo [N,30]; sums3 as in cgs_noise_quant_fwd (may be NULL).
 * Backward: dy* [n, .] (each may be NULL = zeros), dQ_ext [n,3] (may be NULL); rows rows[r] of the FULL-size dxf / dxs / dxo are
 * overwritten with dy[r] (+ row side_map[r] of the compact side arrays when >= 0); dX [n, in_dim] = the input-row gradient
 * (+ row side_map[r] of dx_sub [m, in_dim], may be NULL); dW1 / db1 / dW2q / db2q are ACCUMULATED into, without atomics
 * (per-workgroup images in scratch >= cgs_ctx_level_bwd_scratch_bytes(), summed in a fixed order).
 * n_anchor = rows of anchor / a_mask / xf / xs / xo / dxf / dxs / dxo, n_par = rows of base_f / base_s, m_side = rows of the side
 * arrays and dx_sub: the kernels address every operand through bounds-checked raw buffers (32-bit offsets: an operand of 4 GB or
 * more is refused with CGS_ERR_ARG). */
int cgs_ctx_level_fwd(int in_dim, const float *anchor, int64_t n_anchor, const int64_t *a_rows,
                      const uint8_t *a_mask, const float *base_f, const float *base_s, int64_t n_par,
                      const int64_t *pos, const float *hyp, int64_t n, const float *W1, const float *b1, const float *W2q, const float *b2q,
                      const float *xf, const float *xs, const float *xo, const int64_t *rows, uint64_t seed,
                      float q0f, float q0s, float q0o, float *X, float *yf, float *ys, float *yo, float *Q,
                      double *sums3, void *stream);
size_t cgs_ctx_level_bwd_scratch_bytes(void);
int cgs_ctx_level_bwd(int in_dim, const float *X, const float *W1, const float *b1, const float *W2q,
                      const float *b2q, const float *dyf, const float *dys, const float *dyo,
                      const float *dQ_ext, int64_t n, uint64_t seed, float q0f, float q0s, float q0o,
                      const int64_t *rows, int64_t n_anchor, float *dxf, float *dxs, float *dxo,
                      const int32_t *side_map, int64_t m_side, const float *side_f, const float *side_s, const float *side_o, const float *side_Q,
                      const float *dx_sub, float *dX, float *dW1, float *db1, float *dW2q, float *db2q,
                      void *scratch, size_t scratch_bytes, void *stream);
/* cgs_ctx_level_bwd with one more output: d_hyp_rows [n_anchor, 12] (may be NULL = cgs_ctx_level_bwd) — row rows[r] also receives the
 * last 12 columns of dX[r], the gradient of the level's hyper latents, in the latents' own row order (every anchor belongs to one
 * level: after all levels of a step every row is written, and the hyper prior's backward needs no pass through the inverse coding
 * permutation; scene/gaussian_model.py:1588-1600 concatenates the latents into the level MLP's input). */
int cgs_ctx_level_bwd2(int in_dim, const float *X, const float *W1, const float *b1, const float *W2q,
                       const float *b2q, const float *dyf, const float *dys, const float *dyo,
                       const float *dQ_ext, int64_t n, uint64_t seed, float q0f, float q0s, float q0o,
                       const int64_t *rows, int64_t n_anchor, float *dxf, float *dxs, float *dxo,
                       const int32_t *side_map, int64_t m_side, const float *side_f, const float *side_s, const float *side_o, const float *side_Q,
                       const float *dx_sub, float *dX, float *d_hyp_rows, float *dW1, float *db1, float *dW2q, float *db2q,
                       void *scratch, size_t scratch_bytes, void *stream);

/* Row strides (floats) of the buffers the anchor-MLP forward and backward hand to each other, so that callers size them:
 * out4 = {row stride of Hcat (150: head h in columns 50 h .. 50 h + 49), row stride of X_out of the _rows variant (54),
 * row stride of dZ1cat (192), column pitch of a head in dZ1cat (64: head h in columns 64 h .. 64 h + 49, pad columns hold
 * zeros; dW1cat is [192, 54] and db1cat [192] with the same row numbering)}.  dZ1cat is padded so that every 16-feature chunk
 * the backward stores per row is one aligned 64-byte sector; Hcat and X_out stay packed. */
int cgs_anchor_mlp3_layout(int *out4);

/* The three anchor MLPs (mlp_opacity 54->50->10 tanh, mlp_color 54->50->30
 * sigmoid, mlp_cov 54->50->70; gaussian_renderer/__init__.py:112,122,126) on
 * their shared input in ONE launch each way.  W1/b1/W2/b2 (and dW2/db2) are
 * HOST arrays of three device pointers in the order (opacity, color, cov).
 * Hcat [n,150] holds the three ReLU hidden layers side by side; dZ1cat [n,192] their gradients, head h in
 * columns 64 h .. 64 h + 49 (cgs_anchor_mlp3_layout); dZ2_op [n,10], dZ2_color [n,30] are backward
 * scratch.  dW1cat [192,54] / db1cat [192] hold the three first-layer gradients in rows 64 h .. 64 h + 49;
 * all weight / bias gradients are ACCUMULATED into (scratch as for cgs_mlp2_backward). */
int cgs_anchor_mlp3_forward(const float *X, int64_t ldx,
                            const float *const *W1, const float *const *b1,
                            const float *const *W2, const float *const *b2,
                            float *Y_op, float *Y_color, float *Y_cov,
                            float *Hcat, int64_t n, void *stream);
int cgs_anchor_mlp3_backward(const float *X, int64_t ldx,
                             const float *const *W1, const float *const *W2,
                             const float *Y_op, const float *Y_color,
                             const float *dY_op, const float *dY_color,
                             const float *dY_cov, const float *Hcat, float *dX,
                             int64_t lddx, float *dZ1cat, float *dZ2_op,
                             float *dZ2_color, float *dW1cat, float *db1cat,
                             float *const *dW2, float *const *db2, int64_t n,
                             void *scratch, size_t scratch_bytes, void *stream);
/* The same pair with the MLP input assembled on the fly (gaussian_renderer/__init__.py:106-110 fused into the
 * operand load): X[r] = [feat_src[src_row[r], 0:50] | (a - cam)/|a - cam| | |a - cam|], a = anchor_vis[r],
 * cam3 = camera centre (device float[3]).  X_out [n,54] receives the assembled rows (read by the weight-
 * gradient pass).  The backward stores dX[:, 0:50] into rows src_row[r] of d_feat_src [*,50] (distinct rows;
 * rows that no visible anchor reads are the caller's to zero) and pulls the four view columns back to
 * d_anchor_vis [n,3]; everything else as cgs_anchor_mlp3_backward (X = the forward's X_out). */
int cgs_anchor_mlp3_forward_rows(const float *feat_src, const int64_t *src_row,
                                 const float *anchor_vis, const float *cam3, float *X_out,
                                 const float *const *W1, const float *const *b1,
                                 const float *const *W2, const float *const *b2, float *Y_op,
                                 float *Y_color, float *Y_cov, float *Hcat, int64_t n, void *stream);
/* (round 6) X_out may be NULL in the forward; the backward then takes X == NULL and feat_src (the forward's): its fused
 * data + weight-gradient kernel assembles the rows again from the same operands (needs dW1cat != NULL, n <= 4 M rows). */
int cgs_anchor_mlp3_backward_rows(const float *X, const float *feat_src, const int64_t *src_row, const float *anchor_vis,
                                  const float *cam3, const float *const *W1, const float *const *W2,
                                  const float *Y_op, const float *Y_color, const float *dY_op,
                                  const float *dY_color, const float *dY_cov, const float *Hcat,
                                  float *d_feat_src, float *d_anchor_vis, float *dZ1cat, float *dZ2_op,
                                  float *dZ2_color, float *dW1cat, float *db1cat, float *const *dW2,
                                  float *const *db2, int64_t n, void *scratch, size_t scratch_bytes,
                                  void *stream);
/* (round 6) The _rows pair with Hcat — a buffer nothing but the fused backward reads — in FRAGMENT-MAJOR form (tiled != 0): per
 * 16-row tile the 1 KB register fragments of the kernels in lane order, the columns past the last full 16 behind them; every
 * store / load instruction of a wave is then one contiguous KB instead of sixteen 64-byte chunks a row apart.  Hcat holds
 * ceil(n / 16) * 16 rows; n <= 4 M rows; the backward must ask for the weight gradients in the same call (dW1cat != NULL).
 * X_out / X stay row-major.  tiled == 0: exactly the two functions above.  Same arithmetic either way
 * (gaussian_renderer/__init__.py:106-127 is the contract; the layout of Hcat is internal to the pair). */
int cgs_anchor_mlp3_forward_rows_t(const float *feat_src, const int64_t *src_row,
                                   const float *anchor_vis, const float *cam3, float *X_out,
                                   const float *const *W1, const float *const *b1,
                                   const float *const *W2, const float *const *b2, float *Y_op,
                                   float *Y_color, float *Y_cov, float *Hcat, int64_t n, int tiled, void *stream);
int cgs_anchor_mlp3_backward_rows_t(const float *X, const float *feat_src, const int64_t *src_row, const float *anchor_vis,
                                    const float *cam3, const float *const *W1, const float *const *W2,
                                    const float *Y_op, const float *Y_color, const float *dY_op,
                                    const float *dY_color, const float *dY_cov, const float *Hcat,
                                    float *d_feat_src, float *d_anchor_vis, float *dZ1cat, float *dZ2_op,
                                    float *dZ2_color, float *dW1cat, float *db1cat, float *const *dW2,
                                    float *const *db2, int64_t n, int tiled, void *scratch, size_t scratch_bytes,
                                    void *stream);
/* cgs_anchor_mlp3_backward / _backward_rows with dW1cat == db1cat == dW2 == db2 == NULL write the data gradients
 * only; this is the weight-gradient launch they leave out, on the buffers they wrote (X: the backward's X, ldx = 54 —
 * or the forward's X_out with ldx = cgs_anchor_mlp3_layout()[1] for the _rows pair; dY_cov: the covariance head's
 * incoming gradient).  All gradients are ACCUMULATED into. */
int cgs_anchor_mlp3_wgrad(const float *X, int64_t ldx, const float *Hcat, const float *dZ1cat,
                          const float *dZ2_op, const float *dZ2_color, const float *dY_cov,
                          float *dW1cat, float *db1cat, float *const *dW2, float *const *db2,
                          int64_t n, void *scratch, size_t scratch_bytes, void *stream);

/* out[a, 0:w] = row idx[a] (idx == NULL: row a) of the virtual concatenation of nseg (<= 4) row blocks: block k = rows
 * begin[k] .. begin[k+1]-1, row r at src[k] + (r - begin[k]) * ld[k] floats (src[k] == NULL: zeros).  src / ld / begin are
 * HOST arrays (begin has nseg + 1 entries).  The backward of the hyper latents' per-level slices
 * (scene/gaussian_model.py:1594-1600, :1650-1651 consume them level by level): the levels' gradients, two of them strided
 * column slices, leave through the inverse coding permutation in one pass. */
int cgs_gather_rows_segmented(int nseg, const float *const *src, const int64_t *ld,
                              const int64_t *begin, const int64_t *idx, int64_t n, int w,
                              float *out, void *stream);

/* Factorised-prior likelihood of the hyper latents (EntropyBottleneck.forward,
 * scene/gaussian_model.py:1556; compressai is not in the mount, the density is
 * the one of utils/entropy_models.py:103-138 with filters (3,3,3,3)).  v, lik,
 * g_lik, g_v are [n, C] row-major; raw / g_raw are [C, 58] packed raw
 * parameters per channel: M0 3, B0 3, F0 3, then (M 9, B 3, F 3) x 3, M4 3, B4 1
 * (matrices pass through softplus, factors through tanh inside the kernel).
 * lik = max(|sigmoid(s u) - sigmoid(s l)|, 1e-9). g_raw is ACCUMULATED into. */
int cgs_eb_likelihood_fwd(const float *v, const float *raw, int64_t n, int C,
                          float *lik, void *stream);
int cgs_eb_likelihood_bwd(const float *v, const float *raw, const float *g_lik,
                          int64_t n, int C, float *g_v, float *g_raw,
                          void *stream);

/* ------------------------------------------------------------------ */
/* Entropy coding (torchac / compressai call sites of the reference)     */
/* ------------------------------------------------------------------ */
/* --- table-driven arithmetic coder on the HOST (torchac drop-in:
 * torchac.encode_float_cdf / decode_float_cdf, utils/encodings.py:108,138,
 * 157,178).  cdf is uint16 [n_sym, Lp] (cgs_cdf_float_to_u16_host applies the
 * published float->int conversion), sym int16 in [0, Lp-2].  All pointers
 * are HOST pointers. */
size_t cgs_ac_max_bytes(int64_t n_sym);
int cgs_cdf_float_to_u16_host(const float *cdf, int64_t n_sym, int Lp,
                              uint16_t *out);
int cgs_ac_encode_table_host(const uint16_t *cdf, int Lp, const int16_t *sym,
                             int64_t n_sym, uint8_t *out, size_t out_cap,
                             size_t *out_len);
int cgs_ac_decode_table_host(const uint16_t *cdf, int Lp, int64_t n_sym,
                             const uint8_t *in, size_t in_len,
                             int16_t *sym_out);
/* one CDF row shared by every symbol (the Bernoulli mask stream,
 * utils/encodings.py:147-180) */
int cgs_ac_encode_const_host(const uint16_t *row, int Lp, const int16_t *sym,
                             int64_t n_sym, uint8_t *out, size_t out_cap,
                             size_t *out_len);
int cgs_ac_decode_const_host(const uint16_t *row, int Lp, int64_t n_sym,
                             const uint8_t *in, size_t in_len,
                             int16_t *sym_out);

/* --- batched Gaussian codec on the DEVICE (encoder_gaussian /
 * decoder_gaussian, utils/encodings.py:83-144, for all 1000-anchor chunk
 * streams of a level/attribute at once).  x/mean/scale are flat [n_total]
 * device arrays; element i uses Q[i / q_div]; stream s covers
 * [stream_off[s], stream_off[s+1]) (int64, device).  min_v/max_v [n_streams]
 * hold round(x/Q) extrema per stream (cgs_gaussian_stream_minmax fills them).
 * Encode writes stream s at out + out_off[s] (capacity out_off[s+1]-out_off[s],
 * use cgs_ac_max_bytes) and its byte length to out_len[s]; *status (device
 * int32, zeroed by the caller) becomes non-zero on a range/overflow error.
 * Decode is the exact inverse: x_out[i] = (sym + min) * Q. */
int cgs_gaussian_stream_minmax(const float *x, const float *Q, int64_t q_div,
                               const int64_t *stream_off, int n_streams,
                               int32_t *min_out, int32_t *max_out,
                               void *stream);
int cgs_gaussian_ac_encode(const float *x, const float *mean,
                           const float *scale, const float *Q, int64_t q_div,
                           const int64_t *stream_off, int n_streams,
                           const int32_t *min_v, const int32_t *max_v,
                           uint8_t *out, const int64_t *out_off,
                           uint32_t *out_len, int32_t *status, void *stream);
int cgs_gaussian_ac_decode(const float *mean, const float *scale,
                           const float *Q, int64_t q_div,
                           const int64_t *stream_off, int n_streams,
                           const int32_t *min_v, const int32_t *max_v,
                           const uint8_t *in, const int64_t *in_off,
                           float *x_out, void *stream);
/* Pack the chunk streams a coder launch left in their worst-case slots
 * (src + src_off[s], len[s] bytes) back to back at dst + dst_off[s]: the
 * on-disk layout of featN.b / scalingN.b / offsetsN.b
 * (scene/gaussian_model.py:1235-1238, b"".join of the chunk strings).  `src`
 * must be readable 8 bytes past its last slot. */
int cgs_streams_compact(const uint8_t *src, const int64_t *src_off,
                        const uint32_t *len, const int64_t *dst_off,
                        int n_streams, uint8_t *dst, void *stream);
/* Host file I/O of the container (the .b files conduct_encoding writes with
 * open(...).write(b"".join(...)) and conduct_decoding reads back,
 * scene/gaussian_model.py:1235-1238, 1455-1481): range i = nbytes[i] bytes at
 * file_off[i] of paths[i] <-> memory dst[i] / src[i], moved by `threads` plain
 * C++ workers (no interpreter thread per piece: see csrc/file_io.cpp).  Host
 * pointers (pinned or pageable).  pwrite creates missing files and never
 * truncates.  Blocking; no stream. */
int cgs_pread_ranges(int n, const char *const *paths, const int64_t *file_off,
                     const int64_t *nbytes, void *const *dst, int threads);
int cgs_pwrite_ranges(int n, const char *const *paths, const int64_t *file_off,
                      const int64_t *nbytes, const void *const *src, int threads);
/* Container version 2, Gaussian-coded attributes: the same symbol sequence and
 * the same coder as cgs_gaussian_ac_encode, cut into BLOCKS (blk_off, element
 * offsets as stream_off above) that one wave codes as 64 INTERLEAVED lane
 * streams — lane l codes the block's symbols l, l + 64, ... with its own
 * coder in vector registers — instead of one serial stream per wave.  A block
 * in the file = 64 little-endian uint16 lane-stream byte lengths, then the 64
 * lane streams back to back; out_len[b] is its byte length.  Encode writes
 * block b into its worst-case region out + out_off[b]
 * (cgs_lanes_block_slot_bytes(symbols of the block, 2) bytes, 8-byte aligned),
 * cgs_lanes_compact packs the regions at dst + dst_off[b].  `in` must be
 * readable 4 bytes past its last block.  min_v / max_v / status as above. */
size_t cgs_lanes_block_slot_bytes(int64_t n_symbols, int bytes_per_symbol);
int cgs_gaussian_ac_encode_lanes(const float *x, const float *mean,
                                 const float *scale, const float *Q,
                                 int64_t q_div, const int64_t *blk_off,
                                 int n_blocks, const int32_t *min_v,
                                 const int32_t *max_v, uint8_t *out,
                                 const int64_t *out_off, uint32_t *out_len,
                                 int32_t *status, void *stream);
int cgs_lanes_compact(const uint8_t *src, const int64_t *src_off,
                      const int64_t *blk_off, const int64_t *dst_off,
                      int n_blocks, uint8_t *dst, int bytes_per_symbol,
                      void *stream);
int cgs_gaussian_ac_decode_lanes(const float *mean, const float *scale,
                                 const float *Q, int64_t q_div,
                                 const int64_t *blk_off, int n_blocks,
                                 const int32_t *min_v, const int32_t *max_v,
                                 const uint8_t *in, const int64_t *in_off,
                                 float *x_out, int32_t *status, void *stream);
/* (both lane decoders: in_off has n_blocks + 1 entries; a block whose 64 lane
 * lengths + 128 header bytes do not add up to in_off[b+1] - in_off[b], or whose
 * min / max give a CDF of more than 2^16 entries, is not decoded and
 * *status (device int32, zeroed by the caller, may be NULL) becomes 1 + b.)
 * Container version 2, hyper.b: the hyper latents' integer symbols
 * (scene/gaussian_model.py:1082-1098,1326-1338; compressai's
 * EntropyBottleneck.compress / decompress, per-channel frequency tables) as
 * lane-parallel blocks of the same arithmetic coder.  sym int32 flat
 * [C * n_per_channel] (channel-major); block b covers [blk_off[b], blk_off[b+1])
 * and lies inside channel blk_ch[b]; cdf int32 [C, max_len] (total 2^16, last
 * used slot = escape), cdf_len / offset [C] as EntropyBottleneck.update builds
 * them.  Lane slots hold 6 bytes per symbol (cgs_lanes_block_slot_bytes(n, 6),
 * cgs_lanes_compact(..., 6, ...)); status 2 = a slot overflowed.  Decode writes
 * dequantised rows: out_rows[i * ld_rows + c] = symbol + medians[c]. */
int cgs_table_ac_encode_lanes(const int32_t *sym, const int64_t *blk_off,
                              const int32_t *blk_ch, int n_blocks,
                              const int32_t *cdf, int max_len,
                              const int32_t *cdf_len, const int32_t *offset,
                              uint8_t *out, const int64_t *out_off,
                              uint32_t *out_len, int32_t *status, void *stream);
int cgs_table_ac_decode_lanes(const int64_t *blk_off, const int32_t *blk_ch,
                              int n_blocks, const int32_t *cdf, int max_len,
                              const int32_t *cdf_len, const int32_t *offset,
                              const float *medians, int64_t n_per_channel,
                              const uint8_t *in, const int64_t *in_off,
                              float *out_rows, int64_t ld_rows, int32_t *status,
                              void *stream);
/* Container version 2: the offset-mask symbols (scene/gaussian_model.py:1265-1269,
 * 1348-1353; utils/encodings.py:147-180 code them as ONE serial stream) cut into
 * chunk streams and coded by the same arithmetic coder, one wave per stream.
 * sym01 flat float {0,1} (device); c1 = the interior entry of the integer CDF row
 * [0, c1, 2^16] (cgs_cdf_float_to_u16_host of [0, 1-p, 1]); stream s covers
 * [stream_off[s], stream_off[s+1]) and holds exactly the bytes
 * cgs_ac_encode_const_host produces for those symbols.  Buffers / status as in
 * cgs_gaussian_ac_encode. */
int cgs_bernoulli_ac_encode(const float *sym01, uint32_t c1,
                            const int64_t *stream_off, int n_streams,
                            uint8_t *out, const int64_t *out_off,
                            uint32_t *out_len, int32_t *status, void *stream);
int cgs_bernoulli_ac_decode(uint32_t c1, const int64_t *stream_off,
                            int n_streams, const uint8_t *in,
                            const int64_t *in_off, float *sym_out,
                            void *stream);
/* test hook: the integer CDF table [n, max_v-min_v+2] of one stream */
int cgs_gaussian_cdf_table(const float *mean, const float *scale,
                           const float *Q, int64_t q_div, int64_t n, int min_v,
                           int max_v, uint16_t *table, void *stream);

/* --- range-ANS for the hyper-prior symbols on the HOST
 * (EntropyBottleneck.compress / decompress, scene/gaussian_model.py:1088,
 * 1331).  symbols int32 [C, n] channel-major; cdf int32 [C, max_len] with
 * cdf_len[c] valid entries (last slot = escape), offset[c] = -minima. */
size_t cgs_rans_max_bytes(int64_t n_sym);
int cgs_rans_encode_host(const int32_t *symbols, int C, int64_t n,
                         const int32_t *cdf, int max_len,
                         const int32_t *cdf_len, const int32_t *offset,
                         int prec, uint8_t *out, size_t out_cap,
                         size_t *out_len);
int cgs_rans_decode_host(const uint8_t *in, size_t in_len, int C, int64_t n,
                         const int32_t *cdf, int max_len,
                         const int32_t *cdf_len, const int32_t *offset,
                         int prec, int32_t *symbols);
/* The same decode delivering the DEQUANTISED latents row-major by anchor: out_rows[i * ld_rows + c] =
 * (float)symbol + medians[c] (scene/gaussian_model.py:1326-1338 feeds exactly this to the context model); chunk jobs
 * write disjoint row ranges of one [N, C] host buffer. */
int cgs_rans_decode_rows_host(const uint8_t *in, size_t in_len, int C, int64_t n,
                              const int32_t *cdf, int max_len, const int32_t *cdf_len,
                              const int32_t *offset, int prec, const float *medians,
                              float *out_rows, int64_t ld_rows);

/* ------------------------------------------------------------------ */
/* Fused anchor MLPs + expansion (gaussian_renderer/__init__.py:106-145)  */
/* ------------------------------------------------------------------ */
/* The cgs_anchor_mlp3_*_rows + cgs_expand_* pipeline as one kernel family in
 * which the 110 MLP outputs per anchor, the 150-float hidden layer, the
 * [n,54] MLP input and all their gradients stay in registers / LDS (K must be
 * 10, the shape of the three heads).  Visible anchor r reads its feature row
 * feat_src[feat_row[r]] ([*,50]) and its geometry rows gs_src[geo_row[r]]
 * ([*,6]) / off_src[geo_row[r]] ([*,K*3]); either index may be NULL
 * (identity).  mask [n,K] = get_mask of the visible anchors.
 *   cgs_anchor_gen_count: opacity head + mask (:112-116,129-130) ->
 *     neural_opacity [n*K], y_op [n,K] (tanh output; NULL for inference),
 *     mask_out [n*K] bool bytes (may be NULL), bits [n] (bit k = slot k
 *     survives), base16 [ceil(n/16)+1] exclusive prefix of the survivor
 *     counts per 16 anchors (total last); *count_host = survivors P (the
 *     one stream synchronisation of boolean indexing, :137).
 *   cgs_anchor_gen_write: colour / covariance heads (W1..b2: HOST arrays of
 *     two device pointers: colour, covariance) + the compacted Gaussians
 *     xyz/color/scaling [P,3], opacity [P], rot [P,4], siginv [P,4] (saved
 *     for the backward; NULL for inference).
 *   cgs_anchor_gen_backward: gradients of everything above.  g_* [P,.];
 *     g_neural_opacity [n*K] or NULL.  W1/b1/W2 and dW2/db2: HOST arrays of
 *     three device pointers (opacity, colour, covariance).  Written:
 *     d_feat_src rows feat_row[r], d_gs / d_off rows geo_row[r] (other rows
 *     are the caller's to zero), d_anchor [n,3], d_mask [n,K]; weight / bias
 *     gradients are ACCUMULATED into dW1cat [150,54], db1cat [150], dW2[i],
 *     db2[i].  fused != 0: the weight gradients are formed inside the same
 *     kernel (hidden layer recomputed, nothing but per-workgroup partial sums
 *     leaves the chip); 0: operands written to scratch for a separate launch. */
int cgs_anchor_gen_count(const float *feat_src, const int64_t *feat_row,
                         const float *anchor_vis, const float *cam3, const float *mask,
                         const float *W1, const float *b1, const float *W2, const float *b2,
                         float *y_op, float *neural_opacity, uint8_t *mask_out,
                         uint32_t *bits, uint32_t *base16, int64_t n, int K,
                         int64_t *count_host, void *stream);
int cgs_anchor_gen_write(const float *feat_src, const int64_t *feat_row,
                         const float *anchor_vis, const float *cam3, const float *gs_src,
                         const float *off_src, const int64_t *geo_row,
                         const float *neural_opacity, const uint32_t *bits,
                         const uint32_t *base16, const float *const *W1,
                         const float *const *b1, const float *const *W2,
                         const float *const *b2, float *xyz, float *color, float *opacity,
                         float *scaling, float *rot, float *siginv, int64_t n, int K,
                         void *stream);
size_t cgs_anchor_gen_bwd_scratch_bytes(int64_t n, int fused);
int cgs_anchor_gen_backward(const float *feat_src, const int64_t *feat_row,
                            const float *anchor_vis, const float *cam3, const float *gs_src,
                            const float *off_src, const int64_t *geo_row, const float *mask,
                            const float *y_op, const uint32_t *bits, const uint32_t *base16,
                            const float *color, const float *rot, const float *siginv,
                            const float *g_xyz, const float *g_color, const float *g_opacity,
                            const float *g_scaling, const float *g_rot,
                            const float *g_neural_opacity, const float *const *W1,
                            const float *const *b1, const float *const *W2, float *d_feat_src,
                            float *d_anchor, float *d_gs, float *d_off, float *d_mask,
                            float *dW1cat, float *db1cat, float *const *dW2,
                            float *const *db2, int64_t n, int K, int fused, void *scratch,
                            size_t scratch_bytes, void *stream);

/* ------------------------------------------------------------------ */
/* Anchor -> Gaussian expansion (gaussian_renderer/__init__.py:112-145)   */
/* ------------------------------------------------------------------ */
/* Slots are (anchor n, offset k), i = n*K + k.  op_raw [n,K] is mlp_opacity's
 * output, mask [n,K] the binary offset mask, color_in [n,3K], cov_in [n,7K]
 * the raw mlp_color / mlp_cov outputs, gscaling [n,6], offsets [n,K,3].
 * cgs_expand_count writes neural_opacity [n*K], mask_out (bool bytes), the
 * survivor flags and their compacted rows (pos), and returns the survivor
 * count on the host; cgs_expand_write fills the compacted outputs
 * xyz/color/scaling [P,3], opacity [P], rot [P,4]; cgs_expand_backward
 * returns gradients for every differentiable input (all fully written).
 * src_row (may be NULL): anchor n reads gscaling / offsets row src_row[n] of a larger
 * array (the context model's coding-order output: the visibility gather is fused into
 * the kernel); d_gscaling / d_offsets then have that array's shape, rows src_row[n]
 * are written and the caller pre-zeroes the rest. */
/* SURVIVOR FLAGS are bytes (round 6): `mask_out` [n_anchor*K] (1 = the slot becomes a Gaussian) is what cgs_expand_write,
 * cgs_expand_backward and cgs_raster_preprocess_expand_launch take as `flags`, and what the count's scan runs over (a quarter of
 * the bytes of a uint32 array in three kernels).  The uint32 `flags` output of cgs_expand_count* is OPTIONAL (NULL: not written). */
size_t cgs_expand_scratch_bytes(int64_t n_anchor, int K);
int cgs_expand_count(int64_t n_anchor, int K, const float *op_raw,
                     const float *mask, float *neural_opacity,
                     uint8_t *mask_out, uint32_t *flags, uint32_t *pos,
                     void *scratch, size_t scratch_bytes, int64_t *count_host,
                     void *stream);
/* cgs_expand_count in two halves: _launch enqueues pass A + scan + the 4-byte copy of the count and returns;
 * _wait (same host thread) blocks on that copy alone, so kernels the caller enqueued in between keep the device busy. */
int cgs_expand_count_launch(int64_t n_anchor, int K, const float *op_raw,
                            const float *mask, float *neural_opacity,
                            uint8_t *mask_out, uint32_t *flags, uint32_t *pos,
                            void *scratch, size_t scratch_bytes, void *stream, uint64_t *ticket);
int cgs_expand_count_wait(uint64_t ticket, int64_t *count_host);
int cgs_expand_write(int64_t n_anchor, int K, const uint8_t *flags,
                     const uint32_t *pos, const float *anchor,
                     const float *gscaling, const float *offsets,
                     const float *neural_opacity, const float *color_in,
                     const float *cov_in, float *xyz, float *color,
                     float *opacity, float *scaling, float *rot,
                     const int64_t *src_row, void *stream);
int cgs_expand_backward(int64_t n_anchor, int K, const uint8_t *flags,
                        const uint32_t *pos, const float *gscaling,
                        const float *offsets, const float *op_raw,
                        const float *mask, const float *cov_in,
                        const float *g_xyz, const float *g_color,
                        const float *g_opacity, const float *g_scaling,
                        const float *g_rot, const float *g_neural_opacity,
                        float *d_anchor, float *d_gscaling, float *d_offsets,
                        float *d_op_raw, float *d_mask, float *d_color_in,
                        float *d_cov_in, const int64_t *src_row, void *stream);

/* The three clamp centres of the rate model (scene/gaussian_model.py:1664-1668: _anchor_feat.mean(),
 * get_scaling.mean(), _offset.mean()) in one launch: out3 = (mean a, mean (exp_b ? exp(b) : b), mean c),
 * accumulated in double, deterministic.  scratch from cgs_means3_scratch_bytes(). */
size_t cgs_means3_scratch_bytes(void);
int cgs_means3(const float *a, int64_t na, const float *b, int64_t nb, int exp_b,
               const float *c, int64_t nc, void *scratch, size_t scratch_bytes,
               float *out3, void *stream);

/* Offset-mask accessors of the model (scene/gaussian_model.py:295-310) in one pass:
 *   s = sigmoid(logits), mask = ((s > 0.01) - s) + s  (get_mask's straight-through value, [n,K])
 *   any_alive[a] = sum_k mask[a,k] > 0                (get_mask_anchor, uint8 [n])
 * Either output may be NULL.  Backward: d_logits = (g * (1 - s)) * s over all n*K elements. */
int cgs_mask_ste_fwd(const float *logits, int64_t n, int K, float *mask,
                     uint8_t *any_alive, void *stream);
int cgs_mask_ste_bwd(const float *logits, const float *g, int64_t n_elements,
                     float *d_logits, void *stream);

/* ---- anchor-initialisation kNN ----
 * `distCUDA2` of the simple_knn wheel (called at scene/gaussian_model.py:389,407; the wheel's source is not in
 * the reference checkout): mean_dist2[i] = mean of the squared fp32 distances from points[i] ([n,3], device) to
 * its 3 nearest OTHER points (exact search; coincident points count at distance 0).  Fewer than 4 points leave
 * FLT_MAX terms in the mean (inf), as an empty best-list does upstream.  scratch from cgs_knn_scratch_bytes(n). */
size_t cgs_knn_scratch_bytes(int64_t n);
int cgs_knn_mean_dist2(const float *points, int64_t n, float *mean_dist2,
                       void *scratch, size_t scratch_bytes, void *stream);

/* ---- densification statistics ----
 * scene/gaussian_model.py:696-713 (training_statis) in one pass over the n_vis*K slots of the visible
 * anchors vis_idx [n_vis] (ascending anchor rows): opacity_accum[a] += sum_k max(opacity[.,k], 0),
 * anchor_demon[a] += 1, and for every selected slot (sel != 0; sel_pos = its rank among the selected
 * = its Gaussian index) whose Gaussian passed update_filter: offset_gradient_accum[a*K+k] +=
 * ||grad[j, :2]||, offset_denom[a*K+k] += 1.  grad is [P,3] (viewspace_points.grad).  Accumulators are
 * the model's [N,1] / [N*K,1] fp32 buffers, updated in place. */
int cgs_densify_stats(int64_t n_vis, int K, const int64_t *vis_idx,
                      const float *opacity, const uint8_t *sel,
                      const int64_t *sel_pos, const uint8_t *update_filter,
                      const float *grad, float *opacity_accum, float *anchor_demon,
                      float *offset_gradient_accum, float *offset_denom,
                      void *stream);

/* ---- anchor pruning surgery (scene/gaussian_model.py:715-760, `_prune_anchor_optimizer` / `prune_anchor`, and the
 * statistics compaction of `adjust_anchor` :883-903) ----
 * dst[t][r, :] = src[t][idx[r], :] for r < n_keep, for nt <= 32 row-major fp32 tensors of widths width[t] in ONE
 * launch (the eight per-anchor parameters, their Adam moments, the statistics buffers; the reference boolean-indexes
 * each separately).  src / dst / width / clamp_col0 are HOST arrays.  clamp_col0[t] >= 0: columns >= clamp_col0[t] of
 * tensor t are capped at clamp_max on the way (the `scaling[:, 3:] > 0.05 -> 0.05` of :741-745); NULL = no clamps. */
int cgs_compact_rows(int nt, const float *const *src, float *const *dst, const int *width,
                     const int *clamp_col0, float clamp_max, const int64_t *idx, int64_t n_keep,
                     void *stream);

/* ---- visible-anchor list (gaussian_renderer/__init__.py:44-50: boolean-mask indexing = torch.nonzero) ----
 * Ascending indices of the non-zero bytes of mask [n] into idx_out [n] (first *count_host valid), in two halves:
 * _launch enqueues the kernels and the 4-byte copy of the count, _wait (same host thread) blocks on that copy alone,
 * so work enqueued in between keeps the device busy (torch.nonzero drains the stream). */
size_t cgs_nonzero_scratch_bytes(int64_t n);
int cgs_nonzero_launch(const uint8_t *mask, int64_t n, int64_t *idx_out, void *scratch,
                       size_t scratch_bytes, void *stream, uint64_t *ticket);
int cgs_nonzero_wait(uint64_t ticket, int64_t *count_host);

/* ---- level division of the context model: utils/multi_level.py:3-31
 * `torch_unique_with_indices` on the integer voxel keys of scene/gaussian_model.py:1751-1765 ----
 * cgs_level_key_range: out7 (device floats) = per-column min (3), max (3) of keys [n,3] and 1.0 if some value is not
 *   an integer; scratch8: 8 device ints.  The caller reads out7 (one host read) to choose the key widths.
 * cgs_level_unique: lo / bits = HOST arrays (3 each): column minima and key widths (each <= 31, sum <= 62).  Unique
 *   rows in ascending lexicographic order (-0.0 merged with 0.0); inverse [n] = unique row of every input row;
 *   first [n] = smallest input index of each group, counts [n], unique [n,3]: the first *n_unique_host entries are
 *   valid (one stream synchronisation for that count, as torch.unique has). */
int cgs_level_key_range(const float *keys, int64_t n, float *out7, void *scratch8, void *stream);
size_t cgs_level_unique_scratch_bytes(int64_t n);
int cgs_level_unique(const float *keys, int64_t n, const int32_t *lo, const int32_t *bits, int64_t *inverse,
                     int64_t *first, int64_t *counts, float *unique, int64_t *n_unique_host, void *scratch,
                     size_t scratch_bytes, void *stream);

/* ---- image loss of the training iteration ----
 * train.py:199-204 with utils/loss_utils.py:17-63: L1 = mean|img - gt| and SSIM (11x11 Gaussian
 * window, sigma 1.5, zero padding, per channel) of two [C,H,W] fp32 images, fused.
 * cgs_l1_ssim_fwd writes per-workgroup partial sums: partials [cgs_l1_ssim_partials(C,H,W), 2] =
 * (sum |img - gt|, sum ssim_map) — the caller adds them up and divides by C*H*W — and, for the
 * backward, the three partial-derivative maps [3,C,H,W] (maps may be NULL for evaluation).
 * cgs_l1_ssim_bwd: dimg = g[0] * dL1/dimg + g[1] * dSSIM/dimg with g [2] on the device (the
 * upstream gradients of the two MEANS; no host read). */
size_t cgs_l1_ssim_partials(int C, int H, int W);
int cgs_l1_ssim_fwd(const float *img, const float *gt, int C, int H, int W,
                    float *maps, float *partials, void *stream);
int cgs_l1_ssim_bwd(const float *img, const float *gt, const float *maps,
                    const float *g, int C, int H, int W, float *dimg,
                    void *stream);
/* train.py:199-204 as one value: out3 = (loss, L1, SSIM), loss = (1 - lam) L1 + lam (1 - SSIM), from cgs_l1_ssim_fwd's partials
 * (one single-workgroup launch, fixed summation order); and the backward with the gradient of `loss` read on the device
 * (g_loss [1]; g2 [2] = optional extra gradients of (L1, SSIM); either may be NULL, not both). */
int cgs_l1_ssim_finish(const float *partials, int C, int H, int W, float lam, float *out3,
                       void *stream);
int cgs_l1_ssim_bwd_loss(const float *img, const float *gt, const float *maps, const float *g_loss,
                         const float *g2, float lam, int C, int H, int W, float *dimg, void *stream);

/* sum_i img[i] * w[i] + lam * rate[0] (rate may be NULL) -> out [1], and its backward dimg = g[0] * w, drate[0] = g[0] * lam
 * (drate may be NULL): the linear objective a throughput measurement puts behind render() (train.py:206-209 with the image term
 * made linear), one launch each way, deterministic (per-workgroup partials in double, added in workgroup order).  scratch:
 * cgs_weighted_sum_scratch_bytes() bytes, ZERO-initialised once by the caller and reused across calls on one stream. */
size_t cgs_weighted_sum_scratch_bytes(void);
int cgs_weighted_sum_fwd(const float *img, const float *w, int64_t n, const float *rate, float lam,
                         void *scratch, size_t scratch_bytes, float *out, void *stream);
int cgs_weighted_sum_bwd(const float *g, const float *w, int64_t n, float lam, float *dimg,
                         float *drate, void *stream);

/* ---- the regularisers next to the image loss (train.py:203,209) ----
 * scaling_reg = scaling.prod(dim=1).mean() over the visible Gaussians' scales [P,3], and
 * torch.mean(torch.sigmoid(gaussians._mask)) over the n mask logits.  One streaming launch each way:
 * the forward writes cgs_reg_partials(n) per-workgroup partial SUMS (n = P or the element count;
 * the caller adds them up and divides by P / n), the backward takes the upstream gradient of the MEAN
 * as one float on the device (no host read; torch's prod backward reads an `input == 0` count back
 * on the host) and writes d scaling[i][c] = g / P * (product of the other two) / dx = g / n * s (1 - s).
 * Inputs and gradient outputs must be 16-byte aligned. */
size_t cgs_reg_partials(int64_t n);
int cgs_scaling_reg_fwd(const float *scaling, int64_t P, float *partials, void *stream);
int cgs_scaling_reg_bwd(const float *scaling, const float *g, int64_t P, float *d_scaling, void *stream);
int cgs_sigmoid_mean_fwd(const float *x, int64_t n, float *partials, void *stream);
int cgs_sigmoid_mean_bwd(const float *x, const float *g, int64_t n, float *dx, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CGS_H */
