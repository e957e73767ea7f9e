// Analysis hooks (experiment builds only, not part of include/cgs.h) and one test hook.  First: how many wave iterations the blend kernels need with the current
// mapping (one 8x8 quadrant per wave, one Gaussian per iteration) versus a mapping where the four 16-lane rows of
// a wave own one 4x4 pixel block each and walk their own Gaussian lists (four Gaussians per iteration).
// Uses only the bounding boxes of the alpha >= 1/255 ellipses, like the blend kernels' own quadrant masks.
#include "cgs_internal.h"


// ---- test hook: the per-tile lists of csrc/tile_bin.hip against the round-1 binning (emit_pairs + 32-bit pair sort) ----
// Re-bins the geometry of the last forward into a SECOND binning workspace with the round-1 path and counts the entries
// of gid_sorted and of the tile ranges that differ from what the forward left in bin_ws / img_ws (out2[0], out2[1]).
// tests/test_raster_gpu.py::test_tile_lists_equal_the_pair_sort; declared in include/cgs.h as a test hook.
__global__ void __launch_bounds__(256)
    dbg_count_diff_kernel(int64_t n, const uint32_t *__restrict__ a, const uint32_t *__restrict__ b,
                          unsigned long long *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && a[i] != b[i]) atomicAdd(out, 1ull);
}

extern "C" int cgs_debug_bin_compare(const cgs_raster_cfg *cfg, int64_t P, int64_t R, int64_t R_ws /* the count bin_ws was carved with */,
                                     void *geom_ws, size_t geom_bytes,
                                     void *bin_ws, size_t bin_bytes, void *img_ws, size_t img_bytes, void *bin_ws2,
                                     size_t bin_bytes2, void *ranges2 /* [tiles] uint2 */, int64_t *out2, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CgsGeom g;
    CgsBin b, b2;
    CgsImg im, im2;
    if (!cgs_img_carve(&im, img_ws, img_bytes, cfg->image_height, cfg->image_width) || !cgs_geom_carve(&g, geom_ws, geom_bytes, P) ||
        !cgs_bin_carve(&b, bin_ws, bin_bytes, P, R_ws) || !cgs_bin_carve(&b2, bin_ws2, bin_bytes2, P, R)) {
        cgs_set_error("debug_bin_compare: workspace");
        return CGS_ERR_WORKSPACE;
    }
    CGS_CHECK_HIP(hipMemsetAsync(out2, 0, 2 * sizeof(int64_t), stream));
    if (R == 0) return CGS_OK;
    int rc;
    if ((rc = cgs_launch_emit_pairs(cfg, P, g, b2, stream))) return rc;
    int bits = 0;
    while ((1u << bits) < (uint32_t)(cgs_tiles_x(cfg) * cgs_tiles_y(cfg))) ++bits;
    if ((rc = cgs_sort_pairs_u32(b2.tile_key_a, b2.gid_a, b2.tile_key_c, b2.gid_sorted, b2.tile_key_b, b2.gid_b, R, 0, bits,
                                 b2.scratch, b2.scratch_bytes, stream)))
        return rc;
    im2 = im;
    im2.ranges = (uint2 *)ranges2;
    if ((rc = cgs_launch_ranges(cfg, R, b2, im2, stream))) return rc;
    hipLaunchKernelGGL(dbg_count_diff_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, stream, R,
                       (const uint32_t *)b.gid_sorted, (const uint32_t *)b2.gid_sorted, (unsigned long long *)out2);
    const int64_t nr = 2ll * cgs_tiles_x(cfg) * cgs_tiles_y(cfg);
    hipLaunchKernelGGL(dbg_count_diff_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, stream, nr,
                       (const uint32_t *)im.ranges, (const uint32_t *)ranges2, (unsigned long long *)out2 + 1);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

