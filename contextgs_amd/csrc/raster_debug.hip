// Analysis hook (not part of include/cgs.h): how many wave iterations the blend kernels need with the current
// mapping (one 8x8 quadrant per wave, one Gaussian per iteration) versus a mapping where the four 16-lane rows of
// a wave own one 4x4 pixel block each and walk their own Gaussian lists (four Gaussians per iteration).
// Uses only the bounding boxes of the alpha >= 1/255 ellipses, like the blend kernels' own quadrant masks.
#include "cgs_internal.h"

__global__ void __launch_bounds__(256)
    blend_occupancy_kernel(int tiles_x, const uint2 *__restrict__ ranges, const uint32_t *__restrict__ gid_sorted,
                           const float4 *__restrict__ rec, const uint32_t *__restrict__ tile_last,
                           unsigned long long *__restrict__ out) {
    __shared__ unsigned int cnt[4][5];     // per quadrant: [0] quadrant visits, [1..4] visits of its four 4x4 blocks
    const int tile = blockIdx.x, tid = threadIdx.x;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const uint32_t tlast = tile_last[tile];
    const uint2 range = ranges[tile];
    unsigned long long q_iters = 0, b_iters = 0, b_visits = 0;
    for (uint32_t base = 0; base < tlast; base += 256) {
        if (tid < 20) cnt[tid / 5][tid % 5] = 0;
        __syncthreads();
        const uint32_t pos = base + tid;
        if (pos < tlast) {
            const uint32_t g = gid_sorted[range.x + pos];
            const float4 r0 = rec[3 * (size_t)g], r2 = rec[3 * (size_t)g + 2];
            const float gx = r0.x, gy = r0.y, hx = r2.y, hy = r2.z;
            for (int q = 0; q < 4; ++q) {
                const float x0 = (float)(tx * 16 + (q & 1) * 8), y0 = (float)(ty * 16 + (q >> 1) * 8);
                if ((gx - hx <= x0 + 7.f) && (gx + hx >= x0) && (gy - hy <= y0 + 7.f) && (gy + hy >= y0)) {
                    atomicAdd(&cnt[q][0], 1u);
                    for (int r = 0; r < 4; ++r) {
                        const float bx = x0 + (r & 1) * 4, by = y0 + (r >> 1) * 4;
                        if ((gx - hx <= bx + 3.f) && (gx + hx >= bx) && (gy - hy <= by + 3.f) && (gy + hy >= by))
                            atomicAdd(&cnt[q][1 + r], 1u);
                    }
                }
            }
        }
        __syncthreads();
        if (tid == 0)
            for (int q = 0; q < 4; ++q) {
                q_iters += cnt[q][0];
                unsigned int mx = 0;
                for (int r = 0; r < 4; ++r) { mx = max(mx, cnt[q][1 + r]); b_visits += cnt[q][1 + r]; }
                b_iters += mx;
            }
        __syncthreads();
    }
    if (tid == 0) { atomicAdd(&out[0], q_iters); atomicAdd(&out[1], b_iters); atomicAdd(&out[2], b_visits); }
}

extern "C" int cgs_debug_blend_occupancy(const cgs_raster_cfg *cfg, int64_t P, int64_t R, void *geom_ws, size_t geom_bytes,
                                         void *bin_ws, size_t bin_bytes, void *img_ws, size_t img_bytes, int64_t *out3,
                                         void *stream) {
    CgsGeom g;
    CgsBin b;
    CgsImg im;
    if (!cgs_img_carve(&im, img_ws, img_bytes, cfg->image_height, cfg->image_width) || !cgs_geom_carve(&g, geom_ws, geom_bytes, P) ||
        !cgs_bin_carve(&b, bin_ws, bin_bytes, P, R)) {
        cgs_set_error("debug_blend_occupancy: workspace");
        return CGS_ERR_WORKSPACE;
    }
    CGS_CHECK_HIP(hipMemsetAsync(out3, 0, 3 * sizeof(int64_t), (hipStream_t)stream));
    const int tx = cgs_tiles_x(cfg), ty = cgs_tiles_y(cfg);
    hipLaunchKernelGGL(blend_occupancy_kernel, dim3((unsigned)(tx * ty)), dim3(256), 0, (hipStream_t)stream, tx,
                       (const uint2 *)im.ranges, (const uint32_t *)b.gid_sorted, (const float4 *)g.rec,
                       (const uint32_t *)im.tile_last, (unsigned long long *)out3);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}
