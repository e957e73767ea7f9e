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
    __shared__ unsigned int exact_cnt, oct_cnt, pix_cnt;
    unsigned long long e_visits = 0, o_visits = 0, p_hits = 0;
    for (uint32_t base = 0; base < tlast; base += 256) {
        if (tid < 20) cnt[tid / 5][tid % 5] = 0;
        if (tid == 0) { exact_cnt = 0; oct_cnt = 0; pix_cnt = 0; }
        __syncthreads();
        const uint32_t pos = base + tid;
        if (pos < tlast) {
            const uint32_t g = gid_sorted[range.x + pos];
            const float4 r0 = rec[3 * (size_t)g], r1 = rec[3 * (size_t)g + 1], r2 = rec[3 * (size_t)g + 2];
            // alpha >= 1/255  <=>  A dx^2 + B dx dy + C dy^2 >= thr (log2 domain, A/B/C pre-scaled and negative)
            const float cA = r0.z, cB = r0.w, cC = r1.x, thr = -log2f(255.f * r1.y);
            // diagonal half extents of the same ellipse: for q(d) = -(A dx^2 + B dx dy + C dy^2) <= -thr the extent
            // along u = x + y is sqrt(-thr * (iA + iC + 2 iB)) with the inverse form [[a, b/2],[b/2, c]]^-1
            const float a_ = -cA, b_ = -cB, c_ = -cC, det = a_ * c_ - 0.25f * b_ * b_;
            const float iA = c_ / det, iC = a_ / det, iB = -0.5f * b_ / det;
            const float hu = sqrtf(fmaxf(-thr * (iA + iC + 2.f * iB), 0.f)) * 1.002f + 0.03f;
            const float hv = sqrtf(fmaxf(-thr * (iA + iC - 2.f * iB), 0.f)) * 1.002f + 0.03f;
            const float gx = r0.x, gy = r0.y, hx = r2.y, hy = r2.z;
            for (int q = 0; q < 4; ++q) {
                const float x0 = (float)(tx * 16 + (q & 1) * 8), y0 = (float)(ty * 16 + (q >> 1) * 8);
                if ((gx - hx <= x0 + 7.f) && (gx + hx >= x0) && (gy - hy <= y0 + 7.f) && (gy + hy >= y0)) {
                    atomicAdd(&cnt[q][0], 1u);
                    for (int r = 0; r < 4; ++r) {
                        const float bx = x0 + (r & 1) * 4, by = y0 + (r >> 1) * 4;
                        if ((gx - hx <= bx + 3.f) && (gx + hx >= bx) && (gy - hy <= by + 3.f) && (gy + hy >= by)) {
                            atomicAdd(&cnt[q][1 + r], 1u);
                            bool any = false;
                            unsigned int hits = 0;
                            for (int py = 0; py < 4; ++py)
                                for (int px = 0; px < 4; ++px) {
                                    const float dx = gx - (bx + px), dy = gy - (by + py);
                                    const bool h = (cA * dx * dx + cB * dx * dy + cC * dy * dy) >= thr;
                                    any |= h;
                                    hits += h ? 1u : 0u;
                                }
                            if (any) atomicAdd(&exact_cnt, 1u);
                            atomicAdd(&pix_cnt, hits);
                            // octagon: the block's range of (x + y) and (x - y) against the diagonal extents
                            const float uc = gx + gy, vc = gx - gy;
                            const bool oct = (uc - hu <= bx + by + 6.f) && (uc + hu >= bx + by) && (vc - hv <= bx + 3.f - by) && (vc + hv >= bx - by - 3.f);
                            if (oct) atomicAdd(&oct_cnt, 1u);
                        }
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
        if (tid == 0) { e_visits += exact_cnt; o_visits += oct_cnt; p_hits += pix_cnt; }
        __syncthreads();
    }
    if (tid == 0) { atomicAdd(&out[0], q_iters); atomicAdd(&out[1], b_iters); atomicAdd(&out[2], b_visits); atomicAdd(&out[3], e_visits); atomicAdd(&out[4], o_visits); atomicAdd(&out[5], p_hits); }
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
    CGS_CHECK_HIP(hipMemsetAsync(out3, 0, 6 * sizeof(int64_t), (hipStream_t)stream));
    const int tx = cgs_tiles_x(cfg), ty = cgs_tiles_y(cfg);
    hipLaunchKernelGGL(blend_occupancy_kernel, dim3((unsigned)(tx * ty)), dim3(256), 0, (hipStream_t)stream, tx,
                       (const uint2 *)im.ranges, (const uint32_t *)b.gid_sorted, (const float4 *)g.rec,
                       (const uint32_t *)im.tile_last, (unsigned long long *)out3);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}
