// Preprocess backward (R8, per-Gaussian half): chain dL/d(pixel mean) and
// dL/d(conic) from the blend backward to means3D / scales / rotations, and
// emit the means2D gradient in the NDC-scaled convention the reference's
// densification statistics consume (scene/gaussian_model.py:710,
// arguments/__init__.py:153).  Forward intermediates are recomputed from the
// 40-byte inputs instead of being stored (saves ~100 B/Gaussian of HBM
// traffic each way).
#include "cgs_internal.h"
#include "raster_math.h"
#include "raster_pre.h"

#define PB_THREADS 256

template <bool RAW>
__global__ void __launch_bounds__(PB_THREADS)
    preprocess_bwd_kernel(int64_t P, const float4 *__restrict__ rec, int W, int H, float tanfovx, float tanfovy, float scale_modifier,
                          const float *__restrict__ viewmatrix, const float *__restrict__ projmatrix,
                          const float *__restrict__ means3D, const float *__restrict__ scales,
                          const float *__restrict__ rotations, const int32_t *__restrict__ radii,
                          const float *__restrict__ dL_dmean2D_px, const float *__restrict__ dL_dconic,
                          float *__restrict__ dL_dmeans3D, float *__restrict__ dL_dmeans2D,
                          float *__restrict__ dL_dscales, float *__restrict__ dL_drotations) {
    const int64_t i = (int64_t)blockIdx.x * PB_THREADS + threadIdx.x;
    if (i >= P) return;
    if (radii[i] <= 0) {         // culled in forward: all gradients are zero (the arrays arrive uninitialised)
#pragma unroll
        for (int k = 0; k < 3; ++k) { dL_dmeans3D[3 * i + k] = 0.f; dL_dmeans2D[3 * i + k] = 0.f; dL_dscales[3 * i + k] = 0.f; }
#pragma unroll
        for (int k = 0; k < 4; ++k) dL_drotations[4 * i + k] = 0.f;
        return;
    }

    float V[16], Pm[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { V[k] = viewmatrix[k]; Pm[k] = projmatrix[k]; }
    const float3 p = make_float3(means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]);
    const float3 s_raw = make_float3(scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]);
    const float4 q = make_float4(rotations[4 * i], rotations[4 * i + 1], rotations[4 * i + 2],
                                 rotations[4 * i + 3]);

    const float op = RAW ? rec[3 * i + 1].y : 0.f;       // the record's opacity (the fused view path has no opacity tensor)
    const CgsPreBwd o = cgs_pre_bwd_one<RAW>(p, s_raw, q, dL_dmean2D_px[2 * i], dL_dmean2D_px[2 * i + 1], dL_dconic[3 * i],
                                             dL_dconic[3 * i + 1], dL_dconic[3 * i + 2], V, Pm, W, H, tanfovx, tanfovy,
                                             scale_modifier, op);
#pragma unroll
    for (int k = 0; k < 3; ++k) { dL_dmeans3D[3 * i + k] = o.dp[k]; dL_dmeans2D[3 * i + k] = o.dm2[k]; dL_dscales[3 * i + k] = o.ds[k]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) dL_drotations[4 * i + k] = o.dq[k];
}

// rec_raw != NULL: dL_dmean2D_px / dL_dconic hold the blend backward's RAW sums (CGS_BLEND_BWD_RAW), rec_raw = the geometry
// records (opacity in rec[3 i + 1].y)
int cgs_launch_preprocess_bwd(const cgs_raster_cfg *cfg, int64_t P, const float4 *rec_raw, const float *means3D, const float *scales,
                              const float *rotations, const int32_t *radii, const float *dL_dmean2D_px,
                              const float *dL_dconic, float *dL_dmeans3D, float *dL_dmeans2D,
                              float *dL_dscales, float *dL_drotations, hipStream_t stream) {
    if (P == 0) return CGS_OK;
    CgsProfScope prof(CGS_PROF_PREPROCESS_BWD, stream);
    if (rec_raw)
        hipLaunchKernelGGL(preprocess_bwd_kernel<true>, dim3((unsigned)((P + PB_THREADS - 1) / PB_THREADS)),
                           dim3(PB_THREADS), 0, stream, P, rec_raw, cfg->image_width, cfg->image_height, cfg->tanfovx,
                           cfg->tanfovy, cfg->scale_modifier, cfg->viewmatrix, cfg->projmatrix, means3D, scales,
                           rotations, radii, dL_dmean2D_px, dL_dconic, dL_dmeans3D, dL_dmeans2D, dL_dscales,
                           dL_drotations);
    else
        hipLaunchKernelGGL(preprocess_bwd_kernel<false>, dim3((unsigned)((P + PB_THREADS - 1) / PB_THREADS)),
                           dim3(PB_THREADS), 0, stream, P, rec_raw, cfg->image_width, cfg->image_height, cfg->tanfovx,
                           cfg->tanfovy, cfg->scale_modifier, cfg->viewmatrix, cfg->projmatrix, means3D, scales,
                           rotations, radii, dL_dmean2D_px, dL_dconic, dL_dmeans3D, dL_dmeans2D, dL_dscales,
                           dL_drotations);
    CGS_CHECK_LAUNCH(stream, cfg->debug);
    return CGS_OK;
}
