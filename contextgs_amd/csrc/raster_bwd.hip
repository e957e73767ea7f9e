// Preprocess backward (R8, per-Gaussian half): chain dL/d(pixel mean) and
// dL/d(conic) from the blend backward to means3D / scales / rotations, and
// emit the means2D gradient in the NDC-scaled convention the reference's
// densification statistics consume (scene/gaussian_model.py:710,
// arguments/__init__.py:153).  Forward intermediates are recomputed from the
// 40-byte inputs instead of being stored (saves ~100 B/Gaussian of HBM
// traffic each way).
#include "cgs_internal.h"
#include "raster_math.h"

#define PB_THREADS 256

__global__ void __launch_bounds__(PB_THREADS)
    preprocess_bwd_kernel(int64_t P, int W, int H, float tanfovx, float tanfovy, float scale_modifier,
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
    const float3 s = make_float3(scales[3 * i] * scale_modifier, scales[3 * i + 1] * scale_modifier,
                                 scales[3 * i + 2] * scale_modifier);
    const float4 q = make_float4(rotations[4 * i], rotations[4 * i + 1], rotations[4 * i + 2],
                                 rotations[4 * i + 3]);

    // ---- recompute forward intermediates ---------------------------------
    const float3 t = cgs_to_view(p, V);
    float R[9];
    cgs_quat_to_rot(q, R);
    const CgsCov3 c3 = cgs_cov3d(s, R);
    const CgsJac j = cgs_jacobian(t, V, W, H, tanfovx, tanfovy);
    float x, y, z;   // dilated cov2D = [[x,y],[y,z]]
    cgs_cov2d(j.A, c3, x, y, z);
    x += 0.3f;
    z += 0.3f;
    const float det = x * z - y * y;

    // ---- conic -> cov2D ----------------------------------------------------
    const float ga = dL_dconic[3 * i], gbb = dL_dconic[3 * i + 1], gc = dL_dconic[3 * i + 2];
    float gx = 0.f, gy = 0.f, gz = 0.f;   // dL/d(x,y,z), y = full derivative of the repeated entry
    if (det != 0.f) {
        const float d2 = 1.f / (det * det);
        gx = d2 * (-z * z * ga + y * z * gbb - y * y * gc);
        gy = d2 * (2.f * y * z * ga - (x * z + y * y) * gbb + 2.f * x * y * gc);
        gz = d2 * (-y * y * ga + x * y * gbb - x * x * gc);
    }
    // symmetric matrix form G2 = [[gx, gy/2],[gy/2, gz]]
    const float h = 0.5f * gy;
    const float *A = j.A;

    // ---- cov2D = A Sigma A^T: dL/dSigma = A^T G2 A (matrix form) ------------
    // rows of G2*A
    float GA0[3], GA1[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        GA0[c] = gx * A[c] + h * A[3 + c];
        GA1[c] = h * A[c] + gz * A[3 + c];
    }
    float M[9];   // dL/dSigma as a full symmetric matrix
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) M[3 * r + c] = A[r] * GA0[c] + A[3 + r] * GA1[c];

    // ---- dL/dA = 2 G2 A Sigma ------------------------------------------------
    const float S[9] = {c3.xx, c3.xy, c3.xz, c3.xy, c3.yy, c3.yz, c3.xz, c3.yz, c3.zz};
    float dA[6];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        dA[c] = 2.f * (GA0[0] * S[c] + GA0[1] * S[3 + c] + GA0[2] * S[6 + c]);
        dA[3 + c] = 2.f * (GA1[0] * S[c] + GA1[1] * S[3 + c] + GA1[2] * S[6 + c]);
    }
    // A = J Wv, Wv[i][c] = V[4c+i]:  dL/dJ[r][i] = sum_c dA[r][c] Wv[i][c]
    const float dJ00 = dA[0] * V[0] + dA[1] * V[4] + dA[2] * V[8];
    const float dJ02 = dA[0] * V[2] + dA[1] * V[6] + dA[2] * V[10];
    const float dJ11 = dA[3] * V[1] + dA[4] * V[5] + dA[5] * V[9];
    const float dJ12 = dA[3] * V[2] + dA[4] * V[6] + dA[5] * V[10];
    const float tz = 1.f / j.tz, tz2 = tz * tz, tz3 = tz2 * tz;
    const float dtx = j.clamp_x ? 0.f : (-j.fx * tz2 * dJ02);
    const float dty = j.clamp_y ? 0.f : (-j.fy * tz2 * dJ12);
    const float dtz = -j.fx * tz2 * dJ00 - j.fy * tz2 * dJ11 + (2.f * j.fx * j.tx) * tz3 * dJ02 +
                      (2.f * j.fy * j.ty) * tz3 * dJ12;
    // t = Wv p + trans: dL/dp_c = sum_i Wv[i][c] dL/dt_i
    float dpx = V[0] * dtx + V[1] * dty + V[2] * dtz;
    float dpy = V[4] * dtx + V[5] * dty + V[6] * dtz;
    float dpz = V[8] * dtx + V[9] * dty + V[10] * dtz;

    // ---- projection path: pixel = ((ndc+1) W - 1)/2 -----------------------------
    const float gnx = dL_dmean2D_px[2 * i] * 0.5f * (float)W;   // = dL/d ndc_x
    const float gny = dL_dmean2D_px[2 * i + 1] * 0.5f * (float)H;
    const float hx = Pm[0] * p.x + Pm[4] * p.y + Pm[8] * p.z + Pm[12];
    const float hy = Pm[1] * p.x + Pm[5] * p.y + Pm[9] * p.z + Pm[13];
    const float hwv = Pm[3] * p.x + Pm[7] * p.y + Pm[11] * p.z + Pm[15];
    const float mw = 1.f / (hwv + 0.0000001f);
    const float mx = hx * mw * mw, my = hy * mw * mw;
    dpx += (Pm[0] * mw - Pm[3] * mx) * gnx + (Pm[1] * mw - Pm[3] * my) * gny;
    dpy += (Pm[4] * mw - Pm[7] * mx) * gnx + (Pm[5] * mw - Pm[7] * my) * gny;
    dpz += (Pm[8] * mw - Pm[11] * mx) * gnx + (Pm[9] * mw - Pm[11] * my) * gny;

    dL_dmeans3D[3 * i] = dpx;
    dL_dmeans3D[3 * i + 1] = dpy;
    dL_dmeans3D[3 * i + 2] = dpz;
    dL_dmeans2D[3 * i] = gnx;
    dL_dmeans2D[3 * i + 1] = gny;
    dL_dmeans2D[3 * i + 2] = 0.f;

    // ---- Sigma = (R S)(R S)^T: dL/d(RS) = 2 M (R S) --------------------------------
    const float sv[3] = {s.x, s.y, s.z};
    float dR[9];
    float ds[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            // dL/dMrs[r][k] = 2 sum_c M[r][c] * (R[c][k] s_k)
            const float dm = 2.f * sv[k] * (M[3 * r] * R[k] + M[3 * r + 1] * R[3 + k] + M[3 * r + 2] * R[6 + k]);
            ds[k] += dm * R[3 * r + k];
            dR[3 * r + k] = dm * sv[k];
        }
    dL_dscales[3 * i] = ds[0] * scale_modifier;
    dL_dscales[3 * i + 1] = ds[1] * scale_modifier;
    dL_dscales[3 * i + 2] = ds[2] * scale_modifier;

    const float qr = q.x, qx = q.y, qy = q.z, qz = q.w;
    dL_drotations[4 * i] = 2.f * (-qz * dR[1] + qy * dR[2] + qz * dR[3] - qx * dR[5] - qy * dR[6] + qx * dR[7]);
    dL_drotations[4 * i + 1] =
        2.f * (qy * dR[1] + qz * dR[2] + qy * dR[3] - 2.f * qx * dR[4] - qr * dR[5] + qz * dR[6] + qr * dR[7] -
               2.f * qx * dR[8]);
    dL_drotations[4 * i + 2] =
        2.f * (-2.f * qy * dR[0] + qx * dR[1] + qr * dR[2] + qx * dR[3] + qz * dR[5] - qr * dR[6] + qz * dR[7] -
               2.f * qy * dR[8]);
    dL_drotations[4 * i + 3] =
        2.f * (-2.f * qz * dR[0] - qr * dR[1] + qx * dR[2] + qr * dR[3] - 2.f * qz * dR[4] + qy * dR[5] +
               qx * dR[6] + qy * dR[7]);
}

int cgs_launch_preprocess_bwd(const cgs_raster_cfg *cfg, int64_t P, const float *means3D, const float *scales,
                              const float *rotations, const int32_t *radii, const float *dL_dmean2D_px,
                              const float *dL_dconic, float *dL_dmeans3D, float *dL_dmeans2D,
                              float *dL_dscales, float *dL_drotations, hipStream_t stream) {
    if (P == 0) return CGS_OK;
    CgsProfScope prof(CGS_PROF_PREPROCESS_BWD, stream);
    hipLaunchKernelGGL(preprocess_bwd_kernel, dim3((unsigned)((P + PB_THREADS - 1) / PB_THREADS)),
                       dim3(PB_THREADS), 0, stream, P, cfg->image_width, cfg->image_height, cfg->tanfovx,
                       cfg->tanfovy, cfg->scale_modifier, cfg->viewmatrix, cfg->projmatrix, means3D, scales,
                       rotations, radii, dL_dmean2D_px, dL_dconic, dL_dmeans3D, dL_dmeans2D, dL_dscales,
                       dL_drotations);
    CGS_CHECK_LAUNCH(stream, cfg->debug);
    return CGS_OK;
}
