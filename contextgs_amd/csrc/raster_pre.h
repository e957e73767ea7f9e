// Per-Gaussian bodies of the rasterizer's preprocess forward / backward as device functions, so that the plain kernels
// (csrc/raster_geom.hip, csrc/raster_bwd.hip) and the kernels fused with the anchor expansion (csrc/expand_raster.hip)
// execute the same instructions on the same values.
#pragma once
#include <hip/hip_fp16.h>
#include "cgs_internal.h"
#include "raster_math.h"

// One Gaussian of the rasterizer's preprocess stage (project, cov3D -> conic, radius, tile rectangle, record): the body of
// preprocess_kernel, shared with the kernel that takes its Gaussians straight from the anchor expansion
// (csrc/expand_raster.hip).  i = the Gaussian's row in every output array.
template <bool FILTER_ONLY>
__device__ __forceinline__ void cgs_pre_fwd_one(int64_t i, const float3 p, const float3 s, const float4 q, float op_in, float c0,
                                                float c1, float c2, const float *V, const float *Pm, int W, int H,
                                                float tanfovx, float tanfovy, float scale_modifier, float4 *__restrict__ rec,
                                                uint32_t *__restrict__ depth_key, uint32_t *__restrict__ tiles,
                                                uint2 *__restrict__ rect, int32_t *__restrict__ radii) {
    CgsProj pr;
    const bool ok = cgs_project<float>(p, s, q, V, Pm, W, H, tanfovx, tanfovy, scale_modifier, pr);

    int32_t radius = 0;
    uint32_t ntiles = 0;
    uint2 packed = make_uint2(0u, 0u);
    uint32_t dkey = 0xFFFFFFFFu;
    if (ok) {
        const int gx = (W + CGS_TILE - 1) / CGS_TILE, gy = (H + CGS_TILE - 1) / CGS_TILE;
        // reference tile rect: centre +- 3 sigma radius
        const float r = pr.radius;
        int x0 = min(gx, max(0, (int)((pr.px - r) / (float)CGS_TILE)));
        int y0 = min(gy, max(0, (int)((pr.py - r) / (float)CGS_TILE)));
        int x1 = min(gx, max(0, (int)((pr.px + r + (float)(CGS_TILE - 1)) / (float)CGS_TILE)));
        int y1 = min(gy, max(0, (int)((pr.py + r + (float)(CGS_TILE - 1)) / (float)CGS_TILE)));
        if ((x1 - x0) * (y1 - y0) > 0) {
            radius = (int32_t)r;
            if (!FILTER_ONLY) {
                const float op = op_in;
                // Output-invariant tightening: alpha >= 1/255 needs
                // 0.5 d^T conic d <= tau = ln(255 op); that ellipse's bounding box has
                // half extents sqrt(2 tau cov_xx), sqrt(2 tau cov_yy).  Pixels outside
                // it are skipped by the blend loop anyway, so tiles (and 8x8 quadrants)
                // outside it never need to see this Gaussian.
                float hx = -1.f, hy = -1.f;
                uint32_t diag = 0x7C007C00u;      // (+inf, +inf) as two halves: no diagonal cull
                const float t255 = 255.f * op;
                if (t255 >= 1.f) {
                    const float tau2 = 2.f * logf(t255);
                    hx = sqrtf(tau2 * pr.cov_a) * 1.002f + 0.02f;
                    hy = sqrtf(tau2 * pr.cov_c) * 1.002f + 0.02f;
                    // half extents of the same ellipse along x + y and x - y (the blend kernels cull 4x4 blocks against the
                    // octagon box /\ diagonals): sqrt(tau (1, +-1) cov (1, +-1)^T), padded like hx / hy and rounded UP to fp16
                    const float su = fmaxf(pr.cov_a + pr.cov_c + 2.f * pr.cov_b, 0.f);
                    const float sv = fmaxf(pr.cov_a + pr.cov_c - 2.f * pr.cov_b, 0.f);
                    const float hu = sqrtf(tau2 * su) * 1.002f + 0.03f, hv = sqrtf(tau2 * sv) * 1.002f + 0.03f;
                    diag = (uint32_t)__half_as_ushort(__float2half_ru(hu)) |
                           ((uint32_t)__half_as_ushort(__float2half_ru(hv)) << 16);
                    // pixels are at integer coordinates; first/last pixel inside the box
                    const float fx0 = ceilf(pr.px - hx), fx1 = floorf(pr.px + hx);
                    const float fy0 = ceilf(pr.py - hy), fy1 = floorf(pr.py + hy);
                    if (fx1 >= fx0 && fy1 >= fy0 && fx1 >= 0.f && fy1 >= 0.f && fx0 <= (float)(W - 1) &&
                        fy0 <= (float)(H - 1)) {
                        const int tx0 = max(0, (int)fx0) / CGS_TILE;
                        const int ty0 = max(0, (int)fy0) / CGS_TILE;
                        const int tx1 = min(W - 1, (int)fx1) / CGS_TILE + 1;
                        const int ty1 = min(H - 1, (int)fy1) / CGS_TILE + 1;
                        x0 = max(x0, tx0); y0 = max(y0, ty0);
                        x1 = min(x1, tx1); y1 = min(y1, ty1);
                    } else {
                        x1 = x0; y1 = y0;
                    }
                } else {
                    x1 = x0; y1 = y0;
                }
                if (x1 > x0 && y1 > y0) {
                    ntiles = (uint32_t)((x1 - x0) * (y1 - y0));
                    packed = make_uint2((uint32_t)x0 | ((uint32_t)y0 << 16), (uint32_t)x1 | ((uint32_t)y1 << 16));
                    dkey = __float_as_uint(pr.depth);
                }
                const float k = 1.4426950408889634f;  // log2(e): blend uses exp2
                rec[3 * i + 0] = make_float4(pr.px, pr.py, -0.5f * k * pr.con_a, -k * pr.con_b);
                rec[3 * i + 1] = make_float4(-0.5f * k * pr.con_c, op, c0, c1);
                rec[3 * i + 2] = make_float4(c2, hx, hy, __uint_as_float(diag));
            }
        }
    }
    radii[i] = radius;
    if (!FILTER_ONLY) {
        tiles[i] = ntiles;
        rect[i] = packed;
        depth_key[i] = dkey;
    }
}

// One Gaussian of the preprocess backward (the body of preprocess_bwd_kernel for radius > 0): p / s_raw / q are the forward's
// means3D / scales / rotations, gmean_* = dL/d(pixel mean), gconic_* = dL/d(conic) from the blend backward.  s below is
// s_raw * scale_modifier as in the kernel; o.ds is the gradient of s_raw.
// RAW (csrc/raster_blend_rows.hip, blend_bwd_rows_ga_kernel): the five inputs are the blend backward's raw sums a0..a4 =
// sum gx, sum gy, sum gx dx, sum gx dy, sum gy dy over the Gaussian's pixels and raw_op its opacity; the factors that turn
// them into dL/d(pixel mean) and dL/d(conic) are applied here, from the conic this function recomputes anyway:
//   dL/dmean = -op (con_a a0 + con_b a1, con_c a1 + con_b a0),  dL/dconic = -op (a2 / 2, a3, a4 / 2)
struct CgsPreBwd { float dp[3], dm2[3], ds[3], dq[4]; };
template <bool RAW = false>
__device__ __forceinline__ CgsPreBwd cgs_pre_bwd_one(const float3 p, const float3 s_raw, const float4 q, float gmean_x, float gmean_y,
                                                     float gconic_a, float gconic_b, float gconic_c, const float *V, const float *Pm,
                                                     int W, int H, float tanfovx, float tanfovy, float scale_modifier,
                                                     float raw_op = 0.f) {
    CgsPreBwd o;
    const float3 s = make_float3(s_raw.x * scale_modifier, s_raw.y * scale_modifier, s_raw.z * scale_modifier);
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
    if (RAW) {
        const float inv = det != 0.f ? 1.f / det : 0.f;
        const float con_a = z * inv, con_b = -y * inv, con_c = x * inv;      // as cgs_project forms them
        const float a0 = gmean_x, a1 = gmean_y;
        gmean_x = -raw_op * fmaf(con_a, a0, con_b * a1);
        gmean_y = -raw_op * fmaf(con_c, a1, con_b * a0);
        gconic_a *= -0.5f * raw_op;
        gconic_b *= -raw_op;
        gconic_c *= -0.5f * raw_op;
    }

    // ---- conic -> cov2D ----------------------------------------------------
    const float ga = gconic_a, gbb = gconic_b, gc = gconic_c;
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
    const float gnx = gmean_x * 0.5f * (float)W;   // = dL/d ndc_x
    const float gny = gmean_y * 0.5f * (float)H;
    const float hx = Pm[0] * p.x + Pm[4] * p.y + Pm[8] * p.z + Pm[12];
    const float hy = Pm[1] * p.x + Pm[5] * p.y + Pm[9] * p.z + Pm[13];
    const float hwv = Pm[3] * p.x + Pm[7] * p.y + Pm[11] * p.z + Pm[15];
    const float mw = 1.f / (hwv + 0.0000001f);
    const float mx = hx * mw * mw, my = hy * mw * mw;
    dpx += (Pm[0] * mw - Pm[3] * mx) * gnx + (Pm[1] * mw - Pm[3] * my) * gny;
    dpy += (Pm[4] * mw - Pm[7] * mx) * gnx + (Pm[5] * mw - Pm[7] * my) * gny;
    dpz += (Pm[8] * mw - Pm[11] * mx) * gnx + (Pm[9] * mw - Pm[11] * my) * gny;

    o.dp[0] = dpx;
    o.dp[1] = dpy;
    o.dp[2] = dpz;
    o.dm2[0] = gnx;
    o.dm2[1] = gny;
    o.dm2[2] = 0.f;

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
    o.ds[0] = ds[0] * scale_modifier;
    o.ds[1] = ds[1] * scale_modifier;
    o.ds[2] = ds[2] * scale_modifier;

    const float qr = q.x, qx = q.y, qy = q.z, qz = q.w;
    o.dq[0] = 2.f * (-qz * dR[1] + qy * dR[2] + qz * dR[3] - qx * dR[5] - qy * dR[6] + qx * dR[7]);
    o.dq[1] =
        2.f * (qy * dR[1] + qz * dR[2] + qy * dR[3] - 2.f * qx * dR[4] - qr * dR[5] + qz * dR[6] + qr * dR[7] -
               2.f * qx * dR[8]);
    o.dq[2] =
        2.f * (-2.f * qy * dR[0] + qx * dR[1] + qr * dR[2] + qx * dR[3] + qz * dR[5] - qr * dR[6] + qz * dR[7] -
               2.f * qy * dR[8]);
    o.dq[3] =
        2.f * (-2.f * qz * dR[0] - qr * dR[1] + qx * dR[2] + qr * dR[3] - 2.f * qz * dR[4] + qy * dR[5] +
               qx * dR[6] + qy * dR[7]);
    return o;
}
