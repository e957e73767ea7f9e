// Per-Gaussian projection math shared by the forward preprocess, the
// visibility filter and the preprocess backward (which recomputes the forward
// intermediates instead of storing 100+ bytes per Gaussian in HBM).
//
// Conventions: matrices are row-major fp32 in the
// reference's ROW-vector convention, p' = [x y z 1] @ M.  Quaternions are
// (r, x, y, z) and are used as given.
#pragma once
#include <hip/hip_runtime.h>

struct CgsProj {
    float px, py;          // pixel-space centre
    float depth;           // view-space z
    float con_a, con_b, con_c;   // conic = inverse of the dilated 2-D covariance
    float cov_a, cov_b, cov_c;   // dilated 2-D covariance
    float radius;          // ceil(3 sigma_max)
};

struct CgsCov3 { float xx, xy, xz, yy, yz, zz; };

__device__ __forceinline__ void cgs_quat_to_rot(const float4 q, float R[9]) {
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z);       R[2] = 2.f * (x * z + r * y);
    R[3] = 2.f * (x * y + r * z);       R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
    R[6] = 2.f * (x * z - r * y);       R[7] = 2.f * (y * z + r * x);       R[8] = 1.f - 2.f * (x * x + y * y);
}

// Sigma = R diag(s^2) R^T
__device__ __forceinline__ CgsCov3 cgs_cov3d(const float3 s, const float R[9]) {
    const float sx = s.x * s.x, sy = s.y * s.y, sz = s.z * s.z;
    CgsCov3 c;
    c.xx = R[0] * R[0] * sx + R[1] * R[1] * sy + R[2] * R[2] * sz;
    c.xy = R[0] * R[3] * sx + R[1] * R[4] * sy + R[2] * R[5] * sz;
    c.xz = R[0] * R[6] * sx + R[1] * R[7] * sy + R[2] * R[8] * sz;
    c.yy = R[3] * R[3] * sx + R[4] * R[4] * sy + R[5] * R[5] * sz;
    c.yz = R[3] * R[6] * sx + R[4] * R[7] * sy + R[5] * R[8] * sz;
    c.zz = R[6] * R[6] * sx + R[7] * R[7] * sy + R[8] * R[8] * sz;
    return c;
}

// View-space position (row-vector convention).
__device__ __forceinline__ float3 cgs_to_view(const float3 p, const float *V) {
    return make_float3(V[0] * p.x + V[4] * p.y + V[8] * p.z + V[12],
                       V[1] * p.x + V[5] * p.y + V[9] * p.z + V[13],
                       V[2] * p.x + V[6] * p.y + V[10] * p.z + V[14]);
}

// A = J * Wv: 2x3 Jacobian of the (clamped) perspective map times the
// world->view rotation.  t is the view-space point, returns clamp flags so the
// backward pass can zero the derivative through a clamped coordinate.
struct CgsJac { float A[6]; float tx, ty, tz; bool clamp_x, clamp_y; float fx, fy; };

__device__ __forceinline__ CgsJac cgs_jacobian(const float3 t, const float *V, int W, int H, float tanfovx,
                                               float tanfovy) {
    CgsJac j;
    const float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
    const float txtz = t.x / t.z, tytz = t.y / t.z;
    j.clamp_x = (txtz < -limx) || (txtz > limx);
    j.clamp_y = (tytz < -limy) || (tytz > limy);
    j.tx = fminf(limx, fmaxf(-limx, txtz)) * t.z;
    j.ty = fminf(limy, fmaxf(-limy, tytz)) * t.z;
    j.tz = t.z;
    j.fx = (float)W / (2.f * tanfovx);
    j.fy = (float)H / (2.f * tanfovy);
    const float j00 = j.fx / j.tz, j02 = -(j.fx * j.tx) / (j.tz * j.tz);
    const float j11 = j.fy / j.tz, j12 = -(j.fy * j.ty) / (j.tz * j.tz);
    // Wv[i][c] = V[4*c + i]
    j.A[0] = j00 * V[0] + j02 * V[2];
    j.A[1] = j00 * V[4] + j02 * V[6];
    j.A[2] = j00 * V[8] + j02 * V[10];
    j.A[3] = j11 * V[1] + j12 * V[2];
    j.A[4] = j11 * V[5] + j12 * V[6];
    j.A[5] = j11 * V[9] + j12 * V[10];
    return j;
}

// cov2D = A Sigma A^T (before dilation)
__device__ __forceinline__ void cgs_cov2d(const float A[6], const CgsCov3 &c, float &a, float &b, float &cc) {
    const float u0 = c.xx * A[0] + c.xy * A[1] + c.xz * A[2];
    const float u1 = c.xy * A[0] + c.yy * A[1] + c.yz * A[2];
    const float u2 = c.xz * A[0] + c.yz * A[1] + c.zz * A[2];
    const float w0 = c.xx * A[3] + c.xy * A[4] + c.xz * A[5];
    const float w1 = c.xy * A[3] + c.yy * A[4] + c.yz * A[5];
    const float w2 = c.xz * A[3] + c.yz * A[4] + c.zz * A[5];
    a = A[0] * u0 + A[1] * u1 + A[2] * u2;
    b = A[3] * u0 + A[4] * u1 + A[5] * u2;
    cc = A[3] * w0 + A[4] * w1 + A[5] * w2;
}

template <typename T /*unused, float only*/>
__device__ __forceinline__ bool cgs_project(const float3 p, const float3 s, const float4 q, const float *V,
                                            const float *Pm, int W, int H, float tanfovx, float tanfovy,
                                            float scale_modifier, CgsProj &o) {
    const float3 t = cgs_to_view(p, V);
    if (t.z <= 0.2f) return false;   // near cull: the only frustum test
    const float hx = Pm[0] * p.x + Pm[4] * p.y + Pm[8] * p.z + Pm[12];
    const float hy = Pm[1] * p.x + Pm[5] * p.y + Pm[9] * p.z + Pm[13];
    const float hw = Pm[3] * p.x + Pm[7] * p.y + Pm[11] * p.z + Pm[15];
    const float pw = 1.f / (hw + 0.0000001f);
    const float ndcx = hx * pw, ndcy = hy * pw;

    float R[9];
    cgs_quat_to_rot(q, R);
    const float3 sm = make_float3(s.x * scale_modifier, s.y * scale_modifier, s.z * scale_modifier);
    const CgsCov3 c3 = cgs_cov3d(sm, R);
    const CgsJac j = cgs_jacobian(t, V, W, H, tanfovx, tanfovy);
    float a, b, c;
    cgs_cov2d(j.A, c3, a, b, c);
    a += 0.3f;
    c += 0.3f;
    const float det = a * c - b * b;
    if (det == 0.f) return false;
    const float inv = 1.f / det;
    o.con_a = c * inv; o.con_b = -b * inv; o.con_c = a * inv;
    o.cov_a = a; o.cov_b = b; o.cov_c = c;
    const float mid = 0.5f * (a + c);
    const float disc = sqrtf(fmaxf(0.1f, mid * mid - det));
    o.radius = ceilf(3.f * sqrtf(fmaxf(mid + disc, mid - disc)));
    o.px = ((ndcx + 1.f) * (float)W - 1.f) * 0.5f;
    o.py = ((ndcy + 1.f) * (float)H - 1.f) * 0.5f;
    o.depth = t.z;
    return true;
}
