// Anchor expansion fused with the rasterizer's preprocess stage (training path of render(),
// gaussian_renderer/__init__.py:130-145 -> :179-205).
//
// Unfused, the expansion writes five per-Gaussian tensors (xyz, colour, opacity, scaling, rotation: 56 B) that the
// rasterizer's preprocess reads straight back.  Here one kernel goes from the slots (mask flags, offsets, the anchor MLPs'
// outputs) to the rasterizer's records: slot -> (xyz, scaling, rot, colour, opacity) in registers -> cgs_pre_fwd_one ->
// record / depth key / tile rectangle / radius of compacted row pos[slot].  xyz, scaling and rot are also stored (the loss
// reads scaling, train.py:204; the backward's preprocess stage reads all three); colour and opacity never exist as tensors,
// and nothing is read back: 322 -> 25x us for the pair at 5.8 M Gaussians.
// The same device functions as the unfused kernels (csrc/raster_pre.h) on the same fp32 values: bit-identical records.
//
// The BACKWARD stays two kernels (blend backward -> preprocess_bwd_kernel -> expand_bwd_kernel).  One kernel from the blend
// backward's gradients to the per-slot gradients was built, parity-green, and measured SLOWER (750 us with a lane per slot,
// 887 us with the survivors compacted inside the workgroup, against 516 us for the two streaming kernels: the long dependent
// arithmetic between its loads and stores leaves too few memory requests in flight) — tools/experiments/
// expand_raster_bwd.hip.txt, profiles/r03_fused_view.txt.
#include "cgs_internal.h"
#include "raster_math.h"
#include "raster_pre.h"

#define XR_THREADS 256

struct XrSlot {
    float3 p, s;
    float4 q;
};

// the per-slot arithmetic of expand_write_kernel (csrc/expand.hip), same operation order
__device__ __forceinline__ XrSlot xr_slot(const float *__restrict__ anchor3, const float *__restrict__ gs, const float *__restrict__ off3,
                                          const float *__restrict__ sr) {
    XrSlot t;
    float pv[3], sv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        sv[c] = gs[3 + c] * (1.f / (1.f + __expf(-sr[c])));
        pv[c] = anchor3[c] + off3[c] * gs[c];
    }
    t.p = make_float3(pv[0], pv[1], pv[2]);
    t.s = make_float3(sv[0], sv[1], sv[2]);
    const float q0 = sr[3], q1 = sr[4], q2 = sr[5], q3 = sr[6];
    const float inv = 1.f / fmaxf(sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3), 1e-12f);   // F.normalize eps
    t.q = make_float4(q0 * inv, q1 * inv, q2 * inv, q3 * inv);
    return t;
}

__global__ void __launch_bounds__(XR_THREADS)
    expand_preprocess_kernel(int64_t n_slots, int K, const uint8_t *__restrict__ flags, const uint32_t *__restrict__ pos,
                             const float *__restrict__ anchor, const float *__restrict__ gscaling,
                             const float *__restrict__ offsets, const float *__restrict__ neural_opacity,
                             const float *__restrict__ color_in, const float *__restrict__ cov_in,
                             const int64_t *__restrict__ src_row, int W, int H, float tanfovx, float tanfovy,
                             float scale_modifier, const float *__restrict__ viewmatrix, const float *__restrict__ projmatrix,
                             float *__restrict__ scaling_out, float *__restrict__ xyz_out, float *__restrict__ rot_out,
                             float4 *__restrict__ rec, uint32_t *__restrict__ depth_key, uint32_t *__restrict__ tiles,
                             uint2 *__restrict__ rect, int32_t *__restrict__ radii) {
    const int64_t i = (int64_t)blockIdx.x * XR_THREADS + threadIdx.x;
    if (i >= n_slots || !flags[i]) return;
    const int64_t n = (int64_t)((uint32_t)i / (uint32_t)K);      // n_slots < 2^31 (cgs_expand_count_launch)
    const int64_t j = pos[i];
    const int64_t sn = src_row ? src_row[n] : n;
    const int64_t si = sn * K + (i - n * K);
    const XrSlot t = xr_slot(anchor + 3 * n, gscaling + 6 * sn, offsets + 3 * si, cov_in + 7 * i);
    scaling_out[3 * j] = t.s.x;
    scaling_out[3 * j + 1] = t.s.y;
    scaling_out[3 * j + 2] = t.s.z;
    if (xyz_out) {           // kept for the backward (cgs_raster_backward reads means3D / scales / rotations)
        xyz_out[3 * j] = t.p.x; xyz_out[3 * j + 1] = t.p.y; xyz_out[3 * j + 2] = t.p.z;
        rot_out[4 * j] = t.q.x; rot_out[4 * j + 1] = t.q.y; rot_out[4 * j + 2] = t.q.z; rot_out[4 * j + 3] = t.q.w;
    }
    float V[16], Pm[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { V[k] = viewmatrix[k]; Pm[k] = projmatrix[k]; }
    cgs_pre_fwd_one<false>(j, t.p, t.s, t.q, neural_opacity[i], color_in[3 * i], color_in[3 * i + 1], color_in[3 * i + 2], V, Pm, W,
                           H, tanfovx, tanfovy, scale_modifier, rec, depth_key, tiles, rect, radii);
}

int cgs_launch_expand_preprocess(const cgs_raster_cfg *cfg, const CgsExpandSrc &x, float *scaling_out, float *xyz_out,
                                 float *rot_out, CgsGeom &g, int32_t *radii, hipStream_t stream) {
    const int64_t n = x.n_anchor * x.K;
    if (n == 0) return CGS_OK;
    CgsProfScope prof(CGS_PROF_PREPROCESS, stream);
    hipLaunchKernelGGL(expand_preprocess_kernel, dim3((unsigned)((n + XR_THREADS - 1) / XR_THREADS)), dim3(XR_THREADS), 0, stream,
                       n, x.K, x.flags, x.pos, x.anchor, x.gscaling, x.offsets, x.neural_opacity, x.color_in, x.cov_in, x.src_row,
                       cfg->image_width, cfg->image_height, cfg->tanfovx, cfg->tanfovy, cfg->scale_modifier, cfg->viewmatrix,
                       cfg->projmatrix, scaling_out, xyz_out, rot_out, g.rec, g.depth_key, g.tiles, g.rect, radii);
    CGS_CHECK_LAUNCH(stream, cfg->debug);
    return CGS_OK;
}

