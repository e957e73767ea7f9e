// Anchor -> neural-Gaussian expansion: the
// elementwise / compaction chain of generate_neural_gaussians
// (gaussian_renderer/__init__.py:112-145) as two streaming passes plus one
// backward pass, instead of ~25 torch kernels and ten [N_vis*K, 22]
// temporaries.  The three anchor MLPs are the fused fp32-MFMA kernels of
// csrc/mlp3.hip (rocBLAS, as the north star first suggested, was the round-1 v0
// path: three skinny GEMMs per head and a 75 ms step); this file consumes
// their raw outputs.
//
//   pass A  neural_opacity = mlp_opacity_out * mask;  flag = neural_opacity > 0
//           (then an exclusive scan of flag gives every surviving slot its
//            compacted row — same order as boolean indexing in the reference)
//   pass B  per surviving slot: colour copy, scaling = gs[3:6]*sigmoid(sr[0:3]),
//           rot = normalize(sr[3:7]), xyz = anchor + offset * gs[0:3]
//   bwd     one thread per anchor walks its K slots, scatters the per-slot
//           gradients and reduces the per-anchor ones in registers (no atomics)
#include "cgs_internal.h"

#define EX_THREADS 256

__global__ void __launch_bounds__(EX_THREADS)
    expand_flags_kernel(int64_t n_slots, const float *__restrict__ op_raw, const float *__restrict__ mask,
                        float *__restrict__ neural_opacity, uint32_t *__restrict__ flags,
                        uint8_t *__restrict__ mask_out) {
    const int64_t i = (int64_t)blockIdx.x * EX_THREADS + threadIdx.x;
    if (i >= n_slots) return;
    const float v = op_raw[i] * mask[i];
    neural_opacity[i] = v;
    const bool f = v > 0.f;
    if (flags) flags[i] = f ? 1u : 0u;         // (optional uint32 copy; the kernels of this library read the bytes of mask_out)
    mask_out[i] = f ? 1 : 0;
}

__global__ void __launch_bounds__(EX_THREADS)
    expand_write_kernel(int64_t n_slots, int K, const uint8_t *__restrict__ flags,
                        const uint32_t *__restrict__ pos, const float *__restrict__ anchor,
                        const float *__restrict__ gscaling, const float *__restrict__ offsets,
                        const float *__restrict__ neural_opacity, const float *__restrict__ color_in,
                        const float *__restrict__ cov_in, float *__restrict__ xyz, float *__restrict__ color,
                        float *__restrict__ opacity, float *__restrict__ scaling, float *__restrict__ rot,
                        const int64_t *__restrict__ src_row) {
    const int64_t i = (int64_t)blockIdx.x * EX_THREADS + threadIdx.x;
    if (i >= n_slots || !flags[i]) return;
    const int64_t n = i / K;
    const size_t j = pos[i];
    // src_row: gscaling / offsets live in a larger array (the context model's coding-order output) and anchor n
    // reads its row src_row[n] — the visibility gather fused into this kernel
    const int64_t sn = src_row ? src_row[n] : n;
    const int64_t si = sn * K + (i - n * K);
    const float *gs = gscaling + 6 * sn;
    const float *sr = cov_in + 7 * i;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        color[3 * j + c] = color_in[3 * i + c];
        scaling[3 * j + c] = gs[3 + c] * (1.f / (1.f + __expf(-sr[c])));
        xyz[3 * j + c] = anchor[3 * n + c] + offsets[3 * si + c] * gs[c];
    }
    opacity[j] = neural_opacity[i];
    const float q0 = sr[3], q1 = sr[4], q2 = sr[5], q3 = sr[6];
    const float inv = 1.f / fmaxf(sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3), 1e-12f);   // F.normalize eps
    rot[4 * j] = q0 * inv;
    rot[4 * j + 1] = q1 * inv;
    rot[4 * j + 2] = q2 * inv;
    rot[4 * j + 3] = q3 * inv;
}

#define EX_MAX_K 32
#define EX_APB 32          // anchors per workgroup in the backward

// One thread per slot (coalesced reads/writes of every per-slot array); the nine per-anchor sums
// (d_anchor 3, d_gscaling 6) are reduced over the anchor's K slots through LDS: every slot stores its nine terms
// (part[c][thread], conflict-free), then one thread per (anchor, sum) adds the K terms in slot order — deterministic,
// and no LDS float atomics (ten slots of an anchor hit the same address; ds_add_f32 retires < 1 lane per clock).
// Workgroup = EX_APB anchors x K slots; dynamic LDS = 9 * blockDim floats.
__global__ void __launch_bounds__(EX_APB * EX_MAX_K)
    expand_bwd_kernel(int64_t n_anchor, int K, const uint8_t *__restrict__ flags, const uint32_t *__restrict__ pos,
                      const float *__restrict__ gscaling, const float *__restrict__ offsets,
                      const float *__restrict__ op_raw, const float *__restrict__ mask,
                      const float *__restrict__ cov_in, const float *__restrict__ g_xyz,
                      const float *__restrict__ g_color, const float *__restrict__ g_opacity,
                      const float *__restrict__ g_scaling, const float *__restrict__ g_rot,
                      const float *__restrict__ g_neural_opacity /* may be null */,
                      float *__restrict__ d_anchor, float *__restrict__ d_gscaling, float *__restrict__ d_offsets,
                      float *__restrict__ d_op_raw, float *__restrict__ d_mask, float *__restrict__ d_color_in,
                      float *__restrict__ d_cov_in, const int64_t *__restrict__ src_row) {
    extern __shared__ float part[];            // [9][blockDim.x]
    __shared__ float sgs[EX_APB][6];
    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    const int64_t a0 = (int64_t)blockIdx.x * EX_APB;
    for (int t = tid; t < EX_APB * 6; t += blockDim.x) {
        const int64_t n = a0 + t / 6;
        sgs[t / 6][t % 6] = n < n_anchor ? gscaling[6 * (src_row ? src_row[n] : n) + t % 6] : 0.f;
    }
    __syncthreads();
    const int la = tid / K;                 // local anchor
    const int64_t n = a0 + la;
    const int64_t i = n * K + (tid - la * K);
    if (la < EX_APB && n < n_anchor) {
        const float *gs = sgs[la];
        const int64_t si = src_row ? src_row[n] * K + (tid - la * K) : i;
        float g_no = g_neural_opacity ? g_neural_opacity[i] : 0.f;
        float dcol[3] = {0.f, 0.f, 0.f}, doff[3] = {0.f, 0.f, 0.f}, dsr[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float term[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (flags[i]) {
            const size_t j = pos[i];
            g_no += g_opacity[j];
            const float *sr = cov_in + 7 * i;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                dcol[c] = g_color[3 * j + c];
                const float gx = g_xyz[3 * j + c];
                doff[c] = gx * gs[c];
                const float sig = 1.f / (1.f + __expf(-sr[c]));
                const float gsc = g_scaling[3 * j + c];
                dsr[c] = gsc * gs[3 + c] * sig * (1.f - sig);
                term[c] = gx;
                term[3 + c] = gx * offsets[3 * si + c];
                term[6 + c] = gsc * sig;
            }
            const float q0 = sr[3], q1 = sr[4], q2 = sr[5], q3 = sr[6];
            const float nrm = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
            const float inv = 1.f / fmaxf(nrm, 1e-12f);
            const float g0 = g_rot[4 * j], g1 = g_rot[4 * j + 1], g2 = g_rot[4 * j + 2], g3 = g_rot[4 * j + 3];
            if (nrm > 1e-12f) {
                const float r0 = q0 * inv, r1 = q1 * inv, r2 = q2 * inv, r3 = q3 * inv;
                const float dot = r0 * g0 + r1 * g1 + r2 * g2 + r3 * g3;
                dsr[3] = (g0 - r0 * dot) * inv;
                dsr[4] = (g1 - r1 * dot) * inv;
                dsr[5] = (g2 - r2 * dot) * inv;
                dsr[6] = (g3 - r3 * dot) * inv;
            } else {
                dsr[3] = g0 * inv; dsr[4] = g1 * inv; dsr[5] = g2 * inv; dsr[6] = g3 * inv;
            }
        }
        d_op_raw[i] = g_no * mask[i];
        d_mask[i] = g_no * op_raw[i];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            d_color_in[3 * i + c] = dcol[c];
            d_offsets[3 * si + c] = doff[c];
        }
#pragma unroll
        for (int c = 0; c < 7; ++c) d_cov_in[7 * i + c] = dsr[c];
#pragma unroll
        for (int c = 0; c < 9; ++c) part[c * nthr + tid] = term[c];
    }
    __syncthreads();
    for (int t = tid; t < EX_APB * 9; t += nthr) {
        const int c = t / EX_APB, a = t - c * EX_APB;       // neighbours differ in the anchor: LDS stride K
        const int64_t nn = a0 + a;
        if (nn < n_anchor) {
            const float *q = part + c * nthr + a * K;
            float sum = 0.f;
            for (int k = 0; k < K; ++k) sum += q[k];
            if (c < 3) d_anchor[3 * nn + c] = sum;
            else d_gscaling[6 * (src_row ? src_row[nn] : nn) + (c - 3)] = sum;
        }
    }
}

int cgs_scan_exclusive_u8_total(const uint8_t *in, uint32_t *out, int64_t n, void *scratch, size_t scratch_bytes,
                                uint32_t *grand_total, hipStream_t stream);

extern "C" size_t cgs_expand_scratch_bytes(int64_t n_anchor, int K) {
    return cgs_scan_scratch_bytes(n_anchor * (int64_t)K) + 256;
}

// Pass A + scan.  mask_out (survivor flags, one byte per slot) / pos (uint32) [n_anchor*K] are kept by the caller for pass B
// and the backward; the number of surviving Gaussians goes to the host through a pinned
// slot + event: cgs_expand_count_launch enqueues everything and returns, cgs_expand_count_wait
// blocks on THAT copy only (hipEventSynchronize), so work the caller enqueued in between keeps
// the GPU busy while the host learns the count (the reference's boolean indexing, :137, drains
// the whole stream at this point).  cgs_expand_count = launch + wait.
struct ExpandCountSlot { uint32_t *pinned; hipEvent_t ev; bool pending; uint64_t ticket; };
static thread_local ExpandCountSlot g_expand_slot = {nullptr, nullptr, false, 0};

extern "C" int cgs_expand_count_launch(int64_t n_anchor, int K, const float *op_raw, const float *mask,
                                       float *neural_opacity, uint8_t *mask_out, uint32_t *flags, uint32_t *pos,
                                       void *scratch, size_t scratch_bytes, void *stream_, uint64_t *ticket) {
    hipStream_t stream = (hipStream_t)stream_;
    ExpandCountSlot &sl = g_expand_slot;
    if (!ticket) { cgs_set_error("expand_count_launch: NULL ticket"); return CGS_ERR_ARG; }
    *ticket = 0;
    if (n_anchor < 0 || K < 1 || K > EX_MAX_K) { cgs_set_error("expand_count: bad args"); return CGS_ERR_ARG; }
    if (!sl.pinned) {
        CGS_CHECK_HIP(hipHostMalloc((void **)&sl.pinned, 64, hipHostMallocDefault));
        CGS_CHECK_HIP(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
    }
    sl.pending = false;
    sl.pinned[0] = 0;
    *ticket = sl.ticket = cgs_new_ticket(2);
    const int64_t n = n_anchor * K;
    if (n == 0) return CGS_OK;
    if (n >= (1ll << 31)) { cgs_set_error("expand_count: too many slots"); return CGS_ERR_ARG; }
    if (!op_raw || !mask || !neural_opacity || !mask_out || !pos || !scratch) {
        cgs_set_error("expand_count: NULL");
        return CGS_ERR_ARG;
    }
    hipLaunchKernelGGL(expand_flags_kernel, dim3((unsigned)((n + EX_THREADS - 1) / EX_THREADS)), dim3(EX_THREADS), 0,
                       stream, n, op_raw, mask, neural_opacity, flags, mask_out);
    CGS_CHECK_HIP(hipGetLastError());
    // grand total lands in the last 4 bytes of the scratch area
    if (scratch_bytes < cgs_expand_scratch_bytes(n_anchor, K)) { cgs_set_error("expand_count: scratch too small"); return CGS_ERR_WORKSPACE; }
    uint32_t *total = (uint32_t *)((char *)scratch + scratch_bytes - 256);
    int rc = cgs_scan_exclusive_u8_total(mask_out, pos, n, scratch, scratch_bytes - 256, total, stream);      // (bytes: a quarter of the reads)
    if (rc) return rc;
    CGS_CHECK_HIP(hipMemcpyAsync(sl.pinned, total, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    CGS_CHECK_HIP(hipEventRecord(sl.ev, stream));
    sl.pending = true;
    return CGS_OK;
}

extern "C" int cgs_expand_count_wait(uint64_t ticket, int64_t *count_host) {
    ExpandCountSlot &sl = g_expand_slot;
    if (!count_host) { cgs_set_error("expand_count_wait: NULL count_host"); return CGS_ERR_ARG; }
    *count_host = 0;
    if (!sl.pinned) { cgs_set_error("expand_count_wait: no launch on this thread"); return CGS_ERR_ARG; }
    if (ticket == 0 || ticket != sl.ticket) {
        cgs_set_error("expand_count_wait: stale ticket (another expand_count launch was issued on this thread since)");
        return CGS_ERR_ARG;
    }
    if (sl.pending) {
        CGS_CHECK_HIP(hipEventSynchronize(sl.ev));
        sl.pending = false;
    }
    *count_host = sl.pinned[0];
    return CGS_OK;
}

extern "C" int cgs_expand_count(int64_t n_anchor, int K, const float *op_raw, const float *mask,
                                float *neural_opacity, uint8_t *mask_out, uint32_t *flags, uint32_t *pos,
                                void *scratch, size_t scratch_bytes, int64_t *count_host, void *stream_) {
    if (!count_host) { cgs_set_error("expand_count: NULL count_host"); return CGS_ERR_ARG; }
    *count_host = 0;
    uint64_t ticket = 0;
    int rc = cgs_expand_count_launch(n_anchor, K, op_raw, mask, neural_opacity, mask_out, flags, pos, scratch, scratch_bytes,
                                     stream_, &ticket);
    if (rc) return rc;
    return cgs_expand_count_wait(ticket, count_host);
}

extern "C" int cgs_expand_write(int64_t n_anchor, int K, const uint8_t *flags, const uint32_t *pos,
                                const float *anchor, const float *gscaling, const float *offsets,
                                const float *neural_opacity, const float *color_in, const float *cov_in, float *xyz,
                                float *color, float *opacity, float *scaling, float *rot, const int64_t *src_row,
                                void *stream) {
    const int64_t n = n_anchor * K;
    if (n_anchor < 0 || K < 1 || K > EX_MAX_K) { cgs_set_error("expand_write: bad args"); return CGS_ERR_ARG; }
    if (n == 0) return CGS_OK;
    CgsProfScope prof(CGS_PROF_EXPAND_FWD, (hipStream_t)stream);
    hipLaunchKernelGGL(expand_write_kernel, dim3((unsigned)((n + EX_THREADS - 1) / EX_THREADS)), dim3(EX_THREADS), 0,
                       (hipStream_t)stream, n, K, flags, pos, anchor, gscaling, offsets, neural_opacity, color_in,
                       cov_in, xyz, color, opacity, scaling, rot, src_row);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

extern "C" int cgs_expand_backward(int64_t n_anchor, int K, const uint8_t *flags, const uint32_t *pos,
                                   const float *gscaling, const float *offsets, const float *op_raw,
                                   const float *mask, const float *cov_in, const float *g_xyz, const float *g_color,
                                   const float *g_opacity, const float *g_scaling, const float *g_rot,
                                   const float *g_neural_opacity, float *d_anchor, float *d_gscaling,
                                   float *d_offsets, float *d_op_raw, float *d_mask, float *d_color_in,
                                   float *d_cov_in, const int64_t *src_row, void *stream) {
    if (n_anchor < 0 || K < 1 || K > EX_MAX_K) { cgs_set_error("expand_backward: bad args"); return CGS_ERR_ARG; }
    if (n_anchor == 0) return CGS_OK;
    CgsProfScope prof(CGS_PROF_EXPAND_BWD, (hipStream_t)stream);
    hipLaunchKernelGGL(expand_bwd_kernel, dim3((unsigned)((n_anchor + EX_APB - 1) / EX_APB)),
                       dim3(EX_APB * K), (size_t)9 * EX_APB * K * sizeof(float), (hipStream_t)stream, n_anchor, K, flags, pos, gscaling, offsets, op_raw,
                       mask, cov_in, g_xyz, g_color, g_opacity, g_scaling, g_rot, g_neural_opacity, d_anchor,
                       d_gscaling, d_offsets, d_op_raw, d_mask, d_color_in, d_cov_in, src_row);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}
