// Byte-stream plumbing of the container codec: the device coder writes every chunk stream into its
// own worst-case slot; this packs the streams back to back (the on-disk layout of featN.b / scalingN.b / offsetsN.b,
// scene/gaussian_model.py:1235-1238) ON the device, so that only the real bytes cross PCIe and the host writes the
// packed buffer to the file as it is.
#include "cgs_internal.h"

__global__ void __launch_bounds__(256)
streams_compact_kernel(const uint8_t *__restrict__ src, const int64_t *__restrict__ src_off,
                       const uint32_t *__restrict__ len, const int64_t *__restrict__ dst_off, int S,
                       uint8_t *__restrict__ dst) {
    const int s = blockIdx.x;
    if (s >= S) return;
    const uint8_t *p = src + src_off[s];
    uint8_t *q = dst + dst_off[s];
    const uint32_t n = len[s];
    // head bytes until q is 4-byte aligned, then dword stores assembled from two aligned dword loads
    const uint32_t head = min(n, (uint32_t)((4 - ((uintptr_t)q & 3)) & 3));
    if (threadIdx.x < head) q[threadIdx.x] = p[threadIdx.x];
    const uint32_t body = (n - head) >> 2;
    const uint32_t sh = (uint32_t)(((uintptr_t)(p + head)) & 3) * 8;
    const uint32_t *pw = (const uint32_t *)((uintptr_t)(p + head) & ~(uintptr_t)3);   // slots are 8-byte aligned and padded
    uint32_t *qw = (uint32_t *)(q + head);
    for (uint32_t i = threadIdx.x; i < body; i += 256) {
        const uint32_t a = pw[i];
        qw[i] = sh ? (a >> sh) | (pw[i + 1] << (32 - sh)) : a;
    }
    const uint32_t done = head + (body << 2);
    if (threadIdx.x < n - done) q[done + threadIdx.x] = p[done + threadIdx.x];
}

extern "C" int cgs_streams_compact(const uint8_t *src, const int64_t *src_off, const uint32_t *len,
                                   const int64_t *dst_off, int n_streams, uint8_t *dst, void *stream) {
    if (n_streams < 0) { cgs_set_error("streams_compact: bad args"); return CGS_ERR_ARG; }
    if (n_streams == 0) return CGS_OK;
    hipLaunchKernelGGL(streams_compact_kernel, dim3(n_streams), dim3(256), 0, (hipStream_t)stream, src, src_off, len,
                       dst_off, n_streams, dst);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}
