// Do fp32 MFMAs (v_mfma_f32_16x16x4_f32) and plain VALU work overlap on a gfx950 SIMD — (a) from two waves of one SIMD, (b) inside one
// wave's instruction stream?  Build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_valu_overlap.hip -o /tmp/mvo; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define ITERS 2000
// mode bit 0: this wave issues MFMAs; bit 1: VALU; roles by wave index: role = (mode >> (2 * (wave & 1) ... ))
__global__ void __launch_bounds__(512) k(int mode_even, int mode_odd, float *out, int interleave) {
    const int wave = threadIdx.x >> 6;
    const int mode = ((wave >> 2) & 1) ? mode_odd : mode_even;       // waves w and w + 4 share a SIMD
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
    const float x = 1.0001f, y = 0.5f;
    if (!interleave) {
        if (mode & 1)
            for (int i = 0; i < ITERS; ++i) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
            }
        if (mode & 2)
            for (int i = 0; i < ITERS; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {       // 32 independent fmas = the issue time of 4 MFMAs' pipe time
                    v0 = fmaf(v0, x, y); v1 = fmaf(v1, x, y); v2 = fmaf(v2, x, y); v3 = fmaf(v3, x, y);
                    v4 = fmaf(v4, x, y); v5 = fmaf(v5, x, y); v6 = fmaf(v6, x, y); v7 = fmaf(v7, x, y);
                }
            }
    } else {
        for (int i = 0; i < ITERS; ++i) {
#define STEP(acc)                                                                         \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc, 0, 0, 0);                       \
    v0 = fmaf(v0, x, y); v1 = fmaf(v1, x, y); v2 = fmaf(v2, x, y); v3 = fmaf(v3, x, y);   \
    v4 = fmaf(v4, x, y); v5 = fmaf(v5, x, y); v6 = fmaf(v6, x, y); v7 = fmaf(v7, x, y);   \
    __builtin_amdgcn_sched_barrier(0);
            STEP(a0) STEP(a1) STEP(a2) STEP(a3)
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}
static float run(int me, int mo, int il, float *d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, me, mo, d, il);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, me, mo, d, il);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5 * 1000;
}
int main() {
    float *d; hipMalloc(&d, 256 * 512 * 4);
    printf("per launch, us (2 waves per SIMD, %d iterations of 4 MFMAs / 32 fmas):\n", ITERS);
    printf("  MFMA in one wave, other idle        %8.1f\n", run(1, 0, 0, d));
    printf("  MFMA in both waves                  %8.1f\n", run(1, 1, 0, d));
    printf("  VALU in one wave, other idle        %8.1f\n", run(2, 0, 0, d));
    printf("  VALU in both waves                  %8.1f\n", run(2, 2, 0, d));
    printf("  MFMA in one wave, VALU in the other %8.1f\n", run(1, 2, 0, d));
    printf("  MFMA then VALU in each wave (serial)%8.1f\n", run(3, 3, 0, d));
    printf("  interleaved 1 MFMA : 8 fma, one wave busy (other idle) %8.1f\n", run(3, 0, 1, d));
    printf("  interleaved 1 MFMA : 8 fma, both waves                 %8.1f\n", run(3, 3, 1, d));
    return 0;
}
