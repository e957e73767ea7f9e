// Issue rate of packed vs scalar fp32 VALU on gfx950: N independent FMA chains per lane, one wave per SIMD and four.
// hipcc -O3 --offload-arch=gfx950 tools/micro/pk_rate.hip -o /tmp/pk_rate && /tmp/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a, float b) {
    f2 x[8];
    for (int i = 0; i < 8; ++i) x[i] = (f2){(float)threadIdx.x + i, (float)i};
    const f2 A = {a, a}, B = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) {                 // 2 scalar fma (inline asm: -O3 would SLP-pack the C version into v_pk_fma_f32)
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i].x) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i].y) : "v"(a), "v"(b));
            }
            else if (MODE == 1) x[i] = __builtin_elementwise_fma(x[i], A, B);                                   // 1 packed fma
            else if (MODE == 2) x[i] = x[i] * A;                                                                // 1 packed mul
            else {
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i].x) : "v"(a));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i].y) : "v"(a));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char *name, int blocks, int threads) {
    float *out; hipMalloc(&out, sizeof(float) * blocks * threads);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, threads>>>(out, 100, 1.0001f, 0.5f);
    hipEventRecord(e0);
    k<MODE><<<blocks, threads>>>(out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double pair_ops = (double)iters * 8;            // per lane: (2 fp32 results) x 8 per iteration
    // SIMD time per fp32 PAIR result (one packed instruction or two scalar ones), in 2.4 GHz cycles
    const double wps = blocks * (threads / 64) / 1024.0;
    printf("%-14s %.3f ms -> %.2f cycles@2.4GHz of SIMD time per wave per (pair of fp32 results)\n", name, ms,
           ms * 1e6 / iters / 8.0 * 2.4 / wps);
    hipFree(out);
}
int main() {
    // waves per SIMD = blocks * 4 waves / (256 CUs * 4 SIMDs)
    for (int wps : {1, 2, 4, 8}) {
        printf("-- %d wave(s) per SIMD\n", wps);
        run<0>("2x v_fma_f32", 256 * wps, 256); run<1>("v_pk_fma_f32", 256 * wps, 256);
        run<3>("2x v_mul_f32", 256 * wps, 256); run<2>("v_pk_mul_f32", 256 * wps, 256);
    }
    return 0;
}
