// SIMD time per wave64 VALU instruction on gfx950 by instruction class (16 independent chains per lane, inline asm so
// that the compiler neither packs nor removes anything).  Reported in 2.4 GHz cycles of SIMD time per instruction.
// hipcc -O3 --offload-arch=gfx950 -w tools/micro/valu_rate.hip -o tools/micro/valu_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHAINS 16
#define BODY(ASM)                                                                                   \
    for (int it = 0; it < iters; ++it) {                                                            \
        _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) { ASM; }                                 \
    }
template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a, float b) {
    float x[CHAINS];
    for (int i = 0; i < CHAINS; ++i) x[i] = (float)threadIdx.x * 1e-3f + i;
    const unsigned long long mk = __ballot(x[0] > b * 100.f);
    if (MODE == 0) BODY(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b)))
    if (MODE == 1) BODY(asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i])))
    if (MODE == 2) BODY(asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(x[i])))
    if (MODE == 3) BODY(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a)))
    if (MODE == 4) BODY(asm volatile("v_exp_f32 %0, %0" : "+v"(x[i])))
    if (MODE == 5) BODY(asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i])))
    if (MODE == 6) BODY(asm volatile("v_cmp_ge_f32 vcc, %0, %1" : : "v"(x[i]), "v"(a) : "vcc"))
    if (MODE == 7) BODY(asm volatile("v_min_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a)))
    if (MODE == 8) BODY(asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i])))
    if (MODE == 9) BODY(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "s"(a)))
    if (MODE == 10) BODY(asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a)))
    if (MODE == 11) BODY(asm volatile("v_cmp_ge_f32 %0, %1, %2" : "=s"(*(unsigned long long *)&x[0]) : "v"(x[i]), "v"(a)))
    if (MODE == 12) BODY(asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[i]) : "v"(a)))
    if (MODE == 13) BODY(asm volatile("v_ffbh_u32 %0, %0" : "+v"(x[i])))
    if (MODE == 14) BODY(asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x[i])))
    if (MODE == 16) BODY(asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "s"(mk)))
    if (MODE == 17) BODY(x[i] = (x[i] > a) ? x[i] * b : a)                      // compiler: v_cmp + v_mul + v_cndmask
    if (MODE == 18) BODY(asm volatile("v_cmp_ge_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a) : "vcc"))
    if (MODE == 19) BODY(asm volatile("v_cmp_ge_f32 vcc, %0, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a) : "vcc"))
    if (MODE == 20) BODY(asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(x[(i + 1) % CHAINS])))
    if (MODE == 21) BODY(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a)))
    if (MODE == 22) BODY(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(*(double *)&x[2 * (i / 2)]) : "v"(a), "v"(b) : "vcc"))
    if (MODE == 23) BODY(asm volatile("v_lshl_add_u32 %0, %0, 4, %1" : "+v"(x[i]) : "v"(a)))
    if (MODE == 24) BODY(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x[i]) : "v"(a)))
    if (MODE == 25) BODY(asm volatile("v_mov_b32 %0, 0" : "=v"(x[i])))
    if (MODE == 26) BODY(asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b)))
    if (MODE == 27) BODY(asm volatile("v_sub_f32 %0, %1, %0" : "+v"(x[i]) : "v"(a)))
    if (MODE == 28) BODY(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a)))
    if (MODE == 29) BODY(asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a)))
    if (MODE == 30) BODY(asm volatile("v_bfe_u32 %0, %0, 3, 4" : "+v"(x[i])))
    if (MODE == 31) BODY(asm volatile("v_fma_f32 %0, %0, %1, 2.0" : "+v"(x[i]) : "v"(a)))
    if (MODE == 15) BODY(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double *)&x[2 * (i / 2)]) : "v"(*(double *)&a)))
    float s = 0;
    for (int i = 0; i < CHAINS; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char *name, int wps) {
    const int blocks = 256 * wps, threads = 256, iters = 10000;
    float *out; hipMalloc(&out, sizeof(float) * blocks * threads);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, threads>>>(out, 100, 1.0001f, 0.5f);
    hipEventRecord(e0);
    k<MODE><<<blocks, threads>>>(out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("  %-28s %.2f\n", name, ms * 1e6 / iters / CHAINS * 2.4 / wps);
    hipFree(out);
}
int main() {
    for (int wps : {8}) {
        printf("-- %d waves per SIMD: cycles@2.4GHz of SIMD time per wave64 instruction\n", wps);
        run<0>("v_fma_f32", wps); run<10>("v_add_f32", wps); run<9>("v_mul_f32 (sgpr operand)", wps); run<7>("v_min_f32", wps);
        run<12>("v_and_b32", wps); run<3>("v_cndmask_b32", wps); run<6>("v_cmp_ge_f32 -> vcc", wps);
        run<1>("v_add_f32 dpp quad_perm", wps); run<2>("v_add_f32 dpp row_ror:4", wps); run<8>("v_mov_b32 dpp quad_perm", wps);
        run<16>("v_cndmask_b32_e64 sgpr mask", wps); run<17>("C select: cmp+mul+cndmask", wps);
        run<18>("asm cmp->vcc + cndmask", wps); run<19>("asm cmp + fma + cndmask", wps);
        run<20>("v_mov_b32 v,v", wps); run<25>("v_mov_b32 v,0", wps); run<21>("v_mul_lo_u32", wps); run<22>("v_mad_u64_u32", wps);
        run<23>("v_lshl_add_u32", wps); run<24>("v_mad_u32_u24", wps); run<26>("v_fmac_f32", wps); run<27>("v_sub_f32", wps);
        run<28>("v_mul_f32", wps); run<29>("v_max_f32", wps); run<30>("v_bfe_u32", wps); run<31>("v_fma_f32 inline const", wps);
        run<4>("v_exp_f32", wps); run<5>("v_rcp_f32", wps); run<13>("v_ffbh_u32", wps); run<14>("v_cvt_f32_u32", wps);
    }
    return 0;
}
