// probe: raw buffer loads / stores of 2, 3, 4 dwords at 4-byte aligned offsets on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x3 __attribute__((ext_vector_type(3)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* p, float* q, int nbytes, int flags_variant) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t w = __builtin_amdgcn_make_buffer_rsrc((void*)q, 0, 4096, 0x00020000);
    int lane = threadIdx.x;
    int off = lane * 20 + 4;            // 4-byte aligned only
    i32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    i32x3 b = __builtin_amdgcn_raw_buffer_load_b96(r, off, 0, 0);
    i32x2 c = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
    int d = __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0);
    int base = lane * 64;
    __builtin_amdgcn_raw_buffer_store_b128(a, w, base, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b96(b, w, base + 16, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b64(c, w, base + 28, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(d, w, base + 36, 0, 0);
    // plain global store of the loaded registers, to tell a load problem from a store problem
    float* g = q + 1024 + lane * 4;
    g[0] = __builtin_bit_cast(float, a.x); g[1] = __builtin_bit_cast(float, a.y); g[2] = __builtin_bit_cast(float, a.z); g[3] = __builtin_bit_cast(float, a.w);
}
int main() {
    float h[1024], *p, *q, out[2048];
    for (int i = 0; i < 1024; ++i) h[i] = i;
    hipMalloc(&p, 4096); hipMalloc(&q, 8192);
    hipMemcpy(p, h, 4096, hipMemcpyHostToDevice); hipMemset(q, 0, 8192);
    k<<<1, 4>>>(p, q, 4096, 0);
    hipMemcpy(out, q, 8192, hipMemcpyDeviceToHost);
    for (int l = 0; l < 4; ++l) {
        printf("lane %d (first elem %d): b128->", l, l * 5 + 1);
        for (int i = 0; i < 10; ++i) printf(" %g", out[l * 16 + i]);
        printf(" | regs:");
        for (int i = 0; i < 4; ++i) printf(" %g", out[1024 + l * 4 + i]);
        printf("\n");
    }
    return 0;
}
