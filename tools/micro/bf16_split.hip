// fp32 MLP layers on the fp32 MFMA (v_mfma_f32_16x16x4_f32) against the same layers on the bf16 MFMA with every fp32 operand
// split into three bf16 pieces (hi + mid + lo) and 9 / 6 / 3 cross products accumulated in fp32 (v_mfma_f32_16x16x32_bf16):
// time per launch and error against an fp64 evaluation.  L chained 64 -> 64 layers with ReLU (weights from LDS, activations
// chained through the accumulator layout as in csrc/mlp_frag.h), n rows.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/bf16_split.hip -o tools/micro/bf16_split.bin && tools/micro/bf16_split.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef f32x4 f32x4_a4 __attribute__((aligned(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef f32x2 f32x2_a8 __attribute__((aligned(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
#define D 64
#define LAYERS 6
#define WAVES 8

__device__ __forceinline__ unsigned short bf16_rne(float x) {
    unsigned int u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f(unsigned short h) { return __uint_as_float((unsigned int)h << 16); }

// ---- fp32 MFMA: k-step (q, j) contracts features {16 q + 4 g + j}; lane (g, c) holds features 16 q + 4 g + {0..3} of row c
template <int RT, int NW = WAVES, bool STORE_H = false, int HPAD = 0, int VEC = 4>
__global__ void __launch_bounds__(NW * 64) k_f32(const float *__restrict__ X, const float *__restrict__ W, float *__restrict__ Y, long n,
                                                 float *__restrict__ Hall = nullptr) {
    __shared__ float Ws[LAYERS][D][D + 4];           // [layer][in][out]
    for (int i = threadIdx.x; i < LAYERS * D * D; i += NW * 64) Ws[i / (D * D)][(i / D) % D][i % D] = W[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    const long ntiles = (n + 16 * RT - 1) / (16 * RT);
    for (long tile = (long)blockIdx.x * NW + wave; tile < ntiles; tile += (long)gridDim.x * NW) {
        asm volatile("" ::: "memory");      // keep the LDS weight reads inside the tile loop (hoisted, they spill: csrc/mlp3.hip)
        f32x4 x[RT][4];
        for (int rt = 0; rt < RT; ++rt)
            for (int q = 0; q < 4; ++q) x[rt][q] = *(const f32x4 *)(X + (tile * 16 * RT + rt * 16 + c) * D + 16 * q + 4 * g);
        for (int l = 0; l < LAYERS; ++l) {
            f32x4 acc[RT][4];
            for (int rt = 0; rt < RT; ++rt) for (int t = 0; t < 4; ++t) acc[rt][t] = (f32x4){0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float a = Ws[l][16 * q + 4 * g + j][16 * t + c];
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, x[rt][q][j], acc[rt][t], 0, 0, 0);
                    }
            for (int rt = 0; rt < RT; ++rt) for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r) x[rt][t][r] = fmaxf(acc[rt][t][r], 0.f);
            if (STORE_H)      // every hidden layer goes to memory too (a [n, LAYERS * 64] buffer: 1.5 KB per row, like Hcat + Y)
                for (int rt = 0; rt < RT; ++rt)
                    for (int q = 0; q < 4; ++q)
                    {
                        float *hp = Hall + (tile * 16 * RT + rt * 16 + c) * (LAYERS * D + HPAD) + l * D + 16 * q + 4 * g;
                        if (VEC == 4) *(f32x4_a4 *)hp = x[rt][q];
                        else if (VEC == 2) { *(f32x2_a8 *)hp = (f32x2){x[rt][q][0], x[rt][q][1]}; *(f32x2_a8 *)(hp + 2) = (f32x2){x[rt][q][2], x[rt][q][3]}; }
                        else { hp[0] = x[rt][q][0]; hp[1] = x[rt][q][1]; hp[2] = x[rt][q][2]; hp[3] = x[rt][q][3]; }
                    }
        }
        for (int rt = 0; rt < RT; ++rt)
            for (int q = 0; q < 4; ++q) *(f32x4 *)(Y + (tile * 16 * RT + rt * 16 + c) * D + 16 * q + 4 * g) = x[rt][q];
    }
}

// ---- bf16 split: K-block kb (32 contraction indices) = feature tiles (2 kb, 2 kb + 1); index 8 g + j' <-> feature
// 16 (2 kb + (j' >> 2)) + 4 g + (j' & 3): the lane's own accumulator values are its B operand.  NP = 9, 6 or 3 products.
template <int RT, int NP>
__global__ void __launch_bounds__(WAVES * 64) k_bf16(const float *__restrict__ X, const float *__restrict__ W, float *__restrict__ Y, long n) {
    __shared__ u16x8 Ws[LAYERS][2][4][3][64];        // [layer][kb][out tile][piece][lane] = 8 bf16 of the A operand
    for (int i = threadIdx.x; i < LAYERS * 2 * 4 * 64; i += WAVES * 64) {
        const int ln = i & 63, t = (i >> 6) & 3, kb = (i >> 8) & 1, l = i >> 9;
        const int gg = ln >> 4, m = ln & 15;
        u16x8 p0, p1, p2;
        for (int jj = 0; jj < 8; ++jj) {
            const int feat = 16 * (2 * kb + (jj >> 2)) + 4 * gg + (jj & 3);
            const float w = W[(l * D + feat) * D + 16 * t + m];
            const unsigned short h = bf16_rne(w); const float r1 = w - bf16_f(h);
            const unsigned short md = bf16_rne(r1); const float r2 = r1 - bf16_f(md);
            p0[jj] = h; p1[jj] = md; p2[jj] = bf16_rne(r2);
        }
        Ws[l][kb][t][0][ln] = p0; Ws[l][kb][t][1][ln] = p1; Ws[l][kb][t][2][ln] = p2;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    const long ntiles = (n + 16 * RT - 1) / (16 * RT);
    for (long tile = (long)blockIdx.x * WAVES + wave; tile < ntiles; tile += (long)gridDim.x * WAVES) {
        asm volatile("" ::: "memory");
        f32x4 x[RT][4];
        for (int rt = 0; rt < RT; ++rt)
            for (int q = 0; q < 4; ++q) x[rt][q] = *(const f32x4 *)(X + (tile * 16 * RT + rt * 16 + c) * D + 16 * q + 4 * g);
        for (int l = 0; l < LAYERS; ++l) {
            f32x4 acc[RT][4];
            for (int rt = 0; rt < RT; ++rt) for (int t = 0; t < 4; ++t) acc[rt][t] = (f32x4){0, 0, 0, 0};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                u16x8 b[RT][3];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const float v = x[rt][2 * kb + (jj >> 2)][jj & 3];
                        const unsigned short h = bf16_rne(v); const float r1 = v - bf16_f(h);
                        const unsigned short md = bf16_rne(r1); const float r2 = r1 - bf16_f(md);
                        b[rt][0][jj] = h; b[rt][1][jj] = md; b[rt][2][jj] = bf16_rne(r2);
                    }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    u16x8 a[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) a[p] = Ws[l][kb][t][p][lane];
                    // products ordered small to large: (lo,lo) (lo,mid) (mid,lo) | (mid,mid) (hi,lo) (lo,hi) | (hi,mid) (mid,hi) (hi,hi)
                    const int pa[9] = {2, 2, 1, 1, 0, 2, 0, 1, 0}, pb[9] = {2, 1, 2, 1, 2, 0, 1, 0, 0};
#pragma unroll
                    for (int k = 9 - NP; k < 9; ++k)
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt)
                            acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[pa[k]]),
                                                                                  __builtin_bit_cast(bf16x8, b[rt][pb[k]]), acc[rt][t], 0, 0, 0);
                }
            }
            for (int rt = 0; rt < RT; ++rt) for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r) x[rt][t][r] = fmaxf(acc[rt][t][r], 0.f);
        }
        for (int rt = 0; rt < RT; ++rt)
            for (int q = 0; q < 4; ++q) *(f32x4 *)(Y + (tile * 16 * RT + rt * 16 + c) * D + 16 * q + 4 * g) = x[rt][q];
    }
}

template <typename F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 20;
}

int main() {
    const long n = 1 << 20;
    std::vector<float> hX(n * D), hW(LAYERS * D * D);
    srand(1);
    auto rnd = []() { return (float)((rand() / (double)RAND_MAX) * 2 - 1); };
    for (auto &v : hX) v = rnd();
    for (auto &v : hW) v = rnd() * 0.25f;        // keeps the activations O(1) through the layers
    float *X, *W, *Y;
    hipMalloc(&X, n * D * 4); hipMalloc(&W, hW.size() * 4); hipMalloc(&Y, n * D * 4);
    hipMemcpy(X, hX.data(), n * D * 4, hipMemcpyHostToDevice); hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
    // fp64 reference on the first 256 rows
    const int NR = 256;
    std::vector<double> ref(NR * D);
    for (int r = 0; r < NR; ++r) {
        double a[D], b[D];
        for (int k = 0; k < D; ++k) a[k] = hX[(long)r * D + k];
        for (int l = 0; l < LAYERS; ++l) {
            for (int o = 0; o < D; ++o) { double s = 0; for (int k = 0; k < D; ++k) s += a[k] * (double)hW[(l * D + k) * D + o]; b[o] = s > 0 ? s : 0; }
            for (int k = 0; k < D; ++k) a[k] = b[k];
        }
        for (int k = 0; k < D; ++k) ref[r * D + k] = a[k];
    }
    std::vector<float> hY(NR * D);
    auto report = [&](const char *name, float ms) {
        hipMemcpy(hY.data(), Y, NR * D * 4, hipMemcpyDeviceToHost);
        double emax = 0, scale = 0;
        for (int i = 0; i < NR * D; ++i) { emax = fmax(emax, fabs(hY[i] - ref[i])); scale = fmax(scale, fabs(ref[i])); }
        const double flops = 2.0 * n * D * D * LAYERS;
        printf("%-34s %8.1f us  %6.1f useful TFLOP/s   max |err| / max |y| = %.2e\n", name, ms * 1e3, flops / (ms * 1e-3) / 1e12, emax / scale);
    };
    const int blocks = 256;
    report("fp32 MFMA 16x16x4, 16-row tiles", timeit([&] { k_f32<1><<<blocks, WAVES * 64>>>(X, W, Y, n); }));
    report("fp32 MFMA 16x16x4, 32-row tiles", timeit([&] { k_f32<2><<<blocks, WAVES * 64>>>(X, W, Y, n); }));
    float *Hall; hipMalloc(&Hall, n * D * 4 * LAYERS);
    report("fp32, 16 rows, 16 waves", timeit([&] { k_f32<1, 16><<<blocks, 16 * 64>>>(X, W, Y, n); }));
    report("fp32, 32 rows, 16 waves", timeit([&] { k_f32<2, 16><<<blocks, 16 * 64>>>(X, W, Y, n); }));
    report("fp32, 64 rows, 8 waves", timeit([&] { k_f32<4, 8><<<blocks, 8 * 64>>>(X, W, Y, n); }));
    report("fp32, 16 rows, 16 waves, +H stores", timeit([&] { k_f32<1, 16, true><<<blocks, 16 * 64>>>(X, W, Y, n, Hall); }));
    report("fp32, 32 rows, 8 waves, +H stores", timeit([&] { k_f32<2, 8, true><<<blocks, 8 * 64>>>(X, W, Y, n, Hall); }));
    hipFree(Hall); hipMalloc(&Hall, n * 4 * (LAYERS * D + 8));
    report("  same, row stride + 4 B (misaligned)", timeit([&] { k_f32<2, 8, true, 1><<<blocks, 8 * 64>>>(X, W, Y, n, Hall); }));
    report("  same, row stride + 24 B", timeit([&] { k_f32<2, 8, true, 6><<<blocks, 8 * 64>>>(X, W, Y, n, Hall); }));
    report("  16 rows, 16 waves, stride + 24 B", timeit([&] { k_f32<1, 16, true, 6><<<blocks, 16 * 64>>>(X, W, Y, n, Hall); }));
    report("  stride + 24 B, 8-byte stores", timeit([&] { k_f32<2, 8, true, 6, 2><<<blocks, 8 * 64>>>(X, W, Y, n, Hall); }));
    report("  stride + 4 B, 4-byte stores", timeit([&] { k_f32<2, 8, true, 1, 1><<<blocks, 8 * 64>>>(X, W, Y, n, Hall); }));
    report("  aligned stride, 8-byte stores", timeit([&] { k_f32<2, 8, true, 0, 2><<<blocks, 8 * 64>>>(X, W, Y, n, Hall); }));
    report("bf16 split, 9 products, 16 rows", timeit([&] { k_bf16<1, 9><<<blocks, WAVES * 64>>>(X, W, Y, n); }));
    report("bf16 split, 9 products, 32 rows", timeit([&] { k_bf16<2, 9><<<blocks, WAVES * 64>>>(X, W, Y, n); }));
    report("bf16 split, 6 products, 32 rows", timeit([&] { k_bf16<2, 6><<<blocks, WAVES * 64>>>(X, W, Y, n); }));
    report("bf16 split, 3 products, 32 rows", timeit([&] { k_bf16<2, 3><<<blocks, WAVES * 64>>>(X, W, Y, n); }));
    return 0;
}
