// Fused 2-layer MLPs (Linear -> ReLU -> Linear [-> tanh | sigmoid]) on the gfx950 fp32
// matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, 1024 FMA / instruction).
//
// Why: rocprofv3 of the first end-to-end bench (profiles/r01_rocprof_bench_1m_v0.txt)
// shows the anchor MLPs (gaussian_renderer/__init__.py:112-126, 54 -> 50 -> {10,30,70})
// and the context MLPs (scene/gaussian_model.py:177-188, {71,15} -> 100 -> 175) as THE
// dense-contraction bottleneck: rocBLAS picks poor tiles for these skinny shapes (K = 1 M
// rows for the weight gradients) and torch's bias-gradient reductions cost more than the
// rasterizer.  north_star: "MFMA only if rocprof shows them as a dense contraction
// bottleneck" — it does.
//
// Design: everything is computed TRANSPOSED (weights are the A operand, activations the B
// operand) so that one layer's accumulator registers are directly the next layer's B
// operand: D-layout lane l, reg r holds Z^T[16t + 4(l>>4) + r][row l&15], and a k-step may
// use ANY set of four contraction indices as long as A follows it, so k-step (t, r) uses
// {16t + 4g + r : g = 0..3}.  No LDS transposes; LDS holds only the weights, in layouts
// whose fragment reads are bank-conflict free.  Activations stream straight from/to HBM
// (16 rows x 16 B segments, L1-served across k-steps).
//
//   mlp2_fwd      X [n,IN] -> Y [n,OUT] (+ H = relu(.) saved for the backward)
//   mlp2_bwd      dY, Y, H -> dX, dZ1 (= dH masked), dZ2 (= dY * act')
//   wgrad         dW[a][b] += sum_rows P[row][a] Q[row][b], db[a] += sum_rows P[row][a]
//                 both operands read row-major from HBM in fragment order, split over rows
//                 across workgroups, fp32 atomics at the end.
#include "cgs_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define ACT_NONE 0
#define ACT_TANH 1
#define ACT_SIGMOID 2

// smallest s >= x with s % 32 == 16: the two A-fragment rows (4s+g, g = 0,1) a 32-lane group reads land on
// disjoint halves of the 32 banks
constexpr int pad16mod32(int x) { int s = 16; while (s < x) s += 32; return s; }
// smallest s >= x with s % 8 == 4: rows 16t+4g+r are 4 apart -> shifted by 16 banks
constexpr int pad4mod8(int x) { int s = 4; while (s < x) s += 8; return s; }

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

template <int ACT>
__device__ __forceinline__ float act_fwd(float z) {
    if (ACT == ACT_TANH) return tanhf(z);
    if (ACT == ACT_SIGMOID) return 1.f / (1.f + __expf(-z));
    return z;
}
template <int ACT>
__device__ __forceinline__ float act_grad_from_y(float y) {
    if (ACT == ACT_TANH) return 1.f - y * y;
    if (ACT == ACT_SIGMOID) return y * (1.f - y);
    return 1.f;
}

// ------------------------------------------------------------------------------------------
template <int IN, int HID, int OUT, int ACT, int RT, int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
    mlp2_fwd_kernel(const float *__restrict__ X, int64_t ldx, const float *__restrict__ W1,
                    const float *__restrict__ b1, const float *__restrict__ W2, const float *__restrict__ b2,
                    float *__restrict__ Y, int64_t ldy, float *__restrict__ Hsave, int64_t n) {
    constexpr int KS1 = (IN + 3) / 4, NT1 = (HID + 15) / 16, NT2 = (OUT + 15) / 16;
    constexpr int HP = NT1 * 16, OP = NT2 * 16;
    constexpr int S1 = pad16mod32(HP), S2 = pad4mod8(OP);
    __shared__ float W1s[KS1 * 4 * S1];   // [k][j]
    __shared__ float W2s[HP * S2];        // [h][o]
    __shared__ float b1s[HP];
    __shared__ float b2s[OP];
    const int tid = threadIdx.x, nthr = WAVES * 64;
    for (int i = tid; i < KS1 * 4 * S1; i += nthr) {
        const int k = i / S1, j = i % S1;
        W1s[i] = (k < IN && j < HID) ? W1[j * IN + k] : 0.f;
    }
    for (int i = tid; i < HP * S2; i += nthr) {
        const int h = i / S2, o = i % S2;
        W2s[i] = (h < HID && o < OUT) ? W2[o * HID + h] : 0.f;
    }
    for (int i = tid; i < HP; i += nthr) b1s[i] = i < HID ? b1[i] : 0.f;
    for (int i = tid; i < OP; i += nthr) b2s[i] = i < OUT ? b2[i] : 0.f;
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int64_t ntiles = (n + 16 * RT - 1) / (16 * RT);
    for (int64_t tile = (int64_t)blockIdx.x * WAVES + wave; tile < ntiles; tile += (int64_t)gridDim.x * WAVES) {
        const int64_t row0 = tile * 16 * RT;
        float xb[RT][KS1];
        bool valid[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int64_t row = row0 + rt * 16 + c;
            valid[rt] = row < n;
            const float *xr = X + row * ldx;
#pragma unroll
            for (int s = 0; s < KS1; ++s) {
                const int k = 4 * s + g;
                xb[rt][s] = (valid[rt] && k < IN) ? xr[k] : 0.f;
            }
        }
        f32x4 acc1[NT1][RT];
#pragma unroll
        for (int t = 0; t < NT1; ++t)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc1[t][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS1; ++s)
#pragma unroll
            for (int t = 0; t < NT1; ++t) {
                const float a = W1s[(4 * s + g) * S1 + 16 * t + c];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc1[t][rt] = mfma4(a, xb[rt][s], acc1[t][rt]);
            }
        // bias + ReLU; keep H for the backward
#pragma unroll
        for (int t = 0; t < NT1; ++t)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int64_t row = row0 + rt * 16 + c;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int h = 16 * t + 4 * g + r;
                    const float v = fmaxf(acc1[t][rt][r] + b1s[h], 0.f);
                    acc1[t][rt][r] = v;
                    if (Hsave && valid[rt] && h < HID) Hsave[row * HID + h] = v;
                }
            }
        f32x4 acc2[NT2][RT];
#pragma unroll
        for (int u = 0; u < NT2; ++u)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc2[u][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NT1; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (16 * t + r >= HID) continue;   // every contraction index of this step is padding
#pragma unroll
                for (int u = 0; u < NT2; ++u) {
                    const float a = W2s[(16 * t + 4 * g + r) * S2 + 16 * u + c];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc2[u][rt] = mfma4(a, acc1[t][rt][r], acc2[u][rt]);
                }
            }
#pragma unroll
        for (int u = 0; u < NT2; ++u)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int64_t row = row0 + rt * 16 + c;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = 16 * u + 4 * g + r;
                    if (valid[rt] && o < OUT) Y[row * ldy + o] = act_fwd<ACT>(acc2[u][rt][r] + b2s[o]);
                }
            }
    }
}

// ------------------------------------------------------------------------------------------
template <int IN, int HID, int OUT, int ACT, int RT, int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
    mlp2_bwd_kernel(const float *__restrict__ dY, const float *__restrict__ Y, int64_t ldy,
                    const float *__restrict__ Hsave, const float *__restrict__ W1, const float *__restrict__ W2,
                    float *__restrict__ dX, int64_t lddx, int accumulate_dx, float *__restrict__ dZ2,
                    float *__restrict__ dZ1, int64_t n) {
    constexpr int KS2 = (OUT + 3) / 4, NT1 = (HID + 15) / 16, NTX = (IN + 15) / 16;
    constexpr int HP = NT1 * 16, XP = NTX * 16;
    constexpr int SA = pad16mod32(HP), SB = pad4mod8(XP);
    __shared__ float W2n[KS2 * 4 * SA];   // [o][h]
    __shared__ float W1n[HP * SB];        // [h][k]
    const int tid = threadIdx.x, nthr = WAVES * 64;
    for (int i = tid; i < KS2 * 4 * SA; i += nthr) {
        const int o = i / SA, h = i % SA;
        W2n[i] = (o < OUT && h < HID) ? W2[o * HID + h] : 0.f;
    }
    for (int i = tid; i < HP * SB; i += nthr) {
        const int h = i / SB, k = i % SB;
        W1n[i] = (h < HID && k < IN) ? W1[h * IN + k] : 0.f;
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int64_t ntiles = (n + 16 * RT - 1) / (16 * RT);
    for (int64_t tile = (int64_t)blockIdx.x * WAVES + wave; tile < ntiles; tile += (int64_t)gridDim.x * WAVES) {
        const int64_t row0 = tile * 16 * RT;
        bool valid[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) valid[rt] = row0 + rt * 16 + c < n;
        // dH^T = W2^T dZ2^T
        f32x4 adh[NT1][RT];
#pragma unroll
        for (int t = 0; t < NT1; ++t)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) adh[t][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS2; ++s) {
            float b[RT];
            const int o = 4 * s + g;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int64_t row = row0 + rt * 16 + c;
                float v = 0.f;
                if (valid[rt] && o < OUT) {
                    v = dY[row * ldy + o];
                    if (ACT != ACT_NONE) v *= act_grad_from_y<ACT>(Y[row * ldy + o]);
                    if (dZ2) dZ2[row * OUT + o] = v;
                }
                b[rt] = v;
            }
#pragma unroll
            for (int t = 0; t < NT1; ++t) {
                const float a = W2n[(4 * s + g) * SA + 16 * t + c];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) adh[t][rt] = mfma4(a, b[rt], adh[t][rt]);
            }
        }
        // ReLU mask from the saved activations -> dZ1
#pragma unroll
        for (int t = 0; t < NT1; ++t)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int64_t row = row0 + rt * 16 + c;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int h = 16 * t + 4 * g + r;
                    float d = 0.f;
                    if (valid[rt] && h < HID) {
                        d = Hsave[row * HID + h] > 0.f ? adh[t][rt][r] : 0.f;
                        dZ1[row * HID + h] = d;
                    }
                    adh[t][rt][r] = d;
                }
            }
        if (dX) {
            f32x4 adx[NTX][RT];
#pragma unroll
            for (int v = 0; v < NTX; ++v)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) adx[v][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < NT1; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (16 * t + r >= HID) continue;
#pragma unroll
                    for (int v = 0; v < NTX; ++v) {
                        const float a = W1n[(16 * t + 4 * g + r) * SB + 16 * v + c];
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) adx[v][rt] = mfma4(a, adh[t][rt][r], adx[v][rt]);
                    }
                }
#pragma unroll
            for (int v = 0; v < NTX; ++v)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const int64_t row = row0 + rt * 16 + c;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int k = 16 * v + 4 * g + r;
                        if (valid[rt] && k < IN) {
                            float *p = dX + row * lddx + k;
                            *p = accumulate_dx ? (*p + adx[v][rt][r]) : adx[v][rt][r];
                        }
                    }
                }
        }
    }
}

// ------------------------------------------------------------------------------------------
// dW[a][b] += sum_rows P[row][a] * Q[row][b];  db[a] += sum_rows P[row][a]
// The NA x NB output tiles are dealt round-robin to the WAVES waves of a workgroup (tile id =
// wave + WAVES*j), so every wave has MFMA work whatever the aspect ratio; the row loop is
// unrolled UNR deep so that 2*TPW*UNR fragment loads are in flight per wave (the operands
// come straight from HBM/L2 in fragment order: 4 rows x 64 B per instruction, and the 8
// waves of a workgroup re-touch each other's lines in L1).
template <int TPW, int WAVES, int UNR>
__global__ void __launch_bounds__(WAVES * 64)
    wgrad_kernel(const float *__restrict__ P, int64_t ldp, int DA, const float *__restrict__ Q, int64_t ldq, int DB,
                 float *__restrict__ dW, float *__restrict__ db, int64_t n, int64_t rows_per_block) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int NA = (DA + 15) / 16, NB = (DB + 15) / 16, T = NA * NB;
    const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r_end = min(n, r_begin + rows_per_block);
    f32x4 acc[TPW], accb[TPW];
    int acol[TPW], bcol[TPW];
    bool live[TPW], bias[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int id = wave + WAVES * j;
        live[j] = id < T;
        const int u = live[j] ? id / NB : 0, t = live[j] ? id % NB : 0;
        acol[j] = 16 * u + c;
        bcol[j] = 16 * t + c;
        bias[j] = live[j] && t == 0 && db != nullptr;
        live[j] = live[j] && true;
        acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        accb[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (int64_t row0 = r_begin; row0 < r_end; row0 += 4 * UNR) {
        float a[UNR][TPW], b[UNR][TPW], one[UNR];
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
            const int64_t row = row0 + 4 * q + g;
            const bool valid = row < r_end;
            one[q] = valid ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                a[q][j] = (valid && live[j] && acol[j] < DA) ? P[row * ldp + acol[j]] : 0.f;
                b[q][j] = (valid && live[j] && bcol[j] < DB) ? Q[row * ldq + bcol[j]] : 0.f;
            }
        }
#pragma unroll
        for (int q = 0; q < UNR; ++q)
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                acc[j] = mfma4(a[q][j], b[q][j], acc[j]);
                if (bias[j]) accb[j] = mfma4(a[q][j], one[q], accb[j]);
            }
    }
    // D layout: reg r of lane l <-> (a = 16u + 4g + r, b = 16t + c)
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        if (!live[j]) continue;
        const int u16 = acol[j] - c, b_ = bcol[j];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int arow = u16 + 4 * g + r;
            if (arow >= DA) continue;
            if (b_ < DB) atomicAdd(&dW[(int64_t)arow * DB + b_], acc[j][r]);
            if (bias[j] && c == 0) atomicAdd(&db[arow], accb[j][r]);
        }
    }
}

// ------------------------------------------------------------------------------------------
int cgs_launch_wgrad2(const float *P, int64_t ldp, int DA, const float *Q, int64_t ldq, int DB, float *dW, float *db,
                      int64_t n, int num_cus, hipStream_t s);

static int num_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

template <int IN, int HID, int OUT, int ACT>
static int launch_fwd(const float *X, int64_t ldx, const float *W1, const float *b1, const float *W2, const float *b2,
                      float *Y, int64_t ldy, float *H, int64_t n, hipStream_t s) {
    constexpr int RT = 2, WAVES = 8;
    const int64_t tiles = (n + 16 * RT - 1) / (16 * RT);
    const int64_t want = (tiles + WAVES - 1) / WAVES;
    const int grid = (int)(want < num_cus() ? want : num_cus());
    hipLaunchKernelGGL((mlp2_fwd_kernel<IN, HID, OUT, ACT, RT, WAVES>), dim3(grid), dim3(WAVES * 64), 0, s, X, ldx, W1, b1, W2,
                       b2, Y, ldy, H, n);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

template <int IN, int HID, int OUT, int ACT>
static int launch_bwd(const float *dY, const float *Y, int64_t ldy, const float *H, const float *W1, const float *W2,
                      float *dX, int64_t lddx, int acc, float *dZ2, float *dZ1, int64_t n, hipStream_t s) {
    constexpr int RT = 2, WAVES = 8;
    const int64_t tiles = (n + 16 * RT - 1) / (16 * RT);
    const int64_t want = (tiles + WAVES - 1) / WAVES;
    const int grid = (int)(want < num_cus() ? want : num_cus());
    hipLaunchKernelGGL((mlp2_bwd_kernel<IN, HID, OUT, ACT, RT, WAVES>), dim3(grid), dim3(WAVES * 64), 0, s, dY, Y, ldy, H, W1,
                       W2, dX, lddx, acc, dZ2, dZ1, n);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

static int launch_wgrad(const float *P, int64_t ldp, int DA, const float *Q, int64_t ldq, int DB, float *dW, float *db,
                        int64_t n, hipStream_t s) {
    constexpr int WAVES = 8;
    if (n <= 0) return CGS_OK;
    const int T = ((DA + 15) / 16) * ((DB + 15) / 16);
    const int tpw = (T + WAVES - 1) / WAVES;
    // few, long-running workgroups: every workgroup ends with DA*DB same-address atomics, so their number
    // (not the row count) sets the L2 atomic traffic
    int64_t blocks = (n + 1023) / 1024;
    const int64_t cap = (int64_t)num_cus();
    if (blocks > cap) blocks = cap;
    int64_t rpb = (n + blocks - 1) / blocks;
    rpb = (rpb + 15) / 16 * 16;
    blocks = (n + rpb - 1) / rpb;
#define WG(TPW_, UNR_)                                                                                                  \
    hipLaunchKernelGGL((wgrad_kernel<TPW_, WAVES, UNR_>), dim3((unsigned)blocks), dim3(WAVES * 64), 0, s, P, ldp, DA, Q, ldq, \
                       DB, dW, db, n, rpb)
    switch (tpw) {
        case 1: WG(1, 4); break;
        case 2: WG(2, 4); break;
        case 3: WG(3, 4); break;
        case 4: WG(4, 4); break;
        case 5: WG(5, 2); break;
        case 6: WG(6, 2); break;
        case 7: WG(7, 2); break;
        case 8: WG(8, 2); break;
        case 9: WG(9, 2); break;
        case 10: WG(10, 2); break;
        default: cgs_set_error("wgrad: unsupported dims %d x %d", DA, DB); return CGS_ERR_ARG;
    }
#undef WG
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

#define MLP_CONFIGS(X_)           \
    X_(54, 50, 10, ACT_TANH)      \
    X_(54, 50, 30, ACT_SIGMOID)   \
    X_(54, 50, 70, ACT_NONE)      \
    X_(71, 100, 175, ACT_NONE)    \
    X_(15, 100, 175, ACT_NONE)    \
    X_(71, 100, 3, ACT_NONE)      \
    X_(15, 100, 3, ACT_NONE)

// Y = act(W2 relu(W1 x + b1) + b2); H [n,HID] receives relu(.) (may be NULL for inference).
extern "C" int cgs_mlp2_forward(int in, int hid, int out, int act, const float *X, int64_t ldx, const float *W1,
                                const float *b1, const float *W2, const float *b2, float *Y, int64_t ldy, float *H,
                                int64_t n, void *stream) {
    if (n < 0) { cgs_set_error("mlp2_forward: n < 0"); return CGS_ERR_ARG; }
    if (n == 0) return CGS_OK;
    if (!X || !W1 || !b1 || !W2 || !b2 || !Y) { cgs_set_error("mlp2_forward: NULL"); return CGS_ERR_ARG; }
    CgsProfScope prof(CGS_PROF_MLP_FWD, (hipStream_t)stream);
#define X_(I, Hh, O, A) \
    if (in == I && hid == Hh && out == O && act == A) return launch_fwd<I, Hh, O, A>(X, ldx, W1, b1, W2, b2, Y, ldy, H, n, (hipStream_t)stream);
    MLP_CONFIGS(X_)
#undef X_
    cgs_set_error("mlp2_forward: no kernel instance for %d -> %d -> %d act %d", in, hid, out, act);
    return CGS_ERR_ARG;
}

// dX [n,lddx] (NULL to skip; accumulate_dx adds into it), dZ1 [n,hid], dZ2 [n,out] (may be NULL when act == none:
// then dZ2 == dY).  Weight/bias gradients are ACCUMULATED (atomics) into dW1/db1/dW2/db2: zero or pre-load them.
extern "C" int cgs_mlp2_backward(int in, int hid, int out, int act, const float *X, int64_t ldx, const float *W1,
                                 const float *W2, const float *Y, const float *dY, int64_t ldy, const float *H,
                                 float *dX, int64_t lddx, int accumulate_dx, float *dZ1, float *dZ2, float *dW1,
                                 float *db1, float *dW2, float *db2, int64_t n, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0) { cgs_set_error("mlp2_backward: n < 0"); return CGS_ERR_ARG; }
    if (n == 0) return CGS_OK;
    if (!X || !W1 || !W2 || !dY || !H || !dZ1 || !dW1 || !db1 || !dW2 || !db2 || (act != ACT_NONE && (!Y || !dZ2))) {
        cgs_set_error("mlp2_backward: NULL");
        return CGS_ERR_ARG;
    }
    int rc = CGS_ERR_ARG;
    {
        CgsProfScope prof(CGS_PROF_MLP_BWD, stream);
        bool found = false;
#define X_(I, Hh, O, A)                                                                                              \
    if (!found && in == I && hid == Hh && out == O && act == A) {                                                     \
        found = true;                                                                                                 \
        rc = launch_bwd<I, Hh, O, A>(dY, Y, ldy, H, W1, W2, dX, lddx, accumulate_dx, dZ2, dZ1, n, stream);            \
    }
        MLP_CONFIGS(X_)
#undef X_
        if (!found) { cgs_set_error("mlp2_backward: no kernel instance for %d -> %d -> %d act %d", in, hid, out, act); return CGS_ERR_ARG; }
        if (rc) return rc;
    }
    CgsProfScope prof(CGS_PROF_MLP_WGRAD, stream);
    const float *P2 = (act == ACT_NONE || !dZ2) ? dY : dZ2;
    const int64_t ldp2 = (act == ACT_NONE || !dZ2) ? ldy : out;
    if ((rc = cgs_launch_wgrad2(P2, ldp2, out, H, hid, hid, dW2, db2, n, num_cus(), stream))) return rc;
    return cgs_launch_wgrad2(dZ1, hid, hid, X, ldx, in, dW1, db1, n, num_cus(), stream);
}
