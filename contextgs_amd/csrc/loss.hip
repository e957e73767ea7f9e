// Image loss of the training iteration (train.py:199-204): L1 + (1 - SSIM) with the 11x11, sigma 1.5 Gaussian
// window, zero padding and per-channel convolution of utils/loss_utils.py:17-63 — SURVEY section 8(f) rank 2:
// it sits immediately after the rasterizer in every iteration and is ~20 full-image passes in the reference
// (5 grouped conv2d + their backward + element-wise maps).
//
// One forward and one backward kernel.  A workgroup owns a 32x32 pixel tile of one channel: the 42x42 input patch
// (halo 5) goes to LDS once, the five moment images (x, y, x^2, y^2, xy) are filtered separably (rows into LDS,
// columns from LDS; a thread filters four neighbouring outputs from the 14 values they share), SSIM and its three
// partial-derivative maps are formed in registers.  The backward filters the three derivative maps the same way (the
// window is symmetric, so the transposed convolution is the same filter) and combines them with the pixel values.
// HBM traffic: forward 8 + 12 B / pixel / channel (two reads, three map writes), backward 20 + 4 B.
#include "cgs_internal.h"

#define SS_T 32                    // output tile: SS_T x SS_TY pixels of one channel per workgroup (256 threads, SS_CB outputs each)
#ifndef SS_TY
#define SS_TY 32
#endif
#define SS_CB (SS_T * SS_TY / 256) // outputs of a thread in the column pass (consecutive rows)
#define SS_R 5
#define SS_P (SS_T + 2 * SS_R)     // 42 columns of the input patch
#define SS_PY (SS_TY + 2 * SS_R)   // its rows
#define SS_K 11
#define SS_NG (SS_T / 4)           // groups of four neighbouring outputs per row / column

struct SsimWin { float w[SS_K]; };

// gaussian(11, 1.5) as utils/loss_utils.py:23-25 computes it: exp in double, stored fp32, normalised in fp32
static SsimWin ssim_window() {
    SsimWin g;
    float s = 0.f;
    for (int x = 0; x < SS_K; ++x) {
        g.w[x] = (float)exp(-(double)((x - SS_K / 2) * (x - SS_K / 2)) / (2.0 * 1.5 * 1.5));
        s += g.w[x];
    }
    for (int x = 0; x < SS_K; ++x) g.w[x] /= s;
    return g;
}

__device__ __forceinline__ float block_sum_256(float v, float *sh) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// Round 6: 32 x 32 tiles and REGISTER-BLOCKED filters.  A thread filters four neighbouring outputs of a row (then of a column)
// from the 14 values they share, read once from LDS — 3.5 LDS reads per output and image instead of 11 — and the halo a tile
// loads is 1.7 x its pixels instead of 2.6 x (16 x 16 tiles).  Every output is still the same sum in the same tap order.
// fwd 106 -> see profiles/r06_loss.txt.
__global__ void __launch_bounds__(256)
    l1_ssim_fwd_kernel(const float *__restrict__ img, const float *__restrict__ gt, int H, int W, SsimWin win,
                       float *__restrict__ maps /* [3][C][H][W] or null */, float *__restrict__ partials) {
    __shared__ float sx[SS_PY][SS_P + 1], sy[SS_PY][SS_P + 1];
    __shared__ float hq[5][SS_PY][SS_T + 1];
    __shared__ float red[4];
    const int c = blockIdx.z, C = gridDim.z;
    const int x0 = blockIdx.x * SS_T, y0 = blockIdx.y * SS_TY;
    const int tid = threadIdx.x;
    const size_t plane = (size_t)H * W;
    const float *xi = img + c * plane, *yi = gt + c * plane;
    {
        // the patch: every load of the thread is requested before the first goes to LDS (as a plain loop the seven rounds were seven
        // dependent round trips per workgroup, with three workgroups per CU to hide them)
        constexpr int NL = (SS_PY * SS_P + 255) / 256;
        float vx[NL], vy[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int i = tid + 256 * k;
            const int py = i / SS_P, px = i - py * SS_P;
            const int gy = y0 + py - SS_R, gx = x0 + px - SS_R;
            const bool in = i < SS_PY * SS_P && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const size_t o = in ? (size_t)gy * W + gx : 0;
            vx[k] = in ? xi[o] : 0.f;
            vy[k] = in ? yi[o] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int i = tid + 256 * k;
            if (i < SS_PY * SS_P) {
                const int py = i / SS_P, px = i - py * SS_P;
                sx[py][px] = vx[k];
                sy[py][px] = vy[k];
            }
        }
    }
    __syncthreads();
    // rows: item = (row r of the patch, group of four columns)
    for (int i = tid; i < SS_PY * SS_NG; i += 256) {
        const int r = i / SS_NG, col = 4 * (i - r * SS_NG);
        float xv[SS_K + 3], yv[SS_K + 3];
#pragma unroll
        for (int k = 0; k < SS_K + 3; ++k) { xv[k] = sx[r][col + k]; yv[k] = sy[r][col + k]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
            for (int k = 0; k < SS_K; ++k) {
                const float x = xv[j + k], y = yv[j + k], w = win.w[k];
                a0 += w * x; a1 += w * y; a2 += w * (x * x); a3 += w * (y * y); a4 += w * (x * y);
            }
            hq[0][r][col + j] = a0; hq[1][r][col + j] = a1; hq[2][r][col + j] = a2; hq[3][r][col + j] = a3; hq[4][r][col + j] = a4;
        }
    }
    __syncthreads();
    // columns: thread = (column lx of the 32, group of SS_CB rows)
    const int lx = tid & (SS_T - 1), ly0 = SS_CB * (tid / SS_T);
    float mu1[SS_CB], mu2[SS_CB], e11[SS_CB], e22[SS_CB], e12[SS_CB];
    {
        float v[5][SS_K + SS_CB - 1];
#pragma unroll
        for (int q = 0; q < 5; ++q)
#pragma unroll
            for (int k = 0; k < SS_K + SS_CB - 1; ++k) v[q][k] = hq[q][ly0 + k][lx];
#pragma unroll
        for (int j = 0; j < SS_CB; ++j) {
            float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f, b4 = 0.f;
#pragma unroll
            for (int k = 0; k < SS_K; ++k) {
                const float w = win.w[k];
                b0 += w * v[0][j + k]; b1 += w * v[1][j + k]; b2 += w * v[2][j + k]; b3 += w * v[3][j + k]; b4 += w * v[4][j + k];
            }
            mu1[j] = b0; mu2[j] = b1; e11[j] = b2; e22[j] = b3; e12[j] = b4;
        }
    }
    float m = 0.f, l1 = 0.f;
#pragma unroll
    for (int j = 0; j < SS_CB; ++j) {
        const int ly = ly0 + j, gx = x0 + lx, gy = y0 + ly;
        if (gx < W && gy < H) {
            const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
            const float mu1_sq = mu1[j] * mu1[j], mu2_sq = mu2[j] * mu2[j], mu12 = mu1[j] * mu2[j];
            const float s1 = e11[j] - mu1_sq, s2 = e22[j] - mu2_sq, s12 = e12[j] - mu12;
            const float a1 = 2.f * mu12 + C1, a2 = 2.f * s12 + C2, b1 = mu1_sq + mu2_sq + C1, b2 = s1 + s2 + C2;
            m += (a1 * a2) / (b1 * b2);
            l1 += fabsf(sx[ly + SS_R][lx + SS_R] - sy[ly + SS_R][lx + SS_R]);
            if (maps) {
                const float inv = 1.f / (b1 * b2);
                const float dm_ds1 = -(a1 * a2) * inv / b2;            // = dm / d sigma1_sq
                const float dm_ds12 = 2.f * a1 * inv;
                const float dm_dmu1 = (2.f * mu2[j] * a2 * b1 - 2.f * mu1[j] * a1 * a2) * inv / b1
                                      - 2.f * mu1[j] * dm_ds1 - mu2[j] * dm_ds12;      // through sigma1_sq and sigma12 too
                const size_t o = (size_t)c * plane + (size_t)gy * W + gx, cs = (size_t)C * plane;
                maps[o] = dm_dmu1;
                maps[cs + o] = dm_ds1;
                maps[2 * cs + o] = dm_ds12;
            }
        }
    }
    const float sl = block_sum_256(l1, red);
    __syncthreads();
    const float sm = block_sum_256(m, red);
    if (tid == 0) {
        const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partials[2 * b] = sl;
        partials[2 * b + 1] = sm;
    }
}

// dimg = g[0] / n * sign(img - gt) + g[1] / n * (conv(A) + 2 img conv(B) + gt conv(Cc)),  n = C H W
__global__ void __launch_bounds__(256)
    l1_ssim_bwd_kernel(const float *__restrict__ img, const float *__restrict__ gt, const float *__restrict__ maps,
                       const float *__restrict__ g, const float *__restrict__ g_loss, float lam, int H, int W, SsimWin win,
                       float *__restrict__ dimg) {
    // g [2] (may be null): gradients of (L1, SSIM); g_loss [1] (may be null): gradient of (1 - lam) L1 + lam (1 - SSIM)
    const float gl = g_loss ? g_loss[0] : 0.f;
    const float g0 = (g ? g[0] : 0.f) + gl * (1.f - lam), g1 = (g ? g[1] : 0.f) - gl * lam;
    __shared__ float sm[3][SS_PY][SS_P + 1];
    __shared__ float hq[3][SS_PY][SS_T + 1];
    const int c = blockIdx.z, C = gridDim.z;
    const int x0 = blockIdx.x * SS_T, y0 = blockIdx.y * SS_TY;
    const int tid = threadIdx.x;
    const size_t plane = (size_t)H * W, cs = (size_t)C * plane;
    {
        constexpr int NL = (SS_PY * SS_P + 255) / 256;      // (all loads before the first LDS store: see the forward)
        float vm[NL][3];
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int i = tid + 256 * k;
            const int py = i / SS_P, px = i - py * SS_P;
            const int gy = y0 + py - SS_R, gx = x0 + px - SS_R;
            const bool in = i < SS_PY * SS_P && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const size_t o = in ? (size_t)c * plane + (size_t)gy * W + gx : 0;
#pragma unroll
            for (int q = 0; q < 3; ++q) vm[k][q] = in ? maps[q * cs + o] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int i = tid + 256 * k;
            if (i < SS_PY * SS_P) {
                const int py = i / SS_P, px = i - py * SS_P;
#pragma unroll
                for (int q = 0; q < 3; ++q) sm[q][py][px] = vm[k][q];
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < SS_PY * SS_NG; i += 256) {
        const int r = i / SS_NG, col = 4 * (i - r * SS_NG);
        float v[3][SS_K + 3];
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int k = 0; k < SS_K + 3; ++k) v[q][k] = sm[q][r][col + k];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int k = 0; k < SS_K; ++k) {
                const float w = win.w[k];
                a0 += w * v[0][j + k]; a1 += w * v[1][j + k]; a2 += w * v[2][j + k];
            }
            hq[0][r][col + j] = a0; hq[1][r][col + j] = a1; hq[2][r][col + j] = a2;
        }
    }
    __syncthreads();
    const int lx = tid & (SS_T - 1), ly0 = SS_CB * (tid / SS_T);
    float v[3][SS_K + SS_CB - 1];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int k = 0; k < SS_K + SS_CB - 1; ++k) v[q][k] = hq[q][ly0 + k][lx];
#pragma unroll
    for (int j = 0; j < SS_CB; ++j) {
        float cA = 0.f, cB = 0.f, cC = 0.f;
#pragma unroll
        for (int k = 0; k < SS_K; ++k) {
            const float w = win.w[k];
            cA += w * v[0][j + k]; cB += w * v[1][j + k]; cC += w * v[2][j + k];
        }
        const int gx = x0 + lx, gy = y0 + ly0 + j;
        if (gx < W && gy < H) {
            const size_t o = (size_t)c * plane + (size_t)gy * W + gx;
            const float x = img[o], y = gt[o];
            const float inv_n = 1.f / ((float)C * (float)H * (float)W);
            const float d = x - y;
            const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            dimg[o] = g0 * inv_n * sgn + g1 * inv_n * (cA + 2.f * x * cB + y * cC);
        }
    }
}

extern "C" size_t cgs_l1_ssim_partials(int C, int H, int W) {
    return (size_t)C * ((H + SS_TY - 1) / SS_TY) * ((W + SS_T - 1) / SS_T);
}

extern "C" int cgs_l1_ssim_fwd(const float *img, const float *gt, int C, int H, int W, float *maps, float *partials,
                               void *stream) {
    if (C < 1 || H < 1 || W < 1) { cgs_set_error("l1_ssim_fwd: bad shape"); return CGS_ERR_ARG; }
    if (!img || !gt || !partials) { cgs_set_error("l1_ssim_fwd: NULL"); return CGS_ERR_ARG; }
    static const SsimWin win = ssim_window();
    const dim3 grid((W + SS_T - 1) / SS_T, (H + SS_TY - 1) / SS_TY, C);
    CgsProfScope prof(CGS_PROF_LOSS_FWD, (hipStream_t)stream);
    hipLaunchKernelGGL(l1_ssim_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, img, gt, H, W, win, maps, partials);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

extern "C" int cgs_l1_ssim_bwd(const float *img, const float *gt, const float *maps, const float *g, int C, int H,
                               int W, float *dimg, void *stream) {
    if (C < 1 || H < 1 || W < 1) { cgs_set_error("l1_ssim_bwd: bad shape"); return CGS_ERR_ARG; }
    if (!img || !gt || !maps || !g || !dimg) { cgs_set_error("l1_ssim_bwd: NULL"); return CGS_ERR_ARG; }
    static const SsimWin win = ssim_window();
    const dim3 grid((W + SS_T - 1) / SS_T, (H + SS_TY - 1) / SS_TY, C);
    CgsProfScope prof(CGS_PROF_LOSS_BWD, (hipStream_t)stream);
    hipLaunchKernelGGL(l1_ssim_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, img, gt, maps, g, (const float *)nullptr, 0.f,
                       H, W, win, dimg);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

// The training image loss of train.py:199-204 as ONE value: out3 = (loss, L1, SSIM) with loss = (1 - lam) L1 + lam (1 - SSIM),
// from the forward kernel's per-workgroup partial sums — one single-workgroup launch (fixed summation order) instead of torch's
// sum / div / rsub / mul / add chain (~8 launches); and its backward with the gradient of `loss` (plus, optionally, of L1 / SSIM)
// read on the device: no [2]-vector to build on the host side (~6 launches).
__global__ void __launch_bounds__(256)
    l1_ssim_finish_kernel(const float *__restrict__ partials, int64_t np, float inv_n, float lam, float *__restrict__ out3) {
    __shared__ double sh[2][4];
    double a = 0.0, b = 0.0;
    for (int64_t i = threadIdx.x; i < np; i += 256) { a += (double)partials[2 * i]; b += (double)partials[2 * i + 1]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = a; sh[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float l1 = (float)((sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]) * (double)inv_n);
        const float ss = (float)((sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3]) * (double)inv_n);
        out3[0] = (1.f - lam) * l1 + lam * (1.f - ss);
        out3[1] = l1;
        out3[2] = ss;
    }
}

extern "C" int cgs_l1_ssim_finish(const float *partials, int C, int H, int W, float lam, float *out3, void *stream) {
    if (C < 1 || H < 1 || W < 1 || !partials || !out3) { cgs_set_error("l1_ssim_finish: bad args"); return CGS_ERR_ARG; }
    hipLaunchKernelGGL(l1_ssim_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials,
                       (int64_t)cgs_l1_ssim_partials(C, H, W), 1.f / ((float)C * (float)H * (float)W), lam, out3);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

extern "C" int cgs_l1_ssim_bwd_loss(const float *img, const float *gt, const float *maps, const float *g_loss, const float *g2,
                                    float lam, int C, int H, int W, float *dimg, void *stream) {
    if (C < 1 || H < 1 || W < 1) { cgs_set_error("l1_ssim_bwd: bad shape"); return CGS_ERR_ARG; }
    if (!img || !gt || !maps || (!g_loss && !g2) || !dimg) { cgs_set_error("l1_ssim_bwd: NULL"); return CGS_ERR_ARG; }
    static const SsimWin win = ssim_window();
    const dim3 grid((W + SS_T - 1) / SS_T, (H + SS_TY - 1) / SS_TY, C);
    CgsProfScope prof(CGS_PROF_LOSS_BWD, (hipStream_t)stream);
    hipLaunchKernelGGL(l1_ssim_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, img, gt, maps, g2, g_loss, lam, H, W, win, dimg);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

// ------------------------------------------------------------------------------------------------------------
// The two regularisers of the training loss next to the image terms (train.py:203,209):
//     scaling_reg = scaling.prod(dim=1).mean()                 scaling [P,3], the visible Gaussians' scales
//     mask_reg    = torch.mean(torch.sigmoid(gaussians._mask))
// As torch operators the first is a strided product reduction, a mean and — in the backward — an `input == 0` count that
// is read back on the HOST (prod's zero-safe gradient) followed by grad * result / input: a stream drain in the middle
// of every backward plus ~8 passes over P x 3 floats; the second is ~5 passes over the 10 M mask logits.  Here: one
// streaming launch each way, per-workgroup partial sums (summed by the caller, deterministic), gradients written
// directly: d scaling[i][c] = g / P * (product of the other two), d x = g / n * s (1 - s).
#define RG_THREADS 256
#define RG_MAX_BLOCKS 2048

__global__ void __launch_bounds__(RG_THREADS)
    scaling_reg_fwd_kernel(const float *__restrict__ s, int64_t P, float *__restrict__ partials) {
    __shared__ float sh[4];
    float acc = 0.f;
    const int64_t quads = P / 4;       // four rows = three float4 (the caller checked the 16-byte alignment)
    for (int64_t q = (int64_t)blockIdx.x * RG_THREADS + threadIdx.x; q < quads; q += (int64_t)gridDim.x * RG_THREADS) {
        const float4 a = ((const float4 *)s)[3 * q], b = ((const float4 *)s)[3 * q + 1], c = ((const float4 *)s)[3 * q + 2];
        acc += (a.x * a.y * a.z + a.w * b.x * b.y) + (b.z * b.w * c.x + c.y * c.z * c.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(P - 4 * quads)) {
        const float *r = s + 3 * (4 * quads + threadIdx.x);
        acc += r[0] * r[1] * r[2];
    }
    const float tot = block_sum_256(acc, sh);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(RG_THREADS)
    scaling_reg_bwd_kernel(const float *__restrict__ s, const float *__restrict__ g, int64_t P, float *__restrict__ d) {
    const float c0 = g[0] / (float)P;
    const int64_t quads = P / 4;
    for (int64_t q = (int64_t)blockIdx.x * RG_THREADS + threadIdx.x; q < quads; q += (int64_t)gridDim.x * RG_THREADS) {
        const float4 a = ((const float4 *)s)[3 * q], b = ((const float4 *)s)[3 * q + 1], c = ((const float4 *)s)[3 * q + 2];
        float4 o0, o1, o2;
        o0.x = c0 * (a.y * a.z); o0.y = c0 * (a.x * a.z); o0.z = c0 * (a.x * a.y);
        o0.w = c0 * (b.x * b.y); o1.x = c0 * (a.w * b.y); o1.y = c0 * (a.w * b.x);
        o1.z = c0 * (b.w * c.x); o1.w = c0 * (b.z * c.x); o2.x = c0 * (b.z * b.w);
        o2.y = c0 * (c.z * c.w); o2.z = c0 * (c.y * c.w); o2.w = c0 * (c.y * c.z);
        ((float4 *)d)[3 * q] = o0; ((float4 *)d)[3 * q + 1] = o1; ((float4 *)d)[3 * q + 2] = o2;
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(P - 4 * quads)) {
        const int64_t i = 3 * (4 * quads + threadIdx.x);
        d[i] = c0 * (s[i + 1] * s[i + 2]); d[i + 1] = c0 * (s[i] * s[i + 2]); d[i + 2] = c0 * (s[i] * s[i + 1]);
    }
}

__device__ __forceinline__ float rg_sigmoid(float x) { return 1.f / (1.f + __expf(-x)); }

__global__ void __launch_bounds__(RG_THREADS)
    sigmoid_mean_fwd_kernel(const float *__restrict__ x, int64_t n, float *__restrict__ partials) {
    __shared__ float sh[4];
    float acc = 0.f;
    const int64_t quads = n / 4;
    for (int64_t q = (int64_t)blockIdx.x * RG_THREADS + threadIdx.x; q < quads; q += (int64_t)gridDim.x * RG_THREADS) {
        const float4 a = ((const float4 *)x)[q];
        acc += (rg_sigmoid(a.x) + rg_sigmoid(a.y)) + (rg_sigmoid(a.z) + rg_sigmoid(a.w));
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * quads)) acc += rg_sigmoid(x[4 * quads + threadIdx.x]);
    const float tot = block_sum_256(acc, sh);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(RG_THREADS)
    sigmoid_mean_bwd_kernel(const float *__restrict__ x, const float *__restrict__ g, int64_t n, float *__restrict__ d) {
    const float c0 = g[0] / (float)n;
    const int64_t quads = n / 4;
    for (int64_t q = (int64_t)blockIdx.x * RG_THREADS + threadIdx.x; q < quads; q += (int64_t)gridDim.x * RG_THREADS) {
        const float4 a = ((const float4 *)x)[q];
        const float s0 = rg_sigmoid(a.x), s1 = rg_sigmoid(a.y), s2 = rg_sigmoid(a.z), s3 = rg_sigmoid(a.w);
        ((float4 *)d)[q] = make_float4(c0 * s0 * (1.f - s0), c0 * s1 * (1.f - s1), c0 * s2 * (1.f - s2), c0 * s3 * (1.f - s3));
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * quads)) {
        const float s0 = rg_sigmoid(x[4 * quads + threadIdx.x]);
        d[4 * quads + threadIdx.x] = c0 * s0 * (1.f - s0);
    }
}

static unsigned rg_grid(int64_t quads) {
    const int64_t b = (quads + RG_THREADS - 1) / RG_THREADS;
    return (unsigned)(b < 1 ? 1 : (b > RG_MAX_BLOCKS ? RG_MAX_BLOCKS : b));
}

extern "C" size_t cgs_reg_partials(int64_t n) { return n < 0 ? 0 : (size_t)rg_grid(n / 4); }

static int rg_check(const char *who, const void *x, const void *out, int64_t n) {
    if (n < 1) { cgs_set_error("%s: n < 1", who); return CGS_ERR_ARG; }
    if (!x || !out) { cgs_set_error("%s: NULL", who); return CGS_ERR_ARG; }
    if (((uintptr_t)x & 15) || ((uintptr_t)out & 3)) { cgs_set_error("%s: the input must be 16-byte aligned", who); return CGS_ERR_ARG; }
    return CGS_OK;
}

extern "C" int cgs_scaling_reg_fwd(const float *scaling, int64_t P, float *partials, void *stream) {
    if (int rc = rg_check("scaling_reg_fwd", scaling, partials, P)) return rc;
    CgsProfScope prof(CGS_PROF_LOSS_FWD, (hipStream_t)stream);
    hipLaunchKernelGGL(scaling_reg_fwd_kernel, dim3(rg_grid(P / 4)), dim3(RG_THREADS), 0, (hipStream_t)stream, scaling, P, partials);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

extern "C" int cgs_scaling_reg_bwd(const float *scaling, const float *g, int64_t P, float *d_scaling, void *stream) {
    if (int rc = rg_check("scaling_reg_bwd", scaling, d_scaling, P)) return rc;
    if (!g || ((uintptr_t)d_scaling & 15)) { cgs_set_error("scaling_reg_bwd: g NULL or d_scaling not 16-byte aligned"); return CGS_ERR_ARG; }
    CgsProfScope prof(CGS_PROF_LOSS_BWD, (hipStream_t)stream);
    hipLaunchKernelGGL(scaling_reg_bwd_kernel, dim3(rg_grid(P / 4)), dim3(RG_THREADS), 0, (hipStream_t)stream, scaling, g, P, d_scaling);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

extern "C" int cgs_sigmoid_mean_fwd(const float *x, int64_t n, float *partials, void *stream) {
    if (int rc = rg_check("sigmoid_mean_fwd", x, partials, n)) return rc;
    CgsProfScope prof(CGS_PROF_LOSS_FWD, (hipStream_t)stream);
    hipLaunchKernelGGL(sigmoid_mean_fwd_kernel, dim3(rg_grid(n / 4)), dim3(RG_THREADS), 0, (hipStream_t)stream, x, n, partials);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

extern "C" int cgs_sigmoid_mean_bwd(const float *x, const float *g, int64_t n, float *dx, void *stream) {
    if (int rc = rg_check("sigmoid_mean_bwd", x, dx, n)) return rc;
    if (!g || ((uintptr_t)dx & 15)) { cgs_set_error("sigmoid_mean_bwd: g NULL or dx not 16-byte aligned"); return CGS_ERR_ARG; }
    CgsProfScope prof(CGS_PROF_LOSS_BWD, (hipStream_t)stream);
    hipLaunchKernelGGL(sigmoid_mean_bwd_kernel, dim3(rg_grid(n / 4)), dim3(RG_THREADS), 0, (hipStream_t)stream, x, g, n, dx);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}


// ---- weighted image sum: sum_i img[i] * w[i] (+ lam * rate) and its backward ----------------------------------------------
// The fixed linear objective a throughput measurement puts behind render() (bench.py: sum(image * w) + lambda * bit_per_param,
// train.py:206-209 with the image term made linear) as ONE launch each way instead of dot (2 launches) + add and two muls.
// Deterministic: per-workgroup partial sums in double, the last workgroup to finish adds them in workgroup order and clears the
// ticket for the next call (scratch = WS_BLOCKS doubles + a zero-initialised arrival ticket, cgs_ticket_last; owned by the caller).
#ifndef WS_BLOCKS
#define WS_BLOCKS 512       // (textbook arrival ticket: ~25 ns per workgroup for its __threadfence — 256: 18.8, 512: 26.8, 1024: 47 us; fence-free: 13.4 / 12.5 / 13.1 us)
#endif
__global__ void __launch_bounds__(256)
    wsum_fwd_kernel(const float *__restrict__ a, const float *__restrict__ b, int64_t n, const float *__restrict__ rate, float lam,
                    double *__restrict__ partial, unsigned int *__restrict__ counter, float *__restrict__ out) {
    __shared__ double sh[4];
    __shared__ bool last;
    const int64_t stride = (int64_t)gridDim.x * 256, t0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    // a thread's own terms (a few dozen) are added in fp32, four independent chains with two 16-byte loads per operand in flight;
    // everything across threads and workgroups in double
    float f0 = 0.f, f1 = 0.f, f2 = 0.f, f3 = 0.f;
    const int64_t n4 = ((((uintptr_t)a | (uintptr_t)b) & 15) == 0) ? n / 4 : 0;
    int64_t i = t0;
    for (; i + stride < n4; i += 2 * stride) {
        const float4 x = ((const float4 *)a)[i], y = ((const float4 *)b)[i];
        const float4 x2 = ((const float4 *)a)[i + stride], y2 = ((const float4 *)b)[i + stride];
        f0 = fmaf(x.x, y.x, f0); f1 = fmaf(x.y, y.y, f1); f2 = fmaf(x.z, y.z, f2); f3 = fmaf(x.w, y.w, f3);
        f0 = fmaf(x2.x, y2.x, f0); f1 = fmaf(x2.y, y2.y, f1); f2 = fmaf(x2.z, y2.z, f2); f3 = fmaf(x2.w, y2.w, f3);
    }
    for (; i < n4; i += stride) {
        const float4 x = ((const float4 *)a)[i], y = ((const float4 *)b)[i];
        f0 = fmaf(x.x, y.x, f0); f1 = fmaf(x.y, y.y, f1); f2 = fmaf(x.z, y.z, f2); f3 = fmaf(x.w, y.w, f3);
    }
    for (int64_t j = 4 * n4 + t0; j < n; j += stride) f0 = fmaf(a[j], b[j], f0);
    double acc = ((double)f0 + (double)f1) + ((double)f2 + (double)f3);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        cgs_publish(&partial[blockIdx.x], sh[0] + sh[1] + sh[2] + sh[3]);
        last = cgs_ticket_last(counter);
    }
    __syncthreads();
    if (!last) return;
    double v = 0.0;
    for (int j = threadIdx.x; j < (int)gridDim.x; j += 256) v += cgs_published(&partial[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = (float)(sh[0] + sh[1] + sh[2] + sh[3]) + (rate ? lam * rate[0] : 0.f);
    }
}

__global__ void __launch_bounds__(256)
    wsum_bwd_kernel(const float *__restrict__ g, const float *__restrict__ w, int64_t n, float lam, float *__restrict__ dimg,
                    float *__restrict__ drate) {
    const float gv = g[0];
    const int64_t stride = (int64_t)gridDim.x * 256, t0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n4 = ((((uintptr_t)w | (uintptr_t)dimg) & 15) == 0) ? n / 4 : 0;
    for (int64_t i = t0; i < n4; i += stride) {
        const float4 y = ((const float4 *)w)[i];
        ((float4 *)dimg)[i] = make_float4(gv * y.x, gv * y.y, gv * y.z, gv * y.w);
    }
    for (int64_t i = 4 * n4 + t0; i < n; i += stride) dimg[i] = gv * w[i];
    if (drate && t0 == 0) drate[0] = gv * lam;
}

extern "C" size_t cgs_weighted_sum_scratch_bytes(void) { return (size_t)WS_BLOCKS * sizeof(double) + CGS_TICKET_BYTES; }

extern "C" int cgs_weighted_sum_fwd(const float *img, const float *w, int64_t n, const float *rate, float lam, void *scratch,
                                    size_t scratch_bytes, float *out, void *stream) {
    if (n < 0 || !out || !scratch || scratch_bytes < cgs_weighted_sum_scratch_bytes() || (n && (!img || !w))) {
        cgs_set_error("weighted_sum_fwd: bad args");
        return CGS_ERR_ARG;
    }
    double *partial = (double *)scratch;
    unsigned int *counter = (unsigned int *)((char *)scratch + (size_t)WS_BLOCKS * sizeof(double));
    const int64_t want = (n / 4 + 255) / 256;
    const unsigned grid = (unsigned)(want < 1 ? 1 : (want > WS_BLOCKS ? WS_BLOCKS : want));
    hipLaunchKernelGGL(wsum_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, img, w, n, rate, lam, partial, counter, out);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

extern "C" int cgs_weighted_sum_bwd(const float *g, const float *w, int64_t n, float lam, float *dimg, float *drate, void *stream) {
    if (n < 0 || !g || (n && (!w || !dimg))) { cgs_set_error("weighted_sum_bwd: bad args"); return CGS_ERR_ARG; }
    const int64_t want = (n / 4 + 255) / 256;
    const unsigned grid = (unsigned)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
    hipLaunchKernelGGL(wsum_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, w, n, lam, dimg, drate);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}
