// generate_neural_gaussians' anchor -> Gaussian stage as ONE kernel family
// (gaussian_renderer/__init__.py:106-145): the three anchor MLPs (mlp_opacity 54->50->10 tanh,
// mlp_color 54->50->30 sigmoid, mlp_cov 54->50->70), the opacity mask, the survivor compaction and
// the per-Gaussian element-wise tail, without the 110-float MLP outputs, the 150-float hidden
// layer, their gradients or the [n,54] input ever reaching HBM:
//
//   ag_opacity_kernel   opacity head only -> neural_opacity, selection mask, a survivor bit mask per
//                       anchor and a survivor count per 16 anchors              (:112-116)
//   ag_scan_kernel      exclusive scan of those counts (one workgroup)           (boolean indexing, :137)
//   ag_write_kernel     colour + covariance heads; every lane ends up holding ALL eleven head outputs
//                       of its own (anchor, offset) slots and writes the compacted Gaussians
//                       directly                                                  (:122-145)
//   ag_bwd_kernel       prologue = backward of the element-wise tail straight into the MFMA
//                       B-operand registers; hidden layer RECOMPUTED from the gathered input rows;
//                       dZ2 -> dH -> dZ1 -> dX on the matrix cores; weight / bias gradients of all six
//                       layers in the same kernel (row index as MFMA contraction, 16x16 register
//                       tiles transposed through a 1.25 KB per-wave LDS patch, partial products
//                       accumulated in an LDS image of [dW | db], one image per workgroup summed by
//                       wgrad_multi_reduce_kernel)
//
// The trick that removes every layout shuffle: the OUTPUT index of a second-layer weight matrix is
// free to permute when the matrix is staged into LDS.  With the transposed MFMA chaining of
// mlp_frag.h lane (g, c) owns outputs 16u + 4g + {0..3} of anchor row c; the staging order below
// makes those "virtual" outputs the channels of offset slots 3g, 3g+1, 3g+2 of that anchor, for all
// three heads, so the element-wise tail (and its backward) runs on registers the lane already holds.
// Virtual tiles: opacity 1, colour 3, covariance 6 (8 in the natural order).
#include "cgs_internal.h"
#include "mlp_frag.h"

#define AG_K 10              // offsets per anchor the heads are shaped for (10 / 30 / 70 outputs)
#define AG_IN 54
#define AG_HID 50
#define AG_NTI 4             // ceil(54/16)
#define AG_NT1 4             // ceil(50/16)
#define AG_XP 64
#define AG_HP 64

struct AgRows {
    const float *feat_src;      // [*, 50]
    const int64_t *feat_row;    // [n] row of feat_src per visible anchor, or NULL = identity
    const float *anchor;        // [n, 3]
    const float *cam;           // [3]
};

// virtual output vo = 16u + 4g + j of a head with C channels per slot  ->  actual output, or -1
template <int C>
__host__ __device__ __forceinline__ int ag_vo_to_out(int vo) {
    const int u = vo >> 4, g = (vo >> 2) & 3, j = vo & 3, v = 4 * u + j;
    if (v >= 3 * C) return -1;
    const int slot = 3 * g + v / C;
    return slot < AG_K ? slot * C + v % C : -1;
}

// TM: the last piece in the TAIL ORDER of mlp3.hip's forward (lane g, slot j holds input 48 + g + 4 j: the six inputs 48..53 in two
// MFMA k-steps instead of four) — the forward kernels of this file keep the k order of mlp3_fwd_kernel, bit for bit
template <bool TM = false>
__device__ __forceinline__ f32x4 ag_load_x(const AgRows &R, int64_t row, int q, int g, bool valid) {
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!valid) return v;
    const float *fr = R.feat_src + (R.feat_row ? R.feat_row[row] : row) * AG_HID;
    const int col0 = 16 * q + 4 * g;
    if (col0 + 3 < AG_HID) return *(const f32x4_a4 *)(fr + col0);
    if (TM) {
        const float ux = R.anchor[3 * row] - R.cam[0], uy = R.anchor[3 * row + 1] - R.cam[1], uz = R.anchor[3 * row + 2] - R.cam[2];
        const float dist = sqrtf(ux * ux + uy * uy + uz * uz);
        v[0] = g == 0 ? fr[48] : (g == 1 ? fr[49] : (g == 2 ? ux / dist : uy / dist));
        v[1] = g == 0 ? uz / dist : (g == 1 ? dist : 0.f);
        return v;
    }
    if (col0 >= AG_IN) return v;
    const float ux = R.anchor[3 * row] - R.cam[0], uy = R.anchor[3 * row + 1] - R.cam[1], uz = R.anchor[3 * row + 2] - R.cam[2];
    const float dist = sqrtf(ux * ux + uy * uy + uz * uz);
    if (col0 == 48) { v[0] = fr[48]; v[1] = fr[49]; v[2] = ux / dist; v[3] = uy / dist; }
    else { v[0] = uz / dist; v[1] = dist; }                          // col0 == 52
    return v;
}

// exclusive prefix of x over the 16 lanes that share g (c = lane & 15)
__device__ __forceinline__ int ag_row16_excl_scan(int x, int c) {
    int v = x;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        const int t = __shfl_up(v, d, 16);
        if (c >= d) v += t;
    }
    return v - x;
}

// ---- forward ---------------------------------------------------------------------------------
template <int C>
struct AgFwdLds {
    static constexpr int NT2 = (3 * C + 3) / 4, VO = 16 * NT2, OUT = AG_K * C;
    static constexpr int S1 = frag_pad4mod8(AG_HP), S2 = frag_pad4mod8(VO);
    static constexpr int FLOATS = AG_XP * S1 + AG_HP * S2 + AG_HP + VO;
};

template <int C>
__device__ __forceinline__ void ag_stage_fwd(float *lds, const float *W1, const float *b1, const float *W2,
                                             const float *b2, int tid, int nthr) {
    using L = AgFwdLds<C>;
    float *W1s = lds, *W2s = W1s + AG_XP * L::S1, *b1s = W2s + AG_HP * L::S2, *b2s = b1s + AG_HP;
    for (int i = tid; i < AG_XP * L::S1; i += nthr) {
        const int k = i / L::S1, j = i % L::S1;
        W1s[i] = (k < AG_IN && j < AG_HID) ? W1[j * AG_IN + k] : 0.f;
    }
    for (int i = tid; i < AG_HP * L::S2; i += nthr) {
        const int hh = i / L::S2, vo = i % L::S2;
        const int o = vo < L::VO ? ag_vo_to_out<C>(vo) : -1;
        W2s[i] = (hh < AG_HID && o >= 0) ? W2[o * AG_HID + hh] : 0.f;
    }
    for (int i = tid; i < AG_HP; i += nthr) b1s[i] = i < AG_HID ? b1[i] : 0.f;
    for (int i = tid; i < L::VO; i += nthr) {
        const int o = ag_vo_to_out<C>(i);
        b2s[i] = o >= 0 ? b2[o] : 0.f;
    }
}

// y[u][rt][j] = act(head(x))[virtual output 16u + 4g + j] of row c
template <int C, int ACT, int RT, bool TM = false>
__device__ __forceinline__ void ag_head_fwd(const float *lds, const f32x4 (&xb)[RT][AG_NTI], int g, int c,
                                            f32x4 (&y)[AgFwdLds<C>::NT2][RT]) {
    using L = AgFwdLds<C>;
    const float *W1s = lds, *W2s = W1s + AG_XP * L::S1, *b1s = W2s + AG_HP * L::S2, *b2s = b1s + AG_HP;
    f32x4 acc1[AG_NT1][RT];
#pragma unroll
    for (int t = 0; t < AG_NT1; ++t)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc1[t][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < AG_NTI; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (16 * q + j >= AG_IN) continue;
            if (TM && q == AG_NTI - 1 && 16 * q + 4 * j >= AG_IN) continue;          // tail order: inputs 48 + g + 4 j
#pragma unroll
            for (int t = 0; t < AG_NT1; ++t) {
                const float a = W1s[((TM && q == AG_NTI - 1) ? 16 * q + g + 4 * j : 16 * q + 4 * g + j) * L::S1 + 16 * t + c];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc1[t][rt] = frag_mfma(a, xb[rt][q][j], acc1[t][rt]);
            }
        }
#pragma unroll
    for (int t = 0; t < AG_NT1; ++t)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc1[t][rt][r] = fmaxf(acc1[t][rt][r] + b1s[16 * t + 4 * g + r], 0.f);
#pragma unroll
    for (int u = 0; u < L::NT2; ++u)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) y[u][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < AG_NT1; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (16 * t + r >= AG_HID) continue;
#pragma unroll
            for (int u = 0; u < L::NT2; ++u) {
                const float a = W2s[(16 * t + 4 * g + r) * L::S2 + 16 * u + c];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) y[u][rt] = frag_mfma(a, acc1[t][rt][r], y[u][rt]);
            }
        }
#pragma unroll
    for (int u = 0; u < L::NT2; ++u)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) y[u][rt][r] = frag_act<ACT>(y[u][rt][r] + b2s[16 * u + 4 * g + r]);
}

// Opacity head + mask (:112-116, :129-130).  mask: [n, K] values of get_mask for the visible anchors.
// Outputs: neural_opacity [n*K], y_op [n, K] (tanh output, training only), mask_out [n*K] (bool bytes),
// bits [n] (bit k = slot k survives), cnt16 [ceil(n/16)] survivors per 16 anchors.
template <int RT, int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
    ag_opacity_kernel(AgRows R, const float *__restrict__ W1, const float *__restrict__ b1, const float *__restrict__ W2,
                      const float *__restrict__ b2, const float *__restrict__ mask, float *__restrict__ y_op,
                      float *__restrict__ nop, uint8_t *__restrict__ mask_out, uint32_t *__restrict__ bits,
                      uint32_t *__restrict__ cnt16, int64_t n) {
    __shared__ float lds[AgFwdLds<1>::FLOATS];
    const int tid = threadIdx.x, nthr = WAVES * 64;
    ag_stage_fwd<1>(lds, W1, b1, W2, b2, tid, nthr);
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int64_t ntiles = (n + 16 * RT - 1) / (16 * RT);
    for (int64_t tile = (int64_t)blockIdx.x * WAVES + wave; tile < ntiles; tile += (int64_t)gridDim.x * WAVES) {
        const int64_t row0 = tile * 16 * RT;
        asm volatile("" ::: "memory");
        f32x4 xb[RT][AG_NTI];
        bool valid[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int64_t row = row0 + rt * 16 + c;
            valid[rt] = row < n;
#pragma unroll
            for (int q = 0; q < AG_NTI; ++q) xb[rt][q] = ag_load_x<true>(R, row, q, g, valid[rt]);
        }
        f32x4 y[1][RT];
        ag_head_fwd<1, FRAG_ACT_TANH, RT, true>(lds, xb, g, c, y);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int64_t row = row0 + rt * 16 + c;
            uint32_t lb = 0;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int slot = 3 * g + s;
                if (valid[rt] && slot < AG_K) {
                    const int64_t i = row * AG_K + slot;
                    const float yo = y[0][rt][s];
                    const float v = yo * mask[i];
                    nop[i] = v;
                    if (y_op) y_op[i] = yo;
                    const bool f = v > 0.f;
                    if (mask_out) mask_out[i] = f ? 1 : 0;
                    lb |= f ? (1u << slot) : 0u;
                }
            }
            uint32_t rb = lb;
            rb |= __shfl_xor(rb, 16);
            rb |= __shfl_xor(rb, 32);
            if (g == 0 && valid[rt]) bits[row] = rb;
            // survivors of these 16 anchors: every lane's own slots, summed over the wave
            int cnt = __popc(lb);
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) cnt += __shfl_xor(cnt, d);
            const int64_t blk = tile * RT + rt;
            if (lane == 0 && blk * 16 < n) cnt16[blk] = (uint32_t)cnt;
        }
    }
}

// In-place exclusive scan of cnt[0..nblk) by ONE workgroup; cnt[nblk] receives the total.  Wave w owns a contiguous
// chunk and walks it 64 entries at a time (coalesced): pass 1 sums the chunk, pass 2 scans it with the running carry.
__global__ void __launch_bounds__(1024) ag_scan_kernel(uint32_t *__restrict__ cnt, int64_t nblk) {
    __shared__ uint32_t wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t per = ((nblk + 15) / 16 + 63) / 64 * 64;          // chunk per wave, multiple of 64
    const int64_t b = (int64_t)wave * per, e = min(nblk, b + per);
    uint32_t s = 0;
    for (int64_t i = b + lane; i < e; i += 64) s += cnt[i];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) s += __shfl_xor(s, d);
    if (lane == 0) wsum[wave] = s;
    __syncthreads();
    uint32_t carry = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const uint32_t t = wsum[w];
        if (w < wave) carry += t;
        total += t;
    }
    for (int64_t i0 = b; i0 < e; i0 += 64) {
        const int64_t i = i0 + lane;
        const uint32_t v = i < e ? cnt[i] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        if (i < e) cnt[i] = carry + inc - v;
        carry += __shfl(inc, 63, 64);
    }
    if (tid == 0) cnt[nblk] = total;
}

struct AgGeo {
    const float *gs_src;        // [*, 6]  grid scaling rows
    const float *off_src;       // [*, K*3] offsets rows
    const int64_t *geo_row;     // [n] row of gs_src / off_src per visible anchor, or NULL = identity
};

// Colour + covariance heads and the per-Gaussian tail (:122-145).  siginv [P,4] (training): sigmoid of the three
// scale logits and 1/|q| of the rotation logits (negated when |q| <= 1e-12, where F.normalize divides by eps).
template <int RT, int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
    ag_write_kernel(AgRows R, AgGeo G, const float *__restrict__ nop, const uint32_t *__restrict__ bits,
                    const uint32_t *__restrict__ base16, const float *W1c, const float *b1c, const float *W2c,
                    const float *b2c, const float *W1v, const float *b1v, const float *W2v, const float *b2v,
                    float *__restrict__ xyz, float *__restrict__ color, float *__restrict__ opacity,
                    float *__restrict__ scaling, float *__restrict__ rot, float *__restrict__ siginv, int64_t n) {
    __shared__ float lds[AgFwdLds<3>::FLOATS + AgFwdLds<7>::FLOATS];
    float *lc = lds, *lv = lds + AgFwdLds<3>::FLOATS;
    const int tid = threadIdx.x, nthr = WAVES * 64;
    ag_stage_fwd<3>(lc, W1c, b1c, W2c, b2c, tid, nthr);
    ag_stage_fwd<7>(lv, W1v, b1v, W2v, b2v, tid, nthr);
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int64_t ntiles = (n + 16 * RT - 1) / (16 * RT);
    f32x4 xb[RT][AG_NTI], xn[RT][AG_NTI];
    bool valid[RT], validn[RT];
    const int64_t tile0 = (int64_t)blockIdx.x * WAVES + wave, tstride = (int64_t)gridDim.x * WAVES;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int64_t row = tile0 * 16 * RT + rt * 16 + c;
        valid[rt] = tile0 < ntiles && row < n;
#pragma unroll
        for (int q = 0; q < AG_NTI; ++q) xb[rt][q] = ag_load_x<true>(R, row, q, g, valid[rt]);
    }
    for (int64_t tile = tile0; tile < ntiles; tile += tstride) {
        const int64_t row0 = tile * 16 * RT;
        asm volatile("" ::: "memory");
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int64_t row = (tile + tstride) * 16 * RT + rt * 16 + c;
            validn[rt] = row < n;
#pragma unroll
            for (int q = 0; q < AG_NTI; ++q) xn[rt][q] = ag_load_x<true>(R, row, q, g, validn[rt]);
        }
        f32x4 yc[AgFwdLds<3>::NT2][RT], yv[AgFwdLds<7>::NT2][RT];
        ag_head_fwd<3, FRAG_ACT_SIGMOID, RT, true>(lc, xb, g, c, yc);
        ag_head_fwd<7, FRAG_ACT_NONE, RT, true>(lv, xb, g, c, yv);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int64_t row = row0 + rt * 16 + c;
            const int64_t blk = tile * RT + rt;
            const uint32_t rb = valid[rt] ? bits[row] : 0u;
            const int excl = ag_row16_excl_scan(__popc(rb), c);
            if (!rb) continue;
            size_t j = (size_t)base16[blk] + excl + __popc(rb & ((1u << (3 * g)) - 1u));
            const int64_t sn = G.geo_row ? G.geo_row[row] : row;
            const float *gs = G.gs_src + 6 * sn;
            const float gs0 = gs[0], gs1 = gs[1], gs2 = gs[2], gs3 = gs[3], gs4 = gs[4], gs5 = gs[5];
            const float ax = R.anchor[3 * row], ay = R.anchor[3 * row + 1], az = R.anchor[3 * row + 2];
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int slot = 3 * g + s;
                if (slot >= AG_K || !((rb >> slot) & 1u)) continue;
                const float *of = G.off_src + (sn * AG_K + slot) * 3;
                // registers of this slot: colour channel ch = virtual 3s + ch, covariance 7s + ch
#define AG_YC(ch) yc[(3 * s + (ch)) >> 2][rt][(3 * s + (ch)) & 3]
#define AG_YV(ch) yv[(7 * s + (ch)) >> 2][rt][(7 * s + (ch)) & 3]
                color[3 * j] = AG_YC(0);
                color[3 * j + 1] = AG_YC(1);
                color[3 * j + 2] = AG_YC(2);
                const float s0 = 1.f / (1.f + __expf(-AG_YV(0))), s1 = 1.f / (1.f + __expf(-AG_YV(1))),
                            s2 = 1.f / (1.f + __expf(-AG_YV(2)));
                scaling[3 * j] = gs3 * s0;
                scaling[3 * j + 1] = gs4 * s1;
                scaling[3 * j + 2] = gs5 * s2;
                xyz[3 * j] = ax + of[0] * gs0;
                xyz[3 * j + 1] = ay + of[1] * gs1;
                xyz[3 * j + 2] = az + of[2] * gs2;
                opacity[j] = nop[row * AG_K + slot];
                const float q0 = AG_YV(3), q1 = AG_YV(4), q2 = AG_YV(5), q3 = AG_YV(6);
                const float nrm = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
                const float inv = 1.f / fmaxf(nrm, 1e-12f);                        // F.normalize eps
                *(f32x4 *)(rot + 4 * j) = (f32x4){q0 * inv, q1 * inv, q2 * inv, q3 * inv};
                if (siginv) *(f32x4 *)(siginv + 4 * j) = (f32x4){s0, s1, s2, nrm > 1e-12f ? inv : -inv};
#undef AG_YC
#undef AG_YV
                ++j;
            }
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            valid[rt] = validn[rt];
#pragma unroll
            for (int q = 0; q < AG_NTI; ++q) xb[rt][q] = xn[rt][q];
        }
    }
}

// ---- backward --------------------------------------------------------------------------------
#define AG_S1P 57            // W1 image [hidden row][input col], rows of the three heads packed (150 + 14 pad rows)
#define AG_W1P_ROWS 164
#define AG_W1P_FLOATS (AG_W1P_ROWS * AG_S1P + 8)
#define AG_S2P 52            // W2 image [virtual output][hidden]
#define AG_VO_TOTAL 160      // 16 + 48 + 96
#define AG_W2P_FLOATS ((AG_VO_TOTAL + 1) * AG_S2P)
#define AG_B1P 164
#define AG_TS 20             // row stride of a transposition patch
#define AG_PATCH (16 * AG_TS)
// image of one workgroup's weight-gradient partial sums, laid out [dW | db] per product like mlp_wgrad.hip's
#define AG_E_W1 (150 * 54)
#define AG_OFF_OP (AG_E_W1 + 150)
#define AG_OFF_COL (AG_OFF_OP + 10 * 50 + 10)
#define AG_OFF_COV (AG_OFF_COL + 30 * 50 + 30)
#define AG_E (AG_OFF_COV + 70 * 50 + 70)

#define AG_IMG_ADD(p, v) atomicAdd((p), (v))

struct AgBwdArgs {
    AgRows R;
    AgGeo G;
    const float *mask, *y_op;               // [n, K]
    const uint32_t *bits, *base16;
    const float *color, *rot, *siginv;      // forward outputs [P,3], [P,4], [P,4]
    const float *g_xyz, *g_color, *g_opacity, *g_scaling, *g_rot;   // [P,.]
    const float *g_nop;                      // [n*K] gradient of neural_opacity, may be NULL
    const float *W1[3], *b1[3], *W2[3];      // opacity, colour, covariance
    float *d_feat_src, *d_anchor, *d_gs, *d_off, *d_mask;
    // FUSED == false: operands of the separate weight-gradient launch
    float *X_out, *Hcat, *dZ1cat, *dZ2_op, *dZ2_col, *dZ2_cov;
    // FUSED == true: [gridDim.x][AG_E]
    float *partial;
    int64_t n;
};

template <int C>
__device__ __forceinline__ void ag_stage_w2p(float *W2p, int vo_base, const float *W2, int tid, int nthr) {
    constexpr int VO = 16 * ((3 * C + 3) / 4);
    for (int i = tid; i < VO * AG_S2P; i += nthr) {
        const int vo = i / AG_S2P, hh = i % AG_S2P;
        const int o = ag_vo_to_out<C>(vo);
        W2p[(vo_base + vo) * AG_S2P + hh] = (o >= 0 && hh < AG_HID) ? W2[o * AG_HID + hh] : 0.f;
    }
}

// 16x16 register tile (lane (g,c): columns 4g..4g+3 of row c) -> lane (g,c): rows 4g..4g+3 of column c
__device__ __forceinline__ f32x4 ag_transpose(float *patch, f32x4 v, int g, int c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) patch[(4 * g + j) * AG_TS + c] = v[j];
    __builtin_amdgcn_wave_barrier();
    const f32x4 r = *(const f32x4 *)(patch + c * AG_TS + 4 * g);
    __builtin_amdgcn_wave_barrier();
    return r;
}

// one head: recompute H, dH = W2^T dZ2, dZ1, dX += W1^T dZ1, then either store the wgrad operands or do the
// weight gradients here.  b[u][j] = dZ2 of virtual output 16u + 4g + j of row c.
template <int C, int HEAD, int VO_BASE, int IMG_OFF, bool FUSED>
__device__ __forceinline__ void ag_head_bwd(const float *W1p, const float *W2p, const float *b1p, float *img, float *patch,
                                            const f32x4 (&xb)[AG_NTI], const f32x4 (&xT)[AG_NTI],
                                            const f32x4 (&b)[(3 * C + 3) / 4], bool valid, int64_t row, int g, int c,
                                            const AgBwdArgs &a, f32x4 (&adx)[AG_NTI]) {
    constexpr int NT2 = (3 * C + 3) / 4, OUT = AG_K * C;
    f32x4 H[AG_NT1], dZ[AG_NT1];
#pragma unroll
    for (int t = 0; t < AG_NT1; ++t) { H[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; dZ[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    // hidden layer again (same k order as the forward: bit-identical relu decisions)
#pragma unroll
    for (int q = 0; q < AG_NTI; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (16 * q + j >= AG_IN) continue;
#pragma unroll
            for (int t = 0; t < AG_NT1; ++t) {
                const float w = W1p[(AG_HID * HEAD + 16 * t + c) * AG_S1P + 16 * q + 4 * g + j];
                H[t] = frag_mfma(w, xb[q][j], H[t]);
            }
        }
#pragma unroll
    for (int t = 0; t < AG_NT1; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool real = 16 * t + 4 * g + r < AG_HID;      // rows past 50 of this head's image are the next head's
            const float h = fmaxf(H[t][r] + b1p[AG_HID * HEAD + 16 * t + 4 * g + r], 0.f);
            H[t][r] = real ? h : 0.f;
        }
#pragma unroll
    for (int u = 0; u < NT2; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (4 * u + j >= 3 * C) continue;
#pragma unroll
            for (int t = 0; t < AG_NT1; ++t) {
                const float w = W2p[(VO_BASE + 16 * u + 4 * g + j) * AG_S2P + 16 * t + c];
                dZ[t] = frag_mfma(w, b[u][j], dZ[t]);
            }
        }
#pragma unroll
    for (int t = 0; t < AG_NT1; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) dZ[t][r] = H[t][r] > 0.f ? dZ[t][r] : 0.f;      // also zeroes the padded features
#pragma unroll
    for (int t = 0; t < AG_NT1; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (16 * t + r >= AG_HID) continue;
#pragma unroll
            for (int v = 0; v < AG_NTI; ++v) {
                const float w = W1p[(AG_HID * HEAD + 16 * t + 4 * g + r) * AG_S1P + 16 * v + c];
                adx[v] = frag_mfma(w, dZ[t][r], adx[v]);
            }
        }
    if (!FUSED) {
#pragma unroll
        for (int t = 0; t < AG_NT1; ++t) {
            frag_store4<AG_HID>(a.Hcat + row * 150 + AG_HID * HEAD, t, g, valid, H[t]);
            frag_store4<AG_HID>(a.dZ1cat + row * 150 + AG_HID * HEAD, t, g, valid, dZ[t]);
        }
        return;
    }
    // ---- weight gradients: row index as the MFMA contraction ----
    // dW1[hid][col] += sum_rows dZ1[row][hid] X[row][col]  (col 54 of the padded X tile is 1: the bias gradient)
#pragma unroll
    for (int t = 0; t < AG_NT1; ++t) {
        const f32x4 aT = ag_transpose(patch, dZ[t], g, c);
#pragma unroll
        for (int v = 0; v < AG_NTI; ++v) {
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = frag_mfma(aT[s], xT[v][s], acc);
            const int col = 16 * v + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int hid = 16 * t + 4 * g + r;
                if (16 * t + r >= AG_HID) continue;
                if (hid < AG_HID) {
                    if (col < AG_IN) AG_IMG_ADD(&img[(AG_HID * HEAD + hid) * AG_IN + col], acc[r]);
                    else if (col == AG_IN) AG_IMG_ADD(&img[AG_E_W1 + AG_HID * HEAD + hid], acc[r]);
                }
            }
        }
    }
    // dW2[o][hid] += sum_rows dZ2[row][o] H[row][hid]  (hidden column 50 of the padded H tile is 1: the bias gradient)
    f32x4 hT[AG_NT1];
#pragma unroll
    for (int t = 0; t < AG_NT1; ++t) hT[t] = ag_transpose(patch + AG_PATCH, H[t], g, c);
    if (c == 2) hT[3] = (f32x4){1.f, 1.f, 1.f, 1.f};
#pragma unroll
    for (int u = 0; u < NT2; ++u) {
        const f32x4 aT = ag_transpose(patch, b[u], g, c);
#pragma unroll
        for (int t = 0; t < AG_NT1; ++t) {
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = frag_mfma(aT[s], hT[t][s], acc);
            const int hid = 16 * t + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (4 * u + r >= 3 * C) continue;
                const int slot = 3 * g + (4 * u + r) / C;
                const int o = slot * C + (4 * u + r) % C;
                if (slot < AG_K) {
                    if (hid < AG_HID) AG_IMG_ADD(&img[IMG_OFF + o * AG_HID + hid], acc[r]);
                    else if (hid == AG_HID) AG_IMG_ADD(&img[IMG_OFF + OUT * AG_HID + o], acc[r]);
                }
            }
        }
    }
}

template <int WAVES, bool FUSED>
__global__ void __launch_bounds__(WAVES * 64) ag_bwd_kernel(AgBwdArgs a) {
    __shared__ float lds[AG_W1P_FLOATS + AG_W2P_FLOATS + AG_B1P + (FUSED ? AG_E + WAVES * 2 * AG_PATCH : 0)];
    float *W1p = lds, *W2p = W1p + AG_W1P_FLOATS, *b1p = W2p + AG_W2P_FLOATS, *img = b1p + AG_B1P;
    const int tid = threadIdx.x, nthr = WAVES * 64;
    for (int i = tid; i < AG_W1P_FLOATS; i += nthr) {
        const int row = i / AG_S1P, k = i % AG_S1P;
        W1p[i] = (row < 150 && k < AG_IN) ? a.W1[row / AG_HID][(row % AG_HID) * AG_IN + k] : 0.f;
    }
    for (int i = tid; i < AG_S2P; i += nthr) W2p[AG_VO_TOTAL * AG_S2P + i] = 0.f;
    ag_stage_w2p<1>(W2p, 0, a.W2[0], tid, nthr);
    ag_stage_w2p<3>(W2p, 16, a.W2[1], tid, nthr);
    ag_stage_w2p<7>(W2p, 64, a.W2[2], tid, nthr);
    for (int i = tid; i < AG_B1P; i += nthr) b1p[i] = i < 150 ? a.b1[i / AG_HID][i % AG_HID] : 0.f;
    if (FUSED)
        for (int i = tid; i < AG_E; i += nthr) img[i] = 0.f;
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    float *patch = img + AG_E + wave * 2 * AG_PATCH;
    const int64_t n = a.n, ntiles = (n + 15) / 16;
    for (int64_t tile = (int64_t)blockIdx.x * WAVES + wave; tile < ntiles; tile += (int64_t)gridDim.x * WAVES) {
        const int64_t row = tile * 16 + c;
        const bool valid = row < n;
        asm volatile("" ::: "memory");
        // ---- prologue: backward of the per-Gaussian tail for this lane's slots 3g..3g+2 of anchor `row` ----
        const uint32_t rb = valid ? a.bits[row] : 0u;
        const int excl = ag_row16_excl_scan(__popc(rb), c);
        size_t j = (size_t)a.base16[tile] + excl + __popc(rb & ((1u << (3 * g)) - 1u));
        const int64_t sn = valid ? (a.G.geo_row ? a.G.geo_row[row] : row) : 0;
        float gs[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) gs[k] = valid ? a.G.gs_src[6 * sn + k] : 0.f;
        float sum9[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x4 b_op[1], b_col[3], b_cov[6];
        b_op[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 3; ++u) b_col[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 6; ++u) b_cov[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int slot = 3 * g + s;
            const bool live = valid && slot < AG_K;
            if (!live) continue;
            const int64_t i = row * AG_K + slot, si = sn * AG_K + slot;
            float g_no = a.g_nop ? a.g_nop[i] : 0.f;
            float doff[3] = {0.f, 0.f, 0.f}, dcol[3] = {0.f, 0.f, 0.f}, dsr[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if ((rb >> slot) & 1u) {
                g_no += a.g_opacity[j];
                const f32x4 sv = *(const f32x4 *)(a.siginv + 4 * j);
                const f32x4 rr = *(const f32x4 *)(a.rot + 4 * j), gr = *(const f32x4 *)(a.g_rot + 4 * j);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float y = a.color[3 * j + k];
                    dcol[k] = a.g_color[3 * j + k] * (y * (1.f - y));
                    const float gx = a.g_xyz[3 * j + k], gsc = a.g_scaling[3 * j + k], sig = sv[k];
                    doff[k] = gx * gs[k];
                    dsr[k] = gsc * gs[3 + k] * sig * (1.f - sig);
                    sum9[k] += gx;
                    sum9[3 + k] += gx * a.G.off_src[3 * si + k];
                    sum9[6 + k] += gsc * sig;
                }
                const float inv = sv[3];
                if (inv >= 0.f) {
                    const float dot = rr[0] * gr[0] + rr[1] * gr[1] + rr[2] * gr[2] + rr[3] * gr[3];
#pragma unroll
                    for (int k = 0; k < 4; ++k) dsr[3 + k] = (gr[k] - rr[k] * dot) * inv;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) dsr[3 + k] = gr[k] * (-inv);
                }
                ++j;
            }
            const float yo = a.y_op[i];
            a.d_mask[i] = g_no * yo;
            b_op[0][s] = g_no * a.mask[i] * (1.f - yo * yo);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                a.d_off[3 * si + k] = doff[k];
                b_col[(3 * s + k) >> 2][(3 * s + k) & 3] = dcol[k];
            }
#pragma unroll
            for (int k = 0; k < 7; ++k) b_cov[(7 * s + k) >> 2][(7 * s + k) & 3] = dsr[k];
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            sum9[k] += __shfl_xor(sum9[k], 16);
            sum9[k] += __shfl_xor(sum9[k], 32);
        }
        if (g == 0 && valid) {
#pragma unroll
            for (int k = 0; k < 6; ++k) a.d_gs[6 * sn + k] = sum9[3 + k];
        }
        // ---- MLP backward ----
        f32x4 xb[AG_NTI], xT[AG_NTI], adx[AG_NTI];
#pragma unroll
        for (int q = 0; q < AG_NTI; ++q) {
            xb[q] = ag_load_x(a.R, row, q, g, valid);
            adx[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (FUSED) {
#pragma unroll
            for (int q = 0; q < AG_NTI; ++q) xT[q] = ag_transpose(patch, xb[q], g, c);
            if (c == 6) xT[3] = (f32x4){1.f, 1.f, 1.f, 1.f};
        } else {
#pragma unroll
            for (int q = 0; q < AG_NTI; ++q) {
                xT[q] = xb[q];
                frag_store4<AG_IN>(a.X_out + row * AG_IN, q, g, valid, xb[q]);
            }
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int slot = 3 * g + s;
                if (!valid || slot >= AG_K) continue;
                a.dZ2_op[row * 10 + slot] = b_op[0][s];
#pragma unroll
                for (int k = 0; k < 3; ++k) a.dZ2_col[row * 30 + slot * 3 + k] = b_col[(3 * s + k) >> 2][(3 * s + k) & 3];
#pragma unroll
                for (int k = 0; k < 7; ++k) a.dZ2_cov[row * 70 + slot * 7 + k] = b_cov[(7 * s + k) >> 2][(7 * s + k) & 3];
            }
        }
        ag_head_bwd<1, 0, 0, AG_OFF_OP, FUSED>(W1p, W2p, b1p, img, patch, xb, xT, b_op, valid, row, g, c, a, adx);
        ag_head_bwd<3, 1, 16, AG_OFF_COL, FUSED>(W1p, W2p, b1p, img, patch, xb, xT, b_col, valid, row, g, c, a, adx);
        ag_head_bwd<7, 2, 64, AG_OFF_COV, FUSED>(W1p, W2p, b1p, img, patch, xb, xT, b_cov, valid, row, g, c, a, adx);
        // dX[:, 0:50] -> rows of the feature source's gradient; the four view columns are pulled back to the anchor
        // (u = a - cam, v = u/|u|:  da = (dv - v (v.dv)) / |u| + v d|u|) and joined by the tail's d(xyz) sum
        {
            const int64_t srow = valid ? (a.R.feat_row ? a.R.feat_row[row] : row) : 0;
            float *dst = a.d_feat_src + srow * AG_HID;
#pragma unroll
            for (int v = 0; v < 3; ++v)
                if (valid) *(f32x4_a4 *)(dst + 16 * v + 4 * g) = adx[v];
            const float z52 = __shfl(adx[3][0], 16 + c, 64), z53 = __shfl(adx[3][1], 16 + c, 64);
            if (g == 0 && valid) {
                dst[48] = adx[3][0];
                dst[49] = adx[3][1];
                const float dvx = adx[3][2], dvy = adx[3][3], dvz = z52, dd = z53;
                const float ux = a.R.anchor[3 * row] - a.R.cam[0], uy = a.R.anchor[3 * row + 1] - a.R.cam[1],
                            uz = a.R.anchor[3 * row + 2] - a.R.cam[2];
                const float dist = sqrtf(ux * ux + uy * uy + uz * uz), inv = 1.f / dist;
                const float vx = ux * inv, vy = uy * inv, vz = uz * inv;
                const float dot = vx * dvx + vy * dvy + vz * dvz;
                a.d_anchor[3 * row] = ((dvx - vx * dot) * inv + vx * dd) + sum9[0];
                a.d_anchor[3 * row + 1] = ((dvy - vy * dot) * inv + vy * dd) + sum9[1];
                a.d_anchor[3 * row + 2] = ((dvz - vz * dot) * inv + vz * dd) + sum9[2];
            }
        }
    }
    if (FUSED) {
        __syncthreads();
        float *dst = a.partial + (int64_t)blockIdx.x * AG_E;
        for (int i = tid; i < AG_E; i += nthr) dst[i] = img[i];
    }
}

// ---- host side of the C-ABI --------------------------------------------------------------------
static int ag_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

static int ag_grid(int64_t tiles, int waves) {
    const int64_t want = (tiles + waves - 1) / waves;
    return (int)(want < ag_cus() ? want : ag_cus());
}

// Opacity head, mask, survivor bookkeeping.  base16 [ceil(n/16) + 1]: exclusive prefix of the survivor counts per
// 16 anchors, total in the last entry; *count_host receives the total (ONE stream synchronisation — the reference's
// boolean indexing, gaussian_renderer/__init__.py:137, has the same one).
extern "C" int cgs_anchor_gen_count(const float *feat_src, const int64_t *feat_row, const float *anchor_vis,
                                    const float *cam3, const float *mask, const float *W1, const float *b1,
                                    const float *W2, const float *b2, float *y_op, float *neural_opacity,
                                    uint8_t *mask_out, uint32_t *bits, uint32_t *base16, int64_t n, int K,
                                    int64_t *count_host, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0 || K != AG_K) { cgs_set_error("anchor_gen_count: n < 0 or K != 10"); return CGS_ERR_ARG; }
    if (!count_host) { cgs_set_error("anchor_gen_count: NULL count_host"); return CGS_ERR_ARG; }
    *count_host = 0;
    if (n == 0) return CGS_OK;
    if (n * AG_K >= (1ll << 31)) { cgs_set_error("anchor_gen_count: too many slots"); return CGS_ERR_ARG; }
    if (!feat_src || !anchor_vis || !cam3 || !mask || !W1 || !b1 || !W2 || !b2 || !neural_opacity || !bits || !base16) {
        cgs_set_error("anchor_gen_count: NULL");
        return CGS_ERR_ARG;
    }
    constexpr int RT = 2, WAVES = 8;
    const int64_t tiles = (n + 16 * RT - 1) / (16 * RT), nblk = (n + 15) / 16;
    {
        CgsProfScope prof(CGS_PROF_MLP_FWD, stream);
        hipLaunchKernelGGL((ag_opacity_kernel<RT, WAVES>), dim3(ag_grid(tiles, WAVES)), dim3(WAVES * 64), 0, stream,
                           AgRows{feat_src, feat_row, anchor_vis, cam3}, W1, b1, W2, b2, mask, y_op, neural_opacity,
                           mask_out, bits, base16, n);
        CGS_CHECK_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(ag_scan_kernel, dim3(1), dim3(1024), 0, stream, base16, nblk);
    CGS_CHECK_HIP(hipGetLastError());
    static thread_local uint32_t *pinned = nullptr;
    if (!pinned) CGS_CHECK_HIP(hipHostMalloc((void **)&pinned, 64, hipHostMallocDefault));
    CGS_CHECK_HIP(hipMemcpyAsync(pinned, base16 + nblk, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    CGS_CHECK_HIP(hipStreamSynchronize(stream));
    *count_host = pinned[0];
    return CGS_OK;
}

// Colour / covariance heads + compacted Gaussians.  W1 / b1 / W2 / b2: [2] = (colour, covariance).  gs_src [*,6],
// off_src [*,K*3] read through geo_row (NULL = identity).  Outputs [P,.] with P = the count of cgs_anchor_gen_count;
// siginv [P,4] may be NULL (inference).
extern "C" int cgs_anchor_gen_write(const float *feat_src, const int64_t *feat_row, const float *anchor_vis,
                                    const float *cam3, const float *gs_src, const float *off_src,
                                    const int64_t *geo_row, const float *neural_opacity, const uint32_t *bits,
                                    const uint32_t *base16, const float *const *W1, const float *const *b1,
                                    const float *const *W2, const float *const *b2, float *xyz, float *color,
                                    float *opacity, float *scaling, float *rot, float *siginv, int64_t n, int K,
                                    void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0 || K != AG_K) { cgs_set_error("anchor_gen_write: n < 0 or K != 10"); return CGS_ERR_ARG; }
    if (n == 0) return CGS_OK;
    if (!feat_src || !anchor_vis || !cam3 || !gs_src || !off_src || !neural_opacity || !bits || !base16 || !W1 || !b1 ||
        !W2 || !b2) {
        cgs_set_error("anchor_gen_write: NULL");
        return CGS_ERR_ARG;
    }
    constexpr int RT = 2, WAVES = 8;
    const int64_t tiles = (n + 16 * RT - 1) / (16 * RT);
    CgsProfScope prof(CGS_PROF_EXPAND_FWD, stream);
    hipLaunchKernelGGL((ag_write_kernel<RT, WAVES>), dim3(ag_grid(tiles, WAVES)), dim3(WAVES * 64), 0, stream,
                       AgRows{feat_src, feat_row, anchor_vis, cam3}, AgGeo{gs_src, off_src, geo_row}, neural_opacity, bits,
                       base16, W1[0], b1[0], W2[0], b2[0], W1[1], b1[1], W2[1], b2[1], xyz, color, opacity, scaling, rot,
                       siginv, n);
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}

extern "C" size_t cgs_anchor_gen_bwd_scratch_bytes(int64_t n, int fused) {
    const size_t part = (size_t)ag_cus() * AG_E * sizeof(float) + 256;
    if (fused) return part;
    // X [n,54], Hcat [n,150], dZ1cat [n,150], dZ2 [n,10+30+70] + the two-pass weight-gradient partials
    return (size_t)n * (54 + 150 + 150 + 110) * sizeof(float) + 8 * 256 + cgs_wgrad_scratch_bytes_for(ag_cus());
}



// Backward of count + write.  g_* [P,.] gradients of the five per-Gaussian outputs, g_neural_opacity [n*K] or NULL.
// W1 / b1 / W2: [3] = (opacity, colour, covariance).  Gradients written: d_feat_src rows feat_row[r] (rows no visible
// anchor reads are the caller's to zero; likewise d_gs / d_off through geo_row), d_anchor [n,3], d_mask [n,K]; weight /
// bias gradients are ACCUMULATED into dW1cat [150,54], db1cat [150], dW2[i] [OUT_i,50], db2[i] [OUT_i].
// fused != 0: weight gradients inside the kernel (scratch = workgroup partial images); 0: separate launch.
extern "C" int cgs_anchor_gen_backward(const float *feat_src, const int64_t *feat_row, const float *anchor_vis,
                                       const float *cam3, const float *gs_src, const float *off_src,
                                       const int64_t *geo_row, const float *mask, const float *y_op,
                                       const uint32_t *bits, const uint32_t *base16, const float *color,
                                       const float *rot, const float *siginv, const float *g_xyz, const float *g_color,
                                       const float *g_opacity, const float *g_scaling, const float *g_rot,
                                       const float *g_neural_opacity, const float *const *W1, const float *const *b1,
                                       const float *const *W2, float *d_feat_src, float *d_anchor, float *d_gs,
                                       float *d_off, float *d_mask, float *dW1cat, float *db1cat, float *const *dW2,
                                       float *const *db2, int64_t n, int K, int fused, void *scratch,
                                       size_t scratch_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0 || K != AG_K) { cgs_set_error("anchor_gen_backward: n < 0 or K != 10"); return CGS_ERR_ARG; }
    if (n == 0) return CGS_OK;
    // (the [P,.] arrays are only dereferenced for survivors: NULL is legal when P = 0)
    if (!feat_src || !anchor_vis || !cam3 || !gs_src || !off_src || !mask || !y_op || !bits || !base16 || !W1 || !b1 || !W2 ||
        !d_feat_src || !d_anchor || !d_gs || !d_off || !d_mask || !dW1cat || !db1cat || !dW2 || !db2 || !scratch) {
        cgs_set_error("anchor_gen_backward: NULL");
        return CGS_ERR_ARG;
    }
    if (scratch_bytes < cgs_anchor_gen_bwd_scratch_bytes(n, fused)) { cgs_set_error("anchor_gen_backward: scratch too small"); return CGS_ERR_WORKSPACE; }
    constexpr int WAVES = 8;
    AgBwdArgs a{};
    a.R = AgRows{feat_src, feat_row, anchor_vis, cam3};
    a.G = AgGeo{gs_src, off_src, geo_row};
    a.mask = mask; a.y_op = y_op; a.bits = bits; a.base16 = base16;
    a.color = color; a.rot = rot; a.siginv = siginv;
    a.g_xyz = g_xyz; a.g_color = g_color; a.g_opacity = g_opacity; a.g_scaling = g_scaling; a.g_rot = g_rot;
    a.g_nop = g_neural_opacity;
    for (int i = 0; i < 3; ++i) { a.W1[i] = W1[i]; a.b1[i] = b1[i]; a.W2[i] = W2[i]; }
    a.d_feat_src = d_feat_src; a.d_anchor = d_anchor; a.d_gs = d_gs; a.d_off = d_off; a.d_mask = d_mask;
    a.n = n;
    const int64_t tiles = (n + 15) / 16;
    const int grid = ag_grid(tiles, WAVES);
    const CgsWgProduct shapes[4] = {{nullptr, 0, 150, nullptr, 0, 54, dW1cat, db1cat},
                                    {nullptr, 0, 10, nullptr, 0, 50, dW2[0], db2[0]},
                                    {nullptr, 0, 30, nullptr, 0, 50, dW2[1], db2[1]},
                                    {nullptr, 0, 70, nullptr, 0, 50, dW2[2], db2[2]}};
    if (fused) {
        a.partial = (float *)scratch;
        {
            CgsProfScope prof(CGS_PROF_MLP_BWD, stream);
            hipLaunchKernelGGL((ag_bwd_kernel<WAVES, true>), dim3(grid), dim3(WAVES * 64), 0, stream, a);
            CGS_CHECK_HIP(hipGetLastError());
        }
        CgsProfScope prof(CGS_PROF_MLP_WGRAD, stream);
        return cgs_launch_wgrad_reduce(a.partial, grid, shapes, 4, stream);
    }
    CgsCarver cv(scratch, scratch_bytes);
    a.X_out = cv.take<float>((size_t)n * 54);
    a.Hcat = cv.take<float>((size_t)n * 150);
    a.dZ1cat = cv.take<float>((size_t)n * 150);
    a.dZ2_op = cv.take<float>((size_t)n * 10);
    a.dZ2_col = cv.take<float>((size_t)n * 30);
    a.dZ2_cov = cv.take<float>((size_t)n * 70);
    const size_t wbytes = cgs_wgrad_scratch_bytes_for(ag_cus());
    void *wscratch = cv.take<char>(wbytes);
    if (!cv.ok) { cgs_set_error("anchor_gen_backward: scratch too small"); return CGS_ERR_WORKSPACE; }
    {
        CgsProfScope prof(CGS_PROF_MLP_BWD, stream);
        hipLaunchKernelGGL((ag_bwd_kernel<WAVES, false>), dim3(grid), dim3(WAVES * 64), 0, stream, a);
        CGS_CHECK_HIP(hipGetLastError());
    }
    CgsProfScope prof(CGS_PROF_MLP_WGRAD, stream);
    const CgsWgProduct prods[4] = {{a.dZ1cat, 150, 150, a.X_out, 54, 54, dW1cat, db1cat},
                                   {a.dZ2_op, 10, 10, a.Hcat, 150, 50, dW2[0], db2[0]},
                                   {a.dZ2_col, 30, 30, a.Hcat + 50, 150, 50, dW2[1], db2[1]},
                                   {a.dZ2_cov, 70, 70, a.Hcat + 100, 150, 50, dW2[2], db2[2]}};
    return cgs_launch_wgrad_multi(prods, 4, n, ag_cus(), wscratch, wbytes, stream);
}
