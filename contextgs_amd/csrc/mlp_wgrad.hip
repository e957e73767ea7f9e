// Weight / bias gradients of the fused MLPs:  dW[a][b] += sum_rows P[row][a] Q[row][b],
// db[a] += sum_rows P[row][a]   (P = dZ [n, DA], Q = layer input [n, DB], both row-major in HBM).
//
// fp32 MFMA 16x16x4 with the ROW index as the contraction dimension: one instruction consumes 4
// rows; both operand fragments are 4 rows x 64 B straight from memory (lane l reads
// row0 + (l>>4), column 16*tile + (l&15)), no LDS.
//
// Register blocking: the 8 waves of a workgroup form a WA x WB grid over the NA x NB output
// tiles; a wave owns UA x UB tiles and loads UA + UB fragments per 4-row step for UA*UB MFMAs
// (the first version gave each wave scattered tiles: 2 loads per MFMA, L1-bandwidth bound — see
// profiles/r01_rocprof_bench_1m_v1_mfma_mlp.txt).  Rows are split over workgroups; each ends with
// one fp32 atomic per owned output element.
#include "cgs_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int UA, int UB, int UNR>
__global__ void __launch_bounds__(512)
    wgrad2_kernel(const float *__restrict__ P, int64_t ldp, int DA, const float *__restrict__ Q, int64_t ldq, int DB,
                  float *__restrict__ dW, float *__restrict__ db, int64_t n, int64_t rows_per_block, int WA, int WB) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int wa = wave / WB, wb = wave % WB;
    const int NA = (DA + 15) / 16, NB = (DB + 15) / 16;
    const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r_end = min(n, r_begin + rows_per_block);
    int acol[UA], bcol[UB];
    bool alive[UA], blive[UB];
#pragma unroll
    for (int j = 0; j < UA; ++j) {
        const int u = wa + WA * j;
        alive[j] = u < NA && (16 * u + c) < DA;
        acol[j] = 16 * u + c;
    }
#pragma unroll
    for (int k = 0; k < UB; ++k) {
        const int t = wb + WB * k;
        blive[k] = t < NB && (16 * t + c) < DB;
        bcol[k] = 16 * t + c;
    }
    if (wa >= NA || wb >= NB) return;
    const bool do_bias = db != nullptr && wb == 0;
    f32x4 acc[UA][UB], accb[UA];
#pragma unroll
    for (int j = 0; j < UA; ++j) {
        accb[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < UB; ++k) acc[j][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (int64_t row0 = r_begin; row0 < r_end; row0 += 4 * UNR) {
        float a[UNR][UA], b[UNR][UB], one[UNR];
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
            const int64_t row = row0 + 4 * q + g;
            const bool valid = row < r_end;
            one[q] = valid ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < UA; ++j) a[q][j] = (valid && alive[j]) ? P[row * ldp + acol[j]] : 0.f;
#pragma unroll
            for (int k = 0; k < UB; ++k) b[q][k] = (valid && blive[k]) ? Q[row * ldq + bcol[k]] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < UNR; ++q)
#pragma unroll
            for (int j = 0; j < UA; ++j) {
#pragma unroll
                for (int k = 0; k < UB; ++k)
                    acc[j][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][j], b[q][k], acc[j][k], 0, 0, 0);
                if (do_bias) accb[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][j], one[q], accb[j], 0, 0, 0);
            }
    }
    // D layout: reg r of lane l <-> (a = 16u + 4g + r, b = 16t + c)
#pragma unroll
    for (int j = 0; j < UA; ++j) {
        const int u = wa + WA * j;
        if (u >= NA) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int arow = 16 * u + 4 * g + r;
            if (arow >= DA) continue;
#pragma unroll
            for (int k = 0; k < UB; ++k) {
                const int t = wb + WB * k;
                if (t < NB && bcol[k] < DB) atomicAdd(&dW[(int64_t)arow * DB + bcol[k]], acc[j][k][r]);
            }
            if (do_bias && c == 0) atomicAdd(&db[arow], accb[j][r]);
        }
    }
}

int cgs_launch_wgrad2(const float *P, int64_t ldp, int DA, const float *Q, int64_t ldq, int DB, float *dW, float *db,
                      int64_t n, int num_cus, hipStream_t s) {
    if (n <= 0) return CGS_OK;
    const int NA = (DA + 15) / 16, NB = (DB + 15) / 16;
    // wave grid WA x WB = 8: as square as the tile grid allows
    int WA = 0, WB = 0, UA = 0, UB = 0;
    double best = 1e30;
    for (int wa = 1; wa <= 8; wa *= 2) {
        const int wb = 8 / wa;
        const int ua = (NA + wa - 1) / wa, ub = (NB + wb - 1) / wb;
        if (ua > 4 || ub > 4) continue;
        const int active = (wa < NA ? wa : NA) * (wb < NB ? wb : NB);
        // loads per MFMA, penalised by idle waves
        const double cost = (double)(ua + ub) / (ua * ub) * 8.0 / active;
        if (cost < best) { best = cost; WA = wa; WB = wb; UA = ua; UB = ub; }
    }
    if (!WA) { cgs_set_error("wgrad: tile grid %dx%d too large", NA, NB); return CGS_ERR_ARG; }
    int64_t blocks = (n + 1023) / 1024;
    const int64_t cap = 2 * (int64_t)num_cus;
    if (blocks > cap) blocks = cap;
    int64_t rpb = (n + blocks - 1) / blocks;
    rpb = (rpb + 15) / 16 * 16;
    blocks = (n + rpb - 1) / rpb;
#define WG(UA_, UB_, UNR_)                                                                                          \
    hipLaunchKernelGGL((wgrad2_kernel<UA_, UB_, UNR_>), dim3((unsigned)blocks), dim3(512), 0, s, P, ldp, DA, Q, ldq, DB, dW, \
                       db, n, rpb, WA, WB)
#define ROW(UA_)                                                     \
    switch (UB) {                                                    \
        case 1: WG(UA_, 1, 4); break;                                \
        case 2: WG(UA_, 2, 4); break;                                \
        case 3: WG(UA_, 3, 2); break;                                \
        case 4: WG(UA_, 4, 2); break;                                \
        default: cgs_set_error("wgrad: tile grid %dx%d too large", NA, NB); return CGS_ERR_ARG; \
    }
    switch (UA) {
        case 1: ROW(1); break;
        case 2: ROW(2); break;
        case 3: ROW(3); break;
        case 4: ROW(4); break;
        default: cgs_set_error("wgrad: tile grid %dx%d too large", NA, NB); return CGS_ERR_ARG;
    }
#undef ROW
#undef WG
    CGS_CHECK_HIP(hipGetLastError());
    return CGS_OK;
}
