/*
 * raster_ref.c — CPU ORACLE for the tile rasterizer (forward + backward).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under contextgs_amd/ may import, link or
 * execute this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference's rasterizer (the Scaffold-GS fork of
 * diff-gaussian-rasterization, a CUDA extension) is NOT in /root/reference
 * (SURVEY.md §0 fact 1): there is no source, wheel, golden vector or test to
 * pin this restatement against.  It restates the published 3D Gaussian
 * Splatting rasterization algorithm (Kerbl et al. 2023) exactly as recorded in
 * SURVEY.md Appendix A, under the call-site contract of the reference:
 *   - settings / call:  gaussian_renderer/__init__.py:179-205
 *   - visible_filter:   gaussian_renderer/__init__.py:250-285
 *   - row-vector matrices (transposes): scene/cameras.py:54-56
 *   - means2D gradient convention consumed by densification:
 *     scene/gaussian_model.py:710, arguments/__init__.py:153
 * What pins it instead (tests/test_oracle_raster.py): analytic single-Gaussian
 * images, sum(weights)+final_T == 1, radii==0 <=> culled, and an fp64
 * finite-difference check of every gradient this file produces.
 *
 * Deliberately plain: one loop per stage, no tile/quadrant culling, no
 * pre-scaled conics — none of the tricks the HIP path uses — so that it is an
 * independent statement of the semantics.
 *
 * Build: see oracle/Makefile (REAL=float -> libraster_ref_f32.so, the parity
 * checker and CPU baseline; REAL=double -> libraster_ref_f64.so, gradcheck).
 */
#include <tgmath.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL float
#endif
typedef REAL real;

#define TILE 16

typedef struct {
    int32_t H, W;
    real tanfovx, tanfovy;
    real scale_modifier;
    real view[16];   /* row-vector convention: p_view = [x y z 1] @ view */
    real proj[16];
    real bg[3];
} ref_cfg;

typedef struct {
    int ok;
    real px, py, depth;
    real cov[3];     /* dilated 2-D covariance (a, b, c) */
    real conic[3];
    real radius;
    int x0, y0, x1, y1;   /* tile rect, max exclusive */
    real cov3d[6];
    real t[3];       /* view-space point (unclamped) */
} ref_geom;

static void quat_to_R(const real *q, real R[3][3]) {
    real r = q[0], x = q[1], y = q[2], z = q[3];
    R[0][0] = 1 - 2 * (y * y + z * z); R[0][1] = 2 * (x * y - r * z);     R[0][2] = 2 * (x * z + r * y);
    R[1][0] = 2 * (x * y + r * z);     R[1][1] = 1 - 2 * (x * x + z * z); R[1][2] = 2 * (y * z - r * x);
    R[2][0] = 2 * (x * z - r * y);     R[2][1] = 2 * (y * z + r * x);     R[2][2] = 1 - 2 * (x * x + y * y);
}

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* R1/R2 of SURVEY §2.1: per-Gaussian preprocess. */
static void preprocess_one(const ref_cfg *c, const real *p, const real *s, const real *q, ref_geom *g) {
    const real *V = c->view, *Pm = c->proj;
    g->ok = 0;
    g->radius = 0;
    real t[3];
    for (int j = 0; j < 3; ++j) t[j] = p[0] * V[0 + j] + p[1] * V[4 + j] + p[2] * V[8 + j] + V[12 + j];
    g->t[0] = t[0]; g->t[1] = t[1]; g->t[2] = t[2];
    if (t[2] <= (real)0.2) return;
    real h[4];
    for (int j = 0; j < 4; ++j) h[j] = p[0] * Pm[0 + j] + p[1] * Pm[4 + j] + p[2] * Pm[8 + j] + Pm[12 + j];
    real pw = (real)1 / (h[3] + (real)0.0000001);
    real ndc[2] = {h[0] * pw, h[1] * pw};

    /* cov3D = R S S^T R^T */
    real R[3][3], M[3][3];
    quat_to_R(q, R);
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) M[i][k] = R[i][k] * (s[k] * c->scale_modifier);
    real S3[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            real a = 0;
            for (int k = 0; k < 3; ++k) a += M[i][k] * M[j][k];
            S3[i][j] = a;
        }
    g->cov3d[0] = S3[0][0]; g->cov3d[1] = S3[0][1]; g->cov3d[2] = S3[0][2];
    g->cov3d[3] = S3[1][1]; g->cov3d[4] = S3[1][2]; g->cov3d[5] = S3[2][2];

    /* EWA cov2D */
    real limx = (real)1.3 * c->tanfovx, limy = (real)1.3 * c->tanfovy;
    real txtz = t[0] / t[2], tytz = t[1] / t[2];
    real tx = fmin(limx, fmax(-limx, txtz)) * t[2];
    real ty = fmin(limy, fmax(-limy, tytz)) * t[2];
    real fx = c->W / ((real)2 * c->tanfovx), fy = c->H / ((real)2 * c->tanfovy);
    real J[2][3] = {{fx / t[2], 0, -(fx * tx) / (t[2] * t[2])}, {0, fy / t[2], -(fy * ty) / (t[2] * t[2])}};
    real Wv[3][3]; /* world->view rotation, column-vector convention */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Wv[i][j] = V[4 * j + i];
    real A[2][3];
    for (int r = 0; r < 2; ++r)
        for (int j = 0; j < 3; ++j) {
            real a = 0;
            for (int k = 0; k < 3; ++k) a += J[r][k] * Wv[k][j];
            A[r][j] = a;
        }
    real AS[2][3];
    for (int r = 0; r < 2; ++r)
        for (int j = 0; j < 3; ++j) {
            real a = 0;
            for (int k = 0; k < 3; ++k) a += A[r][k] * S3[k][j];
            AS[r][j] = a;
        }
    real ca = 0, cb = 0, cc = 0;
    for (int k = 0; k < 3; ++k) { ca += AS[0][k] * A[0][k]; cb += AS[0][k] * A[1][k]; cc += AS[1][k] * A[1][k]; }
    ca += (real)0.3;
    cc += (real)0.3;
    real det = ca * cc - cb * cb;
    if (det == 0) return;
    real inv = (real)1 / det;
    g->cov[0] = ca; g->cov[1] = cb; g->cov[2] = cc;
    g->conic[0] = cc * inv; g->conic[1] = -cb * inv; g->conic[2] = ca * inv;
    real mid = (real)0.5 * (ca + cc);
    real disc = sqrt(fmax((real)0.1, mid * mid - det));
    real lam = fmax(mid + disc, mid - disc);
    real radius = ceil((real)3 * sqrt(lam));
    g->px = ((ndc[0] + 1) * c->W - 1) * (real)0.5;
    g->py = ((ndc[1] + 1) * c->H - 1) * (real)0.5;
    int gx = (c->W + TILE - 1) / TILE, gy = (c->H + TILE - 1) / TILE;
    g->x0 = imin(gx, imax(0, (int)((g->px - radius) / TILE)));
    g->y0 = imin(gy, imax(0, (int)((g->py - radius) / TILE)));
    g->x1 = imin(gx, imax(0, (int)((g->px + radius + TILE - 1) / TILE)));
    g->y1 = imin(gy, imax(0, (int)((g->py + radius + TILE - 1) / TILE)));
    if ((g->x1 - g->x0) * (g->y1 - g->y0) == 0) return;
    g->depth = t[2];
    g->radius = radius;
    g->ok = 1;
}

int ref_filter(const ref_cfg *c, int64_t N, const real *means3D, const real *scales, const real *rots,
               int32_t *radii) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        ref_geom g;
        preprocess_one(c, means3D + 3 * i, scales + 3 * i, rots + 4 * i, &g);
        radii[i] = g.ok ? (int32_t)g.radius : 0;
    }
    return 0;
}

typedef struct { uint32_t tile; uint32_t id; real depth; } pair_t;

static int pair_cmp(const void *a_, const void *b_) {
    const pair_t *a = (const pair_t *)a_, *b = (const pair_t *)b_;
    if (a->tile != b->tile) return a->tile < b->tile ? -1 : 1;
    if (a->depth != b->depth) return a->depth < b->depth ? -1 : 1;
    if (a->id != b->id) return a->id < b->id ? -1 : 1;
    return 0;
}

/*
 * Forward (+ optional backward when dL_dout != NULL).
 * Outputs: out_color [3,H,W], radii [P]; aux (may be NULL): final_T [H*W],
 * weight_sum [H*W] (sum of alpha*T per pixel, for the invariant test).
 * Gradients (all may be NULL when dL_dout is NULL): dL_dmeans3D [P,3],
 * dL_dmeans2D [P,3] (NDC-scaled convention), dL_dcolors [P,3], dL_dopac [P],
 * dL_dscales [P,3], dL_drots [P,4].  stats_out[0] = number of (tile,Gaussian)
 * pairs, stats_out[1] = pairs actually visited by the blend loops (max over the
 * tile's pixels), both may be NULL.
 */
int ref_render(const ref_cfg *c, int64_t P, const real *means3D, const real *colors, const real *opac,
               const real *scales, const real *rots, real *out_color, int32_t *radii, real *final_T_out,
               real *weight_sum_out, const real *dL_dout, real *dL_dmeans3D, real *dL_dmeans2D,
               real *dL_dcolors, real *dL_dopac, real *dL_dscales, real *dL_drots, int64_t *stats_out) {
    const int W = c->W, H = c->H;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const int ntiles = gx * gy;
    ref_geom *G = (ref_geom *)malloc(sizeof(ref_geom) * (size_t)(P > 0 ? P : 1));
    if (!G) return 1;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < P; ++i) {
        preprocess_one(c, means3D + 3 * i, scales + 3 * i, rots + 4 * i, &G[i]);
        radii[i] = G[i].ok ? (int32_t)G[i].radius : 0;
    }
    /* R3-R6: pairs, sort by (tile, depth, id), ranges */
    int64_t R = 0;
    for (int64_t i = 0; i < P; ++i)
        if (G[i].ok) R += (int64_t)(G[i].x1 - G[i].x0) * (G[i].y1 - G[i].y0);
    pair_t *pairs = (pair_t *)malloc(sizeof(pair_t) * (size_t)(R > 0 ? R : 1));
    int64_t *rstart = (int64_t *)calloc((size_t)ntiles + 1, sizeof(int64_t));
    if (!pairs || !rstart) return 1;
    int64_t k = 0;
    for (int64_t i = 0; i < P; ++i)
        if (G[i].ok)
            for (int y = G[i].y0; y < G[i].y1; ++y)
                for (int x = G[i].x0; x < G[i].x1; ++x) {
                    pairs[k].tile = (uint32_t)(y * gx + x);
                    pairs[k].id = (uint32_t)i;
                    pairs[k].depth = G[i].depth;
                    ++k;
                }
    qsort(pairs, (size_t)R, sizeof(pair_t), pair_cmp);
    for (int64_t i = 0; i < R; ++i) rstart[pairs[i].tile + 1]++;
    for (int t = 0; t < ntiles; ++t) rstart[t + 1] += rstart[t];

    const size_t HW = (size_t)H * W;
    real *fT = (real *)malloc(sizeof(real) * HW);
    uint32_t *ncontrib = (uint32_t *)malloc(sizeof(uint32_t) * HW);
    if (!fT || !ncontrib) return 1;
    int64_t visited = 0;

    /* R7: blend, front to back */
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : visited)
    for (int t = 0; t < ntiles; ++t) {
        const int tx = t % gx, ty = t / gx;
        uint32_t tile_max = 0;
        for (int ly = 0; ly < TILE; ++ly)
            for (int lx = 0; lx < TILE; ++lx) {
                const int px = tx * TILE + lx, py = ty * TILE + ly;
                if (px >= W || py >= H) continue;
                real T = 1, C[3] = {0, 0, 0}, wsum = 0;
                uint32_t contributor = 0, last = 0;
                for (int64_t e = rstart[t]; e < rstart[t + 1]; ++e) {
                    ++contributor;
                    const ref_geom *g = &G[pairs[e].id];
                    const real dx = g->px - (real)px, dy = g->py - (real)py;
                    const real power = (real)-0.5 * (g->conic[0] * dx * dx + g->conic[2] * dy * dy) -
                                       g->conic[1] * dx * dy;
                    if (power > 0) continue;
                    const real alpha = fmin((real)0.99, opac[pairs[e].id] * exp(power));
                    if (alpha < (real)1 / (real)255) continue;
                    const real test_T = T * (1 - alpha);
                    if (test_T < (real)0.0001) break;   /* done: this Gaussian is not added */
                    for (int ch = 0; ch < 3; ++ch) C[ch] += colors[3 * (size_t)pairs[e].id + ch] * alpha * T;
                    wsum += alpha * T;
                    T = test_T;
                    last = contributor;
                }
                const size_t pix = (size_t)py * W + px;
                fT[pix] = T;
                ncontrib[pix] = last;
                if (last > tile_max) tile_max = last;
                for (int ch = 0; ch < 3; ++ch) out_color[ch * HW + pix] = C[ch] + T * c->bg[ch];
                if (final_T_out) final_T_out[pix] = T;
                if (weight_sum_out) weight_sum_out[pix] = wsum;
            }
        visited += tile_max;
    }
    if (stats_out) { stats_out[0] = R; stats_out[1] = visited; }

    if (dL_dout) {
        real *d_mean_px = (real *)calloc((size_t)(P > 0 ? P : 1) * 2, sizeof(real));
        real *d_conic = (real *)calloc((size_t)(P > 0 ? P : 1) * 3, sizeof(real));
        if (!d_mean_px || !d_conic) return 1;
        memset(dL_dmeans3D, 0, sizeof(real) * 3 * (size_t)P);
        memset(dL_dmeans2D, 0, sizeof(real) * 3 * (size_t)P);
        memset(dL_dcolors, 0, sizeof(real) * 3 * (size_t)P);
        memset(dL_dopac, 0, sizeof(real) * (size_t)P);
        memset(dL_dscales, 0, sizeof(real) * 3 * (size_t)P);
        memset(dL_drots, 0, sizeof(real) * 4 * (size_t)P);
        /* R8 (blend half): back to front, reconstructing T by division */
#pragma omp parallel for schedule(dynamic, 4)
        for (int t = 0; t < ntiles; ++t) {
            const int tx = t % gx, ty = t / gx;
            for (int ly = 0; ly < TILE; ++ly)
                for (int lx = 0; lx < TILE; ++lx) {
                    const int px = tx * TILE + lx, py = ty * TILE + ly;
                    if (px >= W || py >= H) continue;
                    const size_t pix = (size_t)py * W + px;
                    const real T_final = fT[pix];
                    real T = T_final;
                    real dpix[3], accum[3] = {0, 0, 0}, last_color[3] = {0, 0, 0}, last_alpha = 0;
                    real bg_dot = 0;
                    for (int ch = 0; ch < 3; ++ch) {
                        dpix[ch] = dL_dout[ch * HW + pix];
                        bg_dot += c->bg[ch] * dpix[ch];
                    }
                    for (int64_t e = rstart[t] + (int64_t)ncontrib[pix] - 1; e >= rstart[t]; --e) {
                        const uint32_t id = pairs[e].id;
                        const ref_geom *g = &G[id];
                        const real dx = g->px - (real)px, dy = g->py - (real)py;
                        const real power = (real)-0.5 * (g->conic[0] * dx * dx + g->conic[2] * dy * dy) -
                                           g->conic[1] * dx * dy;
                        if (power > 0) continue;
                        const real Gv = exp(power);
                        const real alpha = fmin((real)0.99, opac[id] * Gv);
                        if (alpha < (real)1 / (real)255) continue;
                        T = T / (1 - alpha);
                        const real w = alpha * T;
                        real dL_dalpha = 0;
                        for (int ch = 0; ch < 3; ++ch) {
                            const real col = colors[3 * (size_t)id + ch];
                            accum[ch] = last_alpha * last_color[ch] + (1 - last_alpha) * accum[ch];
                            last_color[ch] = col;
                            dL_dalpha += (col - accum[ch]) * dpix[ch];
                            const real v = w * dpix[ch];
#pragma omp atomic
                            dL_dcolors[3 * (size_t)id + ch] += v;
                        }
                        dL_dalpha *= T;
                        last_alpha = alpha;
                        dL_dalpha += (-T_final / (1 - alpha)) * bg_dot;
                        /* the 0.99 clamp is transparent to the gradient (public algorithm) */
                        const real dL_dG = opac[id] * dL_dalpha;
                        const real gdx = Gv * dx, gdy = Gv * dy;
                        const real dG_ddelx = -gdx * g->conic[0] - gdy * g->conic[1];
                        const real dG_ddely = -gdy * g->conic[2] - gdx * g->conic[1];
                        const real v0 = dL_dG * dG_ddelx, v1 = dL_dG * dG_ddely;
                        const real v2 = (real)-0.5 * gdx * dx * dL_dG;
                        const real v3 = -gdx * dy * dL_dG;  /* full derivative w.r.t. the off-diagonal */
                        const real v4 = (real)-0.5 * gdy * dy * dL_dG;
                        const real v5 = Gv * dL_dalpha;
#pragma omp atomic
                        d_mean_px[2 * (size_t)id] += v0;
#pragma omp atomic
                        d_mean_px[2 * (size_t)id + 1] += v1;
#pragma omp atomic
                        d_conic[3 * (size_t)id] += v2;
#pragma omp atomic
                        d_conic[3 * (size_t)id + 1] += v3;
#pragma omp atomic
                        d_conic[3 * (size_t)id + 2] += v4;
#pragma omp atomic
                        dL_dopac[id] += v5;
                    }
                }
        }
        /* R8 (per-Gaussian half) */
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < P; ++i) {
            if (!G[i].ok) continue;
            const real *V = c->view, *Pm = c->proj;
            const real *p = means3D + 3 * i, *q = rots + 4 * i;
            const ref_geom *g = &G[i];
            /* (1) conic -> cov2D entries (a, b, c), b being the repeated off-diagonal */
            const real a = g->cov[0], b = g->cov[1], cc = g->cov[2];
            const real det = a * cc - b * b;
            const real dca = d_conic[3 * i], dcb = d_conic[3 * i + 1], dcc = d_conic[3 * i + 2];
            real da = 0, db = 0, dc = 0;
            if (det != 0) {
                const real d2 = 1 / (det * det);
                /* conic = (c, -b, a)/det */
                da = d2 * (-cc * cc * dca + b * cc * dcb + (det - a * cc) * dcc);
                dc = d2 * (-a * a * dcc + a * b * dcb + (det - a * cc) * dca);
                db = d2 * (2 * b * cc * dca - (det + 2 * b * b) * dcb + 2 * a * b * dcc);
            }
            /* (2) cov2D = A S3 A^T.  T-matrix form: Tm = A (2x3). */
            real limx = (real)1.3 * c->tanfovx, limy = (real)1.3 * c->tanfovy;
            const real *t = g->t;
            real txtz = t[0] / t[2], tytz = t[1] / t[2];
            real tx = fmin(limx, fmax(-limx, txtz)) * t[2];
            real ty = fmin(limy, fmax(-limy, tytz)) * t[2];
            const real x_mul = (txtz < -limx || txtz > limx) ? 0 : 1;
            const real y_mul = (tytz < -limy || tytz > limy) ? 0 : 1;
            real fx = c->W / ((real)2 * c->tanfovx), fy = c->H / ((real)2 * c->tanfovy);
            real J[2][3] = {{fx / t[2], 0, -(fx * tx) / (t[2] * t[2])}, {0, fy / t[2], -(fy * ty) / (t[2] * t[2])}};
            real Wv[3][3];
            for (int r = 0; r < 3; ++r)
                for (int j = 0; j < 3; ++j) Wv[r][j] = V[4 * j + r];
            real A[2][3];
            for (int r = 0; r < 2; ++r)
                for (int j = 0; j < 3; ++j) A[r][j] = J[r][0] * Wv[0][j] + J[r][1] * Wv[1][j] + J[r][2] * Wv[2][j];
            real S3[3][3] = {{g->cov3d[0], g->cov3d[1], g->cov3d[2]},
                             {g->cov3d[1], g->cov3d[3], g->cov3d[4]},
                             {g->cov3d[2], g->cov3d[4], g->cov3d[5]}};
            /* dL/dS3[j][k] (matrix form, symmetric): sum over cov2D entries */
            real dS3[3][3];
            for (int j = 0; j < 3; ++j)
                for (int k2 = 0; k2 < 3; ++k2)
                    dS3[j][k2] = da * A[0][j] * A[0][k2] + dc * A[1][j] * A[1][k2] +
                                 (real)0.5 * db * (A[0][j] * A[1][k2] + A[1][j] * A[0][k2]);
            /* dL/dA: a = A0 S A0, b = A0 S A1, c = A1 S A1 */
            real SA0[3], SA1[3];
            for (int j = 0; j < 3; ++j) {
                SA0[j] = S3[j][0] * A[0][0] + S3[j][1] * A[0][1] + S3[j][2] * A[0][2];
                SA1[j] = S3[j][0] * A[1][0] + S3[j][1] * A[1][1] + S3[j][2] * A[1][2];
            }
            real dA[2][3];
            for (int j = 0; j < 3; ++j) {
                dA[0][j] = 2 * da * SA0[j] + db * SA1[j];
                dA[1][j] = 2 * dc * SA1[j] + db * SA0[j];
            }
            real dJ[2][3];
            for (int r = 0; r < 2; ++r)
                for (int k2 = 0; k2 < 3; ++k2)
                    dJ[r][k2] = dA[r][0] * Wv[k2][0] + dA[r][1] * Wv[k2][1] + dA[r][2] * Wv[k2][2];
            const real tz = 1 / t[2], tz2 = tz * tz, tz3 = tz2 * tz;
            real dt[3];
            dt[0] = x_mul * (-fx * tz2) * dJ[0][2];
            dt[1] = y_mul * (-fy * tz2) * dJ[1][2];
            dt[2] = -fx * tz2 * dJ[0][0] - fy * tz2 * dJ[1][1] + 2 * fx * tx * tz3 * dJ[0][2] +
                    2 * fy * ty * tz3 * dJ[1][2];
            real dp[3];
            for (int j = 0; j < 3; ++j) dp[j] = Wv[0][j] * dt[0] + Wv[1][j] * dt[1] + Wv[2][j] * dt[2];
            /* (3) projection path */
            real h[4];
            for (int j = 0; j < 4; ++j) h[j] = p[0] * Pm[0 + j] + p[1] * Pm[4 + j] + p[2] * Pm[8 + j] + Pm[12 + j];
            const real mw = 1 / (h[3] + (real)0.0000001);
            const real gnx = d_mean_px[2 * i] * (real)0.5 * W, gny = d_mean_px[2 * i + 1] * (real)0.5 * H;
            for (int j = 0; j < 3; ++j) {
                const real m1 = (Pm[4 * j + 0] * mw - Pm[4 * j + 3] * h[0] * mw * mw);
                const real m2 = (Pm[4 * j + 1] * mw - Pm[4 * j + 3] * h[1] * mw * mw);
                dp[j] += m1 * gnx + m2 * gny;
            }
            for (int j = 0; j < 3; ++j) dL_dmeans3D[3 * i + j] = dp[j];
            dL_dmeans2D[3 * i] = gnx;
            dL_dmeans2D[3 * i + 1] = gny;
            dL_dmeans2D[3 * i + 2] = 0;
            /* (4) S3 = M M^T, M = R diag(s) */
            real Rm[3][3], M[3][3], dM[3][3];
            quat_to_R(q, Rm);
            real sm[3];
            for (int k2 = 0; k2 < 3; ++k2) sm[k2] = scales[3 * i + k2] * c->scale_modifier;
            for (int r = 0; r < 3; ++r)
                for (int k2 = 0; k2 < 3; ++k2) M[r][k2] = Rm[r][k2] * sm[k2];
            for (int r = 0; r < 3; ++r)
                for (int k2 = 0; k2 < 3; ++k2) {
                    real acc = 0;
                    for (int j = 0; j < 3; ++j) acc += (dS3[r][j] + dS3[j][r]) * M[j][k2];
                    dM[r][k2] = acc;
                }
            real dR[3][3];
            for (int k2 = 0; k2 < 3; ++k2) {
                real acc = 0;
                for (int r = 0; r < 3; ++r) { acc += dM[r][k2] * Rm[r][k2]; dR[r][k2] = dM[r][k2] * sm[k2]; }
                dL_dscales[3 * i + k2] = acc * c->scale_modifier;
            }
            const real r = q[0], x = q[1], y = q[2], z = q[3];
            dL_drots[4 * i + 0] = 2 * (z * (dR[1][0] - dR[0][1]) + y * (dR[0][2] - dR[2][0]) + x * (dR[2][1] - dR[1][2]));
            dL_drots[4 * i + 1] = 2 * (y * (dR[0][1] + dR[1][0]) + z * (dR[0][2] + dR[2][0]) + r * (dR[2][1] - dR[1][2])) -
                                  4 * x * (dR[1][1] + dR[2][2]);
            dL_drots[4 * i + 2] = 2 * (x * (dR[0][1] + dR[1][0]) + r * (dR[0][2] - dR[2][0]) + z * (dR[1][2] + dR[2][1])) -
                                  4 * y * (dR[0][0] + dR[2][2]);
            dL_drots[4 * i + 3] = 2 * (r * (dR[1][0] - dR[0][1]) + x * (dR[0][2] + dR[2][0]) + y * (dR[1][2] + dR[2][1])) -
                                  4 * z * (dR[0][0] + dR[1][1]);
        }
        free(d_mean_px);
        free(d_conic);
    }
    free(G);
    free(pairs);
    free(rstart);
    free(fT);
    free(ncontrib);
    return 0;
}

int ref_sizeof_real(void) { return (int)sizeof(real); }
