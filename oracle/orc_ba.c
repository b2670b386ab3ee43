/*
 * oracle/orc_ba.c -- CPU restatement of bundle_adjustment() (TEST INFRASTRUCTURE, see orc.h).
 *
 * Follows NView:142-184 (ReprojectCost) and NView:1162-1244 (problem set-up + ceres::Solve).
 * Ceres (version unpinned by the reference: ceres.lib, OpenCV_SFM.vcxproj:95,130) is not in
 * /root/reference; its documented algorithm is restated [3P]:
 *   residual   : AutoDiffCostFunction<ReprojectCost,2,4,6,3> -> 13-wide forward-mode duals here
 *                (ceres::AngleAxisRotatePoint incl. its theta^2 <= DBL_EPSILON first-order branch)
 *   loss       : HuberLoss(4): rho(s) = s (s <= 16) | 8 sqrt(s) - 16; corrector with rho'' <= 0 =>
 *                r, J scaled by sqrt(rho'); cost = 1/2 sum rho(s)
 *   blocks     : camera 0 constant (NView:1178); ONE shared free intrinsic block (NView:1181)
 *   minimizer  : TRUST_REGION / LEVENBERG_MARQUARDT, Ceres 1.13-2.0 control flow:
 *                jacobi scaling 1/(1+||col||) fixed at x0; D^2 = clamp(diag(J'J),1e-6,1e32)/radius;
 *                solve (J'J + D^2) y = J'r, step = -y; model_cost_change = -(J step).(r + J step/2);
 *                invalid step (<= 0) => radius /= 2; parameter tolerance, then function tolerance
 *                (both BEFORE the step is accepted, candidate discarded), then
 *                rho = cost_change/model_cost_change > 1e-3 => accept, radius /= max(1/3, 1-(2rho-1)^3);
 *                else radius /= nu, nu *= 2; gradient tolerance / max iterations / min radius checked at
 *                the top of each iteration.
 *   linear     : SPARSE_SCHUR = eliminate every point block, factor the reduced camera system, back-
 *                substitute.  Here: dense reduced matrix with an envelope (skyline) Cholesky -- same
 *                solution up to rounding, sparsity of the chain structure exploited like EIGEN_SPARSE does.
 * parity unpinned (no Ceres here, no golden vectors in the reference); cross-checked in tests/ against
 * finite differences and scipy.optimize.least_squares (no-loss case) + a hand-written IRLS Gauss-Newton.
 */
#include "orc.h"
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int orc_get_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_ba_default_options(orc_ba_options* o)
{
    o->max_num_iterations = 50;
    o->initial_trust_region_radius = 1e4;
    o->max_trust_region_radius = 1e16;
    o->min_trust_region_radius = 1e-32;
    o->min_relative_decrease = 1e-3;
    o->min_lm_diagonal = 1e-6;
    o->max_lm_diagonal = 1e32;
    o->function_tolerance = 1e-6;
    o->gradient_tolerance = 1e-10;
    o->parameter_tolerance = 1e-8;
    o->huber_delta = 4.0;
    o->jacobi_scaling = 1;
    o->fix_first_camera = 1;
    o->fix_intrinsics = 0;
    o->verbose = 0;
}

/* ------------------------------------------------------------------ 13-wide duals ("Jets") */
#define NJ 13
typedef struct { double v, d[NJ]; } jet;
static inline jet jc(double c) { jet r; r.v = c; memset(r.d, 0, sizeof r.d); return r; }
static inline jet jvar(double c, int k) { jet r = jc(c); r.d[k] = 1.0; return r; }
static inline jet jadd(jet a, jet b) { for (int i = 0; i < NJ; ++i) a.d[i] += b.d[i]; a.v += b.v; return a; }
static inline jet jsub(jet a, jet b) { for (int i = 0; i < NJ; ++i) a.d[i] -= b.d[i]; a.v -= b.v; return a; }
static inline jet jmul(jet a, jet b)
{
    jet r; r.v = a.v * b.v;
    for (int i = 0; i < NJ; ++i) r.d[i] = a.v * b.d[i] + a.d[i] * b.v;
    return r;
}
static inline jet jdiv(jet a, jet b)
{
    jet r; double h = 1.0 / b.v; r.v = a.v * h;
    for (int i = 0; i < NJ; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * h;
    return r;
}
static inline jet jscale(jet a, double s) { for (int i = 0; i < NJ; ++i) a.d[i] *= s; a.v *= s; return a; }
static inline jet jsqrt(jet a) { jet r; r.v = sqrt(a.v); double h = 0.5 / r.v; for (int i = 0; i < NJ; ++i) r.d[i] = a.d[i] * h; return r; }
static inline jet jsin(jet a) { jet r; r.v = sin(a.v); double c = cos(a.v); for (int i = 0; i < NJ; ++i) r.d[i] = a.d[i] * c; return r; }
static inline jet jcos(jet a) { jet r; r.v = cos(a.v); double s = -sin(a.v); for (int i = 0; i < NJ; ++i) r.d[i] = a.d[i] * s; return r; }

/* ceres::AngleAxisRotatePoint [3P], T = jet */
static void aa_rotate(const jet aa[3], const jet pt[3], jet out[3])
{
    jet th2 = jadd(jadd(jmul(aa[0], aa[0]), jmul(aa[1], aa[1])), jmul(aa[2], aa[2]));
    if (th2.v > DBL_EPSILON) {
        jet th = jsqrt(th2), c = jcos(th), s = jsin(th), inv = jdiv(jc(1.0), th);
        jet w[3] = { jmul(aa[0], inv), jmul(aa[1], inv), jmul(aa[2], inv) };
        jet wx[3] = { jsub(jmul(w[1], pt[2]), jmul(w[2], pt[1])),
                      jsub(jmul(w[2], pt[0]), jmul(w[0], pt[2])),
                      jsub(jmul(w[0], pt[1]), jmul(w[1], pt[0])) };
        jet dot = jadd(jadd(jmul(w[0], pt[0]), jmul(w[1], pt[1])), jmul(w[2], pt[2]));
        jet tmp = jmul(dot, jsub(jc(1.0), c));
        for (int k = 0; k < 3; ++k)
            out[k] = jadd(jadd(jmul(pt[k], c), jmul(wx[k], s)), jmul(w[k], tmp));
    } else {
        jet wx[3] = { jsub(jmul(aa[1], pt[2]), jmul(aa[2], pt[1])),
                      jsub(jmul(aa[2], pt[0]), jmul(aa[0], pt[2])),
                      jsub(jmul(aa[0], pt[1]), jmul(aa[1], pt[0])) };
        for (int k = 0; k < 3; ++k) out[k] = jadd(pt[k], wx[k]);
    }
}

/* NView:151-183.  J columns: [fx fy cx cy | r0 r1 r2 t0 t1 t2 | X Y Z] */
void orc_reproject(const double K4[4], const double ext6[6], const double X[3], const double uv[2],
                   double r[2], double J[26])
{
    jet in[4], ex[6], pt[3], p[3];
    for (int k = 0; k < 4; ++k) in[k] = jvar(K4[k], k);
    for (int k = 0; k < 6; ++k) ex[k] = jvar(ext6[k], 4 + k);
    for (int k = 0; k < 3; ++k) pt[k] = jvar(X[k], 10 + k);
    aa_rotate(ex, pt, p);
    p[0] = jadd(p[0], ex[3]); p[1] = jadd(p[1], ex[4]); p[2] = jadd(p[2], ex[5]);
    jet x = jdiv(p[0], p[2]), y = jdiv(p[1], p[2]);
    jet u = jadd(jmul(in[0], x), in[2]);
    jet v = jadd(jmul(in[1], y), in[3]);
    jet r0 = jsub(u, jc(uv[0])), r1 = jsub(v, jc(uv[1]));
    r[0] = r0.v; r[1] = r1.v;
    for (int k = 0; k < NJ; ++k) { J[k] = r0.d[k]; J[NJ + k] = r1.d[k]; }
}

/* residual only (value path of the same formula) */
static void reproject_value(const double K4[4], const double e[6], const double X[3], const double uv[2], double r[2])
{
    double th2 = e[0] * e[0] + e[1] * e[1] + e[2] * e[2], p[3];
    if (th2 > DBL_EPSILON) {
        double th = sqrt(th2), c = cos(th), s = sin(th), inv = 1.0 / th;
        double w[3] = { e[0] * inv, e[1] * inv, e[2] * inv };
        double wx[3] = { w[1] * X[2] - w[2] * X[1], w[2] * X[0] - w[0] * X[2], w[0] * X[1] - w[1] * X[0] };
        double tmp = (w[0] * X[0] + w[1] * X[1] + w[2] * X[2]) * (1.0 - c);
        for (int k = 0; k < 3; ++k) p[k] = X[k] * c + wx[k] * s + w[k] * tmp;
    } else {
        p[0] = X[0] + (e[1] * X[2] - e[2] * X[1]);
        p[1] = X[1] + (e[2] * X[0] - e[0] * X[2]);
        p[2] = X[2] + (e[0] * X[1] - e[1] * X[0]);
    }
    p[0] += e[3]; p[1] += e[4]; p[2] += e[5];
    double x = p[0] / p[2], y = p[1] / p[2];
    r[0] = K4[0] * x + K4[2] - uv[0];
    r[1] = K4[1] * y + K4[3] - uv[1];
}

/* HuberLoss::Evaluate [3P] */
static inline void huber(double a, double s, double rho[2])
{
    double b = a * a;
    if (a > 0 && s > b) { double r = sqrt(s); rho[0] = 2.0 * a * r - b; rho[1] = fmax(DBL_MIN, a / r); }
    else { rho[0] = s; rho[1] = 1.0; }
}

/* ------------------------------------------------------------------ problem */
typedef struct {
    int nc, np, nobs;
    const int32_t *ocam, *opt; const double* ouv;
    int *pt_start, *pt_obs;
    int fix0, fixK, ncf, n;        /* n = reduced (camera-side) order */
    orc_ba_options o;
    double* scale;                 /* [n camera-side | 3 np points], NULL = no scaling */
} prob;

static inline int cam_off(const prob* P, int c) { return (P->fix0 && c == 0) ? -1 : 6 * (c - P->fix0); }
static inline int k_off(const prob* P) { return P->fixK ? -1 : 6 * P->ncf; }

static int prob_init(prob* P, int nc, int np, const int32_t* ocam, const int32_t* opt, const double* ouv,
                     int nobs, const orc_ba_options* o)
{
    memset(P, 0, sizeof *P);
    P->nc = nc; P->np = np; P->nobs = nobs; P->ocam = ocam; P->opt = opt; P->ouv = ouv; P->o = *o;
    P->fix0 = o->fix_first_camera ? 1 : 0; P->fixK = o->fix_intrinsics ? 1 : 0;
    P->ncf = nc - P->fix0; P->n = 6 * P->ncf + (P->fixK ? 0 : 4);
    P->pt_start = (int*)calloc((size_t)np + 1, sizeof(int));
    P->pt_obs = (int*)malloc(sizeof(int) * (size_t)(nobs > 0 ? nobs : 1));
    for (int k = 0; k < nobs; ++k) {
        if (opt[k] < 0 || opt[k] >= np || ocam[k] < 0 || ocam[k] >= nc) return -1;
        P->pt_start[opt[k] + 1]++;
    }
    for (int p = 0; p < np; ++p) P->pt_start[p + 1] += P->pt_start[p];
    int* fill = (int*)malloc(sizeof(int) * (size_t)(np > 0 ? np : 1));
    memcpy(fill, P->pt_start, sizeof(int) * (size_t)np);
    for (int k = 0; k < nobs; ++k) P->pt_obs[fill[opt[k]]++] = k;
    free(fill);
    return 0;
}
static void prob_free(prob* P) { free(P->pt_start); free(P->pt_obs); free(P->scale); }

/* Evaluate cost (and, if J != NULL, corrected + column-scaled residuals r[2 nobs] and J[26 nobs]).
 * Columns of constant blocks are zeroed. */
static double evaluate(const prob* P, const double* K4, const double* ext, const double* pts, double* r, double* J)
{
    double cost = 0.0;
    const int nobs = P->nobs;
    const int nthr = orc_get_max_threads();
    double* part = (double*)calloc((size_t)nthr, sizeof(double));
#pragma omp parallel
    {
#ifdef _OPENMP
        int tid = omp_get_thread_num();
#else
        int tid = 0;
#endif
        double acc = 0.0;
#pragma omp for schedule(static)
        for (int k = 0; k < nobs; ++k) {
            int c = P->ocam[k], p = P->opt[k];
            double rr[2], JJ[26], rho[2];
            if (J) orc_reproject(K4, ext + 6 * c, pts + 3 * p, P->ouv + 2 * k, rr, JJ);
            else reproject_value(K4, ext + 6 * c, pts + 3 * p, P->ouv + 2 * k, rr);
            double s = rr[0] * rr[0] + rr[1] * rr[1];
            huber(P->o.huber_delta, s, rho);
            acc += 0.5 * rho[0];
            if (J) {
                double sq = sqrt(rho[1]);
                r[2 * k] = sq * rr[0]; r[2 * k + 1] = sq * rr[1];
                int co = cam_off(P, c), ko = k_off(P);
                for (int row = 0; row < 2; ++row) {
                    double* d = J + 26 * (size_t)k + 13 * row;
                    const double* sJ = JJ + 13 * row;
                    for (int j = 0; j < 4; ++j) d[j] = ko < 0 ? 0.0 : sq * sJ[j] * (P->scale ? P->scale[ko + j] : 1.0);
                    for (int j = 0; j < 6; ++j) d[4 + j] = co < 0 ? 0.0 : sq * sJ[4 + j] * (P->scale ? P->scale[co + j] : 1.0);
                    for (int j = 0; j < 3; ++j) d[10 + j] = sq * sJ[10 + j] * (P->scale ? P->scale[P->n + 3 * p + j] : 1.0);
                }
            }
        }
        part[tid] = acc;
    }
    for (int t = 0; t < nthr; ++t) cost += part[t];
    free(part);
    return cost;
}

/* squared column norms of the (current, possibly scaled) J: out[n + 3 np] */
static void col_sqnorm(const prob* P, const double* J, double* out)
{
    memset(out, 0, sizeof(double) * (size_t)(P->n + 3 * P->np));
    int ko = k_off(P);
    for (int k = 0; k < P->nobs; ++k) {
        int co = cam_off(P, P->ocam[k]), p = P->opt[k];
        for (int row = 0; row < 2; ++row) {
            const double* d = J + 26 * (size_t)k + 13 * row;
            if (ko >= 0) for (int j = 0; j < 4; ++j) out[ko + j] += d[j] * d[j];
            if (co >= 0) for (int j = 0; j < 6; ++j) out[co + j] += d[4 + j] * d[4 + j];
            for (int j = 0; j < 3; ++j) out[P->n + 3 * p + j] += d[10 + j] * d[10 + j];
        }
    }
}

/* symmetric 3x3 inverse through Cholesky (Ceres: InvertPSDMatrix -> llt().solve(I) [3P]) */
static int inv3_spd(const double V[9], double Vi[9])
{
    double l00 = V[0]; if (!(l00 > 0)) return -1; l00 = sqrt(l00);
    double l10 = V[3] / l00, l20 = V[6] / l00;
    double l11 = V[4] - l10 * l10; if (!(l11 > 0)) return -1; l11 = sqrt(l11);
    double l21 = (V[7] - l20 * l10) / l11;
    double l22 = V[8] - l20 * l20 - l21 * l21; if (!(l22 > 0)) return -1; l22 = sqrt(l22);
    /* inverse of L */
    double i00 = 1.0 / l00, i11 = 1.0 / l11, i22 = 1.0 / l22;
    double i10 = -l10 * i00 * i11;
    double i21 = -l21 * i11 * i22;
    double i20 = -(l20 * i00 + l21 * i10) * i22;
    /* V^-1 = L^-T L^-1 */
    Vi[0] = i00 * i00 + i10 * i10 + i20 * i20;
    Vi[1] = Vi[3] = i10 * i11 + i20 * i21;
    Vi[2] = Vi[6] = i20 * i22;
    Vi[4] = i11 * i11 + i21 * i21;
    Vi[5] = Vi[7] = i21 * i22;
    Vi[8] = i22 * i22;
    return 0;
}

/* Build the reduced system for damping D2 (per free parameter, [n | 3 np]); S full symmetric n x n.
 * Also returns per-point Vinv (9 np) and b (3 np) for the back-substitution. */
static int g_skip_cam_damping = 0;   /* orc_ba_reduced_system with radius < 0: additive-over-shards form */
static int build_reduced(const prob* P, const double* r, const double* J, const double* D2,
                         double* S, double* rhs, double* Vinv_all, double* b_all)
{
    const int n = P->n, np = P->np;
    const int nthr = orc_get_max_threads();
    double* Sbuf = (double*)calloc((size_t)nthr * ((size_t)n * n + n), sizeof(double));
    if (!Sbuf) return -1;
    int fail = 0;
    const int ko = k_off(P);
#pragma omp parallel
    {
#ifdef _OPENMP
        int tid = omp_get_thread_num();
#else
        int tid = 0;
#endif
        double* St = Sbuf + (size_t)tid * ((size_t)n * n + n);
        double* rt = St + (size_t)n * n;
        double (*W)[30] = NULL; int wcap = 0; int* idx = NULL;
#pragma omp for schedule(static)
        for (int p = 0; p < np; ++p) {
            int s0 = P->pt_start[p], L = P->pt_start[p + 1] - s0;
            double V[9] = { D2[n + 3 * p], 0, 0, 0, D2[n + 3 * p + 1], 0, 0, 0, D2[n + 3 * p + 2] };
            double b[3] = { 0, 0, 0 }, Vi[9];
            if (L > wcap) { wcap = L + 8; W = realloc(W, sizeof(*W) * (size_t)wcap); idx = realloc(idx, sizeof(int) * (size_t)wcap); }
            for (int a = 0; a < L; ++a) {
                int k = P->pt_obs[s0 + a];
                for (int row = 0; row < 2; ++row) {
                    const double* d = J + 26 * (size_t)k + 13 * row;
                    for (int i = 0; i < 3; ++i) {
                        for (int j = 0; j < 3; ++j) V[3 * i + j] += d[10 + i] * d[10 + j];
                        b[i] += d[10 + i] * r[2 * k + row];
                    }
                }
            }
            if (inv3_spd(V, Vi)) { fail = 1; continue; }
            memcpy(Vinv_all + 9 * (size_t)p, Vi, sizeof Vi);
            memcpy(b_all + 3 * (size_t)p, b, sizeof b);
            /* W_a = E_a' F_a (10 x 3), local order [cam 6 | K 4]; U and g on the fly */
            for (int a = 0; a < L; ++a) {
                int k = P->pt_obs[s0 + a];
                int co = cam_off(P, P->ocam[k]);
                idx[a] = co;
                double E[2][10], F[2][3];
                for (int row = 0; row < 2; ++row) {
                    const double* d = J + 26 * (size_t)k + 13 * row;
                    for (int j = 0; j < 6; ++j) E[row][j] = d[4 + j];
                    for (int j = 0; j < 4; ++j) E[row][6 + j] = d[j];
                    for (int j = 0; j < 3; ++j) F[row][j] = d[10 + j];
                }
                for (int i = 0; i < 10; ++i)
                    for (int j = 0; j < 3; ++j) W[a][3 * i + j] = E[0][i] * F[0][j] + E[1][i] * F[1][j];
                for (int i = 0; i < 10; ++i) {
                    int gi = i < 6 ? (co < 0 ? -1 : co + i) : (ko < 0 ? -1 : ko + i - 6);
                    if (gi < 0) continue;
                    rt[gi] += E[0][i] * r[2 * k] + E[1][i] * r[2 * k + 1];
                    for (int j = 0; j < 10; ++j) {
                        int gj = j < 6 ? (co < 0 ? -1 : co + j) : (ko < 0 ? -1 : ko + j - 6);
                        if (gj < 0) continue;
                        St[(size_t)gi * n + gj] += E[0][i] * E[0][j] + E[1][i] * E[1][j];
                    }
                }
            }
            /* Schur: S[a,b] -= W_a Vinv W_b',  rhs[a] -= W_a Vinv b */
            for (int a = 0; a < L; ++a) {
                double T[30];
                for (int i = 0; i < 10; ++i)
                    for (int j = 0; j < 3; ++j)
                        T[3 * i + j] = W[a][3 * i] * Vi[j] + W[a][3 * i + 1] * Vi[3 + j] + W[a][3 * i + 2] * Vi[6 + j];
                for (int i = 0; i < 10; ++i) {
                    int gi = i < 6 ? (idx[a] < 0 ? -1 : idx[a] + i) : (ko < 0 ? -1 : ko + i - 6);
                    if (gi < 0) continue;
                    rt[gi] -= T[3 * i] * b[0] + T[3 * i + 1] * b[1] + T[3 * i + 2] * b[2];
                    for (int bb = 0; bb < L; ++bb)
                        for (int j = 0; j < 10; ++j) {
                            int gj = j < 6 ? (idx[bb] < 0 ? -1 : idx[bb] + j) : (ko < 0 ? -1 : ko + j - 6);
                            if (gj < 0) continue;
                            St[(size_t)gi * n + gj] -= T[3 * i] * W[bb][3 * j] + T[3 * i + 1] * W[bb][3 * j + 1] + T[3 * i + 2] * W[bb][3 * j + 2];
                        }
                }
            }
        }
        free(W); free(idx);
    }
    memset(S, 0, sizeof(double) * (size_t)n * n);
    memset(rhs, 0, sizeof(double) * (size_t)n);
    for (int t = 0; t < nthr; ++t) {
        const double* St = Sbuf + (size_t)t * ((size_t)n * n + n);
#pragma omp parallel for schedule(static)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) S[(size_t)i * n + j] += St[(size_t)i * n + j];
        for (int i = 0; i < n; ++i) rhs[i] += St[(size_t)n * n + i];
    }
    if (!g_skip_cam_damping) for (int i = 0; i < n; ++i) S[(size_t)i * n + i] += D2[i];
    free(Sbuf);
    return fail ? -1 : 0;
}

/* envelope (skyline) Cholesky S = L L' in place (lower), then solve L L' y = rhs */
static int skyline_solve(int n, double* S, const double* rhs, double* y)
{
    int* first = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) {
        int f = 0;
        while (f < i && S[(size_t)i * n + f] == 0.0) ++f;
        first[i] = f;
    }
    for (int i = 0; i < n; ++i) {
        double* Li = S + (size_t)i * n;
        for (int j = first[i]; j <= i; ++j) {
            const double* Lj = S + (size_t)j * n;
            int k0 = first[i] > first[j] ? first[i] : first[j];
            double s = Li[j];
            for (int k = k0; k < j; ++k) s -= Li[k] * Lj[k];
            if (j < i) Li[j] = s / Lj[j];
            else { if (!(s > 0.0) || !isfinite(s)) { free(first); return -1; } Li[i] = sqrt(s); }
        }
    }
    for (int i = 0; i < n; ++i) {
        const double* Li = S + (size_t)i * n;
        double s = rhs[i];
        for (int k = first[i]; k < i; ++k) s -= Li[k] * y[k];
        y[i] = s / Li[i];
    }
    for (int i = n - 1; i >= 0; --i) {
        const double* Li = S + (size_t)i * n;
        y[i] /= Li[i];
        for (int k = first[i]; k < i; ++k) y[k] -= Li[k] * y[i];
    }
    free(first);
    return 0;
}

/* step (scaled coordinates, [n | 3 np]) and model cost change for the given damping */
static int lm_step(const prob* P, const double* r, const double* J, const double* D2,
                   double* S, double* rhs, double* Vinv, double* bp, double* step, double* model_cost_change)
{
    const int n = P->n, np = P->np, ko = k_off(P);
    if (build_reduced(P, r, J, D2, S, rhs, Vinv, bp)) return -1;
    double* y = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    if (skyline_solve(n, S, rhs, y)) { free(y); return -1; }
    for (int i = 0; i < n; ++i) step[i] = -y[i];
    int bad = 0;
#pragma omp parallel for schedule(static)
    for (int p = 0; p < np; ++p) {
        int s0 = P->pt_start[p], L = P->pt_start[p + 1] - s0;
        double t[3] = { bp[3 * p], bp[3 * p + 1], bp[3 * p + 2] };
        for (int a = 0; a < L; ++a) {
            int k = P->pt_obs[s0 + a];
            int co = cam_off(P, P->ocam[k]);
            for (int row = 0; row < 2; ++row) {
                const double* d = J + 26 * (size_t)k + 13 * row;
                double e = 0.0;   /* E_row . y */
                if (co >= 0) for (int j = 0; j < 6; ++j) e += d[4 + j] * y[co + j];
                if (ko >= 0) for (int j = 0; j < 4; ++j) e += d[j] * y[ko + j];
                for (int j = 0; j < 3; ++j) t[j] -= d[10 + j] * e;
            }
        }
        const double* Vi = Vinv + 9 * (size_t)p;
        for (int i = 0; i < 3; ++i) {
            double v = Vi[3 * i] * t[0] + Vi[3 * i + 1] * t[1] + Vi[3 * i + 2] * t[2];
            if (!isfinite(v)) bad = 1;
            step[n + 3 * p + i] = -v;
        }
    }
    free(y);
    if (bad) return -1;
    double mcc = 0.0;
    for (int k = 0; k < P->nobs; ++k) {
        int co = cam_off(P, P->ocam[k]), p = P->opt[k];
        for (int row = 0; row < 2; ++row) {
            const double* d = J + 26 * (size_t)k + 13 * row;
            double m = 0.0;
            if (ko >= 0) for (int j = 0; j < 4; ++j) m += d[j] * step[ko + j];
            if (co >= 0) for (int j = 0; j < 6; ++j) m += d[4 + j] * step[co + j];
            for (int j = 0; j < 3; ++j) m += d[10 + j] * step[n + 3 * p + j];
            mcc -= m * (r[2 * k + row] + 0.5 * m);
        }
    }
    *model_cost_change = mcc;
    return 0;
}

/* pack / unpack free parameters [cams (free) | K | points] */
static void apply_delta(const prob* P, const double* K4, const double* ext, const double* pts, const double* delta,
                        double* K4o, double* exto, double* ptso)
{
    int ko = k_off(P);
    memcpy(K4o, K4, 4 * sizeof(double));
    memcpy(exto, ext, sizeof(double) * 6 * (size_t)P->nc);
    if (ko >= 0) for (int j = 0; j < 4; ++j) K4o[j] += delta[ko + j];
    for (int c = 0; c < P->nc; ++c) {
        int co = cam_off(P, c);
        if (co >= 0) for (int j = 0; j < 6; ++j) exto[6 * c + j] += delta[co + j];
    }
    for (size_t i = 0; i < 3 * (size_t)P->np; ++i) ptso[i] = pts[i] + delta[P->n + i];
}
static double free_norm(const prob* P, const double* K4, const double* ext, const double* pts)
{
    double s = 0;
    if (!P->fixK) for (int j = 0; j < 4; ++j) s += K4[j] * K4[j];
    for (int c = P->fix0; c < P->nc; ++c) for (int j = 0; j < 6; ++j) s += ext[6 * c + j] * ext[6 * c + j];
    for (size_t i = 0; i < 3 * (size_t)P->np; ++i) s += pts[i] * pts[i];
    return sqrt(s);
}
/* max |J' r| in unscaled coordinates from the scaled J */
static double grad_max_norm(const prob* P, const double* r, const double* J)
{
    size_t m = (size_t)P->n + 3 * (size_t)P->np;
    double* g = (double*)calloc(m, sizeof(double));
    int ko = k_off(P);
    for (int k = 0; k < P->nobs; ++k) {
        int co = cam_off(P, P->ocam[k]), p = P->opt[k];
        for (int row = 0; row < 2; ++row) {
            const double* d = J + 26 * (size_t)k + 13 * row; double rr = r[2 * k + row];
            if (ko >= 0) for (int j = 0; j < 4; ++j) g[ko + j] += d[j] * rr;
            if (co >= 0) for (int j = 0; j < 6; ++j) g[co + j] += d[4 + j] * rr;
            for (int j = 0; j < 3; ++j) g[P->n + 3 * p + j] += d[10 + j] * rr;
        }
    }
    double mx = 0;
    for (size_t i = 0; i < m; ++i) { double v = fabs(g[i] / (P->scale ? P->scale[i] : 1.0)); if (v > mx) mx = v; }
    free(g);
    return mx;
}
static void compute_scale(prob* P, const double* K4, const double* ext, const double* pts, double* r, double* J)
{
    size_t m = (size_t)P->n + 3 * (size_t)P->np;
    free(P->scale); P->scale = NULL;
    if (!P->o.jacobi_scaling) return;
    evaluate(P, K4, ext, pts, r, J);
    double* sc = (double*)malloc(sizeof(double) * m);
    col_sqnorm(P, J, sc);
    for (size_t i = 0; i < m; ++i) sc[i] = 1.0 / (1.0 + sqrt(sc[i]));
    P->scale = sc;
}

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

int orc_ba_solve(double* K4, double* ext, int nc, double* pts, int np,
                 const int32_t* ocam, const int32_t* opt, const double* ouv, int nobs,
                 const orc_ba_options* opts, orc_ba_summary* sum,
                 int force_iterations, double* trace_cost, double* trace_radius, int32_t* trace_ok, int trace_cap)
{
    orc_ba_options o; if (opts) o = *opts; else orc_ba_default_options(&o);
    prob P; if (prob_init(&P, nc, np, ocam, opt, ouv, nobs, &o)) { prob_free(&P); return -1; }
    const int n = P.n; const size_t m = (size_t)n + 3 * (size_t)np;
    double t0 = now_s();
    double* r = (double*)malloc(sizeof(double) * 2 * (size_t)(nobs > 0 ? nobs : 1));
    double* J = (double*)malloc(sizeof(double) * 26 * (size_t)(nobs > 0 ? nobs : 1));
    double* S = (double*)malloc(sizeof(double) * ((size_t)n * n + 1));
    double* rhs = (double*)malloc(sizeof(double) * (size_t)(n + 1));
    double* Vinv = (double*)malloc(sizeof(double) * 9 * (size_t)(np + 1));
    double* bp = (double*)malloc(sizeof(double) * 3 * (size_t)(np + 1));
    double* diag = (double*)malloc(sizeof(double) * (m + 1));
    double* D2 = (double*)malloc(sizeof(double) * (m + 1));
    double* step = (double*)malloc(sizeof(double) * (m + 1));
    double* delta = (double*)malloc(sizeof(double) * (m + 1));
    double* K4c = (double*)malloc(sizeof(double) * 4);
    double* extc = (double*)malloc(sizeof(double) * 6 * (size_t)nc);
    double* ptsc = (double*)malloc(sizeof(double) * 3 * (size_t)(np + 1));

    compute_scale(&P, K4, ext, pts, r, J);
    double x_cost = evaluate(&P, K4, ext, pts, r, J);
    double gmax = grad_max_norm(&P, r, J);
    double x_norm = free_norm(&P, K4, ext, pts);
    double radius = o.initial_trust_region_radius, nu = 2.0;
    int reuse_diag = 0, iter = 0, nsucc = 0, ninvalid = 0;
    int termination = 1; /* NO_CONVERGENCE */
    const int forced = force_iterations > 0;
    const int max_it = forced ? force_iterations : o.max_num_iterations;
    sum->initial_cost = x_cost; sum->num_residuals = 2 * nobs;
    if (o.verbose) printf("[orc_ba] it 0 cost %.12e gmax %.3e radius %.3e\n", x_cost, gmax, radius);
    for (;;) {
        if (iter >= max_it) { termination = 1; break; }
        if (!forced && gmax <= o.gradient_tolerance) { termination = 0; break; }
        if (!forced && radius < o.min_trust_region_radius) { termination = 0; break; }
        ++iter;
        if (!reuse_diag) {
            col_sqnorm(&P, J, diag);
            for (size_t i = 0; i < m; ++i) diag[i] = fmin(fmax(diag[i], o.min_lm_diagonal), o.max_lm_diagonal);
        }
        for (size_t i = 0; i < m; ++i) D2[i] = diag[i] / radius;
        reuse_diag = 1;
        double mcc = 0.0;
        int accepted = 0;
        int ok = lm_step(&P, r, J, D2, S, rhs, Vinv, bp, step, &mcc) == 0;
        if (!ok || !(mcc > 0.0)) {
            if (++ninvalid >= 5 && !forced) { termination = 2; if (iter <= trace_cap) { if (trace_cost) trace_cost[iter - 1] = x_cost; if (trace_radius) trace_radius[iter - 1] = radius; if (trace_ok) trace_ok[iter - 1] = 0; } break; }
            radius *= 0.5;
        } else {
            ninvalid = 0;
            for (size_t i = 0; i < m; ++i) delta[i] = step[i] * (P.scale ? P.scale[i] : 1.0);
            apply_delta(&P, K4, ext, pts, delta, K4c, extc, ptsc);
            double cand = evaluate(&P, K4c, extc, ptsc, NULL, NULL);
            if (!isfinite(cand)) cand = DBL_MAX;
            double sn = 0; for (size_t i = 0; i < m; ++i) sn += delta[i] * delta[i];
            sn = sqrt(sn);
            if (!forced && sn <= o.parameter_tolerance * (x_norm + o.parameter_tolerance)) { termination = 0; if (iter <= trace_cap) { if (trace_cost) trace_cost[iter - 1] = x_cost; if (trace_radius) trace_radius[iter - 1] = radius; if (trace_ok) trace_ok[iter - 1] = 0; } break; }
            double cost_change = x_cost - cand;
            if (!forced && fabs(cost_change) <= o.function_tolerance * x_cost) { termination = 0; if (iter <= trace_cap) { if (trace_cost) trace_cost[iter - 1] = x_cost; if (trace_radius) trace_radius[iter - 1] = radius; if (trace_ok) trace_ok[iter - 1] = 0; } break; }
            double rho = cost_change / mcc;
            if (rho > o.min_relative_decrease) {
                memcpy(K4, K4c, 4 * sizeof(double));
                memcpy(ext, extc, sizeof(double) * 6 * (size_t)nc);
                memcpy(pts, ptsc, sizeof(double) * 3 * (size_t)np);
                x_norm = free_norm(&P, K4, ext, pts);
                x_cost = evaluate(&P, K4, ext, pts, r, J);
                gmax = grad_max_norm(&P, r, J);
                double t = 2.0 * rho - 1.0;
                radius = radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
                radius = fmin(o.max_trust_region_radius, radius);
                nu = 2.0; reuse_diag = 0; ++nsucc; accepted = 1;
            } else {
                radius = radius / nu; nu *= 2.0; reuse_diag = 1;
            }
        }
        if (iter <= trace_cap) {
            if (trace_cost) trace_cost[iter - 1] = x_cost;
            if (trace_radius) trace_radius[iter - 1] = radius;
            if (trace_ok) trace_ok[iter - 1] = accepted;
        }
        if (o.verbose) printf("[orc_ba] it %d cost %.12e gmax %.3e radius %.3e %s\n", iter, x_cost, gmax, radius, accepted ? "ok" : "rejected");
    }
    sum->termination = termination; sum->iterations = iter; sum->successful_steps = nsucc;
    sum->final_cost = x_cost; sum->final_radius = radius; sum->final_gradient_max_norm = gmax;
    sum->total_time_s = now_s() - t0;
    free(r); free(J); free(S); free(rhs); free(Vinv); free(bp); free(diag); free(D2); free(step); free(delta);
    free(K4c); free(extc); free(ptsc);
    prob_free(&P);
    return 0;
}

int orc_ba_reduced_system(const double* K4, const double* ext, int nc, const double* pts, int np,
                          const int32_t* ocam, const int32_t* opt, const double* ouv, int nobs,
                          const orc_ba_options* opts, double radius, double* S, double* rhs, double* cost)
{
    orc_ba_options o; if (opts) o = *opts; else orc_ba_default_options(&o);
    prob P; if (prob_init(&P, nc, np, ocam, opt, ouv, nobs, &o)) { prob_free(&P); return -1; }
    const int n = P.n; const size_t m = (size_t)n + 3 * (size_t)np;
    if (!S || !rhs) { prob_free(&P); return n; }
    double* r = (double*)malloc(sizeof(double) * 2 * (size_t)(nobs > 0 ? nobs : 1));
    double* J = (double*)malloc(sizeof(double) * 26 * (size_t)(nobs > 0 ? nobs : 1));
    double* Vinv = (double*)malloc(sizeof(double) * 9 * (size_t)(np + 1));
    double* bp = (double*)malloc(sizeof(double) * 3 * (size_t)(np + 1));
    double* diag = (double*)malloc(sizeof(double) * (m + 1));
    compute_scale(&P, K4, ext, pts, r, J);
    double c = evaluate(&P, K4, ext, pts, r, J);
    if (cost) *cost = c;
    col_sqnorm(&P, J, diag);
    for (size_t i = 0; i < m; ++i) diag[i] = fmin(fmax(diag[i], o.min_lm_diagonal), o.max_lm_diagonal) / fabs(radius);
    g_skip_cam_damping = radius < 0.0;
    int rc = build_reduced(&P, r, J, diag, S, rhs, Vinv, bp);
    g_skip_cam_damping = 0;
    free(r); free(J); free(Vinv); free(bp); free(diag);
    prob_free(&P);
    return rc ? -1 : n;
}
