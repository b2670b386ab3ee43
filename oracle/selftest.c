/*
 * oracle/selftest.c -- sanitizer run of the CPU restatement (TEST INFRASTRUCTURE; SURVEY 5: "ASan/UBSan target for the CPU
 * oracle").  Built by `make -C oracle selftest` with -fsanitize=address,undefined and executed by tests/test_oracle_sanitize.py:
 * every oracle entry point once on small synthetic inputs, incl. the edge cases the tests use (no train rows, one train row,
 * ragged Hamming rows, a point with a single observation).  Exit code 0 = no sanitizer report.
 */
#include "orc.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static unsigned long long rng_s = 88172645463325252ull;
static double urand(void) { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return (double)(rng_s >> 11) / 9007199254740992.0; }

int main(void)
{
    /* ---- matching ---- */
    const int nq = 37, nt = 41, dim = 128;
    float* q = malloc(sizeof(float) * nq * dim); float* t = malloc(sizeof(float) * nt * dim);
    for (int i = 0; i < nq * dim; ++i) q[i] = (float)(int)(urand() * 255);
    for (int i = 0; i < nt * dim; ++i) t[i] = i < nq * dim && (i / dim) % 2 == 0 ? q[i] : (float)(int)(urand() * 255);
    int32_t* idx = malloc(sizeof(int32_t) * 2 * nq); float* dist = malloc(sizeof(float) * 2 * nq);
    orc_dmatch* m = malloc(sizeof(orc_dmatch) * nq);
    for (int ntr = 0; ntr <= nt; ntr += (ntr < 2 ? 1 : nt - 2)) {            /* 0, 1, 2, nt train rows */
        orc_knn2_l2_f32(q, nq, t, ntr, dim, dim, dim, idx, dist);
        const int n = orc_ratio_filter(idx, dist, nq, 0.6, 10.0f, 5.0f, m);
        if (n < 0 || n > nq) return 10;
    }
    float* dm = malloc(sizeof(float) * nq * nt);
    orc_l2_distance_matrix_f32(q, nq, t, nt, dim, dim, dim, dm, nt);
    uint8_t* bq = malloc(61 * nq); uint8_t* bt = malloc(61 * nt);
    for (int i = 0; i < 61 * nq; ++i) bq[i] = (uint8_t)(urand() * 256);
    for (int i = 0; i < 61 * nt; ++i) bt[i] = (uint8_t)(urand() * 256);
    orc_knn2_hamming2_u8(bq, nq, bt, nt, 61, 61, 61, idx, dist);
    orc_knn2_hamming2_u8(bq, nq, bt, nt, 7, 61, 61, idx, dist);              /* ragged: 7-byte rows at stride 61 */
    /* ---- scene: 5 cameras on an arc, 60 points ---- */
    enum { NC = 5, NP = 60 };
    double K4[4] = { 2826.561, 2826.519, 1835.259, 1370.103 }, ext[6 * NC], pts[3 * NP];
    for (int c = 0; c < NC; ++c) { ext[6 * c] = 0.01 * c; ext[6 * c + 1] = -0.05 * c; ext[6 * c + 2] = 0.004 * c; ext[6 * c + 3] = -0.5 * c; ext[6 * c + 4] = 0.02 * c; ext[6 * c + 5] = 0.03 * c; }
    for (int p = 0; p < NP; ++p) { pts[3 * p] = 4 * urand() - 2; pts[3 * p + 1] = 3 * urand() - 1.5; pts[3 * p + 2] = 8 + 3 * urand(); }
    int32_t oc[NC * NP], op[NC * NP]; double uv[2 * NC * NP]; int nobs = 0;
    for (int p = 0; p < NP; ++p)
        for (int c = 0; c < NC; ++c) {
            if (p == 7 && c > 0) continue;                                   /* point 7: a single observation */
            if ((p + c) % 3 == 0 && p != 7) continue;
            double r[2], J[26], z[2] = { 0, 0 };
            orc_reproject(K4, ext + 6 * c, pts + 3 * p, z, r, J);
            oc[nobs] = c; op[nobs] = p; uv[2 * nobs] = r[0] + urand() - 0.5; uv[2 * nobs + 1] = r[1] + urand() - 0.5; ++nobs;
        }
    /* ---- triangulation ---- */
    double Kf[9] = { K4[0], 0, K4[2], 0, K4[1], K4[3], 0, 0, 1 }, R0[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }, T0[3] = { 0, 0, 0 }, T1[3] = { -0.5, 0.02, 0.03 };
    float P1[12], P2[12], xy1[2 * NP], xy2[2 * NP], xyzw[4 * NP]; double xyz[3 * NP];
    orc_projection_matrix(Kf, R0, T0, P1); orc_projection_matrix(Kf, R0, T1, P2);
    for (int p = 0; p < NP; ++p) { xy1[2 * p] = (float)(K4[0] * pts[3 * p] / pts[3 * p + 2] + K4[2]); xy1[2 * p + 1] = (float)(K4[1] * pts[3 * p + 1] / pts[3 * p + 2] + K4[3]);
                                   xy2[2 * p] = (float)(K4[0] * (pts[3 * p] + T1[0]) / (pts[3 * p + 2] + T1[2]) + K4[2]); xy2[2 * p + 1] = (float)(K4[1] * (pts[3 * p + 1] + T1[1]) / (pts[3 * p + 2] + T1[2]) + K4[3]); }
    orc_triangulate2(P1, P2, xy1, xy2, NP, xyzw, xyz);
    orc_triangulate2(P1, P2, xy1, xy2, NP, NULL, xyz);
    double tp[3 * NP]; int32_t nv[NP]; double err[NC * NP];
    orc_triangulate_tracks(K4, ext, NC, oc, op, uv, nobs, NP, tp, nv);
    orc_reprojection_errors(K4, ext, NC, pts, oc, op, uv, nobs, err);
    /* ---- bundle adjustment ---- */
    orc_ba_options o; orc_ba_default_options(&o);
    orc_set_num_threads(2);
    const int n = orc_ba_reduced_system(K4, ext, NC, pts, NP, oc, op, uv, nobs, &o, 1e4, NULL, NULL, NULL);
    double* S = malloc(sizeof(double) * n * n); double* rhs = malloc(sizeof(double) * n); double cost = 0;
    orc_ba_reduced_system(K4, ext, NC, pts, NP, oc, op, uv, nobs, &o, 1e4, S, rhs, &cost);
    orc_ba_reduced_system(K4, ext, NC, pts, NP, oc, op, uv, nobs, &o, -1e4, S, rhs, &cost);
    orc_ba_summary sum; double tc[64], tr[64]; int32_t tk[64];
    double K4b[4], extb[6 * NC], ptsb[3 * NP];
    memcpy(K4b, K4, sizeof K4); memcpy(extb, ext, sizeof ext); memcpy(ptsb, pts, sizeof pts);
    if (orc_ba_solve(K4b, extb, NC, ptsb, NP, oc, op, uv, nobs, &o, &sum, 0, tc, tr, tk, 64) != 0) return 11;
    o.fix_intrinsics = 1; o.huber_delta = 0; o.jacobi_scaling = 0; o.fix_first_camera = 0;
    memcpy(K4b, K4, sizeof K4); memcpy(extb, ext, sizeof ext); memcpy(ptsb, pts, sizeof pts);
    if (orc_ba_solve(K4b, extb, NC, ptsb, NP, oc, op, uv, nobs, &o, &sum, 3, NULL, NULL, NULL, 0) != 0) return 12;
    /* ---- normals ---- */
    double nrm[3 * NP];
    orc_estimate_normals(pts, NP, 10, nrm);
    orc_estimate_normals(pts, 5, 10, nrm);                                  /* fewer points than neighbours */
    if (!(sum.final_cost >= 0) || !isfinite(cost)) return 13;
    free(q); free(t); free(idx); free(dist); free(m); free(dm); free(bq); free(bt); free(S); free(rhs);
    printf("oracle selftest ok (%d observations, reduced order %d, max threads %d)\n", nobs, n, orc_get_max_threads());
    return 0;
}
