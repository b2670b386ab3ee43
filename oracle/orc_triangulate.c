/*
 * oracle/orc_triangulate.c -- CPU restatement of reconstruct() (TEST INFRASTRUCTURE, see orc.h).
 *
 * Follows NView:1117-1159.  cv::triangulatePoints (OpenCV 4.4.0 calib3d, not in /root/reference)
 * restated from its published source [3P]: per correspondence build the 4x4 system in double
 *     A[2j+0][k] = x_j * P_j[2][k] - P_j[0][k]
 *     A[2j+1][k] = y_j * P_j[2][k] - P_j[1][k]        j = 0,1 (views), k = 0..3
 * from float32 points and float32 3x4 projections, cv::SVD::compute(A, w, u, vt), output the last
 * row of vt (right singular vector of the smallest singular value) cast to the point type (float32).
 * The SVD here is a one-sided (Hestenes) Jacobi in double with OpenCV's own stopping rule for double input
 * (JacobiSVDImpl_ [3P]: a column pair is left alone once |a_p . a_q| <= 10 DBL_EPSILON sqrt(|a_p|^2 |a_q|^2), at most
 * 30 sweeps; rounds 1-2 used 1e-16 -- below the rounding noise of the dot product, so most systems ran all 30 sweeps);
 * OpenCV's JacobiSVD differs in sweep order, so the null vector agrees to ~1e-15*cond and in sign only up to +-1 (the
 * sign cancels in the division by w, NView:1154).
 * De-homogenisation NView:1153-1155: `Mat_<float> /= w` is convertTo(alpha = 1./w) [3P], i.e. a
 * float multiply by (float)(1.0/(double)w); then Point3f -> Point3d.
 * parity unpinned; cross-checked vs numpy.linalg.svd in tests/.
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <float.h>
#define ORC_SVD_EPS (10.0 * DBL_EPSILON)

/* null vector (unit 2-norm) of a 4x4 matrix: right singular vector of the smallest singular value */
static void null_vector4(const double Ain[16], double v[4])
{
    double A[4][4], V[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { A[i][j] = Ain[4 * i + j]; V[i][j] = (i == j); }
    for (int sweep = 0; sweep < 30; ++sweep) {
        int rotated = 0;
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 4; ++q) {
                double a = 0, b = 0, g = 0;
                for (int i = 0; i < 4; ++i) { a += A[i][p] * A[i][p]; b += A[i][q] * A[i][q]; g += A[i][p] * A[i][q]; }
                if (fabs(g) <= ORC_SVD_EPS * sqrt(a * b) || g == 0.0) continue;
                rotated = 1;
                double zeta = (b - a) / (2.0 * g);
                double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + tt * tt), s = c * tt;
                for (int i = 0; i < 4; ++i) {
                    double x = A[i][p], y = A[i][q];
                    A[i][p] = c * x - s * y; A[i][q] = s * x + c * y;
                    x = V[i][p]; y = V[i][q];
                    V[i][p] = c * x - s * y; V[i][q] = s * x + c * y;
                }
            }
        if (!rotated) break;
    }
    int m = 0; double best = INFINITY;
    for (int j = 0; j < 4; ++j) {
        double nn = 0;
        for (int i = 0; i < 4; ++i) nn += A[i][j] * A[i][j];
        if (nn < best) { best = nn; m = j; }
    }
    for (int i = 0; i < 4; ++i) v[i] = V[i][m];
}

void orc_triangulate2(const float P1[12], const float P2[12], const float* xy1, const float* xy2,
                      int n, float* xyzw, double* xyz)
{
    const float* P[2] = { P1, P2 };
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        double A[16], v[4];
        const float* pt[2] = { xy1 + 2 * i, xy2 + 2 * i };
        for (int j = 0; j < 2; ++j) {
            double x = pt[j][0], y = pt[j][1];
            for (int k = 0; k < 4; ++k) {
                A[4 * (2 * j) + k]     = x * (double)P[j][8 + k] - (double)P[j][k];
                A[4 * (2 * j + 1) + k] = y * (double)P[j][8 + k] - (double)P[j][4 + k];
            }
        }
        null_vector4(A, v);
        float h[4] = { (float)v[0], (float)v[1], (float)v[2], (float)v[3] };
        if (xyzw) for (int k = 0; k < 4; ++k) xyzw[(size_t)k * n + i] = h[k];
        if (xyz) {
            float inv = (float)(1.0 / (double)h[3]);
            xyz[3 * i + 0] = (double)(h[0] * inv);
            xyz[3 * i + 1] = (double)(h[1] * inv);
            xyz[3 * i + 2] = (double)(h[2] * inv);
        }
    }
}

/* NView:1129-1143: R, T, K converted to CV_32F, then proj = fK * [R|T] as a float32 cv::Mat product.
 * [3P] cv::gemm on CV_32F accumulates each 3-term dot product in double and rounds once to float
 * (GEMMSingleMul<float,double>), restated here. */
void orc_projection_matrix(const double K[9], const double R[9], const double T[3], float P[12])
{
    float fK[9], RT[12];
    for (int i = 0; i < 9; ++i) fK[i] = (float)K[i];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) RT[4 * r + c] = (float)R[3 * r + c];
        RT[4 * r + 3] = (float)T[r];
    }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += (double)fK[3 * r + k] * (double)RT[4 * k + c];
            P[4 * r + c] = (float)s;
        }
}

/* ---- N-view extension (NOT reference behaviour: the reference only ever triangulates from the pair
 * that created a point, NView:1428-1453).  Hartley-Zisserman multi-view DLT on NORMALISED image coordinates: for each
 * observation of the point in camera j with pixel (u, v):  xn = (u - cx)/fx, yn = (v - cy)/fy, [R_j | t_j] from the
 * angle-axis extrinsics (same rotation formula as the BA cost, NView:151-183), rows  xn*Rt[2] - Rt[0],  yn*Rt[2] - Rt[1].
 * M = A'A (4x4) accumulated in double, null vector by the same Jacobi routine as the two-view case, X = v[0:3]/v[3].
 * Points with fewer than 2 observations get NaN.  n_views[p] = number of observations used. */
static void angle_axis_to_Rt(const double* e, double Rt[12])
{
    const double th2 = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
    double R[9];
    if (th2 > 2.220446049250313e-16) {
        const double th = sqrt(th2), c = cos(th), s = sin(th), wx = e[0] / th, wy = e[1] / th, wz = e[2] / th, k = 1.0 - c;
        R[0] = c + wx * wx * k;      R[1] = wx * wy * k - wz * s; R[2] = wx * wz * k + wy * s;
        R[3] = wy * wx * k + wz * s; R[4] = c + wy * wy * k;      R[5] = wy * wz * k - wx * s;
        R[6] = wz * wx * k - wy * s; R[7] = wz * wy * k + wx * s; R[8] = c + wz * wz * k;
    } else {        /* first-order: X + aa x X */
        R[0] = 1; R[1] = -e[2]; R[2] = e[1]; R[3] = e[2]; R[4] = 1; R[5] = -e[0]; R[6] = -e[1]; R[7] = e[0]; R[8] = 1;
    }
    for (int r = 0; r < 3; ++r) { for (int c2 = 0; c2 < 3; ++c2) Rt[4 * r + c2] = R[3 * r + c2]; Rt[4 * r + 3] = e[3 + r]; }
}

void orc_triangulate_tracks(const double K4[4], const double* ext6, int n_cam, const int32_t* obs_cam, const int32_t* obs_pt,
                            const double* obs_uv, int n_obs, int n_pt, double* pts, int32_t* n_views)
{
    double* M = (double*)calloc((size_t)n_pt * 16, sizeof(double));
    int32_t* cnt = (int32_t*)calloc((size_t)n_pt, sizeof(int32_t));
    double* Rt = (double*)malloc((size_t)n_cam * 12 * sizeof(double));
    for (int c = 0; c < n_cam; ++c) angle_axis_to_Rt(ext6 + 6 * c, Rt + 12 * c);
    for (int k = 0; k < n_obs; ++k) {           /* observation order: the accumulation order the GPU kernel uses per point */
        const int p = obs_pt[k]; const double* P = Rt + 12 * obs_cam[k];
        const double xn = (obs_uv[2 * k] - K4[2]) / K4[0], yn = (obs_uv[2 * k + 1] - K4[3]) / K4[1];
        double r0[4], r1[4];
        for (int j = 0; j < 4; ++j) { r0[j] = xn * P[8 + j] - P[j]; r1[j] = yn * P[8 + j] - P[4 + j]; }
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) M[(size_t)p * 16 + 4 * i + j] += r0[i] * r0[j] + r1[i] * r1[j];
        cnt[p]++;
    }
    for (int p = 0; p < n_pt; ++p) {
        if (n_views) n_views[p] = cnt[p];
        if (cnt[p] < 2) { pts[3 * p] = pts[3 * p + 1] = pts[3 * p + 2] = NAN; continue; }
        double v[4];
        null_vector4(M + (size_t)p * 16, v);
        pts[3 * p] = v[0] / v[3]; pts[3 * p + 1] = v[1] / v[3]; pts[3 * p + 2] = v[2] / v[3];
    }
    free(M); free(cnt); free(Rt);
}

/* per-observation reprojection error in pixels at (K4, ext6, pts): |K (R X + t)/z - uv| */
void orc_reprojection_errors(const double K4[4], const double* ext6, int n_cam, const double* pts, const int32_t* obs_cam,
                             const int32_t* obs_pt, const double* obs_uv, int n_obs, double* err)
{
    double* Rt = (double*)malloc((size_t)n_cam * 12 * sizeof(double));
    for (int c = 0; c < n_cam; ++c) angle_axis_to_Rt(ext6 + 6 * c, Rt + 12 * c);
    for (int k = 0; k < n_obs; ++k) {
        const double* P = Rt + 12 * obs_cam[k]; const double* X = pts + 3 * (size_t)obs_pt[k];
        const double x = P[0] * X[0] + P[1] * X[1] + P[2] * X[2] + P[3], y = P[4] * X[0] + P[5] * X[1] + P[6] * X[2] + P[7];
        const double z = P[8] * X[0] + P[9] * X[1] + P[10] * X[2] + P[11];
        const double du = K4[0] * x / z + K4[2] - obs_uv[2 * k], dv = K4[1] * y / z + K4[3] - obs_uv[2 * k + 1];
        err[k] = sqrt(du * du + dv * dv);
    }
    free(Rt);
}
