/*
 * oracle/orc_normals.c -- CPU restatement of estimate_normals + PCAFitPlane (TEST INFRASTRUCTURE).
 *
 * Follows NView:551-599 (all other points pushed into a priority_queue ordered by the Euclidean
 * distance of Pt3dDist NView:468-482, the K nearest popped) and NView:601-690 (mean of the K
 * neighbours, covariance / K, eigen-decomposition, eigenvector of the smallest eigenvalue, flipped
 * when n . mean > 0 (NView:672), L2-normalised).  Eigen::EigenSolver is replaced by a cyclic Jacobi
 * eigen-solver for the symmetric 3x3 (same eigenvectors up to rounding; sign fixed by the flip).
 * Ties at the K-th distance follow std::priority_queue internals in the reference; here (d, index).
 * PINNED: reproduces Viewer/structure_ba.ply normals from Viewer/structure_ba.yml points
 * (tests/test_golden_outputs.py).
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>

static void eig3_sym_min(const double Ain[9], double v[3])
{
    double A[3][3], V[3][3] = { { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[i][j] = Ain[3 * i + j];
    for (int sweep = 0; sweep < 50; ++sweep) {
        double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off == 0.0) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (A[p][q] == 0.0) continue;
                double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) { double x = A[k][p], y = A[k][q]; A[k][p] = c * x - s * y; A[k][q] = s * x + c * y; }
                for (int k = 0; k < 3; ++k) { double x = A[p][k], y = A[q][k]; A[p][k] = c * x - s * y; A[q][k] = s * x + c * y; }
                for (int k = 0; k < 3; ++k) { double x = V[k][p], y = V[k][q]; V[k][p] = c * x - s * y; V[k][q] = s * x + c * y; }
            }
    }
    int m = 0;
    for (int j = 1; j < 3; ++j) if (A[j][j] < A[m][m]) m = j;
    for (int k = 0; k < 3; ++k) v[k] = V[k][m];
}

void orc_estimate_normals(const double* pts, int n, int K, double* normals)
{
#pragma omp parallel
    {
        double* bd = (double*)malloc(sizeof(double) * (size_t)K);
        int* bi = (int*)malloc(sizeof(int) * (size_t)K);
#pragma omp for schedule(static)
        for (int i = 0; i < n; ++i) {
            int cnt = 0;
            const double* pi = pts + 3 * (size_t)i;
            for (int j = 0; j < n; ++j) {
                if (j == i) continue;
                const double* pj = pts + 3 * (size_t)j;
                double dx = pi[0] - pj[0], dy = pi[1] - pj[1], dz = pi[2] - pj[2];
                double d = sqrt(dx * dx + dy * dy + dz * dz);
                if (cnt < K || d < bd[K - 1]) {
                    int k = cnt < K ? cnt++ : K - 1;
                    while (k > 0 && bd[k - 1] > d) { bd[k] = bd[k - 1]; bi[k] = bi[k - 1]; --k; }
                    bd[k] = d; bi[k] = j;
                }
            }
            double mean[3] = { 0, 0, 0 };
            for (int k = 0; k < cnt; ++k) for (int a = 0; a < 3; ++a) mean[a] += pts[3 * (size_t)bi[k] + a];
            for (int a = 0; a < 3; ++a) mean[a] /= (double)cnt;
            double C[9] = { 0 };
            for (int k = 0; k < cnt; ++k) {
                double d[3];
                for (int a = 0; a < 3; ++a) d[a] = pts[3 * (size_t)bi[k] + a] - mean[a];
                for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) C[3 * a + b] += d[a] * d[b];
            }
            for (int a = 0; a < 9; ++a) C[a] /= (double)cnt;
            double v[3];
            eig3_sym_min(C, v);
            if (v[0] * mean[0] + v[1] * mean[1] + v[2] * mean[2] > 0.0) { v[0] = -v[0]; v[1] = -v[1]; v[2] = -v[2]; }
            double nn = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            for (int a = 0; a < 3; ++a) normals[3 * (size_t)i + a] = v[a] / nn;
        }
        free(bd); free(bi);
    }
}
