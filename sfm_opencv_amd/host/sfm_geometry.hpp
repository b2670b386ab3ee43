// sfm_geometry.hpp -- host-side pose estimation between the matcher and the triangulation / BA kernels:
// the reference's find_transform (NViewReconstuct.cpp:1022-1060 = cv::findEssentialMat(RANSAC, 0.999, 1.0) +
// cv::recoverPose + the gates 15 / 0.6 / 0.7) and the per-frame cv::solvePnPRansac (NView:1415, OpenCV defaults:
// 100 iterations, 8 px, confidence 0.99).  Header-only C++17 on the POD mirrors of sfm_ops.hpp, no OpenCV.
//
// PARITY UNPINNED, and un-pinnable: OpenCV's RANSAC draws from cv::RNG streams and uses Nister's five-point solver and
// EPnP minimal sets; none of that is in /root/reference and the reference holds no vectors for it.  These functions
// keep the reference's call surface, constants and gates, and are accepted on reconstruction quality (pose error and
// reprojection RMSE on synthetic scenes, tests/test_geometry_cpu.py), NOT on bit parity:
//   * essential matrix: RANSAC over normalised EIGHT-point samples (Hartley normalisation, essential-manifold
//     projection), Sampson error against threshold / focal like cv::findEssentialMat [3P], refit on the inliers;
//     the eight-point form is degenerate on planar scenes where cv's five-point solver is not, so a plane-induced
//     homography is fitted beside it (four-point RANSAC) and, where it explains the matches as well as the essential
//     matrix does, E = [t]x R comes from the homography's decomposition (Faugeras-Lustman) instead;
//   * recoverPose: cv::decomposeEssentialMat's four candidates, cheirality + the 50-unit distance gate [3P];
//   * PnP: RANSAC over FOUR-point samples -- a P3P solve (Grunert's quartic) on three, the fourth picks the root, as
//     cv's P3P kernel does for four correspondences [3P] (its EPnP kernel takes five) -- then the linear refit and
//     Levenberg-Marquardt on the inliers (what SOLVEPNP_ITERATIVE does after the RANSAC stage [3P]).  The only gate on
//     the number of correspondences is the reference's `< 4` (NView:1410-1414).
// Tiny data, CPU only -- this is plumbing, not a GPU target.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "sfm_ops.hpp"

namespace sfm {
namespace la {

// cyclic Jacobi eigen-solver, symmetric n x n (row-major, destroyed): eigenvalues w, eigenvectors as COLUMNS of V
inline void jacobi_eig_sym(int n, double* A, double* V, double* w)
{
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[i * n + j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; ++i) { diag += A[i * n + i] * A[i * n + i]; for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j]; }
        if (off <= 1e-32 * (diag + off) || off == 0.0) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[p * n + q];
                if (apq == 0.0) continue;
                const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) { const double x = A[k * n + p], y = A[k * n + q]; A[k * n + p] = c * x - s * y; A[k * n + q] = s * x + c * y; }
                for (int k = 0; k < n; ++k) { const double x = A[p * n + k], y = A[q * n + k]; A[p * n + k] = c * x - s * y; A[q * n + k] = s * x + c * y; }
                for (int k = 0; k < n; ++k) { const double x = V[k * n + p], y = V[k * n + q]; V[k * n + p] = c * x - s * y; V[k * n + q] = s * x + c * y; }
            }
    }
    for (int i = 0; i < n; ++i) w[i] = A[i * n + i];
}

// unit vector x minimising |M x| for the m x n matrix M (n <= 12): eigenvector of M'M with the smallest eigenvalue
inline void null_vector(const double* M, int m, int n, double* x)
{
    double G[144], V[144], w[12];
    for (int i = 0; i < n; ++i)
        for (int j = i; j < n; ++j) {
            double s = 0.0;
            for (int r = 0; r < m; ++r) s += M[r * n + i] * M[r * n + j];
            G[i * n + j] = G[j * n + i] = s;
        }
    jacobi_eig_sym(n, G, V, w);
    int k = 0;
    for (int i = 1; i < n; ++i) if (w[i] < w[k]) k = i;
    for (int i = 0; i < n; ++i) x[i] = V[i * n + k];
}

inline double det3(const double* m)
{
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}
inline void mul33(const double* a, const double* b, double* c)
{
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += a[3 * i + k] * b[3 * k + j]; c[3 * i + j] = s; }
}
inline void transpose33(const double* a, double* t) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) t[3 * j + i] = a[3 * i + j]; }

// A = U diag(S) Vt, S descending (3 x 3).  From the eigen-decomposition of A'A; a vanishing third singular value (rank-2
// essential matrices) gets its left vector from the cross product of the first two.
inline void svd3(const double* A, double* U, double* S, double* Vt)
{
    double G[9], V[9], w[3], At[9];
    transpose33(A, At); mul33(At, A, G);
    jacobi_eig_sym(3, G, V, w);
    int idx[3] = { 0, 1, 2 };
    std::sort(idx, idx + 3, [&](int a, int b) { return w[a] > w[b]; });
    double Vs[9];
    for (int k = 0; k < 3; ++k) { S[k] = std::sqrt(std::max(w[idx[k]], 0.0)); for (int i = 0; i < 3; ++i) Vs[3 * i + k] = V[3 * i + idx[k]]; }
    double u[3][3];
    for (int k = 0; k < 3; ++k)
        for (int i = 0; i < 3; ++i) { double s = 0; for (int j = 0; j < 3; ++j) s += A[3 * i + j] * Vs[3 * j + k]; u[k][i] = s; }
    auto norm = [](double* v) { const double n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); if (n > 0) { v[0] /= n; v[1] /= n; v[2] /= n; } return n; };
    norm(u[0]);
    {   // second: orthogonalise against the first (guards a nearly repeated singular value)
        const double d = u[1][0] * u[0][0] + u[1][1] * u[0][1] + u[1][2] * u[0][2];
        for (int i = 0; i < 3; ++i) u[1][i] -= d * u[0][i];
        if (norm(u[1]) < 1e-300) {      // rank <= 1: any unit vector orthogonal to u0
            const int a = std::fabs(u[0][0]) < 0.9 ? 0 : 1;
            double e[3] = { 0, 0, 0 }; e[a] = 1.0;
            const double dd = u[0][a];
            for (int i = 0; i < 3; ++i) u[1][i] = e[i] - dd * u[0][i];
            norm(u[1]);
        }
    }
    // third: the cross product keeps U orthogonal whatever S[2] is; flip it (and nothing else) if A v2 points the other way
    u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1]; u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2]; u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
    {
        double av[3];
        for (int i = 0; i < 3; ++i) { double s = 0; for (int j = 0; j < 3; ++j) s += A[3 * i + j] * Vs[3 * j + 2]; av[i] = s; }
        if (av[0] * u[2][0] + av[1] * u[2][1] + av[2] * u[2][2] < 0.0) for (int i = 0; i < 3; ++i) Vs[3 * i + 2] = -Vs[3 * i + 2];
    }
    for (int k = 0; k < 3; ++k) for (int i = 0; i < 3; ++i) U[3 * i + k] = u[k][i];
    transpose33(Vs, Vt);
}

// Cholesky solve of the SPD n x n system (n <= 6); false if not positive definite
inline bool solve_spd(int n, double* A, double* b)
{
    for (int j = 0; j < n; ++j) {
        double d = A[j * n + j];
        for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0.0)) return false;
        d = std::sqrt(d); A[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) { double s = A[i * n + j]; for (int k = 0; k < j; ++k) s -= A[i * n + k] * A[j * n + k]; A[i * n + j] = s / d; }
    }
    for (int i = 0; i < n; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= A[i * n + k] * b[k]; b[i] = s / A[i * n + i]; }
    for (int i = n - 1; i >= 0; --i) { double s = b[i]; for (int k = i + 1; k < n; ++k) s -= A[k * n + i] * b[k]; b[i] = s / A[i * n + i]; }
    return true;
}

// deterministic sample stream (the reference's RANSAC runs are repeatable too: cv::RNG has a fixed default seed [3P])
struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed = 0x9E3779B97F4A7C15ull) : s(seed) {}
    uint32_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 32); }
    int below(int n) { return (int)(((uint64_t)next() * (uint64_t)n) >> 32); }
    void sample(int n, int k, int* out)
    {
        for (int i = 0; i < k;) {
            const int c = below(n);
            bool dup = false;
            for (int j = 0; j < i; ++j) dup = dup || out[j] == c;
            if (!dup) out[i++] = c;
        }
    }
};

// cv::RANSACUpdateNumIters [3P]: iterations needed to draw an all-inlier sample of `model_points` with probability p
inline int ransac_update_iters(double p, double outlier_ratio, int model_points, int max_iters)
{
    p = std::min(std::max(p, 0.0), 1.0); outlier_ratio = std::min(std::max(outlier_ratio, 0.0), 1.0);
    const double num = std::max(1.0 - p, 2.2250738585072014e-308), denom = 1.0 - std::pow(1.0 - outlier_ratio, model_points);
    if (denom < 2.2250738585072014e-308) return 0;
    const double ln = std::log(num), ld = std::log(denom);
    return ld >= 0 || -ln >= max_iters * (-ld) ? max_iters : (int)std::lround(ln / ld);
}

}  // namespace la

namespace detail {

// eight-point (or more) essential matrix from normalised image coordinates: Hartley conditioning, null vector, projection
// onto the essential manifold (singular values 1, 1, 0)
inline bool essential_from_points(const std::vector<double>& x1, const std::vector<double>& x2, const int* idx, int n, double E[9])
{
    double c1[2] = { 0, 0 }, c2[2] = { 0, 0 };
    for (int k = 0; k < n; ++k) { const int i = idx ? idx[k] : k; c1[0] += x1[2 * i]; c1[1] += x1[2 * i + 1]; c2[0] += x2[2 * i]; c2[1] += x2[2 * i + 1]; }
    c1[0] /= n; c1[1] /= n; c2[0] /= n; c2[1] /= n;
    double d1 = 0, d2 = 0;
    for (int k = 0; k < n; ++k) {
        const int i = idx ? idx[k] : k;
        d1 += std::hypot(x1[2 * i] - c1[0], x1[2 * i + 1] - c1[1]); d2 += std::hypot(x2[2 * i] - c2[0], x2[2 * i + 1] - c2[1]);
    }
    if (!(d1 > 0.0) || !(d2 > 0.0)) return false;
    const double s1 = std::sqrt(2.0) * n / d1, s2 = std::sqrt(2.0) * n / d2;
    std::vector<double> A((size_t)std::max(n, 9) * 9, 0.0);
    for (int k = 0; k < n; ++k) {
        const int i = idx ? idx[k] : k;
        const double a = (x1[2 * i] - c1[0]) * s1, b = (x1[2 * i + 1] - c1[1]) * s1, u = (x2[2 * i] - c2[0]) * s2, v = (x2[2 * i + 1] - c2[1]) * s2;
        double* r = &A[(size_t)k * 9];
        r[0] = u * a; r[1] = u * b; r[2] = u; r[3] = v * a; r[4] = v * b; r[5] = v; r[6] = a; r[7] = b; r[8] = 1.0;      // x2' F x1 = 0
    }
    double f[9];
    la::null_vector(A.data(), std::max(n, 9), 9, f);
    // undo the conditioning: F = T2' f T1
    const double T1[9] = { s1, 0, -s1 * c1[0], 0, s1, -s1 * c1[1], 0, 0, 1 }, T2t[9] = { s2, 0, 0, 0, s2, 0, -s2 * c2[0], -s2 * c2[1], 1 };
    double tmp[9], F[9];
    la::mul33(f, T1, tmp); la::mul33(T2t, tmp, F);
    double U[9], S[3], Vt[9], D[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 0 };
    la::svd3(F, U, S, Vt);
    if (!(S[1] > 0.0)) return false;
    la::mul33(U, D, tmp); la::mul33(tmp, Vt, E);
    return true;
}

// cv::findEssentialMat's point error [3P]: squared Sampson distance of (x1, x2) to the epipolar constraint x2' E x1 = 0
inline double sampson_sq(const double E[9], double a, double b, double u, double v)
{
    const double e0 = E[0] * a + E[1] * b + E[2], e1 = E[3] * a + E[4] * b + E[5], e2 = E[6] * a + E[7] * b + E[8];
    const double t0 = E[0] * u + E[3] * v + E[6], t1 = E[1] * u + E[4] * v + E[7];
    const double r = u * e0 + v * e1 + e2;
    return r * r / (e0 * e0 + e1 * e1 + t0 * t0 + t1 * t1);
}

// linear two-view triangulation in normalised coordinates (P1 = [I | 0], P2 = [R | t]); homogeneous result
inline void triangulate_normalised(const double R[9], const double t[3], double a, double b, double u, double v, double X[4])
{
    const double P2[12] = { R[0], R[1], R[2], t[0], R[3], R[4], R[5], t[1], R[6], R[7], R[8], t[2] };
    double A[16];
    A[0] = -1; A[1] = 0; A[2] = a; A[3] = 0;
    A[4] = 0; A[5] = -1; A[6] = b; A[7] = 0;
    for (int j = 0; j < 4; ++j) { A[8 + j] = u * P2[8 + j] - P2[j]; A[12 + j] = v * P2[8 + j] - P2[4 + j]; }
    la::null_vector(A, 4, 4, X);
}

// plane-induced homography x2 ~ H x1 from four or more matches in normalised coordinates (DLT, Hartley conditioning)
inline bool homography_from_points(const std::vector<double>& x1, const std::vector<double>& x2, const int* idx, int n, double H[9])
{
    if (n < 4) return false;
    double m1[2] = { 0, 0 }, m2[2] = { 0, 0 }, s1 = 0, s2 = 0;
    for (int k = 0; k < n; ++k) { const int i = idx[k]; m1[0] += x1[2 * i]; m1[1] += x1[2 * i + 1]; m2[0] += x2[2 * i]; m2[1] += x2[2 * i + 1]; }
    for (int d = 0; d < 2; ++d) { m1[d] /= n; m2[d] /= n; }
    for (int k = 0; k < n; ++k) {
        const int i = idx[k];
        s1 += std::hypot(x1[2 * i] - m1[0], x1[2 * i + 1] - m1[1]); s2 += std::hypot(x2[2 * i] - m2[0], x2[2 * i + 1] - m2[1]);
    }
    if (s1 <= 0 || s2 <= 0) return false;
    s1 = std::sqrt(2.0) * n / s1; s2 = std::sqrt(2.0) * n / s2;
    double G[81] = { 0 };                                   // A'A of the 2n x 9 system, accumulated row by row
    for (int k = 0; k < n; ++k) {
        const int i = idx[k];
        const double a = (x1[2 * i] - m1[0]) * s1, b = (x1[2 * i + 1] - m1[1]) * s1, u = (x2[2 * i] - m2[0]) * s2, v = (x2[2 * i + 1] - m2[1]) * s2;
        const double r0[9] = { -a, -b, -1, 0, 0, 0, u * a, u * b, u }, r1[9] = { 0, 0, 0, -a, -b, -1, v * a, v * b, v };
        for (int p = 0; p < 9; ++p) for (int q = 0; q < 9; ++q) G[9 * p + q] += r0[p] * r0[q] + r1[p] * r1[q];
    }
    double V[81], w[9];
    la::jacobi_eig_sym(9, G, V, w);
    int kmin = 0;
    for (int i = 1; i < 9; ++i) if (w[i] < w[kmin]) kmin = i;
    double Hn[9];
    for (int i = 0; i < 9; ++i) Hn[i] = V[9 * i + kmin];
    // H = T2^-1 Hn T1 with T = [s 0 -s m; 0 s -s m; 0 0 1]
    const double T1[9] = { s1, 0, -s1 * m1[0], 0, s1, -s1 * m1[1], 0, 0, 1 }, T2i[9] = { 1 / s2, 0, m2[0], 0, 1 / s2, m2[1], 0, 0, 1 };
    double tmp[9];
    la::mul33(Hn, T1, tmp); la::mul33(T2i, tmp, H);
    return true;
}

// squared symmetric transfer error of (x1, x2) under H (normalised coordinates); Hi = H^-1
inline double homography_err(const double H[9], const double Hi[9], double a, double b, double u, double v)
{
    const double w2 = H[6] * a + H[7] * b + H[8], w1 = Hi[6] * u + Hi[7] * v + Hi[8];
    if (std::fabs(w2) < 1e-12 || std::fabs(w1) < 1e-12) return 1e300;
    const double eu = (H[0] * a + H[1] * b + H[2]) / w2 - u, ev = (H[3] * a + H[4] * b + H[5]) / w2 - v;
    const double ea = (Hi[0] * u + Hi[1] * v + Hi[2]) / w1 - a, eb = (Hi[3] * u + Hi[4] * v + Hi[5]) / w1 - b;
    return 0.5 * (eu * eu + ev * ev + ea * ea + eb * eb);
}

inline bool inv33(const double* m, double* o)
{
    const double d = la::det3(m);
    if (std::fabs(d) < 1e-300) return false;
    o[0] = (m[4] * m[8] - m[5] * m[7]) / d; o[1] = (m[2] * m[7] - m[1] * m[8]) / d; o[2] = (m[1] * m[5] - m[2] * m[4]) / d;
    o[3] = (m[5] * m[6] - m[3] * m[8]) / d; o[4] = (m[0] * m[8] - m[2] * m[6]) / d; o[5] = (m[2] * m[3] - m[0] * m[5]) / d;
    o[6] = (m[3] * m[7] - m[4] * m[6]) / d; o[7] = (m[1] * m[6] - m[0] * m[7]) / d; o[8] = (m[0] * m[4] - m[1] * m[3]) / d;
    return true;
}

// Faugeras & Lustman 1988: the (up to eight) motions (R, t / d) compatible with a plane-induced homography in normalised coordinates
inline int decompose_homography(const double H[9], double R[8][9], double t[8][3])
{
    double U[9], w[3], Vt[9];
    la::svd3(H, U, w, Vt);
    const double d1 = w[0], d2 = w[1], d3 = w[2];
    if (d2 <= 0 || d1 / d2 < 1.00001 || d2 / d3 < 1.00001) return 0;        // (near-)pure rotation or degenerate: no translation to recover
    const double s = la::det3(U) * la::det3(Vt);
    const double aux1 = std::sqrt((d1 * d1 - d2 * d2) / (d1 * d1 - d3 * d3)), aux3 = std::sqrt((d2 * d2 - d3 * d3) / (d1 * d1 - d3 * d3));
    const double x1[4] = { aux1, aux1, -aux1, -aux1 }, x3[4] = { aux3, -aux3, aux3, -aux3 };
    int m = 0;
    auto emit = [&](const double Rp[9], const double tp[3]) {
        double tmp[9];
        la::mul33(U, Rp, tmp); la::mul33(tmp, Vt, R[m]);
        for (double& v : R[m]) v *= s;
        for (int r = 0; r < 3; ++r) t[m][r] = U[3 * r] * tp[0] + U[3 * r + 1] * tp[1] + U[3 * r + 2] * tp[2];
        ++m;
    };
    {   // d' = d2
        const double st = std::sqrt((d1 * d1 - d2 * d2) * (d2 * d2 - d3 * d3)) / ((d1 + d3) * d2), ct = (d2 * d2 + d1 * d3) / ((d1 + d3) * d2);
        const double sgn[4] = { 1, -1, -1, 1 };
        for (int i = 0; i < 4; ++i) {
            const double Rp[9] = { ct, 0, -sgn[i] * st, 0, 1, 0, sgn[i] * st, 0, ct }, tp[3] = { (d1 - d3) * x1[i], 0, -(d1 - d3) * x3[i] };
            emit(Rp, tp);
        }
    }
    {   // d' = -d2
        const double sp = std::sqrt((d1 * d1 - d2 * d2) * (d2 * d2 - d3 * d3)) / ((d1 - d3) * d2), cp = (d1 * d3 - d2 * d2) / ((d1 - d3) * d2);
        const double sgn[4] = { 1, -1, -1, 1 };
        for (int i = 0; i < 4; ++i) {
            const double Rp[9] = { cp, 0, sgn[i] * sp, 0, -1, 0, sgn[i] * sp, 0, -cp }, tp[3] = { (d1 + d3) * x1[i], 0, (d1 + d3) * x3[i] };
            emit(Rp, tp);
        }
    }
    return m;
}

inline void essential_from_motion(const double R[9], const double t[3], double E[9])
{
    const double tx[9] = { 0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0 };
    la::mul33(tx, R, E);
}

// rotation exp([w]x) (Rodrigues' formula)
inline void rot_exp(const double w[3], double R[9])
{
    const double th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const double a = th < 1e-9 ? 1.0 - th * th / 6.0 : std::sin(th) / th, b = th < 1e-9 ? 0.5 - th * th / 24.0 : (1.0 - std::cos(th)) / (th * th);
    const double K[9] = { 0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0 };
    double K2[9];
    la::mul33(K, K, K2);
    for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * K[i] + b * K2[i];
}

// Levenberg-Marquardt on the Sampson distances of the listed matches over the five degrees of freedom of (R, t): R <- exp(dw) R,
// t <- normalise(t + B dt) with B spanning the tangent plane of the unit sphere at t; forward-difference Jacobian.  Polishes a
// motion that is roughly right (from a dominant plane's homography, or an ill-conditioned eight-point fit on a shallow scene).
inline void refine_motion(double R[9], double t[3], const std::vector<double>& x1, const std::vector<double>& x2, const std::vector<int>& in, int max_it)
{
    auto residuals = [&](const double Rr[9], const double tt[3], std::vector<double>& r) {
        double E[9];
        essential_from_motion(Rr, tt, E);
        r.resize(in.size());
        double s = 0.0;
        for (size_t k = 0; k < in.size(); ++k) {
            const int i = in[k];
            const double a = x1[2 * i], b = x1[2 * i + 1], u = x2[2 * i], v = x2[2 * i + 1];
            const double e0 = E[0] * a + E[1] * b + E[2], e1 = E[3] * a + E[4] * b + E[5], e2 = E[6] * a + E[7] * b + E[8];
            const double t0 = E[0] * u + E[3] * v + E[6], t1 = E[1] * u + E[4] * v + E[7];
            r[k] = (u * e0 + v * e1 + e2) / std::sqrt(e0 * e0 + e1 * e1 + t0 * t0 + t1 * t1 + 1e-300);
            s += r[k] * r[k];
        }
        return s;
    };
    auto apply = [&](const double d[5], double Rn[9], double tn[3]) {
        double dR[9];
        rot_exp(d, dR);
        la::mul33(dR, R, Rn);
        // tangent basis at t
        double b1[3], b2[3];
        const int k = std::fabs(t[0]) < std::fabs(t[1]) ? (std::fabs(t[0]) < std::fabs(t[2]) ? 0 : 2) : (std::fabs(t[1]) < std::fabs(t[2]) ? 1 : 2);
        double e[3] = { 0, 0, 0 }; e[k] = 1.0;
        b1[0] = t[1] * e[2] - t[2] * e[1]; b1[1] = t[2] * e[0] - t[0] * e[2]; b1[2] = t[0] * e[1] - t[1] * e[0];
        const double n1 = std::sqrt(b1[0] * b1[0] + b1[1] * b1[1] + b1[2] * b1[2]);
        for (double& v : b1) v /= n1;
        b2[0] = t[1] * b1[2] - t[2] * b1[1]; b2[1] = t[2] * b1[0] - t[0] * b1[2]; b2[2] = t[0] * b1[1] - t[1] * b1[0];
        double nn = 0.0;
        for (int a = 0; a < 3; ++a) { tn[a] = t[a] + d[3] * b1[a] + d[4] * b2[a]; nn += tn[a] * tn[a]; }
        nn = std::sqrt(nn);
        for (int a = 0; a < 3; ++a) tn[a] /= nn;
    };
    if (in.size() < 6) return;
    std::vector<double> r0, r1;
    double cost = residuals(R, t, r0), lambda = 1e-3;
    for (int it = 0; it < max_it; ++it) {
        std::vector<double> J(5 * in.size());
        const double h = 1e-6;
        for (int c = 0; c < 5; ++c) {
            double d[5] = { 0, 0, 0, 0, 0 }, Rn[9], tn[3];
            d[c] = h;
            apply(d, Rn, tn);
            residuals(Rn, tn, r1);
            for (size_t k = 0; k < in.size(); ++k) J[5 * k + c] = (r1[k] - r0[k]) / h;
        }
        double A[25] = { 0 }, g[5] = { 0 };
        for (size_t k = 0; k < in.size(); ++k)
            for (int a = 0; a < 5; ++a) { g[a] += J[5 * k + a] * r0[k]; for (int b = 0; b < 5; ++b) A[5 * a + b] += J[5 * k + a] * J[5 * k + b]; }
        bool improved = false;
        for (int tries = 0; tries < 8 && !improved; ++tries) {
            double M[25], rhs[5];
            for (int a = 0; a < 25; ++a) M[a] = A[a];
            for (int a = 0; a < 5; ++a) { M[6 * a] += lambda * (A[6 * a] + 1e-12); rhs[a] = -g[a]; }
            if (!la::solve_spd(5, M, rhs)) { lambda *= 10; continue; }
            double Rn[9], tn[3];
            apply(rhs, Rn, tn);
            const double c1 = residuals(Rn, tn, r1);
            if (c1 < cost) {
                std::copy(Rn, Rn + 9, R); std::copy(tn, tn + 3, t);
                const bool tiny = cost - c1 <= 1e-12 * cost;
                cost = c1; r0.swap(r1); lambda = std::max(lambda * 0.3, 1e-9); improved = true;
                if (tiny) return;
            } else lambda *= 10;
        }
        if (!improved) return;
    }
}

}  // namespace detail

// cv::findEssentialMat(points1, points2, focal, pp, RANSAC, prob, threshold, mask) (NView:1032): returns the 3 x 3 CV_64F
// essential matrix (empty Mat on failure) and the N x 1 CV_8U inlier mask.
enum { RANSAC = 8 };
inline Mat findEssentialMat(const std::vector<Point2f>& p1, const std::vector<Point2f>& p2, double focal, Point2d pp,
                            int /*method: RANSAC*/, double prob, double threshold, Mat& mask)
{
    const int n = (int)p1.size();
    mask = Mat(n, 1, CV_8U);
    if (n < 8 || p2.size() != p1.size()) return Mat();
    std::vector<double> x1(2 * (size_t)n), x2(2 * (size_t)n);
    for (int i = 0; i < n; ++i) {
        x1[2 * i] = (p1[i].x - pp.x) / focal; x1[2 * i + 1] = (p1[i].y - pp.y) / focal;
        x2[2 * i] = (p2[i].x - pp.x) / focal; x2[2 * i + 1] = (p2[i].y - pp.y) / focal;
    }
    const double thr2 = (threshold / focal) * (threshold / focal);        // cv: threshold /= focal [3P]
    auto count = [&](const double E[9], std::vector<uint8_t>* m) {
        int c = 0;
        for (int i = 0; i < n; ++i) {
            const bool in = detail::sampson_sq(E, x1[2 * i], x1[2 * i + 1], x2[2 * i], x2[2 * i + 1]) <= thr2;
            if (m) (*m)[i] = in ? 1 : 0;
            c += in;
        }
        return c;
    };
    la::Rng rng;
    double best[9] = { 0 }; int best_in = 0;
    int iters = 1000;                                                       // cv's maxIters for the essential matrix [3P]
    for (int it = 0; it < iters; ++it) {
        int idx[8];
        rng.sample(n, 8, idx);
        double E[9];
        if (!detail::essential_from_points(x1, x2, idx, 8, E)) continue;
        const int c = count(E, nullptr);
        if (c > std::max(best_in, 7)) {
            best_in = c; std::copy(E, E + 9, best);
            iters = std::min(iters, std::max(it + 1, la::ransac_update_iters(prob, (double)(n - c) / n, 8, iters)));
        }
    }
    if (best_in < 8) return Mat();
    std::vector<uint8_t> m((size_t)n, 0);
    count(best, &m);
    // least-squares refit on the consensus set (twice: the set may grow), kept only while it does not lose support
    for (int pass = 0; pass < 2; ++pass) {
        std::vector<int> in;
        for (int i = 0; i < n; ++i) if (m[i]) in.push_back(i);
        double E[9];
        if (!detail::essential_from_points(x1, x2, in.data(), (int)in.size(), E)) break;
        std::vector<uint8_t> m2((size_t)n, 0);
        const int c = count(E, &m2);
        if (c < best_in) break;
        best_in = c; std::copy(E, E + 9, best); m.swap(m2);
    }
    // Planar and shallow scenes.  Every E compatible with a plane's homography has zero epipolar error on that plane, so the
    // eight-point solution is arbitrary inside that family on a planar scene and ill-conditioned on a shallow one (cv's five-point
    // solver is degenerate in neither [3P]).  So the motion is also read off the dominant plane: homography by four-point RANSAC,
    // Faugeras-Lustman decomposition, the candidates in front of both cameras; every candidate motion -- those and the eight-point
    // one -- is then polished on its own Sampson inliers (five degrees of freedom), and the one that explains the most matches wins
    // (ties: smaller error).  On an exactly planar scene two motions fit alike (the classic twofold ambiguity, the same two answers a
    // five-point solver gives) and either is a correct reading of the data.
    {
        struct Cand { double R[9], t[3]; };
        std::vector<Cand> cands;
        auto front_count = [&](const double Rk[9], const double tk[3], const std::vector<int>& on) {
            int front = 0;
            for (int i : on) {
                double X[4];
                detail::triangulate_normalised(Rk, tk, x1[2 * i], x1[2 * i + 1], x2[2 * i], x2[2 * i + 1], X);
                if (X[3] == 0.0) continue;
                const double x = X[0] / X[3], y = X[1] / X[3], z = X[2] / X[3];
                front += z > 0 && Rk[6] * x + Rk[7] * y + Rk[8] * z + tk[2] > 0;
            }
            return front;
        };
        std::vector<int> e_in;
        for (int i = 0; i < n; ++i) if (m[i]) e_in.push_back(i);
        {   // the eight-point estimate's motion (the decomposition cv::recoverPose would pick)
            double U[9], S[3], Vt[9];
            la::svd3(best, U, S, Vt);
            if (la::det3(U) < 0) for (double& v : U) v = -v;
            if (la::det3(Vt) < 0) for (double& v : Vt) v = -v;
            const double W[9] = { 0, 1, 0, -1, 0, 0, 0, 0, 1 }, Wt[9] = { 0, -1, 0, 1, 0, 0, 0, 0, 1 };
            double tmp[9], Ra[9], Rb[9];
            la::mul33(U, W, tmp); la::mul33(tmp, Vt, Ra);
            la::mul33(U, Wt, tmp); la::mul33(tmp, Vt, Rb);
            Cand best_c; int best_f = -1;
            for (int k = 0; k < 4; ++k) {
                Cand c;
                std::copy(k & 1 ? Rb : Ra, (k & 1 ? Rb : Ra) + 9, c.R);
                for (int a = 0; a < 3; ++a) c.t[a] = (k & 2 ? -1.0 : 1.0) * U[3 * a + 2];
                const int f = front_count(c.R, c.t, e_in);
                if (f > best_f) { best_f = f; best_c = c; }
            }
            if (best_f >= 0) cands.push_back(best_c);
        }
        {   // the dominant plane's motions
            la::Rng hr(0xD1B54A32D192ED03ull);
            double Hb[9] = { 0 }; int h_in = 0, hit = 500;
            auto hcount = [&](const double H[9], std::vector<uint8_t>* mm) {
                double Hi[9];
                if (!detail::inv33(H, Hi)) return 0;
                int c = 0;
                for (int i = 0; i < n; ++i) {
                    const bool inl = detail::homography_err(H, Hi, x1[2 * i], x1[2 * i + 1], x2[2 * i], x2[2 * i + 1]) <= 4.0 * thr2;
                    if (mm) (*mm)[i] = inl ? 1 : 0;
                    c += inl;
                }
                return c;
            };
            for (int it = 0; it < hit; ++it) {
                int idx[4];
                hr.sample(n, 4, idx);
                double H[9];
                if (!detail::homography_from_points(x1, x2, idx, 4, H)) continue;
                const int c = hcount(H, nullptr);
                if (c > std::max(h_in, 3)) {
                    h_in = c; std::copy(H, H + 9, Hb);
                    hit = std::min(hit, std::max(it + 1, la::ransac_update_iters(prob, (double)(n - c) / n, 4, hit)));
                }
            }
            if (h_in >= 12) {
                std::vector<uint8_t> hm((size_t)n, 0);
                hcount(Hb, &hm);
                std::vector<int> on;
                for (int i = 0; i < n; ++i) if (hm[i]) on.push_back(i);
                double H[9];
                if (detail::homography_from_points(x1, x2, on.data(), (int)on.size(), H) && hcount(H, nullptr) >= h_in) std::copy(H, H + 9, Hb);
                double Rc[8][9], tc[8][3];
                const int nc = detail::decompose_homography(Hb, Rc, tc);
                for (int k = 0; k < nc; ++k) {
                    const double tn = std::sqrt(tc[k][0] * tc[k][0] + tc[k][1] * tc[k][1] + tc[k][2] * tc[k][2]);
                    if (tn <= 0) continue;
                    Cand c;
                    std::copy(Rc[k], Rc[k] + 9, c.R);
                    for (int a = 0; a < 3; ++a) c.t[a] = tc[k][a] / tn;
                    if (front_count(c.R, c.t, on) >= 0.9 * (int)on.size()) cands.push_back(c);
                }
            }
        }
        int win = -1, win_c = 0; double win_err = 1e300; double Ewin[9];
        for (size_t k = 0; k < cands.size(); ++k) {
            Cand c = cands[k];
            double E[9];
            for (int round = 0; round < 3; ++round) {           // re-take the consensus set as the motion improves
                detail::essential_from_motion(c.R, c.t, E);
                std::vector<int> on;
                for (int i = 0; i < n; ++i) if (detail::sampson_sq(E, x1[2 * i], x1[2 * i + 1], x2[2 * i], x2[2 * i + 1]) <= (round == 0 ? 9.0 : 1.0) * thr2) on.push_back(i);
                if (on.size() < 8) break;
                detail::refine_motion(c.R, c.t, x1, x2, on, 10);
            }
            detail::essential_from_motion(c.R, c.t, E);
            int cnt = 0; double err = 0.0;
            for (int i = 0; i < n; ++i) { const double e = detail::sampson_sq(E, x1[2 * i], x1[2 * i + 1], x2[2 * i], x2[2 * i + 1]); if (e <= thr2) { ++cnt; err += e; } }
            if (cnt > win_c || (cnt == win_c && err < win_err)) { win = (int)k; win_c = cnt; win_err = err; std::copy(E, E + 9, Ewin); }
        }
        if (win >= 0 && win_c >= best_in) { std::copy(Ewin, Ewin + 9, best); best_in = count(best, &m); }
    }
    for (int i = 0; i < n; ++i) mask.at<uint8_t>(i) = m[i];
    Mat Em(3, 3, CV_64F);
    std::copy(best, best + 9, Em.ptr<double>());
    return Em;
}

inline int countNonZero(const Mat& m)
{
    int c = 0;
    for (int i = 0; i < m.rows * m.cols; ++i) c += m.ptr<uint8_t>()[i] != 0;
    return c;
}

// cv::recoverPose(E, points1, points2, R, t, focal, pp, mask) (NView:1048): the candidate of cv::decomposeEssentialMat
// with the most points in front of both cameras and nearer than 50 baselines [3P]; mask (in/out) keeps those points.
inline int recoverPose(const Mat& E, const std::vector<Point2f>& p1, const std::vector<Point2f>& p2, Mat& R, Mat& T,
                       double focal, Point2d pp, Mat& mask)
{
    const int n = (int)p1.size();
    double U[9], S[3], Vt[9];
    la::svd3(E.ptr<double>(), U, S, Vt);
    if (la::det3(U) < 0) for (double& v : U) v = -v;
    if (la::det3(Vt) < 0) for (double& v : Vt) v = -v;
    const double W[9] = { 0, 1, 0, -1, 0, 0, 0, 0, 1 }, Wt[9] = { 0, -1, 0, 1, 0, 0, 0, 0, 1 };
    double tmp[9], R1[9], R2[9];
    la::mul33(U, W, tmp); la::mul33(tmp, Vt, R1);
    la::mul33(U, Wt, tmp); la::mul33(tmp, Vt, R2);
    const double t[3] = { U[2], U[5], U[8] }, tn[3] = { -U[2], -U[5], -U[8] };
    const double* Rc[4] = { R1, R2, R1, R2 };
    const double* tc[4] = { t, t, tn, tn };
    const double dist = 50.0;
    const bool has_mask = mask.rows * mask.cols == n;
    std::vector<uint8_t> good[4];
    int cnt[4] = { 0, 0, 0, 0 };
    for (int k = 0; k < 4; ++k) {
        good[k].assign((size_t)n, 0);
        for (int i = 0; i < n; ++i) {
            if (has_mask && !mask.ptr<uint8_t>()[i]) continue;
            const double a = (p1[i].x - pp.x) / focal, b = (p1[i].y - pp.y) / focal, u = (p2[i].x - pp.x) / focal, v = (p2[i].y - pp.y) / focal;
            double X[4];
            detail::triangulate_normalised(Rc[k], tc[k], a, b, u, v, X);
            if (X[3] == 0.0) continue;
            const double x = X[0] / X[3], y = X[1] / X[3], z = X[2] / X[3];
            const double z2 = Rc[k][6] * x + Rc[k][7] * y + Rc[k][8] * z + tc[k][2];
            if (z > 0 && z < dist && z2 > 0 && z2 < dist) { good[k][i] = 1; ++cnt[k]; }
        }
    }
    int k = 3;
    if (cnt[0] >= cnt[1] && cnt[0] >= cnt[2] && cnt[0] >= cnt[3]) k = 0;
    else if (cnt[1] >= cnt[0] && cnt[1] >= cnt[2] && cnt[1] >= cnt[3]) k = 1;
    else if (cnt[2] >= cnt[0] && cnt[2] >= cnt[1] && cnt[2] >= cnt[3]) k = 2;
    R = Mat(3, 3, CV_64F); T = Mat(3, 1, CV_64F);
    std::copy(Rc[k], Rc[k] + 9, R.ptr<double>()); std::copy(tc[k], tc[k] + 3, T.ptr<double>());
    mask = Mat(n, 1, CV_8U);
    for (int i = 0; i < n; ++i) mask.at<uint8_t>(i) = good[k][i];
    return cnt[k];
}

// find_transform (NView:1022-1060), gates and prints as the reference; mask comes back N x 1 CV_8U
inline bool find_transform(const Mat& K, const std::vector<Point2f>& p1, const std::vector<Point2f>& p2, Mat& R, Mat& T, Mat& mask)
{
    const double focal_length = 0.5 * (K.ptr<double>()[0] + K.ptr<double>()[4]);
    const Point2d principle_point{ K.ptr<double>()[2], K.ptr<double>()[5] };
    const Mat E = findEssentialMat(p1, p2, focal_length, principle_point, RANSAC, 0.999, 1.0, mask);
    if (E.empty()) return false;
    const double feasible_count = countNonZero(mask);
    if (feasible_count <= 15 || (feasible_count / p1.size()) < 0.6) return false;
    const int pass_count = recoverPose(E, p1, p2, R, T, focal_length, principle_point, mask);
    auto show = [](const char* name, const Mat& m) {
        printf("%s:\n[", name);
        for (int r = 0; r < m.rows; ++r) { for (int c = 0; c < m.cols; ++c) printf("%s%.17g", c ? ", " : "", m.at<double>(r, c)); printf(r + 1 < m.rows ? ";\n " : "]\n"); }
    };
    show("Init R", R); show("Init T", T);
    if (((double)pass_count) / feasible_count < 0.7) return false;
    return true;
}

namespace detail {

struct Pose { double R[9]; double t[3]; };

// six (or more) point DLT pose on normalised image coordinates and centred / scaled object points
inline bool pose_dlt(const std::vector<double>& X, const std::vector<double>& x, const int* idx, int n, Pose& P)
{
    double c[3] = { 0, 0, 0 };
    for (int k = 0; k < n; ++k) { const int i = idx ? idx[k] : k; for (int a = 0; a < 3; ++a) c[a] += X[3 * i + a]; }
    for (double& v : c) v /= n;
    double sc = 0;
    for (int k = 0; k < n; ++k) { const int i = idx ? idx[k] : k; sc += std::sqrt((X[3 * i] - c[0]) * (X[3 * i] - c[0]) + (X[3 * i + 1] - c[1]) * (X[3 * i + 1] - c[1]) + (X[3 * i + 2] - c[2]) * (X[3 * i + 2] - c[2])); }
    if (!(sc > 0.0)) return false;
    sc = std::sqrt(3.0) * n / sc;
    std::vector<double> A((size_t)std::max(2 * n, 12) * 12, 0.0);
    for (int k = 0; k < n; ++k) {
        const int i = idx ? idx[k] : k;
        const double Xc[4] = { (X[3 * i] - c[0]) * sc, (X[3 * i + 1] - c[1]) * sc, (X[3 * i + 2] - c[2]) * sc, 1.0 };
        double* r0 = &A[(size_t)(2 * k) * 12]; double* r1 = r0 + 12;
        for (int a = 0; a < 4; ++a) { r0[a] = Xc[a]; r0[8 + a] = -x[2 * i] * Xc[a]; r1[4 + a] = Xc[a]; r1[8 + a] = -x[2 * i + 1] * Xc[a]; }
    }
    double m[12];
    la::null_vector(A.data(), std::max(2 * n, 12), 12, m);
    double M3[9] = { m[0], m[1], m[2], m[4], m[5], m[6], m[8], m[9], m[10] };
    if (la::det3(M3) < 0) { for (double& v : m) v = -v; for (double& v : M3) v = -v; }
    double U[9], S[3], Vt[9];
    la::svd3(M3, U, S, Vt);
    if (!(S[2] > 1e-12 * S[0])) return false;
    la::mul33(U, Vt, P.R);
    if (la::det3(P.R) < 0) return false;
    const double scale = (S[0] + S[1] + S[2]) / 3.0;
    // M = scale [R | t'] acts on the conditioned points Xc = sc (X - c):  R X + t = R sc^-1... undo: p = R (sc (X - c)) + t'
    const double tp[3] = { m[3] / scale, m[7] / scale, m[11] / scale };
    for (int a = 0; a < 3; ++a) P.t[a] = (tp[a] - sc * (P.R[3 * a] * c[0] + P.R[3 * a + 1] * c[1] + P.R[3 * a + 2] * c[2])) / sc;
    // p = sc (R X + t): a positive factor that the projection ignores
    return true;
}

// roots of a real polynomial c[0] + c[1] x + ... + c[n] x^n (n <= 4) by Durand-Kerner on the monic form, real ones polished by Newton
inline int real_roots(const double* c, int n, double* out)
{
    while (n > 0 && std::fabs(c[n]) <= 1e-14 * (std::fabs(c[0]) + std::fabs(c[1]) + (n > 1 ? std::fabs(c[2]) : 0.0) + (n > 2 ? std::fabs(c[3]) : 0.0) + (n > 3 ? std::fabs(c[4]) : 0.0))) --n;
    if (n <= 0) return 0;
    double a[5];
    for (int i = 0; i <= n; ++i) a[i] = c[i] / c[n];
    double zr[4], zi[4];
    double rad = 0.0;
    for (int i = 0; i < n; ++i) rad = std::max(rad, std::fabs(a[i]));
    rad = 1.0 + rad;
    for (int i = 0; i < n; ++i) { const double ang = 0.4 + 6.283185307179586 * i / n; zr[i] = 0.6 * rad * std::cos(ang); zi[i] = 0.6 * rad * std::sin(ang); }
    for (int it = 0; it < 200; ++it) {
        double moved = 0.0;
        for (int i = 0; i < n; ++i) {
            double pr = 1.0, pi = 0.0;                      // p(z_i), Horner on the monic polynomial
            for (int k = n - 1; k >= 0; --k) { const double t = pr * zr[i] - pi * zi[i] + a[k]; pi = pr * zi[i] + pi * zr[i]; pr = t; }
            double dr = 1.0, di = 0.0;                      // prod_{j != i} (z_i - z_j)
            for (int j = 0; j < n; ++j) if (j != i) { const double xr = zr[i] - zr[j], xi = zi[i] - zi[j]; const double t = dr * xr - di * xi; di = dr * xi + di * xr; dr = t; }
            const double den = dr * dr + di * di;
            if (den == 0.0) continue;
            const double qr = (pr * dr + pi * di) / den, qi = (pi * dr - pr * di) / den;
            zr[i] -= qr; zi[i] -= qi;
            moved = std::max(moved, std::fabs(qr) + std::fabs(qi));
        }
        if (moved <= 1e-15 * rad) break;
    }
    int m = 0;
    for (int i = 0; i < n; ++i) {
        if (std::fabs(zi[i]) > 1e-6 * (1.0 + std::fabs(zr[i]))) continue;
        double x = zr[i];
        for (int it = 0; it < 3; ++it) {
            double f = 0.0, d = 0.0;
            for (int k = n; k >= 0; --k) { d = d * x + f; f = f * x + a[k]; }
            if (d == 0.0) break;
            x -= f / d;
        }
        out[m++] = x;
    }
    return m;
}

// rigid motion taking the three object points X onto the camera-frame points P (Kabsch on the centred triples)
inline bool pose_from_three(const double X[3][3], const double P[3][3], Pose& out)
{
    double xc[3] = { 0, 0, 0 }, pc[3] = { 0, 0, 0 };
    for (int i = 0; i < 3; ++i) for (int d = 0; d < 3; ++d) { xc[d] += X[i][d] / 3.0; pc[d] += P[i][d] / 3.0; }
    double H[9] = { 0 };
    for (int i = 0; i < 3; ++i) for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) H[3 * r + c] += (P[i][r] - pc[r]) * (X[i][c] - xc[c]);
    double U[9], S[3], Vt[9];
    la::svd3(H, U, S, Vt);
    if (S[1] <= 1e-12 * S[0]) return false;                 // collinear sample
    double R[9];
    la::mul33(U, Vt, R);
    if (la::det3(R) < 0) { for (int r = 0; r < 3; ++r) U[3 * r + 2] = -U[3 * r + 2]; la::mul33(U, Vt, R); }
    std::copy(R, R + 9, out.R);
    for (int r = 0; r < 3; ++r) out.t[r] = pc[r] - (R[3 * r] * xc[0] + R[3 * r + 1] * xc[1] + R[3 * r + 2] * xc[2]);
    return true;
}

// P3P (Grunert 1841, as in Haralick et al. 1994): three object points X and the unit bearings f of their images.  With the depths
// s1, s2 = u s1, s3 = v s1 the cosine law of the three sides leaves  u = N(v) / D(v)  (difference of two of the equations, linear in u)
// and a quartic in v (that u in the third); its coefficients are formed by polynomial arithmetic here, not copied from a table.
// Up to four poses with positive depths.
inline int p3p(const double X[3][3], const double f[3][3], Pose out[4])
{
    auto d2 = [&](int i, int j) { double s = 0; for (int k = 0; k < 3; ++k) s += (X[i][k] - X[j][k]) * (X[i][k] - X[j][k]); return s; };
    auto dot = [&](int i, int j) { return f[i][0] * f[j][0] + f[i][1] * f[j][1] + f[i][2] * f[j][2]; };
    const double a2 = d2(1, 2), b2 = d2(0, 2), c2 = d2(0, 1);
    const double ca = dot(1, 2), cb = dot(0, 2), cg = dot(0, 1);
    if (a2 <= 0 || b2 <= 0 || c2 <= 0) return 0;
    // g(v) = 1 + v^2 - 2 v cb;  N(v) = b2 (1 - v^2) + (a2 - c2) g(v);  D(v) = 2 b2 (cg - v ca)
    const double g[3] = { 1.0, -2.0 * cb, 1.0 };
    const double N[3] = { b2 + (a2 - c2) * g[0], (a2 - c2) * g[1], -b2 + (a2 - c2) * g[2] };
    const double D[2] = { 2.0 * b2 * cg, -2.0 * b2 * ca };
    // third equation  b2 u^2 - 2 b2 cg u + (b2 - c2 g) = 0  times D^2:  b2 N^2 - 2 b2 cg N D + (b2 - c2 g) D^2 = 0
    double q[5] = { 0, 0, 0, 0, 0 };
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) q[i + j] += b2 * N[i] * N[j];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 2; ++j) q[i + j] -= 2.0 * b2 * cg * N[i] * D[j];
    const double h[3] = { b2 - c2 * g[0], -c2 * g[1], -c2 * g[2] };
    double DD[3] = { D[0] * D[0], 2.0 * D[0] * D[1], D[1] * D[1] };
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) q[i + j] += h[i] * DD[j];
    double roots[4];
    const int nr = real_roots(q, 4, roots);
    int m = 0;
    for (int r = 0; r < nr && m < 4; ++r) {
        const double v = roots[r];
        const double den = D[0] + D[1] * v, gv = g[0] + g[1] * v + g[2] * v * v;
        if (v <= 0 || std::fabs(den) < 1e-12 * b2 || gv <= 0) continue;
        const double u = (N[0] + N[1] * v + N[2] * v * v) / den;
        if (u <= 0) continue;
        const double s1 = std::sqrt(b2 / gv), s[3] = { s1, u * s1, v * s1 };
        double P[3][3];
        for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) P[i][k] = s[i] * f[i][k];
        // the pair (u, v) must also satisfy the first side (the elimination can introduce a spurious root)
        double chk = 0; for (int k = 0; k < 3; ++k) chk += (P[1][k] - P[2][k]) * (P[1][k] - P[2][k]);
        if (std::fabs(chk - a2) > 1e-6 * a2) continue;
        if (pose_from_three(X, P, out[m])) ++m;
    }
    return m;
}

// reprojection residual (pixels) of point i under pose P with K = (fx, fy, cx, cy)
inline bool project_px(const Pose& P, const double K4[4], const double* X, double uv[2])
{
    const double p0 = P.R[0] * X[0] + P.R[1] * X[1] + P.R[2] * X[2] + P.t[0], p1 = P.R[3] * X[0] + P.R[4] * X[1] + P.R[5] * X[2] + P.t[1],
                 p2 = P.R[6] * X[0] + P.R[7] * X[1] + P.R[8] * X[2] + P.t[2];
    uv[0] = K4[0] * p0 / p2 + K4[2]; uv[1] = K4[1] * p1 / p2 + K4[3];
    return p2 > 0.0;
}

// Levenberg-Marquardt on the pixel reprojection error over the listed points; left-multiplicative rotation update
inline void pose_refine(Pose& P, const double K4[4], const std::vector<double>& X, const std::vector<double>& uv, const std::vector<int>& in, int max_it)
{
    auto cost = [&](const Pose& Q) {
        double s = 0;
        for (int i : in) { double p[2]; project_px(Q, K4, &X[3 * (size_t)i], p); const double a = p[0] - uv[2 * (size_t)i], b = p[1] - uv[2 * (size_t)i + 1]; s += a * a + b * b; }
        return s;
    };
    double lambda = 1e-3, c0 = cost(P);
    for (int it = 0; it < max_it; ++it) {
        double H[36] = { 0 }, g[6] = { 0 };
        for (int i : in) {
            const double* Xi = &X[3 * (size_t)i];
            const double r[3] = { P.R[0] * Xi[0] + P.R[1] * Xi[1] + P.R[2] * Xi[2], P.R[3] * Xi[0] + P.R[4] * Xi[1] + P.R[5] * Xi[2], P.R[6] * Xi[0] + P.R[7] * Xi[1] + P.R[8] * Xi[2] };
            const double p[3] = { r[0] + P.t[0], r[1] + P.t[1], r[2] + P.t[2] };
            const double iz = 1.0 / p[2];
            const double e[2] = { K4[0] * p[0] * iz + K4[2] - uv[2 * (size_t)i], K4[1] * p[1] * iz + K4[3] - uv[2 * (size_t)i + 1] };
            // d(u,v)/dp, dp/d(delta) = -[r]x (delta: R <- exp([delta]x) R), dp/dt = I
            const double du[3] = { K4[0] * iz, 0.0, -K4[0] * p[0] * iz * iz }, dv[3] = { 0.0, K4[1] * iz, -K4[1] * p[1] * iz * iz };
            double J[2][6];
            const double* d[2] = { du, dv };
            for (int a = 0; a < 2; ++a) {
                J[a][0] = d[a][1] * (-r[2]) + d[a][2] * r[1];         // column 0 of -[r]x = (0, -r2... ) see below
                J[a][1] = d[a][0] * r[2] + d[a][2] * (-r[0]);
                J[a][2] = d[a][0] * (-r[1]) + d[a][1] * r[0];
                J[a][3] = d[a][0]; J[a][4] = d[a][1]; J[a][5] = d[a][2];
            }
            for (int a = 0; a < 2; ++a)
                for (int j = 0; j < 6; ++j) { g[j] += J[a][j] * e[a]; for (int k = 0; k <= j; ++k) H[j * 6 + k] += J[a][j] * J[a][k]; }
        }
        bool improved = false;
        for (int tries = 0; tries < 8 && !improved; ++tries) {
            double A[36], b[6];
            for (int j = 0; j < 6; ++j) { b[j] = -g[j]; for (int k = 0; k <= j; ++k) A[j * 6 + k] = A[k * 6 + j] = H[j * 6 + k]; A[j * 6 + j] += lambda * (H[j * 6 + j] + 1e-12); }
            if (!la::solve_spd(6, A, b)) { lambda *= 10; continue; }
            Pose Q = P;
            Mat rv(3, 1, CV_64F), dR;
            rv.at<double>(0) = b[0]; rv.at<double>(1) = b[1]; rv.at<double>(2) = b[2];
            Rodrigues_vec(rv, dR);
            la::mul33(dR.ptr<double>(), P.R, Q.R);
            for (int a = 0; a < 3; ++a) Q.t[a] = P.t[a] + b[3 + a];
            const double c1 = cost(Q);
            if (c1 < c0) {
                const bool tiny = c0 - c1 <= 1e-14 * c0;
                P = Q; c0 = c1; lambda = std::max(lambda * 0.1, 1e-12); improved = true;
                if (tiny) return;
            } else lambda *= 10;
        }
        if (!improved) return;
    }
}

}  // namespace detail

// cv::solvePnPRansac(objectPoints, imagePoints, cameraMatrix, noArray(), rvec, tvec) with OpenCV's defaults
// (iterationsCount 100, reprojectionError 8.0, confidence 0.99, SOLVEPNP_ITERATIVE) (NView:1415).  rvec, tvec: 3 x 1 CV_64F.
inline bool solvePnPRansac(const std::vector<Point3f>& object_points, const std::vector<Point2f>& image_points, const Mat& K,
                           Mat& rvec, Mat& tvec, std::vector<int>* inliers_out = nullptr,
                           int iterations = 100, float reproj_error = 8.0f, double confidence = 0.99)
{
    const int n = (int)object_points.size();
    if (n < 4 || image_points.size() != object_points.size()) return false;
    const double K4[4] = { K.ptr<double>()[0], K.ptr<double>()[4], K.ptr<double>()[2], K.ptr<double>()[5] };
    std::vector<double> X(3 * (size_t)n), uv(2 * (size_t)n), xn(2 * (size_t)n);
    for (int i = 0; i < n; ++i) {
        X[3 * i] = object_points[i].x; X[3 * i + 1] = object_points[i].y; X[3 * i + 2] = object_points[i].z;
        uv[2 * i] = image_points[i].x; uv[2 * i + 1] = image_points[i].y;
        xn[2 * i] = (uv[2 * i] - K4[2]) / K4[0]; xn[2 * i + 1] = (uv[2 * i + 1] - K4[3]) / K4[1];
    }
    const double thr2 = (double)reproj_error * reproj_error;
    auto consensus = [&](const detail::Pose& P, std::vector<int>* in) {
        int c = 0;
        if (in) in->clear();
        for (int i = 0; i < n; ++i) {
            double p[2];
            const bool front = detail::project_px(P, K4, &X[3 * (size_t)i], p);
            const double a = p[0] - uv[2 * i], b = p[1] - uv[2 * i + 1];
            if (front && a * a + b * b <= thr2) { ++c; if (in) in->push_back(i); }
        }
        return c;
    };
    detail::Pose best; int best_in = 0;
    // bearings of the image points
    std::vector<double> fb(3 * (size_t)n);
    for (int i = 0; i < n; ++i) {
        const double nrm = std::sqrt(xn[2 * i] * xn[2 * i] + xn[2 * i + 1] * xn[2 * i + 1] + 1.0);
        fb[3 * i] = xn[2 * i] / nrm; fb[3 * i + 1] = xn[2 * i + 1] / nrm; fb[3 * i + 2] = 1.0 / nrm;
    }
    // one hypothesis: P3P on idx[0..2], the root that reprojects idx[3] best
    auto hypothesis = [&](const int idx[4], detail::Pose& P) {
        double Xs[3][3], fs[3][3];
        for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) { Xs[i][k] = X[3 * (size_t)idx[i] + k]; fs[i][k] = fb[3 * (size_t)idx[i] + k]; }
        detail::Pose cand[4];
        const int m = detail::p3p(Xs, fs, cand);
        double best_e = 1e300; bool ok = false;
        for (int c = 0; c < m; ++c) {
            double p[2];
            if (!detail::project_px(cand[c], K4, &X[3 * (size_t)idx[3]], p)) continue;
            const double e = (p[0] - uv[2 * idx[3]]) * (p[0] - uv[2 * idx[3]]) + (p[1] - uv[2 * idx[3] + 1]) * (p[1] - uv[2 * idx[3] + 1]);
            if (e < best_e) { best_e = e; P = cand[c]; ok = true; }
        }
        return ok;
    };
    la::Rng rng(0x2545F4914F6CDD1Dull);
    int iters = iterations;
    for (int it = 0; it < iters; ++it) {
        int idx[4];
        rng.sample(n, 4, idx);
        detail::Pose P;
        if (!hypothesis(idx, P)) continue;
        const int c = consensus(P, nullptr);
        if (c > std::max(best_in, 3)) {
            best_in = c; best = P;
            iters = std::min(iters, std::max(it + 1, la::ransac_update_iters(confidence, (double)(n - c) / n, 4, iters)));
        }
    }
    if (best_in < 4) return false;
    std::vector<int> in;
    consensus(best, &in);
    // SOLVEPNP_ITERATIVE on the inliers: linear start + Levenberg-Marquardt on the pixel error [3P]; the consensus set is
    // re-taken once with the refined pose
    for (int pass = 0; pass < 2; ++pass) {
        detail::Pose P = best, Pl;
        if (in.size() >= 6 && detail::pose_dlt(X, xn, in.data(), (int)in.size(), Pl) && consensus(Pl, nullptr) >= best_in) P = Pl;
        detail::pose_refine(P, K4, X, uv, in, 30);
        std::vector<int> in2;
        const int c = consensus(P, &in2);
        if (c < best_in && pass > 0) break;
        best = P;
        if (c >= best_in) { best_in = c; in.swap(in2); }
    }
    Mat Rm(3, 3, CV_64F);
    std::copy(best.R, best.R + 9, Rm.ptr<double>());
    Rodrigues(Rm, rvec);
    tvec = Mat(3, 1, CV_64F);
    for (int a = 0; a < 3; ++a) tvec.at<double>(a) = best.t[a];
    if (inliers_out) *inliers_out = in;
    return true;
}

}  // namespace sfm
