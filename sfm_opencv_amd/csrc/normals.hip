// normals.hip -- point-cloud normals for the .ply writer ("next" row 8f-1; replaces estimate_normals
// NViewReconstuct.cpp:551-599 and PCAFitPlane 601-690).
//
// The reference pushes ALL other points into a priority_queue per point (O(N^2 log N) on the host) and pops the
// K nearest; here one thread per point streams the cloud through LDS tiles and keeps a sorted top-16 in
// registers (ordering: distance, then index -- any order among equal distances is a valid K-set for the
// reference).  Then the plane fit of PCAFitPlane: mean of the K neighbours, covariance / K, eigenvector of the
// smallest eigenvalue (cyclic Jacobi instead of Eigen::EigenSolver), flipped when n . mean > 0 (NView:672),
// normalised.  fp64 throughout, no FMA contraction so distances compare exactly like the host's.
#include "common.hpp"
#pragma clang fp contract(off)

#define KMAX 16
#define NTILE 256

__device__ __forceinline__ void eig3_min(const double Cin[9], double v[3])
{
    double A[3][3], V[3][3] = { { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) A[i][j] = Cin[3 * i + j];
    for (int sweep = 0; sweep < 50; ++sweep) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off == 0.0) break;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;      // (0,1) (0,2) (1,2)
            if (A[p][q] == 0.0) continue;
            const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
            for (int k = 0; k < 3; ++k) { const double x = A[k][p], y = A[k][q]; A[k][p] = c * x - s * y; A[k][q] = s * x + c * y; }
#pragma unroll
            for (int k = 0; k < 3; ++k) { const double x = A[p][k], y = A[q][k]; A[p][k] = c * x - s * y; A[q][k] = s * x + c * y; }
#pragma unroll
            for (int k = 0; k < 3; ++k) { const double x = V[k][p], y = V[k][q]; V[k][p] = c * x - s * y; V[k][q] = s * x + c * y; }
        }
    }
    v[0] = V[0][0]; v[1] = V[1][0]; v[2] = V[2][0];
    double best = A[0][0];
    if (A[1][1] < best) { best = A[1][1]; v[0] = V[0][1]; v[1] = V[1][1]; v[2] = V[2][1]; }
    if (A[2][2] < best) { best = A[2][2]; v[0] = V[0][2]; v[1] = V[1][2]; v[2] = V[2][2]; }
}

__global__ __launch_bounds__(NTILE) void normals_kernel(const double* __restrict__ pts, int n, int K, double* __restrict__ normals)
{
    __shared__ double tx[NTILE], ty[NTILE], tz[NTILE];
    const int i = blockIdx.x * NTILE + threadIdx.x;
    const bool active = i < n;
    const double px = active ? pts[3 * (size_t)i] : 0.0, py = active ? pts[3 * (size_t)i + 1] : 0.0, pz = active ? pts[3 * (size_t)i + 2] : 0.0;
    double bd[KMAX]; int bi[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) { bd[k] = INFINITY; bi[k] = -1; }
    // Squared-distance gate in front of the exact test: sqrt(d2) < bd[last] is impossible once d2 > bd[last]^2 (1 + 2^-50) (the real
    // root then exceeds bd[last], and rounding is monotone), so the fp64 square root -- two thirds of the instructions of a candidate --
    // is taken only by candidates that can still enter the list.  Same K-sets, same order, bit for bit (round 3: 113 -> see profiles/).
    double gate2 = INFINITY;
    for (int base = 0; base < n; base += NTILE) {
        const int j0 = base + threadIdx.x;
        __syncthreads();
        if (j0 < n) { tx[threadIdx.x] = pts[3 * (size_t)j0]; ty[threadIdx.x] = pts[3 * (size_t)j0 + 1]; tz[threadIdx.x] = pts[3 * (size_t)j0 + 2]; }
        __syncthreads();
        const int cnt = n - base < NTILE ? n - base : NTILE;
        for (int t = 0; t < cnt; ++t) {
            const int j = base + t;
            const double dx = px - tx[t], dy = py - ty[t], dz = pz - tz[t];
            const double d2 = dx * dx + dy * dy + dz * dz;
            if (!(d2 <= gate2)) continue;
            const double d = sqrt(d2);
            if (j != i && d < bd[KMAX - 1]) {
                bd[KMAX - 1] = d; bi[KMAX - 1] = j;
#pragma unroll
                for (int k = KMAX - 1; k > 0; --k) {
                    if (bd[k] < bd[k - 1]) {
                        const double td = bd[k]; bd[k] = bd[k - 1]; bd[k - 1] = td;
                        const int ti = bi[k]; bi[k] = bi[k - 1]; bi[k - 1] = ti;
                    }
                }
                gate2 = bd[KMAX - 1] * bd[KMAX - 1] * (1.0 + 0x1p-50);       // inf while the list is not full
            }
        }
    }
    if (!active) return;
    int cnt = 0;
    double mean[3] = { 0, 0, 0 };
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        if (k < K && bi[k] >= 0) {
            mean[0] += pts[3 * (size_t)bi[k]]; mean[1] += pts[3 * (size_t)bi[k] + 1]; mean[2] += pts[3 * (size_t)bi[k] + 2];
            ++cnt;
        }
    if (cnt == 0) { normals[3 * (size_t)i] = NAN; normals[3 * (size_t)i + 1] = NAN; normals[3 * (size_t)i + 2] = NAN; return; }
    mean[0] /= (double)cnt; mean[1] /= (double)cnt; mean[2] /= (double)cnt;
    double C[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        if (k < K && bi[k] >= 0) {
            const double d[3] = { pts[3 * (size_t)bi[k]] - mean[0], pts[3 * (size_t)bi[k] + 1] - mean[1], pts[3 * (size_t)bi[k] + 2] - mean[2] };
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) C[3 * a + b] += d[a] * d[b];
        }
#pragma unroll
    for (int a = 0; a < 9; ++a) C[a] /= (double)cnt;
    double v[3];
    eig3_min(C, v);
    if (v[0] * mean[0] + v[1] * mean[1] + v[2] * mean[2] > 0.0) { v[0] = -v[0]; v[1] = -v[1]; v[2] = -v[2]; }
    const double nn = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    normals[3 * (size_t)i] = v[0] / nn; normals[3 * (size_t)i + 1] = v[1] / nn; normals[3 * (size_t)i + 2] = v[2] / nn;
}

extern "C" int sfmhip_estimate_normals(sfmhip_ctx* ctx, const double* pts, int n, int K, double* normals)
{
    SFM_DEVICE_GUARD(ctx);
    SFM_RANGE("sfmhip_estimate_normals");
    SFM_ARG_CHECK(ctx, ctx && n >= 0 && K >= 1 && K <= KMAX);
    if (n == 0) return SFMHIP_OK;
    SFM_ARG_CHECK(ctx, pts && normals);
    double *d_p = nullptr, *d_n = nullptr;
    SFM_HIP_TRY(ctx, hipMalloc((void**)&d_p, (size_t)n * 24));
    hipError_t e = hipMalloc((void**)&d_n, (size_t)n * 24);
    if (e == hipSuccess) e = hipMemcpyAsync(d_p, pts, (size_t)n * 24, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(normals_kernel, dim3(ceil_div(n, NTILE)), dim3(NTILE), 0, ctx->stream, d_p, n, K, d_n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(normals, d_n, (size_t)n * 24, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_p); (void)hipFree(d_n);
    if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); return SFMHIP_E_HIP; }
    return SFMHIP_OK;
}
