// triangulate.hip -- two-view DLT triangulation (replaces cv::triangulatePoints + the float32
// de-homogenisation loop of reconstruct(), NViewReconstuct.cpp:1147-1156).
//
// Per correspondence (one thread): the 4x4 system of cvTriangulatePoints [3P]
//     A[2j+0][k] = x_j P_j[2][k] - P_j[0][k],  A[2j+1][k] = y_j P_j[2][k] - P_j[1][k]   (double, from float32 inputs)
// is reduced by a one-sided (Hestenes) Jacobi SVD in fp64 held entirely in registers; the right singular
// vector of the smallest singular value is cast to float32 (the type of pts4d), divided by w the way
// `Mat_<float> /= w` does (multiply by float(1.0/double(w))) and widened to double (Point3f -> Point3d).
// HBM traffic: 16 B in + 16 B (xyzw) + 24 B (xyz) out per correspondence; the optional fused gather reads
// the two keypoints through the match list instead (get_matched_points, NViewReconstuct.cpp:989-1003).
#include "common.hpp"
// float32 results must be bit-exact with the CPU restatement: no FMA contraction, correctly rounded sqrt
// (HIP's __fsqrt_rn/__fmul_rn are NOT the rounded forms on this toolchain: native sqrt / contractible mul).
#pragma clang fp contract(off)

struct ProjPair { float p1[12]; float p2[12]; };

__device__ __forceinline__ void jacobi_rot(double (&A)[4][4], double (&V)[4][4], const int p, const int q, bool& rotated)
{
    double a = 0.0, b = 0.0, g = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a += A[i][p] * A[i][p]; b += A[i][q] * A[i][q]; g += A[i][p] * A[i][q]; }
    if (fabs(g) <= 1e-16 * sqrt(a * b) || g == 0.0) return;
    rotated = true;
    const double zeta = (b - a) / (2.0 * g);
    const double tt = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    const double c = 1.0 / sqrt(1.0 + tt * tt), s = c * tt;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double x = A[i][p], y = A[i][q];
        A[i][p] = c * x - s * y; A[i][q] = s * x + c * y;
        x = V[i][p]; y = V[i][q];
        V[i][p] = c * x - s * y; V[i][q] = s * x + c * y;
    }
}

__device__ __forceinline__ void triangulate_one(const ProjPair& P, float x1, float y1, float x2, float y2, float h[4])
{
    double A[4][4], V[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        A[0][k] = (double)x1 * (double)P.p1[8 + k] - (double)P.p1[k];
        A[1][k] = (double)y1 * (double)P.p1[8 + k] - (double)P.p1[4 + k];
        A[2][k] = (double)x2 * (double)P.p2[8 + k] - (double)P.p2[k];
        A[3][k] = (double)y2 * (double)P.p2[8 + k] - (double)P.p2[4 + k];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
        jacobi_rot(A, V, 0, 1, rotated); jacobi_rot(A, V, 0, 2, rotated); jacobi_rot(A, V, 0, 3, rotated);
        jacobi_rot(A, V, 1, 2, rotated); jacobi_rot(A, V, 1, 3, rotated); jacobi_rot(A, V, 2, 3, rotated);
        if (!rotated) break;
    }
    double nn[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { nn[j] = 0.0; for (int i = 0; i < 4; ++i) nn[j] += A[i][j] * A[i][j]; }
    double best = nn[0];
    double v0 = V[0][0], v1 = V[1][0], v2 = V[2][0], v3 = V[3][0];
#pragma unroll
    for (int j = 1; j < 4; ++j)
        if (nn[j] < best) { best = nn[j]; v0 = V[0][j]; v1 = V[1][j]; v2 = V[2][j]; v3 = V[3][j]; }
    h[0] = (float)v0; h[1] = (float)v1; h[2] = (float)v2; h[3] = (float)v3;
}

__device__ __forceinline__ void store_point(const float h[4], int i, int n, float* __restrict__ xyzw, double* __restrict__ xyz)
{
    if (xyzw) {
        xyzw[i] = h[0]; xyzw[(size_t)n + i] = h[1]; xyzw[2 * (size_t)n + i] = h[2]; xyzw[3 * (size_t)n + i] = h[3];
    }
    if (xyz) {
        const float inv = (float)(1.0 / (double)h[3]);
        xyz[3 * (size_t)i + 0] = (double)(h[0] * inv);
        xyz[3 * (size_t)i + 1] = (double)(h[1] * inv);
        xyz[3 * (size_t)i + 2] = (double)(h[2] * inv);
    }
}

__global__ __launch_bounds__(256) void triangulate2_kernel(ProjPair P, const float2* __restrict__ xy1, const float2* __restrict__ xy2,
                                                           int n, float* __restrict__ xyzw, double* __restrict__ xyz)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float2 a = xy1[i], b = xy2[i];
    float h[4];
    triangulate_one(P, a.x, a.y, b.x, b.y, h);
    store_point(h, i, n, xyzw, xyz);
}

__global__ __launch_bounds__(256) void triangulate2_matches_kernel(ProjPair P, const sfm_keypoint* __restrict__ kp1,
                                                                   const sfm_keypoint* __restrict__ kp2,
                                                                   const sfm_dmatch* __restrict__ m, int n,
                                                                   float* __restrict__ xyzw, double* __restrict__ xyz)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const sfm_dmatch mm = m[i];
    const sfm_keypoint a = kp1[mm.queryIdx], b = kp2[mm.trainIdx];
    float h[4];
    triangulate_one(P, a.x, a.y, b.x, b.y, h);
    store_point(h, i, n, xyzw, xyz);
}

extern "C" {

int sfmhip_triangulate2_f32_dev(sfmhip_ctx* ctx, const float P1[12], const float P2[12],
                                const float* d_xy1, const float* d_xy2, int n, float* d_xyzw, double* d_xyz)
{
    SFM_ARG_CHECK(ctx, ctx && P1 && P2 && n >= 0);
    if (n == 0) return SFMHIP_OK;
    SFM_ARG_CHECK(ctx, d_xy1 && d_xy2 && (d_xyzw || d_xyz));
    ProjPair P; memcpy(P.p1, P1, sizeof P.p1); memcpy(P.p2, P2, sizeof P.p2);
    hipLaunchKernelGGL(triangulate2_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, P,
                       (const float2*)d_xy1, (const float2*)d_xy2, n, d_xyzw, d_xyz);
    SFM_HIP_TRY(ctx, hipGetLastError());
    return SFMHIP_OK;
}

int sfmhip_triangulate2_matches_dev(sfmhip_ctx* ctx, const float P1[12], const float P2[12],
                                    const sfm_keypoint* d_kp1, const sfm_keypoint* d_kp2,
                                    const sfm_dmatch* d_matches, int n, float* d_xyzw, double* d_xyz)
{
    SFM_ARG_CHECK(ctx, ctx && P1 && P2 && n >= 0);
    if (n == 0) return SFMHIP_OK;
    SFM_ARG_CHECK(ctx, d_kp1 && d_kp2 && d_matches && (d_xyzw || d_xyz));
    ProjPair P; memcpy(P.p1, P1, sizeof P.p1); memcpy(P.p2, P2, sizeof P.p2);
    hipLaunchKernelGGL(triangulate2_matches_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, P,
                       d_kp1, d_kp2, d_matches, n, d_xyzw, d_xyz);
    SFM_HIP_TRY(ctx, hipGetLastError());
    return SFMHIP_OK;
}

// reconstruct()'s arithmetic core on host buffers.  Empty input is an argument error, like the
// reference's "[Err]: empty 2d points." / -1 (NViewReconstuct.cpp:1122-1126).
int sfmhip_triangulate2_f32(sfmhip_ctx* ctx, const float P1[12], const float P2[12],
                            const float* xy1, const float* xy2, int n, float* xyzw, double* xyz)
{
    SFM_ARG_CHECK(ctx, ctx && P1 && P2 && xy1 && xy2 && n > 0 && (xyzw || xyz));
    float *d1 = nullptr, *d2 = nullptr, *dw = nullptr; double* dx = nullptr;
    int rc = SFMHIP_OK;
    hipError_t e = hipMalloc((void**)&d1, (size_t)n * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&d2, (size_t)n * 8);
    if (e == hipSuccess && xyzw) e = hipMalloc((void**)&dw, (size_t)n * 16);
    if (e == hipSuccess && xyz) e = hipMalloc((void**)&dx, (size_t)n * 24);
    if (e == hipSuccess) e = hipMemcpyAsync(d1, xy1, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d2, xy2, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) rc = sfmhip_triangulate2_f32_dev(ctx, P1, P2, d1, d2, n, dw, dx);
    if (e == hipSuccess && rc == SFMHIP_OK && xyzw) e = hipMemcpyAsync(xyzw, dw, (size_t)n * 16, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && rc == SFMHIP_OK && xyz) e = hipMemcpyAsync(xyz, dx, (size_t)n * 24, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); rc = SFMHIP_E_HIP; }
    (void)hipFree(d1); (void)hipFree(d2); (void)hipFree(dw); (void)hipFree(dx);
    return rc;
}

}  // extern "C"
