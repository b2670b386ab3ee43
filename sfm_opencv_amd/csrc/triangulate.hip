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
#include <cmath>
#include <vector>

// OpenCV's stopping rule for double input (JacobiSVDImpl_ [3P]: eps = 10 DBL_EPSILON).  Rounds 1-2 used 1e-16, below the rounding
// noise of the dot product: most systems then ran all 30 sweeps (300k points: 0.32 ms; with this rule: see profiles/README.md).
#define SVD_EPS (10.0 * 2.220446049250313e-16)

struct ProjPair { float p1[12]; float p2[12]; };

__device__ __forceinline__ void jacobi_rot(double (&A)[4][4], double (&V)[4][4], const int p, const int q, bool& rotated)
{
    double a = 0.0, b = 0.0, g = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a += A[i][p] * A[i][p]; b += A[i][q] * A[i][q]; g += A[i][p] * A[i][q]; }
    if (fabs(g) <= SVD_EPS * sqrt(a * b) || g == 0.0) return;
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

// ------------------------------------------------------------------------------------------------
// N-view extension (not reference behaviour): multi-view DLT of every track on normalised image
// coordinates, and per-observation reprojection errors.  Thread per point walks its observation list (CSR by point);
// the 4x4 moment matrix M = A'A stays in registers and goes through the same Jacobi routine as the two-view system.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void triangulate_tracks_kernel(const double* __restrict__ Rt, double fx, double fy, double cx, double cy,
                                                                 const int* __restrict__ pt_start, const int* __restrict__ ocam,
                                                                 const double* __restrict__ ouv, int n_pt,
                                                                 double* __restrict__ pts, int* __restrict__ n_views)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pt) return;
    const int s0 = pt_start[p], s1 = pt_start[p + 1];
    double A[4][4], V[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { A[i][j] = 0.0; V[i][j] = (i == j) ? 1.0 : 0.0; }
    for (int q = s0; q < s1; ++q) {
        const double* P = Rt + 12 * (size_t)ocam[q];
        const double xn = (ouv[2 * q] - cx) / fx, yn = (ouv[2 * q + 1] - cy) / fy;
        double r0[4], r1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { r0[j] = xn * P[8 + j] - P[j]; r1[j] = yn * P[8 + j] - P[4 + j]; }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) A[i][j] += r0[i] * r0[j] + r1[i] * r1[j];
    }
    if (n_views) n_views[p] = s1 - s0;
    if (s1 - s0 < 2) { const double nan = __longlong_as_double(0x7ff8000000000000LL); pts[3 * p] = pts[3 * p + 1] = pts[3 * p + 2] = nan; return; }
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
        jacobi_rot(A, V, 0, 1, rotated); jacobi_rot(A, V, 0, 2, rotated); jacobi_rot(A, V, 0, 3, rotated);
        jacobi_rot(A, V, 1, 2, rotated); jacobi_rot(A, V, 1, 3, rotated); jacobi_rot(A, V, 2, 3, rotated);
        if (!rotated) break;
    }
    double best = 0.0, v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 1.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double nn = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) nn += A[i][j] * A[i][j];
        if (j == 0 || nn < best) { best = nn; v0 = V[0][j]; v1 = V[1][j]; v2 = V[2][j]; v3 = V[3][j]; }
    }
    pts[3 * p] = v0 / v3; pts[3 * p + 1] = v1 / v3; pts[3 * p + 2] = v2 / v3;
}

__global__ __launch_bounds__(256) void reprojection_error_kernel(const double* __restrict__ Rt, double fx, double fy, double cx, double cy,
                                                                 const double* __restrict__ pts, const int* __restrict__ ocam,
                                                                 const int* __restrict__ opt, const double* __restrict__ ouv, int n_obs,
                                                                 double* __restrict__ err)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_obs) return;
    const double* P = Rt + 12 * (size_t)ocam[k]; const double* X = pts + 3 * (size_t)opt[k];
    const double x = P[0] * X[0] + P[1] * X[1] + P[2] * X[2] + P[3], y = P[4] * X[0] + P[5] * X[1] + P[6] * X[2] + P[7];
    const double z = P[8] * X[0] + P[9] * X[1] + P[10] * X[2] + P[11];
    const double du = fx * x / z + cx - ouv[2 * k], dv = fy * y / z + cy - ouv[2 * k + 1];
    err[k] = sqrt(du * du + dv * dv);
}

// angle-axis + translation -> [R | t] (row-major 3x4), the rotation formula of the BA cost (NView:151-183)
static void angle_axis_to_Rt(const double* e, double Rt[12])
{
    const double th2 = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
    double R[9];
    if (th2 > 2.220446049250313e-16) {
        const double th = std::sqrt(th2), c = std::cos(th), s = std::sin(th), wx = e[0] / th, wy = e[1] / th, wz = e[2] / th, k = 1.0 - c;
        R[0] = c + wx * wx * k;      R[1] = wx * wy * k - wz * s; R[2] = wx * wz * k + wy * s;
        R[3] = wy * wx * k + wz * s; R[4] = c + wy * wy * k;      R[5] = wy * wz * k - wx * s;
        R[6] = wz * wx * k - wy * s; R[7] = wz * wy * k + wx * s; R[8] = c + wz * wz * k;
    } else {
        R[0] = 1; R[1] = -e[2]; R[2] = e[1]; R[3] = e[2]; R[4] = 1; R[5] = -e[0]; R[6] = -e[1]; R[7] = e[0]; R[8] = 1;
    }
    for (int r = 0; r < 3; ++r) { for (int c2 = 0; c2 < 3; ++c2) Rt[4 * r + c2] = R[3 * r + c2]; Rt[4 * r + 3] = e[3 + r]; }
}

extern "C" {

int sfmhip_triangulate2_f32_dev(sfmhip_ctx* ctx, const float P1[12], const float P2[12],
                                const float* d_xy1, const float* d_xy2, int n, float* d_xyzw, double* d_xyz)
{
    SFM_DEVICE_GUARD(ctx);
    SFM_RANGE("sfmhip_triangulate2_f32_dev");
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
    SFM_DEVICE_GUARD(ctx);
    SFM_RANGE("sfmhip_triangulate2_matches_dev");
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
    SFM_DEVICE_GUARD(ctx);
    SFM_RANGE("sfmhip_triangulate2_f32");
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


int sfmhip_triangulate_tracks(sfmhip_ctx* ctx, const double K4[4], const double* ext6, int n_cam,
                              const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_uv, int n_obs, int n_pt,
                              double* pts_out, int32_t* n_views_out)
{
    SFM_DEVICE_GUARD(ctx);
    SFM_RANGE("sfmhip_triangulate_tracks");
    SFM_ARG_CHECK(ctx, ctx && K4 && ext6 && n_cam > 0 && n_pt >= 0 && n_obs >= 0 && (pts_out || n_pt == 0));
    SFM_ARG_CHECK(ctx, (obs_cam && obs_pt && obs_uv) || n_obs == 0);
    for (int k = 0; k < n_obs; ++k) SFM_ARG_CHECK(ctx, obs_cam[k] >= 0 && obs_cam[k] < n_cam && obs_pt[k] >= 0 && obs_pt[k] < n_pt);
    if (n_pt == 0) return SFMHIP_OK;
    // CSR by point, observations of a point in the caller's order (the accumulation order of M)
    std::vector<int> pt_start(n_pt + 1, 0), fill(n_pt, 0), ocam(n_obs > 0 ? n_obs : 1);
    std::vector<double> ouv(2 * (size_t)(n_obs > 0 ? n_obs : 1)), Rt(12 * (size_t)n_cam);
    for (int k = 0; k < n_obs; ++k) pt_start[obs_pt[k] + 1]++;
    for (int p = 0; p < n_pt; ++p) pt_start[p + 1] += pt_start[p];
    for (int k = 0; k < n_obs; ++k) {
        const int q = pt_start[obs_pt[k]] + fill[obs_pt[k]]++;
        ocam[q] = obs_cam[k]; ouv[2 * (size_t)q] = obs_uv[2 * (size_t)k]; ouv[2 * (size_t)q + 1] = obs_uv[2 * (size_t)k + 1];
    }
    for (int c = 0; c < n_cam; ++c) angle_axis_to_Rt(ext6 + 6 * c, Rt.data() + 12 * c);
    const size_t b_rt = (Rt.size() * 8 + 255) / 256 * 256, b_st = (pt_start.size() * 4 + 255) / 256 * 256, b_oc = (ocam.size() * 4 + 255) / 256 * 256;
    const size_t b_uv = (ouv.size() * 8 + 255) / 256 * 256, b_pt = ((size_t)n_pt * 24 + 255) / 256 * 256, b_nv = ((size_t)n_pt * 4 + 255) / 256 * 256;
    void* base = nullptr;
    int rc = sfm_scratch(ctx, b_rt + b_st + b_oc + b_uv + b_pt + b_nv, &base); if (rc) return rc;
    char* c0 = (char*)base;
    double* d_rt = (double*)c0; int* d_st = (int*)(c0 + b_rt); int* d_oc = (int*)(c0 + b_rt + b_st); double* d_uv = (double*)(c0 + b_rt + b_st + b_oc);
    double* d_pt = (double*)(c0 + b_rt + b_st + b_oc + b_uv); int* d_nv = (int*)(c0 + b_rt + b_st + b_oc + b_uv + b_pt);
    SFM_HIP_TRY(ctx, hipMemcpyAsync(d_rt, Rt.data(), Rt.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    SFM_HIP_TRY(ctx, hipMemcpyAsync(d_st, pt_start.data(), pt_start.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    SFM_HIP_TRY(ctx, hipMemcpyAsync(d_oc, ocam.data(), ocam.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    SFM_HIP_TRY(ctx, hipMemcpyAsync(d_uv, ouv.data(), ouv.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(triangulate_tracks_kernel, dim3((n_pt + 255) / 256), dim3(256), 0, ctx->stream, d_rt, K4[0], K4[1], K4[2], K4[3],
                       d_st, d_oc, d_uv, n_pt, d_pt, d_nv);
    SFM_HIP_TRY(ctx, hipGetLastError());
    SFM_HIP_TRY(ctx, hipMemcpyAsync(pts_out, d_pt, (size_t)n_pt * 24, hipMemcpyDeviceToHost, ctx->stream));
    if (n_views_out) SFM_HIP_TRY(ctx, hipMemcpyAsync(n_views_out, d_nv, (size_t)n_pt * 4, hipMemcpyDeviceToHost, ctx->stream));
    SFM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SFMHIP_OK;
}

int sfmhip_reprojection_errors(sfmhip_ctx* ctx, const double K4[4], const double* ext6, int n_cam, const double* pts, int n_pt,
                               const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_uv, int n_obs, double* err_out)
{
    SFM_DEVICE_GUARD(ctx);
    SFM_RANGE("sfmhip_reprojection_errors");
    SFM_ARG_CHECK(ctx, ctx && K4 && ext6 && n_cam > 0 && n_pt >= 0 && n_obs >= 0 && (err_out || n_obs == 0));
    SFM_ARG_CHECK(ctx, n_obs == 0 || (pts && obs_cam && obs_pt && obs_uv));
    for (int k = 0; k < n_obs; ++k) SFM_ARG_CHECK(ctx, obs_cam[k] >= 0 && obs_cam[k] < n_cam && obs_pt[k] >= 0 && obs_pt[k] < n_pt);
    if (n_obs == 0) return SFMHIP_OK;
    std::vector<double> Rt(12 * (size_t)n_cam);
    for (int c = 0; c < n_cam; ++c) angle_axis_to_Rt(ext6 + 6 * c, Rt.data() + 12 * c);
    const size_t b_rt = (Rt.size() * 8 + 255) / 256 * 256, b_pt = ((size_t)n_pt * 24 + 255) / 256 * 256, b_i = ((size_t)n_obs * 4 + 255) / 256 * 256;
    const size_t b_uv = ((size_t)n_obs * 16 + 255) / 256 * 256, b_e = ((size_t)n_obs * 8 + 255) / 256 * 256;
    void* base = nullptr;
    int rc = sfm_scratch(ctx, b_rt + b_pt + 2 * b_i + b_uv + b_e, &base); if (rc) return rc;
    char* c0 = (char*)base;
    double* d_rt = (double*)c0; double* d_pt = (double*)(c0 + b_rt); int* d_oc = (int*)(c0 + b_rt + b_pt); int* d_op = (int*)(c0 + b_rt + b_pt + b_i);
    double* d_uv = (double*)(c0 + b_rt + b_pt + 2 * b_i); double* d_e = (double*)(c0 + b_rt + b_pt + 2 * b_i + b_uv);
    SFM_HIP_TRY(ctx, hipMemcpyAsync(d_rt, Rt.data(), Rt.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    SFM_HIP_TRY(ctx, hipMemcpyAsync(d_pt, pts, (size_t)n_pt * 24, hipMemcpyHostToDevice, ctx->stream));
    SFM_HIP_TRY(ctx, hipMemcpyAsync(d_oc, obs_cam, (size_t)n_obs * 4, hipMemcpyHostToDevice, ctx->stream));
    SFM_HIP_TRY(ctx, hipMemcpyAsync(d_op, obs_pt, (size_t)n_obs * 4, hipMemcpyHostToDevice, ctx->stream));
    SFM_HIP_TRY(ctx, hipMemcpyAsync(d_uv, obs_uv, (size_t)n_obs * 16, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(reprojection_error_kernel, dim3((n_obs + 255) / 256), dim3(256), 0, ctx->stream, d_rt, K4[0], K4[1], K4[2], K4[3],
                       d_pt, d_oc, d_op, d_uv, n_obs, d_e);
    SFM_HIP_TRY(ctx, hipGetLastError());
    SFM_HIP_TRY(ctx, hipMemcpyAsync(err_out, d_e, (size_t)n_obs * 8, hipMemcpyDeviceToHost, ctx->stream));
    SFM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SFMHIP_OK;
}

}  // extern "C"
