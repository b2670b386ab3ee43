// ba_kernels.hpp -- device code of the bundle-adjustment path (included by ba.hip only).
//
// Replaces what ceres::Solve does for bundle_adjustment() (NViewReconstuct.cpp:1162-1244):
//   residual + analytic Jacobian of ReprojectCost (NView:151-183; ceres::AngleAxisRotatePoint incl. its
//   small-angle branch), HuberLoss corrector, Jacobi column scaling, the normal-equation blocks, Schur
//   elimination of every point block into the reduced camera system S (cameras 6 each + the shared 4
//   intrinsics), damping, and the back-substitution / model-cost / candidate-cost pass.
//
// Design: NO Jacobian is ever stored.  An observation record is 24 B (camera id, u, v; its point is implied by
// the per-point ordering) while its corrected 2x13 Jacobian would be 208 B; re-deriving it costs ~300 fp64
// flops, i.e. less than the HBM time of reading it back (ridge ~10 flop/B).  Every kernel below therefore
// recomputes the Jacobian rows it needs from (K, camera, point, observation).
//
// All reductions are deterministic (fixed-shape tree/shuffle reductions, single writer per output): no atomics.
#pragma once
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdint>

struct BADev {
    // sizes / layout
    int nc, np, nobs, n, npad, koff, fix0, fixK, cam_split, world, rank;
    double huber_a;
    // parameters: current and candidate
    const double* K; const double* ext; const double* pts;
    double* Kc; double* extc; double* ptsc;
    // observations ordered by point
    const int* pt_start; const int* ocam; const double* ouv;
    // per-camera lists (indices into the by-point ordering) and their points
    const int* blk_crange;   // per block of 256 points (storage order): lowest and highest camera among their observations
    const int* cam_start; const int* cam_pt; const double* cam_uv; const int* opt;   // cam_pt / cam_uv: point slot and pixel of each observation, in camera order
    // layout of the reduced system: position of camera c's 6 columns (-1 = constant camera), koff = intrinsics;
    // posmask[i] = 1 for a real parameter, 0 for a padding slot (segments are padded to whole 32-blocks)
    const int* cam_pos; const int* posmask;
    // per-camera rotation block (CAMPRE doubles: R, the three derivative vectors c_m, branch flag -- see campre_one), current / candidate
    const double* campre; const double* campre_c;
    // column scaling (cam side: npad entries; points: 3 np)
    const double* scale_c; const double* scale_p;
    // per point
    double* Vinv; double* bp; double* WK; double* colsq_p;
    // reduced system message: S (npad x npad) | rhs (npad) | diagU (npad) | graw (npad) | scal
    double* S; double* rhs; double* diagU; double* graw; double* scal;
    // partials
    double* part_pt;    // [pt blocks][32]
    double* part_cam;   // [nc][cam_split][80]
    double* part_back;  // [pt blocks][4]
    // solution of the reduced system (scaled coordinates, y; step = -y)
    const double* y;
    double radius, min_diag, max_diag;
    int xcd_plain;        // measurement knob (SFMHIP_EXP_XCD_PLAIN): 1 = work item = workgroup index, no XCD-aware order
};

#define SCAL_COST 0
#define SCAL_GMAX_SLOTS 8     // scal[8 + rank] = local max |gradient| over this rank's points

__device__ __forceinline__ int cam_off(const BADev& P, int c) { return P.cam_pos[c]; }

// Corrected residual and Jacobian blocks of one observation, column-scaled; blocks of constant parameters are 0.
struct ObsLin {
    double r[2];
    double EK[2][4];   // d r / d (fx fy cx cy)
    double Ec[2][6];   // d r / d (angle-axis, t)
    double F[2][3];    // d r / d X
    double rho0;       // rho(|r|^2) (cost = 1/2 rho0)
};

// R(omega) X and its derivatives d(R X)/d omega_m for one vector X: ceres::AngleAxisRotatePoint [3P] including the
// theta^2 <= DBL_EPSILON first-order branch.  Only ba_campre_kernel calls this (on the three unit vectors): the
// observation kernels read the per-camera matrices instead of redoing sincos + the derivative chain per observation.
__device__ __forceinline__ void rotate_with_derivs(const double* __restrict__ e, const double X[3], double p[3], double dpw[3][3])
{
    const double th2 = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
    if (th2 > DBL_EPSILON) {
        const double th = sqrt(th2);
        double s, c; sincos(th, &s, &c);
        const double inv = 1.0 / th;
        const double w[3] = { e[0] * inv, e[1] * inv, e[2] * inv };
        const double wx[3] = { w[1] * X[2] - w[2] * X[1], w[2] * X[0] - w[0] * X[2], w[0] * X[1] - w[1] * X[0] };
        const double dot = w[0] * X[0] + w[1] * X[1] + w[2] * X[2];
        const double omc = 1.0 - c, tmp = dot * omc;
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] = X[k] * c + wx[k] * s + w[k] * tmp;
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            double dw[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) dw[k] = ((k == m ? 1.0 : 0.0) - w[k] * w[m]) * inv;
            const double dc = -s * w[m], ds = c * w[m];
            const double dwx[3] = { dw[1] * X[2] - dw[2] * X[1], dw[2] * X[0] - dw[0] * X[2], dw[0] * X[1] - dw[1] * X[0] };
            const double ddot = dw[0] * X[0] + dw[1] * X[1] + dw[2] * X[2];
            const double dtmp = ddot * omc + dot * s * w[m];
#pragma unroll
            for (int k = 0; k < 3; ++k)
                dpw[m][k] = X[k] * dc + dwx[k] * s + wx[k] * ds + dw[k] * tmp + w[k] * dtmp;
        }
    } else {
        p[0] = X[0] + (e[1] * X[2] - e[2] * X[1]);
        p[1] = X[1] + (e[2] * X[0] - e[0] * X[2]);
        p[2] = X[2] + (e[0] * X[1] - e[1] * X[0]);
        // d(omega x X)/d omega_m = e_m x X
        dpw[0][0] = 0.0;   dpw[0][1] = -X[2]; dpw[0][2] = X[1];
        dpw[1][0] = X[2];  dpw[1][1] = 0.0;   dpw[1][2] = -X[0];
        dpw[2][0] = -X[1]; dpw[2][1] = X[0];  dpw[2][2] = 0.0;
    }
}

// Per-camera rotation block, CAMPRE = 20 doubles: R row-major (9) | c_0, c_1, c_2 (9) | flag | pad.
// d(R(w) X)/dw_m = (dR/dw_m) X = c_m x (R X) with [c_m]x = (dR/dw_m) R' -- an identity of the exponential map, so the three
// 3x3 derivative matrices (27 doubles, 27 fma per observation) shrink to three vectors (18 mul/fma) and the block fits the
// scalar registers next to K, t and the column scales.  ceres::AngleAxisRotatePoint switches to the first-order formula
// p = X + w x X for theta^2 <= DBL_EPSILON, whose derivative is e_m x X exactly: there flag = 1, c_m = e_m, and the kernels
// take the cross product with X instead of R X (obs_linearize), which reproduces that branch bit for bit.
#define CAMPRE 20
__device__ __forceinline__ void campre_one(const double* __restrict__ e, double* __restrict__ o)
{
    double R[3][3], dR[3][3][3];            // dR[m][k][col] = d R[k][col] / d w_m
#pragma unroll
    for (int col = 0; col < 3; ++col) {
        const double X[3] = { col == 0 ? 1.0 : 0.0, col == 1 ? 1.0 : 0.0, col == 2 ? 1.0 : 0.0 };
        double p[3], dpw[3][3];
        rotate_with_derivs(e, X, p, dpw);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            R[k][col] = p[k];
#pragma unroll
            for (int m = 0; m < 3; ++m) dR[m][k][col] = dpw[m][k];
        }
    }
    const bool small = !(e[0] * e[0] + e[1] * e[1] + e[2] * e[2] > DBL_EPSILON);
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int col = 0; col < 3; ++col) o[3 * k + col] = R[k][col];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        // M = dR_m R' (skew-symmetric up to rounding: take the antisymmetric part)
        double M[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) M[a][b] = dR[m][a][0] * R[b][0] + dR[m][a][1] * R[b][1] + dR[m][a][2] * R[b][2];
        o[9 + 3 * m + 0] = small ? (m == 0 ? 1.0 : 0.0) : 0.5 * (M[2][1] - M[1][2]);
        o[9 + 3 * m + 1] = small ? (m == 1 ? 1.0 : 0.0) : 0.5 * (M[0][2] - M[2][0]);
        o[9 + 3 * m + 2] = small ? (m == 2 ? 1.0 : 0.0) : 0.5 * (M[1][0] - M[0][1]);
    }
    o[18] = small ? 1.0 : 0.0;
    o[19] = 0.0;
}

__global__ void ba_campre_kernel(const double* __restrict__ ext, int nc, double* __restrict__ pre)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < nc) campre_one(ext + 6 * c, pre + CAMPRE * (size_t)c);
}

// 1/sqrt(d) and 1/d by the hardware seed + two Newton steps (~1 ulp): 7 / 5 dependent ops, where sqrt() and the IEEE
// divide expand to ~25 / ~12 instructions each.  Every kernel linearises each observation once or twice per pass, so
// these sit on the busiest (VALU-bound) part of the iteration; the oracle's correctly rounded results differ by ~1e-16.
__device__ __forceinline__ double rsqrt_nr(double d)
{
    double y = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
    double e = fma(-h * y, y, 0.5); y = fma(y, e, y);
    e = fma(-h * y, y, 0.5); y = fma(y, e, y);
    return y;
}
__device__ __forceinline__ double rcp_nr(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}

// Huber: rho(s) and sqrt(rho'(s)) (the corrector that scales residual and Jacobian), s = |r|^2
__device__ __forceinline__ void huber_rho(double a, double s, double& rho0, double& sqrt_rho1)
{
    rho0 = s; sqrt_rho1 = 1.0;
    if (a > 0.0 && s > a * a) {
        const double rs = rsqrt_nr(s);                  // 1/|r|
        rho0 = 2.0 * a * (s * rs) - a * a;
        const double z = fmax(DBL_MIN, a * rs);         // rho' = a / |r|
        sqrt_rho1 = z * rsqrt_nr(z);
    }
}

// cost only: 1/2 rho(|r|^2); pre = the camera's rotation block (only R is read), t = its translation
__device__ __forceinline__ double obs_cost(const double* __restrict__ K4, const double* __restrict__ pre, const double* __restrict__ t,
                                           const double X[3], double u, double v, double huber_a)
{
    const double p0 = pre[0] * X[0] + pre[1] * X[1] + pre[2] * X[2] + t[0];
    const double p1 = pre[3] * X[0] + pre[4] * X[1] + pre[5] * X[2] + t[1];
    const double p2 = pre[6] * X[0] + pre[7] * X[1] + pre[8] * X[2] + t[2];
    const double iz = rcp_nr(p2);
    const double r0 = K4[0] * (p0 * iz) + K4[2] - u;
    const double r1 = K4[1] * (p1 * iz) + K4[3] - v;
    double rho0, sq;
    huber_rho(huber_a, r0 * r0 + r1 * r1, rho0, sq);
    return 0.5 * rho0;
}

__device__ __forceinline__ void obs_linearize(const double* __restrict__ K4, const double* __restrict__ pre, const double* __restrict__ t,
                                              const double X[3], double u, double v, double huber_a,
                                              const double* __restrict__ sK /*4 or null*/, const double* __restrict__ sc /*6 or null*/,
                                              const double sp[3], ObsLin& o)
{
    double R[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) R[i][j] = pre[3 * i + j];
    double q[3], p[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { q[k] = R[k][0] * X[0] + R[k][1] * X[1] + R[k][2] * X[2]; p[k] = q[k] + t[k]; }
    const double iz = rcp_nr(p[2]);
    const double x = p[0] * iz, y = p[1] * iz;
    double r0 = K4[0] * x + K4[2] - u;
    double r1 = K4[1] * y + K4[3] - v;
    double rho0, sq;
    huber_rho(huber_a, r0 * r0 + r1 * r1, rho0, sq);
    o.rho0 = rho0;
    o.r[0] = sq * r0; o.r[1] = sq * r1;
    // d(u,v)/dp, pre-multiplied by the corrector
    const double a00 = sq * K4[0] * iz, a02 = -sq * K4[0] * x * iz;
    const double a11 = sq * K4[1] * iz, a12 = -sq * K4[1] * y * iz;
    if (sK) {
        o.EK[0][0] = sq * x * sK[0]; o.EK[0][1] = 0.0; o.EK[0][2] = sq * sK[2]; o.EK[0][3] = 0.0;
        o.EK[1][0] = 0.0; o.EK[1][1] = sq * y * sK[1]; o.EK[1][2] = 0.0; o.EK[1][3] = sq * sK[3];
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) { o.EK[0][j] = 0.0; o.EK[1][j] = 0.0; }
    }
    if (sc) {
        // dp_m = c_m x (R X), or e_m x X in the first-order branch of the angle-axis formula (campre_one)
        const bool first_order = pre[18] != 0.0;
        const double g0 = first_order ? X[0] : q[0], g1 = first_order ? X[1] : q[1], g2 = first_order ? X[2] : q[2];
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const double* cm = pre + 9 + 3 * m;
            const double dp0 = cm[1] * g2 - cm[2] * g1;
            const double dp1 = cm[2] * g0 - cm[0] * g2;
            const double dp2 = cm[0] * g1 - cm[1] * g0;
            o.Ec[0][m] = (a00 * dp0 + a02 * dp2) * sc[m];
            o.Ec[1][m] = (a11 * dp1 + a12 * dp2) * sc[m];
        }
        o.Ec[0][3] = a00 * sc[3]; o.Ec[0][4] = 0.0;         o.Ec[0][5] = a02 * sc[5];
        o.Ec[1][3] = 0.0;         o.Ec[1][4] = a11 * sc[4]; o.Ec[1][5] = a12 * sc[5];
    } else {
#pragma unroll
        for (int j = 0; j < 6; ++j) { o.Ec[0][j] = 0.0; o.Ec[1][j] = 0.0; }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        o.F[0][j] = (a00 * R[0][j] + a02 * R[2][j]) * sp[j];
        o.F[1][j] = (a11 * R[1][j] + a12 * R[2][j]) * sp[j];
    }
}

// symmetric 3x3 inverse through Cholesky; V = [v00 v10 v11 v20 v21 v22] (lower), same layout out.
__device__ __forceinline__ bool inv3_spd(const double V[6], double Vi[6])
{
    // only the reciprocals of the Cholesky diagonal are needed: rsqrt_nr instead of sqrt + divide
    bool ok = V[0] > 0.0;
    const double i00 = rsqrt_nr(V[0]);
    const double l10 = V[1] * i00, l20 = V[3] * i00;
    const double d11 = V[2] - l10 * l10; ok = ok && d11 > 0.0;
    const double i11 = rsqrt_nr(d11);
    const double l21 = (V[4] - l20 * l10) * i11;
    const double d22 = V[5] - l20 * l20 - l21 * l21; ok = ok && d22 > 0.0;
    const double i22 = rsqrt_nr(d22);
    const double i10 = -l10 * i00 * i11;
    const double i21 = -l21 * i11 * i22;
    const double i20 = -(l20 * i00 + l21 * i10) * i22;
    Vi[0] = i00 * i00 + i10 * i10 + i20 * i20;
    Vi[1] = i10 * i11 + i20 * i21;
    Vi[2] = i11 * i11 + i21 * i21;
    Vi[3] = i20 * i22;
    Vi[4] = i21 * i22;
    Vi[5] = i22 * i22;
    return ok;
}
// y = Vi (sym, lower-packed) * x
__device__ __forceinline__ void symv3(const double Vi[6], const double x[3], double y[3])
{
    y[0] = Vi[0] * x[0] + Vi[1] * x[1] + Vi[3] * x[2];
    y[1] = Vi[1] * x[0] + Vi[2] * x[1] + Vi[4] * x[2];
    y[2] = Vi[3] * x[0] + Vi[4] * x[1] + Vi[5] * x[2];
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    return v;
}

// N per-lane accumulators -> their 64-lane totals, one per lane: every step pairs accumulators (x, y), leaves the lanes
// whose bit `off` is clear with x summed over {l, l ^ off} and the others with y, and so halves the number of live values
// (N + N/2 + ... ~ 2N exchanges instead of 6N butterflies).  Steps 32 and 16 are one v_permlane{32,16}_swap per dword,
// no select.  Returns the total of accumulator wave_scatter_index(lane) (garbage where that index is >= N).  The
// summation order is fixed, so results are run-to-run identical.
__device__ __forceinline__ void lane_swap32(double& a, double& b)
{
    const unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    const auto r0 = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    a = __hiloint2double(r1[0], r0[0]); b = __hiloint2double(r1[1], r0[1]);
}
__device__ __forceinline__ void lane_swap16(double& a, double& b)
{
    const unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    const auto r0 = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    a = __hiloint2double(r1[0], r0[0]); b = __hiloint2double(r1[1], r0[1]);
}
__device__ __forceinline__ int wave_scatter_index(int lane) { return (int)(__brev((unsigned)lane) >> 26); }

template <int OFF, int N>
__device__ __forceinline__ void wave_scatter_step(const double (&in)[N], double (&out)[(N + 1) / 2], int lane)
{
#pragma unroll
    for (int i = 0; i < (N + 1) / 2; ++i) {
        double x = in[2 * i], y = (2 * i + 1 < N) ? in[2 * i + 1] : 0.0;
        if (OFF == 32) { lane_swap32(x, y); out[i] = x + y; }
        else if (OFF == 16) { lane_swap16(x, y); out[i] = x + y; }
        else {
            const bool up = (lane & OFF) != 0;
            const double keep = up ? y : x, send = up ? x : y;
            out[i] = keep + __shfl_xor(send, OFF);
        }
    }
}
template <int N>
__device__ __forceinline__ double wave_reduce_scatter(const double (&a)[N], int lane)
{
    static_assert(N >= 1 && N <= 64, "one accumulator per lane at most");
    double s1[(N + 1) / 2];                     wave_scatter_step<32>(a, s1, lane);
    double s2[((N + 1) / 2 + 1) / 2];           wave_scatter_step<16>(s1, s2, lane);
    constexpr int N2 = ((N + 1) / 2 + 1) / 2;
    double s3[(N2 + 1) / 2];                    wave_scatter_step<8>(s2, s3, lane);
    constexpr int N3 = (N2 + 1) / 2;
    double s4[(N3 + 1) / 2];                    wave_scatter_step<4>(s3, s4, lane);
    constexpr int N4 = (N3 + 1) / 2;
    double s5[(N4 + 1) / 2];                    wave_scatter_step<2>(s4, s5, lane);
    constexpr int N5 = (N4 + 1) / 2;
    double s6[(N5 + 1) / 2];                    wave_scatter_step<1>(s5, s6, lane);
    return s6[0];
}

// ------------------------------------------------------------------------------------------------
// K_pt: one thread per point.  V_p = sum F'F + D_p^2, b_p = sum F'r, WK_p = sum EK'F; stores V_p^-1, b_p, WK_p,
// the raw squared column norms, and per-block partials of: cost, the point-eliminated intrinsic terms
// SKK = sum WK V^-1 WK' (10), gK = sum WK V^-1 b (4), max |gradient| over the block's point columns.
// part_pt[block][32] = { cost, SKK[10], gK[4], gmax, UKK = sum EK'EK [10], sum EK'r [4], - , - }: the last two sums run over
// every observation, so they are taken here, where each observation is linearised with its intrinsic columns anyway
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ba_point_kernel(BADev P, int* __restrict__ err)
{
    __shared__ double red[4][32];
    const int p = blockIdx.x * 256 + threadIdx.x;
    double acc[30];
#pragma unroll
    for (int i = 0; i < 30; ++i) acc[i] = 0.0;
    // per-point results, stored after the block below through a per-wave LDS transposition (see there)
    double Vi[6] = { 0, 0, 0, 0, 0, 0 }, b[3] = { 0, 0, 0 }, WK[12], cs[3] = { 0, 0, 0 };
#pragma unroll
    for (int i = 0; i < 12; ++i) WK[i] = 0.0;
    if (p < P.np) {
        const double X[3] = { P.pts[3 * p], P.pts[3 * p + 1], P.pts[3 * p + 2] };
        const double sp[3] = { P.scale_p[3 * p], P.scale_p[3 * p + 1], P.scale_p[3 * p + 2] };
        double V[6] = { 0, 0, 0, 0, 0, 0 };
        double cost = 0.0;
        const int s0 = P.pt_start[p], s1 = P.pt_start[p + 1];
        for (int k = s0; k < s1; ++k) {
            const int c = P.ocam[k];
            ObsLin o;
            obs_linearize(P.K, P.campre + CAMPRE * (size_t)c, P.ext + 6 * c + 3, X, P.ouv[2 * k], P.ouv[2 * k + 1], P.huber_a,
                          P.fixK ? nullptr : P.scale_c + P.koff, nullptr, sp, o);
            cost += 0.5 * o.rho0;
            // two chained fma per sum (x*y + z*w + acc would be mul, fma, add)
#define ACC2(dst, x0, y0, x1, y1) do { dst = fma(x0, y0, dst); dst = fma(x1, y1, dst); } while (0)
            if (!P.fixK) {
                int a = 16;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j <= i; ++j) { ACC2(acc[a], o.EK[0][i], o.EK[0][j], o.EK[1][i], o.EK[1][j]); ++a; }
#pragma unroll
                for (int i = 0; i < 4; ++i) ACC2(acc[26 + i], o.EK[0][i], o.r[0], o.EK[1][i], o.r[1]);
            }
            ACC2(V[0], o.F[0][0], o.F[0][0], o.F[1][0], o.F[1][0]);
            ACC2(V[1], o.F[0][1], o.F[0][0], o.F[1][1], o.F[1][0]);
            ACC2(V[2], o.F[0][1], o.F[0][1], o.F[1][1], o.F[1][1]);
            ACC2(V[3], o.F[0][2], o.F[0][0], o.F[1][2], o.F[1][0]);
            ACC2(V[4], o.F[0][2], o.F[0][1], o.F[1][2], o.F[1][1]);
            ACC2(V[5], o.F[0][2], o.F[0][2], o.F[1][2], o.F[1][2]);
#pragma unroll
            for (int j = 0; j < 3; ++j) ACC2(b[j], o.F[0][j], o.r[0], o.F[1][j], o.r[1]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) ACC2(WK[3 * i + j], o.EK[0][i], o.F[0][j], o.EK[1][i], o.F[1][j]);
#undef ACC2
        }
        cs[0] = V[0]; cs[1] = V[2]; cs[2] = V[5];
        V[0] += fmin(fmax(cs[0], P.min_diag), P.max_diag) / P.radius;
        V[2] += fmin(fmax(cs[1], P.min_diag), P.max_diag) / P.radius;
        V[5] += fmin(fmax(cs[2], P.min_diag), P.max_diag) / P.radius;
        if (!inv3_spd(V, Vi)) *err = 1;
        // T = WK Vi (4x3); SKK = T WK' ; gK = T b
        double T[12];
#pragma unroll
        for (int i = 0; i < 4; ++i) symv3(Vi, &WK[3 * i], &T[3 * i]);
        acc[0] = cost;
        int q = 1;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j)
                acc[q++] = T[3 * i] * WK[3 * j] + T[3 * i + 1] * WK[3 * j + 1] + T[3 * i + 2] * WK[3 * j + 2];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[11 + i] = T[3 * i] * b[0] + T[3 * i + 1] * b[1] + T[3 * i + 2] * b[2];
        acc[15] = fmax(fabs(b[0] * rcp_nr(sp[0])), fmax(fabs(b[1] * rcp_nr(sp[1])), fabs(b[2] * rcp_nr(sp[2]))));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        // Stores through a per-wave LDS transposition: a wave's 64 records of an array are one contiguous span, written as
        // whole 16-byte (8-byte) pieces lane after lane.  Stored straight from the registers, every instruction scattered
        // 16 B per lane at a 48/96-byte stride: 13 partial-line requests per point at the L2, ~4M per launch.
        __shared__ double stage[4][64 * 13];
        double* buf = stage[wave];
        const int first = blockIdx.x * 256 + wave * 64;               // this wave's first point
        const int npts = min(64, P.np - first);                        // records of this wave that exist (<= 0: none)
        // W_K: 12 doubles per point = 384 double2 per wave
#pragma unroll
        for (int i = 0; i < 12; ++i) buf[lane * 13 + i] = WK[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int j = lane + 64 * i, pt = j / 6, pr = j - 6 * pt;
            if (pt < npts) *(double2*)(P.WK + 12 * (size_t)first + 2 * j) = make_double2(buf[pt * 13 + 2 * pr], buf[pt * 13 + 2 * pr + 1]);
        }
        // V^-1: 6 doubles per point = 192 double2 per wave
#pragma unroll
        for (int i = 0; i < 6; ++i) buf[lane * 7 + i] = Vi[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int j = lane + 64 * i, pt = j / 3, pr = j - 3 * pt;
            if (pt < npts) *(double2*)(P.Vinv + 6 * (size_t)first + 2 * j) = make_double2(buf[pt * 7 + 2 * pr], buf[pt * 7 + 2 * pr + 1]);
        }
        // b_p and the raw column norms: 3 doubles per point, already contiguous as [point][3]
#pragma unroll
        for (int i = 0; i < 3; ++i) { buf[lane * 3 + i] = b[i]; buf[256 + lane * 3 + i] = cs[i]; }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int f = lane + 64 * i;
            if (f < 3 * npts) { P.bp[3 * (size_t)first + f] = buf[f]; P.colsq_p[3 * (size_t)first + f] = buf[256 + f]; }
        }
    }
    {
        double v29[29];                     // the 29 sums: slots 0..14 and 16..29 (slot 15 is a maximum)
#pragma unroll
        for (int i = 0; i < 29; ++i) v29[i] = acc[i < 15 ? i : i + 1];
        const double tot = wave_reduce_scatter(v29, lane);
        const double gm = wave_max(acc[15]);
        const int j = wave_scatter_index(lane);
        if (j < 29) red[wave][j < 15 ? j : j + 1] = tot;
        if (lane == 0) red[wave][15] = gm;
    }
    __syncthreads();
    if (threadIdx.x < 30) {
        const int i = threadIdx.x;
        double v = red[0][i];
        for (int w = 1; w < 4; ++w) v = (i == 15) ? fmax(v, red[w][i]) : v + red[w][i];
        P.part_pt[32 * (size_t)blockIdx.x + i] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// K_cam: block (c, split) walks a slice of camera c's observations.  Per thread 80 accumulators:
//   [0,21)  Scc  = Ec'Ec - T Wc'        (lower 6x6)          T = (Ec'F) V^-1
//   [21,45) ScK  = Ec'EK - T WK_p'      (6x4)
//   [45,51) rhs_c = Ec'r - T b_p
//   [51,65) unused (UKK = EK'EK and gK = EK'r are summed by K_pt)
//   [65,71) diagU_c = diag(Ec'Ec)
//   [71,77) graw_c = Ec'r
// ------------------------------------------------------------------------------------------------
#define CAMACC 80
// The 77 sums are split over two kinds of workgroups (blockIdx.z): part 0 keeps the camera-camera terms (Scc, rhs_c,
// diagU_c, graw_c: 39 accumulators), part 1 the camera-intrinsic ones (ScK, UKK, gK: 38).  Both re-derive the
// linearisation; with all 77 in one thread the kernel needed 324 registers = one wave per SIMD, and PMC showed it
// 42 % waiting on its gathers with nothing else resident to issue (140 us; the split form: see profiles/README.md).
__host__ __device__ constexpr int cam_part_slot(int part, int j)
{
    return part == 0 ? (j < 21 ? j : j < 27 ? 45 + (j - 21) : 65 + (j - 27)) : 21 + j;
}
// Workgroup L of a 1-D grid runs on XCD L % 8 (round-robin dispatch) with its own 4 MB L2.  Work item v = (L % 8) * per + L / 8
// gives every XCD one contiguous range of items: consecutive cameras (and camera pairs) share most of their points, so
// each XCD's L2 serves a point's record to all the observations of it instead of every observation fetching it over the
// fabric (the linearisation moved ~0.5 GB per pass for 90 MB of distinct data and ran at the fabric's rate).
__device__ __forceinline__ int xcd_item_of(int L, int n_blocks, int n_items, int plain)      // n_blocks (and the first block of the range) multiples of 8
{
    const int per = (n_blocks + 7) >> 3;
    const int v = plain ? L : (L & 7) * per + (L >> 3);
    return v < n_items ? v : -1;
}
__device__ __forceinline__ int xcd_item(int n_items, int plain) { return xcd_item_of(blockIdx.x, gridDim.x, n_items, plain); }

template <int PART>
__device__ __forceinline__ void ba_camera_body(const BADev& P, double (*red)[CAMACC], int c, int sp_i)
{
    const int co = cam_off(P, c);
    double acc[CAMACC];
#pragma unroll
    for (int i = 0; i < CAMACC; ++i) acc[i] = 0.0;
    const int s0 = P.cam_start[c], s1 = P.cam_start[c + 1];
    const int len = s1 - s0, per = (len + P.cam_split - 1) / P.cam_split;
    const int b0 = s0 + sp_i * per;
    int b1 = b0 + per; if (b1 > s1) b1 = s1;
    const double* sc = co < 0 ? nullptr : P.scale_c + co;
    const double* sK = P.fixK ? nullptr : P.scale_c + P.koff;
    // the next trip's point index is fetched one trip ahead: the gathers below then start without waiting for it
    int p_next = (b0 + (int)threadIdx.x < b1) ? P.cam_pt[b0 + threadIdx.x] : 0;
    for (int q = b0 + threadIdx.x; q < b1; q += 256) {
        const int p = p_next;
        if (q + 256 < b1) p_next = P.cam_pt[q + 256];
        const double X[3] = { P.pts[3 * p], P.pts[3 * p + 1], P.pts[3 * p + 2] };
        const double spp[3] = { P.scale_p[3 * p], P.scale_p[3 * p + 1], P.scale_p[3 * p + 2] };
        ObsLin o;
        obs_linearize(P.K, P.campre + CAMPRE * (size_t)c, P.ext + 6 * c + 3, X, P.cam_uv[2 * (size_t)q], P.cam_uv[2 * (size_t)q + 1], P.huber_a, sK, sc, spp, o);
        double Vi[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) Vi[i] = P.Vinv[6 * (size_t)p + i];
        // (Ec' F) V^-1 (F' Ec) = Ec' (F V^-1 F') Ec: the 2x2 core G F' is formed first, so the point block never
        // expands to 6x3 per observation
        double G[2][3];
        symv3(Vi, o.F[0], G[0]);
        symv3(Vi, o.F[1], G[1]);
        if (PART == 0) {
            double b[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) b[i] = P.bp[3 * (size_t)p + i];
            // Q = I - F V^-1 F' (symmetric), N = Q Ec
            const double q00 = 1.0 - (G[0][0] * o.F[0][0] + G[0][1] * o.F[0][1] + G[0][2] * o.F[0][2]);
            const double q01 = -(G[0][0] * o.F[1][0] + G[0][1] * o.F[1][1] + G[0][2] * o.F[1][2]);
            const double q11 = 1.0 - (G[1][0] * o.F[1][0] + G[1][1] * o.F[1][1] + G[1][2] * o.F[1][2]);
            // Ec[1][3] and Ec[0][4] are exactly zero (the x / y translation columns touch one residual row each): their products
            // are skipped -- fma(0, finite, acc) == acc, so the sums keep their bits
            double N0[6], N1[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                N0[j] = j == 4 ? q01 * o.Ec[1][j] : j == 3 ? q00 * o.Ec[0][j] : q00 * o.Ec[0][j] + q01 * o.Ec[1][j];
                N1[j] = j == 4 ? q11 * o.Ec[1][j] : j == 3 ? q01 * o.Ec[0][j] : q01 * o.Ec[0][j] + q11 * o.Ec[1][j];
            }
            int a = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) { if (i != 4) acc[a] = fma(o.Ec[0][i], N0[j], acc[a]); if (i != 3) acc[a] = fma(o.Ec[1][i], N1[j], acc[a]); ++a; }
            const double r0 = o.r[0] - (G[0][0] * b[0] + G[0][1] * b[1] + G[0][2] * b[2]);
            const double r1 = o.r[1] - (G[1][0] * b[0] + G[1][1] * b[1] + G[1][2] * b[2]);
#pragma unroll
            for (int i = 0; i < 6; ++i) { if (i != 4) acc[45 + i] = fma(o.Ec[0][i], r0, acc[45 + i]); if (i != 3) acc[45 + i] = fma(o.Ec[1][i], r1, acc[45 + i]); }
#pragma unroll
            for (int i = 0; i < 6; ++i) { if (i != 4) acc[65 + i] = fma(o.Ec[0][i], o.Ec[0][i], acc[65 + i]); if (i != 3) acc[65 + i] = fma(o.Ec[1][i], o.Ec[1][i], acc[65 + i]); }
#pragma unroll
            for (int i = 0; i < 6; ++i) { if (i != 4) acc[71 + i] = fma(o.Ec[0][i], o.r[0], acc[71 + i]); if (i != 3) acc[71 + i] = fma(o.Ec[1][i], o.r[1], acc[71 + i]); }
        } else {
            double WK[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) WK[i] = P.WK[12 * (size_t)p + i];
            double H0[4], H1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                H0[j] = o.EK[0][j] - (G[0][0] * WK[3 * j] + G[0][1] * WK[3 * j + 1] + G[0][2] * WK[3 * j + 2]);
                H1[j] = o.EK[1][j] - (G[1][0] * WK[3 * j] + G[1][1] * WK[3 * j + 1] + G[1][2] * WK[3 * j + 2]);
            }
            int a = 21;
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) { if (i != 4) acc[a] = fma(o.Ec[0][i], H0[j], acc[a]); if (i != 3) acc[a] = fma(o.Ec[1][i], H1[j], acc[a]); ++a; }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // this part's slots: part 0 -> [0,21) u [45,51) u [65,77), part 1 -> [21,45)
    constexpr int NV = (PART == 0) ? 39 : 24;
    double v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = acc[cam_part_slot(PART, j)];
    const double tot = wave_reduce_scatter(v, lane);
    const int j = wave_scatter_index(lane);
    if (j < NV) red[wave][cam_part_slot(PART, j)] = tot;
    __syncthreads();
    if (threadIdx.x < 77) {
        const int i = threadIdx.x;
        const bool mine = (PART == 0) ? (i < 21 || (i >= 45 && i < 51) || i >= 65) : (i >= 21 && i < 45);
        if (mine) P.part_cam[((size_t)c * P.cam_split + sp_i) * CAMACC + i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    }
}

// workgroup L of n_blocks: one (camera, slice, part) item
__device__ __forceinline__ void ba_camera_role(const BADev& P, double (*red)[CAMACC], int L, int n_blocks)
{
    // item = (camera * cam_split + slice) * parts + part: the two parts of a slice sit next to each other
    const int parts = P.fixK ? 1 : 2;
    const int v = xcd_item_of(L, n_blocks, P.nc * P.cam_split * parts, P.xcd_plain);
    if (v < 0) return;
    const int part = v % parts, cs = v / parts, c = cs / P.cam_split, sp_i = cs % P.cam_split;
    if (part == 0) ba_camera_body<0>(P, red, c, sp_i); else ba_camera_body<1>(P, red, c, sp_i);
}

// ------------------------------------------------------------------------------------------------
// K_finalize: block c < nc sums camera c's split partials and writes its S blocks / rhs / diagU / graw;
// block nc reduces the intrinsic terms: S_KK = sum_c UKK - sum_blocks SKK, rhs_K, cost, local gmax.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ba_finalize_role(const BADev& P, int n_pt_blocks, int blk)
{
    __shared__ double sh[256];
    const int tid = threadIdx.x;
    const int ld = P.npad;
    if (blk < P.nc) {
        const int c = blk, co = cam_off(P, c);
        if (co < 0) return;
        if (tid < CAMACC) {
            double v = 0.0;
            for (int s = 0; s < P.cam_split; ++s) v += P.part_cam[((size_t)c * P.cam_split + s) * CAMACC + tid];
            sh[tid] = v;
        }
        __syncthreads();
        if (tid < 36) {
            const int i = tid / 6, j = tid % 6;
            const int hi = i > j ? i : j, lo = i > j ? j : i;
            P.S[(size_t)(co + i) * ld + co + j] = sh[hi * (hi + 1) / 2 + lo];
        } else if (tid < 60 && !P.fixK) {
            const int q = tid - 36, i = q / 4, j = q % 4;
            const double v = sh[21 + q];
            P.S[(size_t)(co + i) * ld + P.koff + j] = v;
            P.S[(size_t)(P.koff + j) * ld + co + i] = v;
        } else if (tid >= 64 && tid < 70) {
            const int i = tid - 64;
            P.rhs[co + i] = sh[45 + i];
            P.diagU[co + i] = sh[65 + i];
            P.graw[co + i] = sh[71 + i];
        }
        return;
    }
    // intrinsic block + scalars: thread (value i = tid & 31, slice = tid >> 5) sums value i over the point blocks of its slice --
    // a wave reads whole 256-byte records, no cross-lane step -- then the eight slices are folded in a fixed order
    {
        __shared__ double red[8][32];
        const int i = tid & 31, slice = tid >> 5;
        double a = 0.0;
        for (int b0 = slice; b0 < n_pt_blocks; b0 += 8 * 16) {      // sixteen records in flight per thread, folded in order
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int b = b0 + 8 * u; v[u] = b < n_pt_blocks ? P.part_pt[32 * (size_t)b + i] : (i == 15 ? 0.0 : 0.0); }
#pragma unroll
            for (int u = 0; u < 16; ++u) a = (i == 15) ? fmax(a, v[u]) : a + v[u];
        }
        red[slice][i] = a;
        __syncthreads();
        if (tid < 30) {
            double v = red[0][tid];
            for (int w = 1; w < 8; ++w) v = (tid == 15) ? fmax(v, red[w][tid]) : v + red[w][tid];
            sh[tid] = v;
        }
        __syncthreads();
    }
    if (tid == 0) {
        P.scal[SCAL_COST] = sh[0];
        P.scal[SCAL_GMAX_SLOTS + P.rank] = sh[15];
    }
    // padding slots of the message tail: nothing else writes them, and the solver leaves its forward-substitution result in rhs --
    // after a factorisation that broke down (NaN) they would poison every later solve
    for (int i = tid; i < P.npad; i += 256)
        if (!P.posmask[i]) { P.rhs[i] = 0.0; P.diagU[i] = 0.0; P.graw[i] = 0.0; }
    if (!P.fixK) {
        if (tid < 16) {
            const int i = tid / 4, j = tid % 4;
            const int hi = i > j ? i : j, lo = i > j ? j : i;
            const int q = hi * (hi + 1) / 2 + lo;
            P.S[(size_t)(P.koff + i) * ld + P.koff + j] = sh[16 + q] - sh[1 + q];
        } else if (tid >= 32 && tid < 36) {
            const int i = tid - 32;
            P.rhs[P.koff + i] = sh[26 + i] - sh[11 + i];
            P.graw[P.koff + i] = sh[26 + i];
            P.diagU[P.koff + i] = sh[16 + i * (i + 1) / 2 + i];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K_schur: the observation pairs (i in camera a, j in camera b, a >= b) that share a point, sorted by (a, b), are cut
// into chunks of <= SCHUR_CHUNK pairs; one wave per chunk accumulates sum T_i W_j' (6x6) and writes it as a partial.
// K_schur_reduce sums each block's partials in chunk order (fixed order => run-to-run identical) and writes
// S_ab = S_ba' = -sum; a == b (one point seen twice by one camera) adds the symmetrised term onto the diagonal
// block written by K_finalize.  chunk_desc: [ca, cb, first item, end item]; items: [obs i, obs j, point, -].
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ba_schur_role(const BADev& P, const int4* __restrict__ chunk_desc, int n_chunk,
                                              const int4* __restrict__ items, double* __restrict__ part, int L, int n_blocks)
{
    const int lane = threadIdx.x & 63;
    const int item = xcd_item_of(L, n_blocks, (n_chunk + 3) >> 2, P.xcd_plain);  // four consecutive chunks per workgroup
    if (item < 0) return;
    const int chunk = item * 4 + (threadIdx.x >> 6);
    if (chunk >= n_chunk) return;
    const int4 cd = chunk_desc[chunk];
    const int ca = __builtin_amdgcn_readfirstlane(cd.x), cb = __builtin_amdgcn_readfirstlane(cd.y);
    const int s0 = __builtin_amdgcn_readfirstlane(cd.z), s1 = __builtin_amdgcn_readfirstlane(cd.w);
    const int oa = cam_off(P, ca), ob = cam_off(P, cb);
    double acc[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) acc[i] = 0.0;
    int4 it_next = (s0 + lane < s1) ? items[s0 + lane] : make_int4(0, 0, 0, 0);
    for (int q = s0 + lane; q < s1; q += 64) {
        const int4 it = it_next;
        if (q + 64 < s1) it_next = items[q + 64];
        const int ki = it.x, kj = it.y, p = it.z;
        const double X[3] = { P.pts[3 * p], P.pts[3 * p + 1], P.pts[3 * p + 2] };
        const double spp[3] = { P.scale_p[3 * p], P.scale_p[3 * p + 1], P.scale_p[3 * p + 2] };
        double Vi[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) Vi[i] = P.Vinv[6 * (size_t)p + i];
        // both pixels requested here, with the point's records: the second one used to be fetched between the two linearisations
        // (C5: 0.69 -> 0.66 ms for the linearisation phase)
        const double ui = P.ouv[2 * ki], vi = P.ouv[2 * ki + 1], uj = P.ouv[2 * kj], vj = P.ouv[2 * kj + 1];
        // T_i W_j' = Ec_a' (F_a V^-1 F_b') Ec_b: 2x2 core first, then 2x6, then the 6x6 outer product (126 fma against
        // 234 for the 6x3 forms)
        double G[2][3], Ea[2][6];
        {
            ObsLin o;
            obs_linearize(P.K, P.campre + CAMPRE * (size_t)ca, P.ext + 6 * ca + 3, X, ui, vi, P.huber_a, nullptr, P.scale_c + oa, spp, o);
            symv3(Vi, o.F[0], G[0]);
            symv3(Vi, o.F[1], G[1]);
#pragma unroll
            for (int i = 0; i < 6; ++i) { Ea[0][i] = o.Ec[0][i]; Ea[1][i] = o.Ec[1][i]; }
        }
        double N0[6], N1[6];
        {
            ObsLin o;
            obs_linearize(P.K, P.campre + CAMPRE * (size_t)cb, P.ext + 6 * cb + 3, X, uj, vj, P.huber_a, nullptr, P.scale_c + ob, spp, o);
            const double m00 = G[0][0] * o.F[0][0] + G[0][1] * o.F[0][1] + G[0][2] * o.F[0][2];
            const double m01 = G[0][0] * o.F[1][0] + G[0][1] * o.F[1][1] + G[0][2] * o.F[1][2];
            const double m10 = G[1][0] * o.F[0][0] + G[1][1] * o.F[0][1] + G[1][2] * o.F[0][2];
            const double m11 = G[1][0] * o.F[1][0] + G[1][1] * o.F[1][1] + G[1][2] * o.F[1][2];
#pragma unroll
            for (int j = 0; j < 6; ++j) {          // Ec[1][3] == Ec[0][4] == 0 exactly (see ba_camera_body): those products are skipped
                N0[j] = j == 4 ? m01 * o.Ec[1][j] : j == 3 ? m00 * o.Ec[0][j] : m00 * o.Ec[0][j] + m01 * o.Ec[1][j];
                N1[j] = j == 4 ? m11 * o.Ec[1][j] : j == 3 ? m10 * o.Ec[0][j] : m10 * o.Ec[0][j] + m11 * o.Ec[1][j];
            }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) { if (i != 4) acc[6 * i + j] = fma(Ea[0][i], N0[j], acc[6 * i + j]); if (i != 3) acc[6 * i + j] = fma(Ea[1][i], N1[j], acc[6 * i + j]); }
    }
    const double tot = wave_reduce_scatter(acc, lane);
    const int e = wave_scatter_index(lane);
    if (e < 36) part[36 * (size_t)chunk + e] = tot;
}

// diag_pass = 0: the off-diagonal blocks (the only writer of those: runs on the pair kernel's stream, beside K_cam and
// K_finalize); diag_pass = 1: the a == b blocks, which add onto what K_finalize wrote (launched after the join, and only
// when such blocks exist).
__device__ __forceinline__ void ba_schur_reduce_role(const BADev& P, const int* __restrict__ blk_cam, const int* __restrict__ blk_chunk,
                                                     int n_blk, const double* __restrict__ part, int diag_pass, int wg)
{
    const int t = wg * 256 + threadIdx.x;
    const int blk = t / 36, e = t % 36;
    if (blk >= n_blk) return;
    const int ca = blk_cam[2 * blk], cb = blk_cam[2 * blk + 1];
    if ((ca == cb) != (diag_pass != 0)) return;
    const int oa = cam_off(P, ca), ob = cam_off(P, cb);
    const int i = e / 6, j = e % 6, eT = j * 6 + i;
    double sum = 0.0, sumT = 0.0;
    for (int c = blk_chunk[blk]; c < blk_chunk[blk + 1]; ++c) { sum += part[36 * (size_t)c + e]; sumT += part[36 * (size_t)c + eT]; }
    const int ld = P.npad;
    if (ca != cb) {
        P.S[(size_t)(oa + i) * ld + ob + j] = -sum;
        P.S[(size_t)(ob + j) * ld + oa + i] = -sum;
    } else {
        P.S[(size_t)(oa + i) * ld + oa + j] -= sum + sumT;
    }
}

// The launches of the build after K_pt.  K_cam and K_schur depend only on K_pt and both leave issue slots idle, so their
// workgroups share ONE launch (first n_cam_blocks: camera items, the rest: pair chunks; both counts multiples of 8, so a
// workgroup's XCD is its role-local index mod 8) -- as two launches on two streams the fork / join events cost 5-7 us of stream
// gap each.  Likewise the two folds: K_finalize's workgroups and the off-diagonal pass of K_schur_reduce write disjoint parts
// of the message; the rare (a, a) blocks add onto what K_finalize wrote and keep their own launch (diag_pass = 1).
__global__ __launch_bounds__(256, 3) void ba_camschur_kernel(BADev P, int n_cam_blocks, const int4* __restrict__ chunk_desc, int n_chunk,
                                                          const int4* __restrict__ items, double* __restrict__ part)
{
    __shared__ double red[4][CAMACC];
    // groups of 8 workgroups (one per XCD) alternate between the roles in proportion, so that both kinds are resident together
    // from the first wave of dispatches to the last (camera items first, then pair chunks: 89 us; mixed: see profiles/README.md)
    const int G = gridDim.x >> 3, Gc = n_cam_blocks >> 3, g = blockIdx.x >> 3;
    const int before = (int)(((long long)g * Gc) / G), upto = (int)(((long long)(g + 1) * Gc) / G);      // camera groups in [0, g) and [0, g]
    if (upto > before) ba_camera_role(P, red, before * 8 + (blockIdx.x & 7), n_cam_blocks);
    else ba_schur_role(P, chunk_desc, n_chunk, items, part, (g - before) * 8 + (blockIdx.x & 7), gridDim.x - n_cam_blocks);
}
// the two roles as launches of their own: large problems run them on two streams (see enqueue_build).  Compiled for TWO waves per SIMD
// (256 registers, nothing spilled), unlike the one-launch form above: at C5, where the gathers miss the L2s, the linearisation phase
// runs 0.746 -> 0.66 ms that way (three waves: 168 registers and up to ten spilled); at C4 the one-launch form measured 82 us at three
// waves against 84 at two.  Reading a trip's gathers one trip ahead (both kernels, round 3) lost in every combination: 0.68-0.77 ms.
__global__ __launch_bounds__(256, 2) void ba_camera_kernel(BADev P)
{
    __shared__ double red[4][CAMACC];
    ba_camera_role(P, red, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(256, 2) void ba_schur_kernel(BADev P, const int4* __restrict__ chunk_desc, int n_chunk,
                                                       const int4* __restrict__ items, double* __restrict__ part)
{
    ba_schur_role(P, chunk_desc, n_chunk, items, part, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(256) void ba_fold_kernel(BADev P, int n_pt_blocks, const int* __restrict__ blk_cam, const int* __restrict__ blk_chunk,
                                                      int n_blk, const double* __restrict__ part)
{
    if ((int)blockIdx.x <= P.nc) ba_finalize_role(P, n_pt_blocks, blockIdx.x);
    else ba_schur_reduce_role(P, blk_cam, blk_chunk, n_blk, part, 0, blockIdx.x - (P.nc + 1));
}
__global__ __launch_bounds__(256) void ba_schur_reduce_kernel(BADev P, const int* __restrict__ blk_cam, const int* __restrict__ blk_chunk,
                                                              int n_blk, const double* __restrict__ part, int diag_pass)
{
    ba_schur_reduce_role(P, blk_cam, blk_chunk, n_blk, part, diag_pass, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// K_damp (after the all-reduce): S_ii += clamp(diagU_i)/radius, padding rows get a unit diagonal;
// scal[1] = max |graw_i / scale_i| over the camera-side columns, folded with the per-rank point maxima.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ba_damp_kernel(BADev P)
{
    __shared__ double red[4];
    double g = 0.0;
    for (int i = threadIdx.x; i < P.npad; i += 256) {
        if (P.posmask[i] && P.diagU[i] > 0.0) {        // diagU == 0: a parameter no residual touches stays put (unit row, like a padding slot)
            const double d = fmin(fmax(P.diagU[i], P.min_diag), P.max_diag) / P.radius;
            P.S[(size_t)i * P.npad + i] += d;
            g = fmax(g, fabs(P.graw[i] / P.scale_c[i]));
        } else {
            P.S[(size_t)i * P.npad + i] = 1.0;
            P.rhs[i] = 0.0;
        }
    }
    g = wave_max(g);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = g;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        for (int r = 0; r < P.world; ++r) m = fmax(m, P.scal[SCAL_GMAX_SLOTS + r]);
        P.scal[1] = m;
    }
}

// jacobi scaling from the raw column norms (first linearisation, scale == 1): s = 1 / (1 + sqrt(colsq))
__global__ void ba_scale_kernel(const double* __restrict__ colsq, double* __restrict__ scale, int n, const int* __restrict__ mask)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) scale[i] = (!mask || mask[i]) ? 1.0 / (1.0 + sqrt(colsq[i])) : 1.0;
}

// ------------------------------------------------------------------------------------------------
// K_camstep: candidate intrinsics / cameras = x + scale * (-y); out[0] = |delta_cam|^2, out[1] = |x_cand,cam|^2
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ba_camstep_kernel(BADev P, double* __restrict__ out2)
{
#pragma clang fp contract(off)          // x + (-y * scale) must round the same way in block 0 and in the other blocks
    if (blockIdx.x > 0) {
        // blocks 1..: the candidate cameras' rotation blocks (what ba_campre_kernel computes), each thread from its own copy
        // of the candidate extrinsics (same arithmetic as block 0's, so nothing waits for block 0)
        const int c = (blockIdx.x - 1) * 256 + threadIdx.x;
        if (c >= P.nc) return;
        const int co = cam_off(P, c);
        double e[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const double x = P.ext[6 * c + j];
            double d = 0.0;
            if (co >= 0) d = -P.y[co + j] * P.scale_c[co + j];
            e[j] = x + d;
        }
        campre_one(e, const_cast<double*>(P.campre_c) + CAMPRE * (size_t)c);
        return;
    }
    __shared__ double red[4][2];
    double dn = 0.0, xn = 0.0;
    for (int i = threadIdx.x; i < 6 * P.nc + 4; i += 256) {
        double x, d = 0.0;
        if (i < 4) {
            x = P.K[i];
            if (!P.fixK) d = -P.y[P.koff + i] * P.scale_c[P.koff + i];
            P.Kc[i] = x + d;
            if (!P.fixK) { dn += d * d; xn += (x + d) * (x + d); }
        } else {
            const int c = (i - 4) / 6, j = (i - 4) % 6;
            const int co = cam_off(P, c);
            x = P.ext[i - 4];
            if (co >= 0) d = -P.y[co + j] * P.scale_c[co + j];
            P.extc[i - 4] = x + d;
            if (co >= 0) { dn += d * d; xn += (x + d) * (x + d); }
        }
    }
    dn = wave_sum(dn); xn = wave_sum(xn);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = dn; red[threadIdx.x >> 6][1] = xn; }
    __syncthreads();                        // also: the candidate extrinsics written above are visible to the whole block
    if (threadIdx.x == 0) {
        out2[0] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
        out2[1] = red[0][1] + red[1][1] + red[2][1] + red[3][1];
    }
    // max |gradient| (what ba_damp_kernel reports when it runs; the sparse solvers damp S themselves): camera-side columns
    // folded with the per-rank point maxima.  graw is not touched by the factorisation.
    {
        __shared__ double gred[4];
        double g = 0.0;
        for (int i = threadIdx.x; i < P.npad; i += 256)
            if (P.posmask[i]) g = fmax(g, fabs(P.graw[i] / P.scale_c[i]));
        g = wave_max(g);
        if ((threadIdx.x & 63) == 0) gred[threadIdx.x >> 6] = g;
        __syncthreads();
        if (threadIdx.x == 0) {
            double m = fmax(fmax(gred[0], gred[1]), fmax(gred[2], gred[3]));
            for (int r = 0; r < P.world; ++r) m = fmax(m, P.scal[SCAL_GMAX_SLOTS + r]);
            P.scal[1] = m;
        }
    }
}

// The scalars the host needs to accept or reject the step, gathered into pinned host memory by one wave, then a sequence
// number with system-scope release: the host polls it.  (Four 8..32-byte copies cost ~5 us each on the stream.)
// host_out: [cost, gmax, mcc, cand, dn_p, xn_p, dn_c, xn_c, err] + seq at [15]
__global__ void ba_publish_kernel(const double* __restrict__ scal2, const double* __restrict__ back4, const double* __restrict__ cam2,
                                  int* __restrict__ err, double* __restrict__ host_out, unsigned long long seq, int clear_err)
{
    const int l = threadIdx.x;
    if (l < 2) host_out[l] = scal2[l];
    else if (l < 6) host_out[l] = back4[l - 2];
    else if (l < 8) host_out[l] = cam2[l - 6];
    else if (l == 8) {
        // clear_err == 0 <=> multi-rank: back4[4] is the all-reduced flag count, the local flag is re-armed by the next build;
        // 2 <=> multi-rank with folded step scalars: *err already belongs to the speculative linearisation, only back4[4] counts
        host_out[8] = clear_err == 1 ? (double)*err : (((clear_err == 0 && *err != 0) || back4[4] > 0.0) ? 1.0 : 0.0);
        if (clear_err == 1) *err = 0;
    }
    __threadfence_system();
    __syncthreads();
    if (l == 0) __hip_atomic_store((unsigned long long*)(host_out + 15), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------------------------------------
// K_back: one thread per point.  y_p = V^-1 (b_p - sum F'(E y_c)), step_p = -y_p, candidate point,
// model cost change -sum m.(r + m/2) with m = J step, candidate cost at (Kc, extc, ptsc).
// part_back[block][4] = { model_cost_change, candidate_cost, |delta_p|^2, |x_cand,p|^2 }
// ------------------------------------------------------------------------------------------------
#define BACK_NCL 24       // cameras of a block's points staged in LDS (points are stored sorted by camera set: a block of 256
                          // consecutive points sees a handful of neighbouring cameras); wider blocks read the cameras from global
#define BACK_REC 60       // doubles per staged camera: rotation block (20) | t (3) | candidate block (20) | candidate t (3) | scale (6) | y (6) | pad

struct BackCam { const double *pre, *t, *pre_c, *t_c, *sc, *y; };     // one camera's inputs of K_back (LDS record or global arrays)

// body of K_back for one point; cam_at(c) -> BackCam
#define BACK_ESAVE 8      // observations per point whose (E y) pair the first pass parks in LDS for the second (longer tracks re-derive it)
template <typename CamAt>
__device__ __forceinline__ void ba_back_point(const BADev& P, int p, CamAt cam_at, double acc[4], double (*esave)[256][2])
{
    const int tid = threadIdx.x;
    const double X[3] = { P.pts[3 * p], P.pts[3 * p + 1], P.pts[3 * p + 2] };
    const double sp[3] = { P.scale_p[3 * p], P.scale_p[3 * p + 1], P.scale_p[3 * p + 2] };
    double t[3] = { P.bp[3 * (size_t)p], P.bp[3 * (size_t)p + 1], P.bp[3 * (size_t)p + 2] };
    double Vi[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) Vi[i] = P.Vinv[6 * (size_t)p + i];
    const int s0 = P.pt_start[p], s1 = P.pt_start[p + 1];
    const double* sK = P.fixK ? nullptr : P.scale_c + P.koff;
    double yK[4] = { 0, 0, 0, 0 };
    if (!P.fixK) { yK[0] = P.y[P.koff]; yK[1] = P.y[P.koff + 1]; yK[2] = P.y[P.koff + 2]; yK[3] = P.y[P.koff + 3]; }
    for (int k = s0; k < s1; ++k) {
        const int c = P.ocam[k];
        const bool free_cam = cam_off(P, c) >= 0;
        const BackCam rec = cam_at(c);
        ObsLin o;
        obs_linearize(P.K, rec.pre, rec.t, X, P.ouv[2 * k], P.ouv[2 * k + 1], P.huber_a, sK, free_cam ? rec.sc : nullptr, sp, o);
        double e0 = 0.0, e1 = 0.0;
        if (free_cam)
#pragma unroll
            for (int j = 0; j < 6; ++j) { const double yy = rec.y[j]; e0 += o.Ec[0][j] * yy; e1 += o.Ec[1][j] * yy; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { e0 += o.EK[0][j] * yK[j]; e1 += o.EK[1][j] * yK[j]; }
#pragma unroll
        for (int j = 0; j < 3; ++j) t[j] -= o.F[0][j] * e0 + o.F[1][j] * e1;
        if (k - s0 < BACK_ESAVE) { esave[k - s0][tid][0] = e0; esave[k - s0][tid][1] = e1; }
    }
    double yp[3];
    symv3(Vi, t, yp);
    const double d[3] = { -yp[0] * sp[0], -yp[1] * sp[1], -yp[2] * sp[2] };
    const double Xc[3] = { X[0] + d[0], X[1] + d[1], X[2] + d[2] };
    P.ptsc[3 * (size_t)p] = Xc[0]; P.ptsc[3 * (size_t)p + 1] = Xc[1]; P.ptsc[3 * (size_t)p + 2] = Xc[2];
    acc[2] = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    acc[3] = Xc[0] * Xc[0] + Xc[1] * Xc[1] + Xc[2] * Xc[2];
    for (int k = s0; k < s1; ++k) {
        const int c = P.ocam[k];
        const bool free_cam = cam_off(P, c) >= 0;
        const BackCam rec = cam_at(c);
        ObsLin o;
        double m0 = 0.0, m1 = 0.0;
        if (k - s0 < BACK_ESAVE) {
            // the camera / intrinsic part of the model step is -(E y) of the first pass (the same sums in the same order, so the
            // same bits): only r and F are re-derived here, without the 2 x 10 camera-side Jacobian
            obs_linearize(P.K, rec.pre, rec.t, X, P.ouv[2 * k], P.ouv[2 * k + 1], P.huber_a, nullptr, nullptr, sp, o);
            m0 = -esave[k - s0][tid][0]; m1 = -esave[k - s0][tid][1];
        } else {
            obs_linearize(P.K, rec.pre, rec.t, X, P.ouv[2 * k], P.ouv[2 * k + 1], P.huber_a, sK, free_cam ? rec.sc : nullptr, sp, o);
            if (free_cam)
#pragma unroll
                for (int j = 0; j < 6; ++j) { const double yy = rec.y[j]; m0 -= o.Ec[0][j] * yy; m1 -= o.Ec[1][j] * yy; }
#pragma unroll
            for (int j = 0; j < 4; ++j) { m0 -= o.EK[0][j] * yK[j]; m1 -= o.EK[1][j] * yK[j]; }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) { m0 -= o.F[0][j] * yp[j]; m1 -= o.F[1][j] * yp[j]; }
        acc[0] -= m0 * (o.r[0] + 0.5 * m0) + m1 * (o.r[1] + 0.5 * m1);
        acc[1] += obs_cost(P.Kc, rec.pre_c, rec.t_c, Xc, P.ouv[2 * k], P.ouv[2 * k + 1], P.huber_a);
    }
}

// fills one camera record (BACK_REC doubles) from the global arrays
__device__ __forceinline__ double back_rec_value(const BADev& P, int c, int f)
{
    if (f < 20) return P.campre[CAMPRE * (size_t)c + f];
    if (f < 23) return P.ext[6 * c + 3 + (f - 20)];
    if (f < 43) return P.campre_c[CAMPRE * (size_t)c + (f - 23)];
    if (f < 46) return P.extc[6 * c + 3 + (f - 43)];
    const int co = cam_off(P, c);
    if (f < 52) return co >= 0 ? P.scale_c[co + (f - 46)] : 0.0;
    if (f < 58) return co >= 0 ? P.y[co + (f - 52)] : 0.0;
    return 0.0;
}

// Workgroups beyond n_pt_blocks zero-fill [z0, z0 + n0) and [z1, z1 + n1) (16-byte aligned, counts even): on a single
// rank the reduced system S and the solver's private buffers, which the factorisation has consumed by now and the next
// linearisation expects empty -- as two hipMemsetAsync calls behind the step's last kernel they sat on the critical path.
#define BACK_ZERO_BLOCKS 512
__global__ __launch_bounds__(256, 3) void ba_back_kernel(BADev P, int n_pt_blocks, double* __restrict__ z0, size_t n0, double* __restrict__ z1, size_t n1)
{
    __shared__ double red[4][4];
    __shared__ __attribute__((aligned(16))) double cam[BACK_NCL][BACK_REC];
    __shared__ double esave[BACK_ESAVE][256][2];
    if ((int)blockIdx.x >= n_pt_blocks) {
        const size_t zb = blockIdx.x - n_pt_blocks, nzb = gridDim.x - n_pt_blocks;
        const size_t total2 = (n0 + n1) / 2, per = (total2 + nzb - 1) / nzb, h0 = n0 / 2;
        const size_t e_end = (zb + 1) * per < total2 ? (zb + 1) * per : total2;
        const double2 zz = make_double2(0.0, 0.0);
        for (size_t e = zb * per + threadIdx.x; e < e_end; e += 256) {
            if (e < h0) ((double2*)z0)[e] = zz; else ((double2*)z1)[e - h0] = zz;
        }
        return;
    }
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double acc[4] = { 0, 0, 0, 0 };
    // camera range of the block's points: fixed by the observation lists, tabulated when the problem is created
    const int cmin = P.blk_crange[2 * blockIdx.x], cmax = P.blk_crange[2 * blockIdx.x + 1];
    const bool staged = cmax >= cmin && cmax - cmin < BACK_NCL;          // block-uniform
    if (staged) {
        const int n = (cmax - cmin + 1) * BACK_REC;
        for (int e = threadIdx.x; e < n; e += 256) cam[e / BACK_REC][e % BACK_REC] = back_rec_value(P, cmin + e / BACK_REC, e % BACK_REC);
    }
    __syncthreads();
    if (p < P.np) {
        if (staged) {
            ba_back_point(P, p, [&](int c) {
                const double* r = &cam[c - cmin][0];
                return BackCam{ r, r + 20, r + 23, r + 43, r + 46, r + 52 }; }, acc, esave);
        } else {            // the block's points span too many cameras: straight from the global arrays
            ba_back_point(P, p, [&](int c) {
                const int co = cam_off(P, c);
                return BackCam{ P.campre + CAMPRE * (size_t)c, P.ext + 6 * c + 3, P.campre_c + CAMPRE * (size_t)c, P.extc + 6 * c + 3,
                                P.scale_c + (co < 0 ? 0 : co), P.y + (co < 0 ? 0 : co) }; }, acc, esave);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = wave_sum(acc[i]);
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[wave][i] = acc[i];
    __syncthreads();
    if (threadIdx.x < 4) {
        const int i = threadIdx.x;
        P.part_back[4 * (size_t)blockIdx.x + i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    }
}

// out4 = sum over blocks of part_back
// With host_out != nullptr (single rank: nothing is all-reduced in between) the block also publishes the decision scalars,
// i.e. does ba_publish_kernel's job in the same launch.
__global__ __launch_bounds__(256) void ba_back_reduce_kernel(const double* __restrict__ part, int nblocks, double* __restrict__ out4,
                                                             const double* __restrict__ scal2, const double* __restrict__ cam2, int* __restrict__ err,
                                                             double* __restrict__ host_out, unsigned long long seq)
{
    __shared__ double red[4][4];
    double acc[4] = { 0, 0, 0, 0 };
    for (int b = threadIdx.x; b < nblocks; b += 256)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += part[4 * (size_t)b + i];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = wave_sum(acc[i]);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[threadIdx.x >> 6][i] = acc[i];
    __syncthreads();
    if (threadIdx.x < 4) {
        const double v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        out4[threadIdx.x] = v;
        if (host_out) host_out[2 + threadIdx.x] = v;
    }
    // multi-rank: the error flag travels with the step scalars (summed by the all-reduce: > 0 on every rank if any rank set it);
    // out4[5..6] keep this linearisation's cost and gradient maximum (scal2[0..1]) for the publish kernel: with the step scalars
    // folded into the next message, the next build has overwritten the message tail by the time the host is told
    if (!host_out && threadIdx.x == 4) out4[4] = *err != 0 ? 1.0 : 0.0;
    if (!host_out && (threadIdx.x == 5 || threadIdx.x == 6)) out4[threadIdx.x] = scal2[threadIdx.x - 5];
    if (host_out) {
        const int l = threadIdx.x;
        if (l >= 64 && l < 66) host_out[l - 64] = scal2[l - 64];
        else if (l >= 66 && l < 68) host_out[6 + l - 66] = cam2[l - 66];
        else if (l == 68) { host_out[8] = (double)*err; *err = 0; }
        __threadfence_system();
        __syncthreads();
        if (l == 0) __hip_atomic_store((unsigned long long*)(host_out + 15), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// cost only at the current parameters (used for |x| bookkeeping at start-up): not needed separately --
// K_pt already returns the cost of the linearisation point.

// |x|^2 over the free parameters of the current point (start-up only)
__global__ __launch_bounds__(256) void ba_xnorm_kernel(BADev P, double* __restrict__ out)
{
    __shared__ double red[4];
    double s = 0.0;
    const size_t total = 4 + 6 * (size_t)P.nc + 3 * (size_t)P.np;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        double v;
        if (i < 4) v = P.fixK ? 0.0 : P.K[i];
        else if (i < 4 + 6 * (size_t)P.nc) { const int c = (int)((i - 4) / 6); v = cam_off(P, c) < 0 ? 0.0 : P.ext[i - 4]; }
        else v = P.pts[i - 4 - 6 * (size_t)P.nc];
        s += v * v;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
