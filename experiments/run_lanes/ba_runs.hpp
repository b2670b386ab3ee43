// ba_runs.hpp -- the linearisation + Schur elimination with ONE LANE PER POINT over runs of points (included by ba.hip only).
//
// Same arithmetic as K_pt / K_cam / K_schur of ba_kernels.hpp (ReprojectCost NView:151-183, HuberLoss(4) NView:1184, the point
// elimination Ceres' SPARSE_SCHUR does for bundle_adjustment NView:1215-1224) and the same decomposition as ba_tiles.hpp:
// points are stored sorted by their camera list, so a RUN of consecutive points sees the same cameras c_0..c_{M-1}; with
// V_p = L L' and F~ = F L^-T an eliminated point contributes -Z Z' (Z = J~' F~, rows [E_c0 | .. | E_cM-1 | E_K | r]: 6M + 5 of them)
// and each of its observations +J~_k' J~_k on the rows [E_ck | E_K | r] (11); summed over the points of a run these are the run's
// whole contribution to [S | rhs | diagU | graw].
//
// Where ba_tiles.hpp spreads a point over four lanes and lets the fp64 matrix pipe contract over points -- which costs an operand
// transposition through LDS and two waves per SIMD -- here a lane owns a point: it linearises the M observations (twice: once for
// the point block and the direct terms, once more after L is known; keeping 18 M Jacobian doubles live would not fit beside the
// 3 (6M + 5) of Z), forms every product of the two lower triangles in registers and the 64 lanes' products are summed by the
// scatter reduction of ba_kernels.hpp (wave_reduce_scatter: ~4 instructions per value for 64 values at a time).  No LDS on the
// data path, no atomics, fixed summation order.  Per point and M = 4: ~6.6k wave instructions per 64 points against ~24k for the three
// per-observation kernels (the point kernel's pass included) -- every observation is linearised twice instead of ~5.5 times, and
// the pair products are taken once per pair of observations instead of re-deriving both Jacobians.
//
// A segment (<= 256 points of one run) is a workgroup, a wave takes 64 of its points; the segment's 66 M + (6M+5)(6M+6)/2 sums are
// added over the four waves in a fixed order and stored compactly; ba_tile_reduce_kernel folds them into the reduced system through
// the table build_tile_tables makes for this layout.
#pragma once
#include "ba_tiles.hpp"
#include <utility>

#define RUN_MMAX 6             // observations per point this path takes (3 (6M + 5) doubles of Z per lane beside a 64-value reduction)
#define RUN_SEG_MAX 256

__host__ __device__ constexpr int run_tri(int r, int c) { return r >= c ? r * (r + 1) / 2 + c : c * (c + 1) / 2 + r; }
__host__ __device__ constexpr int run_nz(int M) { return (6 * M + 5) * (6 * M + 6) / 2; }
__host__ __device__ constexpr int run_nv(int M) { return 66 * M + run_nz(M); }            // values a segment stores
constexpr int run_tri_row(int v) { int a = 0; while ((a + 1) * (a + 2) / 2 <= v) ++a; return a; }

// element V of the lower triangle (row-major packed) of J'J for the 2 x 11 rows J
template <int V>
__device__ __forceinline__ double run_direct_value(const double (&J)[2][11])
{
    constexpr int i = run_tri_row(V), j = V - i * (i + 1) / 2;
    return fma(J[1][i], J[1][j], J[0][i] * J[0][j]);
}
template <int V0, int... I>
__device__ __forceinline__ void run_direct_fill(double (&out)[sizeof...(I)], const double (&J)[2][11], std::integer_sequence<int, I...>)
{
    ((out[I] = run_direct_value<V0 + I>(J)), ...);
}
// element V of the lower triangle of Z Z' for the R x 3 rows Z
template <int R, int V>
__device__ __forceinline__ double run_zz_value(const double (&Z)[R][3])
{
    constexpr int NZV = R * (R + 1) / 2, VV = V < NZV ? V : NZV - 1;
    constexpr int a = run_tri_row(VV), b = VV - a * (a + 1) / 2;
    return V < NZV ? fma(Z[a][2], Z[b][2], fma(Z[a][1], Z[b][1], Z[a][0] * Z[b][0])) : 0.0;
}
template <int R, int V0, int... I>
__device__ __forceinline__ void run_zz_fill(double (&out)[sizeof...(I)], const double (&Z)[R][3], std::integer_sequence<int, I...>)
{
    ((out[I] = run_zz_value<R, V0 + I>(Z)), ...);
}

template <int R, int NZC, int... C>
__device__ __forceinline__ void run_zz_batches(double (&totz)[NZC], const double (&Z)[R][3], int lane, std::integer_sequence<int, C...>)
{
    // NZC batches of 64 values each (compile-time (row, column) of every value)
    (([&] {
        double dv[64];
        run_zz_fill<R, 64 * C>(dv, Z, std::make_integer_sequence<int, 64>{});
        totz[C] = wave_reduce_scatter(dv, lane);
    }()), ...);
}

template <int M>
__device__ __forceinline__ void ba_run_body(const BADev& P, const TileSeg sg, const int* __restrict__ seg_cams,
                                            double* __restrict__ part, double* __restrict__ part_seg, int* __restrict__ err, double* lds)
{
    constexpr int R = 6 * M + 5, NZV = R * (R + 1) / 2, NZC = (NZV + 63) / 64, NV = 66 * M + NZV;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* cam = lds;                                            // [M][TILE_CAMREC], shared by the workgroup
    double* xs = lds + RUN_MMAX * TILE_CAMREC;                    // [4][NV]: the waves' totals
    double* xw = xs + 4 * NV;                                     // [4][2] cost / gmax of the waves
    for (int e = tid; e < M * TILE_CAMREC; e += 256) {
        const int k = e / TILE_CAMREC, f = e % TILE_CAMREC;
        const int c = seg_cams[sg.cams_off + k], co = cam_off(P, c);
        double v = 0.0;
        if (f < 20) v = P.campre[CAMPRE * (size_t)c + f];
        else if (f < 23) v = P.ext[6 * c + 3 + (f - 20)];
        else if (f < 29) v = co >= 0 ? P.scale_c[co + (f - 23)] : 0.0;
        else if (f == 29) v = co >= 0 ? 1.0 : 0.0;
        cam[e] = v;
    }
    __syncthreads();
    const double* sK = P.fixK ? nullptr : P.scale_c + P.koff;
    const int pi = 64 * wave + lane;
    const bool act = pi < sg.npts;
    const int pic = act ? pi : sg.npts - 1;
    const size_t p = (size_t)sg.p0 + pic;
    const double X[3] = { P.pts[3 * p], P.pts[3 * p + 1], P.pts[3 * p + 2] };
    const double sp[3] = { P.scale_p[3 * p], P.scale_p[3 * p + 1], P.scale_p[3 * p + 2] };
    double uv[M][2];
#pragma unroll
    for (int k = 0; k < M; ++k) { const size_t q = (size_t)sg.obs0 + (size_t)pic * M + k; uv[k][0] = P.ouv[2 * q]; uv[k][1] = P.ouv[2 * q + 1]; }
    auto linearize = [&](int k, ObsLin& o) {
        const double* rec = cam + k * TILE_CAMREC;
        obs_linearize(P.K, rec, rec + 20, X, uv[k][0], uv[k][1], P.huber_a, sK, rec[29] != 0.0 ? rec + 23 : nullptr, sp, o);
        if (!act) {                          // a padding lane contributes zeros everywhere
            o.rho0 = 0.0; o.r[0] = o.r[1] = 0.0;
#pragma unroll
            for (int i = 0; i < 4; ++i) { o.EK[0][i] = 0.0; o.EK[1][i] = 0.0; }
#pragma unroll
            for (int i = 0; i < 6; ++i) { o.Ec[0][i] = 0.0; o.Ec[1][i] = 0.0; }
#pragma unroll
            for (int i = 0; i < 3; ++i) { o.F[0][i] = 0.0; o.F[1][i] = 0.0; }
        }
    };
    const int idx = wave_scatter_index(lane);                     // which value of a reduced batch this lane receives
    double totd[M], totd2[M], totz[NZC];
    double cost = 0.0, gmax = 0.0;
    bool bad = false;
    // ---- pass 1: the point block V, b, W_K and the direct terms sum_p J~_k' J~_k of every observation slot
    double s21[21];
#pragma unroll
    for (int i = 0; i < 21; ++i) s21[i] = 0.0;
#pragma unroll
    for (int k = 0; k < M; ++k) {
        ObsLin o;
        linearize(k, o);
        cost += 0.5 * o.rho0;
#define ACC2(dst, x0, y0, x1, y1) do { dst = fma(x0, y0, dst); dst = fma(x1, y1, dst); } while (0)
        ACC2(s21[0], o.F[0][0], o.F[0][0], o.F[1][0], o.F[1][0]);
        ACC2(s21[1], o.F[0][1], o.F[0][0], o.F[1][1], o.F[1][0]);
        ACC2(s21[2], o.F[0][1], o.F[0][1], o.F[1][1], o.F[1][1]);
        ACC2(s21[3], o.F[0][2], o.F[0][0], o.F[1][2], o.F[1][0]);
        ACC2(s21[4], o.F[0][2], o.F[0][1], o.F[1][2], o.F[1][1]);
        ACC2(s21[5], o.F[0][2], o.F[0][2], o.F[1][2], o.F[1][2]);
#pragma unroll
        for (int i = 0; i < 3; ++i) ACC2(s21[6 + i], o.F[0][i], o.r[0], o.F[1][i], o.r[1]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int d = 0; d < 3; ++d) ACC2(s21[9 + 3 * i + d], o.EK[0][i], o.F[0][d], o.EK[1][i], o.F[1][d]);
#undef ACC2
        double J[2][11];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
#pragma unroll
            for (int i = 0; i < 6; ++i) J[e][i] = o.Ec[e][i];
#pragma unroll
            for (int i = 0; i < 4; ++i) J[e][6 + i] = o.EK[e][i];
            J[e][10] = o.r[e];
        }
        {
            double dv[64];
            run_direct_fill<0>(dv, J, std::make_integer_sequence<int, 64>{});
            totd[k] = wave_reduce_scatter(dv, lane);
        }
        {
            double dv2[2];
            run_direct_fill<64>(dv2, J, std::make_integer_sequence<int, 2>{});
            totd2[k] = wave_reduce_scatter(dv2, lane);
        }
    }
    // ---- the point block: damping, V = L L', V^-1
    double V[6], Vi[6], Li[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) V[i] = s21[i];
    const double cs[3] = { V[0], V[2], V[5] };
    V[0] += fmin(fmax(cs[0], P.min_diag), P.max_diag) / P.radius;
    V[2] += fmin(fmax(cs[1], P.min_diag), P.max_diag) / P.radius;
    V[5] += fmin(fmax(cs[2], P.min_diag), P.max_diag) / P.radius;
    if (!inv3_spd_l(V, Vi, Li) && act) bad = true;
    if (act) {
#pragma unroll
        for (int i = 0; i < 6; ++i) P.Vinv[6 * p + i] = Vi[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) { P.bp[3 * p + i] = s21[6 + i]; P.colsq_p[3 * p + i] = cs[i]; }
        gmax = fmax(fabs(s21[6] * rcp_nr(sp[0])), fmax(fabs(s21[7] * rcp_nr(sp[1])), fabs(s21[8] * rcp_nr(sp[2]))));
    }
    // ---- pass 2: Z = J~' F~ with F~ = F L^-T, then the lower triangle of sum_p Z Z'
    {
        double Z[R][3];
#pragma unroll
        for (int k = 0; k < M; ++k) {
            ObsLin o;
            linearize(k, o);
            double Ft[2][3];                                        // F~[e][d] = sum_{c <= d} F[e][c] Li[d][c]
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                Ft[e][0] = o.F[e][0] * Li[0];
                Ft[e][1] = o.F[e][0] * Li[1] + o.F[e][1] * Li[2];
                Ft[e][2] = o.F[e][0] * Li[3] + o.F[e][1] * Li[4] + o.F[e][2] * Li[5];
            }
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int d = 0; d < 3; ++d) Z[6 * k + i][d] = o.Ec[0][i] * Ft[0][d] + o.Ec[1][i] * Ft[1][d];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {                               // E_K rows: W_K L^-T
            const double w0 = s21[9 + 3 * i], w1 = s21[10 + 3 * i], w2 = s21[11 + 3 * i];
            Z[6 * M + i][0] = w0 * Li[0];
            Z[6 * M + i][1] = w0 * Li[1] + w1 * Li[2];
            Z[6 * M + i][2] = w0 * Li[3] + w1 * Li[4] + w2 * Li[5];
        }
        Z[6 * M + 4][0] = s21[6] * Li[0];                           // r row: (L^-1 b)'
        Z[6 * M + 4][1] = s21[6] * Li[1] + s21[7] * Li[2];
        Z[6 * M + 4][2] = s21[6] * Li[3] + s21[7] * Li[4] + s21[8] * Li[5];
        run_zz_batches<R, NZC>(totz, Z, lane, std::make_integer_sequence<int, NZC>{});
    }
    // ---- the segment's totals over its four waves, fixed order
    cost = wave_sum(cost); gmax = wave_max(gmax);
    if (lane == 0) { xw[2 * wave] = cost; xw[2 * wave + 1] = gmax; }
    if (__any(bad) && lane == 0) *err = 1;
    double* mine = xs + wave * NV;
#pragma unroll
    for (int k = 0; k < M; ++k) {
        mine[66 * k + idx] = totd[k];
        if (idx < 2) mine[66 * k + 64 + idx] = totd2[k];
    }
#pragma unroll
    for (int c = 0; c < NZC; ++c) { const int v = 64 * c + idx; if (v < NZV) mine[66 * M + v] = totz[c]; }
    __syncthreads();
    for (int v = tid; v < NV; v += 256) part[(size_t)sg.tile_off + v] = ((xs[v] + xs[NV + v]) + xs[2 * NV + v]) + xs[3 * NV + v];
    if (tid == 0) {
        part_seg[2 * (size_t)blockIdx.x] = ((xw[0] + xw[2]) + xw[4]) + xw[6];
        part_seg[2 * (size_t)blockIdx.x + 1] = fmax(fmax(xw[1], xw[3]), fmax(xw[5], xw[7]));
    }
}

#define RUN_LDS_BYTES ((RUN_MMAX * TILE_CAMREC + 4 * run_nv(RUN_MMAX) + 16) * 8)

__global__ __launch_bounds__(256, 1) void ba_run_kernel(BADev P, const TileSeg* __restrict__ segs, const int* __restrict__ seg_cams,
                                                        double* __restrict__ part, double* __restrict__ part_seg, int* __restrict__ err)
{
    extern __shared__ __attribute__((aligned(16))) double run_lds[];
    const TileSeg sg = segs[blockIdx.x];
    switch (sg.M) {
    case 1: ba_run_body<1>(P, sg, seg_cams, part, part_seg, err, run_lds); break;
    case 2: ba_run_body<2>(P, sg, seg_cams, part, part_seg, err, run_lds); break;
    case 3: ba_run_body<3>(P, sg, seg_cams, part, part_seg, err, run_lds); break;
    case 4: ba_run_body<4>(P, sg, seg_cams, part, part_seg, err, run_lds); break;
    case 5: ba_run_body<5>(P, sg, seg_cams, part, part_seg, err, run_lds); break;
    case 6: ba_run_body<6>(P, sg, seg_cams, part, part_seg, err, run_lds); break;
    default: break;
    }
}
