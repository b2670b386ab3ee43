// ba_tiles.hpp -- the linearisation + Schur elimination as "run tiles" (included by ba.hip only).
//
// Same arithmetic as K_pt / K_cam / K_schur of ba_kernels.hpp (ReprojectCost NView:151-183, HuberLoss(4) NView:1184, the
// point elimination Ceres' SPARSE_SCHUR does for bundle_adjustment NView:1215-1224), organised so that every observation
// is linearised ONCE per pass and no per-point block is gathered again:
//
//   * points are stored sorted by their camera list, so a RUN of consecutive points sees the same cameras c_0..c_{M-1};
//     a run is cut into segments (one workgroup each), a segment into batches of 16 points (dealt to the four waves);
//   * lane (pl, og) = (point of the batch, observation og, og + 4) linearises its observation(s); the four lanes of a
//     point sum V_p = sum F'F + D, b_p = sum F'r, W_K = sum E_K'F across the lane groups and invert V_p = L L';
//   * with F~ = F L^-T the eliminated point contributes the rank-3 term -Z Z', Z = J~' F~, rows = [E_c0 | .. | E_cM-1 | E_K | r]
//     (6M + 5 of them), and each observation the rank-2 term +J~_k' J~_k on the rows [E_ck | E_K | r] (11).  Summed over
//     the points of a run both are symmetric rank-k updates: they go through v_mfma_f64_16x16x4_f64 with the contraction
//     index running over (point, coordinate) resp. (point, residual row) -- the matrix pipe does the cross-lane
//     reduction that the per-observation kernels pay for with shuffles and partial buffers.  fp64 MFMA has the VALU's
//     rate on this part: it is used as the reducer, not for throughput.
//   * operands are transposed through a per-wave LDS buffer (thread-per-observation layout in, MFMA layout out);
//   * the segment's tiles (M "direct" 16x16 tiles + the lower tiles of the (6M+5)^2 product) are summed over its four
//     waves in a fixed order and stored; ba_tile_reduce_kernel folds them into [S | rhs | diagU | graw] through a table
//     built with the solver layout (fixed order => run-to-run identical).
#pragma once
#include "ba_kernels.hpp"

typedef double tile_v4d __attribute__((ext_vector_type(4)));

#define TILE_MMAX 7            // observations per point the tile path takes: 6M + 5 <= 48 rows = 3 row tiles
#define TILE_CAMREC 32         // doubles per staged camera: rotation block (20) | t (3) | column scales (6) | free flag | pad
#define TILE_WAVE_LDS 2352     // doubles per wave: 48 contraction rows x 49 (Z operands, 3 row tiles) >= 4 x 32 rows x 17 (direct operands)
#define TILE_JLD 17
#define TILE_LDS_BYTES ((TILE_MMAX * TILE_CAMREC + 4 * TILE_WAVE_LDS + 16) * 8)

struct TileSeg { int p0, npts, obs0, M, cams_off, tile_off, pad0, pad1; };      // obs0 = first observation of point p0 (M per point, camera-sorted)

// index of element (row, col) of a stored 16x16 tile: D[row = (l >> 4) + 4 g][col = l & 15] sits in register g of lane l
__host__ __device__ inline int tile_elem(int row, int col) { return (row >> 2) * 64 + (row & 3) * 16 + col; }

// symmetric 3x3: V = L L', returns L^-1 (Li = [i00 i10 i11 i20 i21 i22]) and V^-1 (lower-packed like inv3_spd)
__device__ __forceinline__ bool inv3_spd_l(const double V[6], double Vi[6], double Li[6])
{
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
    Li[0] = i00; Li[1] = i10; Li[2] = i11; Li[3] = i20; Li[4] = i21; Li[5] = i22;
    Vi[0] = i00 * i00 + i10 * i10 + i20 * i20;
    Vi[1] = i10 * i11 + i20 * i21;
    Vi[2] = i11 * i11 + i21 * i21;
    Vi[3] = i20 * i22;
    Vi[4] = i21 * i22;
    Vi[5] = i22 * i22;
    return ok;
}

// Sums over the four lanes {pl, pl + 16, pl + 32, pl + 48} of a point (one per 16-lane row) with v_permlane{16,32}_swap: VALU
// only, where ds_bpermute pairs cost ~3.2k cycles per batch on the LDS pipe.  group_sum4: the total in all four lanes (same
// bits in each); group_scatter4: four values in, lane group g gets the total of value g.
__device__ __forceinline__ double group_sum4(double x)
{
    double y = x;
    lane_swap16(x, y); x += y;             // rows (0,1) and (2,3) hold their pair's sum
    y = x;
    lane_swap32(x, y);                      // x = [lower | lower], y = [upper | upper]
    return x + y;
}
__device__ __forceinline__ double group_scatter4(double a0, double a1, double a2, double a3)
{
    lane_swap16(a0, a1); const double s01 = a0 + a1;      // rows: a0(0,1), a1(0,1), a0(2,3), a1(2,3)
    lane_swap16(a2, a3); const double s23 = a2 + a3;
    double u = s01, v = s23;
    lane_swap32(u, v);                                    // u = [s01 rows 0,1 | s23 rows 0,1], v = [s01 rows 2,3 | s23 rows 2,3]
    return u + v;                                         // row g: total of a_g
}

template <int M>
__device__ __forceinline__ void ba_tile_body(const BADev& P, const TileSeg sg, const int* __restrict__ seg_cams,
                                             double* __restrict__ part, double* __restrict__ part_seg, int* __restrict__ err, double* lds)
{
    constexpr int R = 6 * M + 5, T = (R + 15) / 16, NZ = T * (T + 1) / 2, NK = (M + 3) / 4, ZLD = 16 * T + 1, NT = M + NZ;
    static_assert(T <= 3 && 48 * ZLD <= TILE_WAVE_LDS && 4 * 32 * TILE_JLD <= TILE_WAVE_LDS, "staging buffer");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pl = lane & 15, og = lane >> 4;
    double* cam = lds;                                                      // [M][TILE_CAMREC], shared by the workgroup
    double* buf = lds + TILE_MMAX * TILE_CAMREC + wave * TILE_WAVE_LDS;     // this wave's operand staging
    double* xw = lds + TILE_MMAX * TILE_CAMREC + 4 * TILE_WAVE_LDS;         // [4][2] cost / gmax of the waves
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
    tile_v4d dacc[M], zacc[NZ];
#pragma unroll
    for (int i = 0; i < M; ++i) dacc[i] = tile_v4d{ 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
    for (int i = 0; i < NZ; ++i) zacc[i] = tile_v4d{ 0.0, 0.0, 0.0, 0.0 };
    double cost = 0.0, gmax = 0.0;
    bool bad = false;
    const int nbatch = (sg.npts + 15) >> 4;
    const int orow = lane >> 4, ocol = lane & 15;                            // MFMA operand maps: A[i = l & 15][k = l >> 4] = B[k][j = l & 15]
    // the batch's inputs are requested one batch ahead (a wave has little else to hide the ~2 us of a first touch behind)
    double Xn[3], spn[3], uvn[NK][2];
    auto request = [&](int b) {
        const int pi = 16 * b + pl;
        const int pic = pi < sg.npts ? pi : sg.npts - 1;
        const size_t p = (size_t)sg.p0 + pic;
#pragma unroll
        for (int i = 0; i < 3; ++i) { Xn[i] = P.pts[3 * p + i]; spn[i] = P.scale_p[3 * p + i]; }
#pragma unroll
        for (int j = 0; j < NK; ++j) {
            const int k = og + 4 * j, kk = k < M ? k : M - 1;
            const size_t q = (size_t)sg.obs0 + (size_t)pic * M + kk;
            uvn[j][0] = P.ouv[2 * q]; uvn[j][1] = P.ouv[2 * q + 1];
        }
    };
    if (wave < nbatch) request(wave);
    for (int b = wave; b < nbatch; b += 4) {
        const int pi = 16 * b + pl;
        const bool act = pi < sg.npts;
        const int pic = act ? pi : sg.npts - 1;
        const int p = sg.p0 + pic;
        const double X[3] = { Xn[0], Xn[1], Xn[2] }, sp[3] = { spn[0], spn[1], spn[2] };
        double uv[NK][2];
#pragma unroll
        for (int j = 0; j < NK; ++j) { uv[j][0] = uvn[j][0]; uv[j][1] = uvn[j][1]; }
        if (b + 4 < nbatch) request(b + 4);
        // one observation of this lane: slot j -> observation k = og + 4 j of the point (all zero if it does not exist)
        auto linearize = [&](int j, ObsLin& o) -> bool {
            const int k = og + 4 * j;
            const bool has = act && k < M;
            const int kk = k < M ? k : M - 1;
            const double* rec = cam + kk * TILE_CAMREC;
            obs_linearize(P.K, rec, rec + 20, X, uv[j][0], uv[j][1], P.huber_a, sK, rec[29] != 0.0 ? rec + 23 : nullptr, sp, o);
            if (!has) {
                o.rho0 = 0.0; o.r[0] = o.r[1] = 0.0;
#pragma unroll
                for (int i = 0; i < 4; ++i) { o.EK[0][i] = 0.0; o.EK[1][i] = 0.0; }
#pragma unroll
                for (int i = 0; i < 6; ++i) { o.Ec[0][i] = 0.0; o.Ec[1][i] = 0.0; }
#pragma unroll
                for (int i = 0; i < 3; ++i) { o.F[0][i] = 0.0; o.F[1][i] = 0.0; }
            }
            return has;
        };
        ObsLin o0;                                                           // kept across the two phases when the lane has one observation
        double s21[21];                                                      // V (6) | b (3) | W_K (12): this lane's observations
#pragma unroll
        for (int i = 0; i < 21; ++i) s21[i] = 0.0;
#pragma unroll
        for (int j = 0; j < NK; ++j) {
            ObsLin oj;
            ObsLin& o = (NK == 1) ? o0 : oj;
            linearize(j, o);
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
            // direct operands: contraction row (residual row e, point pl) of observation slot og, columns [E_c | E_K | r | 0..]
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                double* row = buf + ((og * 32) + e * 16 + pl) * TILE_JLD;
#pragma unroll
                for (int i = 0; i < 6; ++i) row[i] = o.Ec[e][i];
#pragma unroll
                for (int i = 0; i < 4; ++i) row[6 + i] = o.EK[e][i];
                row[10] = o.r[e];          // columns 11..15 keep whatever the buffer held: they only reach rows / columns >= 11 of the tile, which nothing reads
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q + 4 * j < M) {
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        const double a = buf[((q * 32) + 4 * g + orow) * TILE_JLD + ocol];
                        dacc[q + 4 * j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, dacc[q + 4 * j], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        // the point blocks: totals over the point's lanes
#pragma unroll
        for (int i = 0; i < 9; ++i) s21[i] = group_sum4(s21[i]);
        double wk[3];                        // W_K row og (lane group og forms row og of Z's intrinsic rows)
#pragma unroll
        for (int d = 0; d < 3; ++d) wk[d] = group_scatter4(s21[9 + d], s21[12 + d], s21[15 + d], s21[18 + d]);
        double V[6], Vi[6], Li[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) V[i] = s21[i];
        const double cs[3] = { V[0], V[2], V[5] };
        V[0] += fmin(fmax(cs[0], P.min_diag), P.max_diag) / P.radius;
        V[2] += fmin(fmax(cs[1], P.min_diag), P.max_diag) / P.radius;
        V[5] += fmin(fmax(cs[2], P.min_diag), P.max_diag) / P.radius;
        if (!inv3_spd_l(V, Vi, Li) && act) bad = true;
        if (og == 0 && act) {
#pragma unroll
            for (int i = 0; i < 6; ++i) P.Vinv[6 * (size_t)p + i] = Vi[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) { P.bp[3 * (size_t)p + i] = s21[6 + i]; P.colsq_p[3 * (size_t)p + i] = cs[i]; }
            gmax = fmax(gmax, fmax(fabs(s21[6] * rcp_nr(sp[0])), fmax(fabs(s21[7] * rcp_nr(sp[1])), fabs(s21[8] * rcp_nr(sp[2])))));
        }
        // Z operands: contraction row (coordinate d, point pl), columns = the 6M + 5 rows of Z, zero up to 16 T
#pragma unroll
        for (int j = 0; j < NK; ++j) {
            ObsLin oj;
            if (NK > 1) linearize(j, oj);
            const ObsLin& o = (NK == 1) ? o0 : oj;
            const int k = og + 4 * j;
            if (k < M) {
                double Ft[2][3];                                            // F~ = F L^-T: F~[e][d] = sum_{c <= d} F[e][c] Li[d][c]
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    Ft[e][0] = o.F[e][0] * Li[0];
                    Ft[e][1] = o.F[e][0] * Li[1] + o.F[e][1] * Li[2];
                    Ft[e][2] = o.F[e][0] * Li[3] + o.F[e][1] * Li[4] + o.F[e][2] * Li[5];
                }
#pragma unroll
                for (int d = 0; d < 3; ++d)
#pragma unroll
                    for (int i = 0; i < 6; ++i) buf[(d * 16 + pl) * ZLD + 6 * k + i] = o.Ec[0][i] * Ft[0][d] + o.Ec[1][i] * Ft[1][d];
            }
        }
        {
            // E_K row og: (W_K L^-T)[og]; r row (lane group 1): (L^-1 b)'.  Columns >= 6M + 5 keep stale data (see above).
            const double w0 = act ? wk[0] : 0.0, w1 = act ? wk[1] : 0.0, w2 = act ? wk[2] : 0.0;
            buf[(0 * 16 + pl) * ZLD + 6 * M + og] = w0 * Li[0];
            buf[(1 * 16 + pl) * ZLD + 6 * M + og] = w0 * Li[1] + w1 * Li[2];
            buf[(2 * 16 + pl) * ZLD + 6 * M + og] = w0 * Li[3] + w1 * Li[4] + w2 * Li[5];
            if (og == 1) {
                const double b0 = act ? s21[6] : 0.0, b1 = act ? s21[7] : 0.0, b2 = act ? s21[8] : 0.0;
                buf[(0 * 16 + pl) * ZLD + 6 * M + 4] = b0 * Li[0];
                buf[(1 * 16 + pl) * ZLD + 6 * M + 4] = b0 * Li[1] + b1 * Li[2];
                buf[(2 * 16 + pl) * ZLD + 6 * M + 4] = b0 * Li[3] + b1 * Li[4] + b2 * Li[5];
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int g = 0; g < 12; ++g) {
            double a[T];
#pragma unroll
            for (int t = 0; t < T; ++t) a[t] = buf[(4 * g + orow) * ZLD + 16 * t + ocol];
#pragma unroll
            for (int tr = 0; tr < T; ++tr)
#pragma unroll
                for (int tc = 0; tc <= tr; ++tc)
                    zacc[tr * (tr + 1) / 2 + tc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tr], a[tc], zacc[tr * (tr + 1) / 2 + tc], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // ---- the segment's totals: waves 1..3 hand their tiles to wave 0 through the staging buffers (<= 9 tiles per round)
    cost = wave_sum(cost); gmax = wave_max(gmax);
    if (lane == 0) { xw[2 * wave] = cost; xw[2 * wave + 1] = gmax; }
    if (__any(bad) && lane == 0) *err = 1;
    constexpr int PER = TILE_WAVE_LDS / 256;
#pragma unroll
    for (int t0 = 0; t0 < NT; t0 += PER) {
        __syncthreads();
        if (wave > 0) {
#pragma unroll
            for (int t = t0; t < NT && t < t0 + PER; ++t) {
                const tile_v4d v = t < M ? dacc[t < M ? t : 0] : zacc[t >= M ? t - M : 0];
#pragma unroll
                for (int g = 0; g < 4; ++g) buf[(t - t0) * 256 + g * 64 + lane] = v[g];
            }
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int t = t0; t < NT && t < t0 + PER; ++t) {
                const tile_v4d v = t < M ? dacc[t < M ? t : 0] : zacc[t >= M ? t - M : 0];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    double s = v[g];
#pragma unroll
                    for (int w = 1; w < 4; ++w) s += buf[w * TILE_WAVE_LDS + (t - t0) * 256 + g * 64 + lane];
                    part[((size_t)sg.tile_off + t) * 256 + g * 64 + lane] = s;
                }
            }
        }
    }
    if (tid == 0) {
        part_seg[2 * (size_t)blockIdx.x] = ((xw[0] + xw[2]) + xw[4]) + xw[6];
        part_seg[2 * (size_t)blockIdx.x + 1] = fmax(fmax(xw[1], xw[3]), fmax(xw[5], xw[7]));
    }
}

__global__ __launch_bounds__(256, 2) void ba_tile_kernel(BADev P, const TileSeg* __restrict__ segs, const int* __restrict__ seg_cams,
                                                         double* __restrict__ part, double* __restrict__ part_seg, int* __restrict__ err)
{
    extern __shared__ __attribute__((aligned(16))) double tile_lds[];
    const TileSeg sg = segs[blockIdx.x];
    switch (sg.M) {
    case 1: ba_tile_body<1>(P, sg, seg_cams, part, part_seg, err, tile_lds); break;
    case 2: ba_tile_body<2>(P, sg, seg_cams, part, part_seg, err, tile_lds); break;
    case 3: ba_tile_body<3>(P, sg, seg_cams, part, part_seg, err, tile_lds); break;
    case 4: ba_tile_body<4>(P, sg, seg_cams, part, part_seg, err, tile_lds); break;
    case 5: ba_tile_body<5>(P, sg, seg_cams, part, part_seg, err, tile_lds); break;
    case 6: ba_tile_body<6>(P, sg, seg_cams, part, part_seg, err, tile_lds); break;
    case 7: ba_tile_body<7>(P, sg, seg_cams, part, part_seg, err, tile_lds); break;
    default: break;
    }
}

// dst[i] (and its mirror dst2[i] >= 0) = sum of +-part[src] over the destination's source list (bit 31 of src: subtract), in a
// fixed order.  The first n_long destinations (the intrinsic block: every segment contributes) take a workgroup each --
// thread-strided partial sums, then a fixed tree -- the others a thread each; the last workgroup folds the segments' cost /
// max-gradient partials into scal.
__device__ __forceinline__ double tile_src_value(const double* __restrict__ part, unsigned e)
{
    const double v = part[e & 0x7fffffffu];
    return (e >> 31) ? -v : v;
}
__global__ __launch_bounds__(256) void ba_tile_reduce_kernel(BADev P, const int* __restrict__ rd_start, const int* __restrict__ rd_dst,
                                                             const int* __restrict__ rd_dst2, const unsigned* __restrict__ rd_src, int n_long, int nd,
                                                             const double* __restrict__ part, const double* __restrict__ part_seg, int nseg)
{
    __shared__ double red[4][2];
    const int tid = threadIdx.x;
    if (blockIdx.x + 1 == gridDim.x) {
        double c = 0.0, g = 0.0;
        for (int s = tid; s < nseg; s += 256) { c += part_seg[2 * (size_t)s]; g = fmax(g, part_seg[2 * (size_t)s + 1]); }
        c = wave_sum(c); g = wave_max(g);
        if ((tid & 63) == 0) { red[tid >> 6][0] = c; red[tid >> 6][1] = g; }
        __syncthreads();
        if (tid == 0) {
            P.scal[SCAL_COST] = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0];
            P.scal[SCAL_GMAX_SLOTS + P.rank] = fmax(fmax(red[0][1], red[1][1]), fmax(red[2][1], red[3][1]));
        }
        for (int i = tid; i < P.npad; i += 256)          // padding slots of the message tail (see ba_finalize_role)
            if (!P.posmask[i]) { P.rhs[i] = 0.0; P.diagU[i] = 0.0; P.graw[i] = 0.0; }
        return;
    }
    if ((int)blockIdx.x < n_long) {
        const int i = blockIdx.x;
        double s = 0.0;
        for (int q = rd_start[i] + tid; q < rd_start[i + 1]; q += 256) s += tile_src_value(part, rd_src[q]);
        s = wave_sum(s);
        if ((tid & 63) == 0) red[tid >> 6][0] = s;
        __syncthreads();
        if (tid == 0) {
            const double v = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0];
            P.S[rd_dst[i]] = v;
            if (rd_dst2[i] >= 0) P.S[rd_dst2[i]] = v;
        }
        return;
    }
    const int i = n_long + ((int)blockIdx.x - n_long) * 256 + tid;
    if (i >= nd) return;
    double s = 0.0;
    for (int q = rd_start[i]; q < rd_start[i + 1]; ++q) s += tile_src_value(part, rd_src[q]);
    P.S[rd_dst[i]] = s;
    if (rd_dst2[i] >= 0) P.S[rd_dst2[i]] = s;
}
