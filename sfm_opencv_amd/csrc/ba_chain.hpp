// ba_chain.hpp -- latency-first solver of the reduced camera system for camera CHAINS (round 4).
//
// Replaces, for the structure the reference produces, ceres::Solve's SPARSE_SCHUR back end (NViewReconstuct.cpp:1215-1224).
// In the reference tracks only chain through consecutive frames (NViewReconstuct.cpp:1289-1299), so the reduced system
//     S = [ band of 6x6 camera blocks, half width w cameras | 4 intrinsic rows ]        (+ the right-hand side)
// is block-banded with a dense border.  The level-per-launch multifrontal solver of ba_solver.hpp spends 25-37 us per
// tree level on 32-column panels, launches and dependent first-touch loads; nothing here is throughput-bound (C4: 1198
// unknowns).  This solver is built around the dependency chain instead:
//
//   * nested dissection of the chain into P = 2^m leaves and P - 1 separators of w cameras (same tree as before), but
//     every FRONT lives in LDS from assembly to its last column, and columns go six at a time (one camera): the 6x6
//     pivot block is factored redundantly by every lane of ONE wave in registers (no cross-lane traffic on the chain),
//     the panel rows X = B L^-T one lane per row, the trailing update A -= X X' as 3x3 register blocks over the lower
//     triangle (54 fma per lane and round);
//   * look-ahead: of the G waves that share a front, wave 0 updates the NEXT pivot block first and factors it while the
//     others finish the trailing update; one workgroup barrier per camera;
//   * a leaf: a sliding window of w + 1 cameras (a ring of w + 2 LDS slots: the camera that enters never shares a slot with
//     the panel still being read) + the border [left separator | K | rhs]; cameras enter from S (natural layout, the row
//     block of the camera) two steps ahead of their use;
//   * a separator: its children ADD their remaining blocks straight into the parent's front layout (block copies, the right
//     child's border blocks transposed), no lists, no global round trip inside a workgroup;
//   * the factor records hold W = L^-T X' and L^-T z, so the back-substitution is y_c = zt_c - W_c y_rest: no triangular
//     solves; the part of y_rest outside the leaf / separator is folded for all its cameras at once, the rest is a short
//     serial loop over 8-lane groups;
//   * two launches: chain_sub_kernel (a workgroup = 2^a leaves + a levels of the tree; its root goes to HBM as an image of
//     the parent front) and chain_top_kernel (one workgroup: the remaining levels, the intrinsics, and the WHOLE
//     back-substitution, tree and leaves); small chains take the first kernel only.
//
// Everything is summed in a fixed order: results are run-to-run identical.
//
// The file compiles twice: with hipcc for gfx950, and with g++ -DCHAIN_HOST_EMU against a fiber emulation of a workgroup
// (tests/host/chain_emu_test.cpp) that runs the same index arithmetic on the CPU -- the solver is mostly index
// bookkeeping, and a GPU round trip costs minutes.
#pragma once

#ifdef CHAIN_HOST_EMU
#include <cmath>
#include <cstddef>
namespace chain_emu { int tid(); int bid(); void sync_wg(); void sync_wave(); double shfl_xor(double v, int mask); }
#define CH_DEV static inline
#define CH_HD static inline
#define CH_TID chain_emu::tid()
#define CH_BID chain_emu::bid()
#define CH_SYNC_WG() chain_emu::sync_wg()
#define CH_SYNC_WAVE() chain_emu::sync_wave()
#define CH_UNROLL
CH_DEV double ch_rsqrt(double d) { return 1.0 / std::sqrt(d); }
CH_DEV double ch_fma(double a, double b, double c) { return std::fma(a, b, c); }
CH_DEV double ch_min(double a, double b) { return a < b ? a : b; }
CH_DEV double ch_max(double a, double b) { return a > b ? a : b; }
// sum over the eight lanes of a group (lanes 8 i .. 8 i + 7), every lane gets the total
CH_DEV double ch_sum8(double v) { v += chain_emu::shfl_xor(v, 1); v += chain_emu::shfl_xor(v, 2); v += chain_emu::shfl_xor(v, 4); return v; }
CH_DEV void ch_keep(double) { }
#else
#include <hip/hip_runtime.h>
#define CH_DEV __device__ __forceinline__
#define CH_HD __host__ __device__ __forceinline__
#define CH_TID ((int)threadIdx.x)
#define CH_BID ((int)blockIdx.x)
#define CH_SYNC_WG() __syncthreads()
// lanes of one wave exchange data through LDS: DS operations of a wave execute in order, only the compiler must not move them
#define CH_SYNC_WAVE() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#define CH_UNROLL _Pragma("unroll")
CH_DEV double ch_rsqrt(double d)
{
    double y = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
    double e = fma(-h * y, y, 0.5); y = fma(y, e, y);
    e = fma(-h * y, y, 0.5); y = fma(y, e, y);
    return y;
}
CH_DEV double ch_fma(double a, double b, double c) { return fma(a, b, c); }
CH_DEV double ch_min(double a, double b) { return fmin(a, b); }
CH_DEV double ch_max(double a, double b) { return fmax(a, b); }
template <int CTRL>
CH_DEV double ch_dpp(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror: the same pairing as xor 1, 2, 4
CH_DEV double ch_sum8(double v) { v += ch_dpp<0xB1>(v); v += ch_dpp<0x4E>(v); v += ch_dpp<0x141>(v); return v; }
CH_DEV void ch_keep(double v) { asm volatile("" ::"v"(v)); }
#endif

#if defined(CH_STAMPS) && !defined(CHAIN_HOST_EMU)
#ifndef CH_STAMP_TID
#define CH_STAMP_TID 0
#endif
#define CH_STAMP(A, si) do { if ((A).stamps && CH_BID == 0 && CH_TID == CH_STAMP_TID) (A).stamps[(si)] = (long long)__builtin_amdgcn_s_memtime(); ++(si); } while (0)
#else
#define CH_STAMP(A, si) do { } while (0)
#endif

#define CH_WMAX 3           // band half width (cameras) the register arrays are sized for (6 (3 w - 1) + 4 <= 64 record rows: a lane each)
#define CH_REC_HEAD 8       // record header: L^-T z (6) | pad; then W' rows, 6 doubles per remaining row
#define CH_NW 8             // waves of a workgroup of chain_sub_kernel at most (two per SIMD: 256 registers each)
#define CH_NW_TOP 16        // waves of chain_top_kernel (four per SIMD: 128 registers each)
#define CH_FSCR 128         // doubles behind a front's panel: next camera's panel rows (36) | its updated pivot block (22 + pad) | two factor slots (2 x 32)

struct ChainArgs {
    // geometry (chain_plan fills it)
    int ncf, w, nk, P, m, a, G;     // free cameras; band half width; 4 or 0 intrinsics; leaves = 2^m; tree levels inside chain_sub_kernel; waves per leaf
    int q, r;                       // leaf j has q + (j < r) cameras
    int koff, ld, n, npad;          // position of the intrinsics in S; leading dimension; unknowns; padded length of y
    int nbd, BB;                    // 6 w; border blocks (3 rows each): [left separator (6 w) | K (nk) | rhs | pad]
    int rec_stride, img_doubles;    // doubles per factor record / per exported front image
    int nw_top;                     // waves of chain_top_kernel
    // data
    const double* S; const double* rhs; const double* diagU;    // diagU == nullptr: S is damped already
    double inv_radius, dmin, dmax;  // LM damping: S_ii += clamp(diagU_i) * (1 / radius); diagU_i == 0 (no residual touches it): unit row
    double* rec; double* img; double* y; int* err;
    long long* stamps;              // measurement builds (experiments/chain_bench.hip): s_memtime at phase boundaries of workgroup 0, wave 0
};

CH_HD int ch_tri(int n) { return n * (n + 1) / 2; }
CH_HD int ch_leaf_rb(int w) { return 2 * (w + 2); }
CH_HD int ch_leaf_mat(int w, int BB) { const int RB = ch_leaf_rb(w); return 9 * (RB * RB + BB * RB + ch_tri(BB)); }
CH_HD int ch_node_mat(int w, int BB) { const int RB = 4 * w; return 9 * (ch_tri(RB) + BB * RB + ch_tri(BB)); }
CH_HD int ch_leaf_front(int w, int BB) { return ch_leaf_mat(w, BB) + 18 * (ch_leaf_rb(w) + BB) + CH_FSCR; }
CH_HD int ch_node_front(int w, int BB) { return ch_node_mat(w, BB) + 18 * (4 * w + BB) + CH_FSCR; }
CH_HD int ch_pair_doubles(int w) { return (ch_tri(6 * w) * 2 + 7) / 8; }
CH_HD int ch_back_scratch(int) { return 6 * 64; }        // per wave: the lanes' six partial products, transposed
CH_HD int ch_ysh_doubles(int n) { return (n + 8 + 7) & ~7; }
// dynamic LDS (doubles) of the two kernels
CH_HD size_t ch_sub_lds(int w, int BB, int a, int G, int n, bool single)
{
    const int NL = 1 << a;
    size_t fronts = (size_t)NL * ch_leaf_front(w, BB) + (size_t)(NL - 1) * ch_node_front(w, BB);
    const size_t back = (size_t)ch_ysh_doubles(n) + (size_t)NL * G * ch_back_scratch(w);
    if (single && fronts < back) fronts = back;         // the solution vector and the scratch take the fronts' place
    return (size_t)ch_pair_doubles(w) + fronts;
}
CH_HD size_t ch_top_lds(int w, int BB, int levels, int nw, int n)
{
    size_t fronts = (size_t)((1 << levels) - 1) * ch_node_front(w, BB);
    const size_t back = (size_t)ch_ysh_doubles(n) + (size_t)nw * ch_back_scratch(w);
    if (fronts < back) fronts = back;
    return (size_t)ch_pair_doubles(w) + fronts;
}

// ---- geometry of the dissection ------------------------------------------------------------------------------------
// chain = leaf 0 | sep 0 | leaf 1 | sep 1 | ... | leaf P-1.  Level of separator i: ctz(i + 1) + 1; node k of level l is
// separator (2k + 1) 2^(l-1) - 1, its children are nodes 2k, 2k + 1 of level l - 1 (leaves 2k, 2k + 1 for l = 1), its outer
// neighbours ("LA", "RA") separators i -/+ 2^(l-1).
CH_HD int ch_leaf_lo(const ChainArgs& A, int j) { return j * (A.q + A.w) + (j < A.r ? j : A.r); }
CH_HD int ch_leaf_len(const ChainArgs& A, int j) { return A.q + (j < A.r ? 1 : 0); }
CH_HD int ch_sep_lo(const ChainArgs& A, int i) { return ch_leaf_lo(A, i) + ch_leaf_len(A, i); }

// x mod m for 0 <= x < 2^20, 1 <= m <= 8 without an integer division
CH_DEV int ch_mod(int x, int m) { return x - m * (int)(((float)x + 0.5f) * (1.0f / (float)m)); }

// ---- a front in LDS -------------------------------------------------------------------------------------------------
// Rows: a RING of camera slots (two 3-row blocks each; a leaf's window wraps around, a separator's [own | RA] does not) and
// the BORDER [LA (6 w) | K | rhs | pad].  Stored as 3x3 blocks, row-major inside: ww = ring x ring (square for leaves;
// packed lower for separators, whose ring order is the slot order), bw = border x ring, bb = border x border (packed lower).
// Of a pair of rows the LATER one in elimination order is the block row (ring before border, ring cameras in chain order);
// diagonal blocks keep both triangles.  xp = the current panel X, 6 doubles per remaining row; pn / pq = the pivot wave's
// scratch (panel rows of the next camera, its updated pivot block); lb = two slots for the factor of the next pivot block
// (L 15 | 1 / L_jj 6), written by the pivot wave one step ahead.
struct ChFront { double *ww, *bw, *bb, *xp, *pn, *pq, *lb; int RB, packed; };
CH_DEV ChFront ch_front_at(double* base, int RB, int BB, int packed)
{
    ChFront F; F.RB = RB; F.packed = packed;
    F.ww = base; F.bw = base + 9 * (packed ? ch_tri(RB) : RB * RB); F.bb = F.bw + 9 * BB * RB; F.xp = F.bb + 9 * ch_tri(BB); F.pn = F.xp + 18 * (RB + BB); F.pq = F.pn + 36; F.lb = F.pn + 64;
    return F;
}
CH_DEV int ch_ww_off(const ChFront& F, int pi, int pj) { return 9 * (F.packed ? ch_tri(pi) + pj : pi * F.RB + pj); }
CH_DEV int ch_bw_off(const ChFront& F, int b, int pj) { return (int)(F.bw - F.ww) + 9 * (b * F.RB + pj); }
CH_DEV int ch_bb_off(const ChFront& F, int b1, int b2) { return (int)(F.bb - F.ww) + 9 * (ch_tri(b1) + b2); }
// block (bi >= bj) of the REMAINING rows when the ring head sits at block h0 and nrr2 ring blocks remain behind it
CH_DEV int ch_rem_off(const ChFront& F, int h0, int nrr2, int bi, int bj)
{
    if (bi < nrr2) {
        int pi = h0 + 2 + bi; if (pi >= F.RB) pi -= F.RB;
        int pj = h0 + 2 + bj; if (pj >= F.RB) pj -= F.RB;
        return ch_ww_off(F, pi, pj);
    }
    if (bj < nrr2) { int pj = h0 + 2 + bj; if (pj >= F.RB) pj -= F.RB; return ch_bw_off(F, bi - nrr2, pj); }
    return ch_bb_off(F, bi - nrr2, bj - nrr2);
}

CH_DEV double ch_damped(const ChainArgs& A, double v, int idx)
{
    if (!A.diagU) return v;
    const double du = A.diagU[idx];
    return du > 0.0 ? ch_fma(ch_min(ch_max(du, A.dmin), A.dmax), A.inv_radius, v) : 1.0;
}

// ---- the 6x6 pivot block: Cholesky in registers, every lane the same values --------------------------------------------
struct ChPivot { double L[15], yv[6]; };      // L strictly lower, row-major packed ((i, j) at i (i - 1) / 2 + j); yv = 1 / L_jj
// a: the lower triangle, row-major packed ((i, j) at i (i + 1) / 2 + j).  A non-positive (or NaN) pivot turns every later
// pivot into NaN, so ONE test of the last reciprocal root tells: 24 compares less on the chain every camera waits for.
CH_DEV bool ch_factor6(double (&a)[21], ChPivot& P)
{
    CH_UNROLL
    for (int j = 0; j < 6; ++j) {
        const double y = ch_rsqrt(a[j * (j + 1) / 2 + j]);
        P.yv[j] = y;
        CH_UNROLL
        for (int i = j + 1; i < 6; ++i) P.L[i * (i - 1) / 2 + j] = a[i * (i + 1) / 2 + j] * y;
        CH_UNROLL
        for (int i = j + 1; i < 6; ++i)
            CH_UNROLL
            for (int k = j + 1; k <= i; ++k) a[i * (i + 1) / 2 + k] = ch_fma(-P.L[i * (i - 1) / 2 + j], P.L[k * (k - 1) / 2 + j], a[i * (i + 1) / 2 + k]);
    }
    double t = P.yv[0];
    CH_UNROLL
    for (int j = 1; j < 6; ++j) t = ch_min(t, P.yv[j]);         // min keeps a NaN out only if the other operand hides it: test the sum too
    const double sum = ((P.yv[0] + P.yv[1]) + (P.yv[2] + P.yv[3])) + (P.yv[4] + P.yv[5]);
    return (t > 0.0) && (sum < 1e300);
}
CH_DEV bool ch_factor6_blocks(const double* d00, const double* d10, const double* d11, ChPivot& P)
{
    double a[21];
    a[0] = d00[0]; a[1] = d00[3]; a[2] = d00[4]; a[3] = d00[6]; a[4] = d00[7]; a[5] = d00[8];
    a[6] = d10[0]; a[7] = d10[1]; a[8] = d10[2]; a[9] = d11[0];
    a[10] = d10[3]; a[11] = d10[4]; a[12] = d10[5]; a[13] = d11[3]; a[14] = d11[4];
    a[15] = d10[6]; a[16] = d10[7]; a[17] = d10[8]; a[18] = d11[6]; a[19] = d11[7]; a[20] = d11[8];
    return ch_factor6(a, P);
}

CH_DEV void ch_update_pair(const ChFront& F, const unsigned short* pairs, int h0, int nrr2, int q)
{
    const unsigned pr = pairs[q];
    const int bi = (int)(pr & 255u), bj = (int)(pr >> 8);
    double* blk = F.ww + ch_rem_off(F, h0, nrr2, bi, bj);
    const double* xi = F.xp + 18 * bi; const double* xj = F.xp + 18 * bj;
    double xa[18], v[9];
    CH_UNROLL
    for (int k = 0; k < 18; ++k) xa[k] = xi[k];
    CH_UNROLL
    for (int k = 0; k < 9; ++k) v[k] = blk[k];
    CH_UNROLL
    for (int c = 0; c < 3; ++c) {
        double xb[6];
        CH_UNROLL
        for (int k = 0; k < 6; ++k) xb[k] = xj[6 * c + k];
        CH_UNROLL
        for (int r = 0; r < 3; ++r) {
            double t = v[3 * r + c];
            CH_UNROLL
            for (int k = 0; k < 6; ++k) t = ch_fma(-xa[6 * r + k], xb[k], t);
            v[3 * r + c] = t;
        }
    }
    CH_UNROLL
    for (int k = 0; k < 9; ++k) blk[k] = v[k];
}

// x L' = b for one panel row
CH_DEV void ch_row_solve(const ChPivot& P, const double* s0, const double* s1, double (&x)[6])
{
    const double b[6] = { s0[0], s0[1], s0[2], s1[0], s1[1], s1[2] };
    CH_UNROLL
    for (int c = 0; c < 6; ++c) {
        double t = b[c];
        CH_UNROLL
        for (int k = 0; k < c; ++k) t = ch_fma(-x[k], P.L[c * (c - 1) / 2 + k], t);
        x[c] = t * P.yv[c];
    }
}

// ---- one elimination step: the camera at the ring head (blocks h0, h0 + 1) ---------------------------------------------
// nrr ring cameras remain behind it.  G waves share the front (wig = this wave's number among them).  A lone wave issues an
// instruction every 10-15 cycles (dependent fp64 operations, LDS round trips), so the step is cut by ROLE, each role as few
// instructions as possible:
//   wave 0 ("pivot wave", when the next camera is eliminated here too: factor_next) holds this camera's factor in registers
//          (P; first step: everyone factors the block straight from the front).  It solves only the six panel rows of the NEXT
//          camera, forms that camera's updated pivot block (21 lanes, one element each, from the front + those rows), factors it
//          into P and leaves a copy in lb[parity ^ 1] for the others' next step.  The three block pairs of that pivot block are
//          nobody else's business: after the next step the block is gone.
//   the others take the factor from lb[parity], form the whole panel X = B L^-T, one lane per remaining row, into the shared xp
//          (same values from every wave, so no wave waits for another before its update), and share the block pairs of the
//          trailing update (waves 1 .. G-2, and G-1 too when G <= 3);
//   wave G - 1 also forms W' = X L^-1, the record of the back-substitution (y_c = zt - W y_rest), and -- the caller -- brings
//          the next camera into a leaf's window.
// With G == 1 the one wave does all of it.  The caller synchronises the G waves afterwards.  Returns false on a non-positive pivot.
CH_DEV bool ch_step(const ChainArgs& A, const ChFront& F, const unsigned short* pairs, int h0, int nrr, int cam, bool first, bool factor_next,
                    int G, int wig, int parity, int lane, ChPivot& P, bool reload, int& si)
{
    const int nrr2 = 2 * nrr, nrb = nrr2 + A.BB, nrows = 3 * nrb, h1 = h0 + 1;
    bool ok = true;
    CH_STAMP(A, si);
    if (first) ok = ch_factor6_blocks(F.ww + ch_ww_off(F, h0, h0), F.ww + ch_ww_off(F, h1, h0), F.ww + ch_ww_off(F, h1, h1), P);
    else if (wig != 0 || reload) {
        const double* lb = F.lb + 32 * parity;
        CH_UNROLL
        for (int k = 0; k < 15; ++k) P.L[k] = lb[k];
        CH_UNROLL
        for (int k = 0; k < 6; ++k) P.yv[k] = lb[15 + k];
    }
    CH_STAMP(A, si);
    const bool look = nrr > 0 && factor_next;
    const int npairs = ch_tri(nrb), q0 = look ? 3 : 0;
    if (wig != 0 || G == 1) {
        double* rc = A.rec + (size_t)cam * A.rec_stride;
        const int nx = 6 * nrr + A.nbd + A.nk;      // rows that go to the record; row nx is the right-hand side (-> L^-T z), the rest padding
        for (int r = lane; r < nrows; r += 64) {
            const int lb = r / 3, rr = r - 3 * lb;
            const double *s0, *s1;
            if (lb < nrr2) { int p = h0 + 2 + lb; if (p >= F.RB) p -= F.RB; s0 = F.ww + ch_ww_off(F, p, h0) + 3 * rr; s1 = F.ww + ch_ww_off(F, p, h1) + 3 * rr; }
            else { s0 = F.ww + ch_bw_off(F, lb - nrr2, h0) + 3 * rr; s1 = F.ww + ch_bw_off(F, lb - nrr2, h1) + 3 * rr; }
            double x[6];
            ch_row_solve(P, s0, s1, x);
            double* xo = F.xp + 6 * r;
            CH_UNROLL
            for (int c = 0; c < 6; ++c) xo[c] = x[c];
            if (wig == G - 1 && r <= nx) {
                double wv[6];
                CH_UNROLL
                for (int k = 5; k >= 0; --k) {      // w L = x
                    double t = x[k];
                    CH_UNROLL
                    for (int mm = k + 1; mm < 6; ++mm) t = ch_fma(-wv[mm], P.L[mm * (mm - 1) / 2 + k], t);
                    wv[k] = t * P.yv[k];
                }
                double* o = r < nx ? rc + CH_REC_HEAD + 6 * r : rc;
                CH_UNROLL
                for (int k = 0; k < 6; ++k) o[k] = wv[k];
            }
        }
        CH_SYNC_WAVE();
    }
    CH_STAMP(A, si);
    if (wig == 0 && look) {
        int n0 = h0 + 2; if (n0 >= F.RB) n0 -= F.RB;
        if (G > 1) {            // (a lone wave has just put these rows into xp)
            if (lane < 6) {
                const int lb = lane / 3, rr = lane - 3 * lb;
                double x[6];
                ch_row_solve(P, F.ww + ch_ww_off(F, n0 + lb, h0) + 3 * rr, F.ww + ch_ww_off(F, n0 + lb, h1) + 3 * rr, x);
                CH_UNROLL
                for (int c = 0; c < 6; ++c) F.pn[6 * lane + c] = x[c];
            }
            CH_SYNC_WAVE();
        }
        const double* xn = G > 1 ? F.pn : F.xp;
        if (lane < 21) {
            const int r = lane >= 15 ? 5 : lane >= 10 ? 4 : lane >= 6 ? 3 : lane >= 3 ? 2 : lane >= 1 ? 1 : 0, c = lane - r * (r + 1) / 2;
            double v = F.ww[ch_ww_off(F, n0 + r / 3, n0 + c / 3) + 3 * (r % 3) + c % 3];
            CH_UNROLL
            for (int k = 0; k < 6; ++k) v = ch_fma(-xn[6 * r + k], xn[6 * c + k], v);
            F.pq[lane] = v;
        }
        CH_SYNC_WAVE();
        double a[21];
        CH_UNROLL
        for (int k = 0; k < 21; ++k) a[k] = F.pq[k];
        ok = ch_factor6(a, P) && ok;
        if ((G > 1 || reload) && lane == 0) {
            double* lb = F.lb + 32 * (parity ^ 1);
            CH_UNROLL
            for (int k = 0; k < 15; ++k) lb[k] = P.L[k];
            CH_UNROLL
            for (int k = 0; k < 6; ++k) lb[15 + k] = P.yv[k];
        }
    }
    if (G == 1) {
        for (int q = q0 + lane; q < npairs; q += 64) ch_update_pair(F, pairs, h0, nrr2, q);
    } else if (wig != 0) {
        const int nu = G - 1;
        if (wig <= nu)
            for (int q = q0 + (wig - 1) * 64 + lane; q < npairs; q += 64 * nu) ch_update_pair(F, pairs, h0, nrr2, q);
    }
    CH_STAMP(A, si);
    return ok;
}

// ---- a camera entering a leaf's window --------------------------------------------------------------------------------
// Camera n takes ring slot sl behind `ne` earlier window cameras (n - ne .. n - 1, in the slots before it).  Lane (r = lane >> 3,
// j = lane & 7), r < 6:
//   v[0..3]   row r of the camera x columns j + 8 t of [earlier cameras | itself]: S[6 n + r][6 (n - ne) + c]; its own block damped,
//             kept where block row >= block column
//   v[4]      border rows K_j / rhs (j <= nk) x its column r:  S[6 n + r][koff + j] / rhs[6 n + r]
//   v[5]      diagU of the lane's diagonal element (-1: none): the damping is applied when the value is stored
//   v[6..8]   LA rows j + 8 t x its column r (only while the window is first filled; zero afterwards: the slot's previous
//             occupant left fill there)
// Cameras of the right separator (n >= hi) bring only their coupling to the leaf's own cameras: a leaf hands on its
// UPDATE of the separator, the separator's own entries are staged by the node that eliminates it.
struct ChLeafGeo { int lo, len, hi, ntot, la_lo; bool has_la, has_ra; };
CH_DEV ChLeafGeo ch_leaf_geo(const ChainArgs& A, int j)
{
    ChLeafGeo g;
    g.lo = ch_leaf_lo(A, j); g.len = ch_leaf_len(A, j); g.hi = g.lo + g.len;
    g.has_la = j > 0; g.has_ra = j < A.P - 1;
    g.la_lo = g.has_la ? ch_sep_lo(A, j - 1) : 0;
    g.ntot = g.len + (g.has_ra ? A.w : 0);
    return g;
}
template <int NV>
CH_DEV void ch_cam_load(const ChainArgs& A, const ChLeafGeo& g, int n, int ne, bool with_la, int lane, double (&v)[NV])
{
    const int r = lane >> 3, j = lane & 7;
    const bool is_ra = n >= g.hi, act = r < 6 && n < g.lo + g.ntot;
    const double* row = A.S + (size_t)(6 * n + r) * A.ld;
    CH_UNROLL
    for (int t = 0; t < 4; ++t) {
        const int c = j + 8 * t;
        v[t] = 0.0;
        if (act && c < 6 * (ne + 1)) {
            const int k = c / 6, cc = c - 6 * k;
            const bool own = k == ne;
            const bool need = own ? (!is_ra && r / 3 >= cc / 3) : !(is_ra && n - ne + k >= g.hi);
            if (need) v[t] = row[6 * (n - ne) + c];
        }
    }
    v[4] = 0.0;
    if (act && !is_ra && j <= A.nk) v[4] = j < A.nk ? row[A.koff + j] : A.rhs[6 * n + r];
    // the diagonal element this lane holds (column 6 ne + r of its row) is damped when it is stored: nothing here waits for a load
    v[5] = -1.0;
    if (act && !is_ra && A.diagU && ((6 * ne + r) & 7) == j) v[5] = A.diagU[6 * n + r];
    if (NV > 6) {
        CH_UNROLL
        for (int t = 0; t < 3; ++t) {
            const int br = j + 8 * t;
            v[NV > 6 ? 6 + t : 0] = 0.0;
            if (act && with_la && !is_ra && br < A.nbd) v[NV > 6 ? 6 + t : 0] = row[6 * g.la_lo + br];
        }
    }
}
template <int NV>
CH_DEV void ch_cam_store(const ChainArgs& A, const ChFront& F, const ChLeafGeo& g, int n, int ne, int sl, int lane, const double (&v)[NV])
{
    const int r = lane >> 3, j = lane & 7, RC = F.RB >> 1, b0 = 2 * sl;
    if (r >= 6 || n >= g.lo + g.ntot) return;
    CH_UNROLL
    for (int t = 0; t < 4; ++t) {
        const int c = j + 8 * t;
        if (c < 6 * (ne + 1)) {
            const int k = c / 6, cc = c - 6 * k;
            if (k == ne) {
                double x = v[t];
                if (r == cc && v[5] >= 0.0) x = v[5] > 0.0 ? ch_fma(ch_min(ch_max(v[5], A.dmin), A.dmax), A.inv_radius, x) : 1.0;
                if (r / 3 >= cc / 3) F.ww[ch_ww_off(F, b0 + r / 3, b0 + cc / 3) + 3 * (r % 3) + cc % 3] = x;
            }
            else { int so = sl - ne + k; if (so < 0) so += RC; F.ww[ch_ww_off(F, b0 + r / 3, 2 * so + cc / 3) + 3 * (r % 3) + cc % 3] = v[t]; }
        }
    }
    if (j <= A.nk) { const int br = A.nbd + j; F.ww[ch_bw_off(F, br / 3, b0 + r / 3) + 3 * (br % 3) + r % 3] = v[4]; }
    CH_UNROLL
    for (int t = 0; t < 3; ++t) {
        const int br = j + 8 * t;
        if (br < A.nbd) F.ww[ch_bw_off(F, br / 3, b0 + r / 3) + 3 * (br % 3) + r % 3] = NV > 6 ? v[NV > 6 ? 6 + t : 0] : 0.0;
    }
}

// the intrinsics' own entries (K x K damped, rhs_K) go to the front that ends the elimination (the root separator, or the
// only leaf): border rows nbd .. nbd + nk
CH_DEV void ch_stage_root_K(const ChainArgs& A, const ChFront& F, int t, int nt)
{
    const int T = 3 * (A.BB - 2 * A.w);
    for (int e = t; e < T * T; e += nt) {
        const int it = e / T, jt = e - T * it;
        const int bi = A.nbd + it, bj = A.nbd + jt;
        if (bi / 3 < bj / 3) continue;
        double v = 0.0;
        if (it < A.nk && jt < A.nk) { v = A.S[(size_t)(A.koff + it) * A.ld + A.koff + jt]; if (it == jt) v = ch_damped(A, v, A.koff + it); }
        else if (it == A.nk && jt < A.nk) v = A.rhs[A.koff + jt];
        else if (jt == A.nk && it < A.nk) v = A.rhs[A.koff + it];
        F.ww[ch_bb_off(F, bi / 3, bj / 3) + 3 * (bi % 3) + bj % 3] = v;
    }
}

// The separators' own entries into their (zeroed) fronts: own x own (damped), K x own, rhs x own (own cameras sit at ring slots
// 0 .. w-1), and for the root K x K / rhs_K.  All separators of a workgroup at once -- `nin` children of tree level lvl0, the
// first one number first_in of its level; fronts level-major at node_base -- as ONE index space over the workgroup's threads:
// ch_stage_load requests (every load in flight at the same time, before the barrier that follows the zero fill), ch_stage_store
// writes.  At most CH_STG entries per thread.
#define CH_STG 4
struct ChStage { double v[CH_STG]; int d[CH_STG]; };
CH_DEV void ch_stage_load(const ChainArgs& A, const unsigned short* pairs, int nin, int lvl0, int first_in, int NF, int t, int nt, ChStage& St, bool pairs_ready)
{
    const int w = A.w, n1 = 9 * ch_tri(2 * w), T = 3 * (A.BB - 2 * w), n2 = T * 6 * w, per = n1 + n2 + T * T;
    CH_UNROLL
    for (int u = 0; u < CH_STG; ++u) {
        St.d[u] = -1; St.v[u] = 0.0;
        const int eg = t + u * nt;
        if (eg >= (nin - 1) * per) continue;
        const int ni = (int)(((float)eg + 0.5f) * (1.0f / (float)per)), e = eg - ni * per;
        int lp = 1, off = 0, cntl = nin >> 1;
        while (ni >= off + cntl) { off += cntl; cntl >>= 1; ++lp; }
        const int g = ni - off, l = lvl0 + lp, kglob = (first_in >> lp) + g, half = 1 << (l - 1), own_lo = ch_sep_lo(A, (2 * kglob + 1) * half - 1);
        const int RB = 4 * w, WW = 9 * ch_tri(RB), BW = 9 * A.BB * RB, base = ni * NF;
        if (e < n1) {
            const int blk = e / 9, k = e - 9 * blk;
            int bi, bj;
            if (pairs_ready) { const unsigned pr = pairs[blk]; bi = (int)(pr & 255u); bj = (int)(pr >> 8); }
            else { bi = (int)((sqrtf(8.0f * (float)blk + 1.0f) - 1.0f) * 0.5f); if (ch_tri(bi) > blk) --bi; if (ch_tri(bi + 1) <= blk) ++bi; bj = blk - ch_tri(bi); }
            const int row = 3 * bi + k / 3, col = 3 * bj + k % 3;
            double v = A.S[(size_t)(6 * own_lo + row) * A.ld + 6 * own_lo + col];
            if (row == col) v = ch_damped(A, v, 6 * own_lo + row);
            St.v[u] = v; St.d[u] = base + 9 * blk + k;
        } else if (e < n1 + n2) {
            const int qq = e - n1, kt = qq / (6 * w), c = qq - 6 * w * kt, br = A.nbd + kt;
            if (kt < A.nk) St.v[u] = A.S[(size_t)(6 * own_lo + c) * A.ld + A.koff + kt];
            else if (kt == A.nk) St.v[u] = A.rhs[6 * own_lo + c];
            St.d[u] = base + WW + 9 * ((br / 3) * RB + c / 3) + 3 * (br % 3) + c % 3;
        } else if (l == A.m) {
            const int qq = e - n1 - n2, it = qq / T, jt = qq - T * it, bi = A.nbd + it, bj = A.nbd + jt;
            if (bi / 3 >= bj / 3) {
                double v = 0.0;
                if (it < A.nk && jt < A.nk) { v = A.S[(size_t)(A.koff + it) * A.ld + A.koff + jt]; if (it == jt) v = ch_damped(A, v, A.koff + it); }
                else if (it == A.nk && jt < A.nk) v = A.rhs[A.koff + jt];
                else if (jt == A.nk && it < A.nk) v = A.rhs[A.koff + it];
                St.v[u] = v; St.d[u] = base + WW + BW + 9 * (ch_tri(bi / 3) + bj / 3) + 3 * (bi % 3) + bj % 3;
            }
        }
    }
}
CH_DEV void ch_stage_store(double* node_base, const ChStage& St)
{
    CH_UNROLL
    for (int u = 0; u < CH_STG; ++u)
        if (St.d[u] >= 0) node_base[St.d[u]] = St.v[u];
}

// ---- hand-over of a finished child front to its parent ------------------------------------------------------------
// A finished child keeps [RA_c (ring, nrr2c blocks) | LA_c | K | rhs].  In the parent's front [own | RA] + [LA | K | rhs]:
//   left child  (LA_c = LA, RA_c = own):   ring -> ring blocks 0 .. 2w-1,  border -> border (same index)
//   right child (LA_c = own, RA_c = RA):   ring -> ring blocks 2w .. 4w-1, border LA_c -> ring blocks 0 .. 2w-1, K / rhs -> border
// Returns the block's offset in the parent's matrix (the same for the LDS front and for the image in HBM); tr: the block is
// stored transposed (the parent orders the two row groups the other way round).
CH_DEV int ch_parent_off(int w, int BB, int side, int nrr2c, int bi, int bj, bool& tr)
{
    int ri, ki, rj, kj;
    if (bi < nrr2c) { ri = 1; ki = side == 0 ? bi : 2 * w + bi; } else { const int b = bi - nrr2c; ri = (side == 1 && b < 2 * w) ? 1 : 0; ki = b; }
    if (bj < nrr2c) { rj = 1; kj = side == 0 ? bj : 2 * w + bj; } else { const int b = bj - nrr2c; rj = (side == 1 && b < 2 * w) ? 1 : 0; kj = b; }
    const int oi = ri ? ki : 1000 + ki, oj = rj ? kj : 1000 + kj;
    tr = oi < oj;
    if (tr) { int t = ri; ri = rj; rj = t; t = ki; ki = kj; kj = t; }
    const int RB = 4 * w, WW = 9 * ch_tri(RB), BW = 9 * BB * RB;
    if (ri) return 9 * (ch_tri(ki) + kj);
    if (rj) return WW + 9 * (ki * RB + kj);
    return WW + BW + 9 * (ch_tri(ki) + kj);
}
// child front (final state h0c, nrr2c) -> dst (parent front matrix in LDS: add; image in HBM: store)
CH_DEV void ch_push(const ChainArgs& A, const ChFront& Fc, int h0c, int nrr2c, int side, double* dst, bool store, const unsigned short* pairs, int nw, int wig, int lane)
{
    const int nrb = nrr2c + A.BB, npairs = ch_tri(nrb);
    for (int q = wig * 64 + lane; q < npairs; q += 64 * nw) {
        const unsigned pr = pairs[q];
        const int bi = (int)(pr & 255u), bj = (int)(pr >> 8);
        const double* src = Fc.ww + ch_rem_off(Fc, h0c, nrr2c, bi, bj);
        bool tr;
        double* d = dst + ch_parent_off(A.w, A.BB, side, nrr2c, bi, bj, tr);
        double v[9];
        CH_UNROLL
        for (int k = 0; k < 9; ++k) v[k] = src[k];
        if (store) {
            CH_UNROLL
            for (int k = 0; k < 9; ++k) d[tr ? (k % 3) * 3 + k / 3 : k] = v[k];
        } else {
            CH_UNROLL
            for (int k = 0; k < 9; ++k) d[tr ? (k % 3) * 3 + k / 3 : k] += v[k];
        }
    }
}
// image in HBM (written by ch_push with store = true, already in the parent's orientation) -> += into the parent front
CH_DEV void ch_pull(const ChainArgs& A, const double* img, int nrr2c, int side, double* dst, const unsigned short* pairs, int nw, int wig, int lane)
{
    const int nrb = nrr2c + A.BB, npairs = ch_tri(nrb);
    for (int q = wig * 64 + lane; q < npairs; q += 64 * nw) {
        const unsigned pr = pairs[q];
        bool tr;
        const int off = ch_parent_off(A.w, A.BB, side, nrr2c, (int)(pr & 255u), (int)(pr >> 8), tr);
        double v[9];
        CH_UNROLL
        for (int k = 0; k < 9; ++k) v[k] = img[off + k];
        CH_UNROLL
        for (int k = 0; k < 9; ++k) dst[off + k] += v[k];
    }
}

// ---- a leaf: G waves eliminate cameras lo .. lo + len - 1 ----------------------------------------------------------------
// `steps` >= len: every leaf of the workgroup runs the same number of synchronisation points.  The front holds the first
// window already.  Wave G - 1 brings the later cameras in: camera lo + s + w + 1 during step s, into the slot the camera of
// step s - 1 left (nobody reads that one any more), requested two steps ahead.
CH_DEV bool ch_leaf_steps(const ChainArgs& A, const ChFront& F, const unsigned short* pairs, const ChLeafGeo& g, int steps, int G, int wig, bool multi, int lane, int& si)
{
    const int w = A.w, RC = w + 2;
    double pv[2][6];
    if (wig == G - 1) { ch_cam_load(A, g, g.lo + w + 1, w, false, lane, pv[0]); ch_cam_load(A, g, g.lo + w + 2, w, false, lane, pv[1]); }
    bool ok = true;
    ChPivot P;
    int hs = 0, ins = w + 1;        // ring slot of the head, and of the camera that enters during this step
    CH_STAMP(A, si);
    for (int s = 0; s < steps; ++s) {
        if (s < g.len) {
            const int left = g.ntot - s, nring = left < w + 1 ? left : w + 1;
            ok = ch_step(A, F, pairs, 2 * hs, nring - 1, g.lo + s, s == 0, s + 1 < g.len, G, wig, s & 1, lane, P, false, si) && ok;
            if (wig == G - 1) {
                ch_cam_store(A, F, g, g.lo + s + w + 1, w, ins, lane, pv[0]);
                CH_UNROLL
                for (int t = 0; t < 6; ++t) pv[0][t] = pv[1][t];
                ch_cam_load(A, g, g.lo + s + w + 3, w, false, lane, pv[1]);
            }
        }
        if (++hs == RC) hs = 0;
        if (++ins == RC) ins = 0;
        if (multi) CH_SYNC_WG(); else CH_SYNC_WAVE();
        CH_STAMP(A, si);
    }
    return ok;
}

// ---- the intrinsics: the last nk columns, from the border of the front that ended the elimination ---------------------------
// returns y_K in yk (every lane the same values); false on a non-positive pivot
CH_DEV bool ch_top_solve(const ChainArgs& A, const ChFront& F, double yk[4])
{
    yk[0] = yk[1] = yk[2] = yk[3] = 0.0;
    if (A.nk != 4) return true;
    double a[10], z[4];
    CH_UNROLL
    for (int i = 0; i < 4; ++i) {
        const int bi = A.nbd + i;
        CH_UNROLL
        for (int jx = 0; jx <= i; ++jx) { const int bj = A.nbd + jx; a[i * (i + 1) / 2 + jx] = F.ww[ch_bb_off(F, bi / 3, bj / 3) + 3 * (bi % 3) + bj % 3]; }
        const int br = A.nbd + 4;
        z[i] = F.ww[ch_bb_off(F, br / 3, bi / 3) + 3 * (br % 3) + bi % 3];
    }
    bool ok = true;
    double L[6], yv[4];
    CH_UNROLL
    for (int jx = 0; jx < 4; ++jx) {
        const double d = a[jx * (jx + 1) / 2 + jx];
        ok = ok && (d > 0.0) && (d < 1e300);
        const double y = ch_rsqrt(d);
        yv[jx] = y;
        CH_UNROLL
        for (int i = jx + 1; i < 4; ++i) L[i * (i - 1) / 2 + jx] = a[i * (i + 1) / 2 + jx] * y;
        CH_UNROLL
        for (int i = jx + 1; i < 4; ++i)
            CH_UNROLL
            for (int k = jx + 1; k <= i; ++k) a[i * (i + 1) / 2 + k] = ch_fma(-L[i * (i - 1) / 2 + jx], L[k * (k - 1) / 2 + jx], a[i * (i + 1) / 2 + k]);
    }
    // L z' = z, then L' y = z'
    CH_UNROLL
    for (int c = 0; c < 4; ++c) {
        double t = z[c];
        CH_UNROLL
        for (int k = 0; k < c; ++k) t = ch_fma(-z[k], L[c * (c - 1) / 2 + k], t);
        z[c] = t * yv[c];
    }
    CH_UNROLL
    for (int c = 3; c >= 0; --c) {
        double t = z[c];
        CH_UNROLL
        for (int k = c + 1; k < 4; ++k) t = ch_fma(-yk[k], L[k * (k - 1) / 2 + c], t);
        yk[c] = t * yv[c];
    }
    return ok;
}

// ---- back-substitution of one leaf or separator (one wave) ---------------------------------------------------------------
// Cameras c0 .. c0 + cnt - 1, eliminated in that order; camera idx kept nrr = min(cap, tot - idx) - 1 ring cameras behind it:
// the later ones of this segment, then RA (unknowns at ra_pos ...); then the border LA (la_pos, -1: none) and K.
//     y_c = zt_c - sum_r W_c'[r] y[pos(r)]
// From the last camera to the first.  Lane r holds row r of W_c' (at most 58 rows: CH_WMAX), requested two cameras
// ahead; its six products go to LDS transposed, lane (k, part) of an eight-lane group adds eight of them, the group the rest.
struct ChBackRow { double v[6]; double zt; };
CH_DEV void ch_back_load(const ChainArgs& A, int c, int nx, int lane, ChBackRow& R)
{
    const double* rc = A.rec + (size_t)c * A.rec_stride;
    CH_UNROLL
    for (int k = 0; k < 6; ++k) R.v[k] = lane < nx ? rc[CH_REC_HEAD + 6 * lane + k] : 0.0;
    R.zt = rc[(lane >> 3) < 6 ? (lane >> 3) : 0];
}
CH_DEV void ch_backward_seg(const ChainArgs& A, double* ysh, double* scr, int c0, int cnt, int cap, int tot, int ra_pos, int la_pos, int lane)
{
    auto nrr_of = [&](int idx) { const int left = tot - idx; return (left < cap ? left : cap) - 1; };
    ChBackRow R0, R1;
    { const int i0 = cnt - 1; ch_back_load(A, c0 + i0, 6 * nrr_of(i0) + A.nbd + A.nk, lane, R0); }
    if (cnt > 1) { const int i1 = cnt - 2; ch_back_load(A, c0 + i1, 6 * nrr_of(i1) + A.nbd + A.nk, lane, R1); } else R1 = R0;
    const int kk = lane >> 3, part = lane & 7;
    for (int idx = cnt - 1; idx >= 0; --idx) {
        const int c = c0 + idx, nrr = nrr_of(idx), later = cnt - 1 - idx;
        const int seg = 6 * (nrr < later ? nrr : later), nx = 6 * nrr + A.nbd + A.nk;
        double p[6] = { 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 };
        {
            const int r = lane;
            if (r < nx) {
                int pos;
                if (r < seg) pos = 6 * (c + 1) + r;
                else if (r < 6 * nrr) pos = ra_pos + (r - seg);
                else { const int br = r - 6 * nrr; pos = br < A.nbd ? (la_pos >= 0 ? la_pos + br : -1) : A.koff + (br - A.nbd); }
                const double yv = pos >= 0 ? ysh[pos] : 0.0;
                CH_UNROLL
                for (int k = 0; k < 6; ++k) p[k] = R0.v[k] * yv;
            }
        }
        CH_UNROLL
        for (int k = 0; k < 6; ++k) scr[64 * k + lane] = p[k];
        const double zt = R0.zt;
        R0 = R1;
        if (idx >= 2) ch_back_load(A, c0 + idx - 2, 6 * nrr_of(idx - 2) + A.nbd + A.nk, lane, R1);
        CH_SYNC_WAVE();
        double t = 0.0;
        if (kk < 6) {
            CH_UNROLL
            for (int jx = 0; jx < 8; ++jx) t += scr[64 * kk + part + 8 * jx];
        }
        const double tot8 = ch_sum8(t);
        if (kk < 6 && part == 0) ysh[6 * c + kk] = zt - tot8;
        CH_SYNC_WAVE();
    }
}

// load every 128-byte line of [p, p + n) once (into the L2 this workgroup's XCD uses) without keeping anything
CH_DEV void ch_warm(const double* p, size_t n, int t, int nt)
{
    double acc = 0.0;
    for (size_t i0 = (size_t)t * 16; i0 < n; i0 += (size_t)nt * 16 * 8) {
        double v[8];
        CH_UNROLL
        for (int u = 0; u < 8; ++u) { const size_t i = i0 + (size_t)u * nt * 16; v[u] = i < n ? p[i] : 0.0; }
        CH_UNROLL
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    ch_keep(acc);
}

// top of the tree is done (front Froot): intrinsics, then the whole tree and every leaf backwards, nwb waves
CH_DEV void ch_finish(const ChainArgs& A, const ChFront& Froot, double* ysh, int nwb, int tid, int& si)
{
    const int lane = tid & 63, wave = tid >> 6, w = A.w;
    double* scr = ysh + ch_ysh_doubles(A.n) + (size_t)wave * ch_back_scratch(w);
    double yk[4];
    CH_STAMP(A, si);
    const bool ok = ch_top_solve(A, Froot, yk);
    if (!ok && tid == 0) *A.err = 2;
    CH_SYNC_WG();               // ysh takes the fronts' place: everyone has read the border first
    if (tid < A.nk) ysh[A.koff + tid] = tid == 0 ? yk[0] : tid == 1 ? yk[1] : tid == 2 ? yk[2] : yk[3];
    CH_SYNC_WG();
    CH_STAMP(A, si);
    for (int l = A.m; l >= 1; --l) {
        const int nodes = A.P >> l, half = 1 << (l - 1);
        for (int k = wave; k < nodes; k += nwb) {
            const int i = (2 * k + 1) * half - 1;
            const bool has_la = i - half >= 0, has_ra = i + half <= A.P - 2;
            ch_backward_seg(A, ysh, scr, ch_sep_lo(A, i), w, 1 << 20, has_ra ? 2 * w : w, has_ra ? 6 * ch_sep_lo(A, i + half) : 0, has_la ? 6 * ch_sep_lo(A, i - half) : -1, lane);
        }
        CH_SYNC_WG();
        CH_STAMP(A, si);
    }
    for (int j = wave; j < A.P; j += nwb) {
        const ChLeafGeo g = ch_leaf_geo(A, j);
        ch_backward_seg(A, ysh, scr, g.lo, g.len, w + 1, g.ntot, 6 * g.hi, g.has_la ? 6 * g.la_lo : -1, lane);
    }
    CH_SYNC_WG();
    CH_STAMP(A, si);
    for (int i = tid; i < A.npad; i += 64 * nwb) A.y[i] = i < A.n ? ysh[i] : 0.0;
    CH_STAMP(A, si);
}

CH_DEV void ch_build_pairs(unsigned short* pairs, int w, int tid, int nt)
{
    const int np = ch_tri(6 * w);
    for (int q = tid; q < np; q += nt) {
        int bi = (int)((sqrtf(8.0f * (float)q + 1.0f) - 1.0f) * 0.5f);
        while (ch_tri(bi) > q) --bi;
        while (ch_tri(bi + 1) <= q) ++bi;
        pairs[q] = (unsigned short)(bi | ((q - ch_tri(bi)) << 8));
    }
}

// what a finished front keeps: the ring head of its last step and the ring blocks behind it
struct ChChildState { int h0, nrr2; };
CH_DEV ChChildState ch_leaf_final(const ChainArgs& A, int j)
{
    ChChildState s;
    s.h0 = 2 * ch_mod(ch_leaf_len(A, j) - 1, A.w + 2); s.nrr2 = j < A.P - 1 ? 2 * A.w : 0;
    return s;
}
CH_DEV ChChildState ch_node_final(const ChainArgs& A, int level, int k)
{
    ChChildState s; const int half = 1 << (level - 1), i = (2 * k + 1) * half - 1;
    s.h0 = 2 * (A.w - 1); s.nrr2 = i + half <= A.P - 2 ? 2 * A.w : 0;
    return s;
}

// ---- chain_sub_kernel: 2^a leaves (G waves each) and a levels of the tree per workgroup ------------------------------------
CH_DEV void chain_sub_body(const ChainArgs& A, double* smem)
{
    const int tid = CH_TID, lane = tid & 63, wave = tid >> 6;
    const int G = A.G, NL = 1 << A.a, NW = NL * G, nt = 64 * NW, w = A.w, BB = A.BB;
    const bool multi = NW > 1;
    // wave wv of leaf lw; the roles (wig: 0 = pivot wave) are rotated from leaf to leaf so that the pivot waves, the ones every
    // camera waits for, sit on different SIMDs (waves go to the SIMDs round-robin)
    const int lw = wave / G, wv = wave - lw * G;
    const int rot = G == 4 ? lw : G == 2 ? (((lw + 1) >> 1) & 1) : 0;
    const int wig = (wv + rot) % G;
    unsigned short* pairs = (unsigned short*)smem;
    double* base = smem + ch_pair_doubles(w);
    const int LF = ch_leaf_front(w, BB), NF = ch_node_front(w, BB);
    double* node_base = base + (size_t)NL * LF;
    int si = 0;
    CH_STAMP(A, si);
    const int first_leaf = CH_BID * NL;
    const ChLeafGeo geo = ch_leaf_geo(A, first_leaf + lw);
    const ChFront Fl = ch_front_at(base + (size_t)lw * LF, ch_leaf_rb(w), BB, 0);
    // the first window and the separators' own entries: every load goes out before anything else
    const int ninit = geo.ntot < w + 1 ? geo.ntot : w + 1;
    double iv[CH_WMAX + 1][9];
    if (wig == 0) {
        CH_UNROLL
        for (int i = 0; i < CH_WMAX + 1; ++i)
            if (i < ninit) ch_cam_load(A, geo, geo.lo + i, i, geo.has_la, lane, iv[i]);
    }
    ChStage St;
    ch_stage_load(A, pairs, NL, 0, first_leaf, NF, tid, nt, St, false);
    ch_build_pairs(pairs, w, tid, nt);
    for (int i = tid; i < (NL - 1) * NF; i += nt) node_base[i] = 0.0;
    { const int mat = ch_leaf_mat(w, BB); for (int i = wv * 64 + lane; i < mat; i += 64 * G) Fl.ww[i] = 0.0; }
    if (multi) CH_SYNC_WG(); else CH_SYNC_WAVE();
    ch_stage_store(node_base, St);
    if (wig == 0) {
        if (A.P == 1) ch_stage_root_K(A, Fl, lane, 64);
        CH_UNROLL
        for (int i = 0; i < CH_WMAX + 1; ++i)
            if (i < ninit) ch_cam_store(A, Fl, geo, geo.lo + i, i, i, lane, iv[i]);
    }
    if (multi) CH_SYNC_WG(); else CH_SYNC_WAVE();
    CH_STAMP(A, si);
    const int steps = A.q + (first_leaf < A.r ? 1 : 0);        // the longest leaf of this workgroup (longer leaves come first)
    bool ok = ch_leaf_steps(A, Fl, pairs, geo, steps, G, wig, multi, lane, si);
    for (int l = 1; l <= A.a; ++l) {
        const int nw = G << l, g = wave / nw, wn = (wave - g * nw + g) % nw, hw = nw >> 1;
        const int kglob = (first_leaf >> l) + g, half = 1 << (l - 1), i = (2 * kglob + 1) * half - 1;
        const ChFront Fp = ch_front_at(node_base + (size_t)(NL - (NL >> (l - 1)) + g) * NF, 4 * w, BB, 1);
        const int side = wn >= hw ? 1 : 0, kc = 2 * g + side;
        ChFront Fc; ChChildState cs;
        if (l == 1) { Fc = ch_front_at(base + (size_t)kc * LF, ch_leaf_rb(w), BB, 0); cs = ch_leaf_final(A, first_leaf + kc); }
        else { Fc = ch_front_at(node_base + (size_t)(NL - (NL >> (l - 2)) + kc) * NF, 4 * w, BB, 1); cs = ch_node_final(A, l - 1, (first_leaf >> (l - 1)) + kc); }
        if (side == 0) ch_push(A, Fc, cs.h0, cs.nrr2, 0, Fp.ww, false, pairs, hw, wn, lane);
        CH_SYNC_WG();
        if (side == 1) ch_push(A, Fc, cs.h0, cs.nrr2, 1, Fp.ww, false, pairs, hw, wn - hw, lane);
        CH_SYNC_WG();
        CH_STAMP(A, si);
        const bool has_ra = i + half <= A.P - 2;
        const int own_lo = ch_sep_lo(A, i);
        ChPivot Pn;
        for (int t = 0; t < w; ++t) {
            ok = ch_step(A, Fp, pairs, 2 * t, (has_ra ? 2 * w : w) - t - 1, own_lo + t, t == 0, t + 1 < w, nw, wn, t & 1, lane, Pn, false, si) && ok;
            CH_SYNC_WG();
            CH_STAMP(A, si);
        }
    }
    if (!ok && lane == 0) *A.err = 2;
    // the workgroup's root
    ChFront Fr; ChChildState rs;
    if (A.a == 0) { Fr = Fl; rs = ch_leaf_final(A, first_leaf); }
    else { Fr = ch_front_at(node_base + (size_t)(NL - 2) * NF, 4 * w, BB, 1); rs = ch_node_final(A, A.a, CH_BID); }
    if (A.a == A.m) { if (multi) CH_SYNC_WG(); else CH_SYNC_WAVE(); ch_finish(A, Fr, base, NW, tid, si); return; }
    ch_push(A, Fr, rs.h0, rs.nrr2, CH_BID & 1, A.img + (size_t)CH_BID * A.img_doubles, true, pairs, NW, wave, lane);
    CH_STAMP(A, si);
}

// ---- chain_top_kernel: the levels above, the intrinsics, the back-substitution -----------------------------------------
CH_DEV void chain_top_body(const ChainArgs& A, double* smem)
{
    const int tid = CH_TID, lane = tid & 63, wave = tid >> 6;
    const int NWT = A.nw_top, nt = 64 * NWT, w = A.w, BB = A.BB;
    const int levels = A.m - A.a, NC = 1 << levels;
    unsigned short* pairs = (unsigned short*)smem;
    double* base = smem + ch_pair_doubles(w);
    const int NF = ch_node_front(w, BB);
    int si = 128;
    CH_STAMP(A, si);
    ChStage St;
    ch_stage_load(A, pairs, NC, A.a, 0, NF, tid, nt, St, false);
    ch_build_pairs(pairs, w, tid, nt);
    for (int i = tid; i < (NC - 1) * NF; i += nt) base[i] = 0.0;
    CH_SYNC_WG();
    ch_stage_store(base, St);
    ch_warm(A.rec, (size_t)A.ncf * A.rec_stride, tid, nt);     // the back-substitution reads every record: start them on their way now
    CH_SYNC_WG();
    CH_STAMP(A, si);
    bool ok = true;
    for (int lp = 1; lp <= levels; ++lp) {
        const int l = A.a + lp, nodes = NC >> lp, nw = NWT / nodes;        // nodes <= NWT / 2 (chain_plan): at least two waves per node
        const int g = wave / nw, wig = (wave - g * nw + g) % nw, hw = nw >> 1;
        const int half = 1 << (l - 1), i = (2 * g + 1) * half - 1;
        const ChFront Fp = ch_front_at(base + (size_t)(NC - (NC >> (lp - 1)) + g) * NF, 4 * w, BB, 1);
        for (int side = 0; side < 2; ++side) {
            const int kc = 2 * g + side;
            if (lp == 1) {
                // children = images in HBM, one per workgroup of chain_sub_kernel: child c covers leaves [c 2^a, (c + 1) 2^a)
                ch_pull(A, A.img + (size_t)kc * A.img_doubles, kc + 1 < NC ? 2 * w : 0, side, Fp.ww, pairs, nw, wig, lane);
            } else {
                // children = fronts of the level below, in LDS: the half of the waves that eliminated a child hands it over
                const ChFront Fc = ch_front_at(base + (size_t)(NC - (NC >> (lp - 2)) + kc) * NF, 4 * w, BB, 1);
                const ChChildState cs = ch_node_final(A, l - 1, kc);
                if ((wig >= hw ? 1 : 0) == side) ch_push(A, Fc, cs.h0, cs.nrr2, side, Fp.ww, false, pairs, hw, wig - side * hw, lane);
            }
            CH_SYNC_WG();
            CH_STAMP(A, si);
        }
        const bool has_ra = i + half <= A.P - 2;
        const int own_lo = ch_sep_lo(A, i);
        for (int t = 0; t < w; ++t) {
            ChPivot Pn;         // not kept from step to step here: with sixteen waves (128 registers) the factor is re-read from lb
            ok = ch_step(A, Fp, pairs, 2 * t, (has_ra ? 2 * w : w) - t - 1, own_lo + t, t == 0, t + 1 < w, nw, wig, t & 1, lane, Pn, true, si) && ok;
            CH_SYNC_WG();
            CH_STAMP(A, si);
        }
    }
    if (!ok && lane == 0) *A.err = 2;
    const ChFront Fr = ch_front_at(base + (size_t)(NC - 2) * NF, 4 * w, BB, 1);
    ch_finish(A, Fr, base, NWT, tid, si);
}

#ifndef CHAIN_HOST_EMU
extern __shared__ double ch_smem[];
__global__ __launch_bounds__(64 * CH_NW) void chain_sub_kernel(ChainArgs A) { chain_sub_body(A, ch_smem); }
__global__ __launch_bounds__(64 * CH_NW_TOP) void chain_top_kernel(ChainArgs A) { chain_top_body(A, ch_smem); }
#endif

// ---- host side: plan --------------------------------------------------------------------------------------------
// Fills the geometry for a chain of ncf free cameras with band half width w.  Returns false when the chain solver does not
// apply (band too wide for the register / LDS budget); P, a and G are chosen here (force_*: measurement overrides, <= 0 / < 0: automatic).
static inline bool chain_plan(ChainArgs& A, int ncf, int w, int nk, int ld, int npad, int force_P, int force_a, int force_G, size_t lds_limit_bytes)
{
    if (w < 1) w = 1;
    if (w > CH_WMAX || ncf < 1 || (nk != 0 && nk != 4)) return false;
    A.ncf = ncf; A.w = w; A.nk = nk; A.koff = 6 * ncf; A.ld = ld; A.n = 6 * ncf + nk; A.npad = npad;
    A.nbd = 6 * w; A.BB = 2 * w + (nk + 1 + 2) / 3;
    const int nxr = 6 * (3 * w - 1) + nk;           // record rows of a separator's first camera: [own rest | RA | LA | K]
    A.rec_stride = (CH_REC_HEAD + 6 * nxr + 3) & ~3;
    A.img_doubles = ch_node_mat(w, A.BB);
    A.nw_top = CH_NW_TOP;
    // leaves of at least six cameras (a tree level costs w camera steps, like w cameras of a leaf), at most 32 of them
    int P = 1, m = 0;
    const int min_leaf = w > 6 ? w : 6;
    while (P < 32 && (ncf - (2 * P - 1) * w) / (2 * P) >= min_leaf) { P *= 2; ++m; }
    if (force_P > 0) { P = 1; m = 0; while (P * 2 <= force_P && (ncf - (2 * P - 1) * w) / (2 * P) >= (w > 1 ? w : 1)) { P *= 2; ++m; } }
    for (;; P /= 2, --m) {
        A.P = P; A.m = m;
        const int interior = ncf - (P - 1) * w;
        A.q = interior / P; A.r = interior % P;
        // one launch while the whole tree fits a workgroup's eight waves with at least two waves per leaf; otherwise the top kernel
        // takes at most eight children (three levels) and the workgroups of the first kernel the levels below
        int a = m <= 2 ? m : (m - 3 > 1 ? m - 3 : 1);
        if (force_a >= 0 && force_a <= m) a = force_a;
        int G = CH_NW >> a; if (G > 4) G = 4; if (G < 1) G = 1;
        if (force_G > 0 && (force_G << a) <= CH_NW) G = force_G;
        A.a = a; A.G = G;
        const int levels = m - a;
        const bool shape_ok = (1 << a) * G <= CH_NW && levels <= 3 && (P == 1 || A.q >= w);
        const size_t need1 = 8 * ch_sub_lds(w, A.BB, a, G, A.n, a == m), need2 = levels ? 8 * ch_top_lds(w, A.BB, levels, A.nw_top, A.n) : 0;
        if (shape_ok && need1 <= lds_limit_bytes && need2 <= lds_limit_bytes) return true;
        if (P == 1) return false;
        if (force_a >= 0) force_a = -1;         // a forced split that does not fit: fall back to the automatic one before giving up leaves
    }
}
