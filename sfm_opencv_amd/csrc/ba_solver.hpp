// ba_solver.hpp -- reduced camera system solvers (included by ba.hip only).
//
// S (order n = 6 cams + 4 intrinsics, padded to a multiple of 32) is SPD after damping.  Two paths:
//
//  * chol_sparse_kernel: ONE workgroup factors S = L L' panel by panel (32 columns) and solves, driven by the
//    block fill pattern computed on the host (symbolic factorisation over 32x32 blocks).  In the reference's
//    pipeline tracks only chain through consecutive frames (NViewReconstuct.cpp:1289-1299), so S is block-banded
//    plus the dense intrinsic rows: each panel touches a handful of blocks and the whole solve is a chain of
//    ~n dependent column steps -- latency-bound, so it runs inside one CU with no launches and no inter-workgroup
//    traffic.  The right-hand side rides along as an extra row (forward substitution for free), the backward
//    substitution follows in the same launch.  Used when every panel has <= SRMAX sub-diagonal blocks.
//  * chol_diag/trsm/syrk/solve kernels (ba.hip): dense right-looking blocked Cholesky over many workgroups, the
//    general fallback for wide / unstructured S.
//
// The 32x32 diagonal block is factored by one wave: lane = row, the row in registers; each scaled pivot column is
// published through LDS and read back as wave-uniform wide loads.  Cross-lane visibility inside the wave uses
// wavefront-scope fences + wave_barrier (no instructions, only compiler ordering: DS ops of a wave run in order).
// NOTE: never route these LDS accesses through `volatile` generic pointers -- hipcc turns them into
// flat_load ... sc0 sc1 with a vmcnt(0) wait after every access.
#pragma once
#include <hip/hip_runtime.h>

#define SNB 32
#define SLD 33          // LDS row stride in doubles: conflict-free row and column access
#define SRMAX 8         // max sub-diagonal blocks per panel for the single-workgroup path
#define SAMAX 8         // max blocks outside its own panel range that one elimination-tree node may reach
#define STHREADS 512    // 8 waves, two per SIMD: wave 0 runs the pivot-block chain, the other seven share the trailing update
#define SWAVES (STHREADS / 64)
#define SSTAGE (4096 / STHREADS)   // staging loads in flight per thread: four 32x32 row blocks per round trip
#define SROWS (STHREADS / 32)
#ifndef SPW
#define SPW 16          // panel width of the 32x32 pivot-block factorisation (columns a lane keeps in registers)
#endif

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void wave_sync_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding global store
// (vmcnt(0)), a full L2 round trip, which the panel loop can only afford once per panel
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

typedef double v4d __attribute__((ext_vector_type(4)));

struct DiagLds {
    double D[SNB][SLD];                 // diagonal block / its factor (lower)
    // working copy for the factorisation, 16-byte aligned rows (even stride).  Rows 0..31: L.  Rows 32..63: the inverse,
    // one COLUMN per row: W[32 + c][m] = (L^-1)[m][c] -- i.e. read as a 32x32 array it is (L^-1)', the operand X = B (L^-1)' needs.
    double W[2 * SNB][SNB + 2];
    double G[2 * SNB][SPW + 2];         // one panel of the cross-panel update, handed from the MFMA result layout to lane = row
    double Lc[SPW][2 * SNB];            // the panel's finished columns, one per row: broadcast source of the deferred rank-1 updates
};
// ---- multi-level nested dissection of the camera chain ---------------------------------------------------------------
// The camera chain is cut into P = 2^m leaf segments by P - 1 separators of `w` cameras (w = band width of the camera
// graph).  Elimination order: leaves | separators that split sibling leaves | separators one level up | ... | the last
// separators + the intrinsics ("top").  Nodes of one level are mutually independent: one workgroup each, one launch per
// level; the top (a handful of panels) is factored by a single workgroup, which also starts the back-substitution.
// A node's panels reach a few blocks outside its own range (the separators that bound it and the intrinsics block):
// its updates to pairs of such blocks go to a PRIVATE update buffer U (na*32 square + na*32 right-hand side, zero on
// entry), never to S itself, and the node that owns the lower-numbered block of a pair folds every descendant's
// contribution into S before it factors its own panels -- in a fixed order, so the result is run-to-run identical.
struct NodeDesc {
    int k0, k1;             // own panel (32-block) range [k0, k1)
    int na;                 // blocks outside [k0, k1) that its panels reach (after fill)
    int anc[SAMAX];         // ... their indices, ascending
    int e0, e1, e2;         // fold entries: [e0, e1) 32x32 blocks, [e1, e2) 32-entry pieces of the right-hand side
    long long u_off;        // offset (doubles) of its update buffer in ubuf: [U (na*32)^2 | rhs na*32]
};
// One destination of a node's assembly step (extend-add of the multifrontal method): dst += the sources, in list order.
// Sources are its CHILDREN's update-buffer blocks (child = node whose lowest outside block this node owns); destinations are
// blocks of its own columns in S, or -- for pairs of blocks that lie outside this node too -- blocks of its own update
// buffer (handed on to its parent in turn).  Every diagonal block of its own columns has an entry (possibly without
// sources): the LM damping is applied to it in the same pass.
#define SFOLD_SRC 8
struct FoldEnt {
    double* dst; const double* src[SFOLD_SRC];
    int dst_ld, src_ld[SFOLD_SRC];
    int nsrc, diag0;        // diag0 >= 0: dst is a diagonal block of S whose first row has this position (damping applies);
                            // diag0 == -2: dst lies in this node's own update buffer, which is still all zero (not read)
};

struct SolverLds : DiagLds {
    double B[SRMAX * SNB + 1][SLD];     // stacked row blocks of the panel + the rhs row; the y vector in the backward phase
    double Red[SNB][SLD];
    int Rows3[3][SRMAX];                // block rows of panels k, k + 1, k + 2 (slot = panel % 3): the idle wave fetches two panels
                                        // ahead, so no panel waits for its row list
    // trailing-update block pairs of the current panel: { first row of the A operand in B, first row of the B operand,
    // address of the 32x32 target block (lo, hi) } and the target's leading dimension; double-buffered by panel parity
    // (the idle wave fills the next panel's table while the others walk the current one)
    int4 Pair[2][SRMAX * (SRMAX + 1) / 2];
    int PairLd[2][SRMAX * (SRMAX + 1) / 2];
    int Anc[SAMAX];                     // the node's outside blocks (slot = position in this list)
};

// One wave: in-place Cholesky of the 32x32 block in s.D (lower) AND the inverse of the factor, in the same 32 pivot steps:
//     L[i][j] = (A[i][j] - sum_{m<j} L[i][m] L[j][m]) / L[j][j]                       lanes i = 0..31
//     Z[j][c] = (I[j][c] - sum_{m<j} L[j][m] Z[m][c]) / L[j][j],  Z = L^-1            lanes 32 + c
// Both are "own row . row j of L": one instruction stream, the upper half-wave starts from the identity instead of A
// and its row 32 + c ends up holding column c of L^-1.
// Two panels of SPW = 16 columns.  Inside a panel the lane keeps its 16 entries in registers and the pivot steps are
// right-looking with v_readlane broadcasts: no LDS on the dependent chain (readlane -> rsqrt + Newton -> scale ->
// readlane -> fma: ~20 dependent fp64 ops at ~16 cycles each; the other columns' updates fill its issue gaps).
// Across panels the sum over the finished columns m < p is one small product W[:, :p] W[p:p+16, :p]' on the fp64
// MFMA (operands requested up front), handed back to lane = row through s.G.  (8-column panels: three MFMA phases
// cost 8.7k of 20k cycles; one 32-column panel: the early steps become issue-bound.)
// History (cycles per 32x32 block at C4, in-kernel): sqrt + divide, rows in registers with scratch spills 45k;
// left-looking with LDS rows 23k (+19.5k for the per-row substitutions the inverse now replaces); this form: see
// profiles/README.md.
// 1/sqrt(d) by v_rsq_f64 + two Newton steps (10 dependent ops; sqrt + divide expand to ~32 dependent fp64 ops).
// The panel solve is then X = B (L^-1)' on the fp64 MFMA instead of a 32-step substitution per row.
__device__ __forceinline__ double rsqrt_refined(double d)
{
    double y = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
    double e = fma(-h * y, y, 0.5); y = fma(y, e, y);
    e = fma(-h * y, y, 0.5); y = fma(y, e, y);
    return y;
}

typedef double v2d __attribute__((ext_vector_type(2)));

// fire-and-forget fp64 add into global memory.  The address-space cast matters: a pointer rebuilt from a table is
// generic, the compiler would emit flat_atomic_add_f64, and flat operations also count in lgkmcnt -- every LDS wait
// after them would wait for the atomics' round trip to L2 as well.
typedef __attribute__((address_space(1))) double global_f64;
__device__ __forceinline__ void global_add_f64(double* p, double v)
{
    (void)__builtin_amdgcn_global_atomic_fadd_f64((global_f64*)p, v);
}

#ifdef SFM_CHOL_STAMPS
__device__ long long g_chol_stamps[16];
#define CSTAMP(i) do { if (lane == 0) g_chol_stamps[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define CSTAMP(i) do { } while (0)
#endif
__device__ __forceinline__ bool wave_chol32(DiagLds& s, int lane)
{
    const bool lower = lane < SNB;
    const int ident = lane - SNB;                           // upper half-wave: column index of L^-1
    const int li = lane & 15, lk = lane >> 4;
    const double* dr = &s.D[lane & (SNB - 1)][0];
    bool ok = true;
#pragma unroll
    for (int b = 0; b < SNB / SPW; ++b) {
        const int p = SPW * b;
        double x[SPW];
        CSTAMP(4 * b + 0);
#pragma unroll
        for (int c = 0; c < SPW; ++c) x[c] = lower ? dr[p + c] : (ident == p + c ? 1.0 : 0.0);
        if (b > 0) {
            // G = W[:, :p] W[p:p+SPW, :p]'  (64 x SPW; SPW < 16: the MFMA's surplus columns repeat and are dropped)
            double bop[SNB / 4], aop[4][SNB / 4];
#pragma unroll
            for (int kk = 0; kk < p / 4; ++kk) {
                bop[kk] = s.W[p + (li & (SPW - 1))][4 * kk + lk];
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) aop[rt][kk] = s.W[16 * rt + li][4 * kk + lk];
            }
            v4d acc[4];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) acc[rt] = v4d{ 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
            for (int kk = 0; kk < p / 4; ++kk)
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[rt][kk], bop[kk], acc[rt], 0, 0, 0);
            if (li < SPW) {
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) s.G[16 * rt + lk + 4 * g][li] = acc[rt][g];
            }
            wave_sync_lds();
#pragma unroll
            for (int c = 0; c < SPW; c += 2) { const v2d t = *(const v2d*)&s.G[lane][c]; x[c] -= t.x; x[c + 1] -= t.y; }
        }
        CSTAMP(4 * b + 1);
        // Pivot steps.  Only the NEXT column takes its rank-1 update through v_readlane (it is on the dependent chain); the
        // columns beyond get column j's update one step late, from an LDS broadcast of the finished column (one uniform-
        // address ds_read per two values instead of four v_readlane_b32: 610 of the routine's 1,708 instructions were
        // v_readlane), requested at the end of step j and consumed behind step j + 1's chain.
        // Measured (experiments/chol_bench.hip, cycles per 32x32 block): 10.6k with per-column SPD tests and all updates
        // through v_readlane; 9.6k with the single final test; 9.4k with the deferred updates (1,365 instructions).  Per
        // 16-column panel the pivot loop is 2.7k cycles (166 per column), the cross-panel MFMA product + hand-over 1.6k, loads /
        // stores / the final copy 2.7k.  No gain from: 2x2 block pivots (both reciprocal roots from one Newton latency:
        // 9.7k), dealing the deferred updates out between the chain's operations with pinned order (10.5k), one 32-column
        // panel (10.4k: the early steps become issue-bound).
        double bv[SPW];                 // broadcast values of the column finished one step ago (requested at the end of that step)
#pragma unroll
        for (int jj = 0; jj < SPW; ++jj) {
            const int j = p + jj;
            const double d = readlane_f64(x[jj], j);
            const double y = rsqrt_refined(d);              // d <= 0, NaN or inf: l_jj and everything after it turns NaN, caught by the one test at the end
            const double l = x[jj] * y;                     // lane j: d / sqrt(d)
            x[jj] = l;
            if (jj + 1 < SPW) x[jj + 1] = fma(-l, readlane_f64(l, j + 1), x[jj + 1]);
            if (jj + 2 < SPW) s.Lc[jj][lane] = l;
            __builtin_amdgcn_sched_barrier(0);              // the deferred updates stay BEHIND the chain: their LDS reads have had a whole step to land
            if (jj >= 1) {
#pragma unroll
                for (int c = jj + 1; c < SPW; ++c) x[c] = fma(-x[jj - 1], bv[c], x[c]);
            }
            wave_sync_lds();
            if (jj + 2 < SPW) {
#pragma unroll
                for (int c = jj + 2; c < SPW; ++c) bv[c] = s.Lc[jj][p + c];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        CSTAMP(4 * b + 2);
        // above the diagonal of L (upper half-wave: lane >= 32 > j): zero, off the pivot chain
#pragma unroll
        for (int c = 0; c < SPW; ++c) x[c] = (lane < p + c) ? 0.0 : x[c];
#pragma unroll
        for (int c = 0; c < SPW; c += 2) *(v2d*)&s.W[lane][p + c] = v2d{ x[c], x[c + 1] };
        if (lower) {                    // the factor itself goes straight to s.D (odd row stride: 8-byte stores); a copy W -> D at the end cost 0.7k cycles
#pragma unroll
            for (int c = 0; c < SPW; ++c) s.D[lane][p + c] = x[c];
        }
        wave_sync_lds();
    }
    CSTAMP(8);
    // positive-definiteness: one test of the finished diagonal instead of two compares per pivot step on the issue-bound
    // chain (a failed pivot poisons its own diagonal entry and every later column with NaN)
    {
        const double dg = s.W[lane & (SNB - 1)][lane & (SNB - 1)];
        ok = __builtin_amdgcn_ballot_w64(lower && !((dg > 0.0) && (dg < 1e150))) == 0ull;
    }
    CSTAMP(9);
    return ok;
}

// ------------------------------------------------------------------------------------------------
// Panel sweeps shared by the kernels below.  prow_start[k] .. prow_start[k+1]: ascending block rows i > k with
// L_ik != 0 (after fill).  A node sweeps its own panels [k0, k1); with HAS_EXT, updates whose target lies entirely
// outside that range (both blocks >= k1: they belong to later nodes) are accumulated in the node's PRIVATE update buffer
// (extA: na*32 square with row stride ext_ld, extrhs: na*32; slot of a block = its position in s.Anc) and folded into S
// by their owner later, in a fixed order (deterministic).
// ------------------------------------------------------------------------------------------------
struct SolverPlan {
    const int* prow_start; const int* prow;
    int nb, top_blk;             // blocks; top_blk: first block of the serially factored top (nb: none)
    int dbg;                     // timing experiments only (SFMHIP_EXP_SOLVER): skip phases, results are garbage
    long long* stamps;           // diagnostic (SFMHIP_SOLVER_STAMPS): s_memtime at the phase boundaries of each panel, 8 per panel
    double* linv;                // (L_kk^-1)' of every pivot block, 32x32 each, written by the forward sweep for the backward one
    // LM damping applied by the solver kernels themselves (one launch less per iteration): S_ii += clamp(diagU_i) / radius on
    // real parameters, unit diagonal on padding slots AND on parameters no residual touches (diagU == 0: their row and column of S
    // and their gradient are zero; Ceres never sees such blocks -- only blocks of added residuals enter the problem, NView:1187-1197 --
    // so they stay where they are instead of making S + D singular as the radius grows); damp_diagU == nullptr: the caller has damped S already
    const double* damp_diagU; const int* damp_mask; double damp_radius, damp_min, damp_max;
};

template <bool HAS_EXT>
__device__ __forceinline__ bool forward_panels(SolverLds& s, double* __restrict__ A, int ld, int k0, int k1, const SolverPlan pl,
                                               double* __restrict__ rhs, double* __restrict__ extA, int ext_ld, int na, double* __restrict__ extrhs)
{
    // Look-ahead schedule.  Per panel k (entering with L_kk and its inverse in s.D / s.W, every earlier update visible):
    //   1. all waves stage the panel's row blocks (global -> s.B), write L_kk out                          | lds barrier
    //   2. all waves: panel solve  X = B (L_kk^-1)'  on the fp64 MFMA -> s.B and global                   | lds barrier
    //   3. all waves: the NEXT pivot block: its trailing-update tiles first (pair (k+1,k+1), one 16x16 tile per
    //      wave, result to global and to s.D), or a plain load if this panel does not touch it            | lds barrier
    //   4. wave 0 factors pivot block k+1 (the long dependent chain of the whole solve) WHILE waves 1..3 run the rest
    //      of panel k's trailing update and its right-hand-side update                                    | full barrier
    // so the factorisation (12.7k cycles) no longer adds to the trailing update, the rhs update and the prefetches
    // (11k cycles together): measured per panel 29.6k -> see profiles/README.md.
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index in an SGPR: role tests become scalar branches
    const int r0 = tid >> 5, c = tid & 31;
    // slot of an outside block in the node's update buffer (s.Anc is ascending, na <= SAMAX)
    auto ext_slot = [&](int b) { int q = 0; for (int a = 1; a < SAMAX; ++a) q += (a < na && s.Anc[a] <= b) ? 1 : 0; return q; };
    // The plan arrays are read-only for the kernel: through a constant-address-space pointer their loads are scalar (s_load).
    // As plain global loads (the kernel also stores, so the compiler must assume aliasing) the two prow_start reads at the
    // top of every panel came with an s_waitcnt vmcnt(0) -- a wait for every outstanding store and atomic of the wave,
    // ~2k cycles per panel.
    typedef const int __attribute__((address_space(4)))* cint_p;
    const cint_p prow_start = (cint_p)pl.prow_start;
    bool ok = true;
    constexpr int NT = SWAVES - 2;       // waves that share the trailing update: all but wave 0 (pivot chain) and wave 4, which
                                         // shares wave 0's SIMD (the chain runs 8 % slower with a busy neighbour)
    const int li = lane & 15, lk = lane >> 4;
#define STAMP(i) do { if (pl.stamps && tid == 0) pl.stamps[(size_t)k * 16 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
    if (k0 >= k1) return true;
    // block-pair table of panel kk (rows in s.Rows3[kk % 3]): pairs (qi >= qj) in row-major order of the lower triangle
    auto fill_pairs = [&](int kk, int first, int step) {
        const int Rk = prow_start[kk + 1] - prow_start[kk], np = Rk * (Rk + 1) / 2;
        const int* Rw = s.Rows3[kk % 3];
        for (int pr = first; pr < np; pr += step) {
            int qi = 0, qj = pr; while (qj > qi) { qj -= qi + 1; ++qi; }
            const int bi = Rw[qi], bjb = Rw[qj];
            double* dst; int dld;
            if (HAS_EXT && bjb >= k1) { dst = extA + (size_t)(ext_slot(bi) * SNB) * ext_ld + ext_slot(bjb) * SNB; dld = ext_ld; }
            else { dst = A + (size_t)(bi * SNB) * ld + bjb * SNB; dld = ld; }
            const unsigned long long u = (unsigned long long)dst;
            s.Pair[kk & 1][pr] = make_int4(qi * SNB, qj * SNB, (int)(unsigned)u, (int)(unsigned)(u >> 32));
            s.PairLd[kk & 1][pr] = dld;
        }
    };
    // prologue: first pivot block, and the first panel's block-row list
    { const int q0 = prow_start[k0], Rq = prow_start[k0 + 1] - q0; if (tid < Rq) s.Rows3[k0 % 3][tid] = pl.prow[q0 + tid]; }
    if (k0 + 1 < k1) { const int q1 = prow_start[k0 + 1], Rq = prow_start[k0 + 2] - q1; if (tid >= 64 && tid - 64 < Rq) s.Rows3[(k0 + 1) % 3][tid - 64] = pl.prow[q1 + tid - 64]; }
    for (int r = r0; r < SNB; r += SROWS) s.D[r][c] = A[(size_t)(k0 * SNB + r) * ld + k0 * SNB + c];
    __syncthreads();
    if (wave == 0 && !(pl.dbg & 1)) ok = wave_chol32(s, lane) && ok;
    else if (wave > 0) fill_pairs(k0, tid - 64, STHREADS - 64);
    __syncthreads();
    double rhs_next = 0.0; bool rhs_in_reg = false;      // wave 4, lanes 0..31: the next panel's right-hand-side block, carried in a register
    for (int k = k0; k < k1; ++k) {
        const int p0 = prow_start[k], R = prow_start[k + 1] - p0;
        const int npairs = R * (R + 1) / 2;
        const bool has_next = k + 1 < k1;
        STAMP(0);
        int* Rows = s.Rows3[k % 3];       // fetched two panels ago: no index load in front of the staging
        const bool next_diag = has_next && R > 0 && __builtin_amdgcn_readfirstlane(Rows[0]) == k + 1;     // this panel updates the next pivot block
        // requests whose latency hides behind the staging and the panel solve: this wave's tile of the next pivot block
        // and the right-hand-side entries waves 1..3 update in step 4
        v4d old0 = { 0.0, 0.0, 0.0, 0.0 };
        if (next_diag && wave < 4) {
            const double* src = A + (size_t)((k + 1) * SNB + 16 * (wave >> 1) + lk) * ld + (k + 1) * SNB + 16 * (wave & 1) + li;
#pragma unroll
            for (int g = 0; g < 4; ++g) old0[g] = src[(size_t)(4 * g) * ld];
        }
        double rhs_old[SRMAX / 2];           // wave 4 owns the right-hand-side update: rows lane, lane + 64, ...
#pragma unroll
        for (int u = 0; u < SRMAX / 2; ++u) {
            const int t = lane + 64 * u;
            rhs_old[u] = 0.0;
            if (wave == 4 && t < R * SNB) {
                const int bi = Rows[t >> 5];
                rhs_old[u] = (HAS_EXT && bi >= k1) ? extrhs[ext_slot(bi) * SNB + (t & 31)] : rhs[bi * SNB + (t & 31)];
            }
        }
        // 1. No staging pass: every wave requests the 16 rows of its own solve tile straight into the MFMA operand layout
        //    (a lane's eight values sit 32 B apart in one row), so the unsolved rows never go through LDS and the request is in
        //    flight across the write-back and the barrier below (staging was a 1.6k-cycle L2 round trip plus an LDS write,
        //    a barrier and an LDS read in front of the first MFMA).  The rhs row goes to LDS; L_kk and its inverse to global.
        const int nrows = R * SNB;
        auto tile_src = [&](int rt) -> const double* {
            const int row = 16 * rt + li;
            return A + (size_t)(Rows[row >> 5] * SNB + (row & 31)) * ld + k * SNB + lk;
        };
        double a_pre[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) a_pre[kk] = 0.0;
        if (wave * 16 < nrows && !(pl.dbg & 8)) {
            const double* src = tile_src(wave);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) a_pre[kk] = src[4 * kk];
        }
        {
            // this panel's right-hand-side block: the previous panel's wave 4 still holds it when it was the one to update it
            if (rhs_in_reg) { if (wave == 4 && lane < 32) s.B[R * SNB][lane] = rhs_next; }
            else if (wave == 3 && lane < 32) s.B[R * SNB][lane] = rhs[k * SNB + lane];
            for (int r = r0; r < SNB; r += SROWS) {
                A[(size_t)(k * SNB + r) * ld + k * SNB + c] = s.D[r][c];
                pl.linv[(size_t)k * SNB * SNB + r * SNB + c] = s.W[SNB + r][c];
            }
        }
        lds_barrier();
        STAMP(1);
        // 2. panel rows x L^-T = B (L^-1)' on the fp64 MFMA: one 16-row tile per wave at a time, both 16-column halves
        // (columns < 16 only see k < 16: L^-1 is lower triangular).  The tile's operand rows are in registers before
        // the results overwrite them; tiles of different waves touch disjoint rows.
        // (the right-hand-side row is solved on the VALU by the last wave: as a ninth 16-row tile it cost panels with four row
        //  blocks a second round of tiles)
        if (wave == SWAVES - 1 && !(pl.dbg & 2)) {
            const int j = lane & 31;
            double z4[4] = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
            for (int m = 0; m < SNB; ++m) z4[m & 3] = fma(s.B[R * SNB][m], s.W[SNB + m][j], z4[m & 3]);
            const double z = (z4[0] + z4[1]) + (z4[2] + z4[3]);
            if (lane < SNB) { s.B[R * SNB][j] = z; rhs[k * SNB + j] = z; }
        }
        for (int rt = wave; rt * 16 < nrows && !(pl.dbg & 2); rt += SWAVES) {
            double a[8];
            if (rt == wave) {
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) a[kk] = a_pre[kk];
            } else {                        // more than SWAVES tiles (five or more row blocks): later tiles are requested here
                const double* src = tile_src(rt);
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) a[kk] = src[4 * kk];
            }
            v4d x0 = { 0.0, 0.0, 0.0, 0.0 }, x1 = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], s.W[SNB + 4 * kk + lk][li], x0, 0, 0, 0);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], s.W[SNB + 4 * kk + lk][16 + li], x1, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int t = 16 * rt + lk + 4 * g;
                if (t < nrows) {
                    s.B[t][li] = x0[g]; s.B[t][16 + li] = x1[g];
                    double* gp = A + (size_t)(Rows[t >> 5] * SNB + (t & 31)) * ld + k * SNB;
                    gp[li] = x0[g]; gp[16 + li] = x1[g];
                }
            }
        }
        lds_barrier();                      // the solved rows are in LDS; their global copies are not read again in this kernel
        STAMP(4);
        // 3. the next pivot block into s.D.  Operand / result maps of v_mfma_f64_16x16x4_f64: A[i = l&15][k = l>>4],
        //    B[k = l>>4][j = l&15], D[row = (l>>4) + 4*reg][col = l&15].
        if (has_next && !(pl.dbg & 4)) {
            if (next_diag) {                // pair 0 = (Rows[0], Rows[0]) = (k+1, k+1): tile (tr, tc) = (wave >> 1, wave & 1), waves 0..3
              if (wave < 4) {
                const int tr = wave >> 1, tc = wave & 1;
                v4d acc = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
                for (int kk = 0; kk < 8; ++kk)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(s.B[16 * tr + li][4 * kk + lk], s.B[16 * tc + li][4 * kk + lk], acc, 0, 0, 0);
                double* dst = A + (size_t)((k + 1) * SNB + 16 * tr + lk) * ld + (k + 1) * SNB + 16 * tc + li;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const double v = old0[g] - acc[g];
                    dst[(size_t)(4 * g) * ld] = v;
                    s.D[16 * tr + lk + 4 * g][16 * tc + li] = v;
                }
              }
            } else {
                for (int r = r0; r < SNB; r += SROWS) s.D[r][c] = A[(size_t)((k + 1) * SNB + r) * ld + (k + 1) * SNB + c];
            }
        }
        lds_barrier();
        STAMP(5);
        // 4. wave 0: factor pivot block k+1; waves 1..3: the rest of the trailing update  A_ij -= L_ik L_jk'  (tiles dealt
        //    round-robin, old values of a chunk of target tiles requested first) and  rhs_i -= L_ik z_k.
        if (wave == 0) {
            if (has_next && !(pl.dbg & 1)) ok = wave_chol32(s, lane) && ok;
            STAMP(2);
        } else if (!(pl.dbg & 4)) {
            // this wave's block pairs P = p_first + w3 + NT i: a whole 32x32 target per step (2 x 2 tiles of 16 x 16 from two
            // A-operand and two B-operand row tiles: 32 LDS reads for 32 MFMAs; tile by tile it was 32 reads for 16 and
            // the six waves saturated the LDS pipe: 1.7k of 2.9k cycles per tile pair).
            // A_ij -= X_i X_j' goes out as a hardware fp64 atomic add of the negated product (no return value): each element
            // is touched by exactly one lane per panel and the panels are separated by barriers, so the result is the same
            // single rounding as old - acc in a fixed order, without a read of the old block in front of it.
            const int p_first = next_diag ? 1 : 0, w3 = wave < 4 ? wave - 1 : wave - 2;
            const int n_mine = (wave == 4) ? 0 : (npairs - p_first - w3 + NT - 1) / NT;
            if (wave == 4 && has_next) fill_pairs(k + 1, lane, 64);
            for (int i = 0; i < n_mine; ++i) {
                const int P = p_first + w3 + NT * i;
                const int4 d = s.Pair[k & 1][P];
                const int dld = s.PairLd[k & 1][P];
                double* dst = (double*)(((unsigned long long)(unsigned)d.w << 32) | (unsigned long long)(unsigned)d.z) + (size_t)lk * dld + li;
                const double* pa = &s.B[d.x + li][lk];
                const double* pb = &s.B[d.y + li][lk];
                double a0[8], a1[8], b0[8], b1[8];
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) { a0[kk] = pa[4 * kk]; b0[kk] = pb[4 * kk]; a1[kk] = pa[16 * SLD + 4 * kk]; b1[kk] = pb[16 * SLD + 4 * kk]; }
                v4d c00 = { 0.0, 0.0, 0.0, 0.0 }, c01 = c00, c10 = c00, c11 = c00;
                if (!(pl.dbg & 32)) {
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) {
                        c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[kk], b0[kk], c00, 0, 0, 0);
                        c01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[kk], b1[kk], c01, 0, 0, 0);
                        c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[kk], b0[kk], c10, 0, 0, 0);
                        c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[kk], b1[kk], c11, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    global_add_f64(dst + (size_t)(4 * g) * dld, -c00[g]);
                    global_add_f64(dst + (size_t)(4 * g) * dld + 16, -c01[g]);
                    global_add_f64(dst + (size_t)(16 + 4 * g) * dld, -c10[g]);
                    global_add_f64(dst + (size_t)(16 + 4 * g) * dld + 16, -c11[g]);
                }
            }
            if (pl.stamps && tid == 64) pl.stamps[(size_t)k * 16 + 7] = (long long)__builtin_amdgcn_s_memtime();
#pragma unroll
            for (int u = 0; u < SRMAX / 2; ++u) {
                const int t = lane + 64 * u;
                if (wave == 4 && t < R * SNB) {
                    double v4[4] = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
                    for (int m = 0; m < SNB; m += 4)
#pragma unroll
                        for (int j = 0; j < 4; ++j) v4[j] = fma(s.B[t][m + j], s.B[R * SNB][m + j], v4[j]);
                    const double v = (v4[0] + v4[1]) + (v4[2] + v4[3]);
                    const int bi = Rows[t >> 5];
                    if (HAS_EXT && bi >= k1) extrhs[ext_slot(bi) * SNB + (t & 31)] = rhs_old[u] - v;
                    else rhs[bi * SNB + (t & 31)] = rhs_old[u] - v;
                    if (u == 0 && t < SNB) rhs_next = rhs_old[u] - v;       // rows 0..31 are block Rows[0] (= k + 1 if next_diag)
                }
            }
            // the row list of panel k + 2 (slot (k + 2) % 3 is free: panel k - 1 is done)
            if (wave == 4 && k + 2 < k1) { const int q2 = prow_start[k + 2], R2 = prow_start[k + 3] - q2; if (lane < R2) s.Rows3[(k + 2) % 3][lane] = pl.prow[q2 + lane]; }
            if (pl.stamps && tid == 64) pl.stamps[(size_t)k * 16 + 3] = (long long)__builtin_amdgcn_s_memtime();
        }
        rhs_in_reg = next_diag && !(pl.dbg & 4);
        __syncthreads();                    // full: the next panel stages from what this one wrote to global
        STAMP(6);
    }
#undef STAMP
    return ok;
}

// L' y = z for panels k_hi-1 .. k_lo (descending); sy holds z for those panels and y for every later block they use.
//     t = z_k - sum_i L_ik' y_i,   y_k = (L_kk^-1)' t
// with the inverse the forward sweep left in pl.linv (one 32x32 product instead of a 32-step substitution chain), and
// the next panel's blocks requested one panel ahead (their addresses do not depend on y): a panel costs two barriers
// and two short LDS passes instead of an L2 round trip plus the chain.
__device__ __forceinline__ void backward_panels(SolverLds& s, const double* __restrict__ A, int ld, int k_lo, int k_hi,
                                                const SolverPlan pl, double* sy)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index in an SGPR: role tests become scalar branches
    const int r0 = tid >> 5, c = tid & 31;          // this thread's elements: rows r0 and r0 + 16 of column c
    if (k_hi <= k_lo) return;
    typedef const int __attribute__((address_space(4)))* cint_p;
    const cint_p prow_start = (cint_p)pl.prow_start; const cint_p prow = (cint_p)pl.prow;
    struct PanelRegs { double lb[SRMAX][2]; double wt[2]; int rows[SRMAX]; int R; };
    auto load_panel = [&](PanelRegs& P, int k) {
        const int p0 = prow_start[k];
        P.R = prow_start[k + 1] - p0;
#pragma unroll
        for (int q = 0; q < SRMAX; ++q) {
            P.rows[q] = 0; P.lb[q][0] = 0.0; P.lb[q][1] = 0.0;
            if (q < P.R) {
                const int i = prow[p0 + q];
                P.rows[q] = i;
                P.lb[q][0] = A[(size_t)(i * SNB + r0) * ld + k * SNB + c];
                P.lb[q][1] = A[(size_t)(i * SNB + r0 + 16) * ld + k * SNB + c];
            }
        }
        P.wt[0] = pl.linv[(size_t)k * SNB * SNB + r0 * SNB + c];
        P.wt[1] = pl.linv[(size_t)k * SNB * SNB + (r0 + 16) * SNB + c];
    };
    PanelRegs cur, nxt;
    load_panel(cur, k_hi - 1);
    nxt = cur;
    for (int k = k_hi - 1; k >= k_lo; --k) {
        if (k > k_lo) load_panel(nxt, k - 1);
        double part = 0.0;
#pragma unroll
        for (int q = 0; q < SRMAX; ++q)
            if (q < cur.R) part += cur.lb[q][0] * sy[cur.rows[q] * SNB + r0] + cur.lb[q][1] * sy[cur.rows[q] * SNB + r0 + 16];
        s.Red[r0][c] = part;
        s.D[r0][c] = cur.wt[0]; s.D[r0 + 16][c] = cur.wt[1];
        __syncthreads();
        if (wave == 0) {
            const int l = lane & 31;
            double t = sy[k * SNB + l];
#pragma unroll
            for (int rr = 0; rr < SROWS; ++rr) t -= s.Red[rr][l];
            double y4[4] = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
            for (int m = 0; m < SNB; ++m) y4[m & 3] = fma(s.D[l][m], readlane_f64(t, m), y4[m & 3]);
            if (lane < 32) sy[k * SNB + l] = (y4[0] + y4[1]) + (y4[2] + y4[3]);
        }
        __syncthreads();
        cur = nxt;
    }
}

// rows [i0, i1) of S get their damping term (see SolverPlan); the caller synchronises before the rows are read
__device__ __forceinline__ void damp_rows(double* __restrict__ A, int ld, const SolverPlan& pl, int i0, int i1)
{
    if (!pl.damp_diagU) return;
    for (int i = i0 + (int)threadIdx.x; i < i1; i += STHREADS) {
        double* d = A + (size_t)i * ld + i;
        if (pl.damp_mask[i] && pl.damp_diagU[i] > 0.0) *d += fmin(fmax(pl.damp_diagU[i], pl.damp_min), pl.damp_max) / pl.damp_radius;      // diagU == 0: no residual touches the parameter (see SolverPlan)
        else *d = 1.0;
    }
}

// Assembly step of a node (see FoldEnt).  The node's entries are first copied to LDS by one coalesced burst (read through
// the global table every field is a dependent ~2 us round trip, and hipcc re-reads them after each store: the pass took
// 27-43k cycles that way).  Then a thread owns one 16-byte element pair of every destination block; DB destinations per
// batch with every load of the batch in flight before the first add; sources are summed in list order (fixed =>
// deterministic).  NSRC sources per destination are requested up front, longer lists continue one by one.
#define SFOLD_LDS 28        // entries staged in LDS per pass (a separator has ~14 + 4 right-hand-side pieces, the top ~10 + 4): 12 blocks + 16 pieces
template <int DB, int NSRC>
__device__ __forceinline__ void fold_node(FoldEnt* __restrict__ sE, const FoldEnt* __restrict__ ents, int e0, int e1, int e2, const SolverPlan& pl)
{
    const int tid = threadIdx.x;
    const int r = tid >> 4, c = (tid & 15) * 2;
    for (int base = e0; base < e2; base += SFOLD_LDS) {
        const int cnt = e2 - base < SFOLD_LDS ? e2 - base : SFOLD_LDS;
        __syncthreads();
        {   // sizeof(FoldEnt) is a multiple of 8
            const long long* src = (const long long*)(ents + base);
            long long* dst = (long long*)sE;
            for (int i = tid; i < cnt * (int)(sizeof(FoldEnt) / 8); i += STHREADS) dst[i] = src[i];
        }
        __syncthreads();
        const int nb_blk = e1 - base < cnt ? (e1 - base > 0 ? e1 - base : 0) : cnt;       // block entries in this pass: [0, nb_blk), right-hand-side entries after them
        // right-hand-side entries: one thread per (destination, element) -- up to 16 destinations; their loads are issued
        // together with the first batch of blocks
        const int rdi = nb_blk + (tid >> 5), re = tid & 31;
        const bool rhs_mine = rdi < cnt;
        double rv = 0.0, rq[NSRC];
        if (rhs_mine) {
            const FoldEnt& E = sE[rdi];
            if (E.diag0 != -2) rv = E.dst[re];
#pragma unroll
            for (int q = 0; q < NSRC; ++q)
                if (q < E.nsrc) rq[q] = E.src[q][re];
        }
        for (int d0 = 0; d0 < nb_blk || d0 == 0; d0 += DB) {
            v2d acc[DB], t[DB][NSRC];
            double dmp[DB]; int msk[DB];
#pragma unroll
            for (int u = 0; u < DB; ++u)
                if (d0 + u < nb_blk) {
                    const FoldEnt& E = sE[d0 + u];
                    acc[u] = v2d{ 0.0, 0.0 };
                    if (E.diag0 != -2) acc[u] = *(const v2d*)(E.dst + (size_t)r * E.dst_ld + c);
#pragma unroll
                    for (int q = 0; q < NSRC; ++q)
                        if (q < E.nsrc) t[u][q] = *(const v2d*)(E.src[q] + (size_t)r * E.src_ld[q] + c);
                    dmp[u] = 0.0; msk[u] = 1;
                    if (E.diag0 >= 0 && pl.damp_diagU && (r >> 1) == (tid & 15)) { dmp[u] = pl.damp_diagU[E.diag0 + r]; msk[u] = pl.damp_mask[E.diag0 + r] && dmp[u] > 0.0; }
                }
            if (d0 == 0 && rhs_mine) {
                const FoldEnt& E = sE[rdi];
#pragma unroll
                for (int q = 0; q < NSRC; ++q)
                    if (q < E.nsrc) rv += rq[q];
                for (int q = NSRC; q < E.nsrc; ++q) rv += E.src[q][re];
                E.dst[re] = rv;
            }
#pragma unroll
            for (int u = 0; u < DB; ++u)
                if (d0 + u < nb_blk) {
                    const FoldEnt& E = sE[d0 + u];
#pragma unroll
                    for (int q = 0; q < NSRC; ++q)
                        if (q < E.nsrc) acc[u] += t[u][q];
                    for (int q = NSRC; q < E.nsrc; ++q) acc[u] += *(const v2d*)(E.src[q] + (size_t)r * E.src_ld[q] + c);
                    if (E.diag0 >= 0 && pl.damp_diagU && (r >> 1) == (tid & 15)) {      // this thread's pair holds the diagonal element (r, r)
                        const double add = fmin(fmax(dmp[u], pl.damp_min), pl.damp_max) / pl.damp_radius;
                        if (r & 1) acc[u].y = msk[u] ? acc[u].y + add : 1.0; else acc[u].x = msk[u] ? acc[u].x + add : 1.0;
                    }
                    *(v2d*)(E.dst + (size_t)r * E.dst_ld + c) = acc[u];
                }
        }
        for (int di = rdi + STHREADS / SNB; di < cnt; di += STHREADS / SNB) {      // more than 16 right-hand-side pieces in one pass (not at these sizes)
            const FoldEnt& E = sE[di];
            double v = E.diag0 != -2 ? E.dst[re] : 0.0;
            for (int q = 0; q < E.nsrc; ++q) v += E.src[q][re];
            E.dst[re] = v;
        }
    }
}

// single workgroup: whole factorisation + both substitutions (no dissection)
__global__ __launch_bounds__(STHREADS) void chol_sparse_kernel(double* __restrict__ A, int ld, SolverPlan pl,
                                                               double* __restrict__ rhs, double* __restrict__ y, int* __restrict__ err)
{
    __shared__ SolverLds s;
    const int tid = threadIdx.x;
    damp_rows(A, ld, pl, 0, pl.nb * SNB);
    __syncthreads();
    const bool ok = forward_panels<false>(s, A, ld, 0, pl.nb, pl, rhs, nullptr, 0, 0, nullptr);
    if (!ok && tid == 0) *err = 2;
    double* sy = &s.B[0][0];
    const int n = pl.nb * SNB;
    for (int i = tid; i < n; i += STHREADS) sy[i] = rhs[i];
    __syncthreads();
    backward_panels(s, A, ld, 0, pl.nb, pl, sy);
    for (int i = tid; i < n; i += STHREADS) y[i] = sy[i];
}

// ---- one level of the dissection: a workgroup per node ------------------------------------------------------------
__global__ __launch_bounds__(STHREADS) void chol_node_forward_kernel(double* __restrict__ A, int ld, SolverPlan pl, const NodeDesc* __restrict__ nodes, int first,
                                                                     double* __restrict__ rhs, double* __restrict__ ubuf,
                                                                     const FoldEnt* __restrict__ ents, int* __restrict__ err)
{
    __shared__ SolverLds s;
#define NSTAMP(i) do { if (pl.stamps && threadIdx.x == 0) pl.stamps[(size_t)nd.k0 * 16 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
    const long long t_in = pl.stamps ? (long long)__builtin_amdgcn_s_memtime() : 0;
    const NodeDesc nd = nodes[first + blockIdx.x];
    if (pl.stamps && threadIdx.x == 0) pl.stamps[(size_t)nd.k0 * 16 + 8] = t_in;
    NSTAMP(9);
    if (threadIdx.x < SAMAX) s.Anc[threadIdx.x] = nd.anc[threadIdx.x];
    if (nd.e0 == nd.e2) damp_rows(A, ld, pl, nd.k0 * SNB, nd.k1 * SNB);       // a leaf: nothing to assemble, only the damping
    else fold_node<8, 2>((FoldEnt*)&s.B[0][0], ents, nd.e0, nd.e1, nd.e2, pl);  // two children per separator; the entries are staged in s.B (free until the panels start)
    __syncthreads();
    NSTAMP(10);
    double* U = ubuf + nd.u_off;
    const int ldu = nd.na * SNB;
    const bool ok = forward_panels<true>(s, A, ld, nd.k0, nd.k1, pl, rhs, U, ldu, nd.na, U + (size_t)ldu * ldu);
    if (!ok && threadIdx.x == 0) *err = 2;
    NSTAMP(11);
#undef NSTAMP
}

// the top node (last separators + intrinsics): assemble, factor serially, solve, and its share of the back-substitution
__global__ __launch_bounds__(STHREADS) void chol_top_kernel(double* __restrict__ A, int ld, SolverPlan pl, const NodeDesc* __restrict__ nodes, int top_node,
                                                            double* __restrict__ rhs, const FoldEnt* __restrict__ ents,
                                                            double* __restrict__ y, int* __restrict__ err)
{
    __shared__ SolverLds s;
    const int tid = threadIdx.x;
    const NodeDesc nd = nodes[top_node];
    const int t0 = nd.k0 * SNB, ntop = (nd.k1 - nd.k0) * SNB;
    fold_node<4, 4>((FoldEnt*)&s.B[0][0], ents, nd.e0, nd.e1, nd.e2, pl);       // four children at C4 (up to 8: the nodes of the last parallel level)
    __syncthreads();
    const bool ok = forward_panels<false>(s, A, ld, nd.k0, nd.k1, pl, rhs, nullptr, 0, 0, nullptr);
    if (!ok && tid == 0) *err = 2;
    double* sy = &s.B[0][0];
    for (int i = tid; i < ntop; i += STHREADS) sy[t0 + i] = rhs[t0 + i];
    __syncthreads();
    backward_panels(s, A, ld, nd.k0, nd.k1, pl, sy);
    for (int i = tid; i < ntop; i += STHREADS) y[t0 + i] = sy[t0 + i];
}

__global__ __launch_bounds__(STHREADS) void chol_node_backward_kernel(const double* __restrict__ A, int ld, SolverPlan pl, const NodeDesc* __restrict__ nodes, int first,
                                                                      const double* __restrict__ rhs, double* __restrict__ y)
{
    __shared__ SolverLds s;
    const int tid = threadIdx.x;
    const NodeDesc nd = nodes[first + blockIdx.x];
    double* sy = &s.B[0][0];
    // y of the outside blocks its panels reach (solved by later nodes), z of its own panels
    for (int i = tid; i < nd.na * SNB; i += STHREADS) { const int b = nd.anc[i >> 5]; sy[b * SNB + (i & 31)] = y[b * SNB + (i & 31)]; }
    for (int i = nd.k0 * SNB + tid; i < nd.k1 * SNB; i += STHREADS) sy[i] = rhs[i];
    __syncthreads();
    backward_panels(s, A, ld, nd.k0, nd.k1, pl, sy);
    for (int i = nd.k0 * SNB + tid; i < nd.k1 * SNB; i += STHREADS) y[i] = sy[i];
}
