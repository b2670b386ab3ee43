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
#define STHREADS 256    // 4 waves, one per SIMD: the full register file for the unrolled 32-double register rows
#define SROWS (STHREADS / 32)

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

struct DiagLds {
    double D[SNB][SLD];                 // diagonal block / its factor (lower)
    double LT[SNB][SNB];                // LT[m][c] = L[c][m]: column m of L contiguous for wave-uniform wide reads
    double Inv[SNB];
    double Col[2][SNB];
};
struct SolverLds : DiagLds {
    double B[SRMAX * SNB + 1][SLD];     // stacked row blocks of the panel + the rhs row; the y vector in the backward phase
    double Red[SNB][SLD];
    int Rows[SRMAX];
};

// One wave (row = lane & 31): in-place Cholesky of the 32x32 block in s.D (lower), L' -> s.LT, 1/diag -> s.Inv.
__device__ __forceinline__ bool wave_chol32(DiagLds& s, int lane)
{
    const int row = lane & 31;
    double a[SNB];
#pragma unroll
    for (int c = 0; c < SNB; ++c) a[c] = s.D[row][c];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < SNB; ++j) {
        const double d = readlane_f64(a[j], j);
        ok = ok && (d > 0.0) && (d < 1e300);
        const double sd = sqrt(d > 0.0 ? d : 1.0), inv = 1.0 / sd;
        a[j] = (row == j) ? sd : a[j] * inv;
        s.Col[j & 1][row] = a[j];
        s.LT[j][row] = a[j];
        if (row == j) s.Inv[j] = inv;
        wave_sync_lds();
#pragma unroll
        for (int c = j + 1; c < SNB; ++c) a[c] -= a[j] * s.Col[j & 1][c];
    }
    if (lane < 32) {
#pragma unroll
        for (int c = 0; c < SNB; ++c) s.D[row][c] = (c <= row) ? a[c] : 0.0;
    }
    return ok;
}

// x <- x L^-T for one row held in registers (column-oriented substitution)
__device__ __forceinline__ void row_trsm32(double x[SNB], const DiagLds& s)
{
#pragma unroll
    for (int m = 0; m < SNB; ++m) {
        x[m] *= s.Inv[m];
#pragma unroll
        for (int c = m + 1; c < SNB; ++c) x[c] -= x[m] * s.LT[m][c];
    }
}

// prow_start[k] .. prow_start[k+1]: ascending block rows i > k with L_ik != 0 (after fill)
__global__ __launch_bounds__(STHREADS) void chol_sparse_kernel(double* __restrict__ A, int ld, int nb,
                                                               const int* __restrict__ prow_start, const int* __restrict__ prow,
                                                               double* __restrict__ rhs, double* __restrict__ y, int* __restrict__ err)
{
    __shared__ SolverLds s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = tid >> 5, c = tid & 31;
    bool ok = true;

    for (int k = 0; k < nb; ++k) {
        const int p0 = prow_start[k], R = prow_start[k + 1] - p0;
        if (tid < R) s.Rows[tid] = prow[p0 + tid];
        for (int r = r0; r < SNB; r += SROWS) s.D[r][c] = A[(size_t)(k * SNB + r) * ld + k * SNB + c];
        __syncthreads();
        // wave 0 factors the diagonal block while waves 1.. stage the panel's row blocks and the rhs row
        if (wave > 0) {
            for (int e = tid - 64; e < R * SNB * SNB; e += STHREADS - 64) {
                const int q = e >> 10, rr = (e >> 5) & 31, cc = e & 31;
                s.B[q * SNB + rr][cc] = A[(size_t)(s.Rows[q] * SNB + rr) * ld + k * SNB + cc];
            }
            if (wave == 1 && lane < 32) s.B[R * SNB][lane] = rhs[k * SNB + lane];
        } else {
            ok = wave_chol32(s, lane) && ok;
        }
        __syncthreads();
        for (int r = r0; r < SNB; r += SROWS) A[(size_t)(k * SNB + r) * ld + k * SNB + c] = s.D[r][c];
        // panel rows x L^-T (one row per thread, registers)
        const int nrows = R * SNB + 1;
        for (int t = tid; t < nrows; t += STHREADS) {
            double x[SNB];
#pragma unroll
            for (int m = 0; m < SNB; ++m) x[m] = s.B[t][m];
            row_trsm32(x, s);
#pragma unroll
            for (int m = 0; m < SNB; ++m) s.B[t][m] = x[m];
            double* g = (t < R * SNB) ? A + (size_t)(s.Rows[t >> 5] * SNB + (t & 31)) * ld + k * SNB : rhs + k * SNB;
#pragma unroll
            for (int m = 0; m < SNB; ++m) g[m] = x[m];
        }
        __syncthreads();
        // trailing update: A_ij -= L_ik L_jk' for the panel's block pairs, rhs_i -= L_ik z_k
        for (int qi = 0; qi < R; ++qi)
            for (int qj = 0; qj <= qi; ++qj) {
                double acc[SNB / SROWS];
#pragma unroll
                for (int u = 0; u < SNB / SROWS; ++u) acc[u] = 0.0;
#pragma unroll
                for (int m = 0; m < SNB; ++m) {
                    const double bj = s.B[qj * SNB + c][m];
#pragma unroll
                    for (int u = 0; u < SNB / SROWS; ++u) acc[u] += s.B[qi * SNB + r0 + u * SROWS][m] * bj;
                }
#pragma unroll
                for (int u = 0; u < SNB / SROWS; ++u)
                    A[(size_t)(s.Rows[qi] * SNB + r0 + u * SROWS) * ld + s.Rows[qj] * SNB + c] -= acc[u];
            }
        for (int t = tid; t < R * SNB; t += STHREADS) {
            double v = 0.0;
#pragma unroll
            for (int m = 0; m < SNB; ++m) v += s.B[t][m] * s.B[R * SNB][m];
            rhs[s.Rows[t >> 5] * SNB + (t & 31)] -= v;
        }
        __syncthreads();
    }
    if (!ok && tid == 0) *err = 2;

    // ---- backward: L' y = z (z now sits in rhs); y accumulates in LDS (aliases s.B)
    double* sy = &s.B[0][0];
    const int n = nb * SNB;
    for (int i = tid; i < n; i += STHREADS) sy[i] = rhs[i];
    __syncthreads();
    for (int k = nb - 1; k >= 0; --k) {
        const int p0 = prow_start[k], R = prow_start[k + 1] - p0;
        for (int r = r0; r < SNB; r += SROWS) {
            s.D[r][c] = A[(size_t)(k * SNB + r) * ld + k * SNB + c];
            double part = 0.0;
            for (int q = 0; q < R; ++q) {
                const int i = prow[p0 + q];
                part += A[(size_t)(i * SNB + r) * ld + k * SNB + c] * sy[i * SNB + r];
            }
            s.Red[r][c] = part;
        }
        __syncthreads();
        if (wave == 0) {
            const int l = lane & 31;
            double t = sy[k * SNB + l];
#pragma unroll
            for (int rr = 0; rr < SNB; ++rr) t -= s.Red[rr][l];
            const double inv = 1.0 / s.D[l][l];
#pragma unroll
            for (int cc = SNB - 1; cc >= 0; --cc) {
                const double yc = readlane_f64(t * inv, cc);
                if (l == cc) t = yc;
                else if (l < cc) t -= s.D[cc][l] * yc;
            }
            if (lane < 32) sy[k * SNB + l] = t;
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += STHREADS) y[i] = sy[i];
}
