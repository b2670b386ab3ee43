// ba.hip -- bundle adjustment on gfx950: replaces bundle_adjustment() = ceres::Solve with SPARSE_SCHUR
// (NViewReconstuct.cpp:1162-1244).  Device kernels: ba_kernels.hpp.  This file: problem set-up (per-point and
// per-camera orderings, camera-pair lists), the reduced-system solver (blocked right-looking Cholesky in fp64 +
// blocked triangular solves) and the Levenberg-Marquardt control loop with Ceres' trust-region semantics
// (the same control flow as oracle/orc_ba.c, which documents the [3P] sources).
//
// Per LM iteration the host reads one small block of scalars (one stream sync); everything else stays in HBM.
// Multi-GPU: each rank owns a shard of the points (all their observations) and replicas of the cameras; the
// only exchange is the sum of the reduced-system message [S | rhs | diagU | graw | scalars] plus 4 step scalars,
// done through the caller's all-reduce hook (RCCL over xGMI in production, gloo in the CPU tests of the host logic).
#include "common.hpp"
#include "ba_kernels.hpp"
#include "ba_solver.hpp"
#include <algorithm>
#include <chrono>
#include <cmath>

// ------------------------------------------------------------------------------------------------
// reduced system: blocked Cholesky (NB = 32), lower triangle, row-major, ld = npad
// ------------------------------------------------------------------------------------------------
#define NB 32

// Dense path, step 1: one wave factors the 32x32 diagonal block k (wave_chol32) and writes L_kk and its inverse
// (row-major 32x32 at Linv + k*1024).
__global__ __launch_bounds__(64) void chol_diag_kernel(double* __restrict__ A, int ld, int k, double* __restrict__ Linv, int* __restrict__ err)
{
    __shared__ DiagLds s;
    const int lane = threadIdx.x, row = lane & 31;
    double* base = A + (size_t)(k * NB) * ld + k * NB;
    if (lane < 32)
#pragma unroll
        for (int c = 0; c < NB; ++c) s.D[row][c] = base[(size_t)row * ld + c];
    __syncthreads();
    const bool ok = wave_chol32(s, lane);
    if (!ok && lane == 0) *err = 2;
    __syncthreads();
    if (lane < 32)
#pragma unroll
        for (int c = 0; c < NB; ++c) base[(size_t)row * ld + c] = s.D[row][c];
    // X = L^-1: row i of X is e_i L^-T ... computed as x L^-T with x = e_row, i.e. lane = row of L^-T = column of L^-1
    double x[NB];
#pragma unroll
    for (int m = 0; m < NB; ++m) x[m] = (m == row) ? 1.0 : 0.0;
    row_trsm32(x, s);                      // x = e_row L^-T  =>  x[c] = (L^-1)[c][row]
    if (lane < 32) {
        double* out = Linv + (size_t)k * NB * NB;
#pragma unroll
        for (int i = 0; i < NB; ++i) out[i * NB + row] = x[i];
    }
}

// L_ik = A_ik * Linv_k'   for row blocks i = k+1+blockIdx.x
__global__ __launch_bounds__(256) void chol_trsm_kernel(double* __restrict__ A, int ld, int k, const double* __restrict__ Linv)
{
    __shared__ double sB[NB][NB + 1], sL[NB][NB + 1];
    const int i = k + 1 + blockIdx.x;
    double* blk = A + (size_t)(i * NB) * ld + k * NB;
    const double* li = Linv + (size_t)k * NB * NB;
    const int tid = threadIdx.x;
    for (int e = tid; e < NB * NB; e += 256) { const int r = e / NB, c = e % NB; sB[r][c] = blk[(size_t)r * ld + c]; sL[r][c] = li[e]; }
    __syncthreads();
    for (int e = tid; e < NB * NB; e += 256) {
        const int r = e / NB, c = e % NB;
        double s = 0.0;
#pragma unroll
        for (int m = 0; m < NB; ++m) s += sB[r][m] * sL[c][m];
        blk[(size_t)r * ld + c] = s;
    }
}

// A_ij -= L_ik L_jk'   for k < j <= i (2-D grid over the trailing blocks; upper-triangle blocks exit)
__global__ __launch_bounds__(256) void chol_syrk_kernel(double* __restrict__ A, int ld, int k)
{
    const int i = k + 1 + blockIdx.y, j = k + 1 + blockIdx.x;
    if (j > i) return;
    __shared__ double sI[NB][NB + 1], sJ[NB][NB + 1];
    const double* bi = A + (size_t)(i * NB) * ld + k * NB;
    const double* bj = A + (size_t)(j * NB) * ld + k * NB;
    double* c_ = A + (size_t)(i * NB) * ld + j * NB;
    const int tid = threadIdx.x;
    for (int e = tid; e < NB * NB; e += 256) { const int r = e / NB, c = e % NB; sI[r][c] = bi[(size_t)r * ld + c]; sJ[r][c] = bj[(size_t)r * ld + c]; }
    __syncthreads();
    for (int e = tid; e < NB * NB; e += 256) {
        const int r = e / NB, c = e % NB;
        double s = 0.0;
#pragma unroll
        for (int m = 0; m < NB; ++m) s += sI[r][m] * sJ[c][m];
        c_[(size_t)r * ld + c] -= s;
    }
}

// y = (L L')^-1 rhs with the diagonal-block inverses: one workgroup, rhs held in LDS.
__global__ __launch_bounds__(1024) void chol_solve_kernel(const double* __restrict__ A, int ld, int nb, const double* __restrict__ Linv,
                                                          const double* __restrict__ rhs, double* __restrict__ y)
{
    extern __shared__ __attribute__((aligned(16))) double b[];
    const int tid = threadIdx.x, n = nb * NB;
    const int r = tid >> 5, c = tid & 31;
    for (int i = tid; i < n; i += 1024) b[i] = rhs[i];
    __syncthreads();
    for (int k = 0; k < nb; ++k) {                      // forward: L z = rhs
        double p = Linv[(size_t)k * NB * NB + r * NB + c] * b[k * NB + c];
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) p += __shfl_xor(p, off);
        __syncthreads();
        if (c == 0) b[k * NB + r] = p;
        __syncthreads();
        for (int row = (k + 1) * NB + tid; row < n; row += 1024) {
            const double* Lr = A + (size_t)row * ld + k * NB;
            double s = 0.0;
#pragma unroll
            for (int m = 0; m < NB; ++m) s += Lr[m] * b[k * NB + m];
            b[row] -= s;
        }
        __syncthreads();
    }
    for (int k = nb - 1; k >= 0; --k) {                 // backward: L' y = z
        double p = Linv[(size_t)k * NB * NB + c * NB + r] * b[k * NB + c];   // thread (r, c): Linv[c][r] * z[c], reduce over c
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) p += __shfl_xor(p, off);
        __syncthreads();
        if (c == 0) b[k * NB + r] = p;
        __syncthreads();
        for (int col = tid; col < k * NB; col += 1024) {
            double s = 0.0;
#pragma unroll
            for (int m = 0; m < NB; ++m) s += A[(size_t)(k * NB + m) * ld + col] * b[k * NB + m];
            b[col] -= s;
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += 1024) y[i] = b[i];
}

__global__ void fill_kernel(double* __restrict__ p, size_t n, double v)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ------------------------------------------------------------------------------------------------
// host-side problem
// ------------------------------------------------------------------------------------------------
struct sfmhip_ba {
    sfmhip_ctx* ctx = nullptr;
    sfm_ba_options o;
    int nc = 0, np = 0, nobs = 0, fix0 = 0, fixK = 0, ncf = 0, n = 0, npad = 0, koff = 0, nbk = 0;
    int cam_split = 1, n_pt_blocks = 0, nblk = 0, rank = 0, world = 1;
    std::vector<void*> allocs;
    // parameters
    double *d_K = nullptr, *d_ext = nullptr, *d_pts = nullptr;
    double *d_Kc = nullptr, *d_extc = nullptr, *d_ptsc = nullptr;
    double *d_K0 = nullptr, *d_ext0 = nullptr, *d_pts0 = nullptr;
    // structure
    int *d_pt_start = nullptr, *d_ocam = nullptr, *d_opt = nullptr, *d_cam_start = nullptr, *d_cam_obs = nullptr;
    int *d_blk_cam = nullptr, *d_blk_start = nullptr, *d_items = nullptr;
    int *d_prow_start = nullptr, *d_prow = nullptr; bool use_sparse = false; int max_panel_rows = 0;
    std::vector<int> host_blk_cam;
    double* d_ouv = nullptr;
    // work
    double *d_scale_c = nullptr, *d_scale_p = nullptr, *d_Vinv = nullptr, *d_bp = nullptr, *d_WK = nullptr, *d_colsq_p = nullptr;
    double *d_msg = nullptr; size_t msg_count = 0;
    double *d_part_pt = nullptr, *d_part_cam = nullptr, *d_part_back = nullptr;
    double *d_Linv = nullptr, *d_y = nullptr, *d_back4 = nullptr, *d_cam2 = nullptr, *d_xnorm = nullptr;
    int* d_err = nullptr;
    double* h_scal = nullptr;      // pinned: [cost, gmax, mcc, cand, dn_p, xn_p, dn_c, xn_c, err]
    // all-reduce hook
    sfmhip_allreduce_fn ar_fn = nullptr; void* ar_user = nullptr;
    // LM state
    bool started = false;
    double radius = 0, nu = 2, x_cost = 0, x_norm = 0, gmax = 0, initial_cost = 0;
    int iter = 0, nsucc = 0, ninvalid = 0, termination = SFMHIP_BA_NO_CONVERGENCE;
    hipEvent_t ev[4] = { nullptr, nullptr, nullptr, nullptr };
    double phase_acc[4] = { 0, 0, 0, 0 }; int phase_cnt = 0;
};

template <typename T>
static int dalloc(sfmhip_ba* h, T** p, size_t count)
{
    void* q = nullptr;
    SFM_HIP_TRY(h->ctx, hipMalloc(&q, (count > 0 ? count : 1) * sizeof(T)));
    h->allocs.push_back(q);
    *p = (T*)q;
    return SFMHIP_OK;
}
template <typename T>
static int dupload(sfmhip_ba* h, T** p, const T* src, size_t count)
{
    int rc = dalloc(h, p, count); if (rc) return rc;
    if (count) SFM_HIP_TRY(h->ctx, hipMemcpyAsync(*p, src, count * sizeof(T), hipMemcpyHostToDevice, h->ctx->stream));
    return SFMHIP_OK;
}

static BADev make_dev(const sfmhip_ba* h, double radius)
{
    BADev P;
    memset(&P, 0, sizeof P);
    P.nc = h->nc; P.np = h->np; P.nobs = h->nobs; P.n = h->n; P.npad = h->npad; P.koff = h->koff;
    P.fix0 = h->fix0; P.fixK = h->fixK; P.cam_split = h->cam_split; P.world = h->world; P.rank = h->rank;
    P.huber_a = h->o.huber_delta;
    P.K = h->d_K; P.ext = h->d_ext; P.pts = h->d_pts; P.Kc = h->d_Kc; P.extc = h->d_extc; P.ptsc = h->d_ptsc;
    P.pt_start = h->d_pt_start; P.ocam = h->d_ocam; P.ouv = h->d_ouv;
    P.cam_start = h->d_cam_start; P.cam_obs = h->d_cam_obs; P.opt = h->d_opt;
    P.scale_c = h->d_scale_c; P.scale_p = h->d_scale_p;
    P.Vinv = h->d_Vinv; P.bp = h->d_bp; P.WK = h->d_WK; P.colsq_p = h->d_colsq_p;
    const size_t np2 = (size_t)h->npad * h->npad;
    P.S = h->d_msg; P.rhs = h->d_msg + np2; P.diagU = P.rhs + h->npad; P.graw = P.diagU + h->npad; P.scal = P.graw + h->npad;
    P.part_pt = h->d_part_pt; P.part_cam = h->d_part_cam; P.part_back = h->d_part_back;
    P.y = h->d_y;
    P.radius = radius; P.min_diag = h->o.min_lm_diagonal; P.max_diag = h->o.max_lm_diagonal;
    return P;
}

static int call_allreduce(sfmhip_ba* h, double* buf, size_t count)
{
    if (!h->ar_fn) return SFMHIP_OK;
    const int rc = h->ar_fn(h->ar_user, buf, count, (void*)h->ctx->stream);
    if (rc != 0) { h->ctx->last_error = "all-reduce hook failed"; return SFMHIP_E_COMM; }
    return SFMHIP_OK;
}

// linearise at the current parameters: message = [S | rhs | diagU | graw | scal], summed over ranks, damped.
static int enqueue_linearize(sfmhip_ba* h, double radius, bool damp)
{
    sfmhip_ctx* ctx = h->ctx;
    hipStream_t st = ctx->stream;
    BADev P = make_dev(h, radius);
    SFM_HIP_TRY(ctx, hipMemsetAsync(h->d_msg, 0, h->msg_count * sizeof(double), st));
    SFM_HIP_TRY(ctx, hipMemsetAsync(h->d_err, 0, sizeof(int), st));
    hipLaunchKernelGGL(ba_point_kernel, dim3(h->n_pt_blocks), dim3(256), 0, st, P, h->d_err);
    hipLaunchKernelGGL(ba_camera_kernel, dim3(h->nc, h->cam_split), dim3(256), 0, st, P);
    hipLaunchKernelGGL(ba_finalize_kernel, dim3(h->nc + 1), dim3(256), 0, st, P, h->n_pt_blocks);
    if (h->nblk > 0)
        hipLaunchKernelGGL(ba_schur_kernel, dim3(h->nblk), dim3(64), 0, st, P, h->d_blk_cam, h->d_blk_start, h->d_items);
    SFM_HIP_TRY(ctx, hipGetLastError());
    int rc = call_allreduce(h, h->d_msg, h->msg_count); if (rc) return rc;
    if (damp) { hipLaunchKernelGGL(ba_damp_kernel, dim3(1), dim3(256), 0, st, P); SFM_HIP_TRY(ctx, hipGetLastError()); }
    return SFMHIP_OK;
}

static int enqueue_solve(sfmhip_ba* h)
{
    sfmhip_ctx* ctx = h->ctx;
    hipStream_t st = ctx->stream;
    double* S = h->d_msg;
    const int nb = h->nbk, ld = h->npad;
    if (h->use_sparse) {
        double* rhs_rw = h->d_msg + (size_t)h->npad * h->npad;
        hipLaunchKernelGGL(chol_sparse_kernel, dim3(1), dim3(STHREADS), 0, st, S, ld, nb, h->d_prow_start, h->d_prow, rhs_rw, h->d_y, h->d_err);
        SFM_HIP_TRY(ctx, hipGetLastError());
        return SFMHIP_OK;
    }
    for (int k = 0; k < nb; ++k) {
        hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(64), 0, st, S, ld, k, h->d_Linv, h->d_err);
        const int m = nb - k - 1;
        if (m > 0) {
            hipLaunchKernelGGL(chol_trsm_kernel, dim3(m), dim3(256), 0, st, S, ld, k, h->d_Linv);
            hipLaunchKernelGGL(chol_syrk_kernel, dim3(m, m), dim3(256), 0, st, S, ld, k);
        }
    }
    const double* rhs = h->d_msg + (size_t)h->npad * h->npad;
    hipLaunchKernelGGL(chol_solve_kernel, dim3(1), dim3(1024), (size_t)h->npad * sizeof(double), st, S, ld, nb, h->d_Linv, rhs, h->d_y);
    SFM_HIP_TRY(ctx, hipGetLastError());
    return SFMHIP_OK;
}

static int enqueue_back(sfmhip_ba* h, double radius)
{
    sfmhip_ctx* ctx = h->ctx;
    hipStream_t st = ctx->stream;
    BADev P = make_dev(h, radius);
    hipLaunchKernelGGL(ba_camstep_kernel, dim3(1), dim3(256), 0, st, P, h->d_cam2);
    hipLaunchKernelGGL(ba_back_kernel, dim3(h->n_pt_blocks), dim3(256), 0, st, P);
    hipLaunchKernelGGL(ba_back_reduce_kernel, dim3(1), dim3(256), 0, st, h->d_part_back, h->n_pt_blocks, h->d_back4);
    SFM_HIP_TRY(ctx, hipGetLastError());
    return call_allreduce(h, h->d_back4, 4);
}

// Block fill pattern of the reduced system (32x32 blocks) -> per-panel row lists for chol_sparse_kernel.
// Multi-rank: the pattern is the union over ranks (S is summed), taken through the all-reduce hook.
static int build_solver_plan(sfmhip_ba* h)
{
    sfmhip_ctx* ctx = h->ctx;
    const int nb = h->nbk;
    std::vector<double> pat((size_t)nb * nb, 0.0);
    auto mark = [&](int lo_a, int hi_a, int lo_b, int hi_b) {       // scalar index ranges [lo, hi)
        for (int i = lo_a / NB; i <= (hi_a - 1) / NB; ++i)
            for (int j = lo_b / NB; j <= (hi_b - 1) / NB; ++j) { pat[(size_t)i * nb + j] = 1.0; pat[(size_t)j * nb + i] = 1.0; }
    };
    for (int i = 0; i < nb; ++i) pat[(size_t)i * nb + i] = 1.0;
    for (int c = h->fix0; c < h->nc; ++c) {
        const int co = 6 * (c - h->fix0);
        mark(co, co + 6, co, co + 6);
        if (!h->fixK) mark(co, co + 6, h->koff, h->koff + 4);
    }
    if (!h->fixK) mark(h->koff, h->koff + 4, h->koff, h->koff + 4);
    for (size_t b = 0; b + 1 < h->host_blk_cam.size(); b += 2) {
        const int oa = 6 * (h->host_blk_cam[b] - h->fix0), ob = 6 * (h->host_blk_cam[b + 1] - h->fix0);
        mark(oa, oa + 6, ob, ob + 6);
    }
    if (h->ar_fn) {
        SFM_HIP_TRY(ctx, hipMemcpyAsync(h->d_msg, pat.data(), pat.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        int rc = call_allreduce(h, h->d_msg, pat.size()); if (rc) return rc;
        SFM_HIP_TRY(ctx, hipMemcpyAsync(pat.data(), h->d_msg, pat.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        SFM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    std::vector<char> P((size_t)nb * nb, 0);
    for (int i = 0; i < nb; ++i) for (int j = 0; j <= i; ++j) P[(size_t)i * nb + j] = pat[(size_t)i * nb + j] != 0.0;
    std::vector<int> start(nb + 1, 0), rows;
    int maxR = 0;
    for (int k = 0; k < nb; ++k) {
        std::vector<int> rk;
        for (int i = k + 1; i < nb; ++i) if (P[(size_t)i * nb + k]) rk.push_back(i);
        for (size_t a = 0; a < rk.size(); ++a) for (size_t b = 0; b <= a; ++b) P[(size_t)rk[a] * nb + rk[b]] = 1;
        maxR = std::max(maxR, (int)rk.size());
        rows.insert(rows.end(), rk.begin(), rk.end());
        start[k + 1] = (int)rows.size();
    }
    h->max_panel_rows = maxR;
    h->use_sparse = maxR <= SRMAX && h->npad <= (SRMAX * SNB + 1) * SLD;
    if (getenv("SFMHIP_DENSE_SOLVER")) h->use_sparse = false;
    if (h->use_sparse) {
        if (!h->d_prow_start) { int rc = dalloc(h, &h->d_prow_start, (size_t)nb + 1); if (rc) return rc; rc = dalloc(h, &h->d_prow, (size_t)nb * SRMAX + 1); if (rc) return rc; }
        SFM_HIP_TRY(ctx, hipMemcpyAsync(h->d_prow_start, start.data(), start.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        if (!rows.empty()) SFM_HIP_TRY(ctx, hipMemcpyAsync(h->d_prow, rows.data(), rows.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        SFM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return SFMHIP_OK;
}

// iteration 0 work: jacobi scaling from the column norms at x0, |x0|
static int ba_start(sfmhip_ba* h)
{
    sfmhip_ctx* ctx = h->ctx;
    hipStream_t st = ctx->stream;
    { int rc = build_solver_plan(h); if (rc) return rc; }
    const size_t np3 = 3 * (size_t)h->np;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((h->npad + 255) / 256)), dim3(256), 0, st, h->d_scale_c, (size_t)h->npad, 1.0);
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((np3 + 255) / 256 + 1)), dim3(256), 0, st, h->d_scale_p, np3, 1.0);
    if (h->o.jacobi_scaling) {
        int rc = enqueue_linearize(h, h->o.initial_trust_region_radius, false); if (rc) return rc;
        const double* diagU = h->d_msg + (size_t)h->npad * h->npad + h->npad;
        hipLaunchKernelGGL(ba_scale_kernel, dim3((h->n + 255) / 256 + 1), dim3(256), 0, st, diagU, h->d_scale_c, h->n, 1);
        hipLaunchKernelGGL(ba_scale_kernel, dim3((unsigned)((np3 + 255) / 256 + 1)), dim3(256), 0, st, h->d_colsq_p, h->d_scale_p, (int)np3, 1);
    }
    BADev P = make_dev(h, h->o.initial_trust_region_radius);
    const int nb = 64;
    hipLaunchKernelGGL(ba_xnorm_kernel, dim3(nb), dim3(256), 0, st, P, h->d_xnorm);
    SFM_HIP_TRY(ctx, hipGetLastError());
    std::vector<double> part(nb);
    SFM_HIP_TRY(ctx, hipMemcpyAsync(part.data(), h->d_xnorm, nb * sizeof(double), hipMemcpyDeviceToHost, st));
    SFM_HIP_TRY(ctx, hipStreamSynchronize(st));
    double s = 0; for (double v : part) s += v;
    if (h->ar_fn) {
        // |x|^2 = camera part (replicated) + sum over ranks of the point parts: reduce only the point part
        // (cheap way: subtract the replicated part, reduce, add it back)
        BADev Q = P; (void)Q;
        double cam = 0;
        std::vector<double> K(4), ext(6 * (size_t)h->nc);
        SFM_HIP_TRY(ctx, hipMemcpy(K.data(), h->d_K, 4 * sizeof(double), hipMemcpyDeviceToHost));
        SFM_HIP_TRY(ctx, hipMemcpy(ext.data(), h->d_ext, ext.size() * sizeof(double), hipMemcpyDeviceToHost));
        if (!h->fixK) for (int i = 0; i < 4; ++i) cam += K[i] * K[i];
        for (int c = h->fix0; c < h->nc; ++c) for (int j = 0; j < 6; ++j) cam += ext[6 * c + j] * ext[6 * c + j];
        double pt = s - cam;
        SFM_HIP_TRY(ctx, hipMemcpy(h->d_back4, &pt, sizeof(double), hipMemcpyHostToDevice));
        int rc = call_allreduce(h, h->d_back4, 1); if (rc) return rc;
        SFM_HIP_TRY(ctx, hipStreamSynchronize(st));
        SFM_HIP_TRY(ctx, hipMemcpy(&pt, h->d_back4, sizeof(double), hipMemcpyDeviceToHost));
        s = cam + pt;
    }
    h->x_norm = std::sqrt(s);
    h->radius = h->o.initial_trust_region_radius; h->nu = 2.0;
    h->iter = 0; h->nsucc = 0; h->ninvalid = 0; h->started = true; h->termination = SFMHIP_BA_NO_CONVERGENCE;
    h->initial_cost = -1.0;
    return SFMHIP_OK;
}

// the LM loop.  forced: run exactly max_it more iterations, tolerance checks disabled.
static int ba_loop(sfmhip_ba* h, int max_it, bool forced)
{
    sfmhip_ctx* ctx = h->ctx;
    hipStream_t st = ctx->stream;
    const sfm_ba_options& o = h->o;
    if (!h->started) { int rc = ba_start(h); if (rc) return rc; }
    const int it_end = h->iter + max_it;
    const size_t np2 = (size_t)h->npad * h->npad;
    const double* d_scal = h->d_msg + np2 + 3 * (size_t)h->npad;
    for (;;) {
        if (h->iter >= it_end) { h->termination = SFMHIP_BA_NO_CONVERGENCE; break; }
        if (!forced && h->radius < o.min_trust_region_radius) { h->termination = SFMHIP_BA_CONVERGENCE; break; }
        SFM_HIP_TRY(ctx, hipEventRecord(h->ev[0], st));
        int rc = enqueue_linearize(h, h->radius, true); if (rc) return rc;
        SFM_HIP_TRY(ctx, hipEventRecord(h->ev[1], st));
        rc = enqueue_solve(h); if (rc) return rc;
        SFM_HIP_TRY(ctx, hipEventRecord(h->ev[2], st));
        rc = enqueue_back(h, h->radius); if (rc) return rc;
        SFM_HIP_TRY(ctx, hipEventRecord(h->ev[3], st));
        SFM_HIP_TRY(ctx, hipMemcpyAsync(h->h_scal, d_scal, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
        SFM_HIP_TRY(ctx, hipMemcpyAsync(h->h_scal + 2, h->d_back4, 4 * sizeof(double), hipMemcpyDeviceToHost, st));
        SFM_HIP_TRY(ctx, hipMemcpyAsync(h->h_scal + 6, h->d_cam2, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
        SFM_HIP_TRY(ctx, hipMemcpyAsync(h->h_scal + 8, h->d_err, sizeof(int), hipMemcpyDeviceToHost, st));
        SFM_HIP_TRY(ctx, hipStreamSynchronize(st));
        {
            float a = 0, b = 0, c = 0;
            (void)hipEventElapsedTime(&a, h->ev[0], h->ev[1]); (void)hipEventElapsedTime(&b, h->ev[1], h->ev[2]);
            (void)hipEventElapsedTime(&c, h->ev[2], h->ev[3]);
            h->phase_acc[0] += a; h->phase_acc[1] += b; h->phase_acc[2] += c; h->phase_acc[3] += a + b + c; h->phase_cnt++;
        }
        const double cost = h->h_scal[0], gmax = h->h_scal[1], mcc = h->h_scal[2], cand_raw = h->h_scal[3];
        const double dn = h->h_scal[4] + h->h_scal[6], xn = h->h_scal[5] + h->h_scal[7];
        int err = 0; memcpy(&err, h->h_scal + 8, sizeof(int));
        h->x_cost = cost; h->gmax = gmax;
        if (h->initial_cost < 0.0) h->initial_cost = cost;
        if (!std::isfinite(cost)) { h->termination = SFMHIP_BA_FAILURE; ctx->last_error = "non-finite cost"; break; }
        if (!forced && gmax <= o.gradient_tolerance) { h->termination = SFMHIP_BA_CONVERGENCE; break; }
        ++h->iter;
        bool accepted = false;
        if (err != 0 || !(mcc > 0.0) || !std::isfinite(mcc)) {
            if (++h->ninvalid >= 5 && !forced) { h->termination = SFMHIP_BA_FAILURE; break; }
            h->radius *= 0.5;
        } else {
            h->ninvalid = 0;
            const double cand = std::isfinite(cand_raw) ? cand_raw : DBL_MAX;
            const double step_norm = std::sqrt(dn);
            if (!forced && step_norm <= o.parameter_tolerance * (h->x_norm + o.parameter_tolerance)) { h->termination = SFMHIP_BA_CONVERGENCE; break; }
            const double cost_change = cost - cand;
            if (!forced && std::fabs(cost_change) <= o.function_tolerance * cost) { h->termination = SFMHIP_BA_CONVERGENCE; break; }
            const double rho = cost_change / mcc;
            if (rho > o.min_relative_decrease) {
                std::swap(h->d_K, h->d_Kc); std::swap(h->d_ext, h->d_extc); std::swap(h->d_pts, h->d_ptsc);
                h->x_norm = std::sqrt(xn); h->x_cost = cand;
                const double t = 2.0 * rho - 1.0;
                h->radius = h->radius / std::max(1.0 / 3.0, 1.0 - t * t * t);
                h->radius = std::min(o.max_trust_region_radius, h->radius);
                h->nu = 2.0; ++h->nsucc; accepted = true;
            } else {
                h->radius = h->radius / h->nu; h->nu *= 2.0;
            }
        }
        if (o.verbose)
            printf("[sfmhip_ba] it %d cost %.12e gmax %.3e radius %.3e %s\n", h->iter, h->x_cost, gmax, h->radius, accepted ? "ok" : "rejected");
    }
    return SFMHIP_OK;
}

static void fill_summary(const sfmhip_ba* h, sfm_ba_summary* s, double t_s)
{
    if (!s) return;
    s->termination = h->termination; s->iterations = h->iter; s->successful_steps = h->nsucc;
    s->num_residuals = 2 * h->nobs; s->initial_cost = h->initial_cost < 0 ? 0.0 : h->initial_cost; s->final_cost = h->x_cost;
    s->final_radius = h->radius; s->final_gradient_max_norm = h->gmax; s->total_time_s = t_s;
}

extern "C" {

void sfmhip_ba_default_options(sfm_ba_options* o)
{
    if (!o) return;
    o->max_num_iterations = 50;
    o->initial_trust_region_radius = 1e4;
    o->max_trust_region_radius = 1e16;
    o->min_trust_region_radius = 1e-32;
    o->min_relative_decrease = 1e-3;
    o->min_lm_diagonal = 1e-6;
    o->max_lm_diagonal = 1e32;
    o->function_tolerance = 1e-6;
    o->gradient_tolerance = 1e-10;
    o->parameter_tolerance = 1e-8;
    o->huber_delta = 4.0;
    o->jacobi_scaling = 1;
    o->fix_first_camera = 1;
    o->fix_intrinsics = 0;
    o->verbose = 0;
}

void sfmhip_ba_destroy(sfmhip_ba* h)
{
    if (!h) return;
    (void)hipStreamSynchronize(h->ctx->stream);
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->h_scal) (void)hipHostFree(h->h_scal);
    for (auto& e : h->ev) if (e) (void)hipEventDestroy(e);
    delete h;
}

int sfmhip_ba_create(sfmhip_ctx* ctx, const double* K4, const double* ext6, int n_cam, const double* pts, int n_pt,
                     const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_uv, int n_obs,
                     const sfm_ba_options* opts, sfmhip_ba** out)
{
    SFM_ARG_CHECK(ctx, ctx && out && K4 && ext6 && n_cam > 0 && n_pt >= 0 && n_obs >= 0);
    SFM_ARG_CHECK(ctx, (pts || n_pt == 0) && ((obs_cam && obs_pt && obs_uv) || n_obs == 0));
    for (int k = 0; k < n_obs; ++k)
        SFM_ARG_CHECK(ctx, obs_cam[k] >= 0 && obs_cam[k] < n_cam && obs_pt[k] >= 0 && obs_pt[k] < n_pt);
    sfmhip_ba* h = new sfmhip_ba();
    h->ctx = ctx;
    if (opts) h->o = *opts; else sfmhip_ba_default_options(&h->o);
    h->nc = n_cam; h->np = n_pt; h->nobs = n_obs;
    h->fix0 = h->o.fix_first_camera ? 1 : 0; h->fixK = h->o.fix_intrinsics ? 1 : 0;
    h->ncf = n_cam - h->fix0; h->koff = 6 * h->ncf; h->n = 6 * h->ncf + (h->fixK ? 0 : 4);
    h->npad = std::max(NB, round_up(h->n, NB)); h->nbk = h->npad / NB;
    h->n_pt_blocks = std::max(1, ceil_div(n_pt, 256));

    // ---- orderings (host, once per problem)
    std::vector<int> pt_start(n_pt + 1, 0), fill(n_pt, 0), ocam(n_obs), opt(n_obs), perm(n_obs);
    std::vector<double> ouv(2 * (size_t)n_obs);
    for (int k = 0; k < n_obs; ++k) pt_start[obs_pt[k] + 1]++;
    for (int p = 0; p < n_pt; ++p) pt_start[p + 1] += pt_start[p];
    for (int k = 0; k < n_obs; ++k) { const int p = obs_pt[k]; perm[pt_start[p] + fill[p]++] = k; }
    for (int q = 0; q < n_obs; ++q) { const int k = perm[q]; ocam[q] = obs_cam[k]; opt[q] = obs_pt[k]; ouv[2 * (size_t)q] = obs_uv[2 * (size_t)k]; ouv[2 * (size_t)q + 1] = obs_uv[2 * (size_t)k + 1]; }
    std::vector<int> cam_start(n_cam + 1, 0), cam_obs(n_obs), cfill(n_cam, 0);
    for (int q = 0; q < n_obs; ++q) cam_start[ocam[q] + 1]++;
    for (int c = 0; c < n_cam; ++c) cam_start[c + 1] += cam_start[c];
    for (int q = 0; q < n_obs; ++q) { const int c = ocam[q]; cam_obs[cam_start[c] + cfill[c]++] = q; }
    int max_cam = 1;
    for (int c = 0; c < n_cam; ++c) max_cam = std::max(max_cam, cam_start[c + 1] - cam_start[c]);
    h->cam_split = std::min(32, std::max(1, ceil_div(max_cam, 1024)));
    // camera-pair lists for the off-diagonal Schur blocks (and same-camera pairs)
    struct Item { long long key; int qi, qj; };
    std::vector<Item> items;
    for (int p = 0; p < n_pt; ++p)
        for (int i = pt_start[p]; i < pt_start[p + 1]; ++i)
            for (int j = i + 1; j < pt_start[p + 1]; ++j) {
                int ci = ocam[i], cj = ocam[j], qi = i, qj = j;
                if (ci < cj) { std::swap(ci, cj); std::swap(qi, qj); }
                if ((h->fix0 && ci == 0) || (h->fix0 && cj == 0)) continue;
                items.push_back({ (long long)ci * n_cam + cj, qi, qj });
            }
    std::stable_sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.key < b.key; });
    std::vector<int> blk_cam, blk_start, flat(2 * items.size());
    for (size_t t = 0; t < items.size(); ++t) {
        if (t == 0 || items[t].key != items[t - 1].key) {
            blk_cam.push_back((int)(items[t].key / n_cam)); blk_cam.push_back((int)(items[t].key % n_cam));
            blk_start.push_back((int)t);
        }
        flat[2 * t] = items[t].qi; flat[2 * t + 1] = items[t].qj;
    }
    blk_start.push_back((int)items.size());
    h->nblk = (int)blk_cam.size() / 2;
    h->host_blk_cam = blk_cam;

    int rc = SFMHIP_OK;
#define TRY_RC(x) do { rc = (x); if (rc) { sfmhip_ba_destroy(h); return rc; } } while (0)
    TRY_RC(dupload(h, &h->d_K, K4, 4)); TRY_RC(dupload(h, &h->d_ext, ext6, 6 * (size_t)n_cam)); TRY_RC(dupload(h, &h->d_pts, pts, 3 * (size_t)n_pt));
    TRY_RC(dupload(h, &h->d_K0, K4, 4)); TRY_RC(dupload(h, &h->d_ext0, ext6, 6 * (size_t)n_cam)); TRY_RC(dupload(h, &h->d_pts0, pts, 3 * (size_t)n_pt));
    TRY_RC(dupload(h, &h->d_Kc, K4, 4)); TRY_RC(dupload(h, &h->d_extc, ext6, 6 * (size_t)n_cam)); TRY_RC(dupload(h, &h->d_ptsc, pts, 3 * (size_t)n_pt));
    TRY_RC(dupload(h, &h->d_pt_start, pt_start.data(), pt_start.size())); TRY_RC(dupload(h, &h->d_ocam, ocam.data(), ocam.size()));
    TRY_RC(dupload(h, &h->d_opt, opt.data(), opt.size())); TRY_RC(dupload(h, &h->d_ouv, ouv.data(), ouv.size()));
    TRY_RC(dupload(h, &h->d_cam_start, cam_start.data(), cam_start.size())); TRY_RC(dupload(h, &h->d_cam_obs, cam_obs.data(), cam_obs.size()));
    TRY_RC(dupload(h, &h->d_blk_cam, blk_cam.data(), blk_cam.size())); TRY_RC(dupload(h, &h->d_blk_start, blk_start.data(), blk_start.size()));
    TRY_RC(dupload(h, &h->d_items, flat.data(), flat.size()));
    TRY_RC(dalloc(h, &h->d_scale_c, (size_t)h->npad)); TRY_RC(dalloc(h, &h->d_scale_p, 3 * (size_t)n_pt));
    TRY_RC(dalloc(h, &h->d_Vinv, 6 * (size_t)n_pt)); TRY_RC(dalloc(h, &h->d_bp, 3 * (size_t)n_pt));
    TRY_RC(dalloc(h, &h->d_WK, 12 * (size_t)n_pt)); TRY_RC(dalloc(h, &h->d_colsq_p, 3 * (size_t)n_pt));
    TRY_RC(dalloc(h, &h->d_part_pt, 16 * (size_t)h->n_pt_blocks)); TRY_RC(dalloc(h, &h->d_part_back, 4 * (size_t)h->n_pt_blocks));
    TRY_RC(dalloc(h, &h->d_part_cam, (size_t)CAMACC * n_cam * 32));
    TRY_RC(dalloc(h, &h->d_Linv, (size_t)h->nbk * NB * NB)); TRY_RC(dalloc(h, &h->d_y, (size_t)h->npad));
    TRY_RC(dalloc(h, &h->d_back4, 4)); TRY_RC(dalloc(h, &h->d_cam2, 2)); TRY_RC(dalloc(h, &h->d_xnorm, 64)); TRY_RC(dalloc(h, &h->d_err, 1));
    h->msg_count = (size_t)h->npad * h->npad + 3 * (size_t)h->npad + SCAL_GMAX_SLOTS + 64;   // room for <= 64 ranks
    TRY_RC(dalloc(h, &h->d_msg, h->msg_count));
#undef TRY_RC
    if (hipHostMalloc((void**)&h->h_scal, 16 * sizeof(double)) != hipSuccess) { sfmhip_ba_destroy(h); ctx->last_error = "hipHostMalloc"; return SFMHIP_E_HIP; }
    for (auto& e : h->ev) if (hipEventCreate(&e) != hipSuccess) { sfmhip_ba_destroy(h); ctx->last_error = "hipEventCreate"; return SFMHIP_E_HIP; }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { sfmhip_ba_destroy(h); ctx->last_error = "upload failed"; return SFMHIP_E_HIP; }
    *out = h;
    return SFMHIP_OK;
}

int sfmhip_ba_set_allreduce(sfmhip_ba* h, sfmhip_allreduce_fn fn, void* user, int rank, int world)
{
    if (!h || world < 1 || world > 64 || rank < 0 || rank >= world) return SFMHIP_E_ARG;
    h->ar_fn = fn; h->ar_user = user; h->rank = rank; h->world = world;
    h->started = false;
    return SFMHIP_OK;
}

int sfmhip_ba_reset(sfmhip_ba* h)
{
    if (!h) return SFMHIP_E_ARG;
    sfmhip_ctx* ctx = h->ctx;
    SFM_HIP_TRY(ctx, hipMemcpyAsync(h->d_K, h->d_K0, 4 * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    SFM_HIP_TRY(ctx, hipMemcpyAsync(h->d_ext, h->d_ext0, 6 * (size_t)h->nc * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    if (h->np) SFM_HIP_TRY(ctx, hipMemcpyAsync(h->d_pts, h->d_pts0, 3 * (size_t)h->np * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    h->started = false;
    h->phase_acc[0] = h->phase_acc[1] = h->phase_acc[2] = h->phase_acc[3] = 0; h->phase_cnt = 0;
    return SFMHIP_OK;
}

int sfmhip_ba_run(sfmhip_ba* h, sfm_ba_summary* summary)
{
    if (!h) return SFMHIP_E_ARG;
    const auto t0 = std::chrono::steady_clock::now();
    h->started = false;
    const int rc = ba_loop(h, h->o.max_num_iterations, false);
    fill_summary(h, summary, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    return rc;
}

int sfmhip_ba_iterate(sfmhip_ba* h, int n_iter, sfm_ba_summary* summary)
{
    if (!h || n_iter < 0) return SFMHIP_E_ARG;
    const auto t0 = std::chrono::steady_clock::now();
    h->phase_acc[0] = h->phase_acc[1] = h->phase_acc[2] = h->phase_acc[3] = 0; h->phase_cnt = 0;
    const int rc = ba_loop(h, n_iter, true);
    fill_summary(h, summary, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    return rc;
}

int sfmhip_ba_get_params(sfmhip_ba* h, double* K4, double* ext6, double* pts)
{
    if (!h) return SFMHIP_E_ARG;
    sfmhip_ctx* ctx = h->ctx;
    SFM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (K4) SFM_HIP_TRY(ctx, hipMemcpy(K4, h->d_K, 4 * sizeof(double), hipMemcpyDeviceToHost));
    if (ext6) SFM_HIP_TRY(ctx, hipMemcpy(ext6, h->d_ext, 6 * (size_t)h->nc * sizeof(double), hipMemcpyDeviceToHost));
    if (pts && h->np) SFM_HIP_TRY(ctx, hipMemcpy(pts, h->d_pts, 3 * (size_t)h->np * sizeof(double), hipMemcpyDeviceToHost));
    return SFMHIP_OK;
}

int sfmhip_ba_reduced_system(sfmhip_ba* h, double radius, double* S, double* rhs, int* n, double* cost)
{
    if (!h) return SFMHIP_E_ARG;
    sfmhip_ctx* ctx = h->ctx;
    if (n) *n = h->n;
    if (!S && !rhs) return SFMHIP_OK;
    if (!h->started) { int rc = ba_start(h); if (rc) return rc; }
    if (radius == 0.0) return SFMHIP_E_ARG;
    int rc = enqueue_linearize(h, std::fabs(radius), radius > 0.0); if (rc) return rc;
    SFM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const size_t np2 = (size_t)h->npad * h->npad;
    if (S && h->n) SFM_HIP_TRY(ctx, hipMemcpy2D(S, (size_t)h->n * sizeof(double), h->d_msg, (size_t)h->npad * sizeof(double),
                                                 (size_t)h->n * sizeof(double), h->n, hipMemcpyDeviceToHost));
    if (rhs && h->n) SFM_HIP_TRY(ctx, hipMemcpy(rhs, h->d_msg + np2, (size_t)h->n * sizeof(double), hipMemcpyDeviceToHost));
    if (cost) SFM_HIP_TRY(ctx, hipMemcpy(cost, h->d_msg + np2 + 3 * (size_t)h->npad, sizeof(double), hipMemcpyDeviceToHost));
    return SFMHIP_OK;
}

int sfmhip_ba_phase_ms(sfmhip_ba* h, double out_ms[4])
{
    if (!h || !out_ms) return SFMHIP_E_ARG;
    for (int i = 0; i < 4; ++i) out_ms[i] = h->phase_cnt ? h->phase_acc[i] / h->phase_cnt : 0.0;
    return SFMHIP_OK;
}

int sfmhip_ba_solve(sfmhip_ctx* ctx, double* K4, double* ext6, int n_cam, double* pts, int n_pt,
                    const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_uv, int n_obs,
                    const sfm_ba_options* opts, sfm_ba_summary* summary)
{
    sfmhip_ba* h = nullptr;
    int rc = sfmhip_ba_create(ctx, K4, ext6, n_cam, pts, n_pt, obs_cam, obs_pt, obs_uv, n_obs, opts, &h);
    if (rc) return rc;
    rc = sfmhip_ba_run(h, summary);
    if (rc == SFMHIP_OK) rc = sfmhip_ba_get_params(h, K4, ext6, pts);
    sfmhip_ba_destroy(h);
    return rc;
}

}  // extern "C"
