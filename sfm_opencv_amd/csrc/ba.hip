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
#include "ba_chain.hpp"
#include "ba_tiles.hpp"
#include "ba_setup.hpp"
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
    // L^-1 came out of the same pivot loop, one column per row of s.W[32..63]
    if (lane < 32) {
        double* out = Linv + (size_t)k * NB * NB;
#pragma unroll
        for (int i = 0; i < NB; ++i) out[i * NB + row] = s.W[NB + row][i];
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

// Multi-rank message: only the 32x32 blocks of the lower triangle that the Schur complement can populate (camera
// adjacency + intrinsics row, diagonal blocks whole) plus the tail [rhs | diagU | graw | scalars] travel through the
// all-reduce hook -- at C4 1.4 MB instead of the 12 MB dense square.  dir 0: S -> message, 1: message -> S.
// carry (multi-rank, folded step scalars): the five scalars of the step just taken [model cost change, candidate cost, |dp|^2,
// |x_p|^2, error flag] ride in the free slots scal[2..6] of the message of the NEXT linearisation ("fuse the
// candidate-cost scalar into the next iteration's message"); unpacking hands their sums back.  carry_off = offset of scal[2] in the tail.
__global__ __launch_bounds__(256) void ba_pack_kernel(double* __restrict__ S, int ld, const int* __restrict__ sblk, int n_sblk,
                                                      double* __restrict__ tail, size_t tail_count, double* __restrict__ msg, int dir,
                                                      double* __restrict__ carry, size_t carry_off)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b < n_sblk) {
        const int bi = sblk[2 * b], bj = sblk[2 * b + 1];
        for (int e = tid; e < NB * NB; e += 256) {
            double* g = S + (size_t)(bi * NB + (e >> 5)) * ld + bj * NB + (e & 31);
            double* m = msg + (size_t)b * NB * NB + e;
            if (dir == 0) *m = *g; else *g = *m;
        }
    } else {
        const size_t base = (size_t)n_sblk * NB * NB;
        for (size_t e = (size_t)(b - n_sblk) * 256 + tid; e < tail_count; e += (size_t)(gridDim.x - n_sblk) * 256) {
            if (carry && e >= carry_off && e < carry_off + 5) {
                if (dir == 0) msg[base + e] = carry[e - carry_off]; else carry[e - carry_off] = msg[base + e];
            } else if (dir == 0) msg[base + e] = tail[e]; else tail[e] = msg[base + e];
        }
    }
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
    double *d_campre = nullptr, *d_campre_c = nullptr;     // per-camera rotation blocks (CAMPRE doubles), current / candidate
    // structure
    int *d_pt_start = nullptr, *d_ocam = nullptr, *d_opt = nullptr, *d_cam_start = nullptr, *d_cam_pt = nullptr, *d_blk_crange = nullptr;
    int *d_blk_cam = nullptr, *d_blk_chunk = nullptr; int4 *d_items = nullptr, *d_chunk_desc = nullptr; int nchunk = 0; double* d_part_schur = nullptr;
    int *d_prow_start = nullptr, *d_prow = nullptr; bool use_sparse = false; int max_panel_rows = 0;
    std::vector<int> host_blk_cam;
    // layout of the reduced system (nested-dissection ordering of the camera chain, segments padded to 32-blocks)
    int npad_max = 0, nseg = 1, top_blk = 0, solver_pmax = 1; long long nnz_blocks = 0;      // nseg: leaves of the dissection (1: none)
    std::vector<int> cam_pos, pos_param;      // camera -> first position (-1 constant); position -> natural index (-1 pad)
    int* d_slot = nullptr;                    // caller's point index -> slot in the HBM arrays (points sorted by camera set)
    int *d_cam_pos = nullptr, *d_posmask = nullptr;
    char* arena = nullptr; size_t arena_left = 0, arena_chunk = 0;       // device arrays are carved out of a few large blocks
    double setup_ms[4] = { 0, 0, 0, 0 };      // sfmhip_ba_create, cumulative host clock: [inputs + observation sort, + orderings, + pair lists, whole call]
    double start_ms = 0;                      // ba_start: solver plan + scaling (the rest of Ceres' "preprocessor")
    double* d_topbuf = nullptr; size_t topbuf_count = 0, topbuf_cap = 0;      // the nodes' private update buffers (zero on entry to the solve)
    // elimination tree of the dissection: nodes in elimination order, level l = nodes [lvl_first[l], lvl_first[l + 1]), then the top node
    NodeDesc* d_nodes = nullptr; size_t nodes_cap = 0; FoldEnt* d_ents = nullptr; size_t ents_cap = 0;
    std::vector<int> lvl_first; int top_node = 0;
    int* d_sblk = nullptr; int n_sblk = 0; size_t sblk_cap = 0; double* d_pack = nullptr; size_t pack_cap = 0;     // packed all-reduce message
    double *d_ouv = nullptr, *d_cam_uv = nullptr;
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
    hipEvent_t ev[10] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
    hipStream_t aux = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;      // second stream: the Schur pair kernel runs beside the camera kernel
    hipEvent_t evb[2][5] = {};          // per build parity: [start, camera kernel begin/end, pair kernel begin/end]
    hipEvent_t evi[2][5] = {};          // per iteration parity: [damped, solved, back-substituted, forward kernel begin/end]
    int iter_parity = 0, pending_build = -1, pending_iter = -1;       // timings not yet read (read off the decision path)
    bool cleared = false;               // S (the npad x npad part of d_msg) is already zero for the next build -- ONLY S: the tail [rhs | diagU | graw | scalars]
                                        // and d_err are not refilled, they rely on the finalisation storing every real entry (padding entries stay zero
                                        // through the solve) and on ba_back_reduce_kernel re-arming the error flag; tests/test_ba_gpu.py::
                                        // test_reused_message_tail_with_an_unobserved_camera holds both linearisers to that
    bool top_cleared = false;      // d_topbuf was zero-filled ahead of time (behind the publish kernel, while the host decides)
    int n_diag_blk = 0;            // camera pairs (a, a): a point seen twice by one camera
    bool solver_damps = false; double damp_radius = 0.0;    // the next enqueue_solve applies the LM damping inside its kernels
    bool build_timed = false;      // the pending build recorded its events (timing can be switched between launches)
    bool build_fused[2] = { false, false };      // per build parity: camera items and pair chunks shared one launch (one kernel time, reported in slot 4)
    bool publish_in_back = false, published = false;   // ba_loop asks enqueue_back to publish the step scalars from its reduction kernel
    bool campre_valid = false;     // d_campre matches d_ext (kept across iterations: an accepted step swaps in the candidate's)
    unsigned long long pub_seq = 0; // sequence number of the last ba_publish_kernel
    bool fold_step_scalars = false; // set by ba_loop on several ranks: enqueue_back leaves the step scalars to the next message
    long long ar_calls = 0;         // all-reduce hook invocations since sfmhip_ba_set_allreduce (tests assert one per LM iteration)
    bool force_dense = false;      // SFMHIP_EXPERIMENTS builds: SFMHIP_DENSE_SOLVER routes every problem to the dense fallback
    long long* d_stamps = nullptr; int stamp_calls = 0;      // SFMHIP_EXPERIMENTS builds only: per-panel cycle stamps of the solver
    // run-tile linearisation (ba_tiles.hpp): segments of point runs, their tiles, and the fold table of ba_tile_reduce_kernel
    int fuse_max_blocks = 4096;        // camera + pair workgroups up to which they share one launch
    bool use_tiles = false; int n_tseg = 0; long long n_ttiles = 0;
    std::vector<TileSeg> tsegs; std::vector<int> tcams;
    TileSeg* d_tsegs = nullptr; int* d_tcams = nullptr; double *d_tpart = nullptr, *d_tpart_seg = nullptr;
    int *d_rd_start = nullptr, *d_rd_dst = nullptr, *d_rd_dst2 = nullptr; unsigned* d_rd_src = nullptr; int rd_nd = 0, rd_n_long = 0;
    size_t rd_dst_cap = 0, rd_src_cap = 0;
    std::vector<int> tile_tab_cam_pos; int tile_tab_npad = -1;      // the layout the fold table was built for
    // chain solver (ba_chain.hpp): plan, factor records, the sub-trees' exported fronts
    bool use_chain = false; ChainArgs chain; double *d_chain_rec = nullptr, *d_chain_img = nullptr; size_t chain_rec_cap = 0, chain_img_cap = 0;
    size_t chain_lds1 = 0, chain_lds2 = 0;
    bool built = false; int build_parity = 0;    // d_msg holds the undamped linearisation at the CURRENT parameters (set by a speculative build)
    double phase_acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }; int phase_cnt = 0;
};

static inline double ms_since(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

template <typename T>
static int dalloc(sfmhip_ba* h, T** p, size_t count)
{
    const size_t bytes = (((count > 0 ? count : 1) * sizeof(T)) + 255) & ~(size_t)255;
    if (bytes > h->arena_left) {
        const size_t chunk = std::max(h->arena_chunk, (size_t)1 << 20);
        void* q = nullptr;
        if (bytes >= chunk / 4) {           // large arrays get their own block, the arena keeps its room for the small ones
            int rc = sfm_pool_get(h->ctx, bytes, &q); if (rc) return rc;
            h->allocs.push_back(q);
            *p = (T*)q;
            return SFMHIP_OK;
        }
        int rc = sfm_pool_get(h->ctx, chunk, &q); if (rc) return rc;
        h->allocs.push_back(q);
        h->arena = (char*)q; h->arena_left = chunk;
    }
    *p = (T*)h->arena;
    h->arena += bytes; h->arena_left -= bytes;
    return SFMHIP_OK;
}
template <typename T>
static int dupload(sfmhip_ba* h, T** p, const T* src, size_t count)
{
    int rc = dalloc(h, p, count); if (rc) return rc;
    // through the pinned staging ring: the source (often a function-local vector) has been consumed when this returns -- a
    // hipMemcpyAsync from pageable memory does not promise that
    return count ? sfm_upload(h->ctx, *p, src, count * sizeof(T)) : SFMHIP_OK;
}

static BADev make_dev(const sfmhip_ba* h, double radius, bool at_candidate = false)
{
    BADev P;
    memset(&P, 0, sizeof P);
    P.nc = h->nc; P.np = h->np; P.nobs = h->nobs; P.n = h->n; P.npad = h->npad; P.koff = h->koff;
    P.fix0 = h->fix0; P.fixK = h->fixK; P.cam_split = h->cam_split; P.world = h->world; P.rank = h->rank;
    P.huber_a = h->o.huber_delta;
    P.K = h->d_K; P.ext = h->d_ext; P.pts = h->d_pts; P.Kc = h->d_Kc; P.extc = h->d_extc; P.ptsc = h->d_ptsc;
    P.pt_start = h->d_pt_start; P.ocam = h->d_ocam; P.ouv = h->d_ouv;
    P.cam_start = h->d_cam_start; P.cam_pt = h->d_cam_pt; P.cam_uv = h->d_cam_uv; P.opt = h->d_opt; P.blk_crange = h->d_blk_crange;
    P.cam_pos = h->d_cam_pos; P.posmask = h->d_posmask;
    P.campre = h->d_campre; P.campre_c = h->d_campre_c;
    P.scale_c = h->d_scale_c; P.scale_p = h->d_scale_p;
    P.Vinv = h->d_Vinv; P.bp = h->d_bp; P.WK = h->d_WK; P.colsq_p = h->d_colsq_p;
    const size_t np2 = (size_t)h->npad * h->npad;
    P.S = h->d_msg; P.rhs = h->d_msg + np2; P.diagU = P.rhs + h->npad; P.graw = P.diagU + h->npad; P.scal = P.graw + h->npad;
    P.part_pt = h->d_part_pt; P.part_cam = h->d_part_cam; P.part_back = h->d_part_back;
    P.y = h->d_y;
    P.radius = radius; P.min_diag = h->o.min_lm_diagonal; P.max_diag = h->o.max_lm_diagonal;
#ifdef SFMHIP_EXPERIMENTS
    { static const int plain = getenv("SFMHIP_EXP_XCD_PLAIN") ? 1 : 0; P.xcd_plain = plain; }      // measurement knob (profiles/r01_traffic_pmc.md)
#endif
    if (at_candidate) {
        P.K = h->d_Kc; P.ext = h->d_extc; P.pts = h->d_ptsc; P.Kc = h->d_K; P.extc = h->d_ext; P.ptsc = h->d_pts;
        P.campre = h->d_campre_c; P.campre_c = h->d_campre;
    }
    return P;
}

static int call_allreduce(sfmhip_ba* h, double* buf, size_t count)
{
    if (!h->ar_fn) return SFMHIP_OK;
    ++h->ar_calls;
    const int rc = h->ar_fn(h->ar_user, buf, count, (void*)h->ctx->stream);
    if (rc != 0) { h->ctx->last_error = "all-reduce hook failed"; return SFMHIP_E_COMM; }
    return SFMHIP_OK;
}

static int enqueue_build_exchange(sfmhip_ba* h, double* carry);

// linearise: message = [S | rhs | diagU | graw | scal] (undamped), summed over ranks.  at_candidate: at the candidate
// parameters the last back-substitution produced (speculative build of the next iteration, see ba_loop).
static int enqueue_build(sfmhip_ba* h, double radius, bool at_candidate, bool timed, double* carry = nullptr)
{
    sfmhip_ctx* ctx = h->ctx;
    hipStream_t st = ctx->stream;
    BADev P = make_dev(h, radius, at_candidate);       // the point blocks are damped with the radius while they are built
    hipEvent_t* tv = nullptr;
    // phase / kernel events only while sfmhip_set_kernel_timing is on: thirteen event records cost ~27 us per iteration
    if (timed) h->build_timed = ctx->timing;
    if (timed && ctx->timing) { h->build_parity ^= 1; tv = h->evb[h->build_parity]; (void)hipEventRecord(tv[0], st); }
    if (!h->cleared) {
        SFM_HIP_TRY(ctx, hipMemsetAsync(h->d_msg, 0, h->msg_count * sizeof(double), st));
        SFM_HIP_TRY(ctx, hipMemsetAsync(h->d_err, 0, sizeof(int), st));
    }
    h->cleared = false;
    // rotation blocks: the candidate's are computed with the step (ba_camstep_kernel) and swapped in when it is accepted
    if (!at_candidate && !h->campre_valid) {
        hipLaunchKernelGGL(ba_campre_kernel, dim3(ceil_div(h->nc, 64)), dim3(64), 0, st, h->d_ext, h->nc, h->d_campre);
        h->campre_valid = true;
    }
    if (h->use_tiles) {
        if (tv) { (void)hipEventRecord(tv[3], st); (void)hipEventRecord(tv[4], st); (void)hipEventRecord(tv[1], st); }
        hipLaunchKernelGGL(ba_tile_kernel, dim3(h->n_tseg), dim3(256), TILE_LDS_BYTES, st, P, h->d_tsegs, h->d_tcams, h->d_tpart, h->d_tpart_seg, h->d_err);
        if (tv) (void)hipEventRecord(tv[2], st);
        hipLaunchKernelGGL(ba_tile_reduce_kernel, dim3(h->rd_n_long + ceil_div(h->rd_nd - h->rd_n_long, 256) + 1), dim3(256), 0, st, P, h->d_rd_start, h->d_rd_dst, h->d_rd_dst2, h->d_rd_src,
                           h->rd_n_long, h->rd_nd, h->d_tpart, h->d_tpart_seg, h->n_tseg);
        SFM_HIP_TRY(ctx, hipGetLastError());
        return enqueue_build_exchange(h, carry);
    }
    hipLaunchKernelGGL(ba_point_kernel, dim3(h->n_pt_blocks), dim3(256), 0, st, P, h->d_err);
    // camera items and pair chunks in one launch, then their folds in one launch (ba_kernels.hpp: ba_camschur_kernel)
    const int n_cam_blocks = round_up(h->nc * h->cam_split * (h->fixK ? 1 : 2), 8);
    const int n_schur_blocks = h->nblk > 0 ? round_up(ceil_div(h->nchunk, 4), 8) : 0;
    // Up to a few thousand workgroups one launch is ahead (C3: 0.190 -> 0.169 ms per iteration, C4: 0.314 -> 0.306: the fork /
    // join events cost 5-7 us of stream gap each); beyond that the two kernels on two queues run 15 % faster than the shared
    // launch (C5: 0.44 || 0.48 ms against 0.56), so large problems keep the auxiliary stream.
    bool folded = false;
    if (n_schur_blocks > 0 && n_cam_blocks + n_schur_blocks > h->fuse_max_blocks) {
        SFM_HIP_TRY(ctx, hipEventRecord(h->ev_fork, st));
        SFM_HIP_TRY(ctx, hipStreamWaitEvent(h->aux, h->ev_fork, 0));
        if (tv) { (void)hipEventRecord(tv[3], h->aux); h->build_fused[h->build_parity] = false; }
        hipLaunchKernelGGL(ba_schur_kernel, dim3(n_schur_blocks), dim3(256), 0, h->aux, P, h->d_chunk_desc, h->nchunk, h->d_items, h->d_part_schur);
        if (tv) (void)hipEventRecord(tv[4], h->aux);
        hipLaunchKernelGGL(ba_schur_reduce_kernel, dim3(ceil_div(h->nblk * 36, 256)), dim3(256), 0, h->aux, P, h->d_blk_cam, h->d_blk_chunk, h->nblk, h->d_part_schur, 0);
        SFM_HIP_TRY(ctx, hipEventRecord(h->ev_join, h->aux));
        if (tv) (void)hipEventRecord(tv[1], st);
        hipLaunchKernelGGL(ba_camera_kernel, dim3(n_cam_blocks), dim3(256), 0, st, P);
        if (tv) (void)hipEventRecord(tv[2], st);
        hipLaunchKernelGGL(ba_fold_kernel, dim3(h->nc + 1), dim3(256), 0, st, P, h->n_pt_blocks, h->d_blk_cam, h->d_blk_chunk, 0, h->d_part_schur);      // each stream folds its own kernel's partials
        SFM_HIP_TRY(ctx, hipStreamWaitEvent(st, h->ev_join, 0));
        folded = true;
    } else {
        if (tv) { (void)hipEventRecord(tv[1], st); (void)hipEventRecord(tv[3], st); h->build_fused[h->build_parity] = true; }
        hipLaunchKernelGGL(ba_camschur_kernel, dim3(n_cam_blocks + n_schur_blocks), dim3(256), 0, st, P, n_cam_blocks, h->d_chunk_desc, h->nchunk, h->d_items, h->d_part_schur);
        if (tv) { (void)hipEventRecord(tv[2], st); (void)hipEventRecord(tv[4], st); }
    }
    if (!folded)
        hipLaunchKernelGGL(ba_fold_kernel, dim3(h->nc + 1 + (h->nblk > 0 ? ceil_div(h->nblk * 36, 256) : 0)), dim3(256), 0, st, P, h->n_pt_blocks,
                           h->d_blk_cam, h->d_blk_chunk, h->nblk, h->d_part_schur);
    if (h->n_diag_blk > 0)
        hipLaunchKernelGGL(ba_schur_reduce_kernel, dim3(ceil_div(h->nblk * 36, 256)), dim3(256), 0, st, P, h->d_blk_cam, h->d_blk_chunk, h->nblk, h->d_part_schur, 1);
    SFM_HIP_TRY(ctx, hipGetLastError());
    return enqueue_build_exchange(h, carry);
}

// multi-rank: the packed sum of the reduced-system message over the ranks
static int enqueue_build_exchange(sfmhip_ba* h, double* carry)
{
    sfmhip_ctx* ctx = h->ctx;
    hipStream_t st = ctx->stream;
    if (h->ar_fn) {
        const size_t np2 = (size_t)h->npad * h->npad, tail = h->msg_count - np2, count = (size_t)h->n_sblk * NB * NB + tail;
        const int tail_blocks = (int)std::min<size_t>((tail + 255) / 256, 64);
        const size_t carry_off = 3 * (size_t)h->npad + 2;          // scal[2] inside the tail [rhs | diagU | graw | scal]
        hipLaunchKernelGGL(ba_pack_kernel, dim3(h->n_sblk + tail_blocks), dim3(256), 0, st, h->d_msg, h->npad, h->d_sblk, h->n_sblk,
                           h->d_msg + np2, tail, h->d_pack, 0, carry, carry_off);
        SFM_HIP_TRY(ctx, hipGetLastError());
        int rc = call_allreduce(h, h->d_pack, count); if (rc) return rc;
        hipLaunchKernelGGL(ba_pack_kernel, dim3(h->n_sblk + tail_blocks), dim3(256), 0, st, h->d_msg, h->npad, h->d_sblk, h->n_sblk,
                           h->d_msg + np2, tail, h->d_pack, 1, carry, carry_off);
        SFM_HIP_TRY(ctx, hipGetLastError());
    }
    return SFMHIP_OK;
}

static int enqueue_damp(sfmhip_ba* h, double radius)
{
    BADev P = make_dev(h, radius);
    hipLaunchKernelGGL(ba_damp_kernel, dim3(1), dim3(256), 0, h->ctx->stream, P);
    SFM_HIP_TRY(h->ctx, hipGetLastError());
    return SFMHIP_OK;
}

// linearise at the current parameters (+ damping): start-up and sfmhip_ba_reduced_system
static int enqueue_linearize(sfmhip_ba* h, double radius, bool damp)
{
    h->built = false;
    int rc = enqueue_build(h, radius, false, false); if (rc) return rc;
    return damp ? enqueue_damp(h, radius) : SFMHIP_OK;
}

static int enqueue_solve(sfmhip_ba* h)
{
    sfmhip_ctx* ctx = h->ctx;
    hipStream_t st = ctx->stream;
    double* S = h->d_msg;
    const int nb = h->nbk, ld = h->npad;
    if (h->use_chain) {
        // chain solver: reads S, rhs and diagU (never writes them), factor records and exported fronts in its own buffers
        ChainArgs A = h->chain;
        A.S = S; A.rhs = h->d_msg + (size_t)h->npad * h->npad; A.diagU = nullptr; A.inv_radius = 1.0; A.dmin = A.dmax = 0.0;
        if (h->solver_damps) {
            BADev P = make_dev(h, h->damp_radius);
            A.diagU = P.diagU; A.inv_radius = 1.0 / h->damp_radius; A.dmin = P.min_diag; A.dmax = P.max_diag;
            h->solver_damps = false;
        }
        A.rec = h->d_chain_rec; A.img = h->d_chain_img; A.y = h->d_y; A.err = h->d_err; A.stamps = nullptr;
        hipLaunchKernelGGL(chain_sub_kernel, dim3(A.P >> A.a), dim3(64 * A.G << A.a), h->chain_lds1, st, A);
        if (A.a < A.m) hipLaunchKernelGGL(chain_top_kernel, dim3(1), dim3(64 * A.nw_top), h->chain_lds2, st, A);
        SFM_HIP_TRY(ctx, hipGetLastError());
        return SFMHIP_OK;
    }
    if (h->use_sparse) {
        double* rhs_rw = h->d_msg + (size_t)h->npad * h->npad;
        SolverPlan pl; pl.prow_start = h->d_prow_start; pl.prow = h->d_prow; pl.nb = nb; pl.top_blk = h->top_blk; pl.linv = h->d_Linv;
        pl.damp_diagU = nullptr; pl.damp_mask = nullptr; pl.damp_radius = 1.0; pl.damp_min = pl.damp_max = 0.0;
        if (h->solver_damps) {
            BADev P = make_dev(h, h->damp_radius);
            pl.damp_diagU = P.diagU; pl.damp_mask = P.posmask; pl.damp_radius = h->damp_radius; pl.damp_min = P.min_diag; pl.damp_max = P.max_diag;
            h->solver_damps = false;
        }
        pl.dbg = 0; pl.stamps = nullptr;
#ifdef SFMHIP_EXPERIMENTS
        // timing experiments (results are garbage with dbg != 0) and per-panel cycle stamps; read once per process
        static const int exp_dbg = [] { const char* e = getenv("SFMHIP_EXP_SOLVER"); return e ? atoi(e) : 0; }();
        static const bool exp_stamps = getenv("SFMHIP_SOLVER_STAMPS") != nullptr;
        pl.dbg = exp_dbg;
        if (exp_stamps) {
            if (!h->d_stamps) { int rc = dalloc(h, &h->d_stamps, 16 * 512); if (rc) return rc; (void)hipMemset(h->d_stamps, 0, 16 * 512 * sizeof(long long)); }
            pl.stamps = h->d_stamps;
            if (++h->stamp_calls == 8) {          // dump once, after a few warm iterations
                (void)hipStreamSynchronize(st);
                std::vector<long long> hs(16 * (size_t)nb);
                (void)hipMemcpy(hs.data(), h->d_stamps, hs.size() * sizeof(long long), hipMemcpyDeviceToHost);
                for (int k = 0; k < nb; ++k) {
                    fprintf(stderr, "[stamps] panel %3d: stage %6lld | solve %6lld | next pivot %6lld | factor (wave 0) %6lld | trailing tiles %6lld + rhs %6lld (wave 1) | join %6lld   cycles\n", k,
                            hs[16 * k + 1] - hs[16 * k + 0], hs[16 * k + 4] - hs[16 * k + 1], hs[16 * k + 5] - hs[16 * k + 4],
                            hs[16 * k + 2] - hs[16 * k + 5], hs[16 * k + 7] - hs[16 * k + 5], hs[16 * k + 3] - hs[16 * k + 7], hs[16 * k + 6] - hs[16 * k + 2]);
                    if (hs[16 * k + 8]) fprintf(stderr, "[stamps]   node starting at panel %d: descriptor %lld | assembly / damping %lld | panels %lld | total %lld cycles\n", k,
                                                hs[16 * k + 9] - hs[16 * k + 8], hs[16 * k + 10] - hs[16 * k + 9], hs[16 * k + 11] - hs[16 * k + 10], hs[16 * k + 11] - hs[16 * k + 8]);
                    if (k + 1 < nb) fprintf(stderr, "[stamps]            to the next panel's start: %lld\n", hs[16 * (k + 1)] - hs[16 * k + 6]);
                }
            }
        }
#endif
        if (h->nseg <= 1) {
            pl.top_blk = nb;
            hipLaunchKernelGGL(chol_sparse_kernel, dim3(1), dim3(STHREADS), 0, st, S, ld, pl, rhs_rw, h->d_y, h->d_err);
        } else {
            if (!h->top_cleared) SFM_HIP_TRY(ctx, hipMemsetAsync(h->d_topbuf, 0, h->topbuf_count * sizeof(double), st));
            h->top_cleared = false;
            const int nlev = (int)h->lvl_first.size() - 1;
            for (int l = 0; l < nlev; ++l) {        // one launch per level, a workgroup per node
                if (l == 0 && ctx->timing) (void)hipEventRecord(h->evi[h->iter_parity][3], st);
                hipLaunchKernelGGL(chol_node_forward_kernel, dim3(h->lvl_first[l + 1] - h->lvl_first[l]), dim3(STHREADS), 0, st, S, ld, pl, h->d_nodes, h->lvl_first[l],
                                   rhs_rw, h->d_topbuf, h->d_ents, h->d_err);
                if (l == 0 && ctx->timing) (void)hipEventRecord(h->evi[h->iter_parity][4], st);
            }
            hipLaunchKernelGGL(chol_top_kernel, dim3(1), dim3(STHREADS), 0, st, S, ld, pl, h->d_nodes, h->top_node, rhs_rw, h->d_ents, h->d_y, h->d_err);
            for (int l = nlev - 1; l >= 0; --l)
                hipLaunchKernelGGL(chol_node_backward_kernel, dim3(h->lvl_first[l + 1] - h->lvl_first[l]), dim3(STHREADS), 0, st, S, ld, pl, h->d_nodes, h->lvl_first[l], rhs_rw, h->d_y);
        }
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
    hipLaunchKernelGGL(ba_camstep_kernel, dim3(1 + ceil_div(h->nc, 256)), dim3(256), 0, st, P, h->d_cam2);     // block 0: the step; the others: candidate rotation blocks
    // single rank: the reduction also publishes the decision scalars (ba_loop then skips ba_publish_kernel), and extra
    // workgroups of the back-substitution zero-fill S and the solver's private buffers for the next linearisation.  (The tail
    // [rhs | diagU | graw | scalars] needs no refill there: the finalisation stores every real entry and the padding
    // entries stay zero through the solve; on several ranks the other ranks' slots must be cleared, so the build memsets.)
    const bool fuse_publish = !h->ar_fn && h->publish_in_back;
    size_t n0 = 0, n1 = 0;
    if (fuse_publish) {
        // (the chain solver leaves S as the build wrote it, and the next build stores every entry it can populate: nothing to clear)
        n0 = h->use_chain ? 0 : (size_t)h->npad * h->npad;
        if (h->use_sparse && h->nseg > 1 && h->d_topbuf) { n1 = h->topbuf_count & ~(size_t)1; h->top_cleared = (n1 == h->topbuf_count); }
        h->cleared = true;
    }
    hipLaunchKernelGGL(ba_back_kernel, dim3(h->n_pt_blocks + (n0 + n1 ? BACK_ZERO_BLOCKS : 0)), dim3(256), 0, st, P, h->n_pt_blocks, h->d_msg, n0, h->d_topbuf, n1);
    const double* d_scal = h->d_msg + (size_t)h->npad * h->npad + 3 * (size_t)h->npad;
    hipLaunchKernelGGL(ba_back_reduce_kernel, dim3(1), dim3(256), 0, st, h->d_part_back, h->n_pt_blocks, h->d_back4,
                       d_scal, h->d_cam2, h->d_err, fuse_publish ? h->h_scal : (double*)nullptr, fuse_publish ? ++h->pub_seq : 0ull);
    h->published = fuse_publish;
    SFM_HIP_TRY(ctx, hipGetLastError());
    // five doubles: the four step scalars and this rank's error flag (a non-SPD V of a local point): every rank must take
    // the same accept / invalid branch, or the replicated cameras, radius and nu diverge and the next all-reduce hangs.
    // Folded (ba_loop on several ranks): they travel with the message of the speculative linearisation that follows instead.
    if (h->fold_step_scalars) return SFMHIP_OK;
    return call_allreduce(h, h->d_back4, 5);
}

// Layout + block fill pattern of the reduced system -> per-panel row lists and the elimination tree for the solver kernels.
//  1. camera adjacency (cameras sharing a point); multi-rank: union over ranks through the all-reduce hook.
//  2. ordering: if the camera graph is a narrow band (the reference's tracks only chain through consecutive frames),
//     multi-level nested dissection of the chain (ba_solver.hpp): P = 2^m leaf segments separated by `w` cameras,
//     Lp levels of mutually independent separators, then a serially factored top (remaining separators + intrinsics).
//     Every parallel node is padded to whole 32-blocks.
//  3. symbolic block factorisation in that order -> rows(k); every node's panels may only reach its own range and blocks
//     of LATER levels (at most SAMAX of them, SRMAX per panel), else the next simpler configuration is tried, down to a
//     single node (chol_sparse_kernel) and finally the dense blocked fallback.
static int solver_pick_leaves(int ncf, int w)
{
    // leaves of at least max(3 w, 16) cameras (three panels' worth: every level costs a launch and a handful of dependent
    // round trips to buffers other workgroups wrote, ~25 us at C4 -- measured, profiles/README.md), at most 64 of them
    int P = 1;
    while (P < 64 && w >= 1 && (ncf - (2 * P - 1) * w) / (2 * P) >= std::max(3 * w, 16)) P *= 2;
    return P;
}

static int build_solver_plan(sfmhip_ba* h)
{
    sfmhip_ctx* ctx = h->ctx;
    const int nc = h->nc, ncf = h->ncf, f0 = h->fix0;
    // ---- 1. adjacency among free cameras (index i = c - fix0)
    std::vector<double> adj((size_t)ncf * ncf, 0.0);
    for (size_t b = 0; b + 1 < h->host_blk_cam.size(); b += 2) {
        const int a = h->host_blk_cam[b] - f0, c = h->host_blk_cam[b + 1] - f0;
        adj[(size_t)a * ncf + c] = 1.0; adj[(size_t)c * ncf + a] = 1.0;
    }
    if (h->ar_fn && ncf > 0) {
        if (adj.size() > h->msg_count) { ctx->last_error = "adjacency larger than the message buffer"; return SFMHIP_E_ARG; }
        SFM_HIP_TRY(ctx, hipMemcpyAsync(h->d_msg, adj.data(), adj.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        int rc = call_allreduce(h, h->d_msg, adj.size()); if (rc) return rc;
        SFM_HIP_TRY(ctx, hipMemcpyAsync(adj.data(), h->d_msg, adj.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        SFM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    int w = 0;
    for (int a = 0; a < ncf; ++a) for (int c = 0; c < a; ++c) if (adj[(size_t)a * ncf + c] != 0.0) w = std::max(w, a - c);

    // The chain solver (ba_chain.hpp) takes the natural layout (cameras in chain order, no padding inside).  It applies to bands of
    // at most CH_WMAX = 3 cameras (tracks of up to four frames): its fronts then have <= 59 rows, one lane each.  Measured in the LM
    // loop on such scenes (experiments/chain_vs_levels.py, MI355X): reduced solve 0.094 against 0.128 ms at 200 cameras, 0.071 against
    // 0.084 ms at 50, 0.044 against 0.047 ms at 24, 0.025 against 0.022 ms at 8 -- hence from 12 free cameras on.  Beyond ~640 cameras
    // its leaves get long (32 leaves at most, two waves each) and the level-per-launch solver, whose leaves are throughput-bound
    // panels, is level with it again (999 cameras: 242 us against 238 us).  SURVEY 8d's scenes (tracks of 2..6 frames: band 5) stay
    // with the solver below: at that width a separator is one 32-column MFMA panel there and five serial camera steps here.
    h->use_chain = false;
    ChainArgs chain_args;
    memset(&chain_args, 0, sizeof chain_args);
    bool chain_ok = false;
    if (h->o.solver == 0 && !h->force_dense && ncf >= 12 && ncf <= 640) {
        const int npad_nat = std::max(NB, round_up(6 * ncf + (h->fixK ? 0 : 4), NB));
        chain_ok = npad_nat <= h->npad_max && chain_plan(chain_args, ncf, w, h->fixK ? 0 : 4, npad_nat, npad_nat, 0, -1, 0, (size_t)160 << 10);
    }
    // candidate configurations (leaves, parallel separator levels), best first
    struct Cfg { int P, Lp; };
    std::vector<Cfg> cfgs;
    if (!chain_ok) {
        int P0 = std::min(solver_pick_leaves(ncf, w), h->solver_pmax), Lp0 = -1;
#ifdef SFMHIP_EXPERIMENTS
        if (const char* e = getenv("SFMHIP_ND_LEAVES")) { int v = std::max(1, atoi(e)); P0 = 1; while (P0 * 2 <= v && P0 * 2 <= h->solver_pmax) P0 *= 2; }
        if (const char* e = getenv("SFMHIP_ND_LEVELS")) Lp0 = std::max(0, atoi(e));
#endif
        for (int P = P0; P >= 2; P /= 2) {
            int m = 0; while ((1 << m) < P) ++m;
            // by default every separator level but the last two is parallel (the top keeps <= 3 separators + the intrinsics)
            int Lp = Lp0 >= 0 ? std::min(Lp0, m) : std::max(0, m - 2);
            cfgs.push_back({ P, Lp });
            if (Lp > 0 && Lp0 < 0) cfgs.push_back({ P, 0 });
        }
    }
    cfgs.push_back({ 1, 0 });
    for (const Cfg cfg : cfgs) {
        const int P = cfg.P, Lp = cfg.Lp;
        if (P > 1 && (w < 1 || (ncf - (P - 1) * w) / P < 1)) continue;
        // ---- 2. ordering.  chain = leaf 0 | sep 0 | leaf 1 | sep 1 | ... | leaf P-1; level of sep i = ctz(i + 1) + 1
        std::vector<int> node_of((size_t)std::max(ncf, 1), 0);       // chain camera -> node index (elimination order); -1: top
        std::vector<int> node_level;                                  // per parallel node
        int n_par = 0;
        std::vector<int> first_of_level(1, 0);
        if (P > 1) {
            const int interior = ncf - (P - 1) * w;
            std::vector<int> leaf_lo(P), leaf_hi(P), sep_lo(std::max(P - 1, 1));
            int pos = 0;
            for (int j = 0; j < P; ++j) {
                const int len = interior / P + (j < interior % P ? 1 : 0);
                leaf_lo[j] = pos; pos += len; leaf_hi[j] = pos;
                if (j + 1 < P) { sep_lo[j] = pos; pos += w; }
            }
            for (int j = 0; j < P; ++j) { for (int i = leaf_lo[j]; i < leaf_hi[j]; ++i) node_of[i] = n_par; node_level.push_back(0); ++n_par; }
            first_of_level.push_back(n_par);
            for (int L = 1; L <= Lp; ++L) {
                for (int i = 0; i + 1 < P; ++i)
                    if (__builtin_ctz(i + 1) + 1 == L) { for (int k = 0; k < w; ++k) node_of[sep_lo[i] + k] = n_par; node_level.push_back(L); ++n_par; }
                first_of_level.push_back(n_par);
            }
            // the top: remaining separators by (level, index); their cameras get increasing "top ranks" for the ordering below
            int rank = 0;
            for (int L = Lp + 1; L <= 31; ++L)
                for (int i = 0; i + 1 < P; ++i)
                    if (__builtin_ctz(i + 1) + 1 == L) { for (int k = 0; k < w; ++k) node_of[sep_lo[i] + k] = -1 - (rank++); }
        }
        // positions
        h->cam_pos.assign((size_t)nc, -1);
        std::vector<int> node_k0, node_k1;
        int pos = 0;
        if (P > 1) {
            for (int nd = 0; nd < n_par; ++nd) {
                node_k0.push_back(pos / NB);
                for (int i = 0; i < ncf; ++i) if (node_of[i] == nd) { h->cam_pos[i + f0] = pos; pos += 6; }
                pos = round_up(pos, NB);
                node_k1.push_back(pos / NB);
            }
            const int top_start = pos;
            std::vector<std::pair<int, int>> tops;
            for (int i = 0; i < ncf; ++i) if (node_of[i] < 0) tops.push_back({ -1 - node_of[i], i });
            std::sort(tops.begin(), tops.end());
            for (auto& t : tops) { h->cam_pos[t.second + f0] = pos; pos += 6; }
            h->top_blk = top_start / NB;
        } else {
            for (int i = 0; i < ncf; ++i) { h->cam_pos[i + f0] = pos; pos += 6; }
        }
        h->koff = pos;
        if (!h->fixK) pos += 4;
        const int npad = std::max(NB, round_up(pos, NB));
        if (npad > h->npad_max) { if (P == 1) { ctx->last_error = "internal: npad_max"; return SFMHIP_E_ARG; } continue; }
        const int nb = npad / NB;
        h->npad = npad; h->nbk = nb; h->nseg = P;
        if (P == 1) h->top_blk = nb;
        h->pos_param.assign((size_t)npad, -1);
        for (int c = f0; c < nc; ++c) for (int j = 0; j < 6; ++j) h->pos_param[h->cam_pos[c] + j] = 6 * (c - f0) + j;
        if (!h->fixK) for (int j = 0; j < 4; ++j) h->pos_param[h->koff + j] = 6 * ncf + j;

        // ---- 3. block pattern + symbolic factorisation
        std::vector<char> Pm((size_t)nb * nb, 0);
        auto mark = [&](int lo_a, int lo_b, int len_a, int len_b) {
            for (int i = lo_a / NB; i <= (lo_a + len_a - 1) / NB; ++i)
                for (int j = lo_b / NB; j <= (lo_b + len_b - 1) / NB; ++j) { const int hi = std::max(i, j), lo = std::min(i, j); Pm[(size_t)hi * nb + lo] = 1; }
        };
        for (int i = 0; i < nb; ++i) Pm[(size_t)i * nb + i] = 1;
        for (int c = f0; c < nc; ++c) {
            mark(h->cam_pos[c], h->cam_pos[c], 6, 6);
            if (!h->fixK) mark(h->cam_pos[c], h->koff, 6, 4);
        }
        if (!h->fixK) mark(h->koff, h->koff, 4, 4);
        for (int a = 0; a < ncf; ++a) for (int c = 0; c < a; ++c)
            if (adj[(size_t)a * ncf + c] != 0.0) mark(h->cam_pos[a + f0], h->cam_pos[c + f0], 6, 6);
        std::vector<int> sblk;                      // blocks S can populate (before fill): what a multi-rank all-reduce carries
        for (int i = 0; i < nb; ++i) for (int j = 0; j <= i; ++j) if (Pm[(size_t)i * nb + j]) { sblk.push_back(i); sblk.push_back(j); }
        std::vector<int> blk_node((size_t)nb, n_par);               // block -> node (n_par = the top)
        for (int nd = 0; nd < n_par; ++nd) for (int k = node_k0[nd]; k < node_k1[nd]; ++k) blk_node[k] = nd;
        std::vector<int> start(nb + 1, 0), rows;
        std::vector<std::vector<int>> anc((size_t)n_par);
        int maxR = 0; bool valid = true;
        for (int k = 0; k < nb; ++k) {
            std::vector<int> rk;
            for (int i = k + 1; i < nb; ++i) if (Pm[(size_t)i * nb + k]) rk.push_back(i);
            for (size_t a = 0; a < rk.size(); ++a) for (size_t b = 0; b <= a; ++b) Pm[(size_t)rk[a] * nb + rk[b]] = 1;
            maxR = std::max(maxR, (int)rk.size());
            const int nd = blk_node[k];
            if (nd < n_par)
                for (int i : rk) {
                    const int ni = blk_node[i];
                    if (ni == nd) continue;
                    if (ni < n_par && node_level[ni] <= node_level[nd]) valid = false;      // reaches a sibling: not independent
                    if (std::find(anc[nd].begin(), anc[nd].end(), i) == anc[nd].end()) anc[nd].push_back(i);
                }
            rows.insert(rows.end(), rk.begin(), rk.end());
            start[k + 1] = (int)rows.size();
        }
        for (auto& a : anc) { std::sort(a.begin(), a.end()); if ((int)a.size() > SAMAX) valid = false; }
        if (P > 1 && (!valid || maxR > SRMAX)) continue;       // next simpler configuration
        h->max_panel_rows = maxR;
        h->nnz_blocks = (long long)rows.size() + nb;
        h->use_sparse = maxR <= SRMAX && npad <= (SRMAX * SNB + 1) * SLD;
        if (h->force_dense) h->use_sparse = false;
        if (!h->use_sparse && P > 1) continue;                  // the dense path uses the natural single-node layout
        // ---- elimination tree tables
        std::vector<NodeDesc> nodes((size_t)n_par + 1);
        std::vector<FoldEnt> ents;
        size_t u_total = 0;
        if (P > 1) {
            for (int nd = 0; nd < n_par; ++nd) {
                NodeDesc& N = nodes[nd];
                memset(&N, 0, sizeof N);
                N.k0 = node_k0[nd]; N.k1 = node_k1[nd]; N.na = (int)anc[nd].size();
                for (int a = 0; a < SAMAX; ++a) N.anc[a] = a < N.na ? anc[nd][a] : 0x7fffffff;
                N.u_off = (long long)u_total;
                const size_t ldu = (size_t)N.na * NB;
                u_total += ldu * ldu + ldu;
                u_total = (u_total + 1) & ~(size_t)1;
            }
            NodeDesc& T = nodes[n_par];
            memset(&T, 0, sizeof T);
            T.k0 = h->top_blk; T.k1 = nb; T.na = 0;
            for (int a = 0; a < SAMAX; ++a) T.anc[a] = 0x7fffffff;
            if (u_total > h->topbuf_cap) { int rc = dalloc(h, &h->d_topbuf, u_total); if (rc) return rc; h->topbuf_cap = u_total; }      // addresses go into the entries
            double* const Sd = h->d_msg; double* const rhsd = h->d_msg + (size_t)npad * npad; double* const Ud = h->d_topbuf;
            auto slot_of = [&](int node, int blk) { const auto& a = anc[node]; return (int)(std::find(a.begin(), a.end(), blk) - a.begin()); };
            // assembly lists (extend-add): the parent of a node owns its lowest outside block
            for (int cons = 0; cons <= n_par && valid; ++cons) {
                NodeDesc& C = nodes[cons];
                struct Src { long long key; int node, a, b; };        // key orders destinations; sources of one destination by node index
                std::vector<Src> blocks, rhss;
                const long long BIG = 1ll << 40;
                for (int nd = 0; nd < n_par; ++nd) {
                    if (nd == cons || anc[nd].empty() || blk_node[anc[nd][0]] != cons) continue;       // not a child
                    for (int a = 0; a < (int)anc[nd].size(); ++a) {
                        const int bi = anc[nd][a];
                        if (blk_node[bi] == cons) rhss.push_back({ (long long)bi, nd, a, 0 });
                        else {
                            const int sa = cons < n_par ? slot_of(cons, bi) : -1;
                            if (sa < 0 || sa >= (int)anc[cons].size()) { valid = false; break; }
                            rhss.push_back({ BIG + sa, nd, a, 0 });
                        }
                        for (int b = 0; b <= a; ++b) {
                            const int bj = anc[nd][b];
                            if (blk_node[bj] == cons) blocks.push_back({ (long long)bi * nb + bj, nd, a, b });
                            else {
                                const int sa = slot_of(cons, bi), sb = slot_of(cons, bj);
                                if (cons == n_par || sa >= (int)anc[cons].size() || sb >= (int)anc[cons].size()) { valid = false; break; }
                                blocks.push_back({ BIG + (long long)sa * SAMAX + sb, nd, a, b });
                            }
                        }
                    }
                }
                if (!valid) break;
                // every diagonal block of its own columns is a destination (damping), with or without sources; a node without
                // children (a leaf) has no list at all and damps its rows directly
                if (!blocks.empty() || !rhss.empty() || cons == n_par)
                    for (int k = C.k0; k < C.k1; ++k) blocks.push_back({ (long long)k * nb + k, -1, 0, 0 });
                auto cmp = [](const Src& x, const Src& y) { return x.key != y.key ? x.key < y.key : x.node < y.node; };
                std::sort(blocks.begin(), blocks.end(), cmp); std::sort(rhss.begin(), rhss.end(), cmp);
                const long long ldc = (long long)C.na * NB;
                auto emit = [&](const std::vector<Src>& v, bool is_rhs) {
                    for (size_t i = 0; i < v.size();) {
                        size_t j = i;
                        FoldEnt E; memset(&E, 0, sizeof E);
                        E.diag0 = -1;
                        const long long key = v[i].key;
                        if (key >= BIG) {                       // hand-on into this node's own update buffer (all zero at this point)
                            const long long q = key - BIG;
                            E.diag0 = -2;
                            if (is_rhs) { E.dst = Ud + C.u_off + ldc * ldc + q * NB; E.dst_ld = 0; }
                            else { E.dst = Ud + C.u_off + (q / SAMAX) * NB * ldc + (q % SAMAX) * NB; E.dst_ld = (int)ldc; }
                        } else if (is_rhs) { E.dst = rhsd + key * NB; E.dst_ld = 0; }
                        else {
                            const long long bi = key / nb, bj = key % nb;
                            E.dst = Sd + (size_t)(bi * NB) * npad + bj * NB; E.dst_ld = npad;
                            if (bi == bj) E.diag0 = (int)bi * NB;
                        }
                        for (; j < v.size() && v[j].key == key; ++j) {
                            if (v[j].node < 0) continue;
                            if (E.nsrc == SFOLD_SRC) { valid = false; break; }
                            const NodeDesc& N = nodes[v[j].node];
                            const long long ldu = (long long)N.na * NB;
                            E.src[E.nsrc] = is_rhs ? Ud + N.u_off + ldu * ldu + (long long)v[j].a * NB : Ud + N.u_off + (long long)v[j].a * NB * ldu + (long long)v[j].b * NB;
                            E.src_ld[E.nsrc] = (int)ldu;
                            ++E.nsrc;
                        }
                        ents.push_back(E);
                        i = j;
                    }
                };
                C.e0 = (int)ents.size(); emit(blocks, false); C.e1 = (int)ents.size(); emit(rhss, true); C.e2 = (int)ents.size();
            }
            if (!valid) continue;
        }
        // ---- upload
        std::vector<int> mask((size_t)h->npad_max, 0);
        for (int i = 0; i < npad; ++i) mask[i] = h->pos_param[i] >= 0;
        SFM_HIP_TRY(ctx, hipMemcpyAsync(h->d_cam_pos, h->cam_pos.data(), (size_t)nc * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        SFM_HIP_TRY(ctx, hipMemcpyAsync(h->d_posmask, mask.data(), mask.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        if (h->use_sparse) {
            SFM_HIP_TRY(ctx, hipMemcpyAsync(h->d_prow_start, start.data(), start.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
            if (!rows.empty()) SFM_HIP_TRY(ctx, hipMemcpyAsync(h->d_prow, rows.data(), rows.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
            if (P > 1) {
                int rc = SFMHIP_OK;
                h->topbuf_count = u_total;
                if (nodes.size() > h->nodes_cap) { rc = dalloc(h, &h->d_nodes, nodes.size()); if (rc) return rc; h->nodes_cap = nodes.size(); }
                if (ents.size() + 1 > h->ents_cap) { rc = dalloc(h, &h->d_ents, ents.size() + 1); if (rc) return rc; h->ents_cap = ents.size() + 1; }
                SFM_HIP_TRY(ctx, hipMemcpyAsync(h->d_nodes, nodes.data(), nodes.size() * sizeof(NodeDesc), hipMemcpyHostToDevice, ctx->stream));
                if (!ents.empty()) SFM_HIP_TRY(ctx, hipMemcpyAsync(h->d_ents, ents.data(), ents.size() * sizeof(FoldEnt), hipMemcpyHostToDevice, ctx->stream));
                h->lvl_first = first_of_level; h->top_node = n_par;
                h->top_cleared = false;
            }
        }
        h->msg_count = (size_t)h->npad * h->npad + 3 * (size_t)h->npad + SCAL_GMAX_SLOTS + 64;
        h->n_sblk = (int)sblk.size() / 2;
        if (chain_ok && P == 1) {
            h->chain = chain_args;
            h->chain_lds1 = 8 * ch_sub_lds(chain_args.w, chain_args.BB, chain_args.a, chain_args.G, chain_args.n, chain_args.a == chain_args.m);
            h->chain_lds2 = chain_args.a < chain_args.m ? 8 * ch_top_lds(chain_args.w, chain_args.BB, chain_args.m - chain_args.a, chain_args.nw_top, chain_args.n) : 0;
            const size_t nrec = (size_t)ncf * chain_args.rec_stride, nimg = (size_t)(chain_args.P >> chain_args.a) * chain_args.img_doubles + 8;
            int rc = SFMHIP_OK;
            if (nrec > h->chain_rec_cap) { rc = dalloc(h, &h->d_chain_rec, nrec); if (rc) return rc; h->chain_rec_cap = nrec; }
            if (nimg > h->chain_img_cap) { rc = dalloc(h, &h->d_chain_img, nimg); if (rc) return rc; h->chain_img_cap = nimg; }
            SFM_HIP_TRY(ctx, hipFuncSetAttribute((const void*)chain_sub_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10));      // the largest any plan asks for: problems with different plans may be alive together
            SFM_HIP_TRY(ctx, hipFuncSetAttribute((const void*)chain_top_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10));
            h->use_chain = true;
        }
        {
            int rc = SFMHIP_OK;
            if (sblk.size() > h->sblk_cap) { rc = dalloc(h, &h->d_sblk, sblk.size()); if (rc) return rc; h->sblk_cap = sblk.size(); }     // re-planning (set_allreduce, reset) reuses it
            SFM_HIP_TRY(ctx, hipMemcpyAsync(h->d_sblk, sblk.data(), sblk.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
            const size_t need = (size_t)h->n_sblk * NB * NB + (h->msg_count - (size_t)h->npad * h->npad);
            if (need > h->pack_cap) { rc = dalloc(h, &h->d_pack, need); if (rc) return rc; h->pack_cap = need; }
        }
        SFM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (h->o.verbose)
            printf("[sfmhip_ba] reduced system: %d unknowns (%d free cameras, band %d cameras); solver: %s\n", h->n, ncf, w,
                   h->use_chain ? "chain (fronts in LDS)" : h->use_sparse ? (P > 1 ? "nested dissection, a launch per level" : "one workgroup, sparse panels") : "dense blocked");
        if (h->use_chain && h->o.verbose)
            printf("[sfmhip_ba] chain solver: %d leaves of %d-%d cameras, %d waves each, %d tree level(s) in the first kernel, %d in the second\n",
                   h->chain.P, h->chain.q, h->chain.q + (h->chain.r ? 1 : 0), h->chain.G, h->chain.a, h->chain.m - h->chain.a);
        return SFMHIP_OK;
    }
    ctx->last_error = "internal: no solver layout";
    return SFMHIP_E_ARG;
}

// Fold table of the run tiles for the current solver layout: for every entry of [S | rhs | diagU | graw] that the tiles
// touch, the list of (+/-) tile elements that sum to it, segments in storage order.  With D_k the direct tile of
// observation slot k (rows [E_ck (6) | E_K (4) | r]) and Z the (6M + 5)^2 product (rows [E_c0 .. E_cM-1 | E_K | r]):
//   S[ck,ck] = sum D_k[c,c] - Z[ck,ck]     S[ck,cl] = -Z[ck,cl]     S[ck,K] = D_k[c,K] - Z[ck,K]     S[K,K] = sum_k D_k[K,K] - Z[K,K]
//   rhs_ck = D_k[c,r] - Z[ck,r]     rhs_K = sum_k D_k[K,r] - Z[K,r]     diagU = diag D     graw = D[.,r]
static int build_tile_tables(sfmhip_ba* h)
{
    if (!h->use_tiles) return SFMHIP_OK;
    if (h->tile_tab_npad == h->npad && h->tile_tab_cam_pos == h->cam_pos) return SFMHIP_OK;
    sfmhip_ctx* ctx = h->ctx;
    const int ld = h->npad, koff = h->koff;
    const long long np2 = (long long)ld * ld;
    if (np2 + 3ll * ld >= (1ll << 31)) { ctx->last_error = "reduced system too large for the tile fold table"; return SFMHIP_E_ARG; }
    std::vector<std::pair<int, unsigned>> ents;
    ents.reserve((size_t)h->n_tseg * 600);
    const unsigned NEG = 0x80000000u;
    for (const TileSeg& sg : h->tsegs) {
        const int M = sg.M;
        int co[TILE_MMAX];
        for (int k = 0; k < M; ++k) co[k] = h->cam_pos[h->tcams[sg.cams_off + k]];
        auto D = [&](int k, int row, int col) -> unsigned { return (unsigned)((sg.tile_off + k) * 256 + tile_elem(row, col)); };
        auto Z = [&](int a, int b) -> unsigned {
            if (a < b) std::swap(a, b);
            const int tr = a / 16, tc = b / 16;
            return (unsigned)((sg.tile_off + M + tr * (tr + 1) / 2 + tc) * 256 + tile_elem(a % 16, b % 16));
        };
        auto sdst = [&](int r, int c) -> int { if (r < c) std::swap(r, c); return r * ld + c; };
        for (int k = 0; k < M; ++k) {
            if (co[k] < 0) continue;
            for (int i = 0; i < 6; ++i) {
                for (int j = 0; j <= i; ++j) { ents.push_back({ sdst(co[k] + i, co[k] + j), D(k, i, j) }); ents.push_back({ sdst(co[k] + i, co[k] + j), Z(6 * k + i, 6 * k + j) | NEG }); }
                if (!h->fixK)
                    for (int j = 0; j < 4; ++j) { ents.push_back({ sdst(co[k] + i, koff + j), D(k, i, 6 + j) }); ents.push_back({ sdst(co[k] + i, koff + j), Z(6 * k + i, 6 * M + j) | NEG }); }
                ents.push_back({ (int)np2 + co[k] + i, D(k, i, 10) }); ents.push_back({ (int)np2 + co[k] + i, Z(6 * k + i, 6 * M + 4) | NEG });
                ents.push_back({ (int)np2 + ld + co[k] + i, D(k, i, i) });
                ents.push_back({ (int)np2 + 2 * ld + co[k] + i, D(k, i, 10) });
            }
            for (int l = 0; l < k; ++l) {
                if (co[l] < 0) continue;
                for (int i = 0; i < 6; ++i)
                    for (int j = 0; j < 6; ++j) {
                        ents.push_back({ sdst(co[k] + i, co[l] + j), Z(6 * k + i, 6 * l + j) | NEG });
                        if (co[k] == co[l] && i == j) ents.push_back({ sdst(co[k] + i, co[l] + j), Z(6 * k + i, 6 * l + j) | NEG });     // one camera twice: block + its transpose
                    }
            }
        }
        if (!h->fixK) {
            for (int i = 0; i < 4; ++i) {
                for (int j = 0; j <= i; ++j) {
                    for (int k = 0; k < M; ++k) ents.push_back({ sdst(koff + i, koff + j), D(k, 6 + i, 6 + j) });
                    ents.push_back({ sdst(koff + i, koff + j), Z(6 * M + i, 6 * M + j) | NEG });
                }
                for (int k = 0; k < M; ++k) {
                    ents.push_back({ (int)np2 + koff + i, D(k, 6 + i, 10) });
                    ents.push_back({ (int)np2 + ld + koff + i, D(k, 6 + i, 6 + i) });
                    ents.push_back({ (int)np2 + 2 * ld + koff + i, D(k, 6 + i, 10) });
                }
                ents.push_back({ (int)np2 + koff + i, Z(6 * M + i, 6 * M + 4) | NEG });
            }
        }
    }
    std::stable_sort(ents.begin(), ents.end(), [](const std::pair<int, unsigned>& a, const std::pair<int, unsigned>& b) { return a.first < b.first; });
    // destinations with long source lists first (a workgroup each in ba_tile_reduce_kernel), then the rest in ascending order
    std::vector<int> start, dst, dst2; std::vector<unsigned> src; src.reserve(ents.size());
    h->rd_n_long = 0;
    for (int pass = 0; pass < 2; ++pass)
        for (size_t e = 0; e < ents.size();) {
            size_t f = e;
            while (f < ents.size() && ents[f].first == ents[e].first) ++f;
            if ((f - e > 192) == (pass == 0)) {
                start.push_back((int)src.size()); dst.push_back(ents[e].first);
                int m = -1;
                if (ents[e].first < np2) { const int r = ents[e].first / ld, c = ents[e].first % ld; if (r != c) m = c * ld + r; }
                dst2.push_back(m);
                for (size_t g = e; g < f; ++g) src.push_back(ents[g].second);
                if (pass == 0) ++h->rd_n_long;
            }
            e = f;
        }
    start.push_back((int)src.size());
    h->rd_nd = (int)dst.size();
    int rc = SFMHIP_OK;
    if (dst.size() + 1 > h->rd_dst_cap) {
        rc = dalloc(h, &h->d_rd_start, dst.size() + 1); if (rc) return rc;
        rc = dalloc(h, &h->d_rd_dst, dst.size() + 1); if (rc) return rc;
        rc = dalloc(h, &h->d_rd_dst2, dst.size() + 1); if (rc) return rc;
        h->rd_dst_cap = dst.size() + 1;
    }
    if (src.size() + 1 > h->rd_src_cap) { rc = dalloc(h, &h->d_rd_src, src.size() + 1); if (rc) return rc; h->rd_src_cap = src.size() + 1; }
    SFM_HIP_TRY(ctx, hipMemcpyAsync(h->d_rd_start, start.data(), start.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    if (!dst.empty()) {
        SFM_HIP_TRY(ctx, hipMemcpyAsync(h->d_rd_dst, dst.data(), dst.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        SFM_HIP_TRY(ctx, hipMemcpyAsync(h->d_rd_dst2, dst2.data(), dst2.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    }
    if (!src.empty()) SFM_HIP_TRY(ctx, hipMemcpyAsync(h->d_rd_src, src.data(), src.size() * sizeof(unsigned), hipMemcpyHostToDevice, ctx->stream));
    SFM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    h->tile_tab_npad = h->npad; h->tile_tab_cam_pos = h->cam_pos;
    return SFMHIP_OK;
}

// iteration 0 work: jacobi scaling from the column norms at x0, |x0|
static int ba_start(sfmhip_ba* h)
{
    sfmhip_ctx* ctx = h->ctx;
    hipStream_t st = ctx->stream;
    { int rc = build_solver_plan(h); if (rc) return rc; }
    { int rc = build_tile_tables(h); if (rc) return rc; }
    const size_t np3 = 3 * (size_t)h->np;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((h->npad_max + 255) / 256)), dim3(256), 0, st, h->d_scale_c, (size_t)h->npad_max, 1.0);
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((np3 + 255) / 256 + 1)), dim3(256), 0, st, h->d_scale_p, np3, 1.0);
    if (h->o.jacobi_scaling) {
        int rc = enqueue_linearize(h, h->o.initial_trust_region_radius, false); if (rc) return rc;
        const double* diagU = h->d_msg + (size_t)h->npad * h->npad + h->npad;
        hipLaunchKernelGGL(ba_scale_kernel, dim3((h->npad + 255) / 256 + 1), dim3(256), 0, st, diagU, h->d_scale_c, h->npad, (const int*)h->d_posmask);
        hipLaunchKernelGGL(ba_scale_kernel, dim3((unsigned)((np3 + 255) / 256 + 1)), dim3(256), 0, st, h->d_colsq_p, h->d_scale_p, (int)np3, (const int*)nullptr);
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

// phase / kernel timings of an iteration whose events have completed (kept off the accept/reject path)
static void read_pending_timing(sfmhip_ba* h)
{
    if (h->pending_build < 0) return;
    hipEvent_t* tb = h->evb[h->pending_build]; hipEvent_t* ti = h->evi[h->pending_iter];
    float a = 0, b = 0, c = 0, k1 = 0, k2 = 0, k3 = 0;
    (void)hipEventElapsedTime(&a, tb[0], ti[0]); (void)hipEventElapsedTime(&b, ti[0], ti[1]); (void)hipEventElapsedTime(&c, ti[1], ti[2]);
    (void)hipEventElapsedTime(&k1, tb[1], tb[2]); (void)hipEventElapsedTime(&k2, tb[3], tb[4]);
    if (h->build_fused[h->pending_build] || h->use_tiles) k2 = 0.0f;
    if (h->use_sparse && h->nseg > 1) (void)hipEventElapsedTime(&k3, ti[3], ti[4]);
    h->phase_acc[0] += a; h->phase_acc[1] += b; h->phase_acc[2] += c; h->phase_acc[3] += a + b + c;
    h->phase_acc[4] += k1; h->phase_acc[5] += k2; h->phase_acc[6] += k3; h->phase_cnt++;
    h->pending_build = h->pending_iter = -1;
}

// the LM loop.  forced: run exactly max_it more iterations, tolerance checks disabled.
static int ba_loop(sfmhip_ba* h, int max_it, bool forced)
{
    sfmhip_ctx* ctx = h->ctx;
    hipStream_t st = ctx->stream;
    const sfm_ba_options& o = h->o;
    if (!h->started) { const auto ts = std::chrono::steady_clock::now(); int rc = ba_start(h); if (rc) return rc; h->start_ms = ms_since(ts); }
    const int it_end = h->iter + max_it;
    const size_t np2 = (size_t)h->npad * h->npad;
    const double* d_scal = h->d_msg + np2 + 3 * (size_t)h->npad;
    // Per iteration the host needs four scalars back before it can accept or reject the step.  SFMHIP_SPECULATE=1 enqueues
    // the NEXT linearisation at the candidate parameters right behind the scalar copies (guessing the radius growth of a
    // good step), so that an accepted step finds its build already running.  Measured at C4: 11 of 12 guesses hit, results
    // bit-identical, but no gain (0.695 vs 0.686 ms per step): the round trip is ~10 us of a 0.67 ms iteration and the extra
    // enqueue work costs as much.  Off by default.
    // On several ranks the speculation is how the LM iteration gets by with ONE collective: the step's five scalars ride in the
    // message of the speculative linearisation at the candidate (accepted with the guessed radius: the next iteration starts
    // from it; rejected or another radius: it is discarded and the next iteration linearises again -- one more collective).
#ifdef SFMHIP_EXPERIMENTS
    static const bool speculate_env = getenv("SFMHIP_SPECULATE") != nullptr;
#else
    constexpr bool speculate_env = false;
#endif
    const bool folded = h->ar_fn != nullptr;
    const bool speculate = speculate_env || folded;
    for (;;) {
        if (h->iter >= it_end) { h->termination = SFMHIP_BA_NO_CONVERGENCE; break; }
        if (!forced && h->radius < o.min_trust_region_radius) { h->termination = SFMHIP_BA_CONVERGENCE; break; }
        int rc = SFMHIP_OK;
        if (!h->built) { rc = enqueue_build(h, h->radius, false, true); if (rc) return rc; }
        const int par = h->build_parity;            // events of the build this iteration consumes
        h->iter_parity ^= 1;
        hipEvent_t* ti = h->evi[h->iter_parity];
        h->built = false;                           // damping and the in-place factorisation consume it
        if (h->use_sparse || h->use_chain) { h->solver_damps = true; h->damp_radius = h->radius; }     // damping rides in the solver kernels
        else { rc = enqueue_damp(h, h->radius); if (rc) return rc; }
        const bool timing = ctx->timing && h->build_timed;
        if (timing) SFM_HIP_TRY(ctx, hipEventRecord(ti[0], st));
        rc = enqueue_solve(h); if (rc) return rc;
        if (timing) SFM_HIP_TRY(ctx, hipEventRecord(ti[1], st));
        h->publish_in_back = true; h->fold_step_scalars = folded;
        rc = enqueue_back(h, h->radius); h->publish_in_back = false; h->fold_step_scalars = false; if (rc) return rc;
        if (timing) SFM_HIP_TRY(ctx, hipEventRecord(ti[2], st));
        // the point blocks are damped inside the build, so the speculation must also guess the next radius: a step with
        // rho >= 0.937 (the normal case while LM is making progress) grows it by exactly 1 / (1/3)
        bool speculated = false;
        const double spec_radius = std::min(o.max_trust_region_radius, h->radius / (1.0 / 3.0));
        // folded: ba_back_reduce_kernel has parked this linearisation's cost and gradient maximum in d_back4[5..6] (the build below
        // overwrites the message tail they live in), and the exchange of the build sums d_back4[0..4] over the ranks
        if (folded) {
            read_pending_timing(h);         // the previous iteration's events, before this build re-records the event set they share
            rc = enqueue_build(h, spec_radius, true, true, h->d_back4); if (rc) return rc; speculated = true;
        }
        // one wave gathers the nine scalars into pinned host memory and bumps a sequence number the host polls; it also
        // re-arms the error flag.  The next build's zero-fill does not depend on the decision: it runs while the host decides.
        unsigned long long seq = h->pub_seq;
        if (!h->published) {
            seq = ++h->pub_seq;
            hipLaunchKernelGGL(ba_publish_kernel, dim3(1), dim3(64), 0, st, folded ? (const double*)(h->d_back4 + 5) : d_scal, h->d_back4, h->d_cam2, h->d_err, h->h_scal, seq,
                               folded ? 2 : (h->ar_fn ? 0 : 1));
        }
        if (!folded && speculate && h->iter + 1 < it_end) { rc = enqueue_build(h, spec_radius, true, true); if (rc) return rc; speculated = true; }
        read_pending_timing(h);                     // the PREVIOUS iteration's events, while the GPU works on this one
        if (timing) { h->pending_build = par; h->pending_iter = h->iter_parity; }
        {   // spin on the sequence number; if the stream drains without publishing (a failed launch) the query ends the wait.
            // No event behind the publishing kernel: a hipEventRecord there held the next kernel back by ~10 us.
            // With an all-reduce hook in the iteration a peer that died leaves this stream waiting inside the collective for ever: the
            // wait then has a deadline (60 s: a thousand times the slowest iteration measured) and ends in SFMHIP_E_COMM.
            volatile unsigned long long* flag = (volatile unsigned long long*)(h->h_scal + 15);
            unsigned spins = 0;
            const auto t_wait = std::chrono::steady_clock::now();
            while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
                if ((++spins & 0xffff) != 0) continue;
                if (hipStreamQuery(st) != hipErrorNotReady) {
                    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) { ctx->last_error = "bundle adjustment: the step scalars were not published"; return SFMHIP_E_HIP; }
                    break;
                }
                if (h->ar_fn && ms_since(t_wait) > 60e3) { ctx->last_error = "bundle adjustment: a collective did not complete within 60 s (a peer rank gone?)"; return SFMHIP_E_COMM; }
            }
        }
        const double cost = h->h_scal[0], gmax = h->h_scal[1], mcc = h->h_scal[2], cand_raw = h->h_scal[3];
        const double dn = h->h_scal[4] + h->h_scal[6], xn = h->h_scal[5] + h->h_scal[7];
        const int err = (int)h->h_scal[8];
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
                std::swap(h->d_campre, h->d_campre_c);
                h->x_norm = std::sqrt(xn); h->x_cost = cand;
                const double t = 2.0 * rho - 1.0;
                h->radius = h->radius / std::max(1.0 / 3.0, 1.0 - t * t * t);
                h->radius = std::min(o.max_trust_region_radius, h->radius);
                h->built = speculated && h->radius == spec_radius;      // built at what is now the current point, with the radius it guessed
                h->nu = 2.0; ++h->nsucc; accepted = true;
            } else {
                h->radius = h->radius / h->nu; h->nu *= 2.0;
            }
        }
#ifdef SFMHIP_EXPERIMENTS
        if (o.verbose) fprintf(stderr, "[sfmhip_ba dbg] err %d mcc %.6e cand %.12e dn %.3e\n", err, mcc, cand_raw, dn);
#endif
        if (o.verbose)
            printf("[sfmhip_ba] it %d cost %.12e gmax %.3e radius %.3e %s\n", h->iter, h->x_cost, gmax, h->radius, accepted ? "ok" : "rejected");
    }
    (void)hipStreamSynchronize(st);
    read_pending_timing(h);
    return SFMHIP_OK;
}

static void fill_summary(const sfmhip_ba* h, sfm_ba_summary* s, double t_s)
{
    if (!s) return;
    s->termination = h->termination; s->iterations = h->iter; s->successful_steps = h->nsucc;
    s->num_residuals = 2 * h->nobs; s->initial_cost = h->initial_cost < 0 ? 0.0 : h->initial_cost; s->final_cost = h->x_cost;
    s->final_radius = h->radius; s->final_gradient_max_norm = h->gmax; s->total_time_s = t_s;
    // a resident problem: the call's own share (sfmhip_ba_solve overwrites these with the whole call, as Ceres reports it)
    s->preprocessor_time_s = h->start_ms * 1e-3; s->minimizer_time_s = t_s - h->start_ms * 1e-3; s->postprocessor_time_s = 0.0;
}

// Device temporaries of sfmhip_ba_create, out of the context's block cache.
struct SetupTemps {
    sfmhip_ctx* ctx; std::vector<void*> blocks;
    explicit SetupTemps(sfmhip_ctx* c) : ctx(c) {}
    ~SetupTemps() { for (void* p : blocks) sfm_pool_put(ctx, p); }       // stream-ordered reuse: back to the context's cache
    template <typename T> int get(T** p, size_t count)
    {
        void* q = nullptr;
        int rc = sfm_pool_get(ctx, (count > 0 ? count : 1) * sizeof(T), &q); if (rc) return rc;
        blocks.push_back(q); *p = (T*)q;
        return SFMHIP_OK;
    }
};

// Orderings of the observation list, built on the device (ba_setup.hpp): points sorted by the set of cameras that see
// them (lexicographic on the ascending camera list; internal only, sfmhip_ba_get_params hands them back in the caller's
// order) -- every per-camera and per-camera-pair walk then gathers from runs of neighbouring point records instead of
// from all over HBM (C4 on MI355X: linearisation 0.41 -> 0.31 ms, back-substitution 0.13 -> 0.08 ms) --, a point's
// observations in ascending camera order (the k-th observation of every point of a run then belongs to the same camera,
// which the run tiles rely on), the camera-ordered copy, and the camera-pair lists for the off-diagonal Schur blocks.
static int ba_build_orderings(sfmhip_ba* h, const double* pts, const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_uv)
{
    sfmhip_ctx* ctx = h->ctx;
    hipStream_t st = ctx->stream;
    const int nc = h->nc, np = h->np, nobs = h->nobs;
    SetupTemps T(ctx);
    int rc = SFMHIP_OK;
#define TRY_RC(x) do { rc = (x); if (rc) return rc; } while (0)
    const auto t0 = std::chrono::steady_clock::now();
    // verbose >= 2: host clock at every step, each behind a stream sync (diagnostic; changes the overlap it measures)
    auto stamp = [&](const char* what) { if (h->o.verbose >= 2) { (void)hipStreamSynchronize(st); printf("[sfmhip_ba setup] %-28s %8.3f ms\n", what, ms_since(t0)); } };
    // ---- raw inputs -> HBM
    int *d_rc = nullptr, *d_rp = nullptr; double *d_ruv = nullptr, *d_rpts = nullptr;
    TRY_RC(T.get(&d_rc, (size_t)nobs)); TRY_RC(T.get(&d_rp, (size_t)nobs)); TRY_RC(T.get(&d_ruv, 2 * (size_t)nobs)); TRY_RC(T.get(&d_rpts, 3 * (size_t)np));
    stamp("input buffers");
    // the two index arrays first: the observation sort runs while the pixels and the points are still on their way
    TRY_RC(sfm_upload(ctx, d_rc, obs_cam, (size_t)nobs * sizeof(int)));
    TRY_RC(sfm_upload(ctx, d_rp, obs_pt, (size_t)nobs * sizeof(int)));
    stamp("index arrays uploaded");
    // ---- persistent tables
    TRY_RC(dalloc(h, &h->d_pt_start, (size_t)np + 1)); TRY_RC(dalloc(h, &h->d_ocam, (size_t)nobs)); TRY_RC(dalloc(h, &h->d_opt, (size_t)nobs));
    TRY_RC(dalloc(h, &h->d_ouv, 2 * (size_t)nobs)); TRY_RC(dalloc(h, &h->d_cam_start, (size_t)nc + 1)); TRY_RC(dalloc(h, &h->d_cam_pt, (size_t)nobs));
    TRY_RC(dalloc(h, &h->d_cam_uv, 2 * (size_t)nobs)); TRY_RC(dalloc(h, &h->d_slot, (size_t)np)); TRY_RC(dalloc(h, &h->d_blk_crange, 2 * (size_t)h->n_pt_blocks));
    TRY_RC(dalloc(h, &h->d_pts, 3 * (size_t)np)); TRY_RC(dalloc(h, &h->d_pts0, 3 * (size_t)np)); TRY_RC(dalloc(h, &h->d_ptsc, 3 * (size_t)np));
    // ---- temporaries of the observation / point phase
    const size_t nmax = (size_t)std::max(nobs, np);
    su64 *obsK[2], *ptK[2]; su32 *obsV[2], *ptV[2], *st_pt, *hist, *bsum; int* d_flags; su64* d_total;
    for (int i = 0; i < 2; ++i) { TRY_RC(T.get(&obsK[i], (size_t)nobs)); TRY_RC(T.get(&obsV[i], (size_t)nobs)); TRY_RC(T.get(&ptK[i], (size_t)np)); TRY_RC(T.get(&ptV[i], (size_t)np)); }
    TRY_RC(T.get(&st_pt, (size_t)np + 1));
    TRY_RC(T.get(&hist, 256 * setup_rs_tiles(nmax) + 1)); TRY_RC(T.get(&bsum, setup_scan_tiles(std::max(256 * setup_rs_tiles(nmax), nmax + 1)) + 1));
    TRY_RC(T.get(&d_flags, 4)); TRY_RC(T.get(&d_total, 2));
    SFM_HIP_TRY(ctx, hipMemsetAsync(d_flags, 0, 4 * sizeof(int), st));
    stamp("tables + temporaries");
    const int cb = std::max(1, setup_bit_width((su64)nc - 1)), pb = std::max(1, setup_bit_width((su64)std::max(np, 1) - 1));
    const unsigned gobs = (unsigned)ceil_div(std::max(nobs, 1), 256), gpt = (unsigned)ceil_div(std::max(np, 1), 256);
    // ---- observations by (point, camera, caller's index)
    int ro = 0;
    if (nobs) {
        hipLaunchKernelGGL(setup_obs_key_kernel, dim3(gobs), dim3(256), 0, st, (const int*)d_rc, (const int*)d_rp, nobs, nc, np, cb, obsK[0], d_flags);
        SetupSortBufs B = { { obsK[0], obsK[1] }, { obsV[0], obsV[1] }, hist, bsum };
        ro = setup_radix_sort(st, B, (size_t)nobs, cb + pb, true);
    }
    // first observation of every point in the sorted list (no counters, no atomics: the run boundaries of the sorted keys), longest track
    hipLaunchKernelGGL(setup_starts_kernel, dim3((unsigned)ceil_div(nobs + 1, 256)), dim3(256), 0, st, (const su64*)obsK[ro], (size_t)nobs, cb, (su64)np, st_pt);
    hipLaunchKernelGGL(setup_max_kernel, dim3((unsigned)std::min(1024, (int)gpt)), dim3(256), 0, st, (const su32*)st_pt, np, (su32*)(d_flags + 1));
    int flags[4] = { 0, 0, 0, 0 };
    stamp("observation sort");
    // behind the sort on the stream: the pixels and the points (the host fills the pinned staging buffers meanwhile)
    TRY_RC(sfm_upload(ctx, d_ruv, obs_uv, 2 * (size_t)nobs * sizeof(double)));
    TRY_RC(sfm_upload(ctx, d_rpts, pts, 3 * (size_t)np * sizeof(double)));
    stamp("pixels + points uploaded");
    // (no early return between this copy into a stack array and the synchronisation behind it)
    {
        const hipError_t e1 = hipMemcpyAsync(flags, d_flags, sizeof flags, hipMemcpyDeviceToHost, st), e2 = hipStreamSynchronize(st);
        SFM_HIP_TRY(ctx, e1); SFM_HIP_TRY(ctx, e2);
    }
    if (flags[0]) { ctx->last_error = "bad argument: observation with a camera or point index out of range"; return SFMHIP_E_ARG; }
    h->setup_ms[0] = ms_since(t0);
    // ---- points by camera list: LSD over groups of list positions, last group first
    {
        const int mmax = std::max(flags[1], 1), b = std::max(1, setup_bit_width((su64)nc)), G = 64 / b, groups = ceil_div(mmax, G);
        su64* K[2] = { ptK[0], ptK[1] }; su32* V[2] = { ptV[0], ptV[1] };
        for (int g = groups - 1; g >= 0 && np > 0; --g) {
            const int j0 = g * G, npos = std::min(G, mmax - j0);
            hipLaunchKernelGGL(setup_ptkey_kernel, dim3(gpt), dim3(256), 0, st, g == groups - 1 ? (const su32*)nullptr : (const su32*)V[0], np, (const su32*)st_pt,
                               (const su64*)obsK[ro], (1ull << cb) - 1ull, j0, npos, b, K[0]);
            SetupSortBufs B = { { K[0], K[1] }, { V[0], V[1] }, hist, bsum };
            if (setup_radix_sort(st, B, (size_t)np, npos * b, g == groups - 1)) { std::swap(K[0], K[1]); std::swap(V[0], V[1]); }
        }
        SFM_HIP_TRY(ctx, hipMemsetAsync(h->d_pt_start, 0, ((size_t)np + 1) * sizeof(int), st));
        if (np) hipLaunchKernelGGL(setup_slot_kernel, dim3(gpt), dim3(256), 0, st, (const su32*)V[0], np, (const su32*)st_pt, h->d_slot, (su32*)h->d_pt_start);
        setup_enqueue_scan(st, (su32*)h->d_pt_start, (size_t)np + 1, bsum, nullptr);
    }
    stamp("point sort");
    // ---- observations and points into storage order, the camera-ordered copy, the camera range of every 256 points
    if (nobs) {
        hipLaunchKernelGGL(setup_fill_obs_kernel, dim3(gobs), dim3(256), 0, st, (const su64*)obsK[ro], (const su32*)obsV[ro], nobs, cb, (const su32*)st_pt, (const int*)h->d_slot,
                           (const int*)h->d_pt_start, (const double2*)d_ruv, h->d_ocam, h->d_opt, (double2*)h->d_ouv, obsK[ro ^ 1]);
        SetupSortBufs B = { { obsK[ro ^ 1], obsK[ro] }, { obsV[ro ^ 1], obsV[ro] }, hist, bsum };
        const int r = setup_radix_sort(st, B, (size_t)nobs, cb, true);
        hipLaunchKernelGGL(setup_starts_kernel, dim3((unsigned)ceil_div(nobs + 1, 256)), dim3(256), 0, st, (const su64*)B.k[r], (size_t)nobs, 0, (su64)nc, (su32*)h->d_cam_start);
        hipLaunchKernelGGL(setup_cam_copy_kernel, dim3(gobs), dim3(256), 0, st, (const su32*)B.v[r], nobs, (const int*)h->d_opt, (const double2*)h->d_ouv, h->d_cam_pt,
                           (double2*)h->d_cam_uv);
    }
    if (np) {
        hipLaunchKernelGGL(setup_permute_pts_kernel, dim3(gpt), dim3(256), 0, st, (const double*)d_rpts, (const int*)h->d_slot, np, h->d_pts, 1);
        SFM_HIP_TRY(ctx, hipMemcpyAsync(h->d_pts0, h->d_pts, 3 * (size_t)np * sizeof(double), hipMemcpyDeviceToDevice, st));
        SFM_HIP_TRY(ctx, hipMemcpyAsync(h->d_ptsc, h->d_pts, 3 * (size_t)np * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
    if (!nobs) SFM_HIP_TRY(ctx, hipMemsetAsync(h->d_cam_start, 0, ((size_t)nc + 1) * sizeof(int), st));
    hipLaunchKernelGGL(setup_crange_kernel, dim3(h->n_pt_blocks), dim3(256), 0, st, (const int*)h->d_pt_start, (const int*)h->d_ocam, np, h->d_blk_crange);
    stamp("storage order + camera copy");
    // ---- camera-pair lists for the off-diagonal Schur blocks (and same-camera pairs)
    su32* npair = nullptr;
    TRY_RC(T.get(&npair, (size_t)np + 1));
    SFM_HIP_TRY(ctx, hipMemsetAsync(npair, 0, ((size_t)np + 1) * sizeof(su32), st));
    SFM_HIP_TRY(ctx, hipMemsetAsync(d_total + 1, 0, sizeof(su64), st));
    if (np) hipLaunchKernelGGL(setup_pair_count_kernel, dim3(gpt), dim3(256), 0, st, (const int*)h->d_pt_start, (const int*)h->d_ocam, np, h->fix0, npair, d_total + 1);
    setup_enqueue_scan(st, npair, (size_t)np + 1, bsum, d_total);
    su64 total_pairs = 0;
    std::vector<int> cam_start((size_t)nc + 1);
    SFM_HIP_TRY(ctx, hipMemcpyAsync(&total_pairs, d_total + 1, sizeof(su64), hipMemcpyDeviceToHost, st));      // the count kernel's own 64-bit sum: the scan's counters may have wrapped
    SFM_HIP_TRY(ctx, hipMemcpyAsync(cam_start.data(), h->d_cam_start, cam_start.size() * sizeof(int), hipMemcpyDeviceToHost, st));
    SFM_HIP_TRY(ctx, hipStreamSynchronize(st));
    SFM_HIP_TRY(ctx, hipGetLastError());
    h->setup_ms[1] = ms_since(t0);
    if (total_pairs >= (1ull << 31) - 4096) { ctx->last_error = "bundle adjustment: more than 2^31 observation pairs"; return SFMHIP_E_ARG; }
    const size_t npairs = (size_t)total_pairs;
    int schur_chunk = 512;            // pairs per wave: 8 trips of the 64-lane loop, then one cross-lane reduction
#ifdef SFMHIP_EXPERIMENTS
    if (const char* e = getenv("SFMHIP_SCHUR_CHUNK")) schur_chunk = std::max(64, atoi(e));
#endif
    std::vector<int> blk_cam, blk_chunk;
    std::vector<int4> chunk_desc;
    if (npairs) {
        su64 *pK[2], *blk_key = nullptr; su32 *pV[2], *flag = nullptr, *hist2 = nullptr, *bsum2 = nullptr, *blk_first = nullptr; int2* raw = nullptr;
        for (int i = 0; i < 2; ++i) { TRY_RC(T.get(&pK[i], npairs)); TRY_RC(T.get(&pV[i], npairs)); }
        TRY_RC(T.get(&raw, npairs)); TRY_RC(T.get(&flag, npairs + 1));
        TRY_RC(T.get(&hist2, 256 * setup_rs_tiles(npairs) + 1)); TRY_RC(T.get(&bsum2, setup_scan_tiles(std::max(256 * setup_rs_tiles(npairs), npairs + 1)) + 1));
        hipLaunchKernelGGL(setup_pair_gen_kernel, dim3(gobs), dim3(256), 0, st, (const int*)h->d_pt_start, (const int*)h->d_ocam, (const int*)h->d_opt, nobs, h->fix0, nc,
                           (const su32*)npair, pK[0], raw);
        stamp("pair generation");
        SetupSortBufs B = { { pK[0], pK[1] }, { pV[0], pV[1] }, hist2, bsum2 };
        const int r = setup_radix_sort(st, B, npairs, setup_bit_width((su64)nc * (su64)nc - 1), true);
        stamp("pair sort");
        const unsigned gpair = (unsigned)((npairs + 1 + 255) / 256);
        hipLaunchKernelGGL(setup_flag_kernel, dim3(gpair), dim3(256), 0, st, (const su64*)pK[r], npairs, flag);
        setup_enqueue_scan(st, flag, npairs + 1, bsum2, d_total);
        su64 nblk64 = 0;
        SFM_HIP_TRY(ctx, hipMemcpyAsync(&nblk64, d_total, sizeof(su64), hipMemcpyDeviceToHost, st));
        TRY_RC(dalloc(h, &h->d_items, npairs));
        hipLaunchKernelGGL(setup_items_kernel, dim3(gpair), dim3(256), 0, st, (const su32*)pV[r], (const int2*)raw, (const int*)h->d_opt, npairs, h->d_items);
        SFM_HIP_TRY(ctx, hipStreamSynchronize(st));
        const size_t nblk = (size_t)nblk64;
        TRY_RC(T.get(&blk_key, nblk)); TRY_RC(T.get(&blk_first, nblk));
        hipLaunchKernelGGL(setup_compact_kernel, dim3(gpair), dim3(256), 0, st, (const su64*)pK[r], (const su32*)flag, npairs, blk_key, blk_first);
        std::vector<su64> hkey(nblk); std::vector<su32> hfirst(nblk + 1);
        SFM_HIP_TRY(ctx, hipMemcpyAsync(hkey.data(), blk_key, nblk * sizeof(su64), hipMemcpyDeviceToHost, st));
        SFM_HIP_TRY(ctx, hipMemcpyAsync(hfirst.data(), blk_first, nblk * sizeof(su32), hipMemcpyDeviceToHost, st));
        SFM_HIP_TRY(ctx, hipStreamSynchronize(st));
        SFM_HIP_TRY(ctx, hipGetLastError());
        hfirst[nblk] = (su32)npairs;
        blk_cam.reserve(2 * nblk); blk_chunk.reserve(nblk + 1); chunk_desc.reserve(npairs / schur_chunk + nblk);
        for (size_t bI = 0; bI < nblk; ++bI) {
            const int ca = (int)(hkey[bI] / (su64)nc), cb2 = (int)(hkey[bI] % (su64)nc);
            blk_cam.push_back(ca); blk_cam.push_back(cb2);
            blk_chunk.push_back((int)chunk_desc.size());
            const size_t t = hfirst[bI], u = hfirst[bI + 1];
            const size_t cnt = u - t, nch = (cnt + schur_chunk - 1) / schur_chunk, per = round_up((int)((cnt + nch - 1) / nch), 64);
            for (size_t a = t; a < u; a += per) chunk_desc.push_back(make_int4(ca, cb2, (int)a, (int)std::min(u, a + per)));
        }
    } else {
        TRY_RC(dalloc(h, &h->d_items, (size_t)0));
    }
    blk_chunk.push_back((int)chunk_desc.size());
    h->nblk = (int)blk_cam.size() / 2;
    h->nchunk = (int)chunk_desc.size();
    h->host_blk_cam = blk_cam;
    h->n_diag_blk = 0;
    for (size_t b = 0; b + 1 < blk_cam.size(); b += 2) if (blk_cam[b] == blk_cam[b + 1]) ++h->n_diag_blk;
    TRY_RC(dupload(h, &h->d_blk_cam, blk_cam.data(), blk_cam.size())); TRY_RC(dupload(h, &h->d_blk_chunk, blk_chunk.data(), blk_chunk.size()));
    TRY_RC(dupload(h, &h->d_chunk_desc, chunk_desc.data(), chunk_desc.size()));
    TRY_RC(dalloc(h, &h->d_part_schur, 36 * (size_t)h->nchunk));
    stamp("pair lists");
    h->setup_ms[2] = ms_since(t0);

    int max_cam = 1;
    for (int c = 0; c < nc; ++c) max_cam = std::max(max_cam, cam_start[c + 1] - cam_start[c]);
    int cam_wg_obs = 2048;            // observations per camera workgroup (8 per thread: the 39-value reduction is paid once per wave)
#ifdef SFMHIP_EXPERIMENTS
    if (const char* e = getenv("SFMHIP_CAM_WG_OBS")) cam_wg_obs = std::max(256, atoi(e));
#endif
    h->cam_split = std::min(32, std::max(1, ceil_div(max_cam, cam_wg_obs)));

    // ---- run tiles (opt-in linearizer = 2): runs of points with one camera list, cut into segments of <= seg_max points (one
    // workgroup each).  Host pass over the finished tables.
    h->tsegs.clear(); h->tcams.clear(); h->n_tseg = 0; h->n_ttiles = 0; h->use_tiles = false;
    if (h->o.linearizer == 2 && np > 0) {
        std::vector<int> pt_start((size_t)np + 1), ocam((size_t)nobs);
        SFM_HIP_TRY(ctx, hipMemcpy(pt_start.data(), h->d_pt_start, pt_start.size() * sizeof(int), hipMemcpyDeviceToHost));
        if (nobs) SFM_HIP_TRY(ctx, hipMemcpy(ocam.data(), h->d_ocam, ocam.size() * sizeof(int), hipMemcpyDeviceToHost));
        int seg_max = 320;
#ifdef SFMHIP_EXPERIMENTS
        if (const char* e = getenv("SFMHIP_TILE_SEG")) seg_max = std::max(16, atoi(e));
#endif
        bool fits = true;
        long long tiles = 0;
        for (int p = 0; p < np && fits;) {
            const int M = pt_start[p + 1] - pt_start[p];
            if (M < 1 || M > TILE_MMAX) { fits = false; break; }
            int q = p + 1;
            while (q < np && pt_start[q + 1] - pt_start[q] == M && std::equal(ocam.begin() + pt_start[p], ocam.begin() + pt_start[p + 1], ocam.begin() + pt_start[q])) ++q;
            const int len = q - p, nseg = ceil_div(len, seg_max), per = round_up(ceil_div(len, nseg), 16);
            const int cams_off = (int)h->tcams.size();
            for (int k = 0; k < M; ++k) h->tcams.push_back(ocam[pt_start[p] + k]);
            const int R = 6 * M + 5, Tn = (R + 15) / 16, NT = M + Tn * (Tn + 1) / 2;
            for (int a = p; a < q; a += per) {
                TileSeg sg; sg.p0 = a; sg.npts = std::min(per, q - a); sg.obs0 = pt_start[a]; sg.M = M; sg.cams_off = cams_off; sg.tile_off = (int)tiles; sg.pad0 = sg.pad1 = 0;
                h->tsegs.push_back(sg); tiles += NT;
            }
            p = q;
        }
        // heaviest segments first (work per point grows with M: more tiles, and two observations per lane from M = 5): the
        // dispatcher hands workgroups out in index order, so the short ones fill the tail
        std::stable_sort(h->tsegs.begin(), h->tsegs.end(), [](const TileSeg& a, const TileSeg& b) { return a.M > b.M; });
        if (tiles * 256 >= (1ll << 31)) fits = false;                // the fold table addresses the tile buffer with 31 bits
        // opt-in only: measured on MI355X the tile kernel + fold take 0.125 + 0.04 ms at C4 against 0.13 ms for the whole
        // per-observation pipeline (profiles/README.md, round 2), so 0 = "choose" resolves to the per-observation kernels
        h->use_tiles = fits;
        if (fits) { h->n_tseg = (int)h->tsegs.size(); h->n_ttiles = tiles; } else { h->tsegs.clear(); h->tcams.clear(); }
    }
#undef TRY_RC
    return SFMHIP_OK;
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
    o->linearizer = 0;
    o->solver = 0;
}

void sfmhip_ba_destroy(sfmhip_ba* h)
{
    SFM_DEVICE_GUARD(h ? h->ctx : nullptr);
    if (!h) return;
    (void)hipStreamSynchronize(h->ctx->stream);
    for (void* p : h->allocs) sfm_pool_put(h->ctx, p);          // kept by the context for the next problem (sfmhip_trim releases them)
    if (h->h_scal) (void)hipHostFree(h->h_scal);
    for (auto& e : h->ev) if (e) (void)hipEventDestroy(e);
    if (h->aux) (void)hipStreamSynchronize(h->aux);         // the stream itself belongs to the context
    for (auto& pr : h->evb) for (auto& e : pr) if (e) (void)hipEventDestroy(e);
    for (auto& pr : h->evi) for (auto& e : pr) if (e) (void)hipEventDestroy(e);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    delete h;
}

int sfmhip_ba_create(sfmhip_ctx* ctx, const double* K4, const double* ext6, int n_cam, const double* pts, int n_pt,
                     const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_uv, int n_obs,
                     const sfm_ba_options* opts, sfmhip_ba** out)
{
    SFM_DEVICE_GUARD(ctx);
    SFM_RANGE("sfmhip_ba_create");
    SFM_ARG_CHECK(ctx, ctx && out && K4 && ext6 && n_cam > 0 && n_pt >= 0 && n_obs >= 0);
    SFM_ARG_CHECK(ctx, (pts || n_pt == 0) && ((obs_cam && obs_pt && obs_uv) || n_obs == 0));
    const auto t0 = std::chrono::steady_clock::now();
    sfmhip_ba* h = new sfmhip_ba();
    h->ctx = ctx;
    if (opts) h->o = *opts; else sfmhip_ba_default_options(&h->o);
    h->nc = n_cam; h->np = n_pt; h->nobs = n_obs;
    h->fix0 = h->o.fix_first_camera ? 1 : 0; h->fixK = h->o.fix_intrinsics ? 1 : 0;
    h->ncf = n_cam - h->fix0; h->koff = 6 * h->ncf; h->n = 6 * h->ncf + (h->fixK ? 0 : 4);
    h->npad = std::max(NB, round_up(h->n, NB)); h->nbk = h->npad / NB;
    // room for the per-node padding of the nested-dissection layout: up to solver_pmax leaves and as many separators
    h->solver_pmax = 1; while (h->solver_pmax < 64 && h->solver_pmax * 16 <= h->ncf) h->solver_pmax *= 2;
    h->npad_max = h->npad + NB * (2 * h->solver_pmax + 2);
    h->n_pt_blocks = std::max(1, ceil_div(n_pt, 256));
    // one arena for the many small and medium device arrays of a problem (a hipMalloc each cost more than the uploads)
    h->arena_chunk = (size_t)n_pt * 360 + (size_t)n_obs * 72 + (size_t)n_cam * 4096 + (1u << 20);
#ifdef SFMHIP_EXPERIMENTS
    h->force_dense = getenv("SFMHIP_DENSE_SOLVER") != nullptr;
    if (const char* e = getenv("SFMHIP_FUSE_MAX_BLOCKS")) h->fuse_max_blocks = atoi(e);      // measurement knob: 0 = always two launches on two streams
#endif

    int rc = SFMHIP_OK;
#define TRY_RC(x) do { rc = (x); if (rc) { sfmhip_ba_destroy(h); return rc; } } while (0)
    TRY_RC(ba_build_orderings(h, pts, obs_cam, obs_pt, obs_uv));
    TRY_RC(dupload(h, &h->d_K, K4, 4)); TRY_RC(dupload(h, &h->d_ext, ext6, 6 * (size_t)n_cam));
    TRY_RC(dupload(h, &h->d_K0, K4, 4)); TRY_RC(dupload(h, &h->d_ext0, ext6, 6 * (size_t)n_cam));
    TRY_RC(dupload(h, &h->d_Kc, K4, 4)); TRY_RC(dupload(h, &h->d_extc, ext6, 6 * (size_t)n_cam));
    if (h->use_tiles) {
        TRY_RC(dupload(h, &h->d_tsegs, h->tsegs.data(), h->tsegs.size())); TRY_RC(dupload(h, &h->d_tcams, h->tcams.data(), h->tcams.size()));
        TRY_RC(dalloc(h, &h->d_tpart, (size_t)h->n_ttiles * 256)); TRY_RC(dalloc(h, &h->d_tpart_seg, 2 * (size_t)h->n_tseg));
        if (hipFuncSetAttribute((const void*)ba_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TILE_LDS_BYTES) != hipSuccess) {
            sfmhip_ba_destroy(h); ctx->last_error = "hipFuncSetAttribute(ba_tile_kernel)"; return SFMHIP_E_HIP;
        }
    }
    TRY_RC(dalloc(h, &h->d_scale_c, (size_t)h->npad_max)); TRY_RC(dalloc(h, &h->d_scale_p, 3 * (size_t)n_pt));
    TRY_RC(dalloc(h, &h->d_campre, CAMPRE * (size_t)n_cam)); TRY_RC(dalloc(h, &h->d_campre_c, CAMPRE * (size_t)n_cam));
    TRY_RC(dalloc(h, &h->d_cam_pos, (size_t)n_cam)); TRY_RC(dalloc(h, &h->d_posmask, (size_t)h->npad_max));
    TRY_RC(dalloc(h, &h->d_prow_start, (size_t)h->npad_max / NB + 2)); TRY_RC(dalloc(h, &h->d_prow, ((size_t)h->npad_max / NB + 1) * SRMAX + 1));
    TRY_RC(dalloc(h, &h->d_Vinv, 6 * (size_t)n_pt)); TRY_RC(dalloc(h, &h->d_bp, 3 * (size_t)n_pt));
    TRY_RC(dalloc(h, &h->d_WK, 12 * (size_t)n_pt)); TRY_RC(dalloc(h, &h->d_colsq_p, 3 * (size_t)n_pt));
    TRY_RC(dalloc(h, &h->d_part_pt, 32 * (size_t)h->n_pt_blocks)); TRY_RC(dalloc(h, &h->d_part_back, 4 * (size_t)h->n_pt_blocks));
    TRY_RC(dalloc(h, &h->d_part_cam, (size_t)CAMACC * n_cam * 32));
    TRY_RC(dalloc(h, &h->d_Linv, (size_t)(h->npad_max / NB) * NB * NB)); TRY_RC(dalloc(h, &h->d_y, (size_t)h->npad_max));
    TRY_RC(dalloc(h, &h->d_back4, 8)); TRY_RC(dalloc(h, &h->d_cam2, 2)); TRY_RC(dalloc(h, &h->d_xnorm, 64)); TRY_RC(dalloc(h, &h->d_err, 1));
    h->msg_count = (size_t)h->npad_max * h->npad_max + 3 * (size_t)h->npad_max + SCAL_GMAX_SLOTS + 64;   // room for <= 64 ranks
    TRY_RC(dalloc(h, &h->d_msg, std::max(h->msg_count, (size_t)h->ncf * h->ncf)));
#undef TRY_RC
    if (hipHostMalloc((void**)&h->h_scal, 16 * sizeof(double)) != hipSuccess) { sfmhip_ba_destroy(h); ctx->last_error = "hipHostMalloc"; return SFMHIP_E_HIP; }
    memset(h->h_scal, 0, 16 * sizeof(double));
    for (auto& e : h->ev) if (hipEventCreate(&e) != hipSuccess) { sfmhip_ba_destroy(h); ctx->last_error = "hipEventCreate"; return SFMHIP_E_HIP; }
    for (auto& pr : h->evb) for (auto& e : pr) if (hipEventCreate(&e) != hipSuccess) { sfmhip_ba_destroy(h); ctx->last_error = "hipEventCreate"; return SFMHIP_E_HIP; }
    for (auto& pr : h->evi) for (auto& e : pr) if (hipEventCreate(&e) != hipSuccess) { sfmhip_ba_destroy(h); ctx->last_error = "hipEventCreate"; return SFMHIP_E_HIP; }
    if (!ctx->aux_stream && hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking) != hipSuccess) ctx->aux_stream = nullptr;
    h->aux = ctx->aux_stream;
    if (!h->aux || hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) { sfmhip_ba_destroy(h); ctx->last_error = "auxiliary stream"; return SFMHIP_E_HIP; }
    if (h->o.verbose >= 2) printf("[sfmhip_ba setup] %-28s %8.3f ms\n", "work arrays, events", ms_since(t0));
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { sfmhip_ba_destroy(h); ctx->last_error = "upload failed"; return SFMHIP_E_HIP; }
    h->setup_ms[3] = ms_since(t0);
    if (h->o.verbose)
        printf("[sfmhip_ba] create %.2f ms (inputs + observation sort %.2f, orderings %.2f, pair lists %.2f)\n", h->setup_ms[3], h->setup_ms[0],
               h->setup_ms[1] - h->setup_ms[0], h->setup_ms[2] - h->setup_ms[1]);
    *out = h;
    return SFMHIP_OK;
}

int sfmhip_ba_set_allreduce(sfmhip_ba* h, sfmhip_allreduce_fn fn, void* user, int rank, int world)
{
    SFM_DEVICE_GUARD(h ? h->ctx : nullptr);
    if (!h || world < 1 || world > 64 || rank < 0 || rank >= world) return SFMHIP_E_ARG;
    h->ar_fn = fn; h->ar_user = user; h->rank = rank; h->world = world;
    h->started = false; h->built = false; h->cleared = false; h->campre_valid = false; h->top_cleared = false;
    return SFMHIP_OK;
}

int sfmhip_ba_reset(sfmhip_ba* h)
{
    SFM_DEVICE_GUARD(h ? h->ctx : nullptr);
    if (!h) return SFMHIP_E_ARG;
    sfmhip_ctx* ctx = h->ctx;
    SFM_HIP_TRY(ctx, hipMemcpyAsync(h->d_K, h->d_K0, 4 * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    SFM_HIP_TRY(ctx, hipMemcpyAsync(h->d_ext, h->d_ext0, 6 * (size_t)h->nc * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    if (h->np) SFM_HIP_TRY(ctx, hipMemcpyAsync(h->d_pts, h->d_pts0, 3 * (size_t)h->np * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    h->started = false; h->built = false; h->cleared = false; h->campre_valid = false; h->top_cleared = false;
    for (double& v : h->phase_acc) v = 0; h->phase_cnt = 0;
    return SFMHIP_OK;
}

int sfmhip_ba_run(sfmhip_ba* h, sfm_ba_summary* summary)
{
    SFM_DEVICE_GUARD(h ? h->ctx : nullptr);
    SFM_RANGE("sfmhip_ba_run");
    if (!h) return SFMHIP_E_ARG;
    const auto t0 = std::chrono::steady_clock::now();
    h->started = false; h->built = false; h->cleared = false; h->campre_valid = false; h->top_cleared = false; h->start_ms = 0;
    const int rc = ba_loop(h, h->o.max_num_iterations, false);
    fill_summary(h, summary, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    return rc;
}

int sfmhip_ba_iterate(sfmhip_ba* h, int n_iter, sfm_ba_summary* summary)
{
    SFM_DEVICE_GUARD(h ? h->ctx : nullptr);
    SFM_RANGE("sfmhip_ba_iterate");
    if (!h || n_iter < 0) return SFMHIP_E_ARG;
    const auto t0 = std::chrono::steady_clock::now();
    for (double& v : h->phase_acc) v = 0; h->phase_cnt = 0; h->start_ms = 0;
    const int rc = ba_loop(h, n_iter, true);
    fill_summary(h, summary, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    return rc;
}

int sfmhip_ba_get_params(sfmhip_ba* h, double* K4, double* ext6, double* pts)
{
    SFM_DEVICE_GUARD(h ? h->ctx : nullptr);
    if (!h) return SFMHIP_E_ARG;
    sfmhip_ctx* ctx = h->ctx;
    SFM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (K4) SFM_HIP_TRY(ctx, hipMemcpy(K4, h->d_K, 4 * sizeof(double), hipMemcpyDeviceToHost));
    if (ext6) SFM_HIP_TRY(ctx, hipMemcpy(ext6, h->d_ext, 6 * (size_t)h->nc * sizeof(double), hipMemcpyDeviceToHost));
    if (pts && h->np) {           // back into the caller's point order on the device, then one copy into the caller's array
        void* tmp = nullptr;
        int rc = sfm_scratch(ctx, 3 * (size_t)h->np * sizeof(double), &tmp); if (rc) return rc;
        hipLaunchKernelGGL(setup_permute_pts_kernel, dim3(ceil_div(h->np, 256)), dim3(256), 0, ctx->stream, (const double*)h->d_pts, (const int*)h->d_slot, h->np, (double*)tmp, 0);
        SFM_HIP_TRY(ctx, hipMemcpyAsync(pts, tmp, 3 * (size_t)h->np * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        SFM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return SFMHIP_OK;
}

int sfmhip_ba_reduced_system(sfmhip_ba* h, double radius, double* S, double* rhs, int* n, double* cost)
{
    SFM_DEVICE_GUARD(h ? h->ctx : nullptr);
    SFM_RANGE("sfmhip_ba_reduced_system");
    if (!h) return SFMHIP_E_ARG;
    sfmhip_ctx* ctx = h->ctx;
    if (n) *n = h->n;
    if (!S && !rhs) return SFMHIP_OK;
    if (!h->started) { int rc = ba_start(h); if (rc) return rc; }
    if (radius == 0.0) return SFMHIP_E_ARG;
    int rc = enqueue_linearize(h, std::fabs(radius), radius > 0.0); if (rc) return rc;
    SFM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const size_t np2 = (size_t)h->npad * h->npad;
    if (h->n) {
        std::vector<double> full(np2 + h->npad);
        SFM_HIP_TRY(ctx, hipMemcpy(full.data(), h->d_msg, full.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int i = 0; i < h->npad; ++i) {
            const int pi = h->pos_param[i];
            if (pi < 0) continue;
            if (rhs) rhs[pi] = full[np2 + i];
            // the lower triangle is the authoritative copy (the only one a multi-rank all-reduce carries)
            if (S) for (int j = 0; j < h->npad; ++j) { const int pj = h->pos_param[j]; if (pj >= 0) S[(size_t)pi * h->n + pj] = full[(size_t)std::max(i, j) * h->npad + std::min(i, j)]; }
        }
    }
    if (cost) SFM_HIP_TRY(ctx, hipMemcpy(cost, h->d_msg + np2 + 3 * (size_t)h->npad, sizeof(double), hipMemcpyDeviceToHost));
    return SFMHIP_OK;
}

int sfmhip_ba_debug_table(sfmhip_ba* h, const char* name, void* out, size_t cap_bytes, size_t* n_bytes)
{
    SFM_DEVICE_GUARD(h ? h->ctx : nullptr);
    if (!h || !name) return SFMHIP_E_ARG;
    sfmhip_ctx* ctx = h->ctx;
    const std::string nm(name);
    const void* src = nullptr; size_t bytes = 0;
    const size_t no = (size_t)h->nobs, np = (size_t)h->np, nc = (size_t)h->nc;
    std::vector<int> host;
    if (nm == "pt_slot") { src = h->d_slot; bytes = np * 4; }
    else if (nm == "pt_start") { src = h->d_pt_start; bytes = (np + 1) * 4; }
    else if (nm == "ocam") { src = h->d_ocam; bytes = no * 4; }
    else if (nm == "opt") { src = h->d_opt; bytes = no * 4; }
    else if (nm == "ouv") { src = h->d_ouv; bytes = no * 16; }
    else if (nm == "cam_start") { src = h->d_cam_start; bytes = (nc + 1) * 4; }
    else if (nm == "cam_pt") { src = h->d_cam_pt; bytes = no * 4; }
    else if (nm == "cam_uv") { src = h->d_cam_uv; bytes = no * 16; }
    else if (nm == "blk_crange") { src = h->d_blk_crange; bytes = 2 * (size_t)h->n_pt_blocks * 4; }
    else if (nm == "blk_cam") { src = h->d_blk_cam; bytes = 2 * (size_t)h->nblk * 4; }
    else if (nm == "blk_chunk") { src = h->d_blk_chunk; bytes = ((size_t)h->nblk + 1) * 4; }
    else if (nm == "chunk_desc") { src = h->d_chunk_desc; bytes = (size_t)h->nchunk * 16; }
    else if (nm == "items") {          // the pair items end where the last chunk ends
        int4 last = make_int4(0, 0, 0, 0);
        if (h->nchunk) SFM_HIP_TRY(ctx, hipMemcpy(&last, h->d_chunk_desc + (h->nchunk - 1), sizeof last, hipMemcpyDeviceToHost));
        src = h->d_items; bytes = (size_t)last.w * 16;
    }
    else if (nm == "setup_ms") { if (n_bytes) *n_bytes = sizeof h->setup_ms; if (out && cap_bytes >= sizeof h->setup_ms) memcpy(out, h->setup_ms, sizeof h->setup_ms); return SFMHIP_OK; }
    else { ctx->last_error = "sfmhip_ba_debug_table: unknown table"; return SFMHIP_E_ARG; }
    if (n_bytes) *n_bytes = bytes;
    if (!out) return SFMHIP_OK;
    if (cap_bytes < bytes) { ctx->last_error = "sfmhip_ba_debug_table: buffer too small"; return SFMHIP_E_ARG; }
    SFM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (bytes) SFM_HIP_TRY(ctx, hipMemcpy(out, src, bytes, hipMemcpyDeviceToHost));
    return SFMHIP_OK;
}

int sfmhip_ba_phase_ms(sfmhip_ba* h, double out_ms[8])
{
    if (!h || !out_ms) return SFMHIP_E_ARG;
    for (int i = 0; i < 8; ++i) out_ms[i] = h->phase_cnt ? h->phase_acc[i] / h->phase_cnt : 0.0;
    out_ms[7] = (double)h->nnz_blocks;
    return SFMHIP_OK;
}

int sfmhip_ba_solve(sfmhip_ctx* ctx, double* K4, double* ext6, int n_cam, double* pts, int n_pt,
                    const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_uv, int n_obs,
                    const sfm_ba_options* opts, sfm_ba_summary* summary)
{
    SFM_DEVICE_GUARD(ctx);
    SFM_RANGE("sfmhip_ba_solve");
    // One call = build the problem + solve + write the parameters back, like bundle_adjustment() (NView:1169-1224); the
    // summary's times cover the whole call the way Ceres' do (total_time_in_seconds includes the preprocessor, NView:1239).
    const auto t0 = std::chrono::steady_clock::now();
    sfmhip_ba* h = nullptr;
    int rc = sfmhip_ba_create(ctx, K4, ext6, n_cam, pts, n_pt, obs_cam, obs_pt, obs_uv, n_obs, opts, &h);
    if (rc) return rc;
    const double create_s = ms_since(t0) * 1e-3;
    sfm_ba_summary local;
    sfm_ba_summary* sm = summary ? summary : &local;
    rc = sfmhip_ba_run(h, sm);
    const auto t1 = std::chrono::steady_clock::now();
    if (rc == SFMHIP_OK) rc = sfmhip_ba_get_params(h, K4, ext6, pts);
    sfmhip_ba_destroy(h);
    sm->preprocessor_time_s += create_s;
    sm->postprocessor_time_s = ms_since(t1) * 1e-3;
    sm->total_time_s = ms_since(t0) * 1e-3;
    return rc;
}

}  // extern "C"
