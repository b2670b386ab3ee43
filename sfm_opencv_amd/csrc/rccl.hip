// rccl.hip -- in-library all-reduce hook over RCCL (xGMI), looked up at run time.
//
// The reference is one process on the CPU (NViewReconstuct.cpp:1334-1524; Ceres runs 4 threads, NView:1218).  Here bundle
// adjustment shards its points over one process per GPU and needs one sum per LM iteration over the packed reduced-system
// message (ba.hip).  sfmhip_ba_set_allreduce takes any hook (torch.distributed for the gloo rehearsals on CPU boxes); this
// file provides the production one: ncclAllReduce(double, sum) in place on the context's stream, called straight from the
// LM loop -- no Python, no torch on the critical path of a 0.3 ms iteration.  librccl is dlopen'ed (like roctx in context.hip),
// so libsfmhip.so keeps its single link-time dependency on the HIP runtime; without the library these entry points return
// SFMHIP_E_COMM.  The launcher creates the unique id on rank 0 (sfmhip_rccl_get_unique_id), ships its 128 bytes to the other
// ranks by whatever it has (torch.distributed broadcast, MPI, a file), and every rank calls sfmhip_rccl_comm_create.
#include "common.hpp"
#include <chrono>
#include <dlfcn.h>

namespace {

// the handful of RCCL declarations used (rccl.h: ncclUniqueId is 128 opaque bytes, ncclDouble = 8, ncclSum = 0, ncclSuccess = 0)
struct RcclUniqueId { char internal[128]; };
typedef int (*GetUniqueIdFn)(RcclUniqueId*);
typedef int (*CommInitRankFn)(void** comm, int nranks, RcclUniqueId id, int rank);
typedef int (*CommDestroyFn)(void* comm);
typedef int (*AllReduceFn)(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t stream);
typedef const char* (*GetErrorStringFn)(int);

struct Rccl {
    void* lib = nullptr;
    GetUniqueIdFn get_unique_id = nullptr; CommInitRankFn comm_init_rank = nullptr; CommDestroyFn comm_destroy = nullptr;
    AllReduceFn all_reduce = nullptr; GetErrorStringFn error_string = nullptr;
    Rccl()
    {
        // a process that already carries an RCCL (PyTorch-ROCm ships its own) must use THAT copy: RTLD_NOLOAD first
        for (const char* name : { "librccl.so.1", "librccl.so" }) { lib = dlopen(name, RTLD_LAZY | RTLD_NOLOAD); if (lib) break; }
        if (!lib)
            for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so" }) { lib = dlopen(name, RTLD_LAZY | RTLD_GLOBAL); if (lib) break; }
        if (!lib) return;
        get_unique_id = (GetUniqueIdFn)dlsym(lib, "ncclGetUniqueId");
        comm_init_rank = (CommInitRankFn)dlsym(lib, "ncclCommInitRank");
        comm_destroy = (CommDestroyFn)dlsym(lib, "ncclCommDestroy");
        all_reduce = (AllReduceFn)dlsym(lib, "ncclAllReduce");
        error_string = (GetErrorStringFn)dlsym(lib, "ncclGetErrorString");
        if (!get_unique_id || !comm_init_rank || !comm_destroy || !all_reduce) lib = nullptr;
    }
    static const Rccl& get() { static const Rccl r; return r; }
};

int rccl_allreduce_hook(void* user, void* d_buf, size_t count, void* stream)
{
    const Rccl& R = Rccl::get();
    if (!R.lib || !user) return -1;
    return R.all_reduce(d_buf, d_buf, count, /*ncclDouble*/ 8, /*ncclSum*/ 0, user, (hipStream_t)stream);
}

}  // namespace

extern "C" {

int sfmhip_rccl_available(void) { return Rccl::get().lib != nullptr; }

int sfmhip_rccl_get_unique_id(void* id128)
{
    const Rccl& R = Rccl::get();
    if (!id128) return SFMHIP_E_ARG;
    if (!R.lib) return SFMHIP_E_COMM;
    RcclUniqueId id;
    if (R.get_unique_id(&id) != 0) return SFMHIP_E_COMM;
    memcpy(id128, id.internal, sizeof id.internal);
    return SFMHIP_OK;
}

int sfmhip_rccl_comm_create(sfmhip_ctx* ctx, const void* id128, int rank, int world, void** comm)
{
    SFM_DEVICE_GUARD(ctx);
    SFM_ARG_CHECK(ctx, ctx && id128 && comm && world >= 1 && rank >= 0 && rank < world);
    const Rccl& R = Rccl::get();
    if (!R.lib) { ctx->last_error = "librccl.so not found"; return SFMHIP_E_COMM; }
    SFM_HIP_TRY(ctx, hipSetDevice(ctx->device));          // ncclCommInitRank binds the communicator to the current device
    RcclUniqueId id;
    memcpy(id.internal, id128, sizeof id.internal);
    void* c = nullptr;
    const int rc = R.comm_init_rank(&c, world, id, rank);
    if (rc != 0) { ctx->last_error = std::string("ncclCommInitRank: ") + (R.error_string ? R.error_string(rc) : "failed"); return SFMHIP_E_COMM; }
    *comm = c;
    return SFMHIP_OK;
}

int sfmhip_rccl_comm_destroy(void* comm)
{
    const Rccl& R = Rccl::get();
    if (!comm) return SFMHIP_OK;
    if (!R.lib) return SFMHIP_E_COMM;
    return R.comm_destroy(comm) == 0 ? SFMHIP_OK : SFMHIP_E_COMM;
}

int sfmhip_ba_set_rccl(sfmhip_ba* problem, void* comm, int rank, int world)
{
    if (!problem || !comm) return SFMHIP_E_ARG;
    if (!Rccl::get().lib) return SFMHIP_E_COMM;
    return sfmhip_ba_set_allreduce(problem, rccl_allreduce_hook, comm, rank, world);
}

// Plain in-place sum of `count` doubles at a device pointer over the communicator, on the context's stream (what the hook does;
// exposed for the launcher's own small exchanges and for tests).
int sfmhip_rccl_allreduce_f64(sfmhip_ctx* ctx, void* comm, void* d_buf, size_t count)
{
    SFM_DEVICE_GUARD(ctx);
    SFM_ARG_CHECK(ctx, ctx && comm && (d_buf || count == 0));
    if (rccl_allreduce_hook(comm, d_buf, count, (void*)ctx->stream) != 0) { ctx->last_error = "ncclAllReduce failed"; return SFMHIP_E_COMM; }
    return SFMHIP_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// bundle adjustment over several GPUs of ONE process ("multi-GPU fan-out is internal"; the reference's main() is
// one process, NView:1334-1524): the points are sharded by first camera over the contexts, one host thread per context builds
// and runs its shard, the packed reduced-system message is summed by RCCL -- or, where two contexts share a device (the one-card
// rehearsal of the tests) or librccl is missing, by a host-staged exchange inside the process.
// ------------------------------------------------------------------------------------------------
#include <condition_variable>
#include <mutex>
#include <thread>

namespace {

// Sum of the ranks' device buffers through pinned host memory: every rank copies out, all wait, every rank adds the N copies in
// rank order (so all ranks hold bit-identical sums) and copies back.  A fallback and a test vehicle, not the production path.
struct LocalComm {
    int world = 1;
    std::mutex mu; std::condition_variable cv; int arrived = 0; unsigned long long gen = 0; bool broken = false;
    std::vector<double*> slot; std::vector<size_t> cap;
    explicit LocalComm(int w) : world(w), slot((size_t)w, nullptr), cap((size_t)w, 0) {}
    ~LocalComm() { for (double* p : slot) if (p) (void)hipHostFree(p); }
    bool barrier()
    {
        std::unique_lock<std::mutex> lk(mu);
        if (broken) return false;
        const unsigned long long g = gen;
        if (++arrived == world) { arrived = 0; ++gen; cv.notify_all(); return true; }
        cv.wait(lk, [&] { return gen != g || broken; });
        return !broken;
    }
    void abort() { std::lock_guard<std::mutex> lk(mu); broken = true; cv.notify_all(); }
};
struct LocalRank { LocalComm* comm; int rank; std::vector<double> sum; };

int local_allreduce_hook(void* user, void* d_buf, size_t count, void* stream)
{
    LocalRank* R = (LocalRank*)user;
    LocalComm* C = R->comm;
    const int r = R->rank;
    if (count > C->cap[r]) {
        if (C->slot[r]) (void)hipHostFree(C->slot[r]);
        C->slot[r] = nullptr;
        if (hipHostMalloc((void**)&C->slot[r], (count + count / 4 + 64) * sizeof(double), hipHostMallocDefault) != hipSuccess) { C->abort(); return -1; }
        C->cap[r] = count + count / 4 + 64;
    }
    if (hipMemcpyAsync(C->slot[r], d_buf, count * sizeof(double), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess) { C->abort(); return -1; }
    if (!C->barrier()) return -1;
    R->sum.resize(count);
    for (size_t i = 0; i < count; ++i) { double s = C->slot[0][i]; for (int q = 1; q < C->world; ++q) s += C->slot[q][i]; R->sum[i] = s; }
    if (!C->barrier()) return -1;                      // nobody overwrites its slot before everyone has read it
    if (hipMemcpyAsync(d_buf, R->sum.data(), count * sizeof(double), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess) { C->abort(); return -1; }       // R->sum is reused by the next call
    return 0;
}

}  // namespace

extern "C" int sfmhip_ba_solve_multi(sfmhip_ctx* const* ctxs, int n_ctx, double* K4, double* ext6, int n_cam, double* pts, int n_pt,
                                     const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_uv, int n_obs,
                                     const sfm_ba_options* opts, sfm_ba_summary* summary)
{
    if (!ctxs || n_ctx < 1 || n_ctx > 64 || !ctxs[0]) return SFMHIP_E_ARG;
    sfmhip_ctx* c0 = ctxs[0];
    if (n_ctx == 1) return sfmhip_ba_solve(c0, K4, ext6, n_cam, pts, n_pt, obs_cam, obs_pt, obs_uv, n_obs, opts, summary);
    SFM_RANGE("sfmhip_ba_solve_multi");
    SFM_ARG_CHECK(c0, K4 && ext6 && n_cam > 0 && n_pt >= 0 && n_obs >= 0 && (pts || n_pt == 0) && ((obs_cam && obs_pt && obs_uv) || n_obs == 0));
    for (int r = 0; r < n_ctx; ++r) SFM_ARG_CHECK(c0, ctxs[r] != nullptr);
    for (int k = 0; k < n_obs; ++k) SFM_ARG_CHECK(c0, obs_cam[k] >= 0 && obs_cam[k] < n_cam && obs_pt[k] >= 0 && obs_pt[k] < n_pt);
    const auto t0 = std::chrono::steady_clock::now();
    // ---- shards: points ordered by the first camera that sees them (counting sort, ties by index), cut into runs of equal observation count
    std::vector<int> first((size_t)n_pt, n_cam), cnt((size_t)n_pt, 0);
    for (int k = 0; k < n_obs; ++k) { first[obs_pt[k]] = std::min(first[obs_pt[k]], (int)obs_cam[k]); ++cnt[obs_pt[k]]; }
    std::vector<int> bucket((size_t)n_cam + 2, 0), order((size_t)n_pt);
    for (int p = 0; p < n_pt; ++p) ++bucket[first[p] + 1];
    for (int c = 0; c <= n_cam; ++c) bucket[c + 1] += bucket[c];
    for (int p = 0; p < n_pt; ++p) order[bucket[first[p]]++] = p;
    std::vector<int> rank_of((size_t)n_pt, 0), local_of((size_t)n_pt, 0), n_local((size_t)n_ctx, 0);
    {
        long long run = 0; int r = 0;
        std::vector<int> tmp_rank((size_t)n_pt);
        for (int i = 0; i < n_pt; ++i) {
            while (r + 1 < n_ctx && run >= (long long)n_obs * (r + 1) / n_ctx) ++r;
            tmp_rank[order[i]] = r; run += cnt[order[i]];
        }
        for (int p = 0; p < n_pt; ++p) { rank_of[p] = tmp_rank[p]; local_of[p] = n_local[tmp_rank[p]]++; }     // local order = the caller's order within the shard
    }
    bool distinct = true;
    for (int a = 0; a < n_ctx; ++a) for (int b = 0; b < a; ++b) if (ctxs[a]->device == ctxs[b]->device) distinct = false;
    const bool use_rccl = distinct && Rccl::get().lib != nullptr;
    char uid[128] = { 0 };
    if (use_rccl && sfmhip_rccl_get_unique_id(uid) != SFMHIP_OK) { c0->last_error = "ncclGetUniqueId failed"; return SFMHIP_E_COMM; }
    LocalComm local(n_ctx);
    std::vector<LocalRank> lranks((size_t)n_ctx);
    std::vector<int> rcs((size_t)n_ctx, SFMHIP_OK);
    std::vector<sfm_ba_summary> sums((size_t)n_ctx);
    std::vector<std::vector<double>> Kr((size_t)n_ctx, std::vector<double>(4)), extr((size_t)n_ctx);
    auto work = [&](int r) {
        sfmhip_ctx* ctx = ctxs[r];
        std::vector<double> pl(3 * (size_t)n_local[r]), uvl; std::vector<int32_t> ocl, opl;
        for (int p = 0; p < n_pt; ++p) if (rank_of[p] == r) for (int d = 0; d < 3; ++d) pl[3 * (size_t)local_of[p] + d] = pts[3 * (size_t)p + d];
        for (int k = 0; k < n_obs; ++k) if (rank_of[obs_pt[k]] == r) { ocl.push_back(obs_cam[k]); opl.push_back(local_of[obs_pt[k]]); uvl.push_back(obs_uv[2 * (size_t)k]); uvl.push_back(obs_uv[2 * (size_t)k + 1]); }
        sfmhip_ba* h = nullptr; void* comm = nullptr;
        int rc = sfmhip_ba_create(ctx, K4, ext6, n_cam, pl.data(), n_local[r], ocl.data(), opl.data(), uvl.data(), (int)ocl.size(), opts, &h);
        if (rc == SFMHIP_OK && use_rccl) rc = sfmhip_rccl_comm_create(ctx, uid, r, n_ctx, &comm);
        if (rc == SFMHIP_OK) {
            if (use_rccl) rc = sfmhip_ba_set_rccl(h, comm, r, n_ctx);
            else { lranks[r].comm = &local; lranks[r].rank = r; rc = sfmhip_ba_set_allreduce(h, local_allreduce_hook, &lranks[r], r, n_ctx); }
        }
        if (rc != SFMHIP_OK) local.abort();             // the others must not wait for this rank
        if (rc == SFMHIP_OK) rc = sfmhip_ba_run(h, &sums[r]);
        if (rc != SFMHIP_OK) local.abort();
        if (rc == SFMHIP_OK) {
            extr[r].resize(6 * (size_t)n_cam);
            rc = sfmhip_ba_get_params(h, Kr[r].data(), extr[r].data(), pl.data());
            if (rc == SFMHIP_OK) for (int p = 0; p < n_pt; ++p) if (rank_of[p] == r) for (int d = 0; d < 3; ++d) pts[3 * (size_t)p + d] = pl[3 * (size_t)local_of[p] + d];
        }
        if (h) sfmhip_ba_destroy(h);
        if (comm) (void)sfmhip_rccl_comm_destroy(comm);
        rcs[r] = rc;
    };
    std::vector<std::thread> th;
    for (int r = 1; r < n_ctx; ++r) th.emplace_back(work, r);
    work(0);
    for (auto& t : th) t.join();
    for (int r = 0; r < n_ctx; ++r) if (rcs[r] != SFMHIP_OK) { if (r) c0->last_error = std::string("rank ") + std::to_string(r) + ": " + ctxs[r]->last_error; return rcs[r]; }
    std::copy(Kr[0].begin(), Kr[0].end(), K4); std::copy(extr[0].begin(), extr[0].end(), ext6);       // replicated: every rank holds the same cameras
    if (summary) {
        *summary = sums[0];
        summary->num_residuals = 2 * n_obs;
        summary->total_time_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return SFMHIP_OK;
}
