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
#include <algorithm>
#include <chrono>
#include <dlfcn.h>
#include <memory>
#include <sched.h>

namespace {

// the handful of RCCL declarations used (rccl.h: ncclUniqueId is 128 opaque bytes, ncclDouble = 8, ncclSum = 0, ncclSuccess = 0)
struct RcclUniqueId { char internal[128]; };
typedef int (*GetUniqueIdFn)(RcclUniqueId*);
typedef int (*CommInitRankFn)(void** comm, int nranks, RcclUniqueId id, int rank);
typedef int (*CommDestroyFn)(void* comm);
typedef int (*AllReduceFn)(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t stream);
typedef const char* (*GetErrorStringFn)(int);
typedef int (*CommInitAllFn)(void** comms, int ndev, const int* devlist);
typedef int (*CommAbortFn)(void* comm);

struct Rccl {
    void* lib = nullptr;
    GetUniqueIdFn get_unique_id = nullptr; CommInitRankFn comm_init_rank = nullptr; CommDestroyFn comm_destroy = nullptr;
    AllReduceFn all_reduce = nullptr; GetErrorStringFn error_string = nullptr; CommInitAllFn comm_init_all = nullptr; CommAbortFn comm_abort = nullptr;
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
        comm_init_all = (CommInitAllFn)dlsym(lib, "ncclCommInitAll");
        comm_abort = (CommAbortFn)dlsym(lib, "ncclCommAbort");
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
// Several GPUs of ONE process ("multi-GPU fan-out is internal"; the reference's main() is one process,
// NView:1334-1524).  One host thread per context, every context on its own PCIe link:
//   sfmhip_match_pairs_multi   image pairs in contiguous blocks, a block's images uploaded to its context only (a chain: a block +
//                              one halo image), no exchange at all; lists land in the caller's buffer in pair order;
//   sfmhip_ba_solve_multi      points sharded by the first camera that sees them, cameras replicated, the packed reduced-system
//                              message summed by RCCL -- or, where two contexts share a device (the one-card rehearsal of the tests)
//                              or librccl is missing, by a host-staged exchange inside the process.
// What a first run on a real node must survive (round-3 review): communicators are created ONCE per set of contexts
// (ncclCommInitAll: one call, nobody waits for a rank that failed) and kept until one of the contexts goes; every rank's status
// is agreed through an in-process barrier BEFORE anyone enters a collective, so a shard that fails to build is an error code on
// all ranks, not a hang; the LM loop gives a collective 60 s before it reports SFMHIP_E_COMM (ba.hip); the sharding itself is a
// few parallel passes over the observations.
// ------------------------------------------------------------------------------------------------
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

namespace {

// ranks of one call meet here: barrier() / agree(rc) (everybody leaves with the worst status)
struct Meeting {
    int world = 1;
    std::mutex mu; std::condition_variable cv; int arrived = 0; unsigned long long gen = 0; bool broken = false;
    int worst_now = 0, worst_last = 0;
    explicit Meeting(int w) : world(w) {}
    bool barrier() { int rc = 0; return agree(rc) && rc == 0; }
    // returns false when the meeting was aborted; rc becomes the first non-zero status deposited in this round (0: none)
    bool agree(int& rc)
    {
        std::unique_lock<std::mutex> lk(mu);
        if (broken) return false;
        if (rc != 0 && worst_now == 0) worst_now = rc;
        const unsigned long long g = gen;
        if (++arrived == world) { arrived = 0; worst_last = worst_now; worst_now = 0; ++gen; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g || broken; });
        if (broken) return false;
        rc = worst_last;
        return true;
    }
    void abort() { std::lock_guard<std::mutex> lk(mu); broken = true; cv.notify_all(); }
};

// Sum of the ranks' device buffers through pinned host memory: every rank copies out, all wait, every rank adds the N copies in
// rank order (so all ranks hold bit-identical sums) and copies back.  A fallback and a test vehicle, not the production path.
struct LocalComm {
    Meeting meet;
    std::vector<double*> slot; std::vector<size_t> cap;
    explicit LocalComm(int w) : meet(w), slot((size_t)w, nullptr), cap((size_t)w, 0) {}
    ~LocalComm() { for (double* p : slot) if (p) (void)hipHostFree(p); }
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
        if (hipHostMalloc((void**)&C->slot[r], (count + count / 4 + 64) * sizeof(double), hipHostMallocDefault) != hipSuccess) { C->meet.abort(); return -1; }
        C->cap[r] = count + count / 4 + 64;
    }
    if (hipMemcpyAsync(C->slot[r], d_buf, count * sizeof(double), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess) { C->meet.abort(); return -1; }
    if (!C->meet.barrier()) return -1;
    R->sum.resize(count);
    for (size_t i = 0; i < count; ++i) { double s = C->slot[0][i]; for (int q = 1; q < C->meet.world; ++q) s += C->slot[q][i]; R->sum[i] = s; }
    if (!C->meet.barrier()) return -1;                      // nobody overwrites its slot before everyone has read it
    if (hipMemcpyAsync(d_buf, R->sum.data(), count * sizeof(double), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess) { C->meet.abort(); return -1; }       // R->sum is reused by the next call
    return 0;
}

// RCCL communicators of a set of contexts (order matters: rank r = ctxs[r]), created once and kept
struct CommSet { std::vector<sfmhip_ctx*> ctxs; std::vector<void*> comms; };
std::mutex g_comm_mu;
std::vector<std::unique_ptr<CommSet>> g_comm_sets;

// the cached communicators of these contexts, created on first use (nullptr: RCCL missing / failed, last_error of ctxs[0] says why)
const CommSet* comm_set_get(sfmhip_ctx* const* ctxs, int n)
{
    const Rccl& R = Rccl::get();
    if (!R.lib) return nullptr;
    std::lock_guard<std::mutex> lk(g_comm_mu);
    for (const auto& pcs : g_comm_sets)
        if ((int)pcs->ctxs.size() == n && std::equal(pcs->ctxs.begin(), pcs->ctxs.end(), ctxs)) return pcs.get();
    CommSet cs;
    cs.ctxs.assign(ctxs, ctxs + n); cs.comms.assign((size_t)n, nullptr);
    std::vector<int> devs((size_t)n);
    for (int r = 0; r < n; ++r) devs[r] = ctxs[r]->device;
    int prev = 0; (void)hipGetDevice(&prev);
    int rc = 0;
    if (R.comm_init_all) rc = R.comm_init_all(cs.comms.data(), n, devs.data());       // one call for the whole set: no rank can be left waiting
    else {
        RcclUniqueId id;
        rc = R.get_unique_id(&id);
        std::vector<std::thread> th; std::vector<int> rcs((size_t)n, 0);
        if (rc == 0) {
            for (int r = 0; r < n; ++r) th.emplace_back([&, r] { (void)hipSetDevice(devs[r]); rcs[r] = R.comm_init_rank(&cs.comms[r], n, id, r); });
            for (auto& t : th) t.join();
            for (int r = 0; r < n; ++r) if (rcs[r] != 0) rc = rcs[r];
        }
    }
    (void)hipSetDevice(prev);
    if (rc != 0) {
        for (void* c : cs.comms) if (c) (void)(R.comm_abort ? R.comm_abort(c) : R.comm_destroy(c));
        ctxs[0]->last_error = std::string("RCCL communicator set-up: ") + (R.error_string ? R.error_string(rc) : "failed");
        return nullptr;
    }
    g_comm_sets.push_back(std::unique_ptr<CommSet>(new CommSet(std::move(cs))));
    return g_comm_sets.back().get();
}
// a collective went wrong (or timed out) on this set: its communicators cannot be trusted any more
void comm_set_drop(sfmhip_ctx* const* ctxs, int n, bool abort)
{
    const Rccl& R = Rccl::get();
    std::lock_guard<std::mutex> lk(g_comm_mu);
    for (size_t k = 0; k < g_comm_sets.size(); ++k) {
        CommSet& cs = *g_comm_sets[k];
        if ((int)cs.ctxs.size() != n || !std::equal(cs.ctxs.begin(), cs.ctxs.end(), ctxs)) continue;
        for (void* c : cs.comms) if (c && R.lib) (void)((abort && R.comm_abort) ? R.comm_abort(c) : R.comm_destroy(c));
        g_comm_sets.erase(g_comm_sets.begin() + (long)k);
        return;
    }
}

int host_threads()
{
    int cores = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) cores = CPU_COUNT(&set);
    return std::min(16, std::max(1, cores));
}
// f(t, nt) on nt threads (the caller is thread 0)
template <typename F>
void parallel_do(int nt, F f)
{
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back([&f, t, nt] { f(t, nt); });
    f(0, nt);
    for (auto& x : th) x.join();
}

}  // namespace

// called by sfmhip_destroy: communicators that include this context go with it
void sfm_rccl_forget_ctx(sfmhip_ctx* ctx)
{
    const Rccl& R = Rccl::get();
    std::lock_guard<std::mutex> lk(g_comm_mu);
    for (size_t k = 0; k < g_comm_sets.size();) {
        CommSet& cs = *g_comm_sets[k];
        if (std::find(cs.ctxs.begin(), cs.ctxs.end(), ctx) == cs.ctxs.end()) { ++k; continue; }
        for (void* c : cs.comms) if (c && R.lib) (void)R.comm_destroy(c);
        g_comm_sets.erase(g_comm_sets.begin() + (long)k);
    }
}

extern "C" int sfmhip_match_pairs_multi(sfmhip_ctx* const* ctxs, int n_ctx, int kind, const void* const* desc, const int32_t* rows, int dim,
                                        const size_t* ld, int n_images, const int32_t* pairs, int n_pairs,
                                        double ratio, float floor_, float mult, sfm_dmatch* matches, int max_per_pair, int32_t* counts)
{
    if (!ctxs || n_ctx < 1 || n_ctx > 64 || !ctxs[0]) return SFMHIP_E_ARG;
    sfmhip_ctx* c0 = ctxs[0];
    SFM_RANGE("sfmhip_match_pairs_multi");
    SFM_ARG_CHECK(c0, (kind == SFMHIP_DESC_L2_F32 || kind == SFMHIP_DESC_HAMMING2_U8) && n_images >= 0 && n_pairs >= 0 && dim > 0 && max_per_pair >= 0);
    SFM_ARG_CHECK(c0, (n_images == 0 || (desc && rows)) && (n_pairs == 0 || (pairs && counts && (matches || max_per_pair == 0))));
    for (int r = 0; r < n_ctx; ++r) SFM_ARG_CHECK(c0, ctxs[r] != nullptr);
    for (int p = 0; p < 2 * n_pairs; ++p) SFM_ARG_CHECK(c0, pairs[p] >= 0 && pairs[p] < n_images);
    if (n_pairs == 0) return SFMHIP_OK;
    const int nw = std::min(n_ctx, n_pairs);
    std::vector<int> rcs((size_t)nw, SFMHIP_OK);
    auto work = [&](int r) {
        sfmhip_ctx* ctx = ctxs[r];
        // this context's block of pairs, the images it needs (in index order: a chain block + its halo image), their new numbers
        const int p0 = (int)((long long)n_pairs * r / nw), p1 = (int)((long long)n_pairs * (r + 1) / nw);
        std::vector<int> img;
        for (int p = 2 * p0; p < 2 * p1; ++p) img.push_back(pairs[p]);
        std::sort(img.begin(), img.end()); img.erase(std::unique(img.begin(), img.end()), img.end());
        std::vector<int32_t> lp((size_t)(2 * (p1 - p0)));
        for (int p = 2 * p0; p < 2 * p1; ++p) lp[(size_t)(p - 2 * p0)] = (int32_t)(std::lower_bound(img.begin(), img.end(), pairs[p]) - img.begin());
        const int ni = (int)img.size();
        std::vector<const void*> dp((size_t)ni); std::vector<int32_t> rw((size_t)ni); std::vector<size_t> lds((size_t)ni);
        for (int i = 0; i < ni; ++i) { dp[i] = desc[img[i]]; rw[i] = rows[img[i]]; lds[i] = ld ? ld[img[i]] : (size_t)dim; }
        std::vector<sfmhip_descset*> sets((size_t)ni, nullptr);
        int rc = kind == SFMHIP_DESC_L2_F32 ? sfmhip_descsets_create_l2_host(ctx, (const float* const*)dp.data(), rw.data(), dim, lds.data(), ni, sets.data())
                                            : sfmhip_descsets_create_hamming2_host(ctx, (const uint8_t* const*)dp.data(), rw.data(), dim, lds.data(), ni, sets.data());
        if (rc == SFMHIP_OK)
            rc = sfmhip_match_pairs(ctx, sets.data(), ni, lp.data(), p1 - p0, ratio, floor_, mult, matches + (size_t)p0 * max_per_pair, max_per_pair, counts + p0);
        for (sfmhip_descset* s : sets) if (s) sfmhip_descset_destroy(s);
        rcs[r] = rc;
    };
    std::vector<std::thread> th;
    for (int r = 1; r < nw; ++r) th.emplace_back(work, r);
    work(0);
    for (auto& t : th) t.join();
    for (int r = 0; r < nw; ++r)
        if (rcs[r] != SFMHIP_OK) { if (r) c0->last_error = std::string("context ") + std::to_string(r) + ": " + ctxs[r]->last_error; return rcs[r]; }
    return SFMHIP_OK;
}

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
    const auto t0 = std::chrono::steady_clock::now();
    // ---- shards.  A point belongs to the rank of the FIRST camera that sees it; the cameras are cut into n_ctx consecutive ranges of
    // (nearly) equal observation count.  A few parallel passes over the observations / points on the context's copy threads (persistent:
    // spawning sixteen threads per pass cost more than the passes), no sort; the big arrays live in one grow-only block of the first
    // context (allocated afresh they cost a page fault per 4 KB, 60 ms at C5, and a zero fill on top).
    int NT = 1;
    sfm_parallel(c0, [&](int t, int nt) { if (t == 0) NT = nt; });
    static_assert(sizeof(std::atomic<int>) == sizeof(int), "atomic<int> is an int");
    auto align64 = [](size_t x) { return (x + 63) & ~(size_t)63; };
    const size_t fixed_bytes = 3 * align64((size_t)n_pt * sizeof(int));             // first, rank_of, local_of
    const size_t shard_bytes = align64(3 * (size_t)n_pt * sizeof(double)) + align64(2 * (size_t)n_obs * sizeof(double)) + 2 * align64((size_t)n_obs * sizeof(int32_t)) + 64 * 4 * (size_t)n_ctx;
    if (c0->host_scratch_bytes < fixed_bytes + shard_bytes) {
        free(c0->host_scratch);
        c0->host_scratch_bytes = fixed_bytes + shard_bytes + (fixed_bytes + shard_bytes) / 8;
        c0->host_scratch = malloc(c0->host_scratch_bytes);
        if (!c0->host_scratch) { c0->host_scratch_bytes = 0; c0->last_error = "out of host memory"; return SFMHIP_E_ARG; }
    }
    char* arena = (char*)c0->host_scratch;
    std::atomic<int>* first = (std::atomic<int>*)arena; arena += align64((size_t)n_pt * sizeof(int));
    int* rank_of = (int*)arena; arena += align64((size_t)n_pt * sizeof(int));
    int* local_of = (int*)arena; arena += align64((size_t)n_pt * sizeof(int));
    std::atomic<int> bad_obs{ 0 };
    //  1. the first camera of every point (atomic min: after the first few observations of a point a plain load settles it), validation
    sfm_parallel(c0, [&](int t, int nt) {
        for (long long p = (long long)n_pt * t / nt, e = (long long)n_pt * (t + 1) / nt; p < e; ++p) first[p].store(n_cam, std::memory_order_relaxed);
    });
    sfm_parallel(c0, [&](int t, int nt) {
        for (long long k = (long long)n_obs * t / nt, e = (long long)n_obs * (t + 1) / nt; k < e; ++k) {
            const int c = obs_cam[k], p = obs_pt[k];
            if (c < 0 || c >= n_cam || p < 0 || p >= n_pt) { bad_obs.store(1, std::memory_order_relaxed); continue; }
            int cur = first[p].load(std::memory_order_relaxed);
            while (c < cur && !first[p].compare_exchange_weak(cur, c, std::memory_order_relaxed)) { }
        }
    });
    SFM_ARG_CHECK(c0, bad_obs.load() == 0);
    //  2. observations per (thread, first camera) -> camera ranges of the ranks, and where every thread's observations of a rank start
    std::vector<std::vector<int>> hist((size_t)NT, std::vector<int>((size_t)n_cam + 1, 0));
    sfm_parallel(c0, [&](int t, int nt) {
        int* hh = hist[t].data();
        for (long long k = (long long)n_obs * t / nt, e = (long long)n_obs * (t + 1) / nt; k < e; ++k) ++hh[first[obs_pt[k]].load(std::memory_order_relaxed)];
    });
    std::vector<int> rank_of_cam((size_t)n_cam + 1, n_ctx - 1);
    {
        long long run = 0; int r = 0;
        for (int c = 0; c < n_cam; ++c) {
            while (r + 1 < n_ctx && run >= (long long)n_obs * (r + 1) / n_ctx) ++r;
            rank_of_cam[c] = r;
            for (int t = 0; t < NT; ++t) run += hist[t][c];
        }
    }
    std::vector<std::vector<long long>> pcount((size_t)NT, std::vector<long long>((size_t)n_ctx, 0)), ocount = pcount;
    for (int t = 0; t < NT; ++t) for (int c = 0; c < n_cam; ++c) ocount[t][rank_of_cam[c]] += hist[t][c];
    //  3. rank of every point, points per (thread, rank)
    sfm_parallel(c0, [&](int t, int nt) {
        for (long long p = (long long)n_pt * t / nt, e = (long long)n_pt * (t + 1) / nt; p < e; ++p) { const int r = rank_of_cam[first[p].load(std::memory_order_relaxed)]; rank_of[p] = r; ++pcount[t][r]; }
    });
    std::vector<long long> n_local((size_t)n_ctx, 0), n_lobs((size_t)n_ctx, 0);
    for (int r = 0; r < n_ctx; ++r)
        for (int t = 0; t < NT; ++t) { const long long a = pcount[t][r], b = ocount[t][r]; pcount[t][r] = n_local[r]; ocount[t][r] = n_lobs[r]; n_local[r] += a; n_lobs[r] += b; }
    //  4. scatter: points (their local index = the caller's order inside the shard) and observations into the shards' arrays, order kept
    std::vector<double*> pl((size_t)n_ctx), uvl((size_t)n_ctx);
    std::vector<int32_t*> ocl((size_t)n_ctx), opl((size_t)n_ctx);
    for (int r = 0; r < n_ctx; ++r) {
        pl[r] = (double*)arena; arena += align64(3 * (size_t)n_local[r] * sizeof(double));
        uvl[r] = (double*)arena; arena += align64(2 * (size_t)n_lobs[r] * sizeof(double));
        ocl[r] = (int32_t*)arena; arena += align64((size_t)n_lobs[r] * sizeof(int32_t));
        opl[r] = (int32_t*)arena; arena += align64((size_t)n_lobs[r] * sizeof(int32_t));
    }
    sfm_parallel(c0, [&](int t, int nt) {
        std::vector<long long> at = pcount[t];
        for (long long p = (long long)n_pt * t / nt, e = (long long)n_pt * (t + 1) / nt; p < e; ++p) {
            const int r = rank_of[p]; const long long l = at[r]++;
            local_of[p] = (int)l;
            for (int d = 0; d < 3; ++d) pl[r][3 * (size_t)l + d] = pts[3 * (size_t)p + d];
        }
    });
    sfm_parallel(c0, [&](int t, int nt) {
        std::vector<long long> at = ocount[t];
        for (long long k = (long long)n_obs * t / nt, e = (long long)n_obs * (t + 1) / nt; k < e; ++k) {
            const int p = obs_pt[k], r = rank_of[p]; const long long l = at[r]++;
            ocl[r][(size_t)l] = obs_cam[k]; opl[r][(size_t)l] = local_of[p];
            uvl[r][2 * (size_t)l] = obs_uv[2 * (size_t)k]; uvl[r][2 * (size_t)l + 1] = obs_uv[2 * (size_t)k + 1];
        }
    });
    const double shard_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    bool distinct = true;
    for (int a = 0; a < n_ctx; ++a) for (int b = 0; b < a; ++b) if (ctxs[a]->device == ctxs[b]->device) distinct = false;
    const CommSet* cset = distinct ? comm_set_get(ctxs, n_ctx) : nullptr;       // cached: created by the first call on this set of contexts
    const bool use_rccl = cset != nullptr;
    LocalComm local(n_ctx);
    std::vector<LocalRank> lranks((size_t)n_ctx);
    std::vector<int> rcs((size_t)n_ctx, SFMHIP_OK);
    std::vector<sfm_ba_summary> sums((size_t)n_ctx);
    std::vector<std::vector<double>> Kr((size_t)n_ctx, std::vector<double>(4)), extr((size_t)n_ctx);
    std::vector<double> create_s((size_t)n_ctx, 0.0);
    auto work = [&](int r) {
        sfmhip_ctx* ctx = ctxs[r];
        sfmhip_ba* h = nullptr;
        const auto tc = std::chrono::steady_clock::now();
        int rc = sfmhip_ba_create(ctx, K4, ext6, n_cam, pl[r], (int)n_local[r], ocl[r], opl[r], uvl[r], (int)n_lobs[r], opts, &h);
        if (rc == SFMHIP_OK) {
            if (use_rccl) rc = sfmhip_ba_set_rccl(h, cset->comms[r], r, n_ctx);
            else { lranks[r].comm = &local; lranks[r].rank = r; rc = sfmhip_ba_set_allreduce(h, local_allreduce_hook, &lranks[r], r, n_ctx); }
        }
        create_s[r] = std::chrono::duration<double>(std::chrono::steady_clock::now() - tc).count();
        // every rank's shard is built (or not) before ANYONE enters a collective: a failed shard is an error on all ranks, not a hang
        int agreed = rc;
        const bool met = local.meet.agree(agreed);
        if (rc == SFMHIP_OK && (!met || agreed != SFMHIP_OK)) { rc = SFMHIP_E_COMM; ctx->last_error = "another rank's shard failed"; }
        if (rc == SFMHIP_OK) {
            rc = sfmhip_ba_run(h, &sums[r]);
            if (rc != SFMHIP_OK) local.meet.abort();         // (host-staged exchange: the others leave their barrier; RCCL: their LM loops time out)
        }
        if (rc == SFMHIP_OK) {
            extr[r].resize(6 * (size_t)n_cam);
            rc = sfmhip_ba_get_params(h, Kr[r].data(), extr[r].data(), pl[r]);
        }
        if (h) sfmhip_ba_destroy(h);
        rcs[r] = rc;
    };
    std::vector<std::thread> th;
    for (int r = 1; r < n_ctx; ++r) th.emplace_back(work, r);
    work(0);
    for (auto& t : th) t.join();
    int first_bad = -1;
    for (int r = 0; r < n_ctx; ++r) if (rcs[r] != SFMHIP_OK && (first_bad < 0 || (rcs[first_bad] == SFMHIP_E_COMM && rcs[r] != SFMHIP_E_COMM))) first_bad = r;     // the rank that failed, not the ones it took along
    if (first_bad >= 0) {
        if (use_rccl) comm_set_drop(ctxs, n_ctx, true);      // a collective may be stuck half way: these communicators are not used again
        if (first_bad) c0->last_error = std::string("rank ") + std::to_string(first_bad) + ": " + ctxs[first_bad]->last_error;
        return rcs[first_bad];
    }
    sfm_parallel(c0, [&](int t, int nt) {
        for (long long p = (long long)n_pt * t / nt, e = (long long)n_pt * (t + 1) / nt; p < e; ++p) {
            const int r = rank_of[p]; const size_t l = (size_t)local_of[p];
            for (int d = 0; d < 3; ++d) pts[3 * (size_t)p + d] = pl[r][3 * l + d];
        }
    });
    std::copy(Kr[0].begin(), Kr[0].end(), K4); std::copy(extr[0].begin(), extr[0].end(), ext6);       // replicated: every rank holds the same cameras
    if (opts && opts->verbose) {
        double lo = 1e30, hi = 0.0; for (double v : create_s) { lo = std::min(lo, v); hi = std::max(hi, v); }
        printf("[sfmhip_ba_solve_multi] %d contexts (%s), %d host threads: sharding %.2f ms, shard construction %.2f .. %.2f ms, whole call %.2f ms\n", n_ctx,
               use_rccl ? "RCCL" : "host-staged exchange", NT, 1e3 * shard_s, 1e3 * lo, 1e3 * hi, 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    if (summary) {
        *summary = sums[0];
        summary->num_residuals = 2 * n_obs;
        summary->total_time_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        // like sfmhip_ba_solve: everything before the first LM iteration (sharding on the host threads + the slowest shard's construction + plan)
        double slowest = 0.0; for (double v : create_s) slowest = std::max(slowest, v);
        summary->preprocessor_time_s = shard_s + slowest + sums[0].preprocessor_time_s;
        summary->minimizer_time_s = sums[0].minimizer_time_s;
        summary->postprocessor_time_s = std::max(0.0, summary->total_time_s - summary->preprocessor_time_s - summary->minimizer_time_s);
    }
    return SFMHIP_OK;
}
