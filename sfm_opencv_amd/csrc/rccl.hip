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
