// common.hpp -- internal definitions shared by the translation units of libsfmhip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../../include/sfmhip.h"

struct sfmhip_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;       // the stream every launch goes to (own_stream or an external one)
    std::string last_error;
    // grow-only device scratch (kNN partial results, pair descriptors, rescore lists)
    void*  scratch = nullptr;
    size_t scratch_bytes = 0;
    void*  scratch2 = nullptr;
    size_t scratch2_bytes = 0;
    // grow-only pinned host staging (results the kernels write straight into host memory: sfmhip_match_pairs)
    void*  pinned = nullptr;
    size_t pinned_bytes = 0;
    // pinned staging ring for large uploads from the caller's pageable memory (sfm_upload, context.hip)
    static constexpr size_t STAGE_BYTES = (size_t)16 << 20;
    void*  stage[2] = { nullptr, nullptr };
    hipEvent_t stage_ev[2] = { nullptr, nullptr };
    bool   stage_busy[2] = { false, false };
    int    stage_next = 0;
    struct CopyPool* copy_pool = nullptr;      // three helper threads that fill the staging buffers beside the caller (context.hip)
    // Cache of device blocks (sfm_pool_get / sfm_pool_put): the arrays and the construction temporaries of bundle-adjustment
    // problems.  Giving gigabytes back to the driver costs ~0.1 s that surfaces in whatever HIP call comes next (measured: the
    // second sfmhip_ba_create at C5 took 127 ms against 13 ms for the first), so freed blocks are kept and handed out again;
    // sfmhip_trim / sfmhip_destroy release them.
    struct PoolBlock { void* p; size_t bytes; bool used; };
    std::vector<PoolBlock> pool;
    size_t pool_idle_bytes = 0;
    static constexpr size_t POOL_IDLE_CAP = (size_t)48 << 30;      // idle bytes kept at most (of 288 GB)
    // "every value an integer in [0, 255]" flags of the descriptor sets, one slot each in ONE device array: the preparation
    // kernels raise them, and a batch of sets is resolved with a single copy + sync when a launch first needs to know
    static constexpr int FLAG_SLOTS = 1 << 16;
    int*   d_flagpool = nullptr;
    int    flag_next = 0;
    std::vector<int> flag_free;
    // second stream of the bundle-adjustment problems (created on first use and kept: a hipStreamCreate costs milliseconds)
    hipStream_t aux_stream = nullptr;
    int    num_cus = 256;
    // grow-only host block for the shards' arrays of sfmhip_ba_solve_multi (fresh allocations of that size cost a page fault per 4 KB: 60 ms at C5)
    void*  host_scratch = nullptr; size_t host_scratch_bytes = 0;
    int    inject_alloc_failures = 0;      // sfmhip_debug_fail_allocations: the next N sfm_pool_get calls fail (tests of the error paths)
    // optional per-kernel timing of the matching path (sfmhip_set_kernel_timing): event triples
    // [before kNN kernel, after it, after merge / re-score] for up to TIMING_SLOTS calls since the last query
    static constexpr int TIMING_SLOTS = 64;
    bool   timing = false;
    int    timing_used = 0;
    hipEvent_t tev[TIMING_SLOTS][3] = {};
};

// Every C-ABI entry point binds the calling thread to its context's device for the duration of the call (and restores the
// caller's current device): allocations, event creation and launches then land on ctx->device whatever the caller did
// with hipSetDevice in between, and two contexts on different GPUs can live in one process.
struct SfmDeviceGuard {
    int prev = -1; bool switched = false;
    explicit SfmDeviceGuard(const sfmhip_ctx* c)
    {
        if (c && hipGetDevice(&prev) == hipSuccess && prev != c->device) switched = hipSetDevice(c->device) == hipSuccess;
    }
    ~SfmDeviceGuard() { if (switched) (void)hipSetDevice(prev); }
    SfmDeviceGuard(const SfmDeviceGuard&) = delete;
    SfmDeviceGuard& operator=(const SfmDeviceGuard&) = delete;
};
#define SFM_DEVICE_GUARD(ctx) SfmDeviceGuard _sfm_device_guard(ctx)

// roctx ranges around the C-ABI calls (tracing): visible in `rocprofv3 --marker-trace`.  The marker library is
// looked up at run time (librocprofiler-sdk-roctx, else libroctx64); without it the ranges are no-ops, so libsfmhip.so keeps
// its single dependency on the HIP runtime.
struct SfmRoctx {
    int (*push)(const char*) = nullptr; int (*pop)() = nullptr;
    SfmRoctx();
    static const SfmRoctx& get() { static const SfmRoctx r; return r; }
};
struct SfmRange {
    bool on;
    explicit SfmRange(const char* name) : on(SfmRoctx::get().push != nullptr) { if (on) (void)SfmRoctx::get().push(name); }
    ~SfmRange() { if (on) (void)SfmRoctx::get().pop(); }
    SfmRange(const SfmRange&) = delete;
    SfmRange& operator=(const SfmRange&) = delete;
};
#define SFM_RANGE(name) SfmRange _sfm_range(name)

#define SFM_HIP_TRY(ctx, expr)                                                                   \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);               \
            return SFMHIP_E_HIP;                                                                 \
        }                                                                                        \
    } while (0)

#define SFM_ARG_CHECK(ctx, cond)                                                                 \
    do {                                                                                         \
        if (!(cond)) {                                                                           \
            if (ctx) (ctx)->last_error = std::string("bad argument: ") + #cond;                  \
            return SFMHIP_E_ARG;                                                                 \
        }                                                                                        \
    } while (0)

static inline int sfm_scratch(sfmhip_ctx* ctx, size_t bytes, void** out)
{
    if (bytes > ctx->scratch_bytes) {
        // the stream may still be using the old block: drain it before freeing
        SFM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->scratch) SFM_HIP_TRY(ctx, hipFree(ctx->scratch));
        ctx->scratch = nullptr; ctx->scratch_bytes = 0;
        size_t want = bytes + bytes / 4 + 4096;
        SFM_HIP_TRY(ctx, hipMalloc(&ctx->scratch, want));
        ctx->scratch_bytes = want;
    }
    *out = ctx->scratch;
    return SFMHIP_OK;
}
static inline int sfm_scratch2(sfmhip_ctx* ctx, size_t bytes, void** out)
{
    if (bytes > ctx->scratch2_bytes) {
        SFM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->scratch2) SFM_HIP_TRY(ctx, hipFree(ctx->scratch2));
        ctx->scratch2 = nullptr; ctx->scratch2_bytes = 0;
        size_t want = bytes + bytes / 4 + 4096;
        SFM_HIP_TRY(ctx, hipMalloc(&ctx->scratch2, want));
        ctx->scratch2_bytes = want;
    }
    *out = ctx->scratch2;
    return SFMHIP_OK;
}

static inline int sfm_pinned(sfmhip_ctx* ctx, size_t bytes, void** out)
{
    if (bytes > ctx->pinned_bytes) {
        SFM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->pinned) SFM_HIP_TRY(ctx, hipHostFree(ctx->pinned));
        ctx->pinned = nullptr; ctx->pinned_bytes = 0;
        size_t want = bytes + bytes / 4 + 4096;
        SFM_HIP_TRY(ctx, hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault));
        ctx->pinned_bytes = want;
    }
    *out = ctx->pinned;
    return SFMHIP_OK;
}

// Host -> HBM copy of a caller's (pageable) array, ordered on the context's stream.  hipMemcpyAsync from pageable memory runs at
// 9-13 GB/s on the MI355X boxes (one runtime thread staging); this goes through two pinned 16 MB buffers filled by four host
// threads instead (43 GB/s, experiments/h2d_bench.hip).  The source has been consumed when the call returns.
int sfm_upload(sfmhip_ctx* ctx, void* dst, const void* src, size_t bytes);
// the same, but every pinned piece is PRODUCED on the copy threads on its way into the staging ring (a conversion, a gather of strided
// rows): fill(piece, off, n, t, nt) writes thread t's share of the bytes [off, off + n) of the transfer to piece[0 .. n); pieces are
// multiples of `granule` bytes except the last.  Everything fill reads has been consumed when the call returns.
int sfm_upload_produced(sfmhip_ctx* ctx, void* dst, size_t bytes, size_t granule, const std::function<void(char*, size_t, size_t, int, int)>& fill);
// f(t, nt) on every copy thread of the context, the caller included
void sfm_parallel(sfmhip_ctx* ctx, const std::function<void(int, int)>& f);

// device block of at least `bytes` from the context's cache (an idle block of up to 4x the size, else a new hipMalloc); stream-ordered
// reuse: every user of these blocks works on the context's stream
int  sfm_pool_get(sfmhip_ctx* ctx, size_t bytes, void** out);
void sfm_pool_put(sfmhip_ctx* ctx, void* p);
void sfm_pool_trim(sfmhip_ctx* ctx);
// rccl.hip: communicators cached for sets of contexts go when one of their contexts does
void sfm_rccl_forget_ctx(sfmhip_ctx* ctx);

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
static inline int ceil_div(int x, int m) { return (x + m - 1) / m; }

// descriptor set (one image), see match.hip
struct sfmhip_descset {
    sfmhip_ctx* ctx = nullptr;
    int kind = 0;
    int rows = 0, rows_pad = 0;      // rows_pad: multiple of 128
    int dim = 0;                      // L2: elements; Hamming: bytes
    // L2
    const float* d_f32 = nullptr; size_t ld = 0; bool owns_f32 = false;
    int dim_pad = 0;                  // multiple of 32 (int8 copy row length in bytes)
    int8_t* d_i8 = nullptr;           // rows_pad x dim_pad, value - 128; pad rows zero
    int32_t* d_norm = nullptr;        // 2 x rows_pad: [sum b^2 | sum b^2 + 2 sum b], b = value - 128; pad rows = PAD_NORM
    int exact_u8 = 0;                 // every value an integer in [0,255] and dim <= 128 (valid once !exact_pending)
    bool exact_pending = false;       // the preparation kernel's verdict is still in d_flag (descsets_resolve reads it)
    int flag_slot = -1;               // d_flag = ctx->d_flagpool + flag_slot; -1: a block of its own
    // Hamming2
    uint32_t* d_u32 = nullptr;        // rows_pad x 16 words (64 B rows, zero padded)
    uint32_t* d_f4 = nullptr;         // rows_pad x 96 words: 768 FP4 values per row (prep_hamming_fp4_kernel), nbytes <= 61 only
    int* d_flag = nullptr;
};
