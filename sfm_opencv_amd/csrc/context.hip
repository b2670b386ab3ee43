// context.hip -- context lifetime + the host-only ratio tail of libsfmhip.so.
#include "common.hpp"
#include <cfloat>
#include <dlfcn.h>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <sched.h>

SfmRoctx::SfmRoctx()
{
    for (const char* lib : { "librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so" }) {
        void* h = dlopen(lib, RTLD_LAZY | RTLD_GLOBAL);
        if (!h) continue;
        push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
        pop = (int (*)())dlsym(h, "roctxRangePop");
        if (push && pop) return;
        push = nullptr; pop = nullptr;
    }
}

// Helper threads of sfm_upload / sfm_upload_produced: they sleep on a condition variable and each runs its share of the current job.
// As many as the cores this process may use, between 4 and 16 (the caller is one of them): a memcpy into the staging ring reaches
// 43 GB/s with four, and a float -> byte conversion on the way in is bound by the DRAM read rate of as many as there are.
struct CopyPool {
    int NT;
    std::vector<std::thread> th;
    std::mutex mu; std::condition_variable cv_go, cv_done;
    unsigned long long gen = 0; int pending = 0; bool quit = false;
    const std::function<void(int, int)>* job = nullptr;
    static int pick_threads()
    {
        int cores = (int)std::thread::hardware_concurrency();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) cores = CPU_COUNT(&set);
        return std::min(16, std::max(4, cores));
    }
    CopyPool() : NT(pick_threads())
    {
        for (int t = 1; t < NT; ++t)
            th.emplace_back([this, t] {
                unsigned long long seen = 0;
                for (;;) {
                    std::unique_lock<std::mutex> lk(mu);
                    cv_go.wait(lk, [&] { return quit || gen != seen; });
                    if (quit) return;
                    seen = gen;
                    const std::function<void(int, int)>* f = job;
                    lk.unlock();
                    (*f)(t, NT);
                    lk.lock();
                    if (--pending == 0) cv_done.notify_one();
                }
            });
    }
    ~CopyPool()
    {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv_go.notify_all();
        for (auto& t : th) t.join();
    }
    void run(const std::function<void(int, int)>& f)
    {
        { std::lock_guard<std::mutex> lk(mu); job = &f; pending = NT - 1; ++gen; }
        cv_go.notify_all();
        f(0, NT);
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
    }
};

void sfm_parallel(sfmhip_ctx* ctx, const std::function<void(int, int)>& f)
{
    if (!ctx->copy_pool) ctx->copy_pool = new CopyPool();
    ctx->copy_pool->run(f);
}

static int stage_ring(sfmhip_ctx* ctx)
{
    for (int b = 0; b < 2; ++b)
        if (!ctx->stage[b]) {
            SFM_HIP_TRY(ctx, hipHostMalloc(&ctx->stage[b], sfmhip_ctx::STAGE_BYTES, hipHostMallocDefault));
            SFM_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->stage_ev[b], hipEventDisableTiming));
        }
    return SFMHIP_OK;
}

int sfm_upload_produced(sfmhip_ctx* ctx, void* dst, size_t bytes, size_t granule, const std::function<void(char*, size_t, size_t, int, int)>& fill)
{
    if (bytes == 0) return SFMHIP_OK;
    { const int rc = stage_ring(ctx); if (rc) return rc; }
    const size_t CH = granule > 0 && granule <= sfmhip_ctx::STAGE_BYTES ? sfmhip_ctx::STAGE_BYTES / granule * granule : sfmhip_ctx::STAGE_BYTES;
    for (size_t off = 0; off < bytes; off += CH) {
        const size_t n = std::min(CH, bytes - off);
        const int b = ctx->stage_next;
        if (ctx->stage_busy[b]) SFM_HIP_TRY(ctx, hipEventSynchronize(ctx->stage_ev[b]));
        char* d = (char*)ctx->stage[b];
        if (n >= ((size_t)256 << 10)) sfm_parallel(ctx, [&](int t, int nt) { fill(d, off, n, t, nt); });
        else fill(d, off, n, 0, 1);
        SFM_HIP_TRY(ctx, hipMemcpyAsync((char*)dst + off, d, n, hipMemcpyHostToDevice, ctx->stream));
        SFM_HIP_TRY(ctx, hipEventRecord(ctx->stage_ev[b], ctx->stream));
        ctx->stage_busy[b] = true; ctx->stage_next = b ^ 1;
    }
    return SFMHIP_OK;
}

int sfm_upload(sfmhip_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    return sfm_upload_produced(ctx, dst, bytes, 64, [src](char* piece, size_t off, size_t n, int t, int nt) {
        const size_t a = n * t / nt, e = n * (t + 1) / nt;
        memcpy(piece + a, (const char*)src + off + a, e - a);
    });
}

int sfm_pool_get(sfmhip_ctx* ctx, size_t bytes, void** out)
{
    if (ctx->inject_alloc_failures > 0) { --ctx->inject_alloc_failures; ctx->last_error = "injected allocation failure (sfmhip_debug_fail_allocations)"; return SFMHIP_E_HIP; }
    if (bytes == 0) bytes = 256;
    int best = -1;
    const size_t hi = std::max(4 * bytes, (size_t)1 << 20);
    for (size_t i = 0; i < ctx->pool.size(); ++i) {
        const auto& b = ctx->pool[i];
        if (!b.used && b.bytes >= bytes && b.bytes <= hi && (best < 0 || b.bytes < ctx->pool[best].bytes)) best = (int)i;
    }
    if (best >= 0) {
        ctx->pool[best].used = true; ctx->pool_idle_bytes -= ctx->pool[best].bytes;
        *out = ctx->pool[best].p;
        return SFMHIP_OK;
    }
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, bytes);
    if (e == hipErrorOutOfMemory && ctx->pool_idle_bytes > 0) {
        // idle blocks of other sizes may be all that stands between this request and success (they are only handed out for requests of
        // 1x .. 4x their size): give them back to the driver and try once more
        (void)hipGetLastError();
        sfm_pool_trim(ctx);
        e = hipMalloc(&q, bytes);
    }
    if (e != hipSuccess) { (void)hipGetLastError(); ctx->last_error = std::string("hipMalloc(") + std::to_string(bytes) + " bytes): " + hipGetErrorString(e); return SFMHIP_E_HIP; }
    ctx->pool.push_back({ q, bytes, true });
    *out = q;
    return SFMHIP_OK;
}

void sfm_pool_put(sfmhip_ctx* ctx, void* p)
{
    for (size_t i = 0; i < ctx->pool.size(); ++i) {
        auto& b = ctx->pool[i];
        if (b.p != p) continue;
        if (ctx->pool_idle_bytes + b.bytes > sfmhip_ctx::POOL_IDLE_CAP) {
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipFree(b.p);
            ctx->pool.erase(ctx->pool.begin() + i);
        } else { b.used = false; ctx->pool_idle_bytes += b.bytes; }
        return;
    }
}

void sfm_pool_trim(sfmhip_ctx* ctx)
{
    (void)hipStreamSynchronize(ctx->stream);
    for (size_t i = 0; i < ctx->pool.size();) {
        if (!ctx->pool[i].used) { (void)hipFree(ctx->pool[i].p); ctx->pool_idle_bytes -= ctx->pool[i].bytes; ctx->pool.erase(ctx->pool.begin() + i); }
        else ++i;
    }
}

extern "C" {

const char* sfmhip_version(void) { return "sfmhip 0.1 (gfx950)"; }

int sfmhip_create(int device, sfmhip_ctx** out)
{
    if (!out) return SFMHIP_E_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SFMHIP_E_NODEVICE;
    if (device < 0 || device >= ndev) return SFMHIP_E_ARG;
    sfmhip_ctx* ctx = new sfmhip_ctx();
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete ctx; return SFMHIP_E_HIP; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->num_cus = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return SFMHIP_E_HIP; }
    ctx->stream = ctx->own_stream;
    *out = ctx;
    return SFMHIP_OK;
}

void sfmhip_destroy(sfmhip_ctx* ctx)
{
    SFM_DEVICE_GUARD(ctx);
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    sfm_rccl_forget_ctx(ctx);
    for (auto& b : ctx->pool) (void)hipFree(b.p);
    if (ctx->d_flagpool) (void)hipFree(ctx->d_flagpool);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->scratch2) (void)hipFree(ctx->scratch2);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    free(ctx->host_scratch);
    delete ctx->copy_pool;
    for (int b = 0; b < 2; ++b) { if (ctx->stage[b]) (void)hipHostFree(ctx->stage[b]); if (ctx->stage_ev[b]) (void)hipEventDestroy(ctx->stage_ev[b]); }
    if (ctx->aux_stream) { (void)hipStreamSynchronize(ctx->aux_stream); (void)hipStreamDestroy(ctx->aux_stream); }
    for (auto& t : ctx->tev) for (auto& e : t) if (e) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

// test hook: the next n device allocations of this context fail (what an out-of-memory shard looks like to its callers)
int sfmhip_debug_fail_allocations(sfmhip_ctx* ctx, int n)
{
    if (!ctx || n < 0) return SFMHIP_E_ARG;
    ctx->inject_alloc_failures = n;
    return SFMHIP_OK;
}

int sfmhip_trim(sfmhip_ctx* ctx)
{
    SFM_DEVICE_GUARD(ctx);
    if (!ctx) return SFMHIP_E_ARG;
    sfm_pool_trim(ctx);
    return SFMHIP_OK;
}

int sfmhip_set_kernel_timing(sfmhip_ctx* ctx, int enable)
{
    SFM_DEVICE_GUARD(ctx);
    if (!ctx) return SFMHIP_E_ARG;
    if (enable) {
        for (auto& t : ctx->tev) for (auto& e : t) if (!e) SFM_HIP_TRY(ctx, hipEventCreate(&e));
    }
    ctx->timing = enable != 0;
    ctx->timing_used = 0;
    return SFMHIP_OK;
}

int sfmhip_match_kernel_ms(sfmhip_ctx* ctx, double out_ms[4])
{
    SFM_DEVICE_GUARD(ctx);
    if (!ctx || !out_ms) return SFMHIP_E_ARG;
    out_ms[0] = out_ms[1] = out_ms[2] = out_ms[3] = 0.0;
    SFM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const int n = ctx->timing_used;
    for (int i = 0; i < n; ++i) {
        float a = 0.0f, b = 0.0f;
        SFM_HIP_TRY(ctx, hipEventElapsedTime(&a, ctx->tev[i][0], ctx->tev[i][1]));
        SFM_HIP_TRY(ctx, hipEventElapsedTime(&b, ctx->tev[i][1], ctx->tev[i][2]));
        out_ms[0] += a; out_ms[1] += b;
    }
    if (n > 0) { out_ms[0] /= n; out_ms[1] /= n; }
    out_ms[2] = n;
    ctx->timing_used = 0;
    return SFMHIP_OK;
}

int sfmhip_set_stream(sfmhip_ctx* ctx, void* hip_stream)
{
    if (!ctx) return SFMHIP_E_ARG;
    hipStream_t next = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    // cached device blocks and the staging ring are reused in stream order: drain the old stream before work moves to another
    if (next != ctx->stream) { SFM_DEVICE_GUARD(ctx); (void)hipStreamSynchronize(ctx->stream); }
    ctx->stream = next;
    return SFMHIP_OK;
}

int sfmhip_synchronize(sfmhip_ctx* ctx)
{
    SFM_DEVICE_GUARD(ctx);
    if (!ctx) return SFMHIP_E_ARG;
    SFM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SFMHIP_OK;
}

const char* sfmhip_last_error(sfmhip_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

// match_features' ratio tail on the host, reference arithmetic (NViewReconstuct.cpp:880-908):
// the ratio compare promotes to double (float > 0.6 * float), the absolute gate stays float.
int sfmhip_ratio_filter(const int32_t* idx2, const float* dist2, int nq,
                        double ratio, float floor_, float mult, sfm_dmatch* out, int* n_out)
{
    if (!idx2 || !dist2 || !out || !n_out || nq < 0) return SFMHIP_E_ARG;
    float min_dist = FLT_MAX;
    for (int i = 0; i < nq; ++i) {
        if (idx2[2 * i] < 0 || idx2[2 * i + 1] < 0) continue;
        const float d0 = dist2[2 * i], d1 = dist2[2 * i + 1];
        if ((double)d0 > ratio * (double)d1) continue;
        if (d0 < min_dist) min_dist = d0;
    }
    const float gate = mult * (min_dist > floor_ ? min_dist : floor_);
    int n = 0;
    for (int i = 0; i < nq; ++i) {
        if (idx2[2 * i] < 0 || idx2[2 * i + 1] < 0) continue;
        const float d0 = dist2[2 * i], d1 = dist2[2 * i + 1];
        if ((double)d0 > ratio * (double)d1 || d0 > gate) continue;
        out[n].queryIdx = i; out[n].trainIdx = idx2[2 * i]; out[n].imgIdx = 0; out[n].distance = d0;
        ++n;
    }
    *n_out = n;
    return SFMHIP_OK;
}

}  // extern "C"
