// How should sfmhip_ba_create bring 100+ MB of caller (pageable) memory into HBM?  Times, per strategy, 128 MB + 64 MB + 32 MB + 32 MB
// (C5's pixel / point / index arrays): plain hipMemcpyAsync from pageable memory, hipHostRegister + copy + unregister, and a
// pipeline through two pinned staging buffers filled by host threads.  Also: hipMalloc / hipFree cost by size.
// build: hipcc -O2 --offload-arch=gfx950 -fopenmp -o h2d_bench h2d_bench.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include <thread>
#include <omp.h>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main()
{
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t sizes[4] = { 128u << 20, 64u << 20, 32u << 20, 32u << 20 };
    std::vector<std::vector<char>> src(4);
    void* dst[4];
    for (int i = 0; i < 4; ++i) { src[i].assign(sizes[i], (char)i); CK(hipMalloc(&dst[i], sizes[i])); }
    for (int rep = 0; rep < 3; ++rep) {
        double t = now();
        for (int i = 0; i < 4; ++i) CK(hipMemcpyAsync(dst[i], src[i].data(), sizes[i], hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        printf("pageable hipMemcpyAsync x4 (256 MB): %.2f ms\n", now() - t);
        t = now();
        for (int i = 0; i < 4; ++i) CK(hipHostRegister(src[i].data(), sizes[i], hipHostRegisterDefault));
        double t1 = now();
        for (int i = 0; i < 4; ++i) CK(hipMemcpyAsync(dst[i], src[i].data(), sizes[i], hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        double t2 = now();
        for (int i = 0; i < 4; ++i) CK(hipHostUnregister(src[i].data()));
        printf("register %.2f + copy %.2f + unregister %.2f = %.2f ms\n", t1 - t, t2 - t1, now() - t2, now() - t);
        // staged: two pinned buffers of 16 MB, memcpy by T threads, async copy
        for (int threads : { 1, 4, 8 }) {
            const size_t CH = 16u << 20;
            static void* pin[2] = { nullptr, nullptr }; static hipEvent_t ev[2] = { nullptr, nullptr };
            if (!pin[0]) for (int b = 0; b < 2; ++b) { CK(hipHostMalloc(&pin[b], CH)); CK(hipEventCreateWithFlags(&ev[b], hipEventDisableTiming)); }
            t = now();
            int b = 0; bool used[2] = { false, false };
            for (int i = 0; i < 4; ++i)
                for (size_t off = 0; off < sizes[i]; off += CH) {
                    const size_t n = std::min(CH, sizes[i] - off);
                    if (used[b]) CK(hipEventSynchronize(ev[b]));
                    const char* s = src[i].data() + off; char* d = (char*)pin[b];
#pragma omp parallel for num_threads(threads)
                    for (int k = 0; k < 64; ++k) { const size_t a = n * k / 64, e = n * (k + 1) / 64; memcpy(d + a, s + a, e - a); }
                    CK(hipMemcpyAsync((char*)dst[i] + off, pin[b], n, hipMemcpyHostToDevice, st));
                    CK(hipEventRecord(ev[b], st)); used[b] = true; b ^= 1;
                }
            CK(hipStreamSynchronize(st));
            printf("staged through 2 x 16 MB pinned, %d memcpy thread(s): %.2f ms\n", threads, now() - t);
        }
    }
    for (size_t mb : { 1, 16, 128, 1024 }) {
        void* p; double t = now(); CK(hipMalloc(&p, mb << 20)); double t1 = now(); CK(hipFree(p));
        printf("hipMalloc %zu MB: %.3f ms, hipFree %.3f ms\n", mb, t1 - t, now() - t1);
    }
    { void* p; double t = now(); for (int i = 0; i < 40; ++i) CK(hipMalloc(&p, 4096)); printf("40 x hipMalloc 4 KB: %.3f ms\n", now() - t); }
    { hipEvent_t evx; double t = now(); for (int i = 0; i < 30; ++i) CK(hipEventCreate(&evx)); printf("30 x hipEventCreate: %.3f ms\n", now() - t); }
    { hipStream_t s2; double t = now(); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)); printf("hipStreamCreate: %.3f ms\n", now() - t); }
    { void* p; double t = now(); CK(hipHostMalloc(&p, 128)); printf("hipHostMalloc 128 B: %.3f ms\n", now() - t); }
    return 0;
}
