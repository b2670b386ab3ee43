// Store-policy sweep for a pure 400 MB write stream: cache-policy bits of global_store_dwordx4 (inline asm), grid sizes.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int MODE> __global__ __launch_bounds__(256) void wr(v4f* p, size_t n4)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    v4f v = { 1.f, 2.f, 3.f, (float)i };
    for (; i < n4; i += stride) {
        v4f* d = p + i;
        if (MODE == 0) *d = v;
        else if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(d), "v"(v) : "memory");
        else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(d), "v"(v) : "memory");
        else if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(d), "v"(v) : "memory");
        else if (MODE == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(d), "v"(v) : "memory");
        else if (MODE == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(d), "v"(v) : "memory");
        else if (MODE == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(d), "v"(v) : "memory");
        else if (MODE == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" :: "v"(d), "v"(v) : "memory");
    }
}
int main()
{
    const size_t n = 100000000; float* p; hipMalloc(&p, (n + 2000000) * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto t = [&](const char* name, auto f) { for (int i = 0; i < 3; ++i) f(); hipEventRecord(a); for (int i = 0; i < 20; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); printf("%-28s %.1f us  %.2f TB/s\n", name, ms / 20 * 1e3, n * 4 / (ms / 20 * 1e-3) / 1e12); };
    const char* names[8] = { "plain", "nt", "sc0", "sc1", "sc0 sc1", "sc0 sc1 nt", "sc1 nt", "sc0 nt" };
    for (int g : { 2048, 16384 }) {
        printf("grid %d x 256\n", g);
        t(names[0], [&] { hipLaunchKernelGGL(wr<0>, dim3(g), dim3(256), 0, 0, (v4f*)p, n / 4); });
        t(names[1], [&] { hipLaunchKernelGGL(wr<1>, dim3(g), dim3(256), 0, 0, (v4f*)p, n / 4); });
        t(names[2], [&] { hipLaunchKernelGGL(wr<2>, dim3(g), dim3(256), 0, 0, (v4f*)p, n / 4); });
        t(names[3], [&] { hipLaunchKernelGGL(wr<3>, dim3(g), dim3(256), 0, 0, (v4f*)p, n / 4); });
        t(names[4], [&] { hipLaunchKernelGGL(wr<4>, dim3(g), dim3(256), 0, 0, (v4f*)p, n / 4); });
        t(names[5], [&] { hipLaunchKernelGGL(wr<5>, dim3(g), dim3(256), 0, 0, (v4f*)p, n / 4); });
        t(names[6], [&] { hipLaunchKernelGGL(wr<6>, dim3(g), dim3(256), 0, 0, (v4f*)p, n / 4); });
        t(names[7], [&] { hipLaunchKernelGGL(wr<7>, dim3(g), dim3(256), 0, 0, (v4f*)p, n / 4); });
    }
    return 0;
}
