#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int NT> __global__ void wr(v4f* p, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    v4f v = { 1.f, 2.f, 3.f, (float)i };
    for (; i < n4; i += stride) { if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v; }
}
// 2-D tiled write like the distance matrix: wave writes 8 rows x 128 B per instruction, rows ld floats apart
template <int NT> __global__ void wr2d(float* p, int rows, int cols, size_t ld) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q0 = blockIdx.x * 128 + wave * 32;
    for (int t = blockIdx.y * 512; t < blockIdx.y * 512 + 512 && t < cols; t += 32)
        for (int pass = 0; pass < 4; ++pass) {
            const int row = q0 + pass * 8 + (lane >> 3), col = t + (lane & 7) * 4;
            if (row < rows && col + 3 < cols) { v4f v = { 1.f, 2.f, 3.f, (float)col }; v4f* d = (v4f*)(p + (size_t)row * ld + col);
                if (NT) __builtin_nontemporal_store(v, d); else *d = v; }
        }
}
int main() {
    const size_t n = 100000000; float* p; hipMalloc(&p, (n + 2000000) * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto t = [&](const char* name, auto f) { for (int i = 0; i < 3; ++i) f(); hipEventRecord(a); for (int i = 0; i < 20; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); printf("%-28s %.1f us  %.2f TB/s\n", name, ms / 20 * 1e3, n * 4 / (ms / 20 * 1e-3) / 1e12); };
    t("linear plain 2048 blocks", [&] { hipLaunchKernelGGL(wr<0>, dim3(2048), dim3(256), 0, 0, (v4f*)p, n / 4); });
    t("linear nt 2048 blocks", [&] { hipLaunchKernelGGL(wr<1>, dim3(2048), dim3(256), 0, 0, (v4f*)p, n / 4); });
    t("linear nt 8192 blocks", [&] { hipLaunchKernelGGL(wr<1>, dim3(8192), dim3(256), 0, 0, (v4f*)p, n / 4); });
    t("2d ld=10000 plain", [&] { hipLaunchKernelGGL(wr2d<0>, dim3(79, 20), dim3(256), 0, 0, p, 10000, 10000, (size_t)10000); });
    t("2d ld=10000 nt", [&] { hipLaunchKernelGGL(wr2d<1>, dim3(79, 20), dim3(256), 0, 0, p, 10000, 10000, (size_t)10000); });
    t("2d ld=10112 plain", [&] { hipLaunchKernelGGL(wr2d<0>, dim3(79, 20), dim3(256), 0, 0, p, 10000, 10000, (size_t)10112); });
    t("2d ld=10112 nt", [&] { hipLaunchKernelGGL(wr2d<1>, dim3(79, 20), dim3(256), 0, 0, p, 10000, 10000, (size_t)10112); });
    return 0;
}
