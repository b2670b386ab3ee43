// Shape of one wave's store instruction in a row-major 10000 x ld float matrix: 8 rows x 128 B, 4 x 256 B, 2 x 512 B, 1 x 1 KB.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int ROWS_PER_INSTR>       // lanes per row = 64 / ROWS_PER_INSTR, each lane 16 B
__global__ __launch_bounds__(256) void wr2d(float* p, int rows, int cols, size_t ld)
{
    constexpr int LPR = 64 / ROWS_PER_INSTR, W = LPR * 4;      // floats per row per instruction
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q0 = blockIdx.x * 128 + wave * 32;               // each wave owns 32 rows x 128 columns of the block
    const int t0 = blockIdx.y * 128;
    for (int c0 = 0; c0 < 128; c0 += W)
        for (int r0 = 0; r0 < 32; r0 += ROWS_PER_INSTR) {
            const int row = q0 + r0 + lane / LPR, col = t0 + c0 + (lane % LPR) * 4;
            if (row < rows && col + 3 < cols) { v4f v = { 1.f, 2.f, 3.f, (float)col }; *(v4f*)(p + (size_t)row * ld + col) = v; }
        }
}
int main()
{
    const size_t n = 100000000; float* p; hipMalloc(&p, (n + 4000000) * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto t = [&](const char* name, auto f) { for (int i = 0; i < 3; ++i) f(); hipEventRecord(a); for (int i = 0; i < 20; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); printf("%-28s %.1f us  %.2f TB/s\n", name, ms / 20 * 1e3, 4e8 / (ms / 20 * 1e-3) / 1e12); };
    for (size_t ld : { (size_t)10000, (size_t)10112 }) {
        printf("ld = %zu\n", ld);
        t("8 rows x 128 B", [&] { hipLaunchKernelGGL(wr2d<8>, dim3(79, 79), dim3(256), 0, 0, p, 10000, 10000, ld); });
        t("4 rows x 256 B", [&] { hipLaunchKernelGGL(wr2d<4>, dim3(79, 79), dim3(256), 0, 0, p, 10000, 10000, ld); });
        t("2 rows x 512 B", [&] { hipLaunchKernelGGL(wr2d<2>, dim3(79, 79), dim3(256), 0, 0, p, 10000, 10000, ld); });
    }
    return 0;
}
