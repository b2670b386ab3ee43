// Peak check: v_mfma_i32_32x32x32_i8 issue rate (operands in registers), one accumulator chain vs four.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(int* out, int iters, long long* cyc)
{
    v4i a = { (int)threadIdx.x, 2, 3, 4 }, b = { 5, (int)threadIdx.x * 3, 7, 8 };
    v16i acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16 / NACC; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC>
static void run(int wgs, const char* what)
{
    int* o; long long* c; hipMalloc(&o, (size_t)wgs * 256 * 4); hipMalloc(&c, 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(wgs), dim3(256), 0, 0, o, 10, c);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<NACC>, dim3(wgs), dim3(256), 0, 0, o, iters, c);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long hc; hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
    const double n_mfma = (double)wgs * 4 * iters * 16;
    printf("%-28s wgs %5d: %.3f ms, %.1f TOP/s, %.1f ticks per MFMA per wave (wave 0)\n", what, wgs, ms, n_mfma * 65536.0 / (ms * 1e-3) / 1e12, (double)hc / (iters * 16.0));
    hipFree(o); hipFree(c);
}
int main()
{
    run<1>(256, "1 chain, 1 wave/SIMD");
    run<4>(256, "4 chains, 1 wave/SIMD");
    run<1>(512, "1 chain, 2 waves/SIMD");
    run<4>(512, "4 chains, 2 waves/SIMD");
    run<4>(2048, "4 chains, 2048 WGs");
    return 0;
}
