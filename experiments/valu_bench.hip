// VALU issue rate of the kNN epilogue mix (v_lshl_add_u32 + v_med3_i32 + v_min_i32 per value) at 1, 2, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(int* out, int iters, int seed)
{
    int b1[16], b2[16], a[16];
    for (int i = 0; i < 16; ++i) { b1[i] = 0x7fffffff; b2[i] = 0x7fffffff; a[i] = threadIdx.x * 977 + i * 131 + seed; }
    int nbt = seed * 3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int key = (int)(((unsigned)a[i] << 8) + (unsigned)nbt);
            const int lo = b1[i] < b2[i] ? b1[i] : b2[i], hi = b1[i] < b2[i] ? b2[i] : b1[i];
            const int t = hi < key ? hi : key;
            b2[i] = lo > t ? lo : t;
            b1[i] = b1[i] < key ? b1[i] : key;
            asm volatile("" : "+v"(b1[i]), "+v"(b2[i]), "+v"(a[i]));
        }
        nbt += 1;
    }
    int s = 0;
    for (int i = 0; i < 16; ++i) s += b1[i] ^ b2[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    int* o; hipMalloc(&o, 4096 * 256 * 4);
    const int iters = 20000;
    for (int wgs : {256, 512, 1024, 2048}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, o, 10, 1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, o, iters, 1);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)wgs * 4 / 1024.0 * iters * 48;
        printf("%d waves/SIMD: %.3f ms, %.2f ns per VALU instr per SIMD (= %.2f cycles at 2.4 GHz)\n", wgs / 256, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    }
    return 0;
}
