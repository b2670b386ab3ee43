// Do the kNN tile's two halves overlap across waves?  Each wave repeats [4 x v_mfma_i32_32x32x32_i8] + [48 VALU ops of
// the top-2 epilogue on OTHER registers]; variants: MFMA only, VALU only, both; 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
template <int MODE>   // 1 = MFMA, 2 = VALU, 3 = both (epilogue consumes the previous group's accumulator like the real kernel)
__global__ __launch_bounds__(256) void k(int* out, int iters, int seed)
{
    v4i a = { (int)threadIdx.x, 2, 3, seed }, b = { 5, (int)threadIdx.x * 3, 7, 8 };
    int b1[16], b2[16];
    for (int i = 0; i < 16; ++i) { b1[i] = 0x7fffffff; b2[i] = 0x7fffffff; }
    v16i prev = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int nbt = seed;
    for (int it = 0; it < iters; ++it) {
        v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (MODE & 1) {
#pragma unroll
            for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
        }
        if (MODE & 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = (int)(((unsigned)prev[i] << 8) + (unsigned)nbt);
                const int lo = b1[i] < b2[i] ? b1[i] : b2[i], hi = b1[i] < b2[i] ? b2[i] : b1[i];
                const int t = hi < key ? hi : key;
                b2[i] = lo > t ? lo : t;
                b1[i] = b1[i] < key ? b1[i] : key;
                asm volatile("" : "+v"(b1[i]), "+v"(b2[i]));
            }
        }
        if (MODE == 3) prev = acc; else { asm volatile("" : "+v"(acc)); if (MODE == 2) prev[it & 15] += it; }
        nbt += 1;
    }
    int s = 0;
    for (int i = 0; i < 16; ++i) s += b1[i] ^ b2[i] ^ prev[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
static void run(int wgs, const char* what)
{
    int* o; hipMalloc(&o, 4096 * 256 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, o, 10, 1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, o, iters, 1);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double groups_per_simd = (double)wgs * 4 / 1024.0 * iters;
    printf("%-10s %d waves/SIMD: %8.3f ms  -> %.1f ns per tile-group per SIMD\n", what, wgs / 256, ms, ms * 1e6 / groups_per_simd);
    hipFree(o);
}
int main()
{
    for (int w : {1, 2, 4}) { run<1>(256 * w, "MFMA"); run<2>(256 * w, "VALU"); run<3>(256 * w, "both"); }
    return 0;
}
