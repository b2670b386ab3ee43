// Issue rate of candidate kNN-epilogue instruction mixes, integer against float forms, at 1 / 2 / 4 / 8 waves per SIMD:
//   INT3   v_lshl_add_u32 + v_med3_i32 + v_min_i32          (the shipped exact top-2: key = (d2 << 7) + slot)
//   INT2   v_med3_i32 + v_min_i32                            (value-only top-2 on integers)
//   F4     v_cvt_f32_i32 + v_fma_f32 + v_med3_f32 + v_min_f32 (value-only top-2 on float d2 = fma(-2, float(acc), norm))
//   F3PK   v_cvt_f32_i32 + 1/2 v_pk_fma_f32 + v_med3_f32 + v_min_f32
//   F2     v_med3_f32 + v_min_f32
// build: hipcc -O3 --offload-arch=gfx950 -o valu_bench2 valu_bench2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(int* out, int iters, int seed)
{
    int b1[16], b2[16], a[16]; float f1[16], f2[16];
    for (int i = 0; i < 16; ++i) { b1[i] = 0x7fffffff; b2[i] = 0x7fffffff; f1[i] = 3e38f; f2[i] = 3e38f; a[i] = threadIdx.x * 977 + i * 131 + seed; }
    int nbt = seed * 3; float nf = (float)seed;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = MODE == 0 ? (int)(((unsigned)a[i] << 8) + (unsigned)nbt) : a[i];
                { const int lo = b1[i] < b2[i] ? b1[i] : b2[i], hi = b1[i] < b2[i] ? b2[i] : b1[i]; const int t = hi < key ? hi : key; b2[i] = lo > t ? lo : t; }   // -> v_med3_i32
                b1[i] = b1[i] < key ? b1[i] : key;
                asm volatile("" : "+v"(b1[i]), "+v"(b2[i]), "+v"(a[i]));
            }
        } else if (MODE == 2 || MODE == 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float e = MODE == 2 ? fmaf(-2.0f, (float)a[i], nf) : __int_as_float(a[i]);
                f2[i] = __builtin_amdgcn_fmed3f(f1[i], f2[i], e);
                f1[i] = fminf(f1[i], e);
                asm volatile("" : "+v"(f1[i]), "+v"(f2[i]), "+v"(a[i]));
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                v2f x = { (float)a[i], (float)a[i + 1] }, n2 = { nf, nf }, m2 = { -2.0f, -2.0f };
                v2f e = __builtin_elementwise_fma(m2, x, n2);
                f2[i] = __builtin_amdgcn_fmed3f(f1[i], f2[i], e.x); f1[i] = fminf(f1[i], e.x);
                f2[i + 1] = __builtin_amdgcn_fmed3f(f1[i + 1], f2[i + 1], e.y); f1[i + 1] = fminf(f1[i + 1], e.y);
                asm volatile("" : "+v"(f1[i]), "+v"(f2[i]), "+v"(a[i]), "+v"(f1[i + 1]), "+v"(f2[i + 1]), "+v"(a[i + 1]));
            }
        }
        nbt += 1; nf += 1.0f;
    }
    int s = 0;
    for (int i = 0; i < 16; ++i) s += b1[i] ^ b2[i] ^ __float_as_int(f1[i]) ^ __float_as_int(f2[i]);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, double ops_per_value, int* o)
{
    const int iters = 20000;
    for (int wgs : { 256, 512, 1024, 2048 }) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, o, 10, 1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, o, iters, 1);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double values_per_simd = (double)wgs * 4 / 1024.0 * iters * 16;
        printf("%-6s %d waves/SIMD: %7.3f ms  %.2f ns per value per SIMD  (%.2f ns per instruction at %.1f instr/value)\n", name, wgs / 256, ms, ms * 1e6 / values_per_simd,
               ms * 1e6 / values_per_simd / ops_per_value, ops_per_value);
    }
}
int main()
{
    int* o; hipMalloc(&o, 4096 * 256 * 4);
    run<0>("INT3", 3, o); run<1>("INT2", 2, o); run<2>("F4", 4, o); run<3>("F3PK", 3.5, o); run<4>("F2", 2, o);
    return 0;
}
