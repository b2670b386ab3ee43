// Sustained rate of the two FP4 block-scaled MFMA shapes on random +-1 data, operands in registers, 2 waves per SIMD on every CU:
//   v_mfma_scale_f32_32x32x64_f8f6f4 (32 cycles) against v_mfma_scale_f32_16x16x128_f8f6f4 (16 cycles), same flops per cycle on paper.
// The chip lowers its clock under matrix load and the shape can change by how much (MI355X_MICROARCH.md, DVFS item 7).
// build: hipcc --offload-arch=gfx950 -O3 -o experiments/_exp/mfma_fp4_bench experiments/mfma_fp4_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int   v4i  __attribute__((ext_vector_type(4)));
typedef int   v8i  __attribute__((ext_vector_type(8)));
typedef float v4f  __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void k_bench(const v4i* __restrict__ data, float* __restrict__ out, int iters)
{
    const int lane = threadIdx.x & 63;
    v4i a[12], b[4];
    for (int i = 0; i < 12; ++i) a[i] = data[(blockIdx.x * 7 + i) % 64 * 64 + lane];
    for (int i = 0; i < 4; ++i) b[i] = data[(blockIdx.x * 3 + i + 17) % 64 * 64 + lane];
    float sum = 0.0f;
    if (SHAPE == 32) {
        v16f c0, c1;
        for (int i = 0; i < 16; ++i) { c0[i] = 0.0f; c1[i] = 0.0f; }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 12; ++s) {
                const v8i a0 = { a[s][0], a[s][1], a[s][2], a[s][3], 0, 0, 0, 0 };
                const v8i a1 = { a[11 - s][0], a[11 - s][1], a[11 - s][2], a[11 - s][3], 0, 0, 0, 0 };
                const v8i b8 = { b[s & 3][0], b[s & 3][1], b[s & 3][2], b[s & 3][3], 0, 0, 0, 0 };
                c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a0, b8, c0, 4, 4, 0, 127, 0, 127);
                c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a1, b8, c1, 4, 4, 0, 127, 0, 127);
            }
        }
        for (int i = 0; i < 16; ++i) sum += c0[i] + c1[i];
    } else {
        v4f c[4];
        for (int t = 0; t < 4; ++t) for (int i = 0; i < 4; ++i) c[t][i] = 0.0f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 12; ++s) {        // 12 x 4 MFMAs of 16 cycles = the cycles of 12 x 2 of 32
                const v8i b8 = { b[s & 3][0], b[s & 3][1], b[s & 3][2], b[s & 3][3], 0, 0, 0, 0 };
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const v4i av = a[(s + 3 * t) % 12];
                    const v8i a8 = { av[0], av[1], av[2], av[3], 0, 0, 0, 0 };
                    c[t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, c[t], 4, 4, 0, 127, 0, 127);
                }
            }
        }
        for (int t = 0; t < 4; ++t) for (int i = 0; i < 4; ++i) sum += c[t][i];
    }
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

int main()
{
    std::vector<uint32_t> h(64 * 64 * 4);
    srand(3);
    for (auto& w : h) { w = 0; for (int n = 0; n < 8; ++n) w |= (uint32_t)((rand() & 1) ? 0x2 : 0xA) << (4 * n); }
    v4i* d; float* o;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, 2048 * 256 * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, grid = 2048;       // 2048 workgroups of 4 waves: 4 rounds of 2 workgroups per CU
    for (int rep = 0; rep < 3; ++rep)
        for (int shape : { 32, 16 }) {
            hipEventRecord(e0);
            if (shape == 32) hipLaunchKernelGGL(k_bench<32>, dim3(grid), dim3(256), 0, 0, d, o, iters);
            else             hipLaunchKernelGGL(k_bench<16>, dim3(grid), dim3(256), 0, 0, d, o, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)grid * 4 * iters * (shape == 32 ? 24.0 * 2 * 32 * 32 * 64 : 48.0 * 2 * 16 * 16 * 128);
            const double cyc = (double)grid * 4 / 1024.0 * iters * 24 * 32;          // matrix-pipe cycles per SIMD
            printf("shape %2d: %.3f ms  %.2f PFLOP/s  pipe-bound clock %.3f GHz\n", shape, ms, flops / ms / 1e12, cyc / ms / 1e6);
        }
    return 0;
}
