#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../sfm_opencv_amd/csrc/ba_solver.hpp"
__global__ __launch_bounds__(256) void kbench(const double* g, double* out, long long* cyc, int reps, int mode)
{
    __shared__ SolverLds s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave != 0 && mode == 0) return;
    long long t0 = 0, t1 = 0;
    double accum = 0;
    for (int r = 0; r < reps; ++r) {
        if (wave == 0) { for (int c = 0; c < 32; ++c) s.D[lane & 31][c] = g[(lane & 31) * 32 + c]; }
        __syncthreads();
        if (tid == 0) t0 += __builtin_amdgcn_s_memtime();
        if (wave == 0) { bool ok = wave_chol32(s, lane); accum += ok; }
        if (tid == 0) t1 += __builtin_amdgcn_s_memtime();
        __syncthreads();
        accum += s.D[lane & 31][lane & 15];
    }
    out[tid] = accum;
    if (tid == 0) cyc[0] = (t1 - t0) / reps;
}
int main()
{
    std::vector<double> h(1024, 0.0);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) h[i * 32 + j] = (i == j) ? 40.0 + i : 1.0 / (1 + abs(i - j));
    double *g, *o; long long* c;
    hipMalloc(&g, 8192); hipMalloc(&o, 8192); hipMalloc(&c, 64);
    hipMemcpy(g, h.data(), 8192, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode)
        for (int reps : {1, 20}) {
            hipLaunchKernelGGL(kbench, dim3(1), dim3(256), 0, 0, g, o, c, reps, mode);
            hipDeviceSynchronize();
            long long hc; hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
            printf("mode %d reps %2d: wave_chol32 = %lld cycles per call\n", mode, reps, hc);
        }
    return 0;
}
