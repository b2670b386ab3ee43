// Microbenchmark + check of the 32x32 pivot-block factorisation (one wave): cycles per call, |L L' - A|, |L Z - I|.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "../sfm_opencv_amd/csrc/ba_solver.hpp"
__global__ __launch_bounds__(256) void kbench(const double* g, double* out, long long* cyc, int reps, int mode)
{
    __shared__ SolverLds s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave != 0 && mode == 0) return;
    long long t0 = 0, t1 = 0;
    for (int r = 0; r < reps; ++r) {
        if (wave == 0) { for (int c = 0; c < 32; ++c) s.D[lane & 31][c] = g[(lane & 31) * 32 + c]; }
        __syncthreads();
        if (tid == 0) t0 += __builtin_amdgcn_s_memtime();
        if (wave == 0) { bool ok = wave_chol32(s, lane); if (!ok && lane == 0) out[4096] = -1; }
        if (tid == 0) t1 += __builtin_amdgcn_s_memtime();
        __syncthreads();
    }
    if (wave == 0) {
        if (lane < 32) for (int c = 0; c < 32; ++c) out[lane * 32 + c] = s.D[lane][c];          // L
        else for (int c = 0; c < 32; ++c) out[1024 + c * 32 + (lane - 32)] = s.W[lane][c];       // Z[m][c] = W[32+c][m]
    }
    if (tid == 0) cyc[0] = (t1 - t0) / reps;
}
int main()
{
    std::vector<double> h(1024, 0.0);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) h[i * 32 + j] = (i == j) ? 40.0 + i : 1.0 / (1 + abs(i - j)) + 0.3 * sin(i * j);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < i; ++j) h[j * 32 + i] = h[i * 32 + j];
    double *g, *o; long long* c;
    hipMalloc(&g, 8192); hipMalloc(&o, 8 * 5000); hipMalloc(&c, 64);
    hipMemset(o, 0, 8 * 5000);
    hipMemcpy(g, h.data(), 8192, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(kbench, dim3(1), dim3(256), 0, 0, g, o, c, 2, mode);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kbench, dim3(1), dim3(256), 0, 0, g, o, c, 200, mode);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long hc; hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
        printf("mode %d: %6lld s_memtime ticks per call, %.2f us per call (whole loop)\n", mode, hc, ms * 1000.0 / 200);
    }
    std::vector<double> r(4097);
    hipMemcpy(r.data(), o, 8 * 4097, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double a = 0, b = 0;
        for (int m = 0; m < 32; ++m) { a += r[i * 32 + m] * r[j * 32 + m]; b += r[i * 32 + m] * r[1024 + m * 32 + j]; }
        e1 = fmax(e1, fabs(a - h[i * 32 + j])); e2 = fmax(e2, fabs(b - (i == j)));
    }
#ifdef SFM_CHOL_STAMPS
    { long long st[16]; hipMemcpyFromSymbol(st, HIP_SYMBOL(g_chol_stamps), sizeof st);
      printf("stamps (cycles): panel0 init %lld pivots %lld tail %lld | panel1 init+cross %lld pivots %lld tail %lld | final %lld | total %lld\n",
             st[1] - st[0], st[2] - st[1], st[4] - st[2], st[5] - st[4], st[6] - st[5], st[8] - st[6], st[9] - st[8], st[9] - st[0]); }
#endif
    printf("ok flag %g  max|LL'-A| = %.3e  max|L Z - I| = %.3e  L[5][3]=%.6f Z[5][3]=%.6f upper L[3][5]=%g\n", r[4096], e1, e2, r[5 * 32 + 3], r[1024 + 5 * 32 + 3], r[3 * 32 + 5]);
    return 0;
}
