// Microbenchmark + check of the 32x32 pivot-block factorisation (one wave): cycles per call, |L L' - A|, |L Z - I|.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "../sfm_opencv_amd/csrc/ba_solver.hpp"
template <int V> __device__ __forceinline__ bool chol_var(DiagLds& s, int lane)
{
    const bool lower = lane < SNB;
    const int ident = lane - SNB;                           // upper half-wave: column index of L^-1
    const int li = lane & 15, lk = lane >> 4;
    const double* dr = &s.D[lane & (SNB - 1)][0];
    bool ok = true;
#pragma unroll 1
    for (int b = 0; b < SNB / 8; ++b) {
        const int p = 8 * b;
        double x[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) x[c] = (V & 8) ? 40.0 + lane + c : (lower ? dr[p + c] : (ident == p + c ? 1.0 : 0.0));
        if (b > 0 && !(V & 1)) {
            // G = W[:, :p] W[p:p+8, :p]'  (64 x 8; the MFMA's columns 8..15 repeat 0..7 and are dropped)
            v4d acc[4];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) acc[rt] = v4d{ 0.0, 0.0, 0.0, 0.0 };
#pragma unroll 2
            for (int kk = 0; kk < 2 * b; ++kk) {
                const double bop = s.W[p + (li & 7)][4 * kk + lk];
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
                    acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(s.W[16 * rt + li][4 * kk + lk], bop, acc[rt], 0, 0, 0);
            }
            if (li < 8) {
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) s.G[16 * rt + lk + 4 * g][li] = acc[rt][g];
            }
            wave_sync_lds();
#pragma unroll
            for (int c = 0; c < 8; c += 2) { const v2d t = *(const v2d*)&s.G[lane][c]; x[c] -= t.x; x[c + 1] -= t.y; }
        }
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int j = p + jj;
            const double d = readlane_f64(x[jj], j);
            ok = ok && (d > 0.0) && (d < 1e300);
            const double dd = d > 0.0 ? d : 1.0;
            const double y = (V & 2) ? dd * 0.01 : rsqrt_refined(dd);
            double sd = dd * y;
            sd = fma(fma(-sd, sd, dd), 0.5 * y, sd);
            double l = (lane == j) ? sd : x[jj] * y;
            if (lane < j) l = 0.0;                          // above the diagonal of L (upper half-wave: lane >= 32 > j)
            x[jj] = l;
#pragma unroll
            for (int c = jj + 1; c < ((V & 4) ? jj + 2 : 8) && c < 8; ++c) x[c] = fma(-l, readlane_f64(l, p + c), x[c]);
        }
#pragma unroll
        for (int c = 0; c < 8; c += 2) *(v2d*)&s.W[lane][p + c] = v2d{ x[c], x[c + 1] };
        wave_sync_lds();
    }
    if (lower) {
        v2d t[SNB / 2];                 // all reads first: interleaved with the stores the compiler waits after every one
        const double* wr = &s.W[lane][0];
#pragma unroll
        for (int c = 0; c < SNB / 2; ++c) t[c] = *(const v2d*)(wr + 2 * c);
#pragma unroll
        for (int c = 0; c < SNB / 2; ++c) { s.D[lane][2 * c] = t[c].x; s.D[lane][2 * c + 1] = t[c].y; }
    }
    return ok;
}


template <int V>
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
        if (wave == 0) { bool ok = chol_var<V>(s, lane); if (!ok && lane == 0) out[4096] = -1; }
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
    auto run = [&](auto kern, const char* what) {
    for (int mode = 0; mode < 1; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(kern, dim3(1), dim3(256), 0, 0, g, o, c, 2, mode);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(1), dim3(256), 0, 0, g, o, c, 200, mode);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long hc; hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
        printf("%-30s: %6lld s_memtime ticks per call, %.2f us per call (whole loop)\n", what, hc, ms * 1000.0 / 200);
    } };
    run(kbench<1>, "no MFMA cross-panel"); run(kbench<2>, "no rsqrt"); run(kbench<4>, "only next-column update"); run(kbench<8>, "no init LDS reads");
    run(kbench<15>, "all knocked out"); run(kbench<0>, "full");
    std::vector<double> r(4097);
    hipMemcpy(r.data(), o, 8 * 4097, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double a = 0, b = 0;
        for (int m = 0; m < 32; ++m) { a += r[i * 32 + m] * r[j * 32 + m]; b += r[i * 32 + m] * r[1024 + m * 32 + j]; }
        e1 = fmax(e1, fabs(a - h[i * 32 + j])); e2 = fmax(e2, fabs(b - (i == j)));
    }
    printf("ok flag %g  max|LL'-A| = %.3e  max|L Z - I| = %.3e  L[5][3]=%.6f Z[5][3]=%.6f upper L[3][5]=%g\n", r[4096], e1, e2, r[5 * 32 + 3], r[1024 + 5 * 32 + 3], r[3 * 32 + 5]);
    return 0;
}
