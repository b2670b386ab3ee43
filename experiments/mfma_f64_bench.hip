// v_mfma_f64_16x16x4_f64 on gfx950: cycles per instruction for 1, 2 and 4 independent accumulator chains, one wave
// per SIMD and two waves per SIMD (the reduced-system solver runs 8 waves per workgroup = 2 per SIMD).
//   hipcc -O3 --offload-arch=gfx950 mfma_f64_bench.hip -o mfma_f64_bench && ./mfma_f64_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int CH>
__global__ __launch_bounds__(512) void k(double* out, long long* cyc, int iters)
{
    const int lane = threadIdx.x & 63;
    double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-3;
    v4d c[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) c[i] = v4d{ 0.0, 0.0, 0.0, 0.0 };
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < CH; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < CH; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int CH>
void run(int threads, const char* what)
{
    double* out; long long* cyc;
    hipMalloc((void**)&out, 1024 * sizeof(double)); hipMalloc((void**)&cyc, 8 * sizeof(long long));
    const int iters = 2000;
    k<CH><<<1, threads>>>(out, cyc, iters);
    k<CH><<<1, threads>>>(out, cyc, iters);
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-22s %d chains: %6.1f s_memtime ticks per MFMA per wave\n", what, CH, (double)h / (iters * CH));
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<1>(256, "1 wave/SIMD"); run<2>(256, "1 wave/SIMD"); run<4>(256, "1 wave/SIMD");
    run<1>(512, "2 waves/SIMD"); run<2>(512, "2 waves/SIMD"); run<4>(512, "2 waves/SIMD");
    // s_memtime tick vs wall clock: time a long kernel
    double* out; long long* cyc; hipMalloc((void**)&out, 1024 * sizeof(double)); hipMalloc((void**)&cyc, 8 * sizeof(long long));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<4><<<1, 256>>>(out, cyc, 200000); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h; hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("s_memtime: %.1f ticks per microsecond (kernel %.3f ms, %lld ticks); %.1f ns per MFMA\n", h / (ms * 1e3), ms, h, ms * 1e6 / (200000.0 * 4));
    return 0;
}
