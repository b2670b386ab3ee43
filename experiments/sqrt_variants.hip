// Exhaustive check of cheaper correctly-rounded sqrt sequences for integer-valued floats in [0, 2^24) (the distance-matrix epilogue):
// every variant against sqrtf (IEEE, correctly rounded on this target) for all 2^24 inputs.
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o sqrt_variants sqrt_variants.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float v_cur(float x)          // the shipped sequence: v_sqrt_f32 + neighbour test (8 ops + sqrt)
{
    float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __int_as_float(__float_as_int(s) - 1), s_up = __int_as_float(__float_as_int(s) + 1);
    const float r_dn = fmaf(-s_dn, s, x), r_up = fmaf(-s_up, s, x);
    s = r_dn <= 0.0f ? s_dn : s;
    s = r_up > 0.0f ? s_up : s;
    return s;
}
__device__ __forceinline__ float v_rsq(float x)          // Markstein: y = rsq, g = x y, h = y / 2, d = x - g g, g' = g + d h
{
    const float y = __builtin_amdgcn_rsqf(fmaxf(x, 1.0f));
    const float g = x * y, h = 0.5f * y;
    const float d = fmaf(-g, g, x);
    return fmaf(d, h, g);
}
__device__ __forceinline__ float v_sqrt_newton(float x)  // s = v_sqrt, one fma-residual Newton step with h = 0.5 * rcp(s)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rcpf(fmaxf(s, 1.0f));
    const float d = fmaf(-s, s, x);
    return fmaf(d, h, s);
}
__device__ __forceinline__ float v_sqrt_rsq(float x)     // s = v_sqrt, h = 0.5 * rsq(x): two transcendentals, two fma
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rsqf(fmaxf(x, 1.0f));
    const float d = fmaf(-s, s, x);
    return fmaf(d, h, s);
}
template <int V> __global__ void check(int* bad, int* first)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float x = (float)i;
    float r = V == 0 ? v_cur(x) : V == 1 ? v_rsq(x) : V == 2 ? v_sqrt_newton(x) : v_sqrt_rsq(x);
    if (__float_as_int(r) != __float_as_int(sqrtf(x))) { atomicAdd(bad, 1); atomicMin(first, i); }
}
int main()
{
    int *d; hipMalloc(&d, 8);
    const char* names[4] = { "v_sqrt + neighbour test (shipped)", "rsq: g = x y, d = x - g g, g + d h", "v_sqrt + fma residual * 0.5 rcp(s)", "v_sqrt + fma residual * 0.5 rsq(x)" };
    for (int v = 0; v < 4; ++v) {
        int h[2] = { 0, 1 << 30 }; hipMemcpy(d, h, 8, hipMemcpyHostToDevice);
        if (v == 0) hipLaunchKernelGGL(check<0>, dim3(65536), dim3(256), 0, 0, d, d + 1);
        if (v == 1) hipLaunchKernelGGL(check<1>, dim3(65536), dim3(256), 0, 0, d, d + 1);
        if (v == 2) hipLaunchKernelGGL(check<2>, dim3(65536), dim3(256), 0, 0, d, d + 1);
        if (v == 3) hipLaunchKernelGGL(check<3>, dim3(65536), dim3(256), 0, 0, d, d + 1);
        hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("%-40s mismatches %d (first at %d)\n", names[v], h[0], h[0] ? h[1] : -1);
    }
    return 0;
}
