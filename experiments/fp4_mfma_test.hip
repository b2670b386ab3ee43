// Semantics check of v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 (e2m1) operands and per-lane E8M0 block scales, on exact data.
// Hypothesis: lane l (r = l & 31, h = l >> 5) supplies 32 K-values (block h) of A row r / B column r in 4 dwords (8 nibbles each);
// its scale byte multiplies that block by 2^(scale-127); C/D: col = lane & 31 (B), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (A).
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/fp4_test experiments/fp4_mfma_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef int   v8i  __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void k_test(const uint32_t* A, const uint32_t* B, const int* sa, const int* sb, float* C, int mode)
{
    const int lane = threadIdx.x;
    v8i a = {0,0,0,0,0,0,0,0}, b = {0,0,0,0,0,0,0,0};
    for (int j = 0; j < 4; ++j) { a[j] = (int)A[lane * 4 + j]; b[j] = (int)B[lane * 4 + j]; }
    v16f c;
    for (int i = 0; i < 16; ++i) c[i] = 0.0f;
    if (mode == 0) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, 0, 0, 0);
    else           c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, sa[lane], 0, sb[lane]);
    for (int i = 0; i < 16; ++i) C[lane * 16 + i] = c[i];
}

static float fp4_val(int n) { static const float m[8] = {0.f, .5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f}; return (n & 8) ? -m[n & 7] : m[n & 7]; }

int main()
{
    uint32_t hA[256], hB[256]; int hsa[64], hsb[64]; float hC[1024];
    srand(7);
    for (int i = 0; i < 256; ++i) { hA[i] = 0; hB[i] = 0; for (int n = 0; n < 8; ++n) { hA[i] |= (uint32_t)(rand() & 15) << (4 * n); hB[i] |= (uint32_t)(rand() & 15) << (4 * n); } }
    for (int l = 0; l < 64; ++l) { hsa[l] = 127 + (l % 5); hsb[l] = 127 - (l % 3) + ((l >> 5) ? 2 : 0); }
    uint32_t *dA, *dB; int *dsa, *dsb; float* dC;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dC, 4096);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k_test, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dC, mode);
        hipMemcpy(hC, dC, 4096, hipMemcpyDeviceToHost);
        int bad = 0; double maxerr = 0;
        for (int lane = 0; lane < 64; ++lane) for (int reg = 0; reg < 16; ++reg) {
            const int col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            double ref = 0;
            for (int h = 0; h < 2; ++h) {
                const int la = row + 32 * h, lb = col + 32 * h;
                double s = 0;
                for (int j = 0; j < 32; ++j) s += (double)fp4_val((hA[la * 4 + j / 8] >> (4 * (j % 8))) & 15) * fp4_val((hB[lb * 4 + j / 8] >> (4 * (j % 8))) & 15);
                if (mode == 1) s *= ldexp(1.0, hsa[la] - 127) * ldexp(1.0, hsb[lb] - 127);
                ref += s;
            }
            const double e = fabs(ref - hC[lane * 16 + reg]);
            if (e > 0) { ++bad; if (e > maxerr) maxerr = e; if (bad <= 4) printf("  mode %d lane %d reg %d: got %g want %g\n", mode, lane, reg, hC[lane * 16 + reg], ref); }
        }
        printf("mode %d (%s): %d mismatches of 1024, max err %g\n", mode, mode ? "per-lane scales" : "literal zero scales", bad, maxerr);
    }
    return 0;
}
