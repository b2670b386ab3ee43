// The inner loop of knn2_hamming2_fp4_kernel (csrc/match.hip) with BOTH block-scaled FP4 MFMA shapes, everything but the staging:
// per K-step one ds_read_b128 of a train fragment out of LDS (the kernel's swizzled addresses), the step's MFMAs, the finished tile's
// top-2 updates (v_med3_f32 pairs) behind them, one barrier per 64-row stage; 8 waves per workgroup, 64 stationary query rows per wave.
//   shape 32: v_mfma_scale_f32_32x32x64_f8f6f4,  2 per K-step (32 cycles each), 12 K-steps per 32-train tile, 32 top-2 slots per lane
//   shape 16: v_mfma_scale_f32_16x16x128_f8f6f4, 4 per K-step (16 cycles each),  6 K-steps per 16-train tile, 16 top-2 slots per lane
// Same flops, LDS bytes and VALU ops per distance; what differs is the clock the part holds (mfma_fp4_bench.hip: 16x16x128 runs 3-9 % higher
// in a bare loop) and how the shorter instructions share the SIMD with the other wave.  Verdict item 6 of round 3 ("16x16x128 shape").
// build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o experiments/_exp/fp4_loop_bench experiments/fp4_loop_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int   v4i  __attribute__((ext_vector_type(4)));
typedef int   v8i  __attribute__((ext_vector_type(8)));
typedef float v4f  __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define RB 384
#define TROWS 64
#define BUF_BYTES (TROWS * RB)
#define BUF_STRIDE (TROWS * 512)
#define TOP2(B1v, B2v, KEY) asm volatile("v_med3_f32 %1, %0, %1, %2\n\tv_med3_f32 %0, %0, %2, %3" : "+v"(B1v), "+v"(B2v) : "v"(KEY), "s"(neg_inf))

template <int SHAPE, int mode>
__global__ __launch_bounds__(512, 2) void k_loop(const v4i* __restrict__ data, float* __restrict__ out, int nblocks)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[BUF_STRIDE + BUF_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < (BUF_STRIDE + BUF_BYTES) / 16; i += 512) ((v4i*)lds)[i] = data[(i * 5 + blockIdx.x) & 4095];
    __syncthreads();
    float neg_inf = -__builtin_inff();
    asm volatile("" : "+s"(neg_inf));
    const int sa = 133, sb = 127;
    float sum = 0.0f;
    if (SHAPE == 32) {
        const int l31 = lane & 31, half = lane >> 5;
        v4i afrag[2][12];
#pragma unroll
        for (int at = 0; at < 2; ++at)
#pragma unroll
            for (int s = 0; s < 12; ++s) { v4i a = data[(blockIdx.x * 7 + at * 12 + s) % 64 * 64 + lane]; asm volatile("" : "+v"(a)); afrag[at][s] = a; }
        float best1[2][16], best2[2][16];
#pragma unroll
        for (int at = 0; at < 2; ++at)
#pragma unroll
            for (int i = 0; i < 16; ++i) { best1[at][i] = 1.0e9f; best2[at][i] = 1.0e9f; }
        int rd_off[4];
        {
            const int base_lane = l31 * RB, k16 = 16 * ((l31 >> 1) & 7);
#pragma unroll
            for (int j = 0; j < 4; ++j) rd_off[j] = base_lane + ((32 * j + 16 * half) ^ k16);
        }
        v16f accA0, accA1, accB0, accB1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { accB0[i] = 1.0e9f; accB1[i] = 1.0e9f; }
        auto top2_of = [&](v16f& p0, v16f& p1, int v) {
            if (v < 16) TOP2(best1[0][v], best2[0][v], p0[v]);
            else if (v < 32) TOP2(best1[1][v - 16], best2[1][v - 16], p1[v - 16]);
        };
        constexpr int NG = 24;
        for (int blk = 0; blk < nblocks; ++blk) {
            auto rd = [&](int g) { return *(const v4i*)(lds + rd_off[(g % 12) & 3] + ((g / 12) * (32 * RB) + 128 * ((g % 12) >> 2))); };
            v4i bq[2];
            bq[0] = rd(0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int tile = (g / 12) & 1, s = g % 12;
                if (g + 1 < NG && !(mode & 8)) bq[(g + 1) & 1] = rd(g + 1);
                const v4i b4 = bq[g & 1];
                const v8i b8 = { b4[0], b4[1], b4[2], b4[3], 0, 0, 0, 0 };
                const v8i a0 = { afrag[0][s][0], afrag[0][s][1], afrag[0][s][2], afrag[0][s][3], 0, 0, 0, 0 };
                const v8i a1 = { afrag[1][s][0], afrag[1][s][1], afrag[1][s][2], afrag[1][s][3], 0, 0, 0, 0 };
                v16f& c0 = tile == 0 ? accA0 : accB0;
                v16f& c1 = tile == 0 ? accA1 : accB1;
                v16f& p0 = tile == 0 ? accB0 : accA0;
                v16f& p1 = tile == 0 ? accB1 : accA1;
                if (s == 0) {
                    v16f z;
#pragma unroll
                    for (int i = 0; i < 16; ++i) z[i] = 0.0f;
                    c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a0, b8, z, 4, 4, 0, sa, 0, sb);
                    c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a1, b8, z, 4, 4, 0, sa, 0, sb);
                } else {
                    c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a0, b8, c0, 4, 4, 0, sa, 0, sb);
                    c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a1, b8, c1, 4, 4, 0, sa, 0, sb);
                    if (!(mode & 4)) { top2_of(p0, p1, 3 * (s - 1)); top2_of(p0, p1, 3 * (s - 1) + 1); top2_of(p0, p1, 3 * (s - 1) + 2); }
                }
                asm volatile("" : "+v"(c0), "+v"(c1));
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) rd_off[j] ^= BUF_STRIDE;
            if (!(mode & 2)) __syncthreads();
        }
#pragma unroll
        for (int at = 0; at < 2; ++at)
#pragma unroll
            for (int i = 0; i < 16; ++i) sum += best1[at][i] + best2[at][i];
        for (int i = 0; i < 16; ++i) sum += accA0[i] + accA1[i] + accB0[i] + accB1[i];
    } else {
        const int l15 = lane & 15, grp = lane >> 4;
        v4i afrag[4][6];
#pragma unroll
        for (int at = 0; at < 4; ++at)
#pragma unroll
            for (int s = 0; s < 6; ++s) { v4i a = data[(blockIdx.x * 7 + at * 6 + s) % 64 * 64 + lane]; asm volatile("" : "+v"(a)); afrag[at][s] = a; }
        float best1[4][4], best2[4][4];
#pragma unroll
        for (int at = 0; at < 4; ++at)
#pragma unroll
            for (int i = 0; i < 4; ++i) { best1[at][i] = 1.0e9f; best2[at][i] = 1.0e9f; }
        // lane (train row l15 of the 16-row tile, 32-value block grp of the K-step): chunk 4 s + grp of its row, swizzled as the kernel's rows are
        int rd_off[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) rd_off[j] = l15 * RB + 16 * ((4 * j + grp) ^ (l15 >> 1));
        v4f accA[4], accB[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) { accB[t][i] = 1.0e9f; accA[t][i] = 0.0f; }
        constexpr int NG = 24;            // 4 train tiles of 16 rows x 6 K-steps
        for (int blk = 0; blk < nblocks; ++blk) {
            auto rd = [&](int g) { return *(const v4i*)(lds + rd_off[(g % 6) & 1] + ((g / 6) * (16 * RB) + 128 * ((g % 6) >> 1))); };
            v4i bq[2];
            bq[0] = rd(0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int tile = (g / 6) & 1, s = g % 6;
                if (g + 1 < NG && !(mode & 8)) bq[(g + 1) & 1] = rd(g + 1);
                const v4i b4 = bq[g & 1];
                const v8i b8 = { b4[0], b4[1], b4[2], b4[3], 0, 0, 0, 0 };
                v4f* c = tile == 0 ? accA : accB;
                v4f* p = tile == 0 ? accB : accA;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const v8i a8 = { afrag[t][s][0], afrag[t][s][1], afrag[t][s][2], afrag[t][s][3], 0, 0, 0, 0 };
                    const v4f z = { 0.0f, 0.0f, 0.0f, 0.0f };
                    c[t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, s == 0 ? z : c[t], 4, 4, 0, sa, 0, sb);
                    // one top-2 update (two VALU ops, 8 issue cycles) behind each 16-cycle MFMA: six in a row leave the matrix pipe idle
                    if (!(mode & 4)) {
                        const int v = 3 * s + t;
                        if (t < 3 && v < 16) TOP2(best1[v >> 2][v & 3], best2[v >> 2][v & 3], p[v >> 2][v & 3]);
                        if (t == 3 && s == 5) TOP2(best1[3][3], best2[3][3], p[3][3]);
                    }
                    if (mode & 16) __builtin_amdgcn_sched_barrier(0);
                }
                asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]));
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) rd_off[j] ^= BUF_STRIDE;
            if (!(mode & 2)) __syncthreads();
        }
#pragma unroll
        for (int at = 0; at < 4; ++at)
#pragma unroll
            for (int i = 0; i < 4; ++i) sum += best1[at][i] + best2[at][i] + accA[at][i] + accB[at][i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = sum;
}

int main()
{
    std::vector<uint32_t> h(4096 * 4);
    srand(3);
    for (auto& w : h) { w = 0; for (int n = 0; n < 8; ++n) w |= (uint32_t)((rand() & 1) ? 0x2 : 0xA) << (4 * n); }
    v4i* d; float* o;
    const int grid = 1024;              // 4 rounds of one 8-wave workgroup per CU
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, (size_t)grid * 512 * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int nblocks = 600;            // 64-row stages per workgroup (the C4 chunk: 80)
    const char* what[] = { "full loop", "full loop, 16: top-2 pinned per MFMA", "no barriers", "no top-2 updates", "no fragment reads after the first" };
    const int modes[] = { 0, 16, 2, 4, 8 };
    for (int mi = 0; mi < 5; ++mi)
        for (int rep = 0; rep < 3; ++rep) {
            float ms32 = 0, ms16 = 0;
            for (int shape : { 32, 16 }) {
                hipEventRecord(e0);
#define LAUNCH(M) case M: if (shape == 32) hipLaunchKernelGGL((k_loop<32, M>), dim3(grid), dim3(512), 0, 0, d, o, nblocks); \
                          else hipLaunchKernelGGL((k_loop<16, M>), dim3(grid), dim3(512), 0, 0, d, o, nblocks); break;
                switch (modes[mi]) { LAUNCH(0) LAUNCH(16) LAUNCH(2) LAUNCH(4) LAUNCH(8) }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                (shape == 32 ? ms32 : ms16) = ms;
            }
            const double flops = (double)grid * 8 * nblocks * 64.0 * 64.0 * 768.0 * 2.0;
            printf("%-34s 32x32x64: %.3f ms %.2f PFLOP/s | 16x16x128: %.3f ms %.2f PFLOP/s | ratio %.3f\n", what[mi], ms32, flops / ms32 / 1e12, ms16, flops / ms16 / 1e12, ms16 / ms32);
        }
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("error: %s\n", hipGetErrorString(e)); return 1; }
    return 0;
}
