// Store-ceiling check against MI355X_MICROARCH.md's plain-store figure (6.0-6.2 TB/s: one dword per lane, 256 B per
// wave-instruction, 8 waves per CU).  A 400 MB (and a 1.6 GB) write-only stream in several shapes:
//   lin1 / lin4   grid-stride linear stream, 4 B / 16 B per lane, persistent grids of 2 / 4 / 8 / 16 waves per SIMD-set
//   row1          each wave sweeps whole 40,000-B rows with dword stores (the guide's shape on our matrix)
//   tile          the distance-matrix kernel's tiling: 128 x 128 tiles, a wave stores R rows x (256 / R) B per instruction
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void lin1(float* p, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) p[i] = (float)i;
}
__global__ __launch_bounds__(256) void lin4(v4f* p, size_t n4)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n4; i += stride) { v4f v = { 1.f, 2.f, 3.f, (float)i }; p[i] = v; }
}
// chunked linear: each workgroup owns a contiguous chunk (what a tile kernel's rows look like to the memory system)
__global__ __launch_bounds__(256) void chunk4(v4f* p, size_t n4, size_t per)
{
    const size_t b = (size_t)blockIdx.x * per, e = b + per < n4 ? b + per : n4;
    for (size_t i = b + threadIdx.x; i < e; i += 256) { v4f v = { 1.f, 2.f, 3.f, (float)i }; p[i] = v; }
}
__global__ __launch_bounds__(256) void row1(float* p, int rows, int cols, size_t ld)
{
    const int lane = threadIdx.x & 63, w = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    for (int r = w; r < rows; r += nw)
        for (int c = lane; c < cols; c += 64) p[(size_t)r * ld + c] = (float)c;
}
template <int R, int TW>       // R rows per instruction; tile TW columns wide, 128 rows high (4 waves x 32 rows)
__global__ __launch_bounds__(256) void tile(float* p, int rows, int cols, size_t ld)
{
    constexpr int LPR = 64 / R, W = LPR * 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q0 = blockIdx.x * 128 + wave * 32, t0 = blockIdx.y * TW;
    for (int c0 = 0; c0 < TW; c0 += W)
        for (int r0 = 0; r0 < 32; r0 += R) {
            const int row = q0 + r0 + lane / LPR, col = t0 + c0 + (lane % LPR) * 4;
            if (row < rows && col + 3 < cols) { v4f v = { 1.f, 2.f, 3.f, (float)col }; *(v4f*)(p + (size_t)row * ld + col) = v; }
        }
}
// the same 128 x 128 tiles, dealt so that ONE XCD (linear workgroup id % 8) walks along a band of 128 rows: horizontally
// adjacent tiles -- which share the 128-B lines that straddle their common edge when rows are not line-aligned -- pass
// through the same L2 one after the other
template <int R>
__global__ __launch_bounds__(256) void tile_xcd(float* p, int rows, int cols, size_t ld, int nqb, int ntb)
{
    constexpr int LPR = 64 / R, W = LPR * 4;
    const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
    const int qb = (slot / ntb) * 8 + xcd, tb = slot % ntb;
    if (qb >= nqb) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q0 = qb * 128 + wave * 32, t0 = tb * 128;
    for (int c0 = 0; c0 < 128; c0 += W)
        for (int r0 = 0; r0 < 32; r0 += R) {
            const int row = q0 + r0 + lane / LPR, col = t0 + c0 + (lane % LPR) * 4;
            if (row < rows && col + 3 < cols) { v4f v = { 1.f, 2.f, 3.f, (float)col }; *(v4f*)(p + (size_t)row * ld + col) = v; }
        }
}
// torch's fill kernel shape (vectorized_elementwise_kernel: 256 threads x one 16-byte store = a 4 KB chunk per workgroup)
__global__ __launch_bounds__(256) void fill4k(v4f* p, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) { v4f v = { 1.f, 1.f, 1.f, 1.f }; p[i] = v; }
}
// distance-matrix tiling with the row-parity trick of distmat_i8_kernel for strides of 16 mod 32 floats: a workgroup takes the rows
// of one parity out of 256 and odd ones shift their column window by 16, so every 512-B segment starts on a line boundary
template <int R>
__global__ __launch_bounds__(256) void tile_par(float* p, int rows, int cols, size_t ld, int nqb, int ntb)
{
    constexpr int LPR = 64 / R, W = LPR * 4;
    const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
    const int qb = (slot / ntb) * 8 + xcd, tb = slot % ntb;
    if (qb >= nqb) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qbase = (qb >> 1) * 256 + (qb & 1), t0 = tb * 128 - ((qb & 1) ? 16 : 0);
    for (int c0 = 0; c0 < 128; c0 += W)
        for (int r0 = 0; r0 < 32; r0 += R) {
            const int row = qbase + 2 * (wave * 32 + r0 + lane / LPR), col = t0 + c0 + (lane % LPR) * 4;
            if (row < rows && col >= 0 && col + 3 < cols) { v4f v = { 1.f, 2.f, 3.f, (float)col }; *(v4f*)(p + (size_t)row * ld + col) = v; }
        }
}
// S consecutive 4 KB pieces per workgroup, one 16-byte store per thread and piece (S = 1: fill4k)
template <int S>
__global__ __launch_bounds__(256) void fillS(v4f* p, size_t n4)
{
    const size_t base = (size_t)blockIdx.x * 256 * S + threadIdx.x;
#pragma unroll
    for (int s = 0; s < S; ++s) { const size_t i = base + (size_t)s * 256; if (i < n4) { v4f v = { 1.f, 1.f, 1.f, (float)s }; p[i] = v; } }
}
// one 16-byte store per thread, tile shaped: a workgroup writes 8 rows x 512 B (R8) or 32 rows x 128 B (R32) of the 10000 x 10000 matrix
template <int ROWS>
__global__ __launch_bounds__(256) void tile1(float* p, int rows, int cols, size_t ld)
{
    constexpr int TPR = 256 / ROWS;                  // threads per row
    const int row = blockIdx.y * ROWS + threadIdx.x / TPR, col = (blockIdx.x * TPR + threadIdx.x % TPR) * 4;
    if (row < rows && col + 3 < cols) { v4f v = { 1.f, 2.f, 3.f, (float)col }; *(v4f*)(p + (size_t)row * ld + col) = v; }
}
// round 4: short-and-wide workgroup tiles under the XCD-banded mapping (one XCD walks along a band of ROWS rows): ROWS rows x SEGF floats
// per workgroup, consecutive threads along a row (16-byte stores), a wave covering 1 KB of one row per instruction
template <int ROWS, int SEGF>
__global__ __launch_bounds__(256) void tile_band(float* p, int rows, int cols, size_t ld, int nqb, int ntb)
{
    const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
    const int qb = (slot / ntb) * 8 + xcd, tb = slot % ntb;
    if (qb >= nqb) return;
    constexpr int TPR = SEGF / 4;                    // threads (16 B each) per row segment
    for (int e = threadIdx.x; e < ROWS * TPR; e += 256) {
        const int row = qb * ROWS + e / TPR, col = tb * SEGF + (e % TPR) * 4;
        if (row < rows && col + 3 < cols) { v4f v = { 1.f, 2.f, 3.f, (float)col }; *(v4f*)(p + (size_t)row * ld + col) = v; }
    }
}
int main(int argc, char** argv)
{
    if (argc > 1) {
        // sustained mode: every shape 300 launches back to back after 50 warm-up launches, mean of blocks of 100
        const size_t n = 100000000; float* p; hipMalloc(&p, (n + 4000000) * 4);
        hipEvent_t ev[4]; for (auto& e : ev) hipEventCreate(&e);
        auto run = [&](const char* name, auto f) {
            for (int i = 0; i < 50; ++i) f();
            for (int b = 0; b < 3; ++b) { hipEventRecord(ev[b]); for (int i = 0; i < 100; ++i) f(); }
            hipEventRecord(ev[3]); hipEventSynchronize(ev[3]);
            float ms[3]; for (int b = 0; b < 3; ++b) hipEventElapsedTime(&ms[b], ev[b], ev[b + 1]);
            printf("%-52s %6.1f %6.1f %6.1f us per launch (blocks of 100)  -> %.2f TB/s\n", name, ms[0] * 10, ms[1] * 10, ms[2] * 10, 4e8 / (ms[2] * 10 * 1e-6) / 1e12);
        };
        for (int rep = 0; rep < 2; ++rep) {
            run("fill4k (torch fill shape: 97,657 x 4 KB)", [&] { hipLaunchKernelGGL(fill4k, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, 0, (v4f*)p, n / 4); });
            run("fillS<2>: 8 KB per workgroup, 2 stores per thread", [&] { hipLaunchKernelGGL(fillS<2>, dim3((unsigned)((n / 4 + 511) / 512)), dim3(256), 0, 0, (v4f*)p, n / 4); });
            run("fillS<4>: 16 KB per workgroup", [&] { hipLaunchKernelGGL(fillS<4>, dim3((unsigned)((n / 4 + 1023) / 1024)), dim3(256), 0, 0, (v4f*)p, n / 4); });
            run("fillS<16>: 64 KB per workgroup", [&] { hipLaunchKernelGGL(fillS<16>, dim3((unsigned)((n / 4 + 4095) / 4096)), dim3(256), 0, 0, (v4f*)p, n / 4); });
            run("chunk4 65536 workgroups of 5 KB (loop)", [&] { hipLaunchKernelGGL(chunk4, dim3(65536), dim3(256), 0, 0, (v4f*)p, n / 4, (n / 4 + 65535) / 65536); });
            run("tile1<8>: 8 rows x 512 B per workgroup, 1 store per thread, ld 10000", [&] { hipLaunchKernelGGL(tile1<8>, dim3(79, 1250), dim3(256), 0, 0, p, 10000, 10000, (size_t)10000); });
            run("tile1<32>: 32 rows x 128 B per workgroup, ld 10000", [&] { hipLaunchKernelGGL(tile1<32>, dim3(313, 313), dim3(256), 0, 0, p, 10000, 10000, (size_t)10000); });
            run("tile1<8>, ld 10016", [&] { hipLaunchKernelGGL(tile1<8>, dim3(79, 1250), dim3(256), 0, 0, p, 10000, 10000, (size_t)10016); });
            run("tile1<1>: 1 row x 4 KB per workgroup, ld 10000", [&] { hipLaunchKernelGGL(tile1<1>, dim3(10, 10000), dim3(256), 0, 0, p, 10000, 10000, (size_t)10000); });
            run("lin4 grid 16384", [&] { hipLaunchKernelGGL(lin4, dim3(16384), dim3(256), 0, 0, (v4f*)p, n / 4); });
            run("lin1 grid 4096", [&] { hipLaunchKernelGGL(lin1, dim3(4096), dim3(256), 0, 0, p, n); });
            run("tile 128x128 XCD-banded 8 rows x 128 B, ld 10016", [&] { hipLaunchKernelGGL((tile_xcd<8>), dim3(80 * 79), dim3(256), 0, 0, p, 10000, 10000, (size_t)10016, 79, 79); });
            run("tile 128x128 XCD-banded 8 rows x 128 B, ld 10000", [&] { hipLaunchKernelGGL((tile_xcd<8>), dim3(80 * 79), dim3(256), 0, 0, p, 10000, 10000, (size_t)10000, 79, 79); });
            run("tile 128x128 row-parity 8 rows x 128 B, ld 10000", [&] { hipLaunchKernelGGL((tile_par<8>), dim3(80 * 80), dim3(256), 0, 0, p, 10000, 10000, (size_t)10000, 80, 80); });
            run("tile 128x10000 row bands 1 row x 1 KB, ld 10000", [&] { hipLaunchKernelGGL((tile<1, 10000>), dim3(79, 1), dim3(256), 0, 0, p, 10000, 10000, (size_t)10000); });
            run("row1 grid 2048, ld 10000", [&] { hipLaunchKernelGGL(row1, dim3(2048), dim3(256), 0, 0, p, 10000, 10000, (size_t)10000); });
#define BAND(R, F) do { const int nqb = (10000 + R - 1) / R, ntb = (10000 + F - 1) / F; \
            run("band tile " #R " rows x " #F " floats, XCD-banded, ld 10000", [&] { hipLaunchKernelGGL((tile_band<R, F>), dim3(((nqb + 7) / 8) * 8 * ntb), dim3(256), 0, 0, p, 10000, 10000, (size_t)10000, nqb, ntb); }); } while (0)
            BAND(128, 128); BAND(64, 256); BAND(32, 512); BAND(16, 1024); BAND(8, 2048); BAND(32, 1024); BAND(16, 2048); BAND(4, 4096); BAND(32, 128); BAND(64, 128);
        }
        return 0;
    }
    const size_t nbig = 400000000; float* p; hipMalloc(&p, (nbig + 4000000) * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto t = [&](const char* name, double bytes, auto f) { for (int i = 0; i < 3; ++i) f(); hipEventRecord(a); for (int i = 0; i < 20; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); printf("%-44s %7.1f us  %.2f TB/s\n", name, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12); };
    char nm[128];
    for (size_t n : { (size_t)100000000, nbig }) {
        printf("---- %.1f GB\n", n * 4 / 1e9);
        for (int g : { 512, 1024, 2048, 4096, 16384 }) {
            snprintf(nm, sizeof nm, "lin1 grid %5d (%d waves/CU)", g, g * 4 / 256);
            t(nm, n * 4.0, [&] { hipLaunchKernelGGL(lin1, dim3(g), dim3(256), 0, 0, p, n); });
            snprintf(nm, sizeof nm, "lin4 grid %5d", g);
            t(nm, n * 4.0, [&] { hipLaunchKernelGGL(lin4, dim3(g), dim3(256), 0, 0, (v4f*)p, n / 4); });
        }
        for (int g : { 2048, 6241, 16384, 65536 }) {
            snprintf(nm, sizeof nm, "chunk4 %5d workgroups of %zu KB", g, (n * 4 / g) >> 10);
            const size_t per = (n / 4 + g - 1) / g;
            t(nm, n * 4.0, [&] { hipLaunchKernelGGL(chunk4, dim3(g), dim3(256), 0, 0, (v4f*)p, n / 4, per); });
        }
    }
    for (size_t ld : { (size_t)10000, (size_t)10016 }) {
        printf("---- 10000 x 10000 floats, row stride %zu\n", ld);
        for (int g : { 512, 2048 }) { snprintf(nm, sizeof nm, "row1 grid %d", g); t(nm, 4e8, [&] { hipLaunchKernelGGL(row1, dim3(g), dim3(256), 0, 0, p, 10000, 10000, ld); }); }
        t("tile 128x128, 8 rows x 128 B", 4e8, [&] { hipLaunchKernelGGL((tile<8, 128>), dim3(79, 79), dim3(256), 0, 0, p, 10000, 10000, ld); });
        t("tile 128x128 XCD-banded, 8 rows x 128 B", 4e8, [&] { hipLaunchKernelGGL((tile_xcd<8>), dim3(80 * 79), dim3(256), 0, 0, p, 10000, 10000, ld, 79, 79); });
        t("tile 128x128 XCD-banded, 4 rows x 256 B", 4e8, [&] { hipLaunchKernelGGL((tile_xcd<4>), dim3(80 * 79), dim3(256), 0, 0, p, 10000, 10000, ld, 79, 79); });
        t("tile 128x256, 4 rows x 256 B", 4e8, [&] { hipLaunchKernelGGL((tile<4, 256>), dim3(79, 40), dim3(256), 0, 0, p, 10000, 10000, ld); });
        t("tile 128x512, 2 rows x 512 B", 4e8, [&] { hipLaunchKernelGGL((tile<2, 512>), dim3(79, 20), dim3(256), 0, 0, p, 10000, 10000, ld); });
        t("tile 128x1024, 1 row x 1 KB", 4e8, [&] { hipLaunchKernelGGL((tile<1, 1024>), dim3(79, 10), dim3(256), 0, 0, p, 10000, 10000, ld); });
        t("tile 128x2560, 1 row x 1 KB", 4e8, [&] { hipLaunchKernelGGL((tile<1, 2560>), dim3(79, 4), dim3(256), 0, 0, p, 10000, 10000, ld); });
        t("tile 128x10000, 1 row x 1 KB (row bands)", 4e8, [&] { hipLaunchKernelGGL((tile<1, 10000>), dim3(79, 1), dim3(256), 0, 0, p, 10000, 10000, ld); });
    }
    return 0;
}
