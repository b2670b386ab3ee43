// Stand-alone check + timing of the chain solver (csrc/ba_chain.hpp) on the GPU: random band + border systems of the
// benchmark shapes (C3: 49 free cameras, C4: 199, C5: 999; w = 3, 4 intrinsics), compared with a dense Cholesky on the
// host (shapes up to 1300 unknowns), timed as back-to-back launches with HIP events.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o chain_bench chain_bench.hip && ./chain_bench [P a]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "../sfm_opencv_amd/csrc/ba_chain.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static int run(int ncf, int w, int nk, int force_P, int force_a, int force_G, int reps)
{
    const int n = 6 * ncf + nk, npad = (n + 31) / 32 * 32, ld = npad;
    ChainArgs A; memset(&A, 0, sizeof A);
    if (!chain_plan(A, ncf, w, nk, ld, npad, force_P, force_a, force_G, (size_t)160 << 10)) { printf("ncf %d: no plan\n", ncf); return 1; }
    std::mt19937_64 rng(1234 + ncf);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    std::vector<double> S((size_t)npad * ld, 0.0), rhs(npad, 0.0), diagU(npad, 0.0);
    for (int i = 0; i < ncf; ++i)
        for (int rep = 0; rep < 3; ++rep) {
            const int span = (int)(rng() % (unsigned)(w + 1));
            std::vector<std::pair<int, double>> row;
            for (int c = i; c <= i + span && c < ncf; ++c) for (int k = 0; k < 6; ++k) row.push_back({ 6 * c + k, U(rng) });
            for (int k = 0; k < nk; ++k) row.push_back({ 6 * ncf + k, 0.3 * U(rng) });
            const double res = U(rng);
            for (auto& a : row) { for (auto& b : row) S[(size_t)a.first * ld + b.first] += a.second * b.second; rhs[a.first] += a.second * res; diagU[a.first] += a.second * a.second; }
        }
    const double radius = 1e2, dmin = 1e-6, dmax = 1e32;
    std::vector<double> yref;
    if (n <= 1300) {
        std::vector<double> Lm((size_t)n * n);
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Lm[(size_t)i * n + j] = S[(size_t)i * ld + j];
        for (int i = 0; i < n; ++i) Lm[(size_t)i * n + i] += std::min(std::max(diagU[i], dmin), dmax) / radius;
        for (int j = 0; j < n; ++j) {
            double d = Lm[(size_t)j * n + j];
            for (int k = 0; k < j; ++k) d -= Lm[(size_t)j * n + k] * Lm[(size_t)j * n + k];
            const double l = std::sqrt(d); Lm[(size_t)j * n + j] = l;
            for (int i = j + 1; i < n; ++i) { double s = Lm[(size_t)i * n + j]; for (int k = 0; k < j; ++k) s -= Lm[(size_t)i * n + k] * Lm[(size_t)j * n + k]; Lm[(size_t)i * n + j] = s / l; }
        }
        std::vector<double> z(n); yref.resize(n);
        for (int i = 0; i < n; ++i) { double s = rhs[i]; for (int k = 0; k < i; ++k) s -= Lm[(size_t)i * n + k] * z[k]; z[i] = s / Lm[(size_t)i * n + i]; }
        for (int i = n - 1; i >= 0; --i) { double s = z[i]; for (int k = i + 1; k < n; ++k) s -= Lm[(size_t)k * n + i] * yref[k]; yref[i] = s / Lm[(size_t)i * n + i]; }
    }
    double *dS, *drhs, *ddiag, *drec, *dimg, *dy; int* derr;
    CK(hipMalloc(&dS, S.size() * 8)); CK(hipMalloc(&drhs, npad * 8)); CK(hipMalloc(&ddiag, npad * 8));
    CK(hipMalloc(&drec, (size_t)ncf * A.rec_stride * 8)); CK(hipMalloc(&dimg, ((size_t)(A.P >> A.a) * A.img_doubles + 8) * 8)); CK(hipMalloc(&dy, npad * 8)); CK(hipMalloc(&derr, 4));
    CK(hipMemcpy(dS, S.data(), S.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(drhs, rhs.data(), npad * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(ddiag, diagU.data(), npad * 8, hipMemcpyHostToDevice)); CK(hipMemset(derr, 0, 4)); CK(hipMemset(dy, 0xff, npad * 8));
    A.S = dS; A.rhs = drhs; A.diagU = ddiag; A.inv_radius = 1.0 / radius; A.dmin = dmin; A.dmax = dmax; A.rec = drec; A.img = dimg; A.y = dy; A.err = derr;
    const size_t lds1 = 8 * ch_sub_lds(A.w, A.BB, A.a, A.G, A.n, A.a == A.m), lds2 = A.a < A.m ? 8 * ch_top_lds(A.w, A.BB, A.m - A.a, A.nw_top, A.n) : 0;
    CK(hipFuncSetAttribute((const void*)chain_sub_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
    if (lds2) CK(hipFuncSetAttribute((const void*)chain_top_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    auto launch = [&] {
        hipLaunchKernelGGL(chain_sub_kernel, dim3(A.P >> A.a), dim3(64 * A.G << A.a), lds1, 0, A);
        if (lds2) hipLaunchKernelGGL(chain_top_kernel, dim3(1), dim3(64 * A.nw_top), lds2, 0, A);
    };
    launch();
    CK(hipDeviceSynchronize());
    std::vector<double> y(npad); int err = 0;
    CK(hipMemcpy(y.data(), dy, npad * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&err, derr, 4, hipMemcpyDeviceToHost));
    double emax = 0, ymax = 0;
    if (!yref.empty()) for (int i = 0; i < n; ++i) { emax = std::max(emax, std::fabs(y[i] - yref[i])); ymax = std::max(ymax, std::fabs(yref[i])); if (y[i] != y[i]) emax = 1e300; }
    std::vector<double> y2(npad);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) launch();
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(y2.data(), dy, npad * 8, hipMemcpyDeviceToHost));
#ifdef CH_STAMPS
    {
        long long* dst; CK(hipMalloc(&dst, 512 * 8)); CK(hipMemset(dst, 0, 512 * 8));
        A.stamps = dst; launch(); CK(hipDeviceSynchronize()); A.stamps = nullptr;
        std::vector<long long> st(512); CK(hipMemcpy(st.data(), dst, 512 * 8, hipMemcpyDeviceToHost)); (void)hipFree(dst);
        for (int base = 0; base < 512; base += 128) {
            if (!st[base]) continue;
            printf("  stamps of kernel %d (s_memtime ticks since its start; delta):", base / 128);
            for (int i = base; i < base + 128 && st[i]; ++i) printf(" %lld(+%lld)", st[i] - st[base], i > base ? st[i] - st[i - 1] : 0ll);
            printf("\n");
        }
    }
#endif
    const bool same = memcmp(y.data(), y2.data(), npad * 8) == 0;
    printf("ncf %4d w %d nk %d: P %2d a %d G %d top levels %d, LDS %zu / %zu B: %.2f us per solve (%d back-to-back), max |y - y_ref| = %.3e (|y| <= %.2e) err %d, rerun bitwise %s\n",
           ncf, w, nk, A.P, A.a, A.G, A.m - A.a, lds1, lds2, ms * 1000.0 / reps, reps, emax, ymax, err, same ? "identical" : "DIFFERENT");
    (void)hipFree(dS); (void)hipFree(drhs); (void)hipFree(ddiag); (void)hipFree(drec); (void)hipFree(dimg); (void)hipFree(dy); (void)hipFree(derr);
    return (err == 0 && (yref.empty() || emax <= 1e-9 * (1 + ymax)) && same) ? 0 : 1;
}

int main(int argc, char** argv)
{
    const int fP = argc > 1 ? atoi(argv[1]) : 0, fa = argc > 2 ? atoi(argv[2]) : -1, fG = argc > 3 ? atoi(argv[3]) : 0;
    int bad = 0;
    if (argc > 4) { bad += run(atoi(argv[4]), 3, 4, fP, fa, fG, 200); return bad; }
    bad += run(6, 3, 4, fP, fa, fG, 200);
    bad += run(49, 3, 4, fP, fa, fG, 200);
    bad += run(199, 3, 4, fP, fa, fG, 200);
    bad += run(999, 3, 4, fP, fa, fG, 200);
    if (fP == 0) {
        const int shapes[][4] = { { 6, 1, 0, 1 }, { 6, 1, 0, 2 }, { 49, 2, 1, 4 }, { 49, 4, 1, 2 }, { 49, 4, 2, 1 }, { 49, 4, 0, 4 }, { 199, 8, 1, 4 }, { 199, 16, 2, 2 }, { 199, 16, 1, 2 }, { 199, 32, 2, 2 }, { 199, 16, 2, 1 }, { 999, 16, 1, 4 }, { 999, 32, 2, 1 } };
        for (auto& s : shapes) bad += run(s[0], 3, 4, s[1], s[2], s[3], 200);
        bad += run(60, 4, 4, 0, -1, 0, 100);
        bad += run(50, 2, 0, 0, -1, 0, 100);
        bad += run(50, 1, 4, 0, -1, 0, 100);
    }
    printf(bad ? "FAILED (%d)\n" : "all ok\n", bad);
    return bad;
}
