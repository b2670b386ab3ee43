// sfm_synth.cpp -- the synthetic workloads of SURVEY 8d in C++ on std::mt19937_64 (seed 20240607 + image index for the descriptor
// chains, 20240607 for the track scene), built as libsfmsynth.so for bench.py and the tests (plain C ABI, host only).
//
// The engine is the standard's (its output sequence is fixed by the C++ standard); the transforms on top of it are written out here
// -- 53-bit uniforms, Box-Muller normals, Fisher-Yates permutations, Lemire-free modulo-rejection integers -- because the standard
// leaves std::normal_distribution / std::uniform_int_distribution to the implementation and these files must generate the same
// numbers wherever they are compiled.  Same constructions as sfm_opencv_amd/synth.py (which keeps numpy's PCG64 streams for the
// tests' fixed expectations); no reference counterpart: the reference ships no generator, only datasets.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <random>
#include <vector>

namespace {

struct Rng {
    std::mt19937_64 e;
    bool have = false; double spare = 0.0;
    explicit Rng(uint64_t seed) : e(seed) {}
    double uniform() { return (double)(e() >> 11) * (1.0 / 9007199254740992.0); }          // [0, 1), 53 bits
    double normal()
    {
        if (have) { have = false; return spare; }
        double u1;
        do u1 = uniform(); while (u1 <= 0.0);
        const double u2 = uniform(), r = std::sqrt(-2.0 * std::log(u1)), a = 6.283185307179586476925 * u2;
        spare = r * std::sin(a); have = true;
        return r * std::cos(a);
    }
    uint64_t below(uint64_t n)                      // unbiased integer in [0, n)
    {
        const uint64_t lim = UINT64_MAX - UINT64_MAX % n;
        uint64_t x;
        do x = e(); while (x >= lim);
        return x % n;
    }
    void permutation(std::vector<int>& p, int n)
    {
        p.resize((size_t)n); std::iota(p.begin(), p.end(), 0);
        for (int i = n - 1; i > 0; --i) std::swap(p[(size_t)i], p[(size_t)below((uint64_t)i + 1)]);
    }
};

// OpenCV-SIFT-shaped row: |N(0,1)|, L2-normalise, clip 0.2, renormalise, min(255, floor(512 v))
void sift_like(Rng& g, int dim, float* row)
{
    std::vector<double> v((size_t)dim);
    double s = 0.0;
    for (int k = 0; k < dim; ++k) { v[k] = std::fabs(g.normal()); s += v[k] * v[k]; }
    s = std::sqrt(s);
    double s2 = 0.0;
    for (int k = 0; k < dim; ++k) { v[k] = std::fmin(v[k] / s, 0.2); s2 += v[k] * v[k]; }
    s2 = std::sqrt(s2);
    for (int k = 0; k < dim; ++k) row[k] = (float)std::fmin(255.0, std::floor(512.0 * v[k] / s2));
}

}  // namespace

extern "C" {

// n_img matrices of n_desc x dim float32 (integer-valued in [0, 255]), concatenated in `out`: image i + 1 = a random permutation of
// round(overlap n_desc) rows copied from image i with U{-2..2} integer noise (clipped) plus fresh rows; RNG seeded seed + i per image
int sfmsynth_sift_chain(int n_img, int n_desc, int dim, uint64_t seed, double overlap, float* out)
{
    if (n_img < 0 || n_desc < 0 || dim < 1 || !out) return -1;
    const size_t per = (size_t)n_desc * dim;
    const int n_copy = (int)std::llround(overlap * n_desc);
    std::vector<float> tmp(per);
    std::vector<int> perm;
    for (int i = 0; i < n_img; ++i) {
        Rng g(seed + (uint64_t)i);
        float* cur = out + (size_t)i * per;
        if (i == 0) { for (int r = 0; r < n_desc; ++r) sift_like(g, dim, cur + (size_t)r * dim); continue; }
        const float* prev = out + (size_t)(i - 1) * per;
        g.permutation(perm, n_desc);
        for (int r = 0; r < n_copy; ++r)
            for (int k = 0; k < dim; ++k) {
                const double v = (double)prev[(size_t)perm[(size_t)r] * dim + k] + (double)((int)g.below(5) - 2);
                tmp[(size_t)r * dim + k] = (float)std::fmin(255.0, std::fmax(0.0, v));
            }
        for (int r = n_copy; r < n_desc; ++r) sift_like(g, dim, &tmp[(size_t)r * dim]);
        g.permutation(perm, n_desc);
        for (int r = 0; r < n_desc; ++r) std::memcpy(cur + (size_t)r * dim, &tmp[(size_t)perm[(size_t)r] * dim], (size_t)dim * sizeof(float));
    }
    return 0;
}

// binary rows (AKAZE M-LDB shape): nbytes random bytes; copies get `flip` of their bits flipped; RNG seeded seed + 7919 + i
int sfmsynth_akaze_chain(int n_img, int n_desc, int nbytes, uint64_t seed, double overlap, double flip, uint8_t* out)
{
    if (n_img < 0 || n_desc < 0 || nbytes < 1 || !out) return -1;
    const size_t per = (size_t)n_desc * nbytes;
    const int n_copy = (int)std::llround(overlap * n_desc);
    std::vector<uint8_t> tmp(per);
    std::vector<int> perm;
    for (int i = 0; i < n_img; ++i) {
        Rng g(seed + 7919ull + (uint64_t)i);
        uint8_t* cur = out + (size_t)i * per;
        auto fresh = [&](uint8_t* row) { for (int k = 0; k < nbytes; k += 8) { uint64_t w = g.e(); for (int b = 0; b < 8 && k + b < nbytes; ++b) row[k + b] = (uint8_t)(w >> (8 * b)); } };
        if (i == 0) { for (int r = 0; r < n_desc; ++r) fresh(cur + (size_t)r * nbytes); continue; }
        const uint8_t* prev = out + (size_t)(i - 1) * per;
        g.permutation(perm, n_desc);
        for (int r = 0; r < n_copy; ++r)
            for (int k = 0; k < nbytes; ++k) {
                uint8_t m = 0;
                for (int b = 0; b < 8; ++b) if (g.uniform() < flip) m |= (uint8_t)(1u << b);
                tmp[(size_t)r * nbytes + k] = prev[(size_t)perm[(size_t)r] * nbytes + k] ^ m;
            }
        for (int r = n_copy; r < n_desc; ++r) fresh(&tmp[(size_t)r * nbytes]);
        g.permutation(perm, n_desc);
        for (int r = 0; r < n_desc; ++r) std::memcpy(cur + (size_t)r * nbytes, &tmp[(size_t)perm[(size_t)r] * nbytes], (size_t)nbytes);
    }
    return 0;
}

// Ring scene of SURVEY 8d (the construction of synth.ba_scene): cameras on a ring of radius 10 (+-1 radial, +-2 height modulation)
// looking at the origin, points uniform in a radius-3 ball, each seen by L ~ U{min_len..max_len} CONSECUTIVE cameras, pixel noise,
// gross outliers, perturbed start (camera 0 exact).  Two calls: with obs_* == NULL it only returns the observation count.
// Observations are sorted by (camera, point), the order bundle_adjustment adds residual blocks (NView:1187-1197).
// Outputs: K_true[4], ext_true[6 n_cam], pts_true[3 n_pt], K0[4], ext0[6 n_cam], pts0[3 n_pt], obs_cam / obs_pt [n_obs], obs_uv[2 n_obs].
long long sfmsynth_ba_scene(int n_cam, int n_pt, uint64_t seed, double noise_px, double outlier_frac, int min_len, int max_len,
                            double* K_true, double* ext_true, double* pts_true, double* K0, double* ext0, double* pts0,
                            int32_t* obs_cam, int32_t* obs_pt, double* obs_uv)
{
    if (n_cam < 1 || n_pt < 0) return -1;
    const double KREF[4] = { 2826.561, 2826.519, 1835.259, 1370.103 };          // NViewReconstuct.cpp:1353-1356
    Rng g(seed);
    if (max_len > n_cam) max_len = n_cam;
    if (min_len > max_len) min_len = max_len;
    std::vector<double> ext((size_t)6 * n_cam), R9((size_t)9 * n_cam);
    for (int c = 0; c < n_cam; ++c) {
        const double phi = 6.283185307179586476925 * c / n_cam, rad = 10.0 + std::cos(5.0 * phi);
        const double C[3] = { rad * std::sin(phi), 2.0 * std::sin(3.0 * phi), -rad * std::cos(phi) };
        const double nC = std::sqrt(C[0] * C[0] + C[1] * C[1] + C[2] * C[2]);
        const double z[3] = { -C[0] / nC, -C[1] / nC, -C[2] / nC };
        double x[3] = { z[2], 0.0, -z[0] };                                       // (0, 1, 0) x z
        const double nx = std::sqrt(x[0] * x[0] + x[2] * x[2]);
        x[0] /= nx; x[2] /= nx;
        const double y[3] = { z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0] };
        double* R = &R9[(size_t)9 * c];
        for (int k = 0; k < 3; ++k) { R[k] = x[k]; R[3 + k] = y[k]; R[6 + k] = z[k]; }
        // Rodrigues log
        const double tr = std::fmin(1.0, std::fmax(-1.0, (R[0] + R[4] + R[8] - 1.0) / 2.0)), th = std::acos(tr);
        const double w[3] = { R[7] - R[5], R[2] - R[6], R[3] - R[1] };
        double aa[3];
        if (th < 1e-12) { for (int k = 0; k < 3; ++k) aa[k] = 0.5 * w[k]; }
        else if (3.141592653589793 - th < 1e-6) {
            double A[9]; for (int k = 0; k < 9; ++k) A[k] = (R[k] + (k % 4 == 0 ? 1.0 : 0.0)) / 2.0;
            int kk = 0; for (int k = 1; k < 3; ++k) if (A[4 * k] > A[4 * kk]) kk = k;
            const double d = std::sqrt(std::fmax(A[4 * kk], 0.0));
            double ax[3] = { A[3 * kk] / d, A[3 * kk + 1] / d, A[3 * kk + 2] / d };
            const double na = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
            for (int k = 0; k < 3; ++k) aa[k] = th * ax[k] / na;
        } else { for (int k = 0; k < 3; ++k) aa[k] = th * w[k] / (2.0 * std::sin(th)); }
        for (int k = 0; k < 3; ++k) { ext[(size_t)6 * c + k] = aa[k]; ext[(size_t)6 * c + 3 + k] = -(R[3 * k] * C[0] + R[3 * k + 1] * C[1] + R[3 * k + 2] * C[2]); }
    }
    std::vector<double> P((size_t)3 * n_pt);
    for (int p = 0; p < n_pt; ++p) {
        double d[3] = { g.normal(), g.normal(), g.normal() };
        const double nd = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), r = 3.0 * std::cbrt(g.uniform());
        for (int k = 0; k < 3; ++k) P[(size_t)3 * p + k] = d[k] / nd * r;
    }
    std::vector<int> L((size_t)n_pt), start((size_t)n_pt);
    long long n_obs = 0;
    for (int p = 0; p < n_pt; ++p) { L[p] = min_len + (int)g.below((uint64_t)(max_len - min_len + 1)); n_obs += L[p]; }
    for (int p = 0; p < n_pt; ++p) start[p] = (int)(g.uniform() * (n_cam - L[p] + 1));
    if (!obs_cam || !obs_pt || !obs_uv) return n_obs;
    // observations by (camera, point): counting sort over cameras, points ascending inside a camera
    std::vector<long long> cstart((size_t)n_cam + 1, 0);
    for (int p = 0; p < n_pt; ++p) for (int j = 0; j < L[p]; ++j) ++cstart[(size_t)start[p] + j + 1];
    for (int c = 0; c < n_cam; ++c) cstart[c + 1] += cstart[c];
    std::vector<long long> fill(cstart.begin(), cstart.end() - 1);
    for (int p = 0; p < n_pt; ++p)
        for (int j = 0; j < L[p]; ++j) { const int c = start[p] + j; const long long at = fill[c]++; obs_cam[at] = c; obs_pt[at] = p; }
    // pixels: ReprojectCost's forward model (NView:151-177) with the rotation matrices, + noise, + outliers -- drawn in observation order
    for (long long k = 0; k < n_obs; ++k) {
        const int c = obs_cam[k], p = obs_pt[k];
        const double* R = &R9[(size_t)9 * c]; const double* t = &ext[(size_t)6 * c + 3]; const double* X = &P[(size_t)3 * p];
        const double q[3] = { R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0], R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1], R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2] };
        obs_uv[2 * k] = KREF[0] * q[0] / q[2] + KREF[2] + noise_px * g.normal();
        obs_uv[2 * k + 1] = KREF[1] * q[1] / q[2] + KREF[3] + noise_px * g.normal();
    }
    for (long long k = 0; k < n_obs; ++k)
        if (g.uniform() < outlier_frac) { obs_uv[2 * k] += -50.0 + 100.0 * g.uniform(); obs_uv[2 * k + 1] += -50.0 + 100.0 * g.uniform(); }
    for (int k = 0; k < 4; ++k) { K_true[k] = KREF[k]; K0[k] = KREF[k] * 1.01; }
    std::memcpy(ext_true, ext.data(), ext.size() * sizeof(double)); std::memcpy(ext0, ext.data(), ext.size() * sizeof(double));
    std::memcpy(pts_true, P.data(), P.size() * sizeof(double));
    for (int c = 1; c < n_cam; ++c) { for (int k = 0; k < 3; ++k) ext0[(size_t)6 * c + k] += 0.01 * g.normal(); for (int k = 3; k < 6; ++k) ext0[(size_t)6 * c + k] += 0.05 * g.normal(); }
    for (size_t i = 0; i < P.size(); ++i) pts0[i] = P[i] + 0.05 * g.normal();
    return n_obs;
}

}  // extern "C"
