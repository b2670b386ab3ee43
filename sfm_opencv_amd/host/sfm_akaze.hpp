// sfm_akaze.hpp -- AKAZE key points + M-LDB binary descriptors for the drivers (host C++, CPU, header only).
//
// The reference's live extractor is cv::AKAZE::create() with its defaults (NViewReconstuct.cpp:797-812): MLDB descriptors of 486
// bits (61 bytes), 3 channels, threshold 0.001, 4 octaves x 4 sublevels, Perona-Malik g2 diffusivity; the rows go to
// BFMatcher(NORM_HAMMING2) (NView:876).  OpenCV is absent here, so this restates the PUBLISHED method [3P]: Alcantarilla, Nuevo,
// Bartoli, "Fast Explicit Diffusion for Accelerated Features in Nonlinear Scale Spaces" (BMVC 2013) and the structure of its
// reference implementation -- nonlinear scale space by FED cycles, scale-normalised determinant of the Hessian from Scharr-type
// derivative kernels of growing step, 3x3 extrema with suppression against the neighbouring sublevels, 2D sub-pixel fit, dominant
// orientation from a sliding pi/3 window, and the rotated 2x2 / 3x3 / 4x4 grid comparisons of (intensity, dx, dy) cell means.
// PARITY UNPINNED and un-pinnable: the reference ships no key points or descriptors; accepted on behaviour (tests/test_features_cpu.py:
// repeatability and Hamming matching under a known similarity, 61-byte rows, the two padding bits clear).
#pragma once
// included by sfm_features.hpp (after the image / Gaussian helpers it builds on); include that header, not this one

namespace sfm {
namespace akaze {

using sift::Gray;

struct Params {
    int omax = 4, nsublevels = 4;
    float soffset = 1.6f, derivative_factor = 1.5f, dthreshold = 0.001f, kperc = 0.7f;
    int knbins = 300, pattern_size = 10;
};

struct Level {
    Gray Lt, Lsmooth, Lx, Ly, Ldet;
    float esigma = 0, etime = 0;
    int octave = 0, sublevel = 0, sigma_size = 0;
    std::vector<float> tau;
};

inline int fround(float v) { return (int)(v + (v >= 0 ? 0.5f : -0.5f)); }
inline int reflect101(int i, int n) { if (n == 1) return 0; while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i; return i; }

// cv::Scharr(src, dst, CV_32F, dx, dy): [-3 0 3; -10 0 10; -3 0 3] (not normalised), BORDER_REFLECT_101
inline void scharr(const Gray& s, Gray& gx, Gray& gy)
{
    gx = s; gy = s;
    const int H = s.rows, W = s.cols;
    for (int y = 0; y < H; ++y) {
        const int ym = reflect101(y - 1, H), yp = reflect101(y + 1, H);
        for (int x = 0; x < W; ++x) {
            const int xm = reflect101(x - 1, W), xp = reflect101(x + 1, W);
            gx.at(y, x) = 3.0f * (s.at(ym, xp) - s.at(ym, xm)) + 10.0f * (s.at(y, xp) - s.at(y, xm)) + 3.0f * (s.at(yp, xp) - s.at(yp, xm));
            gy.at(y, x) = 3.0f * (s.at(yp, xm) - s.at(ym, xm)) + 10.0f * (s.at(yp, x) - s.at(ym, x)) + 3.0f * (s.at(yp, xp) - s.at(ym, xp));
        }
    }
}

// contrast factor: the kperc percentile of the gradient magnitudes of the (sigma = 1) smoothed image
inline float k_percentile(const Gray& img, const Params& P)
{
    Gray sm = sift::blur(img, 1.0), gx, gy;
    scharr(sm, gx, gy);
    float hmax = 0.0f;
    for (int y = 1; y < img.rows - 1; ++y)
        for (int x = 1; x < img.cols - 1; ++x) hmax = std::max(hmax, std::sqrt(gx.at(y, x) * gx.at(y, x) + gy.at(y, x) * gy.at(y, x)));
    if (!(hmax > 0.0f)) return 0.03f;
    std::vector<int> hist((size_t)P.knbins, 0);
    size_t npoints = 0;
    for (int y = 1; y < img.rows - 1; ++y)
        for (int x = 1; x < img.cols - 1; ++x) {
            const float m = std::sqrt(gx.at(y, x) * gx.at(y, x) + gy.at(y, x) * gy.at(y, x));
            if (m != 0.0f) { int b = (int)std::floor(P.knbins * (m / hmax)); if (b >= P.knbins) b = P.knbins - 1; hist[(size_t)b]++; ++npoints; }
        }
    const size_t nthreshold = (size_t)(npoints * P.kperc);
    size_t nel = 0; int k = 0;
    for (; nel < nthreshold && k < P.knbins; ++k) nel += (size_t)hist[(size_t)k];
    return nel < nthreshold ? 0.03f : hmax * (float)k / (float)P.knbins;
}

// FED step sizes of one cycle that reaches process time T (Grewenig, Weickert, Bruhn 2010), kappa-cycle reordering
inline std::vector<float> fed_taus(float T, float tau_max = 0.25f)
{
    const int n = (int)(std::ceil(std::sqrt(3.0 * T / tau_max + 0.25) - 0.5 - 1.0e-8) + 0.5);
    if (n <= 0) return {};
    const double scale = 3.0 * T / (tau_max * (double)(n * (n + 1)));
    std::vector<float> tauh((size_t)n), tau((size_t)n);
    const double c = 1.0 / (4.0 * n + 2.0), d = scale * tau_max / 2.0;
    for (int k = 0; k < n; ++k) { const double h = std::cos(3.14159265358979323846 * (2.0 * k + 1.0) * c); tauh[(size_t)k] = (float)(d / (h * h)); }
    const int kappa = n / 2;
    int prime = n + 1;
    auto is_prime = [](int v) { if (v < 2) return false; for (int q = 2; q * q <= v; ++q) if (v % q == 0) return false; return true; };
    while (!is_prime(prime)) ++prime;
    for (int k = 0, l = 0; l < n; ++k) { const int idx = ((k + 1) * kappa) % prime - 1; if (idx >= 0 && idx < n) tau[(size_t)l++] = tauh[(size_t)idx]; if (k > 4 * prime) { tau = tauh; break; } }
    return tau;
}

// one explicit diffusion step  L += tau/2 * div(c grad L)  with the arithmetic-mean conductivities of the 4-neighbourhood
inline void nld_step(Gray& L, const Gray& c, float tau)
{
    const int H = L.rows, W = L.cols;
    Gray step = L;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const float l = L.at(y, x), cc = c.at(y, x);
            float a = 0.0f;
            if (x + 1 < W) a += (cc + c.at(y, x + 1)) * (L.at(y, x + 1) - l);
            if (x > 0) a -= (c.at(y, x - 1) + cc) * (l - L.at(y, x - 1));
            if (y + 1 < H) a += (cc + c.at(y + 1, x)) * (L.at(y + 1, x) - l);
            if (y > 0) a -= (c.at(y - 1, x) + cc) * (l - L.at(y - 1, x));
            step.at(y, x) = 0.5f * tau * a;
        }
    for (size_t i = 0; i < L.v.size(); ++i) L.v[i] += step.v[i];
}

inline Gray halfsample(const Gray& s, int rows, int cols)       // cv::resize(INTER_AREA) to half: 2x2 means
{
    Gray d; d.rows = rows; d.cols = cols; d.v.resize((size_t)rows * cols);
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            const int y0 = std::min(2 * y, s.rows - 1), y1 = std::min(2 * y + 1, s.rows - 1), x0 = std::min(2 * x, s.cols - 1), x1 = std::min(2 * x + 1, s.cols - 1);
            d.at(y, x) = 0.25f * (s.at(y0, x0) + s.at(y0, x1) + s.at(y1, x0) + s.at(y1, x1));
        }
    return d;
}

// derivative kernels of step `scale` (Scharr weights 1 : 10/3 : 1 across, -1 0 +1 along), normalised: separable, 3 taps each
inline Gray deriv(const Gray& s, int scale, bool along_x)
{
    const float w = 10.0f / 3.0f, norm = 1.0f / (2.0f * scale * (w + 2.0f));
    const int H = s.rows, W = s.cols;
    Gray t = s, d = s;
    if (along_x) {
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) t.at(y, x) = s.at(y, reflect101(x + scale, W)) - s.at(y, reflect101(x - scale, W));
        for (int y = 0; y < H; ++y) {
            const int ym = reflect101(y - scale, H), yp = reflect101(y + scale, H);
            for (int x = 0; x < W; ++x) d.at(y, x) = norm * (t.at(ym, x) + w * t.at(y, x) + t.at(yp, x));
        }
    } else {
        for (int y = 0; y < H; ++y) {
            const int ym = reflect101(y - scale, H), yp = reflect101(y + scale, H);
            for (int x = 0; x < W; ++x) t.at(y, x) = s.at(yp, x) - s.at(ym, x);
        }
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) d.at(y, x) = norm * (t.at(y, reflect101(x - scale, W)) + w * t.at(y, x) + t.at(y, reflect101(x + scale, W)));
    }
    return d;
}

inline std::vector<Level> build_scale_space(const Gray& img01, const Params& P)
{
    std::vector<Level> ev;
    for (int o = 0; o < P.omax; ++o) {
        const int rows = img01.rows >> o, cols = img01.cols >> o;
        if (rows < 80 || cols < 80) break;
        for (int s = 0; s < P.nsublevels; ++s) {
            Level L;
            L.esigma = P.soffset * std::pow(2.0f, (float)s / P.nsublevels + o);
            L.sigma_size = fround(L.esigma * P.derivative_factor / (float)(1 << o));
            L.etime = 0.5f * L.esigma * L.esigma;
            L.octave = o; L.sublevel = s;
            L.Lt.rows = rows; L.Lt.cols = cols;
            ev.push_back(std::move(L));
        }
    }
    for (size_t i = 1; i < ev.size(); ++i) ev[i].tau = fed_taus(ev[i].etime - ev[i - 1].etime);
    if (ev.empty()) return ev;
    ev[0].Lt = sift::blur(img01, P.soffset);
    ev[0].Lsmooth = ev[0].Lt;
    float kcontrast = k_percentile(img01, P);
    for (size_t i = 1; i < ev.size(); ++i) {
        if (ev[i].octave > ev[i - 1].octave) { ev[i].Lt = halfsample(ev[i - 1].Lt, ev[i].Lt.rows, ev[i].Lt.cols); kcontrast *= 0.75f; }
        else ev[i].Lt = ev[i - 1].Lt;
        ev[i].Lsmooth = sift::blur(ev[i].Lt, 1.0);
        Gray gx, gy, flow = ev[i].Lt;
        scharr(ev[i].Lsmooth, gx, gy);
        const float ik2 = 1.0f / (kcontrast * kcontrast);
        for (size_t q = 0; q < flow.v.size(); ++q) flow.v[q] = 1.0f / (1.0f + ik2 * (gx.v[q] * gx.v[q] + gy.v[q] * gy.v[q]));      // Perona-Malik g2
        for (float tau : ev[i].tau) nld_step(ev[i].Lt, flow, tau);
    }
    // scale-normalised derivatives and the determinant of the Hessian
    for (auto& L : ev) {
        const float ss = (float)L.sigma_size;
        L.Lx = deriv(L.Lsmooth, L.sigma_size, true); L.Ly = deriv(L.Lsmooth, L.sigma_size, false);
        for (float& v : L.Lx.v) v *= ss;
        for (float& v : L.Ly.v) v *= ss;
        Gray Lxx = deriv(L.Lx, L.sigma_size, true), Lyy = deriv(L.Ly, L.sigma_size, false), Lxy = deriv(L.Lx, L.sigma_size, false);
        L.Ldet = L.Lx;
        for (size_t q = 0; q < L.Ldet.v.size(); ++q) { const float xx = Lxx.v[q] * ss, yy = Lyy.v[q] * ss, xy = Lxy.v[q] * ss; L.Ldet.v[q] = xx * yy - xy * xy; }
    }
    return ev;
}

struct Cand { float x, y, size, response; int octave, level; };      // x, y in image pixels

inline std::vector<Cand> find_extrema(const std::vector<Level>& ev, const Params& P)
{
    std::vector<Cand> aux;
    const float smax = 12.0f * std::sqrt(2.0f);
    for (size_t i = 0; i < ev.size(); ++i) {
        const Level& L = ev[i];
        const float ratio = (float)(1 << L.octave);
        const int border = fround(smax * L.sigma_size) + 1;
        for (int y = border; y < L.Ldet.rows - border; ++y)
            for (int x = border; x < L.Ldet.cols - border; ++x) {
                const float v = L.Ldet.at(y, x);
                if (!(v > P.dthreshold)) continue;
                if (!(v > L.Ldet.at(y, x - 1) && v > L.Ldet.at(y, x + 1) && v > L.Ldet.at(y - 1, x - 1) && v > L.Ldet.at(y - 1, x) && v > L.Ldet.at(y - 1, x + 1) &&
                      v > L.Ldet.at(y + 1, x - 1) && v > L.Ldet.at(y + 1, x) && v > L.Ldet.at(y + 1, x + 1))) continue;
                Cand c{ x * ratio, y * ratio, L.esigma * P.derivative_factor, v, L.octave, (int)i };
                // the same blob one sublevel down (or here): keep the stronger
                bool is_extremum = true, repeated = false; size_t id_rep = 0;
                for (size_t k = 0; k < aux.size(); ++k) {
                    if (aux[k].level != c.level && aux[k].level != c.level - 1) continue;
                    const float dx = c.x - aux[k].x, dy = c.y - aux[k].y;
                    if (dx * dx + dy * dy <= c.size * c.size) {
                        if (c.response > aux[k].response) { repeated = true; id_rep = k; } else is_extremum = false;
                        break;
                    }
                }
                if (!is_extremum) continue;
                if (repeated) aux[id_rep] = c; else aux.push_back(c);
            }
    }
    // ... and against the sublevel above
    std::vector<Cand> out;
    for (size_t i = 0; i < aux.size(); ++i) {
        bool keep = true;
        for (size_t j = i + 1; j < aux.size() && keep; ++j) {
            if (aux[j].level != aux[i].level + 1 && aux[j].level != aux[i].level) continue;
            const float dx = aux[i].x - aux[j].x, dy = aux[i].y - aux[j].y;
            if (dx * dx + dy * dy <= aux[j].size * aux[j].size && aux[i].response < aux[j].response) keep = false;
        }
        if (keep) out.push_back(aux[i]);
    }
    return out;
}

// 2D quadratic fit of the response around the integer extremum; false if it moves by more than a pixel
inline bool subpixel(const std::vector<Level>& ev, Cand& c)
{
    const Level& L = ev[(size_t)c.level];
    const float ratio = (float)(1 << c.octave);
    const int x = fround(c.x / ratio), y = fround(c.y / ratio);
    const Gray& D = L.Ldet;
    if (x < 1 || y < 1 || x + 1 >= D.cols || y + 1 >= D.rows) return false;
    const float Dx = 0.5f * (D.at(y, x + 1) - D.at(y, x - 1)), Dy = 0.5f * (D.at(y + 1, x) - D.at(y - 1, x));
    const float Dxx = D.at(y, x + 1) + D.at(y, x - 1) - 2.0f * D.at(y, x), Dyy = D.at(y + 1, x) + D.at(y - 1, x) - 2.0f * D.at(y, x);
    const float Dxy = 0.25f * (D.at(y + 1, x + 1) + D.at(y - 1, x - 1) - D.at(y - 1, x + 1) - D.at(y + 1, x - 1));
    const float det = Dxx * Dyy - Dxy * Dxy;
    if (det == 0.0f) return false;
    const float dx = (-Dyy * Dx + Dxy * Dy) / det, dy = (Dxy * Dx - Dxx * Dy) / det;
    if (!(std::fabs(dx) <= 1.0f && std::fabs(dy) <= 1.0f)) return false;
    c.x = (x + dx) * ratio + 0.5f * (ratio - 1.0f);
    c.y = (y + dy) * ratio + 0.5f * (ratio - 1.0f);
    c.size *= 2.0f;                                   // diameter, the convention of cv::KeyPoint::size
    return true;
}

inline float main_orientation(const Level& L, float xf, float yf, int s)
{
    float resX[109], resY[109], ang[109];
    int n = 0;
    for (int i = -6; i <= 6; ++i)
        for (int j = -6; j <= 6; ++j) {
            if (i * i + j * j >= 36) continue;
            const int iy = fround(yf + j * s), ix = fround(xf + i * s);
            if (iy < 0 || ix < 0 || iy >= L.Lx.rows || ix >= L.Lx.cols) { resX[n] = resY[n] = 0.0f; ang[n] = 0.0f; ++n; continue; }
            const float g = std::exp(-(float)(i * i + j * j) / (2.0f * 2.5f * 2.5f));
            resX[n] = g * L.Lx.at(iy, ix); resY[n] = g * L.Ly.at(iy, ix);
            float a = std::atan2(resY[n], resX[n]); if (a < 0) a += 6.28318530718f;
            ang[n] = a; ++n;
        }
    float best = 0.0f, angle = 0.0f;
    for (float a1 = 0.0f; a1 < 6.28318530718f; a1 += 0.15f) {
        const float a2 = a1 + 1.0471975512f > 6.28318530718f ? a1 - 5.2359877560f : a1 + 1.0471975512f;
        float sx = 0.0f, sy = 0.0f;
        for (int k = 0; k < n; ++k) {
            const float a = ang[k];
            if ((a1 < a2 && a1 < a && a < a2) || (a2 < a1 && ((a > 0 && a < a2) || (a > a1 && a < 6.28318530718f)))) { sx += resX[k]; sy += resY[k]; }
        }
        if (sx * sx + sy * sy > best) { best = sx * sx + sy * sy; angle = std::atan2(sy, sx); if (angle < 0) angle += 6.28318530718f; }
    }
    return angle;
}

// 486 bits: for the 2x2, 3x3 and 4x4 grids over the rotated 20s x 20s patch, every pair of cells compared on the mean
// intensity, then on the mean derivative across, then along the key point's orientation
inline void mldb(const Level& L, float xf, float yf, float angle, int scale, const Params& P, uint8_t* desc)
{
    memset(desc, 0, 61);
    const float co = std::cos(angle), si = std::sin(angle);
    const int steps[3] = { P.pattern_size, (P.pattern_size * 2 + 2) / 3, (P.pattern_size + 1) / 2 };
    int dpos = 0;
    for (int lvl = 0; lvl < 3; ++lvl) {
        const int step = steps[lvl];
        float vals[16][3];
        int nv = 0;
        for (int i = -P.pattern_size; i < P.pattern_size; i += step)
            for (int j = -P.pattern_size; j < P.pattern_size; j += step) {
                float di = 0, dx = 0, dy = 0; int ns = 0;
                for (int k = i; k < i + step; ++k)
                    for (int l = j; l < j + step; ++l) {
                        const float sy = yf + (l * co * scale + k * si * scale), sx = xf + (-l * si * scale + k * co * scale);
                        const int y1 = fround(sy), x1 = fround(sx);
                        if (x1 < 0 || y1 < 0 || x1 >= L.Lt.cols || y1 >= L.Lt.rows) continue;
                        const float rx = L.Lx.at(y1, x1), ry = L.Ly.at(y1, x1);
                        di += L.Lt.at(y1, x1); dx += -rx * si + ry * co; dy += rx * co + ry * si; ++ns;
                    }
                if (ns > 0) { di /= ns; dx /= ns; dy /= ns; }
                if (nv < 16) { vals[nv][0] = di; vals[nv][1] = dx; vals[nv][2] = dy; ++nv; }
            }
        for (int ch = 0; ch < 3; ++ch)
            for (int a = 0; a < nv; ++a)
                for (int b = a + 1; b < nv; ++b) { if (vals[a][ch] > vals[b][ch]) desc[dpos >> 3] |= (uint8_t)(1u << (dpos & 7)); ++dpos; }
    }
}

}  // namespace akaze

// cv::AKAZE::create()->detectAndCompute: key points (pt, size = diameter, angle in degrees, response, octave, class_id = level)
// and CV_8U descriptors of 61 bytes; max_features > 0 keeps the strongest
inline void akaze_detect_and_compute(const Image& img, std::vector<KeyPoint>& key_points, Mat& descriptors, int max_features = 0,
                                     const akaze::Params& P = akaze::Params())
{
    key_points.clear();
    sift::Gray g = sift::to_gray(img);
    for (float& v : g.v) v *= 1.0f / 255.0f;
    const std::vector<akaze::Level> ev = akaze::build_scale_space(g, P);
    std::vector<akaze::Cand> cand = akaze::find_extrema(ev, P), kept;
    for (auto& c : cand) if (akaze::subpixel(ev, c)) kept.push_back(c);
    std::stable_sort(kept.begin(), kept.end(), [](const akaze::Cand& a, const akaze::Cand& b) { return a.response > b.response; });
    if (max_features > 0 && (int)kept.size() > max_features) kept.resize((size_t)max_features);
    descriptors = Mat((int)kept.size(), 61, CV_8U);
    for (size_t i = 0; i < kept.size(); ++i) {
        const akaze::Cand& c = kept[i];
        const akaze::Level& L = ev[(size_t)c.level];
        const float ratio = (float)(1 << c.octave), xf = c.x / ratio, yf = c.y / ratio;
        const int s = std::max(1, akaze::fround(0.5f * c.size / ratio));
        const float angle = akaze::main_orientation(L, xf, yf, s);
        akaze::mldb(L, xf, yf, angle, s, P, descriptors.ptr<uint8_t>((int)i));
        KeyPoint kp;
        kp.pt.x = c.x; kp.pt.y = c.y; kp.size = c.size; kp.angle = angle * 57.29577951308232f; kp.response = c.response; kp.octave = c.octave; kp.class_id = c.level;
        key_points.push_back(kp);
    }
}

}  // namespace sfm
