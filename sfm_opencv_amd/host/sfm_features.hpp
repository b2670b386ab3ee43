// sfm_features.hpp -- host-side feature extraction for the driver programs: what the reference's
// extract_features() does (NViewReconstuct.cpp:785-848: imread, detect + compute, drop images with <= 10 key points,
// sample the BGR colour under every key point) with a from-scratch SIFT in place of cv::AKAZE / cv::SIFT.
//
// Detector / descriptor: Lowe's SIFT with the parameters of the reference's SIFT twin, cv::SIFT::create(0, 3, 0.04, 10)
// (TwoViewReconstruct.cpp:112; sigma 1.6): image doubled first, 3 layers per octave, DoG extrema refined by the 3-D
// quadratic fit, contrast and edge tests, 36-bin orientation histograms (peaks >= 0.8 max), 4 x 4 x 8 descriptors,
// clipped at 0.2, scaled by 512 and saturated to [0, 255] -- INTEGER-VALUED floats like OpenCV's, which is what puts them
// on the exact int8 MFMA matching path of libsfmhip.so.  The live reference configuration is AKAZE (NView:797); its
// nonlinear scale space and MLDB descriptor are not rebuilt -- binary descriptors from any extractor still go through the
// Hamming2 path via a features file.
//
// PARITY UNPINNED and un-pinnable: OpenCV is absent and the reference holds no key-point files.  Accepted on behaviour:
// repeatability / matching under known warps (tests/test_features_cpu.py) and on the reconstructions the drivers obtain.
// Images come as binary PPM / PGM (the harness decodes JPEGs with PIL; no JPEG decoder is built).  CPU only: plumbing.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "sfm_ops.hpp"
#include "sfm_jpeg.hpp"

namespace sfm {

// cv::Mat CV_8UC3 (BGR, as cv::imread returns it) or CV_8UC1
struct Image {
    int rows = 0, cols = 0, channels = 0;
    std::vector<uint8_t> data;
    bool empty() const { return rows == 0 || cols == 0; }
    const uint8_t* at(int y, int x) const { return &data[((size_t)y * cols + x) * channels]; }
};

// baseline JPEG (sfm_jpeg.hpp: the pixels libjpeg / cv::imread produce) or binary PPM (P6, RGB -> stored BGR) / PGM (P5), maxval 255
inline Image imread(const std::string& path)
{
    Image img;
    std::ifstream f(path, std::ios::binary);
    if (!f) return img;
    if (f.get() == 0xFF && f.get() == 0xD8) {
        f.seekg(0, std::ios::end);
        std::vector<uint8_t> bytes((size_t)f.tellg());
        f.seekg(0);
        f.read((char*)bytes.data(), (std::streamsize)bytes.size());
        if (!f || !jpeg::decode_jpeg(bytes.data(), bytes.size(), img.rows, img.cols, img.channels, img.data)) return Image();
        return img;
    }
    f.clear(); f.seekg(0);
    std::string magic;
    f >> magic;
    if (magic != "P6" && magic != "P5") return img;
    auto next_int = [&]() {
        int v = 0; char c;
        for (;;) {
            f >> std::ws;
            if (f.peek() == '#') { while (f.get(c) && c != '\n') {} continue; }
            break;
        }
        f >> v;
        return v;
    };
    const int w = next_int(), h = next_int(), maxv = next_int();
    f.get();                                            // the single whitespace after maxval
    if (!f || w <= 0 || h <= 0 || maxv != 255) return img;
    img.rows = h; img.cols = w; img.channels = magic == "P6" ? 3 : 1;
    img.data.resize((size_t)w * h * img.channels);
    f.read((char*)img.data.data(), (std::streamsize)img.data.size());
    if (!f) return Image();
    if (img.channels == 3)
        for (size_t i = 0; i + 2 < img.data.size(); i += 3) std::swap(img.data[i], img.data[i + 2]);       // RGB -> BGR
    return img;
}

namespace sift {

struct Gray { int rows = 0, cols = 0; std::vector<float> v; float& at(int y, int x) { return v[(size_t)y * cols + x]; } float at(int y, int x) const { return v[(size_t)y * cols + x]; } };

inline Gray to_gray(const Image& img)
{
    Gray g; g.rows = img.rows; g.cols = img.cols; g.v.resize((size_t)img.rows * img.cols);
    for (int y = 0; y < img.rows; ++y)
        for (int x = 0; x < img.cols; ++x) {
            const uint8_t* p = img.at(y, x);
            g.at(y, x) = img.channels == 3 ? 0.114f * p[0] + 0.587f * p[1] + 0.299f * p[2] : (float)p[0];       // cv::COLOR_BGR2GRAY weights
        }
    return g;
}

// separable Gaussian, kernel radius ceil(4 sigma) (cv: cvRound(sigma * 8 + 1) | 1 taps for float images), replicated border
inline Gray blur(const Gray& src, double sigma)
{
    const int r = std::max(1, (int)std::ceil(4.0 * sigma));
    std::vector<float> k(2 * r + 1);
    double s = 0;
    for (int i = -r; i <= r; ++i) { k[i + r] = (float)std::exp(-0.5 * i * i / (sigma * sigma)); s += k[i + r]; }
    for (float& v : k) v = (float)(v / s);
    Gray tmp, dst;
    tmp.rows = dst.rows = src.rows; tmp.cols = dst.cols = src.cols;
    tmp.v.resize(src.v.size()); dst.v.resize(src.v.size());
    const int W = src.cols, H = src.rows;
    for (int y = 0; y < H; ++y) {
        const float* row = &src.v[(size_t)y * W];
        float* out = &tmp.v[(size_t)y * W];
        for (int x = 0; x < W; ++x) {
            float a = 0;
            if (x >= r && x + r < W) for (int i = -r; i <= r; ++i) a += k[i + r] * row[x + i];
            else for (int i = -r; i <= r; ++i) a += k[i + r] * row[std::min(std::max(x + i, 0), W - 1)];
            out[x] = a;
        }
    }
    std::vector<float> col((size_t)W);
    for (int y = 0; y < H; ++y) {
        float* out = &dst.v[(size_t)y * W];
        std::fill(out, out + W, 0.0f);
        for (int i = -r; i <= r; ++i) {
            const float* row = &tmp.v[(size_t)std::min(std::max(y + i, 0), H - 1) * W];
            const float kw = k[i + r];
            for (int x = 0; x < W; ++x) out[x] += kw * row[x];
        }
    }
    return dst;
}

inline Gray upsample2(const Gray& s)         // bilinear, like cv::resize(..., INTER_LINEAR) to twice the size
{
    Gray d; d.rows = 2 * s.rows; d.cols = 2 * s.cols; d.v.resize((size_t)d.rows * d.cols);
    for (int y = 0; y < d.rows; ++y) {
        const float fy = std::max(0.0f, (y + 0.5f) * 0.5f - 0.5f);
        const int y0 = std::min((int)fy, s.rows - 1), y1 = std::min(y0 + 1, s.rows - 1); const float wy = fy - y0;
        for (int x = 0; x < d.cols; ++x) {
            const float fx = std::max(0.0f, (x + 0.5f) * 0.5f - 0.5f);
            const int x0 = std::min((int)fx, s.cols - 1), x1 = std::min(x0 + 1, s.cols - 1); const float wx = fx - x0;
            d.at(y, x) = (1 - wy) * ((1 - wx) * s.at(y0, x0) + wx * s.at(y0, x1)) + wy * ((1 - wx) * s.at(y1, x0) + wx * s.at(y1, x1));
        }
    }
    return d;
}
inline Gray downsample2(const Gray& s)       // every second pixel (cv: INTER_NEAREST)
{
    Gray d; d.rows = s.rows / 2; d.cols = s.cols / 2; d.v.resize((size_t)d.rows * d.cols);
    for (int y = 0; y < d.rows; ++y) for (int x = 0; x < d.cols; ++x) d.at(y, x) = s.at(2 * y, 2 * x);
    return d;
}

struct Params { int nfeatures = 0, layers = 3; double contrast = 0.04, edge = 10.0, sigma = 1.6; };

struct Pyramid { int n_oct = 0, layers = 3; std::vector<Gray> gauss, dog; const Gray& G(int o, int i) const { return gauss[(size_t)o * (layers + 3) + i]; } const Gray& D(int o, int i) const { return dog[(size_t)o * (layers + 2) + i]; } };

inline Pyramid build_pyramid(const Gray& gray, const Params& P)
{
    Pyramid py; py.layers = P.layers;
    // base: doubled image (first octave -1), blurred from an assumed 0.5 (-> 1.0 after doubling) up to sigma
    Gray base = blur(upsample2(gray), std::sqrt(std::max(P.sigma * P.sigma - 4.0 * 0.25, 0.01)));
    py.n_oct = std::max(1, (int)std::lround(std::log2((double)std::min(base.cols, base.rows)) - 2.0));
    std::vector<double> sig((size_t)P.layers + 3);
    sig[0] = P.sigma;
    const double k = std::pow(2.0, 1.0 / P.layers);
    for (int i = 1; i < P.layers + 3; ++i) {
        const double prev = std::pow(k, i - 1) * P.sigma, total = prev * k;
        sig[i] = std::sqrt(total * total - prev * prev);
    }
    for (int o = 0; o < py.n_oct; ++o) {
        if (o > 0 && (py.G(o - 1, P.layers).rows < 24 || py.G(o - 1, P.layers).cols < 24)) { py.n_oct = o; break; }      // nothing survives the 5-pixel border below that
        for (int i = 0; i < P.layers + 3; ++i) {
            if (o == 0 && i == 0) py.gauss.push_back(base);
            else if (i == 0) py.gauss.push_back(downsample2(py.G(o - 1, P.layers)));
            else py.gauss.push_back(blur(py.gauss.back(), sig[i]));
        }
        for (int i = 0; i < P.layers + 2; ++i) {
            const Gray &a = py.G(o, i), &b = py.G(o, i + 1);
            Gray d; d.rows = a.rows; d.cols = a.cols; d.v.resize(a.v.size());
            for (size_t q = 0; q < a.v.size(); ++q) d.v[q] = b.v[q] - a.v[q];
            py.dog.push_back(std::move(d));
        }
    }
    return py;
}

struct Raw { float x, y, size, response, angle; int octave, layer; float scl_octv; };

// 3-D quadratic refinement of a DoG extremum (Lowe 2004, sec. 4); false: rejected
inline bool adjust_extremum(const Pyramid& py, const Params& P, int o, int& layer, int& r, int& c, Raw& out)
{
    const float img_scale = 1.0f / 255.0f, d1 = img_scale * 0.5f, d2 = img_scale, dc = img_scale * 0.25f;
    float xi = 0, xr = 0, xc = 0, contr = 0;
    int it = 0;
    for (; it < 5; ++it) {
        const Gray &pr = py.D(o, layer - 1), &cu = py.D(o, layer), &nx = py.D(o, layer + 1);
        const float dD[3] = { (cu.at(r, c + 1) - cu.at(r, c - 1)) * d1, (cu.at(r + 1, c) - cu.at(r - 1, c)) * d1, (nx.at(r, c) - pr.at(r, c)) * d1 };
        const float v2 = cu.at(r, c) * 2;
        const float dxx = (cu.at(r, c + 1) + cu.at(r, c - 1) - v2) * d2, dyy = (cu.at(r + 1, c) + cu.at(r - 1, c) - v2) * d2, dss = (nx.at(r, c) + pr.at(r, c) - v2) * d2;
        const float dxy = (cu.at(r + 1, c + 1) - cu.at(r + 1, c - 1) - cu.at(r - 1, c + 1) + cu.at(r - 1, c - 1)) * dc;
        const float dxs = (nx.at(r, c + 1) - nx.at(r, c - 1) - pr.at(r, c + 1) + pr.at(r, c - 1)) * dc;
        const float dys = (nx.at(r + 1, c) - nx.at(r - 1, c) - pr.at(r + 1, c) + pr.at(r - 1, c)) * dc;
        // solve H X = -dD (3 x 3, Cramer)
        const double H[9] = { dxx, dxy, dxs, dxy, dyy, dys, dxs, dys, dss };
        const double det = H[0] * (H[4] * H[8] - H[5] * H[7]) - H[1] * (H[3] * H[8] - H[5] * H[6]) + H[2] * (H[3] * H[7] - H[4] * H[6]);
        if (std::fabs(det) < 1e-30) return false;
        const double b[3] = { -dD[0], -dD[1], -dD[2] };
        auto det3 = [](const double* m) { return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]); };
        double M[9], X[3];
        for (int k = 0; k < 3; ++k) { for (int q = 0; q < 9; ++q) M[q] = H[q]; for (int q = 0; q < 3; ++q) M[3 * q + k] = b[q]; X[k] = det3(M) / det; }
        xc = (float)X[0]; xr = (float)X[1]; xi = (float)X[2];
        if (std::fabs(xi) < 0.5f && std::fabs(xr) < 0.5f && std::fabs(xc) < 0.5f) break;
        if (std::fabs(xi) > 1e6f || std::fabs(xr) > 1e6f || std::fabs(xc) > 1e6f) return false;
        c += (int)std::lround(xc); r += (int)std::lround(xr); layer += (int)std::lround(xi);
        if (layer < 1 || layer > P.layers || c < 5 || c >= cu.cols - 5 || r < 5 || r >= cu.rows - 5) return false;
    }
    if (it >= 5) return false;
    {
        const Gray &pr = py.D(o, layer - 1), &cu = py.D(o, layer), &nx = py.D(o, layer + 1);
        const float dD[3] = { (cu.at(r, c + 1) - cu.at(r, c - 1)) * d1, (cu.at(r + 1, c) - cu.at(r - 1, c)) * d1, (nx.at(r, c) - pr.at(r, c)) * d1 };
        contr = cu.at(r, c) * img_scale + 0.5f * (dD[0] * xc + dD[1] * xr + dD[2] * xi);
        if (std::fabs(contr) * P.layers < P.contrast) return false;
        const float v2 = cu.at(r, c) * 2;
        const float dxx = (cu.at(r, c + 1) + cu.at(r, c - 1) - v2) * d2, dyy = (cu.at(r + 1, c) + cu.at(r - 1, c) - v2) * d2;
        const float dxy = (cu.at(r + 1, c + 1) - cu.at(r + 1, c - 1) - cu.at(r - 1, c + 1) + cu.at(r - 1, c - 1)) * dc;
        const float tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
        if (det <= 0 || tr * tr * P.edge >= (P.edge + 1) * (P.edge + 1) * det) return false;
    }
    const float oscale = std::ldexp(1.0f, o);
    out.x = (c + xc) * oscale; out.y = (r + xr) * oscale;
    out.octave = o; out.layer = layer;
    out.scl_octv = (float)(P.sigma * std::pow(2.0, (layer + xi) / P.layers));
    out.size = out.scl_octv * oscale * 2;
    out.response = std::fabs(contr);
    return true;
}

// 36-bin gradient orientation histogram around (r, c) of a Gaussian image; returns its maximum
inline float orientation_hist(const Gray& img, int r0, int c0, int radius, float sigma, float* hist)
{
    const int n = 36;
    float raw[n + 4];
    for (float& v : raw) v = 0;
    float* tmp = raw + 2;
    const float expf_scale = -1.0f / (2.0f * sigma * sigma);
    for (int i = -radius; i <= radius; ++i) {
        const int y = r0 + i;
        if (y <= 0 || y >= img.rows - 1) continue;
        for (int j = -radius; j <= radius; ++j) {
            const int x = c0 + j;
            if (x <= 0 || x >= img.cols - 1) continue;
            const float dx = img.at(y, x + 1) - img.at(y, x - 1), dy = img.at(y - 1, x) - img.at(y + 1, x);
            const float w = std::exp((i * i + j * j) * expf_scale), mag = std::sqrt(dx * dx + dy * dy);
            float ori = std::atan2(dy, dx) * 57.29577951308232f;
            if (ori < 0) ori += 360.0f;
            int bin = (int)std::lround(ori * n / 360.0f);
            if (bin >= n) bin -= n;
            if (bin < 0) bin += n;
            tmp[bin] += w * mag;
        }
    }
    tmp[-1] = tmp[n - 1]; tmp[-2] = tmp[n - 2]; tmp[n] = tmp[0]; tmp[n + 1] = tmp[1];
    float mx = 0;
    for (int i = 0; i < n; ++i) {
        hist[i] = (tmp[i - 2] + tmp[i + 2]) * (1.0f / 16) + (tmp[i - 1] + tmp[i + 1]) * (4.0f / 16) + tmp[i] * (6.0f / 16);
        mx = std::max(mx, hist[i]);
    }
    return mx;
}

// 4 x 4 x 8 descriptor (Lowe sec. 6): rotated, Gaussian-weighted, trilinearly interpolated gradient histograms
inline void descriptor(const Gray& img, float px, float py_, float ori_deg, float scl, float* dst)
{
    const int d = 4, n = 8;
    const float cos_t = std::cos(ori_deg * 0.017453292519943295f), sin_t = std::sin(ori_deg * 0.017453292519943295f);
    const float bins_per_rad = n / 360.0f, exp_scale = -1.0f / (d * d * 0.5f), hist_width = 3.0f * scl;
    int radius = (int)std::lround(hist_width * 1.4142135623730951f * (d + 1) * 0.5f);
    radius = std::min(radius, (int)std::sqrt((double)img.cols * img.cols + (double)img.rows * img.rows));
    const float ct = cos_t / hist_width, st = sin_t / hist_width;
    const int cx = (int)std::lround(px), cy = (int)std::lround(py_);
    float hist[(4 + 2) * (4 + 2) * (8 + 2)];
    for (float& v : hist) v = 0;
    for (int i = -radius; i <= radius; ++i)
        for (int j = -radius; j <= radius; ++j) {
            const float c_rot = j * ct - i * st, r_rot = j * st + i * ct;
            const float rbin = r_rot + d / 2 - 0.5f, cbin = c_rot + d / 2 - 0.5f;
            const int r = cy + i, c = cx + j;
            if (!(rbin > -1 && rbin < d && cbin > -1 && cbin < d && r > 0 && r < img.rows - 1 && c > 0 && c < img.cols - 1)) continue;
            const float dx = img.at(r, c + 1) - img.at(r, c - 1), dy = img.at(r - 1, c) - img.at(r + 1, c);
            float ori = std::atan2(dy, dx) * 57.29577951308232f;
            if (ori < 0) ori += 360.0f;
            const float mag = std::sqrt(dx * dx + dy * dy) * std::exp((c_rot * c_rot + r_rot * r_rot) * exp_scale);
            float obin = (ori - ori_deg) * bins_per_rad;
            const int r0 = (int)std::floor(rbin), c0 = (int)std::floor(cbin); int o0 = (int)std::floor(obin);
            const float fr = rbin - r0, fc = cbin - c0, fo = obin - o0;
            if (o0 < 0) o0 += n;
            if (o0 >= n) o0 -= n;
            const float v_r1 = mag * fr, v_r0 = mag - v_r1;
            const float v_rc11 = v_r1 * fc, v_rc10 = v_r1 - v_rc11, v_rc01 = v_r0 * fc, v_rc00 = v_r0 - v_rc01;
            const float v111 = v_rc11 * fo, v110 = v_rc11 - v111, v101 = v_rc10 * fo, v100 = v_rc10 - v101;
            const float v011 = v_rc01 * fo, v010 = v_rc01 - v011, v001 = v_rc00 * fo, v000 = v_rc00 - v001;
            const int idx = ((r0 + 1) * (d + 2) + c0 + 1) * (n + 2) + o0;
            hist[idx] += v000; hist[idx + 1] += v001;
            hist[idx + (n + 2)] += v010; hist[idx + (n + 3)] += v011;
            hist[idx + (d + 2) * (n + 2)] += v100; hist[idx + (d + 2) * (n + 2) + 1] += v101;
            hist[idx + (d + 3) * (n + 2)] += v110; hist[idx + (d + 3) * (n + 2) + 1] += v111;
        }
    // the orientation histogram is circular
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
            const int idx = ((i + 1) * (d + 2) + (j + 1)) * (n + 2);
            hist[idx] += hist[idx + n]; hist[idx + 1] += hist[idx + n + 1];
            for (int k = 0; k < n; ++k) dst[(i * d + j) * n + k] = hist[idx + k];
        }
    const int len = d * d * n;
    float nrm2 = 0;
    for (int k = 0; k < len; ++k) nrm2 += dst[k] * dst[k];
    const float thr = std::sqrt(nrm2) * 0.2f;
    nrm2 = 0;
    for (int k = 0; k < len; ++k) { dst[k] = std::min(dst[k], thr); nrm2 += dst[k] * dst[k]; }
    const float sc = 512.0f / std::max(std::sqrt(nrm2), 1.1920929e-7f);
    for (int k = 0; k < len; ++k) dst[k] = (float)std::min(255, std::max(0, (int)std::lround(dst[k] * sc)));      // saturate_cast<uchar>: integer-valued floats
}

}  // namespace sift

// cv::SIFT::create(nfeatures = 0, nOctaveLayers = 3, contrastThreshold = 0.04, edgeThreshold = 10)->detect + compute
// (TwoViewReconstruct.cpp:112, 130-131).  descriptors: CV_32F, one 128-column row per key point.
inline void sift_detect_and_compute(const Image& img, std::vector<KeyPoint>& key_points, Mat& descriptors, const sift::Params& P = sift::Params())
{
    using namespace sift;
    key_points.clear();
    const Gray gray = to_gray(img);
    const Pyramid py = build_pyramid(gray, P);
    const float thr = (float)std::floor(0.5 * P.contrast / P.layers * 255.0);
    std::vector<Raw> raws;
    for (int o = 0; o < py.n_oct; ++o)
        for (int i = 1; i <= P.layers; ++i) {
            const Gray &pr = py.D(o, i - 1), &cu = py.D(o, i), &nx = py.D(o, i + 1);
            for (int r = 5; r < cu.rows - 5; ++r)
                for (int c = 5; c < cu.cols - 5; ++c) {
                    const float v = cu.at(r, c);
                    if (!(std::fabs(v) > thr)) continue;
                    bool ext = true;
                    if (v > 0) {
                        for (int dy = -1; dy <= 1 && ext; ++dy) for (int dx = -1; dx <= 1; ++dx) {
                            if (v < pr.at(r + dy, c + dx) || v < nx.at(r + dy, c + dx) || ((dx | dy) && v < cu.at(r + dy, c + dx))) { ext = false; break; } }
                    } else {
                        for (int dy = -1; dy <= 1 && ext; ++dy) for (int dx = -1; dx <= 1; ++dx) {
                            if (v > pr.at(r + dy, c + dx) || v > nx.at(r + dy, c + dx) || ((dx | dy) && v > cu.at(r + dy, c + dx))) { ext = false; break; } }
                    }
                    if (!ext) continue;
                    int layer = i, r1 = r, c1 = c;
                    Raw kp;
                    if (!adjust_extremum(py, P, o, layer, r1, c1, kp)) continue;
                    float hist[36];
                    const float scl = kp.scl_octv;
                    const float mx = orientation_hist(py.G(o, layer), r1, c1, (int)std::lround(3 * 1.5f * scl), 1.5f * scl, hist);
                    const float mag_thr = mx * 0.8f;
                    for (int j = 0; j < 36; ++j) {
                        const int l = j > 0 ? j - 1 : 35, r2 = j < 35 ? j + 1 : 0;
                        if (hist[j] > hist[l] && hist[j] > hist[r2] && hist[j] >= mag_thr) {
                            float bin = j + 0.5f * (hist[l] - hist[r2]) / (hist[l] - 2 * hist[j] + hist[r2]);
                            bin = bin < 0 ? 36 + bin : (bin >= 36 ? bin - 36 : bin);
                            Raw k2 = kp;
                            k2.angle = 360.0f - (360.0f / 36) * bin;
                            if (std::fabs(k2.angle - 360.0f) < 1.1920929e-7f) k2.angle = 0.0f;
                            raws.push_back(k2);
                        }
                    }
                }
        }
    // duplicates out (same position, size, angle), strongest first when a budget is given
    std::sort(raws.begin(), raws.end(), [](const Raw& a, const Raw& b) {
        if (a.x != b.x) return a.x < b.x;
        if (a.y != b.y) return a.y < b.y;
        if (a.size != b.size) return a.size > b.size;
        return a.angle < b.angle;
    });
    raws.erase(std::unique(raws.begin(), raws.end(), [](const Raw& a, const Raw& b) { return a.x == b.x && a.y == b.y && a.size == b.size && a.angle == b.angle; }), raws.end());
    if (P.nfeatures > 0 && (int)raws.size() > P.nfeatures) {
        std::stable_sort(raws.begin(), raws.end(), [](const Raw& a, const Raw& b) { return a.response > b.response; });
        raws.resize((size_t)P.nfeatures);
    }
    descriptors = Mat((int)raws.size(), 128, CV_32F);
    key_points.resize(raws.size());
    for (size_t q = 0; q < raws.size(); ++q) {
        const Raw& k = raws[q];
        // the pyramid starts at the doubled image (first octave -1): halve the coordinates for the caller
        KeyPoint kp;
        kp.pt.x = k.x * 0.5f; kp.pt.y = k.y * 0.5f; kp.size = k.size * 0.5f; kp.angle = k.angle; kp.response = k.response;
        kp.octave = (k.octave - 1) & 255; kp.octave |= k.layer << 8; kp.class_id = -1;
        key_points[q] = kp;
        const float oscale = std::ldexp(1.0f, -k.octave);
        float angle = 360.0f - k.angle;
        if (std::fabs(angle - 360.0f) < 1.1920929e-7f) angle = 0.0f;
        descriptor(py.G(k.octave, k.layer), k.x * oscale, k.y * oscale, angle, k.scl_octv, descriptors.ptr<float>((int)q));
    }
}

// extract_features (NView:785-848; SIFT as in TwoViewReconstruct.cpp:112): images that cannot be read or give <= 10 key
// points are skipped; colours are the BGR pixel under each key point (bounds test as written there, clamped to the image)
}  // namespace sfm
#include "sfm_akaze.hpp"
namespace sfm {

// which extractor extract_features runs: the reference's live one is AKAZE (NView:797), its commented twin SIFT (NView:798, TwoView:112)
enum Extractor { EXTRACT_SIFT = 0, EXTRACT_AKAZE = 1 };

inline void extract_features(std::vector<std::string>& image_names, std::vector<std::vector<KeyPoint>>& key_points_for_all,
                             std::vector<Mat>& descriptor_for_all, std::vector<std::vector<Vec3b>>& colors_for_all, int max_features = 0,
                             Extractor extractor = EXTRACT_SIFT)
{
    key_points_for_all.clear(); descriptor_for_all.clear(); colors_for_all.clear();
    sift::Params P; P.nfeatures = max_features;
    // images are independent: extracted in parallel (OpenMP, when the driver is built with it), reported and stored in file order
    const int n = (int)image_names.size();
    std::vector<std::vector<KeyPoint>> kps((size_t)n); std::vector<Mat> descs((size_t)n); std::vector<std::vector<Vec3b>> cols((size_t)n);
    std::vector<int> state((size_t)n, 0);           // 0: unreadable, 1: too few key points, 2: kept
    std::vector<std::string> warnings((size_t)n);
#pragma omp parallel for schedule(dynamic, 1)
    for (int k = 0; k < n; ++k) {
        const Image img = imread(image_names[(size_t)k]);
        if (img.empty()) continue;
        std::vector<KeyPoint>& key_points = kps[(size_t)k]; Mat& descriptor = descs[(size_t)k];
        if (extractor == EXTRACT_AKAZE) akaze_detect_and_compute(img, key_points, descriptor, max_features);
        else sift_detect_and_compute(img, key_points, descriptor, P);
        state[(size_t)k] = 1;
        if (key_points.size() <= 10) continue;
        state[(size_t)k] = 2;
        std::vector<Vec3b>& colors = cols[(size_t)k];
        colors.resize(key_points.size());
        for (size_t i = 0; i < key_points.size(); ++i) {
            const int y = (int)key_points[i].pt.y, x = (int)key_points[i].pt.x;
            if (y <= img.rows && x <= img.cols) {
                const uint8_t* p = img.at(std::min(std::max(y, 0), img.rows - 1), std::min(std::max(x, 0), img.cols - 1));
                for (int ch = 0; ch < 3; ++ch) colors[i][ch] = img.channels == 3 ? p[ch] : p[0];
            } else {
                char line[160];
                snprintf(line, sizeof line, "[Warning]: pt2d[%.3f, %.3f] out of image range.\n", key_points[i].pt.x, key_points[i].pt.y);
                warnings[(size_t)k] += line;
            }
        }
    }
    for (int k = 0; k < n; ++k) {
        if (state[(size_t)k] == 0) continue;
        printf("Extracting features for image %s...\n", image_names[(size_t)k].c_str());
        if (state[(size_t)k] == 1) continue;
        printf("%zd 2D feature point detected.\n", kps[(size_t)k].size());
        if (!warnings[(size_t)k].empty()) fputs(warnings[(size_t)k].c_str(), stdout);
        key_points_for_all.push_back(std::move(kps[(size_t)k])); descriptor_for_all.push_back(std::move(descs[(size_t)k])); colors_for_all.push_back(std::move(cols[(size_t)k]));
    }
}

}  // namespace sfm
