// geom_test.cpp -- CPU-only harness for sfm_opencv_amd/host/sfm_geometry.hpp (no GPU call is made: only the pose-estimation
// host code runs).  Driven by tests/test_geometry_cpu.py through raw binary files.
//   geom_test essential <in.bin> <out.bin>   in: K(9) n, p1 (n x 2 float), p2 (n x 2 float)
//                                            out: ok(int) R(9) T(3) n_mask mask(n bytes)
//   geom_test pnp <in.bin> <out.bin>         in: K(9) n, obj (n x 3 float), img (n x 2 float)
//                                            out: ok(int) rvec(3) T(3) R(9) n_inliers
//   geom_test rodrigues <in.bin> <out.bin>   in: n, rvec (n x 3 double)   out: R (n x 9), back (n x 3)
//   geom_test features <image.ppm> <out.bin> [max]  out: n, key points (n x 28 B), descriptors (n x 128 float), colours (n x 3)
//   geom_test features_akaze <image> <out.bin> [max]  same with AKAZE: descriptors n x 61 bytes
#include "../../sfm_opencv_amd/host/sfm_features.hpp"
#include "../../sfm_opencv_amd/host/sfm_geometry.hpp"
using namespace sfm;

int main(int argc, char** argv)
{
    if (argc < 4) return 2;
    const std::string mode = argv[1];
    if (mode == "features" || mode == "features_akaze") {
        std::vector<std::string> names = { argv[2] };
        std::vector<std::vector<KeyPoint>> kps; std::vector<Mat> descs; std::vector<std::vector<Vec3b>> cols;
        extract_features(names, kps, descs, cols, argc > 4 ? std::atoi(argv[4]) : 0, mode == "features" ? EXTRACT_SIFT : EXTRACT_AKAZE);
        std::ofstream o(argv[3], std::ios::binary);
        const int n = kps.empty() ? 0 : (int)kps[0].size();
        o.write((const char*)&n, 4);
        if (n) { o.write((const char*)kps[0].data(), (std::streamsize)sizeof(KeyPoint) * n); o.write((const char*)descs[0].buf.data(), (std::streamsize)descs[0].buf.size()); o.write((const char*)cols[0].data(), 3 * (std::streamsize)n); }
        return 0;
    }
    std::ifstream in(argv[2], std::ios::binary);
    std::ofstream out(argv[3], std::ios::binary);
    auto rd = [&](void* p, size_t n) { in.read((char*)p, (std::streamsize)n); };
    auto wr = [&](const void* p, size_t n) { out.write((const char*)p, (std::streamsize)n); };
    if (mode == "rodrigues") {
        int n = 0; rd(&n, 4);
        std::vector<double> v(3 * (size_t)n); rd(v.data(), v.size() * 8);
        for (int i = 0; i < n; ++i) {
            Mat r(3, 1, CV_64F), R, back;
            for (int k = 0; k < 3; ++k) r.at<double>(k) = v[3 * i + k];
            Rodrigues_vec(r, R); Rodrigues(R, back);
            wr(R.ptr<double>(), 72); wr(back.ptr<double>(), 24);
        }
        return 0;
    }
    Mat K(3, 3, CV_64F); rd(K.ptr<double>(), 72);
    int n = 0; rd(&n, 4);
    if (mode == "essential") {
        std::vector<Point2f> p1((size_t)n), p2((size_t)n);
        rd(p1.data(), 8 * (size_t)n); rd(p2.data(), 8 * (size_t)n);
        Mat R, T, mask;
        const int ok = find_transform(K, p1, p2, R, T, mask) ? 1 : 0;
        wr(&ok, 4);
        if (R.empty()) { R = Mat(3, 3, CV_64F); T = Mat(3, 1, CV_64F); }
        wr(R.ptr<double>(), 72); wr(T.ptr<double>(), 24);
        const int nm = mask.rows * mask.cols; wr(&nm, 4); wr(mask.ptr<uint8_t>(), (size_t)nm);
        return 0;
    }
    if (mode == "pnp") {
        std::vector<Point3f> obj((size_t)n); std::vector<Point2f> img((size_t)n);
        rd(obj.data(), 12 * (size_t)n); rd(img.data(), 8 * (size_t)n);
        Mat r, T, R;
        std::vector<int> inl;
        const int ok = solvePnPRansac(obj, img, K, r, T, &inl) ? 1 : 0;
        wr(&ok, 4);
        if (!ok) { r = Mat(3, 1, CV_64F); T = Mat(3, 1, CV_64F); }
        Rodrigues_vec(r, R);
        wr(r.ptr<double>(), 24); wr(T.ptr<double>(), 24); wr(R.ptr<double>(), 72);
        const int ni = (int)inl.size(); wr(&ni, 4);
        return 0;
    }
    return 2;
}
