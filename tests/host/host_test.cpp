// host_test.cpp -- exercises sfm_opencv_amd/host/sfm_ops.hpp (the C++ mirror of the reference's function names)
// against libsfmhip.so.  Driven by tests/test_host_cpp.py through small raw binary files.
//   host_test yml  <in.bin> <out.yml>            (no GPU) save_structure
//   host_test ply  <in.bin> <out.ply>            (no GPU) get_ply_pts3d + write_ply_binary
//   host_test pipe <in.bin> <out.bin>            (GPU) match_features_for_all -> reconstruct -> fuse -> bundle_adjustment -> normals
#include "../../sfm_opencv_amd/host/sfm_ops.hpp"
#include <cstdlib>
using namespace sfm;

struct Reader {
    std::ifstream f;
    explicit Reader(const char* p) : f(p, std::ios::binary) {}
    int i32() { int v; f.read((char*)&v, 4); return v; }
    template <typename T> std::vector<T> arr(size_t n) { std::vector<T> v(n); if (n) f.read((char*)v.data(), n * sizeof(T)); return v; }
};
struct Writer {
    std::ofstream f;
    explicit Writer(const char* p) : f(p, std::ios::binary) {}
    void i32(int v) { f.write((const char*)&v, 4); }
    template <typename T> void arr(const T* p, size_t n) { if (n) f.write((const char*)p, n * sizeof(T)); }
};

static Mat mat_from(const double* d, int r, int c) { Mat m(r, c, CV_64F); std::memcpy(m.ptr<double>(), d, sizeof(double) * r * c); return m; }

int main(int argc, char** argv)
{
    if (argc < 4) { printf("usage\n"); return 2; }
    const std::string mode = argv[1];
    Reader in(argv[2]);
    if (mode == "yml" || mode == "ply") {
        const int nc = in.i32(), np = in.i32();
        auto R = in.arr<double>(9 * (size_t)nc), T = in.arr<double>(3 * (size_t)nc), P = in.arr<double>(3 * (size_t)np);
        auto C = in.arr<uint8_t>(3 * (size_t)np);
        std::vector<Mat> rot, mot; std::vector<Point3d> pts; std::vector<Vec3b> col;
        for (int c = 0; c < nc; ++c) { rot.push_back(mat_from(&R[9 * c], 3, 3)); mot.push_back(mat_from(&T[3 * c], 3, 1)); }
        for (int p = 0; p < np; ++p) { pts.emplace_back(P[3 * p], P[3 * p + 1], P[3 * p + 2]); Vec3b v; v[0] = C[3 * p]; v[1] = C[3 * p + 1]; v[2] = C[3 * p + 2]; col.push_back(v); }
        if (mode == "yml") { save_structure(argv[3], rot, mot, pts, col); return 0; }
        auto N = in.arr<double>(3 * (size_t)np);
        std::vector<Point3d> nrm; for (int p = 0; p < np; ++p) nrm.emplace_back(N[3 * p], N[3 * p + 1], N[3 * p + 2]);
        std::vector<Pt3DPly> ply;
        if (get_ply_pts3d(pts, nrm, col, ply) != 0) return 1;
        write_ply_binary(argv[3], ply);
        std::vector<Point3d> shortn(nrm.begin(), nrm.end() - 1);
        return get_ply_pts3d(pts, shortn, col, ply) == -1 ? 0 : 1;      // size mismatch must be the reference's -1
    }
    if (mode == "pipe") {
        // input: n_img, n_desc, dim, then per image: descriptors float32 (n_desc x dim), keypoints xy float32 (n_desc x 2);
        //        K (9), per image R (9), T (3) doubles
        const int n_img = in.i32(), n_desc = in.i32(), dim = in.i32();
        std::vector<Mat> descs; std::vector<std::vector<KeyPoint>> kps; std::vector<std::vector<Vec3b>> colors_all;
        for (int i = 0; i < n_img; ++i) {
            Mat d(n_desc, dim, CV_32F); in.f.read((char*)d.ptr<float>(), sizeof(float) * (size_t)n_desc * dim); descs.push_back(d);
            auto xy = in.arr<float>(2 * (size_t)n_desc);
            std::vector<KeyPoint> kp((size_t)n_desc);
            for (int k = 0; k < n_desc; ++k) { kp[k].pt.x = xy[2 * k]; kp[k].pt.y = xy[2 * k + 1]; }
            kps.push_back(kp);
            colors_all.emplace_back((size_t)n_desc);
        }
        auto Kd = in.arr<double>(9);
        Mat K = mat_from(Kd.data(), 3, 3);
        std::vector<Mat> rotations, motions;
        for (int i = 0; i < n_img; ++i) { auto r = in.arr<double>(9); auto t = in.arr<double>(3); rotations.push_back(mat_from(r.data(), 3, 3)); motions.push_back(mat_from(t.data(), 3, 1)); }

        std::vector<std::vector<DMatch>> matches_for_all;
        match_features_for_all(descs, matches_for_all);                                     // NView:1369
        std::vector<DMatch> single; match_features(descs[0], descs[1], single);             // NView:873
        if (single.size() != matches_for_all[0].size()) return 3;

        // first pair (poses are given: the RANSAC pose stages are out of scope), NView:916-987 without find_transform
        std::vector<Point2f> p1, p2; std::vector<Vec3b> colors, c2;
        get_matched_points(kps[0], kps[1], matches_for_all[0], p1, p2);
        get_matched_colors(colors_all[0], colors_all[1], matches_for_all[0], colors, c2);
        std::vector<Point3d> pts3d;
        if (reconstruct(K, rotations[0], motions[0], rotations[1], motions[1], p1, p2, pts3d) != 0) return 4;
        std::vector<std::vector<int>> inds;
        init_correspondence(kps, matches_for_all[0], std::vector<uint8_t>(), inds);
        for (int i = 1; i < (int)matches_for_all.size(); ++i) {                              // NView:1393-1455 without solvePnPRansac
            std::vector<Point2f> a, b; std::vector<Vec3b> ca, cb;
            get_matched_points(kps[i], kps[i + 1], matches_for_all[i], a, b);
            get_matched_colors(colors_all[i], colors_all[i + 1], matches_for_all[i], ca, cb);
            std::vector<Point3d> next;
            if (reconstruct(K, rotations[i], motions[i], rotations[i + 1], motions[i + 1], a, b, next) != 0) return 5;
            fuse_structure(matches_for_all[i], inds[i], inds[i + 1], pts3d, next, colors, ca);
            printf("Frame %d point cloud fused, total %d points now.\n", i, (int)pts3d.size());
        }
        std::vector<Point2f> e1, e2; std::vector<Point3d> es;
        if (reconstruct(K, rotations[0], motions[0], rotations[1], motions[1], e1, e2, es) != -1) return 6;   // empty input -> -1

        Mat intrinsic(4, 1, CV_64F);                                                         // NView:1464-1471
        intrinsic.at<double>(0) = K.at<double>(0, 0); intrinsic.at<double>(1) = K.at<double>(1, 1);
        intrinsic.at<double>(2) = K.at<double>(0, 2); intrinsic.at<double>(3) = K.at<double>(1, 2);
        std::vector<Mat> extrinsics;
        for (size_t i = 0; i < rotations.size(); ++i) {                                      // NView:1475-1487
            Mat e(6, 1, CV_64F), r; Rodrigues(rotations[i], r);
            for (int k = 0; k < 3; ++k) { e.at<double>(k) = r.at<double>(k); e.at<double>(3 + k) = motions[i].at<double>(k); }
            extrinsics.push_back(e);
        }
        std::vector<Point3d> before = pts3d;
        bundle_adjustment(intrinsic, extrinsics, inds, kps, pts3d);                          // NView:1491
        std::vector<Point3d> normals(pts3d.size());
        estimate_normals(pts3d, 10, normals);                                                // NView:1502

        Writer out(argv[3]);
        out.i32((int)matches_for_all.size());
        for (auto& m : matches_for_all) { out.i32((int)m.size()); out.arr(m.data(), m.size()); }
        out.i32((int)pts3d.size());
        out.arr(&before[0].x, 3 * before.size()); out.arr(&pts3d[0].x, 3 * pts3d.size()); out.arr(&normals[0].x, 3 * normals.size());
        out.arr(intrinsic.ptr<double>(), 4);
        for (auto& e : extrinsics) out.arr(e.ptr<double>(), 6);
        for (auto& v : inds) { out.i32((int)v.size()); out.arr(v.data(), v.size()); }
        return 0;
    }
    return 2;
}
