"""CPU: the from-scratch SIFT of the driver programs (sfm_opencv_amd/host/sfm_features.hpp = extract_features
NViewReconstuct.cpp:785-848 with cv::SIFT::create(0, 3, 0.04, 10), TwoViewReconstruct.cpp:112).

PARITY UNPINNED and un-pinnable (OpenCV absent, the reference ships no key-point files): accepted on behaviour --
descriptors are integer-valued floats in [0, 255] (what puts them on the exact int8 matching path), key points repeat and
match under a known similarity warp, colours are the BGR pixel under the key point."""
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle as orc
from sfm_opencv_amd import api

HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "host", "geom_test")


@pytest.fixture(scope="module")
def exe():
    subprocess.check_call(["make", "-C", os.path.join(HERE, "host"), "geom_test"], stdout=subprocess.DEVNULL)
    return EXE


def _texture(h, w, seed):
    """smooth random blobs on a gradient: plenty of DoG extrema at several scales"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = 90 + 30 * np.sin(xx / 97.0) + 20 * np.cos(yy / 71.0)
    for _ in range(700):
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        s = rng.uniform(2.0, 9.0); a = rng.uniform(-70, 70)
        img += a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
    return np.clip(img, 0, 255)


def _warp(img, A, t):
    """out(p) = img(A^-1 (p - t)), bilinear; A 2x2, t 2"""
    h, w = img.shape[:2]
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    Ai = np.linalg.inv(A)
    sx = Ai[0, 0] * (xx - t[0]) + Ai[0, 1] * (yy - t[1]); sy = Ai[1, 0] * (xx - t[0]) + Ai[1, 1] * (yy - t[1])
    x0 = np.clip(np.floor(sx).astype(int), 0, w - 2); y0 = np.clip(np.floor(sy).astype(int), 0, h - 2)
    fx = np.clip(sx - x0, 0, 1); fy = np.clip(sy - y0, 0, 1)
    out = (img[y0, x0] * (1 - fx) * (1 - fy) + img[y0, x0 + 1] * fx * (1 - fy) + img[y0 + 1, x0] * (1 - fx) * fy + img[y0 + 1, x0 + 1] * fx * fy)
    inside = (sx >= 0) & (sx <= w - 1) & (sy >= 0) & (sy <= h - 1)
    return np.where(inside, out, 100.0)


def _write_ppm(path, gray, tint=(1.0, 0.9, 0.8)):
    g = np.clip(gray, 0, 255)
    rgb = np.stack([g * tint[0], g * tint[1], g * tint[2]], 2).astype(np.uint8)
    with open(path, "wb") as f:
        f.write(b"P6\n# made by the test\n%d %d\n255\n" % (gray.shape[1], gray.shape[0]))
        f.write(rgb.tobytes())
    return rgb


def _features(exe, path, out, nmax=0):
    subprocess.check_call([exe, "features", str(path), str(out)] + ([str(nmax)] if nmax else []), stdout=subprocess.DEVNULL)
    raw = open(out, "rb").read()
    n = struct.unpack_from("<i", raw, 0)[0]
    kp = np.frombuffer(raw, api.KEYPOINT, n, 4)
    d = np.frombuffer(raw, "<f4", n * 128, 4 + 28 * n).reshape(n, 128)
    c = np.frombuffer(raw, np.uint8, 3 * n, 4 + 28 * n + 512 * n).reshape(n, 3)
    return kp, d, c


def test_sift_descriptors_and_matching_under_a_similarity(exe, tmp_path):
    img = _texture(360, 480, 5)
    th = np.radians(17.0); sc = 1.25
    A = sc * np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]); t = np.array([40.0, -55.0])
    rgb1 = _write_ppm(tmp_path / "a.ppm", img); _write_ppm(tmp_path / "b.ppm", _warp(img, A, t))
    k1, d1, c1 = _features(exe, tmp_path / "a.ppm", tmp_path / "a.bin")
    k2, d2, c2 = _features(exe, tmp_path / "b.ppm", tmp_path / "b.bin")
    assert len(k1) > 250 and len(k2) > 150
    for d in (d1, d2):
        assert d.min() >= 0 and d.max() <= 255 and np.array_equal(d, np.round(d))          # saturate_cast<uchar>: integer-valued
        nrm = np.linalg.norm(d, axis=1)
        assert np.percentile(nrm, 5) > 400 and nrm.max() < 620                               # ~512 after clipping + rounding
    assert (k1["size"] > 1).all() and (k1["angle"] >= 0).all() and (k1["angle"] < 360).all() and (k1["response"] > 0).all()
    assert (k1["x"] >= 0).all() and (k1["x"] < 480).all() and (k1["y"] >= 0).all() and (k1["y"] < 360).all()
    # colours: the BGR pixel under the key point (NView:829-838)
    xi = k1["x"].astype(int); yi = k1["y"].astype(int)
    assert np.array_equal(c1, rgb1[yi, xi][:, ::-1])
    # the reference's matcher (kNN-2 + ratio 0.6, oracle restatement) on the two descriptor sets: matches follow the warp
    m = orc.match_features_l2(d1, d2)
    assert len(m) > 50                     # the reference's absolute gate 5 * max(min_dist, 10) keeps about half of the ratio-test survivors
    p1 = np.stack([k1["x"][m["queryIdx"]], k1["y"][m["queryIdx"]]], 1); p2 = np.stack([k2["x"][m["trainIdx"]], k2["y"][m["trainIdx"]]], 1)
    err = np.linalg.norm(p1 @ A.T + t - p2, axis=1)
    assert (err < 1.5).mean() > 0.95 and np.median(err) < 0.4
    # scale and orientation follow the warp too
    ok = err < 1.5
    ratio = k2["size"][m["trainIdx"]][ok] / k1["size"][m["queryIdx"]][ok]
    assert abs(np.median(ratio) - sc) < 0.08
    # key-point angles follow cv::KeyPoint's convention (degrees, clockwise in image coordinates = the warp's rotation here)
    dang = (k2["angle"][m["trainIdx"]][ok] - k1["angle"][m["queryIdx"]][ok] + 540) % 360 - 180
    assert abs(abs(np.median(dang)) - 17.0) < 3.0


def test_feature_budget_and_poor_images(exe, tmp_path):
    img = _texture(240, 320, 9)
    _write_ppm(tmp_path / "a.ppm", img)
    k_all, _, _ = _features(exe, tmp_path / "a.ppm", tmp_path / "a.bin")
    k_top, _, _ = _features(exe, tmp_path / "a.ppm", tmp_path / "t.bin", nmax=50)
    assert len(k_top) == 50 and k_top["response"].min() >= np.sort(k_all["response"])[-60]
    # a flat image has no key points: extract_features drops it (<= 10 key points, NView:820-823)
    _write_ppm(tmp_path / "flat.ppm", np.full((120, 160), 128.0))
    k0, _, _ = _features(exe, tmp_path / "flat.ppm", tmp_path / "f.bin")
    assert len(k0) == 0


# ---------------------------------------------------------------------------------------------------------------------
# AKAZE + M-LDB (sfm_akaze.hpp): the reference's live extractor (NViewReconstuct.cpp:797), parity unpinned like the SIFT
# ---------------------------------------------------------------------------------------------------------------------
def _features_akaze(exe, path, out, nmax=0):
    subprocess.check_call([exe, "features_akaze", str(path), str(out)] + ([str(nmax)] if nmax else []), stdout=subprocess.DEVNULL)
    raw = open(out, "rb").read()
    n = struct.unpack_from("<i", raw, 0)[0]
    kp = np.frombuffer(raw, api.KEYPOINT, n, 4)
    d = np.frombuffer(raw, np.uint8, n * 61, 4 + 28 * n).reshape(n, 61)
    return kp, d


def test_akaze_rows_and_hamming2_matching_under_a_similarity(exe, tmp_path):
    img = _texture(480, 640, 6)
    th = np.radians(-23.0); sc = 1.4          # two sublevels of the scale space: AKAZE's scales are quantised (integer derivative steps, no interpolation)
    A = sc * np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]); t = np.array([-30.0, 140.0])
    _write_ppm(tmp_path / "a.ppm", img); _write_ppm(tmp_path / "b.ppm", _warp(img, A, t))
    k1, d1 = _features_akaze(exe, tmp_path / "a.ppm", tmp_path / "a.bin")
    k2, d2 = _features_akaze(exe, tmp_path / "b.ppm", tmp_path / "b.bin")
    assert len(k1) > 150 and len(k2) > 100
    for d in (d1, d2):
        assert d.shape[1] == 61 and (d[:, 60] >> 6 == 0).all()          # 486 bits: the last two of byte 60 stay clear
        ones = np.unpackbits(d, axis=1).sum(1)
        assert 150 < np.median(ones) < 340                              # comparisons of cell means: neither empty nor saturated rows
    assert (k1["size"] > 2).all() and (k1["angle"] >= 0).all() and (k1["angle"] < 360).all() and (k1["response"] > 0.001).all()
    # the reference's live matcher: kNN-2 under NORM_HAMMING2 + ratio 0.6 + the absolute gate (oracle restatement)
    m = orc.match_features_hamming2(d1, d2)
    assert len(m) > 40
    p1 = np.stack([k1["x"][m["queryIdx"]], k1["y"][m["queryIdx"]]], 1); p2 = np.stack([k2["x"][m["trainIdx"]], k2["y"][m["trainIdx"]]], 1)
    err = np.linalg.norm(p1 @ A.T + t - p2, axis=1)
    assert (err < 2.5).mean() > 0.9 and np.median(err) < 1.0
    ok = err < 2.5
    ratio = k2["size"][m["trainIdx"]][ok] / k1["size"][m["queryIdx"]][ok]
    assert abs(np.median(ratio) - sc) < 0.25
    dang = (k2["angle"][m["trainIdx"]][ok] - k1["angle"][m["queryIdx"]][ok] + 540) % 360 - 180
    assert abs(abs(np.median(dang)) - 23.0) < 5.0


def test_akaze_budget_and_flat_image(exe, tmp_path):
    img = _texture(300, 400, 12)
    _write_ppm(tmp_path / "a.ppm", img)
    k_all, _ = _features_akaze(exe, tmp_path / "a.ppm", tmp_path / "a.bin")
    k_top, d_top = _features_akaze(exe, tmp_path / "a.ppm", tmp_path / "t.bin", nmax=40)
    assert len(k_all) > 60 and len(k_top) == 40 and d_top.shape == (40, 61)
    assert k_top["response"].min() >= np.sort(k_all["response"])[-41]
    _write_ppm(tmp_path / "flat.ppm", np.full((200, 240), 128.0))
    k0, _ = _features_akaze(exe, tmp_path / "flat.ppm", tmp_path / "f.bin")
    assert len(k0) == 0
