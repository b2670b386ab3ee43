"""CPU: the pose-estimation host code between matching and triangulation (sfm_opencv_amd/host/sfm_geometry.hpp =
find_transform NViewReconstuct.cpp:1022-1060, solvePnPRansac + Rodrigues NView:1415-1418).

PARITY UNPINNED and un-pinnable (OpenCV's RANSAC / RNG internals are not in the reference, which holds no vectors for
them): accepted on reconstruction quality -- pose error against the synthetic truth and reprojection RMSE -- with the
reference's gates (15 inliers / 0.6 / 0.7) behaving as written."""
import os
import struct
import subprocess

import numpy as np
import pytest

from sfm_opencv_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "host", "geom_test")
K = np.array([[synth.K_REF[0], 0, synth.K_REF[2]], [0, synth.K_REF[1], synth.K_REF[3]], [0, 0, 1.0]])


@pytest.fixture(scope="module")
def exe():
    subprocess.check_call(["make", "-C", os.path.join(HERE, "host"), "geom_test"], stdout=subprocess.DEVNULL)
    return EXE


def _two_views(n, seed, outliers=0.0, noise=0.3, planar=False):
    rng = np.random.default_rng(seed)
    R = synth.angle_axis_to_rotmat(np.array([0.03, -0.2, 0.015])); T = np.array([-1.0, 0.08, 0.15])
    X = np.stack([rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), np.full(n, 9.0) if planar else rng.uniform(6, 14, n)], 1)

    def proj(R_, T_):
        p = X @ R_.T + T_
        return np.stack([K[0, 0] * p[:, 0] / p[:, 2] + K[0, 2], K[1, 1] * p[:, 1] / p[:, 2] + K[1, 2]], 1)
    p1 = proj(np.eye(3), np.zeros(3)) + noise * rng.standard_normal((n, 2))
    p2 = proj(R, T) + noise * rng.standard_normal((n, 2))
    bad = rng.random(n) < outliers
    p2[bad] = rng.uniform([0, 0], [3600, 2700], (int(bad.sum()), 2))
    return R, T, X, p1.astype(np.float32), p2.astype(np.float32), bad


def _run_essential(exe, tmp_path, p1, p2):
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(K.astype("<f8").tobytes()); f.write(struct.pack("<i", len(p1))); f.write(p1.tobytes()); f.write(p2.tobytes())
    out = subprocess.run([exe, "essential", str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    raw = open(tmp_path / "out.bin", "rb").read()
    ok = struct.unpack_from("<i", raw, 0)[0]
    R = np.frombuffer(raw, "<f8", 9, 4).reshape(3, 3); T = np.frombuffer(raw, "<f8", 3, 76)
    nm = struct.unpack_from("<i", raw, 100)[0]
    mask = np.frombuffer(raw, np.uint8, nm, 104)
    return ok, R, T, mask, out.stdout


def _rot_err_deg(Ra, Rb):
    return np.degrees(np.arccos(np.clip((np.trace(Ra.T @ Rb) - 1) / 2, -1, 1)))


@pytest.mark.parametrize("seed,outliers", [(1, 0.0), (2, 0.2), (3, 0.3)])
def test_find_transform_recovers_the_relative_pose(exe, tmp_path, seed, outliers):
    R, T, X, p1, p2, bad = _two_views(600, seed, outliers)
    ok, Re, Te, mask, log = _run_essential(exe, tmp_path, p1, p2)
    assert ok == 1 and "Init R:" in log and "Init T:" in log
    assert abs(np.linalg.det(Re) - 1) < 1e-9 and abs(np.linalg.norm(Te) - 1) < 1e-9          # unit baseline, like cv::recoverPose
    assert _rot_err_deg(Re, R) < 0.25
    assert np.degrees(np.arccos(np.clip(Te @ T / np.linalg.norm(T), -1, 1))) < 1.5
    # the mask keeps (nearly) all true correspondences and drops the gross outliers
    assert mask[~bad].mean() > 0.9 and (not bad.any() or mask[bad].mean() < 0.05)
    # quality bar: triangulating the kept matches with the recovered pose reprojects to the noise level
    s = np.linalg.norm(T)
    P1 = K @ np.hstack([np.eye(3), np.zeros((3, 1))]); P2 = K @ np.hstack([Re, (Te * s)[:, None]])
    sel = mask > 0
    err = []
    for a, b in zip(p1[sel].astype(np.float64), p2[sel].astype(np.float64)):
        A = np.stack([a[0] * P1[2] - P1[0], a[1] * P1[2] - P1[1], b[0] * P2[2] - P2[0], b[1] * P2[2] - P2[1]])
        Xh = np.linalg.svd(A)[2][-1]; Xh /= Xh[3]
        q1 = P1 @ Xh; q2 = P2 @ Xh
        err += [np.hypot(*(q1[:2] / q1[2] - a)), np.hypot(*(q2[:2] / q2[2] - b))]
    assert np.sqrt(np.mean(np.square(err))) < 1.0


def _plane_alternative(R, T, nrm, d):
    """the second motion compatible with the homography H = R + T n'/d of a plane (Faugeras-Lustman's twofold ambiguity)"""
    H = R + np.outer(T, nrm) / d
    U, w, Vt = np.linalg.svd(H)
    d1, d2, d3 = w
    s = np.linalg.det(U) * np.linalg.det(Vt)
    st = np.sqrt((d1 * d1 - d2 * d2) * (d2 * d2 - d3 * d3)) / ((d1 + d3) * d2); ct = (d2 * d2 + d1 * d3) / ((d1 + d3) * d2)
    out = []
    for sg in (1, -1):
        Rp = np.array([[ct, 0, -sg * st], [0, 1, 0], [sg * st, 0, ct]])
        out.append(s * U @ Rp @ Vt)
    return out


@pytest.mark.parametrize("seed,outliers,relief", [(21, 0.0, 0.0), (22, 0.2, 0.0), (23, 0.1, 0.25), (24, 0.0, 0.25)])
def test_find_transform_on_planar_and_near_planar_scenes(exe, tmp_path, seed, outliers, relief):
    """Round-2 verdict / advisor: the eight-point RANSAC standing in for cv::findEssentialMat's five-point solver is degenerate when
    the matches lie on one plane (any E compatible with the plane's homography fits them).  findEssentialMat now also fits that
    homography and, where it explains the matches, takes the motion from its decomposition.  relief = 0: an exactly planar scene
    has TWO valid motions (the same two a five-point solver returns) -- the result must be one of them; with a little relief
    (+-2.8 % of the depth) the off-plane matches pick the true one."""
    rng = np.random.default_rng(seed)
    n = 500
    R = synth.angle_axis_to_rotmat(np.array([0.03, -0.2, 0.015])); T = np.array([-1.0, 0.08, 0.15])
    X = np.stack([rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), 9.0 + relief * rng.uniform(-1, 1, n)], 1)

    def proj(R_, T_):
        p = X @ R_.T + T_
        return np.stack([K[0, 0] * p[:, 0] / p[:, 2] + K[0, 2], K[1, 1] * p[:, 1] / p[:, 2] + K[1, 2]], 1)
    p1 = (proj(np.eye(3), np.zeros(3)) + 0.3 * rng.standard_normal((n, 2))).astype(np.float32)
    p2 = proj(R, T) + 0.3 * rng.standard_normal((n, 2))
    bad = rng.random(n) < outliers
    p2[bad] = rng.uniform([0, 0], [3600, 2700], (int(bad.sum()), 2))
    ok, Re, Te, mask, log = _run_essential(exe, tmp_path, p1, p2.astype(np.float32))
    assert ok == 1
    assert mask[~bad].mean() > 0.9 and (not bad.any() or mask[bad].mean() < 0.05)
    if relief == 0.0:
        errs = [_rot_err_deg(Re, Ra) for Ra in _plane_alternative(R, T, np.array([0, 0, 1.0]), 9.0)]
        assert min(errs) < 0.5, errs
    else:
        assert _rot_err_deg(Re, R) < 0.6, _rot_err_deg(Re, R)
        assert np.degrees(np.arccos(np.clip(Te @ T / np.linalg.norm(T), -1, 1))) < 4.0


def test_find_transform_gates(exe, tmp_path):
    # fewer than eight matches: no model; mostly outliers: the 0.6 inlier-ratio gate of NView:1042 refuses
    R, T, X, p1, p2, bad = _two_views(7, 5)
    assert _run_essential(exe, tmp_path, p1, p2)[0] == 0
    R, T, X, p1, p2, bad = _two_views(400, 6, outliers=0.7)
    assert _run_essential(exe, tmp_path, p1, p2)[0] == 0
    # 15 or fewer inliers (NView:1042: feasible_count <= 15)
    R, T, X, p1, p2, bad = _two_views(14, 7, noise=0.05)
    assert _run_essential(exe, tmp_path, p1, p2)[0] == 0


@pytest.mark.parametrize("seed,outliers", [(11, 0.0), (12, 0.3)])
def test_solve_pnp_ransac_recovers_the_camera(exe, tmp_path, seed, outliers):
    rng = np.random.default_rng(seed)
    n = 400
    R = synth.angle_axis_to_rotmat(np.array([0.1, 0.4, -0.05])); T = np.array([0.5, -0.2, 1.0])
    X = np.stack([rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), rng.uniform(6, 14, n)], 1).astype(np.float32)
    p = X.astype(np.float64) @ R.T + T
    uv = np.stack([K[0, 0] * p[:, 0] / p[:, 2] + K[0, 2], K[1, 1] * p[:, 1] / p[:, 2] + K[1, 2]], 1) + 0.4 * rng.standard_normal((n, 2))
    bad = rng.random(n) < outliers
    uv[bad] = rng.uniform([0, 0], [3600, 2700], (int(bad.sum()), 2))
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(K.astype("<f8").tobytes()); f.write(struct.pack("<i", n)); f.write(X.tobytes()); f.write(uv.astype(np.float32).tobytes())
    subprocess.check_call([exe, "pnp", str(tmp_path / "in.bin"), str(tmp_path / "out.bin")])
    raw = open(tmp_path / "out.bin", "rb").read()
    ok = struct.unpack_from("<i", raw, 0)[0]
    rvec = np.frombuffer(raw, "<f8", 3, 4); Te = np.frombuffer(raw, "<f8", 3, 28); Re = np.frombuffer(raw, "<f8", 9, 52).reshape(3, 3)
    n_in = struct.unpack_from("<i", raw, 124)[0]
    assert ok == 1
    assert _rot_err_deg(Re, R) < 0.05 and np.abs(Te - T).max() < 0.02
    assert np.abs(synth.angle_axis_to_rotmat(rvec) - Re).max() < 1e-12                           # rvec <-> R consistent
    assert abs(n_in - int((~bad).sum())) <= 0.02 * n
    q = X[~bad].astype(np.float64) @ Re.T + Te
    rep = np.stack([K[0, 0] * q[:, 0] / q[:, 2] + K[0, 2], K[1, 1] * q[:, 1] / q[:, 2] + K[1, 2]], 1) - uv[~bad]
    assert np.sqrt((rep ** 2).sum(1).mean()) < 0.8                                                # 0.4 px noise per axis


def _run_pnp(exe, tmp_path, X, uv):
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(K.astype("<f8").tobytes()); f.write(struct.pack("<i", len(X))); f.write(X.astype(np.float32).tobytes()); f.write(uv.astype(np.float32).tobytes())
    subprocess.check_call([exe, "pnp", str(tmp_path / "in.bin"), str(tmp_path / "out.bin")])
    raw = open(tmp_path / "out.bin", "rb").read()
    ok = struct.unpack_from("<i", raw, 0)[0]
    return ok, np.frombuffer(raw, "<f8", 3, 28), np.frombuffer(raw, "<f8", 9, 52).reshape(3, 3), struct.unpack_from("<i", raw, 124)[0]


@pytest.mark.parametrize("n", [4, 5, 6, 9])
def test_solve_pnp_ransac_with_the_minimum_number_of_correspondences(exe, tmp_path, n):
    """The reference registers a frame as soon as it has FOUR 2D-3D pairs (the only gate is `< 4`, NViewReconstuct.cpp:1410-1414;
    cv::solvePnPRansac then runs its P3P kernel [3P]).  Rounds 1-2 needed six (DLT minimal sets) and skipped such frames."""
    rng = np.random.default_rng(40 + n)
    R = synth.angle_axis_to_rotmat(np.array([0.1, 0.4, -0.05])); T = np.array([0.5, -0.2, 1.0])
    X = np.stack([rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), rng.uniform(6, 14, n)], 1).astype(np.float32)
    p = X.astype(np.float64) @ R.T + T
    uv = np.stack([K[0, 0] * p[:, 0] / p[:, 2] + K[0, 2], K[1, 1] * p[:, 1] / p[:, 2] + K[1, 2]], 1) + 0.2 * rng.standard_normal((n, 2))
    ok, Te, Re, n_in = _run_pnp(exe, tmp_path, X, uv)
    assert ok == 1 and n_in == n
    q = X.astype(np.float64) @ Re.T + Te
    rep = np.stack([K[0, 0] * q[:, 0] / q[:, 2] + K[0, 2], K[1, 1] * q[:, 1] / q[:, 2] + K[1, 2]], 1) - uv
    assert np.sqrt((rep ** 2).sum(1).mean()) < 1.0
    # four noisy points leave the pose loosely determined; from six on it is tight
    assert _rot_err_deg(Re, R) < (2.0 if n < 6 else 0.5) and np.abs(Te - T).max() < (0.5 if n < 6 else 0.1)
    # three points are refused (the reference's own gate would have skipped the frame)
    assert _run_pnp(exe, tmp_path, X[:3], uv[:3])[0] == 0


def test_rodrigues_round_trip(exe, tmp_path):
    rng = np.random.default_rng(0)
    v = np.concatenate([rng.standard_normal((50, 3)), 1e-9 * rng.standard_normal((5, 3)), np.zeros((1, 3)),
                        (np.pi - 1e-8) * np.eye(3), [[2.0, -1.0, 0.5]]])
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(struct.pack("<i", len(v))); f.write(v.astype("<f8").tobytes())
    subprocess.check_call([exe, "rodrigues", str(tmp_path / "in.bin"), str(tmp_path / "out.bin")])
    raw = np.fromfile(tmp_path / "out.bin", "<f8").reshape(len(v), 12)
    for i in range(len(v)):
        R = raw[i, :9].reshape(3, 3)
        assert np.abs(R - synth.angle_axis_to_rotmat(v[i])).max() < 1e-12
        assert np.abs(synth.angle_axis_to_rotmat(raw[i, 9:]) - R).max() < 1e-6                   # back through the log map
