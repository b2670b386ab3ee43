"""GPU: the drop-in driver programs sfm_opencv_amd/host/{NViewReconstruct,TwoViewReconstruct} (the reference's two
main()s, NViewReconstuct.cpp:1334-1524 / TwoViewReconstruct.cpp:50-97) on synthetic scenes.

With --poses-from-file the pose stages are bypassed and every number is checked against the oracle pipeline (oracle
matching -> oracle triangulation -> a literal transcription of the bookkeeping -> oracle BA -> oracle normals) fed to
formats.py; the files the driver wrote must be byte-identical to formats.py applied to the driver's own numbers (the
C++ writers against the golden-file-pinned Python ones), and those numbers within the stated tolerances of the oracle's.
Without it the drivers run stand-alone (find_transform + solvePnPRansac, parity unpinned): accepted on the final RMSE."""
import os
import re
import subprocess

import numpy as np
import pytest

import oracle as orc
from sfm_opencv_amd import features_io, formats, synth

pytestmark = pytest.mark.gpu
HOST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sfm_opencv_amd", "host")


@pytest.fixture(scope="module")
def drivers():
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    return os.path.join(HOST, "NViewReconstruct"), os.path.join(HOST, "TwoViewReconstruct")


def _scene(n_img=5, n_pts=500, n_desc=560, seed=3):
    rng = np.random.default_rng(seed)
    K = np.array([[synth.K_REF[0], 0, synth.K_REF[2]], [0, synth.K_REF[1], synth.K_REF[3]], [0, 0, 1.0]])
    X = np.stack([rng.uniform(-2, 2, n_pts), rng.uniform(-1.5, 1.5, n_pts), rng.uniform(7, 11, n_pts)], 1)
    base = synth._sift_like(rng, n_pts)
    Rs, Ts, descs, kps, cols = [], [], [], [], []
    for i in range(n_img):
        R = synth.angle_axis_to_rotmat(np.array([0.01 * i, -0.06 * i, 0.005 * i])); T = np.array([-0.6 * i, 0.02 * i, 0.05 * i])
        p = X @ R.T + T
        uv = np.stack([K[0, 0] * p[:, 0] / p[:, 2] + K[0, 2], K[1, 1] * p[:, 1] / p[:, 2] + K[1, 2]], 1) + 0.2 * rng.standard_normal((n_pts, 2))
        d = np.clip(base + rng.integers(-2, 3, base.shape), 0, 255).astype(np.float32)
        extra = n_desc - n_pts
        d = np.concatenate([d, synth._sift_like(rng, extra)]); uv = np.concatenate([uv, rng.uniform(0, 3000, (extra, 2))])
        perm = rng.permutation(n_desc)
        descs.append(np.ascontiguousarray(d[perm])); kps.append(uv[perm].astype(np.float32))
        cols.append(rng.integers(0, 256, (n_desc, 3), dtype=np.uint8))
        Rs.append(R); Ts.append(T)
    return K, Rs, Ts, descs, kps, cols, X


def _oracle_pipeline(K, Rs, Ts, descs, kps, cols):
    n_img, n_desc = len(descs), descs[0].shape[0]
    match = orc.match_features_hamming2 if descs[0].dtype == np.uint8 else orc.match_features_l2
    ms = [match(descs[i], descs[i + 1]) for i in range(n_img - 1)]
    Ps = [orc.projection_matrix(K, R, T) for R, T in zip(Rs, Ts)]
    inds = [np.full(n_desc, -1, np.int32) for _ in range(n_img)]
    m0 = ms[0]
    _, pts = orc.triangulate2(Ps[0], Ps[1], kps[0][m0["queryIdx"]], kps[1][m0["trainIdx"]])
    pts = list(pts); colors = list(cols[0][m0["queryIdx"]])
    inds[0][m0["queryIdx"]] = np.arange(len(m0)); inds[1][m0["trainIdx"]] = np.arange(len(m0))
    for i in range(1, n_img - 1):
        m = ms[i]
        _, nxt = orc.triangulate2(Ps[i], Ps[i + 1], kps[i][m["queryIdx"]], kps[i + 1][m["trainIdx"]])
        for j in range(len(m)):
            q, t = m["queryIdx"][j], m["trainIdx"][j]
            if inds[i][q] >= 0:
                inds[i + 1][t] = inds[i][q]
            else:
                pts.append(nxt[j]); colors.append(cols[i][q]); inds[i][q] = inds[i + 1][t] = len(pts) - 1
    pts = np.array(pts); colors = np.array(colors, np.uint8)
    oc = np.concatenate([np.full((v >= 0).sum(), i, np.int32) for i, v in enumerate(inds)])
    op = np.concatenate([v[v >= 0] for v in inds]); uv = np.concatenate([kps[i][v >= 0] for i, v in enumerate(inds)]).astype(np.float64)
    ext0 = np.array([np.concatenate([synth.rotmat_to_angle_axis(R), T]) for R, T in zip(Rs, Ts)])
    Ko, exto, ptso, so, _ = orc.ba_solve(np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]]), ext0, pts, oc, op, uv)
    return pts, colors, ptso, so


def test_nview_driver_with_given_poses_against_the_oracle_pipeline(drivers, tmp_path):
    K, Rs, Ts, descs, kps, cols, X = _scene()
    feat = tmp_path / "features.bin"
    features_io.write_features(feat, K, kps, descs, cols, poses=list(zip(Rs, Ts)))
    out = subprocess.run([drivers[0], str(feat), str(tmp_path), "--poses-from-file"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    log = out.stdout
    for line in ("Total 5 image files.", "Matching images 0 - 1", "Construct from the first two frames...", "Incremental SFM...",
                 "Frame 3 reconstructed.", "Bundle adjustment fo SFM...", "Bundle Adjustment statistics (approximated RMSE):",
                 "Point3d 0 offset: [", "structure_ba.yml saved.", "Saving structure to ply...", "Save structure done."):
        assert line in log, line
    y0 = formats.read_structure_yml(tmp_path / "structure.yml"); y1 = formats.read_structure_yml(tmp_path / "structure_ba.yml")
    # (1) the C++ writers == the golden-file-pinned Python writers on the same numbers, byte for byte
    for name, y in (("structure.yml", y0), ("structure_ba.yml", y1)):
        assert (tmp_path / name).read_bytes() == formats.structure_yml_text(y["rotations"], y["motions"], y["points"], y["colors"]).encode()
    # (2) against the oracle pipeline
    pts, colors, ptso, so = _oracle_pipeline(K, Rs, Ts, descs, kps, cols)
    assert y0["points"].shape == pts.shape and np.array_equal(y0["colors"], colors) and np.array_equal(y1["colors"], colors)
    assert np.array_equal(y0["points"], y0["points"].astype(np.float32).astype(np.float64))          # Point3f -> Point3d (NView:1155)
    assert (np.linalg.norm(y0["points"] - pts, axis=1) <= 1e-5 * np.linalg.norm(pts, axis=1)).all()
    assert np.abs(y1["points"] - ptso).max() <= 1e-5 * np.abs(ptso).max()
    for a, R, T in zip(range(5), Rs, Ts):
        assert np.array_equal(y0["rotations"][a], R) and np.array_equal(y0["motions"][a].reshape(3), T)
        assert np.array_equal(y1["rotations"][a], R) and np.array_equal(y1["motions"][a].reshape(3), T)   # quirk 1: pre-BA poses
    m = re.search(r"Final   RMSE\(pixel\): ([0-9.eE+-]+)", log)
    assert m and abs(float(m.group(1)) - np.sqrt(so["final_cost"] / so["num_residuals"])) <= 1e-4
    # (3) the .ply: points + GPU normals + colours (BGR -> RGB), byte-identical to formats.py on the oracle normals where they agree
    ply = formats.read_ply_binary(tmp_path / "structure_ba.ply")
    nrm = orc.estimate_normals(y1["points"], 10)
    ret, v = formats.get_ply_pts3d(y1["points"], nrm, y1["colors"])
    assert len(ply) == len(v)
    for k in ("x", "y", "z", "r", "g", "b"):
        assert np.array_equal(ply[k], v[k])
    got = np.stack([ply["nx"], ply["ny"], ply["nz"]], 1); ref = np.stack([v["nx"], v["ny"], v["nz"]], 1)
    assert np.abs(got - ref).max() <= 1e-6
    assert (tmp_path / "structure_ba.ply").read_bytes()[:300] == formats.ply_bytes(v)[:300]          # header + first vertices


def test_nview_driver_binary_descriptors_hamming2(drivers, tmp_path):
    """The reference's LIVE configuration: AKAZE's 61-byte MLDB rows matched with NORM_HAMMING2 (NView:797, 876).  The extractor
    itself is not rebuilt, so the rows are synthetic (a random 486-bit code per scene point, 3 % of the bits flipped per view):
    the driver must route CV_8U descriptors to the Hamming2 kernels and reproduce the oracle pipeline."""
    K, Rs, Ts, descs, kps, cols, X = _scene()
    rng = np.random.default_rng(11)
    n_pts, n_desc = X.shape[0], descs[0].shape[0]
    code = rng.integers(0, 256, (n_pts, 61), dtype=np.uint8)
    # _scene() permuted every image's rows; recover which row shows which scene point from the noise-free part of the key points
    bdescs = []
    for i in range(len(descs)):
        p = X @ Rs[i].T + Ts[i]
        uv = np.stack([K[0, 0] * p[:, 0] / p[:, 2] + K[0, 2], K[1, 1] * p[:, 1] / p[:, 2] + K[1, 2]], 1)
        d2 = ((kps[i][:, None, :].astype(np.float64) - uv[None, :, :]) ** 2).sum(-1)
        owner = d2.argmin(1); is_pt = d2.min(1) < 4.0
        b = rng.integers(0, 256, (n_desc, 61), dtype=np.uint8)
        flips = np.packbits(rng.random((n_desc, 61, 8)) < 0.03, axis=2).reshape(n_desc, 61)
        b[is_pt] = code[owner[is_pt]] ^ flips[is_pt]
        bdescs.append(np.ascontiguousarray(b))
    feat = tmp_path / "features.bin"
    features_io.write_features(feat, K, kps, bdescs, cols, poses=list(zip(Rs, Ts)))
    out = subprocess.run([drivers[0], str(feat), str(tmp_path), "--poses-from-file"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    y0 = formats.read_structure_yml(tmp_path / "structure.yml"); y1 = formats.read_structure_yml(tmp_path / "structure_ba.yml")
    pts, colors, ptso, so = _oracle_pipeline(K, Rs, Ts, bdescs, kps, cols)
    assert len(pts) > 300                                   # most scene points were matched through the binary rows
    assert y0["points"].shape == pts.shape and np.array_equal(y0["colors"], colors)
    assert (np.linalg.norm(y0["points"] - pts, axis=1) <= 1e-5 * np.linalg.norm(pts, axis=1)).all()
    assert np.abs(y1["points"] - ptso).max() <= 1e-5 * np.abs(ptso).max()


def test_nview_driver_stand_alone_quality(drivers, tmp_path):
    """No poses given: find_transform + solvePnPRansac (parity unpinned) -- the reconstruction must reach the noise floor."""
    K, Rs, Ts, descs, kps, cols, X = _scene(n_img=6, seed=9)
    feat = tmp_path / "features.bin"
    features_io.write_features(feat, K, kps, descs, cols)
    out = subprocess.run([drivers[0], str(feat), str(tmp_path), "--quiet", "--write-back-poses"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    log = out.stdout
    assert "Init R:" in log and "Point3d 0 offset" not in log
    m = re.search(r"Final   RMSE\(pixel\): ([0-9.eE+-]+)", log)
    assert m and float(m.group(1)) < 0.6                         # 0.2 px noise per axis + the few wrong matches under Huber
    y = formats.read_structure_yml(tmp_path / "structure_ba.yml")
    assert len(y["rotations"]) == 6 and y["points"].shape[0] > 450
    # poses up to the gauge (first camera at the origin, unit first baseline): rotations are absolute
    for i in range(6):
        dR = y["rotations"][i].T @ Rs[i]
        assert np.degrees(np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1))) < 0.2
    base = np.linalg.norm(y["motions"][1])
    for i in range(1, 6):
        assert np.abs(y["motions"][i].reshape(3) / base - Ts[i] / np.linalg.norm(Ts[1])).max() < 0.03


def test_twoview_driver(drivers, tmp_path):
    K, Rs, Ts, descs, kps, cols, X = _scene(n_img=2, seed=4)
    feat = tmp_path / "features.bin"
    features_io.write_features(feat, K, kps, descs, cols, poses=list(zip(Rs, Ts)))
    out = subprocess.run([drivers[1], str(feat), str(tmp_path), "--poses-from-file"], capture_output=True, text=True)
    assert out.returncode == 0 and "successful!!!" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    txt = (tmp_path / "structure.yml").read_text()
    m = orc.match_features_l2(descs[0], descs[1])
    P1 = orc.projection_matrix(K, np.eye(3), np.zeros(3)); P2 = orc.projection_matrix(K, Rs[1], Ts[1])
    xyzw, xyz = orc.triangulate2(P1, P2, kps[0][m["queryIdx"]], kps[1][m["trainIdx"]])
    y = formats.read_structure_yml(tmp_path / "structure.yml")
    assert y["points"].shape == xyz.shape and np.array_equal(y["colors"], cols[0][m["queryIdx"]])
    assert (np.linalg.norm(y["points"] - xyz, axis=1) <= 1e-5 * np.linalg.norm(xyz, axis=1)).all()
    # float tokens: "%.8e" (Point3f), unlike the N-view file's "%.16e"
    first = txt[txt.index("Points:"):].splitlines()[1]
    assert re.fullmatch(r"   - \[ -?\d\.\d{8}e[+-]\d\d, -?\d\.\d{8}e[+-]\d\d, -?\d\.\d{8}e[+-]\d\d \]", first), first
    # writer against formats.py on values that print to the same tokens: re-emit from the parsed float32 values
    h = np.concatenate([y["points"].astype(np.float32).T, np.ones((1, len(xyz)), np.float32)])
    assert txt == formats.structure_yml_text_twoview(y["rotations"], y["motions"], h, y["colors"])
    # stand-alone (essential matrix): same point count order of magnitude, unit baseline
    out = subprocess.run([drivers[1], str(feat), str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0 and "Init R:" in out.stdout
    y2 = formats.read_structure_yml(tmp_path / "structure.yml")
    assert abs(np.linalg.norm(y2["motions"][1]) - 1.0) < 1e-9 and y2["points"].shape[0] > 0.9 * len(xyz)


GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_nview_driver_on_the_crazyhorse_sequence(drivers, tmp_path):
    """BASELINE.json configs[1]: NViewReconstuct on dataset/crazyhorse, full sequence, BA to convergence.  Input: the
    features fixture made from the reference's seven JPEGs by this repo's own SIFT (tests/golden/make_crazyhorse_features.py;
    the reference's AKAZE is not rebuilt, so this is a quality test: parity unpinned).  Everything after the features runs
    here: matching, essential matrix, PnP, triangulation, fusion, BA, normals, the three output files."""
    out = subprocess.run([drivers[0], os.path.join(GOLD, "crazyhorse_features.bin"), str(tmp_path), "--quiet"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    log = out.stdout
    assert "Total 7 image files." in log and "Frame 5 point cloud fused" in log and "Save structure done." in log
    v = re.search(r"#views: (\d+)\n #residuals: (\d+)\n Initial RMSE\(pixel\): ([0-9.eE+-]+)\n Final   RMSE\(pixel\): ([0-9.eE+-]+)", log)
    assert v, log[-2000:]
    views, nres, r0, r1 = int(v.group(1)), int(v.group(2)), float(v.group(3)), float(v.group(4))
    assert views == 7 and nres > 2000
    assert r1 < 0.5 and r1 < r0                                   # sub-pixel after BA (0.18 px when the fixture was made)
    y0 = formats.read_structure_yml(tmp_path / "structure.yml"); y1 = formats.read_structure_yml(tmp_path / "structure_ba.yml")
    assert len(y0["rotations"]) == 7 and y0["points"].shape[0] > 350 and y1["points"].shape == y0["points"].shape
    assert np.array_equal(y0["points"], y0["points"].astype(np.float32).astype(np.float64))        # float32-exact before BA (NView:1155)
    ply = formats.read_ply_binary(tmp_path / "structure_ba.ply")
    assert len(ply) == y1["points"].shape[0]
    # the cameras sweep around the object: monotone rotation about the vertical axis, baseline growing along x
    yaw = [np.degrees(np.arctan2(R[0, 2], R[2, 2])) for R in y0["rotations"]]
    assert all(b > a for a, b in zip(yaw, yaw[1:])) and 5 < yaw[-1] < 40
    tx = [float(np.asarray(T).reshape(3)[0]) for T in y0["motions"]]
    assert all(b < a for a, b in zip(tx, tx[1:]))
    # the bulk of the structure lies in front of the first camera at a sensible depth (unit = first baseline)
    z = y1["points"][:, 2]
    assert np.median(z) > 2 and (z > 0).mean() > 0.95


def test_twoview_driver_on_crazyhorse(drivers, tmp_path):
    """BASELINE.json configs[0]: TwoViewReconstruct on dataset/crazyhorse (first two images of the features fixture)."""
    out = subprocess.run([drivers[1], os.path.join(GOLD, "crazyhorse_features.bin"), str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0 and "successful!!!" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    y = formats.read_structure_yml(tmp_path / "structure.yml")
    assert len(y["rotations"]) == 2 and np.array_equal(y["rotations"][0], np.eye(3)) and not y["motions"][0].any()
    assert abs(np.linalg.norm(y["motions"][1]) - 1.0) < 1e-9 and abs(np.linalg.det(y["rotations"][1]) - 1.0) < 1e-9
    assert y["points"].shape[0] > 120 and (y["points"][:, 2] > 0).mean() > 0.95
    # reprojection of the triangulated points into both views with the recovered pose: sub-pixel
    from sfm_opencv_amd import features_io
    f = features_io.read_features(os.path.join(GOLD, "crazyhorse_features.bin"))
    K = f["K"]
    X = y["points"]
    for R, T in ((np.eye(3), np.zeros(3)), (y["rotations"][1], y["motions"][1].reshape(3))):
        p = X @ R.T + T
        uv = np.stack([K[0, 0] * p[:, 0] / p[:, 2] + K[0, 2], K[1, 1] * p[:, 1] / p[:, 2] + K[1, 2]], 1)
        assert ((uv[:, 0] > -50) & (uv[:, 0] < 1074) & (uv[:, 1] > -50) & (uv[:, 1] < 818)).mean() > 0.98


def test_nview_driver_on_crazyhorse_with_akaze_rows(drivers, tmp_path):
    """BASELINE.json configs[1] in the reference's LIVE configuration: AKAZE key points + 61-byte M-LDB rows (this repo's
    sfm_akaze.hpp on the reference's seven JPEGs, tests/golden/make_crazyhorse_features.py) matched under NORM_HAMMING2 on the GPU,
    then the same stages as above.  Quality test, parity unpinned."""
    out = subprocess.run([drivers[0], os.path.join(GOLD, "crazyhorse_features_akaze.bin"), str(tmp_path), "--quiet"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    log = out.stdout
    assert "Total 7 image files." in log and "Save structure done." in log
    v = re.search(r"#views: (\d+)\n #residuals: (\d+)\n Initial RMSE\(pixel\): ([0-9.eE+-]+)\n Final   RMSE\(pixel\): ([0-9.eE+-]+)", log)
    assert v, log[-2000:]
    views, nres, r0, r1 = int(v.group(1)), int(v.group(2)), float(v.group(3)), float(v.group(4))
    assert views == 7 and nres > 2000 and r1 < 0.8 and r1 < r0
    y0 = formats.read_structure_yml(tmp_path / "structure.yml"); y1 = formats.read_structure_yml(tmp_path / "structure_ba.yml")
    assert len(y0["rotations"]) == 7 and y0["points"].shape[0] > 300
    yaw = [np.degrees(np.arctan2(R[0, 2], R[2, 2])) for R in y0["rotations"]]
    assert all(b > a for a, b in zip(yaw, yaw[1:])) and 5 < yaw[-1] < 40          # the same sweep the SIFT run recovers
    z = y1["points"][:, 2]
    assert np.median(z) > 2 and (z > 0).mean() > 0.9


def test_nview_driver_spreads_bundle_adjustment_over_contexts(drivers, tmp_path):
    """`NViewReconstruct ... --gpus=0,0`: bundle_adjustment() of the C++ driver over two contexts (the one-card rehearsal of a
    multi-GPU process: points sharded by first camera, one thread per context, the sums exchanged inside the library) must write
    the structure the single-context run writes."""
    feat = os.path.join(GOLD, "crazyhorse_features.bin")
    a = tmp_path / "one"; b = tmp_path / "two"; a.mkdir(); b.mkdir()
    # SIFT-rows fixture; extractor flags do not matter for a features file
    o1 = subprocess.run([drivers[0], feat, str(a), "--quiet"], capture_output=True, text=True)
    o2 = subprocess.run([drivers[0], feat, str(b), "--quiet", "--gpus=0,0"], capture_output=True, text=True)
    assert o1.returncode == 0 and o2.returncode == 0, o2.stdout[-2000:] + o2.stderr[-2000:]
    r = re.compile(r"Final   RMSE\(pixel\): ([0-9.eE+-]+)")
    f1, f2 = float(r.search(o1.stdout).group(1)), float(r.search(o2.stdout).group(1))
    assert abs(f1 - f2) <= 1e-4 * f1, (f1, f2)
    y1 = formats.read_structure_yml(a / "structure_ba.yml"); y2 = formats.read_structure_yml(b / "structure_ba.yml")
    assert y1["points"].shape == y2["points"].shape and np.abs(y1["points"] - y2["points"]).max() <= 1e-3 * max(1.0, np.abs(y1["points"]).max())
    # match_features_for_all spreads the chain's pairs over the same two contexts (sfmhip_match_pairs_multi): the lists, hence everything
    # written before bundle adjustment, are the single-context ones byte for byte
    assert (a / "structure.yml").read_bytes() == (b / "structure.yml").read_bytes()
