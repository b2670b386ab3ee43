"""The C++ host mirror (sfm_opencv_amd/host/sfm_ops.hpp: the reference's function names over the C-ABI).
CPU part: save_structure / write_ply_binary reproduce the reference's files.  GPU part: the incremental pipeline
match_features_for_all -> reconstruct -> fuse_structure -> bundle_adjustment -> estimate_normals vs the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle as orc
from sfm_opencv_amd import formats, synth, api

HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "host", "host_test")
G = os.path.join(HERE, "golden")


@pytest.fixture(scope="module")
def exe():
    subprocess.check_call(["make", "-C", os.path.join(HERE, "host"), "host_test"], stdout=subprocess.DEVNULL)
    return EXE


def _dump_structure(path, y, normals=None):
    with open(path, "wb") as f:
        f.write(struct.pack("<ii", len(y["rotations"]), len(y["points"])))
        f.write(np.stack(y["rotations"]).astype("<f8").tobytes()); f.write(np.stack(y["motions"]).astype("<f8").tobytes())
        f.write(y["points"].astype("<f8").tobytes()); f.write(y["colors"].astype(np.uint8).tobytes())
        if normals is not None:
            f.write(normals.astype("<f8").tobytes())


@pytest.mark.parametrize("name", ["structure.yml", "structure_ba.yml", "structure_fountain.yml"])
def test_cpp_save_structure_byte_exact(exe, tmp_path, name):
    y = formats.read_structure_yml(os.path.join(G, name))
    _dump_structure(tmp_path / "in.bin", y)
    subprocess.check_call([exe, "yml", str(tmp_path / "in.bin"), str(tmp_path / "out.yml")])
    assert (tmp_path / "out.yml").read_bytes() == open(os.path.join(G, name), "rb").read()


def test_cpp_write_ply_binary(exe, tmp_path):
    y = formats.read_structure_yml(os.path.join(G, "structure_ba.yml"))
    nrm = orc.estimate_normals(y["points"], 10)
    nrm[7] = np.nan                                              # NaN rows are skipped (NView:238-246)
    _dump_structure(tmp_path / "in.bin", y, nrm)
    out = subprocess.run([exe, "ply", str(tmp_path / "in.bin"), str(tmp_path / "out.ply")], capture_output=True, text=True)
    assert out.returncode == 0 and "Total 3190 3D points." in out.stdout and "[Err]: items size not equal." in out.stdout
    _, v = formats.get_ply_pts3d(y["points"], nrm, y["colors"])
    assert (tmp_path / "out.ply").read_bytes() == formats.ply_bytes(v)
    assert len(formats.read_ply_binary(tmp_path / "out.ply")) == 3189


def _scene(n_img=4, n_pts=300, n_desc=360, seed=3):
    rng = np.random.default_rng(seed)
    K = np.array([[synth.K_REF[0], 0, synth.K_REF[2]], [0, synth.K_REF[1], synth.K_REF[3]], [0, 0, 1.0]])
    X = np.stack([rng.uniform(-2, 2, n_pts), rng.uniform(-1.5, 1.5, n_pts), rng.uniform(7, 11, n_pts)], 1)
    base = synth._sift_like(rng, n_pts)
    Rs, Ts, descs, kps, perms = [], [], [], [], []
    for i in range(n_img):
        R = synth.angle_axis_to_rotmat(np.array([0.01 * i, -0.06 * i, 0.005 * i])); T = np.array([-0.6 * i, 0.02 * i, 0.05 * i])
        p = X @ R.T + T
        uv = np.stack([K[0, 0] * p[:, 0] / p[:, 2] + K[0, 2], K[1, 1] * p[:, 1] / p[:, 2] + K[1, 2]], 1) + 0.2 * rng.standard_normal((n_pts, 2))
        d = np.clip(base + rng.integers(-2, 3, base.shape), 0, 255).astype(np.float32)
        extra = n_desc - n_pts
        d = np.concatenate([d, synth._sift_like(rng, extra)]); uv = np.concatenate([uv, rng.uniform(0, 3000, (extra, 2))])
        perm = rng.permutation(n_desc)
        descs.append(np.ascontiguousarray(d[perm])); kps.append(uv[perm].astype(np.float32)); perms.append(perm)
        Rs.append(R); Ts.append(T)
    return K, Rs, Ts, descs, kps, X


@pytest.mark.gpu
def test_cpp_pipeline_matches_oracle(exe, tmp_path):
    K, Rs, Ts, descs, kps, X = _scene()
    n_img, n_desc = len(descs), descs[0].shape[0]
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(struct.pack("<iii", n_img, n_desc, 128))
        for d, k in zip(descs, kps):
            f.write(d.astype("<f4").tobytes()); f.write(k.astype("<f4").tobytes())
        f.write(K.astype("<f8").tobytes())
        for R, T in zip(Rs, Ts):
            f.write(R.astype("<f8").tobytes()); f.write(T.astype("<f8").tobytes())
    out = subprocess.run([exe, "pipe", str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "[Err]: empty 2d points." in out.stdout and "Bundle Adjustment statistics (approximated RMSE):" in out.stdout
    raw = open(tmp_path / "out.bin", "rb").read(); off = 0

    def take(fmt):
        nonlocal off
        v = struct.unpack_from(fmt, raw, off); off += struct.calcsize(fmt); return v

    def arr(dt, n):
        nonlocal off
        a = np.frombuffer(raw, dt, n, off).copy(); off += a.nbytes; return a

    (npairs,) = take("<i")
    matches = []
    for _ in range(npairs):
        (m,) = take("<i"); matches.append(arr(api.DMATCH, m))
    (npt,) = take("<i")
    before = arr("<f8", 3 * npt).reshape(-1, 3); after = arr("<f8", 3 * npt).reshape(-1, 3); normals = arr("<f8", 3 * npt).reshape(-1, 3)
    Kout = arr("<f8", 4); ext = arr("<f8", 6 * n_img).reshape(-1, 6)
    inds = []
    for _ in range(n_img):
        (m,) = take("<i"); inds.append(arr("<i4", m))

    # expected, with the oracle + a literal transcription of the bookkeeping (NView:959-983, 1275-1301)
    exp_m = [orc.match_features_l2(descs[i], descs[i + 1]) for i in range(n_img - 1)]
    for a, b in zip(matches, exp_m):
        assert np.array_equal(a, b) and len(a) > 200
    Ps = [orc.projection_matrix(K, R, T) for R, T in zip(Rs, Ts)]
    e_inds = [np.full(n_desc, -1, np.int32) for _ in range(n_img)]
    m0 = exp_m[0]
    _, pts = orc.triangulate2(Ps[0], Ps[1], kps[0][m0["queryIdx"]], kps[1][m0["trainIdx"]])
    pts = list(pts)
    e_inds[0][m0["queryIdx"]] = np.arange(len(m0)); e_inds[1][m0["trainIdx"]] = np.arange(len(m0))
    for i in range(1, n_img - 1):
        m = exp_m[i]
        _, nxt = orc.triangulate2(Ps[i], Ps[i + 1], kps[i][m["queryIdx"]], kps[i + 1][m["trainIdx"]])
        for j in range(len(m)):
            q, t = m["queryIdx"][j], m["trainIdx"][j]
            if e_inds[i][q] >= 0:
                e_inds[i + 1][t] = e_inds[i][q]
            else:
                pts.append(nxt[j]); e_inds[i][q] = e_inds[i + 1][t] = len(pts) - 1
    pts = np.array(pts)
    for a, b in zip(inds, e_inds):
        assert np.array_equal(a, b)                                  # integer bookkeeping: exact
    assert before.shape == pts.shape
    assert (np.linalg.norm(before - pts, axis=1) <= 1e-5 * np.linalg.norm(pts, axis=1)).all()
    # BA input exactly as bundle_adjustment builds it (NView:1187-1212), solved by the oracle
    oc = np.concatenate([np.full((v >= 0).sum(), i, np.int32) for i, v in enumerate(e_inds)])
    op = np.concatenate([v[v >= 0] for v in e_inds]); uv = np.concatenate([kps[i][v >= 0] for i, v in enumerate(e_inds)]).astype(np.float64)
    ext0 = np.array([np.concatenate([synth.rotmat_to_angle_axis(R), T]) for R, T in zip(Rs, Ts)])
    Ko, exto, ptso, so, _ = orc.ba_solve(synth.K_REF, ext0, before, oc, op, uv)
    scale = np.abs(ptso).max()
    assert np.abs(after - ptso).max() <= 1e-5 * scale and np.abs(ext - exto).max() <= 1e-5 * scale
    assert np.abs(Kout - Ko).max() <= 1e-5 * 3000
    assert np.array_equal(ext[0], ext0[0])
    ref_n = orc.estimate_normals(after, 10)
    assert np.abs(normals - ref_n).max() <= 1e-6
