"""CPU: the SURVEY 8d generators on std::mt19937_64 (sfm_opencv_amd/host/sfm_synth.cpp -> libsfmsynth.so).  The engine's sequence is fixed
by the C++ standard and the transforms on top are written out in the source, so the first values are pinned here; the constructions
are checked against what the numpy generators of synth.py promise (60 % copied rows that survive the ratio test, consecutive-camera
tracks, (camera, point) order, noise and outlier levels)."""
import os
import subprocess

import numpy as np
import pytest

import oracle as orc
from sfm_opencv_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def built():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "sfm_opencv_amd", "host"), "../libsfmsynth.so"], stdout=subprocess.DEVNULL)


def test_first_values_are_pinned():
    ch = synth.sift_descriptor_chain_mt(2, 16)
    assert ch[0][0][:8].tolist() == [25.0, 38.0, 38.0, 65.0, 26.0, 1.0, 87.0, 76.0]
    assert ch[1][3][:6].tolist() == [22.0, 29.0, 37.0, 37.0, 52.0, 76.0]
    ah = synth.akaze_descriptor_chain_mt(2, 8)
    assert ah[0][0][:6].tolist() == [38, 42, 229, 78, 181, 165] and ah[1][2][:6].tolist() == [158, 147, 39, 197, 128, 241]
    sc = synth.ba_scene_mt(6, 20)
    assert sc["n_obs"] == 76 and sc["obs_pt"][:6].tolist() == [0, 2, 3, 4, 5, 9] and sc["obs_cam"][:6].tolist() == [0] * 6
    assert np.abs(sc["obs_uv"][:2] - [[2207.161634, 816.665379], [1507.85212, 924.502572]]).max() < 1e-5
    assert np.abs(sc["pts0"][0] - [1.162520899, -1.7495975, -1.825502458]).max() < 1e-8


def test_descriptor_chains_have_the_promised_structure():
    ch = synth.sift_descriptor_chain_mt(3, 1500)
    for c in ch:
        assert c.dtype == np.float32 and c.shape == (1500, 128) and np.array_equal(c, np.floor(c)) and c.min() >= 0 and c.max() <= 255
    for a, b in zip(ch, ch[1:]):
        m = orc.match_features_l2(a, b)
        assert len(m) == 900                                        # the 60 % copied rows, and only they, pass the 0.6 ratio test
        assert (np.abs(a[m["queryIdx"]] - b[m["trainIdx"]]).max(1) <= 2).all()
    ah = synth.akaze_descriptor_chain_mt(3, 1500)
    for a, b in zip(ah, ah[1:]):
        assert a.dtype == np.uint8 and a.shape == (1500, 61)
        assert len(orc.match_features_hamming2(a, b)) == 900
    # seeds: image i depends on (seed + i) and on image i - 1 only
    assert np.array_equal(synth.sift_descriptor_chain_mt(2, 300)[1], synth.sift_descriptor_chain_mt(3, 300)[1])
    assert not np.array_equal(synth.sift_descriptor_chain_mt(1, 300, seed=1)[0], synth.sift_descriptor_chain_mt(1, 300, seed=2)[0])


def test_track_scene_has_the_promised_structure():
    sc = synth.ba_scene_mt(30, 6000)
    oc, op, uv = sc["obs_cam"], sc["obs_pt"], sc["obs_uv"]
    assert np.array_equal(np.lexsort((op, oc)), np.arange(sc["n_obs"]))              # (camera, point) order, NView:1187-1197
    L = np.bincount(op, minlength=6000)
    assert L.min() == 2 and L.max() == 6 and abs(L.mean() - 4.0) < 0.1              # No = 4 Np
    for p in (0, 17, 5999):                                                         # consecutive cameras
        cams = np.sort(oc[op == p]); assert np.array_equal(cams, np.arange(cams[0], cams[0] + len(cams)))
    r = np.linalg.norm(synth.project(sc["K_true"], sc["ext_true"][oc], sc["pts_true"][op]) - uv, axis=1)
    assert 0.5 < np.median(r) < 0.7 and 0.01 < (r > 5).mean() < 0.03                # 0.5 px noise per axis, 2 % gross outliers
    assert np.array_equal(sc["ext0"][0], sc["ext_true"][0]) and np.allclose(sc["K0"], 1.01 * sc["K_true"])
    assert np.linalg.norm(sc["pts_true"], axis=1).max() <= 3.0 + 1e-9
    s = orc.ba_solve(sc["K0"], sc["ext0"], sc["pts0"], oc, op, uv)[3]
    assert s["final_cost"] < 0.05 * s["initial_cost"]
