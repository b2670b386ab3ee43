"""CPU: the oracle's matching restatement vs an independent numpy brute force, and the host-side ratio tail of
libsfmhip.so (pure C, needs no GPU).  The reference ships no vectors for this stage => parity unpinned; these
tests pin the oracle to the published cv::batchDistance semantics (ascending distance, ties -> lower train index)."""
import numpy as np
import pytest

import oracle as orc
from sfm_opencv_amd import api, synth

TAB2 = np.array([sum(((v >> (2 * k)) & 3) != 0 for k in range(4)) for v in range(256)])


def np_knn2(d):
    order = np.argsort(d, axis=1, kind="stable")[:, :2]
    return order.astype(np.int32), np.take_along_axis(d, order, 1)


def np_l2(q, t):
    d2 = ((q[:, None, :].astype(np.float64) - t[None].astype(np.float64)) ** 2).sum(-1)
    return np.sqrt(d2.astype(np.float32))          # integer-valued inputs: d2 is exact in float32


def py_ratio(idx, dist, ratio=0.6, floor_=np.float32(10), mult=np.float32(5)):
    """NViewReconstuct.cpp:880-908 transcribed literally (double ratio compare, float gate)."""
    min_dist = np.float32(np.finfo(np.float32).max)
    for i in range(len(idx)):
        if float(dist[i, 0]) > ratio * float(dist[i, 1]):
            continue
        if dist[i, 0] < min_dist:
            min_dist = dist[i, 0]
    out = []
    with np.errstate(over="ignore"):
        gate = np.float32(mult) * max(min_dist, np.float32(floor_))
    for i in range(len(idx)):
        if float(dist[i, 0]) > ratio * float(dist[i, 1]) or dist[i, 0] > gate:
            continue
        out.append((i, idx[i, 0], 0, dist[i, 0]))
    return np.array(out, api.DMATCH) if out else np.zeros(0, api.DMATCH)


@pytest.mark.parametrize("nq,nt", [(1, 2), (17, 5), (300, 400)])
def test_knn2_l2_vs_numpy(nq, nt):
    d = synth.sift_descriptor_chain(2, max(nq, nt), seed=nq)
    q, t = d[0][:nq], d[1][:nt]
    idx, dist = orc.knn2_l2(q, t)
    ri, rd = np_knn2(np_l2(q, t))
    assert np.array_equal(idx, ri) and np.array_equal(dist, rd)


def test_knn2_l2_ties_and_short_train():
    rng = np.random.default_rng(0)
    q = rng.integers(0, 3, (50, 128)).astype(np.float32)
    t = np.repeat(rng.integers(0, 3, (10, 128)), 5, axis=0).astype(np.float32)
    idx, dist = orc.knn2_l2(q, t)
    ri, rd = np_knn2(np_l2(q, t))
    assert np.array_equal(idx, ri)
    assert (idx[:, 0] % 5 == 0).all() and (idx[:, 1] == idx[:, 0] + 1).all()     # duplicates: lowest copies win, in order
    idx, dist = orc.knn2_l2(q, t[:1])
    assert (idx[:, 0] == 0).all() and (idx[:, 1] == -1).all() and (dist[:, 1] == np.finfo(np.float32).max).all()


def test_knn2_l2_generic_floats_sse_order():
    # generic floats: the oracle's accumulation order is OpenCV's SSE2 normL2Sqr_; a float64 reference agrees to rounding
    rng = np.random.default_rng(1)
    q = rng.standard_normal((40, 128)).astype(np.float32); t = rng.standard_normal((90, 128)).astype(np.float32)
    idx, dist = orc.knn2_l2(q, t)
    d = np.sqrt(((q[:, None].astype(np.float64) - t[None]) ** 2).sum(-1))
    assert np.abs(dist - np.sort(d, 1)[:, :2]).max() < 1e-5
    assert np.array_equal(orc.l2_distance_matrix(q, t)[np.arange(40), idx[:, 0]], dist[:, 0])


@pytest.mark.parametrize("nb", [61, 32])
def test_knn2_hamming2_vs_numpy(nb):
    d = synth.akaze_descriptor_chain(2, 300, nbytes=nb, seed=3)
    q, t = d[0][:200], d[1]
    idx, dist = orc.knn2_hamming2(q, t)
    dh = TAB2[q[:, None, :] ^ t[None]].sum(-1)
    ri, rd = np_knn2(dh)
    assert np.array_equal(idx, ri) and np.array_equal(dist, rd.astype(np.float32))


def test_ratio_filter_oracle_host_lib_and_literal_transcription_agree():
    d = synth.sift_descriptor_chain(2, 800, seed=9)
    idx, dist = orc.knn2_l2(d[0], d[1])
    a = orc.ratio_filter(idx, dist); b = api.ratio_filter(idx, dist); c = py_ratio(idx, dist)
    assert len(a) > 300
    assert np.array_equal(a, b) and np.array_equal(a, c)
    assert (np.diff(a["queryIdx"]) > 0).all() and (a["imgIdx"] == 0).all()


def test_ratio_filter_edge_cases():
    # no survivor of the ratio test: min_dist stays FLT_MAX, nothing is kept (NView:880, 897-908)
    idx = np.array([[0, 1], [1, 0]], np.int32); dist = np.array([[5, 6], [7, 8]], np.float32)
    assert len(orc.ratio_filter(idx, dist)) == 0 and len(api.ratio_filter(idx, dist)) == 0
    # the absolute gate 5 * max(min_dist, 10): d0 = 51 > 50 is rejected although it passes the ratio test
    idx = np.array([[0, 1], [1, 0], [2, 3]], np.int32); dist = np.array([[1, 100], [51, 1000], [50, 1000]], np.float32)
    for f in (orc.ratio_filter, api.ratio_filter):
        m = f(idx, dist)
        assert list(m["queryIdx"]) == [0, 2]
    # rows with a missing neighbour are dropped, empty input is fine
    idx = np.array([[0, -1]], np.int32); dist = np.array([[1, np.finfo(np.float32).max]], np.float32)
    assert len(orc.ratio_filter(idx, dist)) == 0 and len(api.ratio_filter(idx, dist)) == 0
    assert len(api.ratio_filter(np.zeros((0, 2), np.int32), np.zeros((0, 2), np.float32))) == 0
    # the ratio compare is done in double: 0.6f*d1 would round differently for this pair
    d1 = np.float32(16777216.0); d0 = np.float32(0.6 * float(d1))
    idx = np.array([[0, 1]], np.int32); dist = np.array([[d0, d1]], np.float32)
    assert len(orc.ratio_filter(idx, dist)) == (0 if float(d0) > 0.6 * float(d1) else 1)
