"""Random-shape sweep of the device-side problem construction against the numpy restatement of tests/test_ba_setup_gpu.py:
camera counts 2..300, 1..4000 points, track lengths up to 40, shuffled observations, duplicate cameras, empty points / cameras, both
fix_first_camera settings.  usage: python experiments/sweep_setup.py [seed0 [n]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from sfm_opencv_amd import api
from test_ba_setup_gpu import host_tables
ctx = api.Context(0)
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for seed in range(seed0, seed0 + n):
    rng = np.random.default_rng(seed)
    n_cam = int(rng.integers(2, 300)); n_pt = int(rng.integers(1, 4000)); fix0 = int(rng.integers(0, 2))
    maxlen = int(rng.choice([3, 6, 12, 40]))
    oc, op = [], []
    for p in range(n_pt):
        if rng.random() < 0.03:
            continue
        L = int(rng.integers(1, min(maxlen, n_cam) + 1))
        if rng.random() < 0.5:
            c0 = int(rng.integers(0, n_cam - L + 1)); cams = np.arange(c0, c0 + L)
        else:
            cams = rng.choice(n_cam, size=L, replace=False)
        if rng.random() < 0.05:
            cams = np.concatenate([cams, cams[:1]])
        oc += list(cams); op += [p] * len(cams)
    if not oc:
        continue
    oc = np.asarray(oc, np.int32); op = np.asarray(op, np.int32)
    sh = rng.permutation(len(oc)); oc, op = oc[sh], op[sh]
    uv = rng.uniform(0, 1000, size=(len(oc), 2)); pts = rng.normal(size=(n_pt, 3))
    K0 = np.array([1000.0, 1000.0, 500.0, 400.0]); ext = np.zeros((n_cam, 6)); ext[:, 5] = 5.0
    pb = ctx.ba_create(K0, ext, pts, oc, op, uv, ctx.ba_options(fix_first_camera=fix0))
    ref = host_tables(n_cam, n_pt, oc, op, uv, fix0)
    ok = all(np.array_equal(pb.debug_table(k), v) for k, v in ref.items()) and np.array_equal(pb.params()[2], pts)
    pb.close()
    bad += not ok
    print(f"seed {seed}: {n_cam} cameras, {n_pt} points, {len(oc)} observations, max track {maxlen}, fix0 {fix0}: {'ok' if ok else 'MISMATCH'}", flush=True)
print("mismatches:", bad)
