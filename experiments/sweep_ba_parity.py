import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import oracle as orc
from sfm_opencv_amd import synth, api
ctx = api.Context(0, use_torch_stream=True)
bad = 0
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for seed in range(seed0, seed0 + (int(sys.argv[2]) if len(sys.argv) > 2 else 40)):
    rng = np.random.default_rng(seed)
    n_cam = int(rng.choice([3, 4, 6, 11, 17, 31, 45, 66, 90, 121, 150, 200]))
    n_pt = int(rng.integers(40, 100)) * n_cam
    max_len = int(rng.integers(2, 11))
    sc = synth.ba_scene(n_cam, n_pt, seed=int(rng.integers(1, 1 << 30)), max_len=max_len, outlier_frac=float(rng.choice([0.0, 0.03])))
    kw = dict(fix_intrinsics=int(rng.integers(0, 2)), fix_first_camera=int(rng.integers(0, 2)), jacobi_scaling=int(rng.integers(0, 2)),
              huber_delta=float(rng.choice([0.0, 1.0, 4.0])))
    lin = int(rng.integers(0, 3))
    args = (sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"])
    pb = ctx.ba_create(*args, opts=ctx.ba_options(linearizer=lin, **kw))
    S, rhs, cost = pb.reduced_system(1e3)
    So, rhso, costo = orc.ba_reduced_system(*args, 1e3, opts=orc.ba_default_options(**kw))
    e1 = np.abs(S - So).max() / np.abs(So).max(); e2 = np.abs(rhs - rhso).max() / max(np.abs(rhso).max(), 1e-300)
    s = pb.iterate(4)
    so = orc.ba_solve(*args, opts=orc.ba_default_options(**kw), force_iterations=4)[3]
    e3 = abs(s["final_cost"] - so["final_cost"]) / so["final_cost"]
    ok = e1 <= 1e-9 and e2 <= 1e-9 and e3 <= 1e-8 and s["successful_steps"] == so["successful_steps"]
    bad += not ok
    print(seed, n_cam, n_pt, max_len, "lin", lin, kw, "%.1e %.1e %.1e" % (e1, e2, e3), s["successful_steps"], so["successful_steps"], "OK" if ok else "BAD", flush=True)
    pb.close()
print("bad", bad)
