"""In-process A/B of BA knobs (env read at problem creation): prints per-phase / per-kernel ms at C4."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sfm_opencv_amd import api, synth

def sort_points(sc):
    """relabel points so that points with the same camera set are contiguous (first camera, then track length)"""
    n_pt = sc["n_pt"]
    first = np.full(n_pt, 1 << 30); np.minimum.at(first, sc["obs_pt"], sc["obs_cam"])
    cnt = np.bincount(sc["obs_pt"], minlength=n_pt)
    order = np.lexsort((cnt, first))
    new_of_old = np.empty(n_pt, np.int64); new_of_old[order] = np.arange(n_pt)
    out = dict(sc)
    out["pts0"] = np.ascontiguousarray(sc["pts0"][order])
    op = new_of_old[sc["obs_pt"]].astype(np.int32)
    o2 = np.lexsort((op, sc["obs_cam"]))
    out["obs_cam"] = sc["obs_cam"][o2]; out["obs_pt"] = op[o2]; out["obs_uv"] = np.ascontiguousarray(sc["obs_uv"][o2])
    return out

def run(ctx, sc, env, steps=8):
    env = dict(env)
    if env.pop("SORT", 0): sc = sort_points(sc)
    for k, v in env.items(): os.environ[k] = str(v)
    pb = ctx.ba_create(sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"])
    pb.iterate(2)
    pb.reset()
    cost = pb.iterate(steps)["final_cost"]
    ph = pb.phase_ms()
    pb.close()
    for k in env: os.environ.pop(k, None)
    return ph, cost

if __name__ == "__main__":
    ctx = api.Context(0, use_torch_stream=True)
    ctx.set_kernel_timing(True)       # sfmhip_ba_phase_ms needs the event instrumentation
    cfg = synth.CONFIGS["C4"]
    sc = synth.ba_scene(cfg["n_img"], cfg["n_pt"])
    variants = [json.loads(a) for a in sys.argv[1:]] or [{}]
    for rep in range(2):
        for env in variants:
            ph, cost = run(ctx, sc, env)
            print(rep, env, "lin %.3f solve %.3f back %.3f total %.3f | camera %.3f schur %.3f fwd %.3f | cost %.6e" % (ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6], cost), flush=True)
