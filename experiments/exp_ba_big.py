"""BA at C5 size on one GPU (1000 cameras, 2M points, 8M observations): per-phase / per-kernel times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_opencv_amd import api, synth
n_cam = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n_pt = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
t0 = time.time(); sc = synth.ba_scene(n_cam, n_pt); print("scene %.1f s, obs %d" % (time.time() - t0, sc["n_obs"]), flush=True)
ctx = api.Context(0, use_torch_stream=True)
ctx.set_kernel_timing(True)       # sfmhip_ba_phase_ms needs the event instrumentation
t0 = time.time(); pb = ctx.ba_create(sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"]); print("create %.1f s" % (time.time() - t0), flush=True)
s = pb.iterate(2); pb.reset()
t0 = time.time(); s = pb.iterate(8); dt = time.time() - t0
ph = pb.phase_ms()
print("it/s %.1f | lin %.3f solve %.3f back %.3f total %.3f | camera %.3f schur %.3f fwd %.3f | nnz blocks %d | cost %.4e -> %.4e, succ %d" % (
    8 / dt, ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6], ph[7], s["initial_cost"], s["final_cost"], s["successful_steps"]))
