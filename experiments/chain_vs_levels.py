"""chain solver (sfm_ba_options.solver = 0, bands of <= 3 cameras) against the level-per-launch solver (solver = 1) on scenes
with tracks of 2..4 frames: same LM trajectory, phase timings"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import oracle as orc
from sfm_opencv_amd import api, synth
ctx = api.Context(0)
for nc, npt in ((8, 400), (24, 3000), (50, 80000), (200, 300000)):
    sc = synth.ba_scene(nc, npt, max_len=4)
    res = {}
    for solver in (0, 1):
        pb = ctx.ba_create(sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], opts=ctx.ba_options(verbose=1 if nc == 200 else 0, solver=solver))
        s = pb.iterate(6)
        ctx.set_kernel_timing(True)
        pb.iterate(40)
        ph = pb.phase_ms()
        ctx.set_kernel_timing(False)
        res[solver] = (s["final_cost"], ph)
        pb.close()
    print(nc, npt, "cost chain %.12e  levels %.12e  rel %.2e | reduced solve ms: chain %.4f levels %.4f | iteration %.4f vs %.4f" % (
        res[0][0], res[1][0], abs(res[0][0] - res[1][0]) / res[1][0], res[0][1][1], res[1][1][1], res[0][1][3], res[1][1][3]))
