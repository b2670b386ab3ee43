"""Stability soak of the chain solver (sfm_ba_options.solver = 0 on a band-3 scene): the LM trajectory re-run many times must reproduce
bit for bit (catches races between the wave roles that share a front: pivot wave / update waves / the wave that brings cameras in).
python3 experiments/soak_chain.py [repeats]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_opencv_amd import api, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
tot_bad = 0
for nc, npt in ((200, 300000), (50, 80000), (30, 4000)):
    sc = synth.ba_scene(nc, npt, max_len=4)
    ctx = api.Context(0, use_torch_stream=True)
    pb = ctx.ba_create(sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"])
    ref = None; bad = 0; t0 = time.perf_counter()
    for r in range(reps):
        pb.reset()
        s = pb.iterate(12)
        K, ext, pts = pb.params()
        sig = (s["final_cost"], float(np.abs(ext).sum()), float(np.abs(pts).sum()), float(K.sum()))
        if ref is None: ref = sig
        elif sig != ref: bad += 1; print("MISMATCH at repeat", r, sig, ref, flush=True)
    print(f"{nc} cameras: {reps * 12} iterations, {bad} mismatches, {time.perf_counter() - t0:.1f} s; final cost {ref[0]}", flush=True)
    tot_bad += bad
    pb.close(); ctx.close()
sys.exit(1 if tot_bad else 0)
