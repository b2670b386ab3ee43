"""Stability soak: the C4 LM trajectory re-run many times must reproduce bit for bit (catches races in the solver's
atomics / barriers and in the publish + poll handshake).  python3 experiments/soak_ba.py [repeats]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_opencv_amd import api, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cfg = synth.CONFIGS["C4"]
sc = synth.ba_scene(cfg["n_img"], cfg["n_pt"])
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
    if r % 50 == 49: print(r + 1, "repeats,", bad, "mismatches, %.1f s" % (time.perf_counter() - t0), flush=True)
print("done:", reps * 12, "iterations,", bad, "mismatches; final cost", ref[0])
