"""One-off: C4 BA to convergence on the GPU vs the oracle (same termination, iterations, final cost)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle as orc
from sfm_opencv_amd import api, synth
cfg = synth.CONFIGS["C4"]
sc = synth.ba_scene(cfg["n_img"], cfg["n_pt"])
args = (sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"])
ctx = api.Context(0, use_torch_stream=True)
t0 = time.time(); K, ext, pts, s = ctx.ba_solve(*args); tg = time.time() - t0
orc.set_num_threads(16)
t0 = time.time(); Ko, exto, ptso, so, _ = orc.ba_solve(*args); tc = time.time() - t0
print("gpu:", s, "wall %.3f s" % tg)
print("cpu:", so, "wall %.3f s" % tc)
print("final cost rel diff %.3e  max|ext diff| %.3e  max|pts diff| %.3e (median %.3e)  K diff %.3e" % (
    abs(s["final_cost"] - so["final_cost"]) / so["final_cost"], np.abs(ext - exto).max(), np.abs(pts - ptso).max(),
    np.median(np.abs(pts - ptso)), np.abs(K - Ko).max()))
