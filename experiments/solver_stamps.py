"""Per-panel cycle stamps of the level-per-launch reduced solve at C4 (-DSFMHIP_EXPERIMENTS build, SFMHIP_SOLVER_STAMPS=1): the library
prints them to stderr at the eighth solve.  SFMHIP_LIB=experiments/_exp/libsfmhip_exp.so SFMHIP_SOLVER_STAMPS=1 python experiments/solver_stamps.py [C3|C4]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_opencv_amd import api, synth
cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C4"]
sc = synth.ba_scene_mt(cfg["n_img"], cfg["n_pt"])
ctx = api.Context(0)
pb = ctx.ba_create(sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], opts=ctx.ba_options(verbose=1))
s = pb.iterate(12)
print("cost", s["final_cost"])
