"""C4 BA iterations only (for rocprofv3 --pmc runs): python3 experiments/run_ba.py [iterations]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_opencv_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = synth.CONFIGS["C4"]
sc = synth.ba_scene(cfg["n_img"], cfg["n_pt"])
ctx = api.Context(0, use_torch_stream=True)
pb = ctx.ba_create(sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"])
print(pb.iterate(n)["final_cost"])
