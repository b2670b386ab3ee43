"""Minimal driver for rocprofv3 passes: N forced LM iterations of the C4 scene (env knobs of an experiments build apply)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_opencv_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = synth.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "C4"]
ctx = api.Context(0)
sc = synth.ba_scene(cfg["n_img"], cfg["n_pt"])
pb = ctx.ba_create(sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"])
s = pb.iterate(n)
print(s)
