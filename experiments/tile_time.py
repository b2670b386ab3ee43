"""Kernel time of the run-tile linearisation (HIP events around the kernel), C4 scene."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_opencv_amd import api, synth
cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C4"]
ctx = api.Context(0)
sc = synth.ba_scene(cfg["n_img"], cfg["n_pt"])
pb = ctx.ba_create(sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], opts=ctx.ba_options(linearizer=2))
pb.iterate(3)
ctx.set_kernel_timing(True); pb.iterate(10); ph = pb.phase_ms()
print("tile kernel ms %.4f  build phase %.4f" % (ph[4], ph[0]), flush=True)
