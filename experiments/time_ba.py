import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_opencv_amd import api, synth
cfg = synth.CONFIGS["C4"]
sc = synth.ba_scene(cfg["n_img"], cfg["n_pt"])
ctx = api.Context(0, use_torch_stream=True)
pb = ctx.ba_create(sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"])
pb.iterate(2)
for rep in range(3):
    pb.reset(); pb.iterate(2); ctx.synchronize()
    t = time.perf_counter(); pb.iterate(10); ctx.synchronize(); dt = time.perf_counter() - t
    print("ms per step %.4f" % (dt * 100))
