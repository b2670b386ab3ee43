"""sfmhip_ba_create step by step (verbose = 2: a host clock stamp behind a stream sync after every step)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_opencv_amd import api, synth
ctx = api.Context(0)
for name in (sys.argv[1:] or ["C4"]):
    cfg = synth.CONFIGS[name]
    sc = synth.ba_scene(cfg["n_img"], cfg["n_pt"])
    print(name, {k: (v.dtype, v.shape) for k, v in sc.items() if hasattr(v, "dtype")}, flush=True)
    for rep in range(3):
        print(f"--- {name} rep {rep}", flush=True)
        t = time.perf_counter()
        pb = ctx.ba_create(sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], ctx.ba_options(verbose=2))
        print(f"python wall {1e3 * (time.perf_counter() - t):.2f} ms", flush=True)
        pb.close()
