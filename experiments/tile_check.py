"""Run-tile linearisation against the per-observation kernels and the oracle: reduced system, then timing of forced iterations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle as orc
from sfm_opencv_amd import api, synth

def args(sc): return sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"]
def rel(a, b): return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
ctx = api.Context(0)
for shape, kw in [((12, 700), {}), ((12, 700), dict(fix_intrinsics=1)), ((12, 700), dict(fix_first_camera=0, huber_delta=0.0)), ((60, 20000), {})]:
    sc = synth.ba_scene(*shape)
    res = {}
    for lin in (1, 2):
        pb = ctx.ba_create(*args(sc), opts=ctx.ba_options(linearizer=lin, **kw))
        res[lin] = pb.reduced_system(1e4); pb.close()
    So, ro, co = orc.ba_reduced_system(*args(sc), 1e4, opts=orc.ba_default_options(**kw))
    for lin in (1, 2):
        S, r, c = res[lin]
        print(shape, kw, "lin", lin, "S", rel(S, So), "rhs", rel(r, ro), "cost", abs(c - co) / co, "sym", np.abs(S - S.T).max() / np.abs(S).max(), flush=True)
if len(sys.argv) > 1:
    cfg = synth.CONFIGS[sys.argv[1]]
    sc = synth.ba_scene(cfg["n_img"], cfg["n_pt"])
    for lin in (1, 2):
        pb = ctx.ba_create(*args(sc), opts=ctx.ba_options(linearizer=lin))
        pb.iterate(5)
        ctx.synchronize(); t = time.perf_counter(); s = pb.iterate(100); ctx.synchronize(); dt = time.perf_counter() - t
        ctx.set_kernel_timing(True); pb.iterate(20); ph = pb.phase_ms(); ctx.set_kernel_timing(False)
        print(sys.argv[1], "lin", lin, "ms/it %.4f" % (dt * 10), "cost", s["final_cost"], "phases", [round(float(x), 4) for x in ph], flush=True)
        pb.close()
