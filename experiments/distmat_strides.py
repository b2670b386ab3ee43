"""Distance-matrix kernel time against the output row stride (floats), same box, interleaved repetitions."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfm_opencv_amd import api, synth
ctx = api.Context(0, use_torch_stream=True)
nq = nt = 10000
dd = synth.sift_descriptor_chain(2, nq, seed=synth.SEED + 100000)
q = torch.from_numpy(dd[0]).cuda(); t = torch.from_numpy(dd[1]).cuda()
qs, ts = ctx.descset_l2(q), ctx.descset_l2(t)
alg = 4.0 * nq * nt + 4.0 * 128 * (nq + nt)
strides = [10000, 10016, 10048, 10112, 10240, 10496, 12288]
res = {s: [] for s in strides}
stream = torch.cuda.current_stream()
for rep in range(4):
    for ld in strides:
        buf = torch.empty((nq, ld), dtype=torch.float32, device="cuda"); out = buf[:, :nt]
        for _ in range(3): ctx.l2_distance_matrix_dev(qs, ts, out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(stream)
        for _ in range(20): ctx.l2_distance_matrix_dev(qs, ts, out)
        e1.record(stream); torch.cuda.synchronize()
        res[ld].append(e0.elapsed_time(e1) / 20 * 1e3)
        del out, buf
for ld in strides:
    v = res[ld]; print("stride %5d: %s us  -> best %.1f us = %.2f TB/s" % (ld, " ".join("%.1f" % x for x in v), min(v), alg / min(v) / 1e6))
