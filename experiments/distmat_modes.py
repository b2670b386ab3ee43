"""Where does the distance-matrix kernel's time go?  One process, the -DSFMHIP_EXPERIMENTS build (SFMHIP_LIB), every variant
sustained (50 warm-up launches, then 3 blocks of 100): the real kernel, without the exact-sqrt fix-up, with plain sqrtf,
compute only (no stores), stores only (no matrix products, no sqrt), and torch's fill of the same 400 MB as the write ceiling
of THIS device (boxes differ by ~10 %).  usage: SFMHIP_LIB=experiments/_exp/libsfmhip_exp.so python experiments/distmat_modes.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfm_opencv_amd import api, synth
ctx = api.Context(0, use_torch_stream=True)
nq = nt = 10000
dd = synth.sift_descriptor_chain(2, nq, seed=synth.SEED + 100000)
q = torch.from_numpy(dd[0]).cuda(); t = torch.from_numpy(dd[1]).cuda()
qs, ts = ctx.descset_l2(q), ctx.descset_l2(t)
stream = torch.cuda.current_stream()
alg = 4.0 * nq * nt + 4.0 * 128 * (nq + nt)

def sustained(fn):
    for _ in range(50): fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for b in range(3):
        ev[b].record(stream)
        for _ in range(100): fn()
    ev[3].record(stream); torch.cuda.synchronize()
    return [ev[b].elapsed_time(ev[b + 1]) * 10 for b in range(3)]

for ld in (10000, 10016):
    buf = torch.empty((nq, ld), dtype=torch.float32, device="cuda"); out = buf[:, :nt]
    flat = torch.empty(nq * nt, dtype=torch.float32, device="cuda")
    for rep in range(2):
        us = sustained(lambda: flat.fill_(1.0))
        print(f"ld {ld}  torch fill of 400 MB (write ceiling of this device)   {us[0]:6.1f} {us[1]:6.1f} {us[2]:6.1f} us  -> {4e8 / us[2] / 1e6:.2f} TB/s", flush=True)
        for nwv in (("4", "8") if os.environ.get("DM_AB") else ("",)):
          if nwv:
            os.environ["SFMHIP_EXP_DM_NW"] = nwv; print("  -- waves per workgroup:", nwv)
          for mode, name in ((0, "distmat_i8_kernel (real)"), (1, "no sqrt (d^2 as float)"), (4, "plain sqrtf, no exactness fix-up"), (2, "compute only, no stores"),
                           (16, "stores only, no MFMA / sqrt"), (16 | 32, "stores only, no operand loads either"), (32, "real kernel without operand loads"), (8, "real kernel, nontemporal stores")):
            os.environ["SFMHIP_EXP_DISTMAT"] = str(mode)
            us = sustained(lambda: ctx.l2_distance_matrix_dev(qs, ts, out))
            print(f"ld {ld}  mode {mode:2d} {name:36s} {us[0]:6.1f} {us[1]:6.1f} {us[2]:6.1f} us  -> {alg / us[2] / 1e6:.2f} TB/s", flush=True)
            if nwv and mode == 4:
                break
    del out, buf, flat
