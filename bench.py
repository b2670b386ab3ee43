#!/usr/bin/env python3
"""bench.py -- headline benchmark of the matching -> BA hot path on MI355X (BASELINE.json metric:
"BA iterations/sec + matched-pairs/sec, 200-img/300k-pt scene").

    python bench.py --gpus N --steps K --warmup W

N>1: either launched by torch.distributed.run (one rank per GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the env), or
run bare -- then this process only spawns the N ranks (before anything touches the GPU), waits, relays rank 0's JSON
line and exits non-zero if a rank failed.

Workload (config C4 of BASELINE.json, synthetic): 200 images x 5000 integer-valued SIFT-like descriptors
(199 chain pairs) and a 200-camera / 300k-point / ~1.2M-observation BA scene.  The scene is FIXED as N grows
("strong" scaling): pairs and points are sharded over the ranks; BA exchanges one all-reduce per LM iteration.

Timed regions (each: barrier + synchronize on both sides, exactly K steps, MAX over ranks):
  A (primary, `value`)  K LM iterations (linearise + Schur build + reduced solve + back-substitution + candidate cost),
                         parameters/observations resident in HBM.
  B (`matched_pairs_per_sec`) K passes over this rank's chain pairs: descriptor preparation + kNN-2 + ratio tail on
                         the device, match lists copied to pinned host memory; float descriptors resident in HBM.
  C (`ba_solve_end_to_end`, rank 0 at N=1) the drop-in call itself: sfmhip_ba_solve from HOST arrays = problem construction +
                         LM to Ceres-default convergence + parameters written back (what bundle_adjustment() is, NView:1162-1244),
                         and `match_pairs_from_host`: the chain matched from host descriptor matrices (uploads inside the call).
  D (`matched_pairs_per_sec_hamming2`) region B on 61-byte binary rows through the Hamming2 kernel: the reference's LIVE
                         configuration (AKAZE + BFMatcher(NORM_HAMMING2), NView:797, 876).
Plus the materialised 10k x 10k x 128 distance matrix (north-star HBM-roofline case) timed with events on the stream.
Rank 0 prints ONE JSON line.  The CPU baseline is this repo's own C restatement (oracle/, kind "port"), NOT OpenCV/Ceres.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec); ~6.3 TB/s measured copy
I8_MFMA_PEAK_TOPS = 5000.0     # dense int8 MFMA = 2x bf16 dense (~2.5 PFLOP/s)
FP4_MFMA_PEAK_TFLOPS = 10000.0  # dense FP4 MFMA (block-scaled v_mfma_scale_f32_32x32x64_f8f6f4), ~10 PFLOP/s
VALU_PEAK_TLANEOPS = 256 * 4 * 32 * 2.4e9 / 1e12      # 78.6: 256 CUs x 4 SIMD-32 x 2.4 GHz, one 32-bit lane-op per lane and cycle (MI355X_MICROARCH.md: v_fma_f32 wave64 = 2 cycles)
# TCC FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE per launch from separate rocprofv3 --pmc passes over this command at
# N=1 / C4: recorded figures, NOT measured in the run that prints them (counters are not collectable in-process)
# what a kernel that only stores can reach is measured in the run itself: a 400 MB fill (one 16-byte store per thread, 4 KB per
# workgroup -- the fastest write shape found, 6.6-6.9 TB/s sustained; devices differ by ~10 %, experiments/wbw4.hip, profiles/README.md)
SOLVER_KERNEL = "chol_node_forward_kernel"     # the leaf level of the dissection (the longest solver launch)
RECORDED_TRAFFIC = {"knn2_i8_kernel<4>": (3.045e8, "profiles/r03_traffic_pmc.md"),
                    "distmat_i8_kernel<4>": (4.881e8, "profiles/r03_traffic_pmc.md"),
                    "knn2_hamming2_kernel": (6.368e8, "profiles/r03_traffic_pmc.md"),     # the VALU kernel (62..64-byte rows), not on the bench path any more
                    "knn2_hamming2_fp4_kernel": (8.934e8, "profiles/r03_hamming_fp4.md"),  # 2 x 428,280 KiB FETCH_SIZE (LDS-DMA row streams) + 15,920 KiB WRITE_SIZE
                    "ba_camschur_kernel": (9.94e7, "profiles/r03_traffic_pmc.md")}      # 2 x FETCH_SIZE (streaming reads; raw for gathers / scalar loads) + WRITE_SIZE


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children of a parent that never touches the GPU
    (no torch import, no HIP call), wait for them and relay rank 0's output."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(n):
        # HSA_ENABLE_IPC_MODE_LEGACY=0: the pool's host driver only supports dmabuf IPC; without it RCCL's cross-process buffer
        # exchange fails with "hipIpcGetMemHandle: invalid argument" (the box exports it already; kept for bare environments)
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # poll every child: a rank that dies during init would otherwise leave rank 0 waiting in a collective until torch's timeout
    import threading
    buf = []
    rd = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    rd.start()
    rcs = [None] * n
    while any(rc is None for rc in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
        if any(rc not in (None, 0) for rc in rcs):
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    p.terminate()
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    try:
                        rcs[r] = p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill(); rcs[r] = p.wait()
            break
        time.sleep(0.05)
    rd.join(timeout=5)
    sys.stdout.write((buf[0] if buf else b"").decode()); sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        print(f"[bench] ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C4", choices=["C3", "C4", "C5", "small"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gemm", action="store_true")
    ap.add_argument("--generator", default="mt19937_64", choices=["mt19937_64", "numpy"],
                    help="synthetic inputs: SURVEY 8d's std::mt19937_64 generators (libsfmsynth.so) or the numpy PCG64 ones the tests use")
    ap.add_argument("--no-match", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip region C (the drop-in calls from host arrays)")
    ap.add_argument("--no-hamming", action="store_true", help="skip region D (Hamming2 chain)")
    ap.add_argument("--staged-match-copy", action="store_true", help="match lists into device buffers, then one D2H copy per pass (default: written to pinned host memory by the kernel)")
    ap.add_argument("--match-steps", type=int, default=0, help="matching passes timed (default: min(steps, 20))")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (default: the box's cores, at most 16)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist
    from sfm_opencv_amd import synth, api
    from sfm_opencv_amd import dist as sdist

    rank, world, local = sdist.init_process_group()
    if world != args.gpus and rank == 0:
        print(f"[bench] warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    torch.cuda.set_device(local)
    ctx = api.Context(local, use_torch_stream=True)
    stream = ctx.torch_stream

    mt = args.generator == "mt19937_64"
    gen_scene = synth.ba_scene_mt if mt else synth.ba_scene
    gen_sift = synth.sift_descriptor_chain_mt if mt else synth.sift_descriptor_chain
    gen_akaze = synth.akaze_descriptor_chain_mt if mt else synth.akaze_descriptor_chain
    cfg = dict(synth.CONFIGS[args.config]) if args.config in synth.CONFIGS else dict(n_img=12, n_desc=1000, n_pt=5000)
    n_img, n_desc, n_pt = cfg["n_img"], cfg["n_desc"], cfg["n_pt"]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ------------------------------------------------------------------ materialised 10k x 10k distance matrix
    gemm = None
    if not args.no_gemm and rank == 0:
        nq = nt = 10000
        dd = gen_sift(2, nq, seed=synth.SEED + 100000)
        q = torch.from_numpy(dd[0]).cuda(); t = torch.from_numpy(dd[1]).cuda()
        qs, ts = ctx.descset_l2(q), ctx.descset_l2(t)
        alg = 4.0 * nq * nt + 4.0 * 128 * (nq + nt)          # 410.2 MB
        legs = {}
        # the write ceiling of THIS device: torch's fill of the same 400 MB, sustained (50 warm-up launches, mean of the next 100)
        flat = torch.empty(nq * nt, dtype=torch.float32, device="cuda")
        for _ in range(50):
            flat.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(100):
            flat.fill_(1.0)
        e1.record(stream); torch.cuda.synchronize()
        store_ceiling_gbs = 4.0 * nq * nt / (e0.elapsed_time(e1) / 100 * 1e-3) / 1e9
        del flat
        # `ms` is the SUSTAINED figure (mean of launches 151..200 of a back-to-back run), `ms_burst` the first ten after 0.3 s of idle;
        # every block of ten is listed.  Round 2's 80 -> 110 -> 90 us curve came from the kernel's compute side (a pure writer holds its
        # rate from the first launch, experiments/distmat_power.py): with the epilogue cut from ~19 to ~11 issue slots per element the
        # kernel runs 80-82 us sustained, 4-5 us above its own store stream (experiments/distmat_modes.py).
        for name, ld in (("ld_10000", nt), ("ld_10016_rows_128B_aligned", 10016)):
            buf = torch.empty((nq, ld), dtype=torch.float32, device="cuda")
            out = buf[:, :nt]
            for _ in range(3):
                ctx.l2_distance_matrix_dev(qs, ts, out)
            torch.cuda.synchronize()
            time.sleep(0.3)
            blocks = []
            for _ in range(20):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(10):
                    ctx.l2_distance_matrix_dev(qs, ts, out)
                e1.record(stream); torch.cuda.synchronize()
                blocks.append(e0.elapsed_time(e1) / 10)
            del out, buf
            ms = sum(blocks[15:]) / 5.0
            legs[name] = dict(ms=ms, ms_burst=blocks[0], ms_per_block_of_10=[round(b, 4) for b in blocks],
                              achieved=alg / (ms * 1e-3) / 1e9, frac=alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              frac_of_store_ceiling=alg / (ms * 1e-3) / 1e9 / store_ceiling_gbs,
                              burst_achieved=alg / (blocks[0] * 1e-3) / 1e9, burst_frac=alg / (blocks[0] * 1e-3) / 1e9 / HBM_PEAK_GBS)
        ref = legs["ld_10000"]                                # the reference's layout (a dense cv::Mat: row stride = nt)
        tr, src = RECORDED_TRAFFIC["distmat_i8_kernel<4>"]
        gemm = dict(kernel="distmat_i8_kernel<4>", workload="10000x10000x128 float32 distance matrix", ms=ref["ms"],
                    bound="hbm", achieved=ref["achieved"], peak=HBM_PEAK_GBS, unit="GB/s", frac=ref["frac"], algorithmic_bytes=alg,
                    store_ceiling_gbs=store_ceiling_gbs, store_ceiling_source="a 400 MB torch fill timed in this run, sustained (100 launches after 50): the write-only ceiling of this device", legs=legs,
                    traffic=tr, traffic_source=src + " (recorded by separate --pmc passes, not measured in this run)")
        del qs, ts

    # ------------------------------------------------------------------ region A: bundle adjustment
    sc = gen_scene(n_img, n_pt)
    pts_l, oc_l, op_l, uv_l, _ = sdist.shard_points(sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], sc["pts0"], rank, world)
    pb = ctx.ba_create(sc["K0"], sc["ext0"], pts_l, oc_l, op_l, uv_l)
    native_comm = None
    if world > 1:
        # production: the in-library RCCL hook (ncclAllReduce on the context's stream, no Python between the LM loop and the
        # collective).  gloo rehearsals (several ranks on one card, SFM_DIST_BACKEND=gloo) go through torch.distributed instead.
        want_native = dist.get_backend() == "nccl" and ctx.rccl_available() and os.environ.get("SFM_NATIVE_RCCL", "1") != "0"
        if want_native:
            try:
                native_comm = sdist.make_native_rccl(ctx)
            except Exception as exc:       # e.g. a second RCCL copy that cannot initialise: every rank must then take the same other path
                print(f"[bench] rank {rank}: in-library RCCL communicator failed ({exc}); falling back to the torch.distributed hook", file=sys.stderr)
                native_comm = None
            ok = torch.tensor([1 if native_comm is not None else 0], dtype=torch.int32, device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0 and native_comm is not None:
                ctx.rccl_comm_destroy(native_comm); native_comm = None
        if native_comm is not None:
            pb.set_rccl(native_comm, rank, world)
        else:
            pb.set_allreduce(sdist.make_allreduce_hook(), rank, world)
    s0 = pb.iterate(args.warmup) if args.warmup > 0 else None
    barrier()
    t0 = time.perf_counter()
    s1 = pb.iterate(args.steps)
    barrier()
    t_ba = max_over_ranks(time.perf_counter() - t0)
    # per-phase / per-kernel device times: the same number of steps again with the library's HIP-event instrumentation on
    # (thirteen event records per iteration cost ~27 us, so the headline steps above run without them)
    ctx.set_kernel_timing(True)
    pb.iterate(args.steps)
    phase = pb.phase_ms()
    ctx.set_kernel_timing(False)
    barrier()
    # What a pair of HIP events around ONE kernel reads beyond the kernel itself (the second event's marker waits for the kernel's
    # completion signal, the kernel's dispatch waits for the first marker): calibrated by bracketing a one-element kernel
    # (~2 us in rocprof) the same way on the same stream.  rocprofv3's kernel durations are begin-to-end of the dispatch and
    # do not contain it; `avg_launch_ms_net` below = event bracket - this overhead, which is what agrees with profiles/.
    xcal = torch.zeros(1, device="cuda")
    br = []
    for _ in range(60):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream); xcal.add_(1.0); e1.record(stream)
        torch.cuda.synchronize()
        br.append(e0.elapsed_time(e1))
    ev_overhead_ms = max(0.0, float(np.median(br[10:])) - 0.002)
    ba_its = args.steps / t_ba
    n_red = 6 * (n_img - 1) + 4
    n_obs = sc["n_obs"]
    # algorithmic bytes per LM iteration: observation records read by the linearisation and by the
    # candidate-cost pass, points read + written, cameras, S written + read by the factorisation + factor written
    b_it = 2 * 24 * n_obs + 2 * 24 * n_pt + 2 * 48 * n_img + 3 * 8 * n_red * n_red

    # ------------------------------------------------------------------ region B: matching
    # C5 names 1000 images x 10k descriptors (999 pairs, 5.1 GB of float rows): that chain is generated ON THE DEVICE (same
    # construction, torch's random streams); the host-side copies that the CPU baseline and the from-host calls need are the
    # first 13 images only
    device_chain = args.config == "C5"
    n_img_match = n_img
    m_steps = args.match_steps if args.match_steps > 0 else min(args.steps, 20)
    m_warm = min(args.warmup, 3)
    pairs_g, images, pairs_l = sdist.shard_pairs(n_img_match, rank, world)
    chain = None
    d_desc, sets = [], []
    if len(images) and not args.no_match:
        # descriptors of image i depend on image i-1: generate the chain up to this rank's last image
        if device_chain:
            dchain = synth.sift_descriptor_chain_device(images[-1] + 1, n_desc)
            chain = [t.cpu().numpy() for t in dchain[:13]] if rank == 0 else None
            for i in images:
                d_desc.append(dchain[i]); sets.append(ctx.descset_l2(dchain[i]))
            del dchain
        else:
            chain = gen_sift(images[-1] + 1, n_desc)
            for i in images:
                t = torch.from_numpy(chain[i]).cuda()
                d_desc.append(t); sets.append(ctx.descset_l2(t))
    n_pairs_l = 0 if args.no_match else pairs_l.shape[0]
    d_matches = torch.zeros((max(n_pairs_l, 1), n_desc, 4), dtype=torch.int32, device="cuda")
    d_counts = torch.zeros((max(n_pairs_l, 1),), dtype=torch.int32, device="cuda")
    h_matches = torch.zeros_like(d_matches, device="cpu").pin_memory()
    h_counts = torch.zeros_like(d_counts, device="cpu").pin_memory()

    # match lists: the ratio-tail kernel writes the surviving matches (and the counts) straight into pinned host memory -- only
    # what survives crosses PCIe (C4: ~9.5 MB per pass instead of the 15.9 MB capacity of the per-pair lists as a D2H copy behind
    # the kernels: 1.34 -> 1.21 ms per pass).  --staged-match-copy restores device buffers + one copy.
    zero_copy = not args.staged_match_copy

    def match_pass():
        if n_pairs_l == 0:
            return
        ctx.refresh_descsets(sets)
        if zero_copy:       # the ratio-tail kernel writes the surviving matches straight into pinned host memory
            ctx.match_pairs_dev(sets, pairs_l, h_matches, n_desc, h_counts)
        else:
            ctx.match_pairs_dev(sets, pairs_l, d_matches, n_desc, d_counts)
            h_matches.copy_(d_matches, non_blocking=True)
            h_counts.copy_(d_counts, non_blocking=True)

    def timed_block(one_pass):
        """m_steps passes between two barriers, twice: the MEAN of the two blocks is quoted, both are reported (round 3 quoted the
        shorter one; the spread between the blocks of one run is 3-15 % on these boxes, slices of shared machines)"""
        blocks = []
        for _ in range(2):
            barrier()
            t0 = time.perf_counter()
            for _ in range(m_steps):
                one_pass()
            barrier()
            blocks.append(max_over_ranks(time.perf_counter() - t0))
        return sum(blocks) / len(blocks), blocks

    for _ in range(m_warm):
        match_pass()
    t_match, t_match_blocks = timed_block(match_pass) if n_pairs_l else (1.0, [1.0, 1.0])
    # the kNN kernel's launch time: HIP events inside the library, on its stream, over further passes (with the events on the library
    # keeps each pass to ONE kNN launch; the timed passes above run uninstrumented, like the BA steps)
    ctx.set_kernel_timing(True)
    for _ in range(min(m_steps, 8) if n_pairs_l else 0):
        match_pass()
    barrier()
    knn_kernel_ms, knn_merge_ms, knn_calls, _ = ctx.match_kernel_ms()
    ctx.set_kernel_timing(False)
    pairs_per_s = (n_img_match - 1) * m_steps / t_match if not args.no_match else None
    n_matches = int(h_counts.sum().item())

    # ------------------------------------------------------------------ region D: the same chain on binary rows (Hamming2)
    ham = None
    if not args.no_hamming and not args.no_match and n_pairs_l:
        if device_chain:
            dbchain = synth.akaze_descriptor_chain_device(images[-1] + 1, n_desc)
            bchain = [t.cpu().numpy() for t in dbchain[:13]]
            d_bin = [dbchain[i] for i in images]
            del dbchain
        else:
            bchain = gen_akaze(images[-1] + 1, n_desc)
            d_bin = [torch.from_numpy(bchain[i]).cuda() for i in images]
        bsets = [ctx.descset_hamming2(t) for t in d_bin]

        def ham_pass():
            ctx.match_pairs_dev(bsets, pairs_l, h_matches, n_desc, h_counts)

        for _ in range(m_warm):
            ham_pass()
        t_ham, t_ham_blocks = timed_block(ham_pass)
        ctx.set_kernel_timing(True)
        for _ in range(min(m_steps, 8)):
            ham_pass()
        barrier()
        ham_kernel_ms, ham_merge_ms, ham_calls, _ = ctx.match_kernel_ms()
        ctx.set_kernel_timing(False)
        ham = dict(t=t_ham, blocks=t_ham_blocks, kernel_ms=ham_kernel_ms, merge_ms=ham_merge_ms, calls=ham_calls, matches=int(h_counts.sum().item()), chain=bchain)
        del bsets, d_bin

    # ------------------------------------------------------------------ region E: two-view DLT triangulation (reconstruct, NView:1117-1159), resident inputs
    tri = None
    if rank == 0 and not args.no_match:
        n_tri = n_pt
        Kc = np.array([[sc["K0"][0], 0, sc["K0"][2]], [0, sc["K0"][1], sc["K0"][3]], [0, 0, 1]], np.float64)
        def proj(ext6):
            w = np.asarray(ext6[:3], np.float64); th = np.linalg.norm(w)
            Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
            R = np.eye(3) if th < 1e-12 else np.eye(3) + np.sin(th) / th * Kx + (1 - np.cos(th)) / th ** 2 * Kx @ Kx
            return (Kc @ np.concatenate([R, np.asarray(ext6[3:6], np.float64).reshape(3, 1)], 1)).astype(np.float32)
        P1, P2 = proj(sc["ext0"].reshape(-1, 6)[0]), proj(sc["ext0"].reshape(-1, 6)[1])
        # the points cameras 0 and 1 both observe (what a matched image pair hands to reconstruct()), tiled to the scene's point count
        oc_a = np.asarray(sc["obs_cam"]); op_a = np.asarray(sc["obs_pt"])
        both = np.intersect1d(op_a[oc_a == 0], op_a[oc_a == 1])
        if both.size == 0:
            both = np.arange(min(n_tri, 1000))
        X3 = np.asarray(sc["pts0"], np.float64).reshape(-1, 3)[both]
        X = np.concatenate([np.tile(X3, (n_tri // X3.shape[0] + 1, 1))[:n_tri], np.ones((n_tri, 1))], 1)
        def pix(P):
            h = X @ P.astype(np.float64).T
            return (h[:, :2] / h[:, 2:3]).astype(np.float32)
        d_xy1 = torch.from_numpy(pix(P1)).cuda(); d_xy2 = torch.from_numpy(pix(P2)).cuda()
        d_xyzw = torch.empty((4, n_tri), dtype=torch.float32, device="cuda"); d_xyz = torch.empty((n_tri, 3), dtype=torch.float64, device="cuda")
        for _ in range(5):
            ctx.triangulate2_dev(P1, P2, d_xy1, d_xy2, d_xyzw, d_xyz)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            ctx.triangulate2_dev(P1, P2, d_xy1, d_xy2, d_xyzw, d_xyz)
        torch.cuda.synchronize()
        t_tri = (time.perf_counter() - t0) / 50
        tri = {"value": n_tri / t_tri, "unit": "points/s", "points_per_call": n_tri, "ms_per_call": 1e3 * t_tri, "calls_timed": 50,
               "algorithmic_bytes_per_point": 56, "achieved_gbs": 56.0 * n_tri / t_tri / 1e9,
               "distinct_points": int(both.size),
               "what": "sfmhip_triangulate2_f32_dev: the points cameras 0 and 1 both observe, tiled to the scene's point count (float32 projection matrices and pixels, fp64 Jacobi SVD of the 4x4 "
                       "system per point, float32 homogeneous + fp64 de-homogenised output); 16 B in + 40 B out per point: a compute-bound kernel"}
        del d_xy1, d_xy2, d_xyzw, d_xyz

    # ------------------------------------------------------------------ region C: the drop-in calls, from host arrays
    e2e = None
    if not args.no_end_to_end and rank == 0 and world == 1:
        import ctypes as C
        from sfm_opencv_amd._lib import BASummary
        oc = np.ascontiguousarray(sc["obs_cam"], np.int32); op_ = np.ascontiguousarray(sc["obs_pt"], np.int32)
        uv = np.ascontiguousarray(sc["obs_uv"], np.float64)
        opts = ctx.ba_options()
        runs = []
        for rep in range(4):        # rep 0 also pays first-use costs (staging buffers, cached device blocks): listed, not used
            K = sc["K0"].copy(); ext = sc["ext0"].copy(); pts = sc["pts0"].copy()
            sm = BASummary()
            torch.cuda.synchronize()
            tc = time.perf_counter()
            rc = ctx.lib.sfmhip_ba_solve(ctx.h, K.ctypes.data, ext.ctypes.data, n_img, pts.ctypes.data, n_pt, oc.ctypes.data, op_.ctypes.data,
                                         uv.ctypes.data, oc.shape[0], C.byref(opts), C.byref(sm))
            wall = time.perf_counter() - tc
            assert rc == 0, rc
            runs.append(dict(wall_ms=1e3 * wall, create_ms=1e3 * sm.preprocessor_time_s, run_ms=1e3 * sm.minimizer_time_s,
                             writeback_ms=1e3 * sm.postprocessor_time_s, summary_total_ms=1e3 * sm.total_time_s, iterations=sm.iterations,
                             successful_steps=sm.successful_steps, termination=sm.termination, initial_cost=sm.initial_cost, final_cost=sm.final_cost))
        best = min(runs[1:], key=lambda r: r["wall_ms"])
        e2e = dict(best, calls=[round(r["wall_ms"], 3) for r in runs],
                   what="sfmhip_ba_solve(host arrays): uploads + orderings + pair lists on the device + solver plan (create_ms), LM loop to Ceres-default "
                        "termination (run_ms), parameters copied back in the caller's order + teardown (writeback_ms); best of calls 2-4, every call listed")
        if not args.no_match and chain is not None:
            mruns = []
            for rep in range(3):
                tc = time.perf_counter()
                got = api.match_features_for_all(chain[:n_img_match], ctx=ctx)
                mruns.append(1e3 * (time.perf_counter() - tc))
            n_host = min(len(chain), n_img_match)
            e2e["match_pairs_from_host"] = dict(ms=min(mruns[1:]), calls=[round(x, 3) for x in mruns], pairs=n_host - 1,
                                                pairs_per_sec=(n_host - 1) / (min(mruns[1:]) * 1e-3), matches=int(sum(len(g) for g in got)),
                                                host_bytes_uploaded=int(sum(c.nbytes for c in chain[:n_img_match])),
                                                what="match_features_for_all on host descriptor matrices: ONE sfmhip_descsets_create_l2_host for the chain (the staging threads "
                                                     "convert + verify the integer-valued float rows to bytes on their way into the pinned ring: 128 B per row over PCIe instead "
                                                     "of 512; one preparation launch), one batched kNN-2 + ratio tail, match lists back on the host.  Host DRAM-read bound: "
                                                     "5.1-8.3 ms from box to box (rounds 1-3: 16.8-22 ms)")
            if ham is not None:
                hruns = []
                for rep in range(3):
                    tc = time.perf_counter()
                    got = api.match_features_for_all(ham["chain"][:n_img_match], ctx=ctx)
                    hruns.append(1e3 * (time.perf_counter() - tc))
                e2e["match_pairs_from_host_hamming2"] = dict(ms=min(hruns[1:]), calls=[round(x, 3) for x in hruns], pairs=n_host - 1,
                                                             pairs_per_sec=(n_host - 1) / (min(hruns[1:]) * 1e-3), matches=int(sum(len(g) for g in got)))

    # ------------------------------------------------------------------ CPU baseline (rank 0, bounded sample)
    cpu = cpu4 = None
    if not args.no_cpu_baseline and rank == 0 and world == 1:      # reported at N=1 only
        import oracle as orc
        cores = args.cpu_threads if args.cpu_threads > 0 else min(len(os.sched_getaffinity(0)), 16)     # the 1-GPU box's CPU share is 16 cores
        big = args.config in ("C4", "C5")

        def cpu_ba(threads, n_it):
            orc.set_num_threads(threads)
            tc = time.perf_counter()
            orc.ba_solve(sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"], force_iterations=n_it)
            return n_it / (time.perf_counter() - tc)

        def cpu_match(threads, n_pairs_cpu):
            if args.no_match or n_pairs_cpu <= 0:
                return None
            orc.set_num_threads(threads)
            ch = chain if chain is not None and len(chain) > n_pairs_cpu else gen_sift(n_pairs_cpu + 1, n_desc)
            tc = time.perf_counter()
            for i in range(n_pairs_cpu):
                orc.match_features_l2(ch[i], ch[i + 1])
            return n_pairs_cpu / (time.perf_counter() - tc)

        what = ("forced LM iterations of the same %d-camera/%d-point scene (incl. one extra linearisation for the column "
                "scaling) and %d chain pairs of %dx%dx128 matching; oracle/ C restatement with OpenMP (an unoptimised "
                "checker: 13-wide dual-number Jacobians, skyline Cholesky), not OpenCV/Ceres (unbuildable offline)")
        n_it_cpu = (30 if args.config == "C4" else 6) if big else 10          # ~10 s of CPU work at C4 (3 it/s on 16 threads)
        cpu_pairs = min(100 if args.config != "C5" else 12, n_img_match - 1, (len(chain) - 1) if chain is not None else 0)
        cpu = dict(value=cpu_ba(cores, n_it_cpu), unit="it/s", cores=cores, kind="port",
                   sample=("%d " % n_it_cpu) + what % (n_img, n_pt, cpu_pairs, n_desc, n_desc),
                   matched_pairs_per_sec=cpu_match(cores, cpu_pairs))
        # what the reference asks of Ceres: options.num_threads = 4 (NViewReconstuct.cpp:1218)
        n_it4 = (10 if args.config == "C4" else 3) if big else 10
        pairs4 = min(25 if args.config != "C5" else 4, n_img_match - 1, (len(chain) - 1) if chain is not None else 0)
        cpu4 = dict(value=cpu_ba(4, n_it4), unit="it/s", cores=4, kind="port",
                    sample=("%d " % n_it4) + what % (n_img, n_pt, pairs4, n_desc, n_desc),
                    matched_pairs_per_sec=cpu_match(4, pairs4))
        if e2e is not None and args.config != "C5":
            # the same call on the CPU restatement, to convergence (C4: ~15 s on 16 threads; at C5 it would take minutes: not run)
            orc.set_num_threads(cores)
            tc = time.perf_counter()
            so = orc.ba_solve(sc["K0"], sc["ext0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_uv"])[3]
            e2e["cpu_port"] = dict(wall_ms=1e3 * (time.perf_counter() - tc), cores=cores, iterations=so["iterations"], final_cost=so["final_cost"],
                                   kind="port", what="oracle/ orc.ba_solve end to end (problem set-up + LM to the same termination rules), not Ceres")
        if ham is not None:
            orc.set_num_threads(cores)
            hp = min(40 if args.config != "C5" else 6, n_img_match - 1, len(ham["chain"]) - 1)
            tc = time.perf_counter()
            for i in range(hp):
                orc.match_features_hamming2(ham["chain"][i], ham["chain"][i + 1])
            ham["cpu"] = dict(value=hp / (time.perf_counter() - tc), unit="pairs/s", cores=cores, kind="port", sample="%d chain pairs" % hp)

    if rank == 0:
        # single kernels of the LM iteration timed with HIP events inside the library (same stream, instrumented steps):
        # `roofline` is the one with the largest launch time -- the kernel that dominates the region `value` is measured on
        L = np.bincount(sc["obs_pt"], minlength=n_pt).astype(np.int64)
        n_pairs_items = int((L * (L - 1) // 2).sum())
        # algorithmic bytes of a kernel = what it must move once (compulsory traffic): its records, every point block it reads,
        # its partial sums; `gather_bytes` = what its threads request (a point's blocks once per observation / pair), mostly
        # served by the XCD L2s because neighbouring cameras share their points -- reported beside it, not priced against HBM
        pt_blocks = n_pt * (24 + 24 + 48)                       # point, column scales, V^-1
        pair_alg = n_pairs_items * 16 + pt_blocks + n_obs * 16 + 36 * 8 * (n_pairs_items // 512 + 6 * n_img)
        cam_alg = n_obs * 20 + pt_blocks + n_pt * (24 + 96) + 2 * 80 * 8 * n_img
        pair_gather = n_pairs_items * (16 + 24 + 24 + 48 + 32)
        cam_gather = 2 * n_obs * (4 + 16 + 24 + 24 + 48) + n_obs * (24 + 96)
        pair_what = "pair records 16 B + every point's 96 B of blocks once + pixels 16 B per observation + 288 B per partial"
        cam_what = "observation records 20 B + every point's 96 B of blocks + b_p / W_K 120 B once + partials"
        kern = {SOLVER_KERNEL: dict(ms=phase[6], bytes=2 * 8 * 1024 * phase[7], gather=None, what="non-zero 32x32 blocks of S read and of L written")}
        if phase[5] > 0:            # large problems: the pair kernel and the camera kernel as two launches on two streams
            kern["ba_schur_kernel"] = dict(ms=phase[5], bytes=pair_alg, gather=pair_gather, what=pair_what)
            kern["ba_camera_kernel"] = dict(ms=phase[4], bytes=cam_alg, gather=cam_gather, what=cam_what)
        else:                       # one launch runs both kinds of workgroup; the point blocks are shared between them
            kern["ba_camschur_kernel"] = dict(ms=phase[4], bytes=pair_alg + cam_alg - pt_blocks, gather=pair_gather + cam_gather,
                                              what=pair_what + "; " + cam_what + " (point blocks counted once)")
        dom = max(kern, key=lambda k: kern[k]["ms"])
        kd = kern[dom]
        # SURVEY 8d prices an LM iteration at B_it = 2*24*No + 2*24*Np + 2*48*Nc + 3*8*n^2; the linearisation's share of it is the
        # observation records read once, the points read once and S written once.  That share is the `algorithmic_bytes` of the
        # linearisation launch (one shared launch up to a few thousand workgroups; two concurrent launches on two streams beyond:
        # the share then belongs to the pair, priced against the longer one).  The builder's own byte models of round 2 (what the
        # kernel must move once incl. its partial sums / what its threads request) are kept as extra keys, not used for `frac`.
        lin_share = 24 * n_obs + 24 * n_pt + 8 * n_red * n_red
        is_lin = dom in ("ba_camschur_kernel", "ba_camera_kernel", "ba_schur_kernel")
        alg_bytes = lin_share if is_lin else kd["bytes"]
        net_ms = max(kd["ms"] - ev_overhead_ms, 1e-9)
        c4_single = world == 1 and args.config == "C4"
        roof = {"kernel": dom, "bound": "hbm", "achieved": alg_bytes / (net_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": alg_bytes / (net_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "traffic": RECORDED_TRAFFIC[dom][0] if (dom in RECORDED_TRAFFIC and c4_single) else None,
                "traffic_source": (RECORDED_TRAFFIC[dom][1] + " (recorded by separate --pmc passes, not measured in this run)")
                                  if (dom in RECORDED_TRAFFIC and c4_single) else None,
                "algorithmic_bytes": alg_bytes,
                "algorithmic_bytes_model": ("SURVEY 8d share of B_it for the linearisation: 24*No + 24*Np + 8*n^2" if is_lin else kd["what"]),
                "avg_launch_ms": kd["ms"], "event_bracket_overhead_ms": ev_overhead_ms, "avg_launch_ms_net": net_ms,
                "timing": "HIP events on the launch stream inside libsfmhip; _net subtracts the bracket overhead calibrated in this run "
                          "(a one-element kernel bracketed the same way) and is the figure comparable with rocprofv3's kernel duration in profiles/",
                # what actually limits this launch: fp64 VALU issue.  SQ_INSTS_VALU of a separate --pmc pass (recorded, C4 only);
                # an fp64 wave instruction holds its SIMD for 4 cycles; 1,024 SIMDs at the 2.35 GHz the kernel runs at
                "limiter": "valu_fp64" if is_lin else "latency",
                "valu_issue": ({"wave_instructions": 2.62e7, "frac_of_issue_slots": 2.62e7 * 4 / (1024 * 2.35e9 * net_ms * 1e-3),
                                "source": "profiles/r03_traffic_pmc.md: SQ_INSTS_VALU 26.2 M, GRBM_GUI_ACTIVE / 8 XCDs = 2.35 GHz (recorded by a separate --pmc pass of round 3, "
                                          "not measured in this run)"}
                               if (dom == "ba_camschur_kernel" and c4_single) else None),
                "builder_models": {"compulsory_bytes": kd["bytes"], "compulsory_bytes_what": kd["what"], "gather_bytes_requested": kd["gather"]},
                "note": f"largest kernel of one LM iteration (the region `value` is measured on); the iteration as a whole: "
                        f"{phase[3]:.3f} ms of device time for B_it = {b_it / 1e6:.1f} MB, see roofline_lm_iteration and DESIGN.md 7"}
        roof_knn = None
        if n_pairs_l and knn_calls:
            ops = 2.0 * n_desc * n_desc * 128 * n_pairs_l       # 2 * nq * nt * dim per pair
            tr, src = RECORDED_TRAFFIC["knn2_i8_kernel<4>"] if (world == 1 and args.config == "C4") else (None, None)
            knn_net = max(knn_kernel_ms - ev_overhead_ms, 1e-9)
            roof_knn = {"kernel": "knn2_i8_kernel<4>", "bound": "mfma", "achieved": ops / (knn_net * 1e-3) / 1e12,
                        "peak": I8_MFMA_PEAK_TOPS, "unit": "TOP/s", "frac": ops / (knn_net * 1e-3) / 1e12 / I8_MFMA_PEAK_TOPS,
                        "event_bracket_overhead_ms": ev_overhead_ms, "avg_launch_ms_net": knn_net,
                        "traffic": tr, "traffic_source": (src + " (recorded by separate --pmc passes, not measured in this run)") if src else None,
                        "traffic_unit": "bytes per launch, L2 fabric side: 2 x FETCH_SIZE + WRITE_SIZE",
                        "algorithmic_ops": ops, "avg_launch_ms": knn_kernel_ms, "launches_timed": int(knn_calls),
                        "merge_rescore_ms": knn_merge_ms,
                        "note": "one launch = all chain pairs of this rank; peak = dense int8 MFMA at 2.4 GHz (measured sustained "
                                "4.2 POP/s, experiments/mfma_i8_bench.hip)"}
        roof_ham = ham_out = None
        if ham is not None and ham["calls"]:
            # knn2_hamming2_fp4_kernel (round 3): every two-bit cell as the simplex triple (s0, s1, s0 s1) in FP4, the distance out of
            # the dot product of two rows' 3 x 244 = 732 values (padded to 768 = 12 K-steps of v_mfma_scale_f32_32x32x64_f8f6f4).
            # Priced on the 732 useful multiply-adds per row pair against the dense FP4 MFMA peak.
            flops = 2.0 * 732 * n_desc * n_desc * n_pairs_l
            ham_net = max(ham["kernel_ms"] - ev_overhead_ms, 1e-9)
            roof_ham = {"kernel": "knn2_hamming2_fp4_kernel", "bound": "mfma", "achieved": flops / (ham_net * 1e-3) / 1e12,
                        "peak": FP4_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flops / (ham_net * 1e-3) / 1e12 / FP4_MFMA_PEAK_TFLOPS,
                        "algorithmic_ops": flops, "avg_launch_ms": ham["kernel_ms"], "event_bracket_overhead_ms": ev_overhead_ms,
                        "avg_launch_ms_net": ham_net, "launches_timed": int(ham["calls"]), "merge_ms": ham["merge_ms"],
                        "traffic": RECORDED_TRAFFIC["knn2_hamming2_fp4_kernel"][0] if (world == 1 and args.config == "C4") else None,
                        "traffic_source": (RECORDED_TRAFFIC["knn2_hamming2_fp4_kernel"][1] + " (recorded by separate --pmc passes, not measured in this run)") if (world == 1 and args.config == "C4") else None,
                        "algorithmic_bytes": 2.0 * 384 * n_desc * n_pairs_l + 16.0 * n_desc * n_pairs_l,
                        "issued": "12 K-steps x 2 MFMAs of 32 cycles per (64 queries x 32 trains) on padded 5120-row sets: 1.91e6 matrix-pipe cycles per SIMD and launch "
                                  "(SQ_VALU_MFMA_BUSY_CYCLES, profiles/r03_hamming_fp4_pmc.md) = 0.80 ms at 2.4 GHz",
                        "note": "one launch = all chain pairs of this rank; the VALU popcount kernel of rounds 1-2 (knn2_hamming2_kernel, 3.33 ms, still used "
                                "for 62..64-byte rows) ran at the VALU issue rate; peak = dense FP4 MFMA (MI355X_MICROARCH.md)"}
            ham_out = {"value": (n_img_match - 1) * m_steps / ham["t"], "ms_per_pass": 1e3 * ham["t"] / m_steps, "pairs": n_img_match - 1,
                       "passes_timed": m_steps, "blocks_ms_per_pass": [1e3 * b / m_steps for b in ham["blocks"]], "matches_rank0": ham["matches"], "descriptor": "61-byte rows (AKAZE M-LDB shape), NORM_HAMMING2",
                       "cpu_baseline": ham.get("cpu"),
                       "includes": "kNN-2 (Hamming2) + ratio tail + match lists written to pinned host memory; re-encoded rows resident in HBM"}
        out = {
            "metric": "ba_iterations_per_sec", "value": ba_its, "unit": "it/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * t_ba / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic (SURVEY 8d constructions; " + ("std::mt19937_64, seed 20240607 + i" if mt else "numpy PCG64, seed 20240607 + i") + (", C5 descriptor chain: torch generator on the device" if device_chain else "") + ")",
            "config": {"workload": f"{args.config}: {n_img} images x {n_desc} SIFT-like descriptors ({n_img_match - 1} chain pairs matched), "
                                   f"{n_img} cameras / {n_pt} points / {n_obs} observations BA",
                       "parallelism": f"points (by first camera) + pairs sharded over {world} rank(s), cameras replicated, 1 all-reduce per LM iteration (packed reduced-system "
                                      f"message of the speculative next linearisation + the 5 step scalars; {'in-library RCCL hook' if native_comm is not None else ('torch.distributed hook' if world > 1 else 'no exchange')})",
                       "reduced_system_order": n_red},
            "roofline": roof,
            "roofline_lm_iteration": {"bound": "hbm", "achieved": b_it / (phase[3] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": b_it / (phase[3] * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": b_it, "device_ms": phase[3]},
            "roofline_knn": roof_knn,
            "ba_kernel_ms": {k: v["ms"] for k, v in kern.items()},
            "ba_phase_ms": {"linearize_schur": phase[0], "reduced_solve": phase[1], "backsub_cost": phase[2], "total_device": phase[3],
                            "measured": f"HIP events over {args.steps} further steps after the timed ones (instrumentation off during the timed steps)"},
            "ba_cost": {"initial": (s0 or s1)["initial_cost"], "after_timed_steps": s1["final_cost"],
                        "successful_steps": s1["successful_steps"], "iterations": s1["iterations"]},
            "matched_pairs_per_sec": None if args.no_match else
                {"value": pairs_per_s, "ms_per_pass": 1e3 * t_match / m_steps, "pairs": n_img_match - 1, "passes_timed": m_steps,
                 "blocks_ms_per_pass": [1e3 * b / m_steps for b in t_match_blocks],
                 "matches_rank0": n_matches, "includes": "prep + kNN-2 + ratio tail + match lists in host memory (" + ("device buffers + one D2H copy" if args.staged_match_copy else "written to pinned host memory by the ratio-tail kernel") + ")"},
            "matched_pairs_per_sec_hamming2": ham_out,
            "roofline_hamming2": roof_ham,
            "triangulated_points_per_sec": tri,
            "ba_solve_end_to_end": e2e,
            "roofline_gemm": gemm,
            "cpu_baseline": cpu,
            "cpu_baseline_4thr": cpu4,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
