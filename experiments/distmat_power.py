"""is the 80 -> 110 -> 90 us curve of the distance-matrix kernel a clock / power effect, and does a pure writer
of the same bytes show it too?  Three back-to-back runs of 300 launches, each in blocks of 10 timed with events, with a side thread
sampling sclk / mclk / socket power from sysfs:
  (i)   the distance-matrix kernel (ld = 10000);
  (ii)  a write-only stream of the same 400 MB (torch's fill kernel);
  (iii) the distance-matrix kernel again after 1 s of idle.
usage: python experiments/distmat_power.py > profiles/r03_distmat_power.log"""
import io, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfm_opencv_amd import api, synth
import sample_power

ctx = api.Context(0, use_torch_stream=True)
nq = nt = 10000
dd = synth.sift_descriptor_chain(2, nq, seed=synth.SEED + 100000)
q = torch.from_numpy(dd[0]).cuda(); t = torch.from_numpy(dd[1]).cuda()
qs, ts = ctx.descset_l2(q), ctx.descset_l2(t)
stream = torch.cuda.current_stream()
out = torch.empty((nq, nt), dtype=torch.float32, device="cuda")
alg = 4.0 * nq * nt + 4.0 * 128 * (nq + nt)


def run(name, fn, bytes_):
    buf = io.StringIO()
    th = threading.Thread(target=sample_power.main, args=(1e9, buf), daemon=True)
    stop = [False]
    # sampler with a stop flag
    def sampler():
        d = sample_power.find()
        t0 = time.perf_counter()
        while not stop[0]:
            if d is None:
                break
            import glob
            hw = sorted(glob.glob(os.path.join(d, "hwmon", "hwmon*")))
            hw = hw[0] if hw else ""
            pw = sample_power.first([os.path.join(hw, "power1_input"), os.path.join(hw, "power1_average")])
            f1 = sample_power.first([os.path.join(hw, "freq1_input")])
            buf.write("%9.2f sclk %s freq1 %s mclk %s power %s\n" % (1e3 * (time.perf_counter() - t0), sample_power.cur_mhz(os.path.join(d, "pp_dpm_sclk")), f1,
                                                                      sample_power.cur_mhz(os.path.join(d, "pp_dpm_mclk")), pw))
            time.sleep(0.004)
    th = threading.Thread(target=sampler, daemon=True)
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); time.sleep(1.0)
    th.start()
    blocks = []
    t_start = time.perf_counter()
    for b in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(10):
            fn()
        e1.record(stream); torch.cuda.synchronize()
        blocks.append((1e3 * (time.perf_counter() - t_start), e0.elapsed_time(e1) / 10 * 1e3))
    stop[0] = True; th.join()
    print("== %s: us per launch in blocks of 10 (block end ms: us -> TB/s)" % name)
    print("   " + "  ".join("%.0f:%.1f->%.2f" % (tb, us, bytes_ / us / 1e6) for tb, us in blocks))
    lines = buf.getvalue().strip().splitlines()
    print("   sysfs samples (%d), every 5th:" % len(lines))
    for l in lines[::5]:
        print("   " + l)
    sys.stdout.flush()


run("(i) distmat_i8_kernel, ld 10000", lambda: ctx.l2_distance_matrix_dev(qs, ts, out), alg)
run("(ii) write-only stream of 400 MB (torch fill)", lambda: out.fill_(1.0), 4.0 * nq * nt)
run("(iii) distmat_i8_kernel again", lambda: ctx.l2_distance_matrix_dev(qs, ts, out), alg)
