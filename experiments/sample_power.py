"""Clock / power sampler: prints `t_ms sclk_MHz mclk_MHz power_W` every ~5 ms from the first amdgpu device's sysfs nodes (what
rocm-smi reads), until killed or for argv[1] seconds.  Run beside a kernel loop to see whether a rate change is a clock change."""
import glob, os, sys, time

def find():
    for d in sorted(glob.glob("/sys/class/drm/card*/device")):
        if os.path.exists(os.path.join(d, "pp_dpm_sclk")):
            return d
    return None

def cur_mhz(path):
    try:
        for l in open(path):
            if "*" in l:
                return l.split(":")[1].strip().split("Mhz")[0].strip()
    except OSError:
        pass
    return "nan"

def first(paths):
    for p in paths:
        try:
            return open(p).read().strip()
        except OSError:
            continue
    return "nan"

def main(duration=10.0, out=sys.stdout):
    d = find()
    if d is None:
        print("no amdgpu sysfs device readable", file=out); return
    hw = sorted(glob.glob(os.path.join(d, "hwmon", "hwmon*")))
    hw = hw[0] if hw else ""
    t0 = time.perf_counter()
    print("# device", d, "hwmon", hw, file=out)
    while time.perf_counter() - t0 < duration:
        sclk = cur_mhz(os.path.join(d, "pp_dpm_sclk")); mclk = cur_mhz(os.path.join(d, "pp_dpm_mclk"))
        f1 = first([os.path.join(hw, "freq1_input")])
        pw = first([os.path.join(hw, "power1_input"), os.path.join(hw, "power1_average")])
        try:
            pw = "%.0f" % (float(pw) / 1e6)
        except ValueError:
            pass
        try:
            f1 = "%.0f" % (float(f1) / 1e6)
        except ValueError:
            pass
        print("%9.2f %s %s %s %s" % (1e3 * (time.perf_counter() - t0), sclk, f1, mclk, pw), file=out, flush=True)
        time.sleep(0.005)

if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 10.0)
