"""Repeatability check of the matching pass: runs bench.py's matching region a few times in fresh processes."""
import os, sys, json, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-gemm", "--steps", "4", "--warmup", "1"],
                         capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    print("rep", rep, "pass ms %.3f  kNN kernel ms %.3f  merge+rescore ms %.3f  frac %.3f" % (
        d["matched_pairs_per_sec"]["ms_per_pass"], d["roofline"]["avg_launch_ms"], d["roofline"]["merge_rescore_ms"], d["roofline"]["frac"]), flush=True)
