"""Subprocess A/B of kNN kernel knock-outs (SFMHIP_EXP_KNN is read once per process)."""
import os, sys, json, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for mode in sys.argv[1:] or ["0", "1", "3"]:
    env = dict(os.environ, SFMHIP_EXP_KNN=mode)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-gemm", "--steps", "4", "--warmup", "1"],
                         env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    print("mode", mode, "pass ms %.3f  knn+merge+tail ms %.3f" % (d["matched_pairs_per_sec"]["ms_per_pass"], d["roofline_knn"]["ms_per_pass_rank0"]), flush=True)
