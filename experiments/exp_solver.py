import os, sys, json, subprocess
for dbg in (0, 1, 2, 4, 8, 15):
    env = dict(os.environ, SFMHIP_EXP_SOLVER=str(dbg))
    out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-gemm", "--steps", "6", "--warmup", "1"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    print("dbg", dbg, "solve ms %.3f" % d["ba_phase_ms"]["reduced_solve"])
