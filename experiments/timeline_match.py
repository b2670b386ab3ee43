"""Kernel timeline of one matching pass from a rocprofv3 --kernel-trace CSV of `bench.py --no-gemm --no-cpu-baseline`."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("prep_l2_batched")]
j = idx[int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) - 3]
t0 = int(rows[j]["Start_Timestamp"])
nxt = idx[idx.index(j) + 1]
for r in rows[j:nxt + 1]:
    print(f"{(int(r['Start_Timestamp']) - t0) / 1000:8.1f} {(int(r['End_Timestamp']) - t0) / 1000:8.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000:7.1f} {r['Kernel_Name'][:80]}")
