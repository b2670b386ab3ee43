"""Print the kernel timeline (start offset, duration, queue) of the last passes from a rocprofv3 --kernel-trace CSV.
usage: python experiments/trace_timeline.py <kernel_trace.csv> [n_last_kernels]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the matching region: the last ratio_tail launches
idx = [i for i, r in enumerate(rows) if "ratio_tail" in r["Kernel_Name"]]
if idx:
    hi = idx[len(idx) // 2]
    rows = rows[max(0, hi - n):hi + 4]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s = int(r["Start_Timestamp"]) - t0; e = int(r["End_Timestamp"]) - t0
    print("%10.1f %10.1f %8.1f us  q=%s  %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:60]))
