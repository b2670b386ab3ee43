"""Print the kernel timeline of one LM iteration from a rocprofv3 --kernel-trace CSV (start, end, duration in us, queue)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
which = int(sys.argv[2]) if len(sys.argv) > 2 else 150
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("ba_point")]
j = idx[which]
t0 = int(rows[j]["Start_Timestamp"])
for r in rows[j:idx[which + 1] + 1]:
    print(f"{(int(r['Start_Timestamp']) - t0) / 1000:8.1f} {(int(r['End_Timestamp']) - t0) / 1000:8.1f} "
          f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000:7.1f} q{r['Queue_Id']} {r['Kernel_Name'][:70]}")
per = [(int(rows[idx[i + 1]]["Start_Timestamp"]) - int(rows[idx[i]]["Start_Timestamp"])) / 1000 for i in range(40, len(idx) - 1)]
print("median period %.1f us over %d iterations" % (sorted(per)[len(per) // 2], len(per)))
