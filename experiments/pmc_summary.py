"""median per kernel of one counter from a rocprofv3 --pmc run (counter_collection.csv): python experiments/pmc_summary.py <csv> [counter]"""
import csv, sys, statistics as st
rows = list(csv.DictReader(open(sys.argv[1])))
cn = sys.argv[2] if len(sys.argv) > 2 else rows[0]["Counter_Name"]
by = {}
for r in rows:
    if r["Counter_Name"] != cn:
        continue
    by.setdefault(r["Kernel_Name"].split("(")[0][:60], []).append(float(r["Counter_Value"]))
for k, v in sorted(by.items(), key=lambda kv: -st.median(kv[1])):
    print(f"{cn:12s} {k:62s} n={len(v):5d} median {st.median(v):14.1f} max {max(v):14.1f}")
