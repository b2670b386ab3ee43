"""median of every counter of one kernel (name substring) in rocprofv3 --pmc counter_collection.csv files:
python experiments/pmc_kernel.py <substring> <csv> [<csv> ...]"""
import csv, sys, statistics as st
sub = sys.argv[1]
for f in sys.argv[2:]:
    rows = [r for r in csv.DictReader(open(f)) if sub in r["Kernel_Name"]]
    by = {}
    for r in rows:
        by.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in by.items():
        print("%-28s n=%3d median %.5g" % (k, len(v), st.median(v)))
    if rows:
        r = rows[0]
        print("   ", {k: r[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size") if k in r})
