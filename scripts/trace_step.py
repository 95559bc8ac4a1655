#!/usr/bin/env python3
"""Launch-by-launch timeline of ONE training step from a rocprofv3 kernel trace (…_kernel_trace.csv): every launch between two
consecutive standardize_kernel launches (the step's first kernel), with its start offset, duration, gap to the previous launch and
grid.  python scripts/trace_step.py <kernel_trace.csv> [step index from the end, default 2] [min us to list, default 0]"""
import csv
import sys

src = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
rows = sorted(csv.DictReader(open(src)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "standardize_kernel" in r["Kernel_Name"]]
lo, hi = starts[-back - 1], starts[-back]
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = t0
busy = 0
print(f"# step = launches {lo}..{hi - 1} of {src.split('/')[-1]}: {(int(rows[hi]['Start_Timestamp']) - t0) / 1e6:.3f} ms\n")
print("| # | start ms | us | gap us | grid | kernel |\n|---|---|---|---|---|---|")
for i in range(lo, hi):
    r = rows[i]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:64]
    busy += e - s
    if (e - s) / 1e3 >= min_us:
        print(f"| {i - lo} | {(s - t0) / 1e6:.3f} | {(e - s) / 1e3:.1f} | {(s - prev_end) / 1e3:.1f} | "
              f"{int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)}x{r['Grid_Size_Y']} | {name} |")
    prev_end = max(prev_end, e)
print(f"\nsum of kernel durations {busy / 1e6:.3f} ms")
