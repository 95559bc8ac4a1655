#!/usr/bin/env python3
"""kernel_stats.csv (rocprofv3 --kernel-trace --stats) -> short markdown table for profiles/."""
import csv, sys
src, title = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""
rows = list(csv.DictReader(open(src)))
print(f"# {title}\n")
print("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
for r in rows[:28]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    name = name.split("(")[0][:70] if not name.startswith("at::") else name[:60]
    print(f"| {name} | {r['Calls']} | {int(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |")
