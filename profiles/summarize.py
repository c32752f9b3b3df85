"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


for f in find("*kernel_stats.csv"):
    print("== kernel stats:", f)
    for row in list(csv.DictReader(open(f)))[:6]:
        d = {k: row[k] for k in row if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")}
        d["Name"] = d["Name"][:70]
        print(d)

for f in find("*counter_collection.csv"):
    print("== counters:", f)
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        if "rcsh::k_run" not in name and "rcsh::k_cartesian" not in name and "rcsh::k_render" not in name:
            continue
        acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[(name, row["Counter_Name"])] += 1
    for name, d in acc.items():
        print(name[:80])
        for c, v in sorted(d.items()):
            n = cnt[(name, c)]
            print(f"   {c}: total {v:.4g} over {n} dispatches -> {v / n:.6g} per dispatch")
