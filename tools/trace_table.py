"""dev tool: per-launch durations of the stepping kernels from a rocprofv3 kernel trace (csv), in launch order.
    python tools/trace_table.py <kernel_trace.csv> [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 80
k = 0
for r in rows:
    nm = r["Kernel_Name"]
    if "k_run_team" not in nm:
        continue
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tag = nm[nm.index("Topo"):nm.index("(")][:60]
    if k < nshow:
        print(k, tag, "%.1f us" % dur)
    k += 1
print("launches", k)
