"""dev tool: per-launch durations of the lean and the contact-resolving launch over a rollout, from a rocprofv3 kernel trace
    rocprofv3 --kernel-trace --output-format csv -d D -o t -- python bench.py --steps 1000 --warmup 50 --no-cpu-baseline
    python tools/esc_trace.py D/t_kernel_trace.csv [bucket]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        nm = r["Kernel_Name"]
        if "k_run_team" not in nm:
            continue
        con = "false, false, true, false" in nm or "false, false, true, true" in nm
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), con))
rows.sort()
bucket = int(sys.argv[2]) if len(sys.argv) > 2 else 100
lean = [(s, e) for s, e, c in rows if not c]
con = [(s, e) for s, e, c in rows if c]
print(f"{len(lean)} lean launches, {len(con)} contact-resolving launches")
for name, seq in (("lean", lean), ("contact-resolving", con)):
    for i in range(0, len(seq), bucket):
        chunk = seq[i:i + bucket]
        d = [e - s for s, e in chunk]
        span = (chunk[-1][1] - chunk[0][0]) / len(chunk)
        print(f"  {name:18s} launches {i:5d}..{i + len(chunk) - 1:5d}: mean {sum(d) / len(d) / 1e3:8.1f} us  max {max(d) / 1e3:8.1f} us  wall per launch {span / 1e3:8.1f} us")
