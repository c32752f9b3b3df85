#!/bin/bash
# dev tool: per-launch duration of the env-step kernel over a bench run (rocprofv3 kernel trace), to tell a uniform slowdown from a tail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/lt && mkdir -p gpurun_out/lt
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lt -o lt -- python bench.py --steps 300 --warmup 30 --no-cpu-baseline "$@" > gpurun_out/lt/bench.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/lt/**/lt_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_run_team" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print(len(d), "launches; us:", " ".join(f"{sum(d[i:i+30])/len(d[i:i+30]):.1f}" for i in range(0, len(d), 30)))
print("max per block of 30:", " ".join(f"{max(d[i:i+30]):.0f}" for i in range(0, len(d), 30)))
PY
tail -1 gpurun_out/lt/bench.log | cut -c1-160
