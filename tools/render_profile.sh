#!/bin/bash
# dev tool (GPU box): per-kernel times of the ray-casting path -> gpurun_out/prof_render/
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_render
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python tools/render_bench.py ${1:-4096} > "$OUT/render_bench.log" 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_render/stats/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print(r["Name"][:90], r["Calls"], "avg", r["AverageNs"], "min", r["MinNs"], "max", r["MaxNs"])
PY
tail -6 "$OUT/render_bench.log"
