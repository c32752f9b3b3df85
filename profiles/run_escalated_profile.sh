#!/bin/bash
# Usage (on the GPU box, from the repo root): bash profiles/run_escalated_profile.sh <tag>
# rocprofv3 kernel stats + one PMC pass of the headline's 1000-step rollout: the lean launch next to the contact-resolving launch
# over the escalated environments (k_run_team<Topo<7,true>, false, false, CON>) -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r5_escalated}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
CMD="python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- $CMD > "$OUT/bench_stats.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU -d "$OUT/pmc1" -o pmc1 -- $CMD > "$OUT/bench_pmc1.log" 2>&1
python profiles/summarize.py "$OUT" > "$OUT/summary.txt" 2>&1
tail -1 "$OUT/bench_stats.log" | cut -c1-300 >> "$OUT/summary.txt"
python tools/esc_trace.py $(find "$OUT/stats" -name "*kernel_trace.csv" | head -1) 100 >> "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
