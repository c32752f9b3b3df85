#!/bin/bash
# Usage (on the GPU box, from the repo root): bash profiles/run_profile.sh <tag> [extra bench.py arguments, e.g. --control cartesian]
# Produces gpurun_out/prof_<tag>/{stats,pmc*} CSVs; copy the summaries you want judged into profiles/.
set -u
TAG=${1:-r1}
shift || true
EXTRA="$*"
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
CMD="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- $CMD > "$OUT/bench_stats.log" 2>&1
# PMC passes, each in its own run, kernel-trace only
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU -d "$OUT/pmc1" -o pmc1 -- $CMD > "$OUT/bench_pmc1.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM -d "$OUT/pmc2" -o pmc2 -- $CMD > "$OUT/bench_pmc2.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc3" -o pmc3 -- $CMD > "$OUT/bench_pmc3.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$OUT/pmc4" -o pmc4 -- $CMD > "$OUT/bench_pmc4.log" 2>&1
python profiles/summarize.py "$OUT" > "$OUT/summary.txt" 2>&1
python tools/host_rate.py >> "$OUT/summary.txt" 2>&1
tail -1 "$OUT/bench_stats.log" >> "$OUT/summary.txt"
cat "$OUT/summary.txt"
