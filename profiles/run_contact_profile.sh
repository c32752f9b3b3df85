#!/bin/bash
# Usage (on the GPU box, from the repo root): bash profiles/run_contact_profile.sh <tag>
# rocprofv3 kernel stats + two PMC passes of the contact-resolving kernel while EVERY environment pinches its cube
# (tools/grasp_bench.py 4096: fr3_simple_pick_up, k_run_team<Topo<7,true>, false, BOX, CON>) -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r4_contact}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
CMD="python tools/grasp_bench.py 4096"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- $CMD > "$OUT/bench_stats.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU -d "$OUT/pmc1" -o pmc1 -- $CMD > "$OUT/bench_pmc1.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT -d "$OUT/pmc2" -o pmc2 -- $CMD > "$OUT/bench_pmc2.log" 2>&1
python profiles/summarize.py "$OUT" > "$OUT/summary.txt" 2>&1
grep -E "substeps" "$OUT/bench_stats.log" >> "$OUT/summary.txt"
cat "$OUT/summary.txt"
