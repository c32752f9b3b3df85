#!/bin/bash
# The occupancy experiment's evidence: per-kernel durations and SQ counters for builds with one and two
# resident wavefronts per SIMD.  usage (GPU box, repo root): tools/occ_profile.sh  -> gpurun_out/r3_occ2/<case>/summary.txt
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
run() {  # tag, RCSH_OCC2, bench args
  local tag=$1 occ=$2; shift 2
  local out=gpurun_out/r3_occ2/$tag; mkdir -p $out
  local cmd="python bench.py --steps 60 --warmup 10 --no-cpu-baseline $*"
  RCSH_OCC2=$occ rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- $cmd > $out/bench.log 2>&1
  RCSH_OCC2=$occ rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $out/pmc1 -o pmc1 -- $cmd > $out/bench_pmc1.log 2>&1
  RCSH_OCC2=$occ rocprofv3 --kernel-trace --output-format csv --pmc SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES -d $out/pmc2 -o pmc2 -- $cmd > $out/bench_pmc2.log 2>&1
  python profiles/summarize.py $out > $out/summary.txt 2>&1
  tail -1 $out/bench.log >> $out/summary.txt
  rm -rf $out/stats/*/*_agent_info.csv
}
run ur5e_4096 0 --robot ur5e --envs 4096
run ur5e_8192 0 --robot ur5e --envs 8192
run fr3_8192_occ1 0 --robot fr3 --envs 8192
run fr3_8192_occ2 1 --robot fr3 --envs 8192
run so101_8192_occ1 0 --robot so101 --envs 8192
run so101_8192_occ2 1 --robot so101 --envs 8192
for d in gpurun_out/r3_occ2/*/; do echo "=== $d"; grep -A12 "k_run_team" $d/summary.txt | head -40; done
