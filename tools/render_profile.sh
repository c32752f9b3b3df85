#!/bin/bash
# dev tool (GPU box, repo root): rocprofv3 kernel stats + one PMC pass of tools/render_bench.py -> gpurun_out/prof_render/
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_render
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python tools/render_bench.py > "$OUT/stats.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU -d "$OUT/pmc1" -o pmc1 -- python tools/render_bench.py > "$OUT/pmc1.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d "$OUT/pmc2" -o pmc2 -- python tools/render_bench.py > "$OUT/pmc2.log" 2>&1
find "$OUT" -name "*kernel_stats.csv" | head -1 | xargs head -12
