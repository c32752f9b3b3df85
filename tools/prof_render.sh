cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_render
rm -rf $OUT; mkdir -p $OUT
CMD="python bench.py --no-cpu-baseline --steps 20 --warmup 3 --robot xarm7_pick --cameras side_cam --resolution 256x256"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $CMD > $OUT/log1 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU -d $OUT/pmc1 -o pmc1 -- $CMD > $OUT/log2 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU -d $OUT/pmc2 -o pmc2 -- $CMD > $OUT/log3 2>&1
python profiles/summarize.py $OUT 2>&1 | grep -v "^$" | head -80
