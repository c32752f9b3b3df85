#!/bin/bash
# ur5e / so101 / mixed bench lines (one MI355X): gpurun_out/bench_new_robots.jsonl
mkdir -p gpurun_out
out=gpurun_out/bench_new_robots.jsonl
: > $out
for r in ur5e so101 mixed; do
  python bench.py --no-cpu-baseline --robot $r --steps 200 --warmup 20 2>>gpurun_out/bench_new_robots.err | tail -1 >> $out
done
python bench.py --no-cpu-baseline --robot mixed --envs 16384 --steps 100 --warmup 10 2>>gpurun_out/bench_new_robots.err | tail -1 >> $out
cat $out
