#!/bin/bash
# dev tool: every bench.py configuration once, short (does each still print its line?)  -> gpurun_out/bench_smoke.jsonl
mkdir -p gpurun_out; out=gpurun_out/bench_smoke.jsonl; : > $out
run() { echo "== $*"; python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | tee -a $out | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print('   ', round(d['value'] / 1e6, 2), 'M env-steps/s', d['ms_per_step'], 'ms/step; kernel', d['roofline']['kernel_ms_avg'], 'finite', d['config']['obs_finite'], '| exchange:', d['config']['exchange'][:80])
except Exception as e: print('   FAILED', e)"; }
run --steps 20 --warmup 5
run --steps 300 --warmup 30
run --steps 100 --warmup 10 --control cartesian
run --steps 100 --warmup 10 --task pick_up
run --steps 60 --warmup 10 --task pick_up --cameras wrist_0 --resolution 64x64
run --steps 100 --warmup 10 --robot xarm7
run --steps 100 --warmup 10 --robot mixed --envs 4096
run --steps 100 --warmup 10 --robot mixed --envs 16384
run --steps 60 --warmup 10 --mode convergence
run --steps 40 --warmup 5 --gpus 2 --dist-backend host --envs 1024
run --steps 100 --warmup 10 --robot xarm7_pick
run --steps 40 --warmup 5 --robot xarm7_pick --cameras side_cam --resolution 256x256
run --steps 40 --warmup 5 --gpus 2 --dist-backend sdma --envs 1024
run --steps 40 --warmup 5 --gpus 2 --envs 1024
