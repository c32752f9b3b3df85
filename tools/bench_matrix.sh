#!/bin/bash
# The benchmark matrix of BASELINE.md section 3 (one JSON line per workload) -> gpurun_out/bench_matrix.jsonl; run on the GPU box.
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/bench_matrix.jsonl
: > $OUT
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 >> $OUT; }
run --steps 20 --warmup 5                                     # headline at the driver's length
run --steps 300 --warmup 30                                   # headline: 4096 x fr3_empty_world JOINTS async
run --steps 1000 --warmup 50                                  # BASELINE.md's rollout length
run --steps 300 --warmup 30 --contacts flag --contact-check-every 16   # the round-4 configuration
run --steps 300 --warmup 5 --mode convergence                 # ... 300 steps without a reset
run --steps 30 --warmup 5 --mode convergence                  # reference default: step_until_convergence
run --steps 200 --warmup 20 --control cartesian               # BASELINE configs[2]: CARTESIAN_TRPY -> CLIK + gripper
run --steps 200 --warmup 20 --robot xarm7
run --steps 200 --warmup 20 --robot arm6
run --steps 200 --warmup 20 --robot xarm7_box
run --steps 200 --warmup 20 --task pick_up
run --steps 100 --warmup 10 --task pick_up --cameras wrist_0 --resolution 64x64
run --steps 100 --warmup 10 --envs 65536
run --steps 200 --warmup 20 --robot ur5e
run --steps 200 --warmup 20 --robot so101
run --steps 200 --warmup 20 --robot mixed --envs 4096
run --steps 200 --warmup 20 --robot xarm7_pick
run --steps 50 --warmup 5 --robot xarm7_pick --cameras side_cam --resolution 256x256
python - <<'PY'
import json
for l in open("gpurun_out/bench_matrix.jsonl"):
    d = json.loads(l)
    print(f"{d['value']/1e6:8.2f} M env-steps/s  {d['ms_per_step']:.3f} ms  {d['metric']}  [{d['config']['mode']}]")
PY
