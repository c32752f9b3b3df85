#!/bin/bash
# dev tool: env-steps/s of each kernel variant over batch sizes
cd "$GRAFT_REPO_ROOT"
for n in 64 1024 4096 8192 16384 32768 65536 131072 262144 524288; do
  for k in 0 1; do
    RCSH_KERNEL=$([ $k = 1 ] && echo team || echo lane) python bench.py --no-cpu-baseline --steps 30 --warmup 5 --envs $n | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('envs %7d team=$k  %8.2f M env-steps/s  %.3f ms' % ($n, d['value']/1e6, d['ms_per_step']))"
  done
done
