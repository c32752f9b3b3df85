#!/bin/bash
# dev tool: bench alternative builds of the library (copies each over librcs_hip.so in the scratch copy)
cd "$GRAFT_REPO_ROOT"
D=robot-control-stack_amd/rcs_amd
cp $D/librcs_hip.so /tmp/orig.so
for v in "$@"; do
  cp $D/librcs_hip_$v.so $D/librcs_hip.so
  for fw in 0 1; do
    RCSH_FOUR_WAVE=$fw python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v four_wave=$fw', round(d['value']/1e6,2), 'M env-steps/s', round(d['ms_per_step'],4), 'ms')"
  done
done
cp /tmp/orig.so $D/librcs_hip.so
