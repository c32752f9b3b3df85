#!/bin/bash
# dev tool: the occupancy experiment: team kernel as compiled (one wavefront per SIMD fits) against the
# <= 256-register build (two resident wavefronts per SIMD), over batch sizes and robots.  Output: gpurun_out/occ_sweep.jsonl
mkdir -p gpurun_out
out=gpurun_out/occ_sweep.jsonl; : > $out
for robot in fr3 xarm7 ur5e so101; do
 for envs in 4096 8192 16384 65536; do
  for occ in 0 1; do
    RCSH_OCC2=$occ python bench.py --robot $robot --envs $envs --steps 200 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'robot': '$robot', 'envs': $envs, 'occ2': $occ, 'Msteps_s': round(d['value'] / 1e6, 2), 'ms_per_step': d['ms_per_step'], 'kernel_ms': d.get('roofline', {}).get('kernel_ms_avg')}))" | tee -a $out
  done
 done
done
