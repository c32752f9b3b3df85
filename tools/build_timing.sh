#!/bin/bash
# dev tool: builds the instrumented library tools/team_timing.py loads (phase cycle counters compiled in)
cd "$(dirname "$0")/../robot-control-stack_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
  -Wno-unused-value -mllvm -amdgpu-sched-strategy=iterative-ilp -DRCSH_PHASE_TIMING rcs_hip.hip model.cpp -o ../rcs_amd/librcs_hip_timing.so
