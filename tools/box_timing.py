"""dev tool: cycle counts of the free-box substep inside the team kernel (library built with tools/build_timing.sh)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "robot-control-stack_amd"), os.path.join(ROOT, "tests")]
import rcs_amd._lib as lib
lib.LIB_PATH = os.path.join(ROOT, "robot-control-stack_amd", "rcs_amd", "librcs_hip_timing.so")
import numpy as np
from rcs_amd.envs import FR3SimplePickUpSimEnvCreator
n, T = 4096, 20
env = FR3SimplePickUpSimEnvCreator()(n_envs=n)
rng = np.random.default_rng(0)
env.reset()
for t in range(5): env.step({"xyzrpy": rng.uniform(-0.02, 0.02, (n, 6)), "gripper": rng.uniform(0, 1, n)})
out = (C.c_ulonglong * 24)()
env._L.rcsh_debug_team_cycles(out)
base = np.array(out[:], dtype=np.float64)
for t in range(T): env.step({"xyzrpy": rng.uniform(-0.02, 0.02, (n, 6)), "gripper": rng.uniform(0, 1, n)})
env._L.rcsh_debug_team_cycles(out)
a = np.array(out[:], dtype=np.float64) - base
sub = T * 17
print(f"robot substep {a[:9].sum() / sub:.0f}  (+ leader post / loop sync incl. imp0 {a[9] / sub:.0f})")
for i, name in ((16, "box: state, contacts, rows, smooth"), (17, "box: warm start choice"), (18, "box: Newton"), (19, "box: noslip"), (20, "box: integrate")):
    print(f"  {name:36s} {a[i] / sub:9.0f}")
print(f"counts per substep (lane 0 of block 0): Newton iterations {a[21] / sub:.2f}, line-search evaluations {a[22] / sub:.2f}, QCQP iterations {a[23] / sub:.2f}")
