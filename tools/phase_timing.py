"""dev tool: per-wave, per-phase cycle counts of the 4-wave kernel (library built with -DRCSH_PHASE_TIMING)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "robot-control-stack_amd"), os.path.join(ROOT, "tests")]
os.environ["RCSH_FOUR_WAVE"] = "1"
import rcs_amd._lib as lib
lib.LIB_PATH = os.path.join(ROOT, "robot-control-stack_amd", "rcs_amd", "librcs_hip_timing.so")
import numpy as np
from parity_util import make_vec_env, synthetic_actions
n, T = 4096, 20
env = make_vec_env(n, True)
j, g = synthetic_actions(64, T, 0)
j = np.tile(j, (1, n // 64, 1)); g = np.tile(g, (1, n // 64))
env.reset()
for t in range(T): env.step({"joints": j[t], "gripper": g[t]})
out = (C.c_ulonglong * 32)()
env._L.rcsh_debug_phase_cycles(out)
a = np.array(out[:], dtype=np.float64).reshape(4, 8) / (T * 17 + 1)
names = ["A", "B", "C", "D1", "D2(solve)", "D3(integ)", "-", "barrier wait"]
print("cycles per substep (block 0, lane 0):")
print("wave  " + "  ".join(f"{x:>12}" for x in names))
for w in range(4): print(f"W{w}    " + "  ".join(f"{x:12.0f}" for x in a[w]), "  total", round(a[w].sum()))
