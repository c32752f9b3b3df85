"""dev tool: how often the contact phase wakes during the headline rollout (library built with tools/build_timing.sh)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "robot-control-stack_amd")]
import rcs_amd._lib as lib
lib.LIB_PATH = os.path.join(ROOT, "robot-control-stack_amd", "rcs_amd", "librcs_hip_timing.so")
import numpy as np
from rcs_amd.envs import make_vec_env
n = 4096
env = make_vec_env(n, True)
env.reset()
rng = np.random.default_rng(0)
out = (C.c_ulonglong * 48)()
prev = np.zeros(48)
for t in range(331):
    env.step({"joints": rng.uniform(-0.0873, 0.0873, (n, 7)), "gripper": rng.uniform(0, 1, n).astype(np.float32)})
    if t % 30 == 0:
        env._L.rcsh_debug_team_cycles48(out); a = np.array(out[:], dtype=np.float64); d = a - prev; prev = a
        q = env.sim.qpos
        print(f"step {t:4d}: wave-substeps {d[41]:.0f}, with a woken contact phase {d[40]:.0f} ({100 * d[40] / max(d[41], 1):.2f} %), coupled phases (block 0) {a[34]:.0f}; "
              f"collision flags {int(env.robot.get_state().collision.sum())}")
