"""dev tool: as self_timing.py, but in the bench's convergence-mode loop (env.step with random relative actions): after a few
steps the four environments of a wavefront no longer share their callback phases."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "robot-control-stack_amd"), os.path.join(ROOT, "tests")]
import rcs_amd._lib as lib
lib.LIB_PATH = os.path.join(ROOT, "robot-control-stack_amd", "rcs_amd", "librcs_hip_timing.so")
import numpy as np
from rcs_amd.envs import make_vec_env, MAX_JOINT_MOV
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = make_vec_env(n, async_control=False, gripper=True, relative=True)
env.reset()
rng = np.random.default_rng(0)
out = (C.c_ulonglong * 48)()
for step in range(12):
    env.sim._L.rcsh_debug_team_cycles48(out); base = np.array(out[:], dtype=np.float64)
    obs, _, _, _, info = env.step({"joints": rng.uniform(-MAX_JOINT_MOV, MAX_JOINT_MOV, (n, 7)), "gripper": rng.uniform(0, 1, n).astype(np.float32)})
    env.sim._L.rcsh_debug_team_cycles48(out); a = np.array(out[:], dtype=np.float64) - base
    tot = sum(a[i] for i in range(48) if i not in (29, 33, 34, 35, 36, 39, 40, 41, 46, 47))
    print(f"step {step}: substeps wg0 {info['substeps'][:4]}, calls {a[47]:.0f}, candidates {a[46]:.0f}, pair test cycles {a[42]+a[43]+a[44]+a[37]+a[38]+a[45]:.0f} of {tot:.0f}")
