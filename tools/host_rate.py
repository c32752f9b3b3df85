"""PCIe-inclusive rate: drive the fused env-step with host (numpy) buffers, 4096 envs, 17 substeps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "robot-control-stack_amd"), os.path.join(ROOT, "tests")]
import numpy as np
from parity_util import make_vec_env, synthetic_actions
n, T = 4096, 60
env = make_vec_env(n, True)
rng = np.random.default_rng(0)
j = rng.uniform(-0.087, 0.087, size=(T, n, 7)); g = rng.uniform(0, 1, size=(T, n)).astype(np.float32)
env.reset()
for t in range(10): env.step({"joints": j[t], "gripper": g[t]})
t0 = time.perf_counter()
for t in range(10, T): env.step({"joints": j[t], "gripper": g[t]})
dt = time.perf_counter() - t0
print(f"host-buffer env.step (numpy in/out, PCIe + sync each step): {n*(T-10)/dt:.3e} env-steps/s, {dt/(T-10)*1e3:.3f} ms/step")
