"""dev tool (GPU, timing build): step_until_convergence over a long rollout without resets -- what the self-collision test costs as the arms wander."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "robot-control-stack_amd"), os.path.join(ROOT, "tests")]
import rcs_amd._lib as lib
lib.LIB_PATH = os.path.join(ROOT, "robot-control-stack_amd", "rcs_amd", "librcs_hip_timing.so")
import numpy as np, time
from parity_util import make_vec_env, synthetic_actions
n, T = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 300
env = make_vec_env(n, False)
j, g = synthetic_actions(n, T, 0, dof=env.dof)
env.reset()
out = (C.c_ulonglong * 64)()
env._L.rcsh_debug_team_cycles64(out)
prev = np.array(out[:], dtype=np.float64)
t0 = time.perf_counter()
for t in range(T):
    info = env.step({"joints": j[t], "gripper": g[t]})[4]
    if t % 50 == 49:
        env.sim.synchronize() if hasattr(env.sim, "synchronize") else None
        dt = (time.perf_counter() - t0) / 50; t0 = time.perf_counter()
        env._L.rcsh_debug_team_cycles64(out)
        a = np.array(out[:], dtype=np.float64); d = a - prev; prev = a
        coll = float(np.asarray(info["collision"]).mean())
        print(f"steps {t-49}..{t}: {dt*1e3:.2f} ms/step, substeps mean {np.asarray(info['substeps']).mean():.0f}, in collision {coll:.3f}; wave-0 cycles per env-step: slack+spheres {d[42]/50:.0f} (mark40 {d[40]/50:.0f}) boxes {d[43]/50:.0f} stage {d[44]/50:.0f} refine {d[45]/50:.0f} m37 {d[37]/50:.0f} m38 {d[38]/50:.0f} m39 {d[39]/50:.0f}; counts per env-step: calls {d[47]/50:.1f} narrow pairs {d[46]/50:.1f} sphere-survivor calls {d[41]/50:.1f}")
