"""dev tool (CPU): how far two runs of the ORACLE part when one of them is nudged by 1e-13 rad -- the conditioning of the rollout itself.
    python tools/oracle_sensitivity.py seed env nudge_step [steps] [eps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("robot-control-stack_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import parity_util as PU  # noqa: E402

seed, e, nudge = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
eps = float(sys.argv[5]) if len(sys.argv) > 5 else 1e-13
a, b = PU.make_oracle_envs(2, True)
joints, grip = PU.synthetic_actions(1, steps, seed + e)
a.reset(); b.reset()
worst = 0.0
for t in range(steps):
    if t == nudge:
        b.sim.s.d.qpos[3] += eps
    a.step({"joints": joints[t, 0], "gripper": grip[t, 0]})
    b.step({"joints": joints[t, 0], "gripper": grip[t, 0]})
    if t >= nudge:
        dq = float(np.abs(np.array(a.sim.qpos) - np.array(b.sim.qpos)).max())
        worst = max(worst, dq)
        if t < nudge + 40 or t % 20 == 0:
            print(t, "ncon", a.sim.s.d.ncon, b.sim.s.d.ncon, "niter", a.sim.s.d.solver_niter, a.sim.s.d.noslip_niter, "dq %.3e" % dq)
print("worst", worst)
