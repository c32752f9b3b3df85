"""dev tool (CPU): the most contacts the uncapped oracle sees, per environment, on the headline rollout.
    python tools/oracle_ncon_probe.py seed steps env [env ...]     (environment e of seed s draws from rng(s + e))"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("robot-control-stack_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import parity_util as PU  # noqa: E402

seed, steps = int(sys.argv[1]), int(sys.argv[2])
envs = [int(a) for a in sys.argv[3:]]
for e in envs:
    oe = PU.make_oracle_envs(1, True)[0]
    joints, grip = PU.synthetic_actions(1, steps, seed + e)
    oe.reset()
    mx, when, hist = 0, -1, []
    for t in range(steps):
        oe.step({"joints": joints[t, 0], "gripper": grip[t, 0]})
        nc = oe.sim.s.d.ncon
        if nc > mx:
            mx, when = nc, t
        hist.append(nc)
    h = np.array(hist)
    print("env", e, "max ncon", mx, "at step", when, "; steps with > 48:", int((h > 48).sum()), "> 64:", int((h > 64).sum()), "> 96:", int((h > 96).sum()), flush=True)
