"""dev tool: the 1000-step headline rollout with contacts resolved environment by environment, per-environment report.
    python tools/headline_resolved_probe.py [n_envs] [n_steps] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("robot-control-stack_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import parity_util as PU  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
venv = PU.make_vec_env(n, True)
oenvs = PU.make_oracle_envs(n, True)
cm = oenvs[0].sim.cm
joints, grip = PU.synthetic_actions(n, steps, seed)
venv.reset()
for oe in oenvs:
    oe.reset()
first = np.full(n, -1)
bad = np.full(n, -1)
err = np.zeros(n)
kinds = {}
log = {}
for t in range(steps):
    venv.step({"joints": joints[t], "gripper": grip[t]})
    q = venv.sim.qpos
    now, ever = venv.sim.contact_escalated()
    for e, oe in enumerate(oenvs):
        oe.sim.s.d.pen_seen = 0.0
        oe.step({"joints": joints[t, e], "gripper": grip[t, e]})
        d = oe.sim.s.d
        if d.pen_seen > 0 and first[e] < 0:
            first[e] = t
        if d.ncon:
            kinds.setdefault(e, set()).update((cm.geom_names[d.contact[c].geom[0]] if d.contact[c].geom[0] < cm.ngeom else "box", cm.geom_names[d.contact[c].geom[1]]) for c in range(d.ncon))
        dq = float(np.abs(q[e] - oe.sim.qpos[: q.shape[1]]).max())
        err[e] = max(err[e], dq)
        if dq > 1e-9 and bad[e] < 0:
            bad[e] = t
            log[e] = (t, dq, d.ncon, d.solver_niter, d.noslip_niter, bool(now[e]), float(d.pen_seen))
for e in range(n):
    if first[e] >= 0 or bad[e] >= 0:
        print(e, "first contact step", first[e], "first bad step", bad[e], "max err %.2e" % err[e], log.get(e), sorted(kinds.get(e, []))[:4])
print("overall", err.max())
