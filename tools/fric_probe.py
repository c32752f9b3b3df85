"""dev tool (GPU): the dry-friction FR3's convergence-mode rollout against the oracle, environment by environment
(tests/test_gpu_parity.py::test_joint_friction_on_the_other_archetypes).    python tools/fric_probe.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("robot-control-stack_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import parity_util as PU
n, steps = 32, 2
venv = PU.make_vec_env(n, False, gripper=True, robot="fr3_fric")
oenvs = PU.make_oracle_envs(n, False, gripper=True, robot="fr3_fric")
joints, grip = PU.synthetic_actions(n, steps, 23, dof=7)
venv.reset()
for oe in oenvs:
    oe.reset()
for t in range(steps):
    obs, _, _, _, info = venv.step({"joints": joints[t], "gripper": grip[t]})
    for e, oe in enumerate(oenvs):
        oe.step({"joints": joints[t, e], "gripper": grip[t, e]})
    q = venv.sim.qpos
    now, ever = venv.sim.contact_escalated()
    for e, oe in enumerate(oenvs):
        err = np.abs(q[e][:9] - oe.sim.qpos[:9])
        nc = oe.sim.s.ncon if hasattr(oe.sim.s, "ncon") else -1
        print(f"step {t} env {e}: arm err {err[:7].max():.2e} finger err {err[7:].max():.2e} grip cmd {grip[t, e]:.2f} escalated now {int(now[e])} ever {int(ever[e])} substeps gpu {int(info['substeps'][e])} oracle {int(oe.sim.s.convergence_steps)} oracle ncon {nc} fingers {oe.sim.qpos[7]:.5f} {oe.sim.qpos[8]:.5f}")
