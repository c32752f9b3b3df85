"""Capture MuJoCo reference trajectories as fixtures (run on ANY machine where `pip install mujoco==3.2.6` works).

    python tools/capture_mujoco_fixture.py            # writes tests/golden/mujoco/*.npz

Neither the build container nor the GPU box has MuJoCo (reference pin: pyproject.toml:23), so the physics oracle
(oracle/rcs_physics.c, rcs_object.c, rcs_contact.c) is "parity unpinned" against it.  This script is the other half of the
pin: it loads the repository's physics-only scenes in real MuJoCo, replays seeded controls and stores

  * the controls and the qpos / qvel trajectories (sampled every `k` substeps),
  * the model constants the restatement derives itself (dof_invweight0, body_invweight0, body_mass, body_inertia,
    stat.meaninertia, actuator gains, joint ranges),

as data.  tests/test_mujoco_optional.py::test_oracle_matches_mujoco_fixture replays every fixture it finds in the oracle and
applies the north-star tolerance (1e-5 on joint positions / velocities); with no fixture committed it skips.  Mesh geoms
are stripped before loading (their hull vertex sets live in an .npz MuJoCo cannot read; they are massless collision
geoms): joint-space dynamics are unaffected, the contact scenarios below only involve primitive geoms (pads, cube, floor).
"""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = os.path.join(ROOT, "robot-control-stack_amd", "rcs_amd", "scenes")
OUT = os.path.join(ROOT, "tests", "golden", "mujoco")

FR3_HOME = [0.0, -np.pi / 4, 0.0, -3 * np.pi / 4, 0.0, np.pi / 2, np.pi / 4]


def load(scene: str):
    import mujoco

    path = os.path.join(SCENES, scene, "scene.xml")
    xml = open(path).read()

    def inline(m):
        inc = open(os.path.join(os.path.dirname(path), m.group(1))).read()
        return re.search(r"<mujoco[^>]*>(.*)</mujoco>", inc, re.S).group(1)

    xml = re.sub(r'<include file="([^"]*)"\s*/>', inline, xml)
    xml = re.sub(r"<geom[^>]*\bmesh=\"[^\"]*\"[^>]*/>", "", xml)
    return mujoco.MjModel.from_xml_string(xml)


def constants(m):
    return dict(dof_invweight0=np.array(m.dof_invweight0), body_invweight0=np.array(m.body_invweight0), body_mass=np.array(m.body_mass),
                body_inertia=np.array(m.body_inertia), meaninertia=np.array(m.stat.meaninertia), jnt_range=np.array(m.jnt_range),
                actuator_gainprm=np.array(m.actuator_gainprm[:, :3]), actuator_biasprm=np.array(m.actuator_biasprm[:, :3]),
                timestep=np.array(m.opt.timestep))


def joint_rollout(scene, joints, actuators, home, steps=20, k=17, seed=0):
    import mujoco

    m = load(scene)
    d = mujoco.MjData(m)
    jadr = [m.joint(n).qposadr[0] for n in joints]
    aid = [m.actuator(n).id for n in actuators]
    q = np.array(home, dtype=np.float64)
    d.qpos[jadr] = q
    d.ctrl[aid] = q
    rng = np.random.default_rng(seed)
    ctrl, qpos, qvel = [], [], []
    for _ in range(steps):
        q = q + rng.uniform(-0.0873, 0.0873, size=q.shape)
        d.ctrl[aid] = q
        for _ in range(k):
            mujoco.mj_step(m, d)
        ctrl.append(q.copy()); qpos.append(np.array(d.qpos)); qvel.append(np.array(d.qvel))
    return dict(kind="joint_rollout", scene=scene, joints=np.array(joints), actuators=np.array(actuators), home=np.array(home), k=np.array(k),
                ctrl=np.array(ctrl), qpos=np.array(qpos), qvel=np.array(qvel), **constants(m))


def pinch(scene="fr3_simple_pick_up", k=25):
    """Joint-space pinch of the cube: the arm is driven through joint targets (computed once by this repository's CLIK and
    stored here as data), the fingers close on the cube, the arm lifts.  Contacts: floor-cube, pads-cube."""
    import mujoco

    m = load(scene)
    d = mujoco.MjData(m)
    joints = [f"fr3_joint{i}_0" for i in range(1, 8)]
    jadr = [m.joint(n).qposadr[0] for n in joints]
    aid = [m.actuator(n).id for n in joints]
    grip = m.actuator("actuator8_0").id
    d.qpos[jadr] = FR3_HOME
    d.ctrl[aid] = FR3_HOME
    targets = np.load(os.path.join(OUT, "pinch_targets.npy"))  # [stage][7] joint targets + [7] = gripper ctrl, [8] = substeps
    ctrl, qpos, qvel, ncon = [], [], [], []
    for row in targets:
        d.ctrl[aid] = row[:7]
        d.ctrl[grip] = row[7]
        for _ in range(int(row[8]) // k):
            for _ in range(k):
                mujoco.mj_step(m, d)
            ctrl.append(np.array(d.ctrl)); qpos.append(np.array(d.qpos)); qvel.append(np.array(d.qvel)); ncon.append(d.ncon)
    return dict(kind="pinch", scene=scene, k=np.array(k), ctrl=np.array(ctrl), qpos=np.array(qpos), qvel=np.array(qvel), ncon=np.array(ncon), **constants(m))


def main():
    import mujoco

    os.makedirs(OUT, exist_ok=True)
    print("mujoco", mujoco.__version__)
    fr3 = [f"fr3_joint{i}_0" for i in range(1, 8)]
    np.savez_compressed(os.path.join(OUT, "fr3_empty_world_joint_rollout.npz"), version=mujoco.__version__, **joint_rollout("fr3_empty_world", fr3, fr3, FR3_HOME))
    xj = [f"joint{i}" for i in range(1, 8)]
    xa = [f"act{i}" for i in range(1, 8)]
    np.savez_compressed(os.path.join(OUT, "xarm7_empty_world_joint_rollout.npz"), version=mujoco.__version__,
                        **joint_rollout("xarm7_empty_world", xj, xa, [0, 0, 0, 0, 0, 0, 0]))
    if os.path.exists(os.path.join(OUT, "pinch_targets.npy")):
        np.savez_compressed(os.path.join(OUT, "fr3_simple_pick_up_pinch.npz"), version=mujoco.__version__, **pinch())
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    sys.exit(main())
