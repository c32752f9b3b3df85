"""Opportunistic comparison against real MuJoCo: runs only where `import mujoco` works.

Neither this container nor the GPU box has MuJoCo (pinned 3.2.6 by the reference, pyproject.toml:23), so here the
module skips; on a machine that has it, it steps the repository's physics-only scenes in MuJoCo and in the CPU
oracle on the same controls and applies the north-star tolerance (1e-5 on joint positions / velocities).  This is the
test that would turn "parity unpinned" into a pin.  MuJoCo is a third-party wheel, not reference code.
"""

import os
import re

import numpy as np
import pytest

mujoco = pytest.importorskip("mujoco")

import rcs_oracle as O  # noqa: E402
from parity_util import SCENE, XARM7_SCENE  # noqa: E402
from rcs_amd.mjcf import compile_mjcf  # noqa: E402
from rcs_env_oracle import FR3, XARM7  # noqa: E402


def _mujoco_model(path: str):
    """The scene without its mesh geoms (mass 0, collision only; their hull vertices live in an .npz MuJoCo cannot read)."""
    xml = open(path).read()
    xml = re.sub(r"<geom[^>]*\bmesh=\"[^\"]*\"[^>]*/>", "", xml)
    return mujoco.MjModel.from_xml_string(xml)


@pytest.mark.parametrize("scene,robot", [(SCENE, FR3), (XARM7_SCENE, XARM7)])
def test_oracle_matches_mujoco_on_joint_rollout(scene, robot):
    mm = _mujoco_model(scene)
    md = mujoco.MjData(mm)
    cm = compile_mjcf(scene)
    s = O.Sim(cm, robot["joints"], robot["actuators"], robot["site"], robot["base"], robot["q_home"], None,
              gripper_joint=robot["gripper_joint"], gripper_actuator=robot["gripper_actuator"],
              arm_collision_geoms=robot.get("arm_collision_geoms"))
    rng = np.random.default_rng(0)
    q = np.array(robot["q_home"], dtype=np.float64)
    jadr = [mm.joint(n).qposadr[0] for n in robot["joints"]]
    aid = [mm.actuator(n).id for n in robot["actuators"]]
    for i, a in zip(jadr, q):
        md.qpos[i] = a
        s.s.d.qpos[i] = a
    for u, a in zip(aid, q):
        md.ctrl[u] = a
    s.set_joint_position(q)
    for _ in range(20):
        q = q + rng.uniform(-0.0873, 0.0873, size=q.shape)
        for u, a in zip(aid, q):
            md.ctrl[u] = a
        s.set_joint_position(q)
        for _ in range(17):
            mujoco.mj_step(mm, md)
        s.step(17)
        assert np.abs(np.asarray(md.qpos)[jadr] - np.asarray(s.qpos)[: len(jadr)]).max() < 1e-5
        assert np.abs(np.asarray(md.qvel)[jadr] - np.asarray(s.qvel)[: len(jadr)]).max() < 1e-5
    assert os.path.exists(scene)


def test_oracle_free_box_matches_mujoco():
    """The cube of fr3_simple_pick_up against MuJoCo: contact count, the regularised friction of the elliptic cone, the
    pop-out after RandomCubePos' placement, a kicked slide and the rest pose.  With its mesh geoms removed the robot
    cannot touch the cube in MuJoCo either, which is the configuration the oracle restates (floor contacts only)."""
    from parity_util import PICKUP_SCENE

    xml = open(PICKUP_SCENE).read()
    inc = open(os.path.join(os.path.dirname(PICKUP_SCENE), "..", "fr3_empty_world", "scene.xml")).read()
    inc = re.sub(r"<geom[^>]*\bmesh=\"[^\"]*\"[^>]*/>", "", inc)
    # the pads are box geoms and would collide with the cube in MuJoCo: take them out of the collision set as well
    inc = re.sub(r"(<default class=\"pad\d\"><geom )", r"\1contype=\"0\" conaffinity=\"0\" ", inc)
    body = re.search(r"<mujoco[^>]*>(.*)</mujoco>", inc, re.S).group(1)
    xml = xml.replace('<include file="../fr3_empty_world/scene.xml"/>', body)
    mm = mujoco.MjModel.from_xml_string(xml)
    md = mujoco.MjData(mm)
    cm = compile_mjcf(PICKUP_SCENE)
    s = O.Sim(cm, FR3["joints"], FR3["actuators"], FR3["site"], FR3["base"], FR3["q_home"], None,
              gripper_joint=FR3["gripper_joint"], gripper_actuator=FR3["gripper_actuator"])
    assert abs(mm.stat.meaninertia - s.model.box.meaninertia) < 1e-9
    badr = mm.joint("box_joint").qposadr[0]
    vadr = mm.joint("box_joint").dofadr[0]
    jadr = [mm.joint(n).qposadr[0] for n in FR3["joints"]]
    for i, a in zip(jadr, FR3["q_home"]):
        md.qpos[i] = a
        s.s.d.qpos[i] = a
    for n, a in zip(FR3["actuators"], FR3["q_home"]):
        md.ctrl[mm.actuator(n).id] = a
    s.set_joint_position(FR3["q_home"])
    place = [0.5, -0.03, 0.0288 / 2, 0.3, 0, 0, 1]
    md.qpos[badr:badr + 7] = place
    s.box_qpos = place
    for k in range(600):
        if k == 300:  # kick the resting cube sideways
            md.qvel[vadr:vadr + 3] = [0.4, 0.2, 0]
            v = s.box_qvel
            v[:3] = [0.4, 0.2, 0]
            s.box_qvel = v
        mujoco.mj_step(mm, md)
        s.step(1)
        assert md.ncon == s.s.d.box.ncon, k
        assert np.abs(np.asarray(md.qpos)[badr:badr + 7] - s.box_qpos).max() < 1e-5, k
        assert np.abs(np.asarray(md.qvel)[vadr:vadr + 6] - s.box_qvel).max() < 1e-4, k
