"""Replay of MuJoCo fixtures (tests/golden/mujoco/*.npz, captured by tools/capture_mujoco_fixture.py on a machine that has
mujoco==3.2.6) in the CPU oracle, at the north-star tolerance: 1e-5 on joint positions / velocities, model constants
1e-9.  This is the pin the physics oracle is waiting for: MuJoCo cannot be installed in the build
container or on the GPU box, so the comparison has to travel as data.  With no fixture committed every case skips -- and the
oracle stays "parity unpinned".
"""

import glob
import os

import numpy as np
import pytest

import rcs_oracle as O
from parity_util import ROOT
from rcs_amd.mjcf import compile_mjcf

FIX = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "mujoco", "*.npz")))
SCENES = os.path.join(ROOT, "robot-control-stack_amd", "rcs_amd", "scenes")


def _scene(name: str) -> str:
    return os.path.join(SCENES, name, "scene.xml")


@pytest.mark.skipif(not FIX, reason="no MuJoCo fixture committed (run tools/capture_mujoco_fixture.py where mujoco==3.2.6 is installable)")
@pytest.mark.parametrize("path", FIX or [None])
def test_oracle_matches_mujoco_fixture(path):
    f = np.load(path, allow_pickle=False)
    cm = compile_mjcf(_scene(str(f["scene"])))
    kind = str(f["kind"])
    if kind == "joint_rollout":
        joints, acts = [str(x) for x in f["joints"]], [str(x) for x in f["actuators"]]
        grip = ("finger_joint1_0", "actuator8_0") if "finger_joint1_0" in cm.jnt_names else (None, None)
        s = O.Sim(cm, joints, acts, cm.site_names[0], cm.body_names[1], list(f["home"]), None, gripper_joint=grip[0], gripper_actuator=grip[1],
                  arm_collision_geoms=[])
        m = s.model
        # constants the restatement derives itself
        nv = cm.nv
        assert np.abs(np.array(m.dof_invweight0[:nv]) - f["dof_invweight0"][:nv]).max() < 1e-9
        assert abs(float(m.timestep) - float(f["timestep"])) == 0
        k = int(f["k"])
        for i, a in enumerate(f["home"]):
            s.s.d.qpos[i] = a
        s.set_joint_position(f["home"])
        for t in range(len(f["ctrl"])):
            s.set_joint_position(f["ctrl"][t])
            s.step(k)
            assert np.abs(np.asarray(s.qpos)[:nv] - f["qpos"][t][:nv]).max() < 1e-5, t
            assert np.abs(np.asarray(s.qvel)[:nv] - f["qvel"][t][:nv]).max() < 1e-5, t
    elif kind == "pinch":
        from rcs_env_oracle import FR3_Q_HOME

        arm = [f"fr3_joint{i}_0" for i in range(1, 8)]
        s = O.Sim(cm, arm, arm, "attachment_site_0", "base_0", FR3_Q_HOME, O.franka_hand_tcp_offset(), "finger_joint1_0", "actuator8_0")
        assert abs(s.model.box.meaninertia - float(f["meaninertia"])) < 1e-9
        # body_invweight0 of the finger bodies and the cube (mjModel.body_invweight0[:, 0])
        for i, a in enumerate(FR3_Q_HOME):
            s.s.d.qpos[i] = a
        s.set_joint_position(FR3_Q_HOME)
        k = int(f["k"])
        for t in range(len(f["ctrl"])):
            for u in range(8):
                s.s.d.ctrl[u] = float(f["ctrl"][t][u])
            s.step(k)
            assert int(s.s.d.ncon) == int(f["ncon"][t]), t
            assert np.abs(np.asarray(s.qpos)[:9] - f["qpos"][t][:9]).max() < 1e-5, t
            assert np.abs(s.box_qpos - f["qpos"][t][9:16]).max() < 1e-5, t
    else:
        pytest.fail(f"unknown fixture kind {kind}")
