"""The robot dynamics against FIRST PRINCIPLES (tests/golden/*_dynamics.json, tools/derive_fr3_dynamics.py).

The golden vectors are a Lagrangian derivation in 40-digit arithmetic from the constants an independent reader took out of the
reference's own MJCF files (tests/golden/reference_models.json): mass matrix from differentiated forward kinematics,
Coriolis forces from the Christoffel symbols, gravity from the potential, gravity compensation, the MJCF actuator formulas and
clamps, one implicitfast step.  Nothing in them comes from the oracle, from the kernels, from rcs_amd/mjcf.py or from anybody's
memory of MuJoCo's pipeline -- they are what pins `orc_step1 / orc_step2` (CPU, here) and the HIP substep (`-m gpu`, through the
C-ABI) on the 9-dof FR3 + hand, the FR3 arm alone and the xArm7 chain.

Bars (written here): every intermediate of the oracle within 1e-10 of the golden value relative to the vector's largest entry
(measured 4e-13); the kernel's velocities / positions after one substep within 1e-10 (measured below 1e-13), i.e. its qacc
(v+ - v) / h within 1e-7 absolute of accelerations of order 1e2..1e3 rad/s^2.
"""
import ctypes as C
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, "..", "oracle"), os.path.join(HERE, "..", "robot-control-stack_amd"), HERE]

import parity_util as P  # noqa: E402

REL = 1e-10
CASES = {"fr3": lambda: P.SCENE, "fr3_arm": P.fr3_arm_only_scene, "xarm7": P.xarm7_frictionless_scene}


def golden(tag):
    return json.load(open(os.path.join(HERE, "golden", f"{tag}_dynamics.json")))


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


@pytest.mark.parametrize("tag", list(CASES))
def test_oracle_intermediates_match_first_principles(tag):
    import rcs_oracle as O
    from rcs_amd.mjcf import compile_mjcf

    g = golden(tag)
    nv = g["nv"]
    m = O.make_model(compile_mjcf(CASES[tag]()), False)
    assert m.njnt == nv
    L = O.lib()
    worst = {}
    for s in g["samples"]:
        d = O.OrcData()
        L.orc_reset_data(C.byref(m), C.byref(d))
        for i in range(nv):
            d.qpos[i], d.qvel[i] = s["qpos"][i], s["qvel"][i]
        for i, u in enumerate(s["ctrl"]):
            d.ctrl[i] = u
        L.orc_step1(C.byref(m), C.byref(d))
        L.orc_step2(C.byref(m), C.byref(d))
        qM = np.array(d.qM[:]).reshape(O.MAXV, O.MAXV)[:nv, :nv]
        got = {"qM": qM, "qvel_next": d.qvel[:nv], "qpos_next": d.qpos[:nv]}
        for k in ("qfrc_bias", "qfrc_gravcomp", "qfrc_passive", "qfrc_actuator", "qfrc_smooth", "qacc_smooth"):
            got[k] = getattr(d, k)[:nv]
        got["actuator_force"] = d.actuator_force[: len(s["actuator_force"])]
        if "efc_force" in s:  # FR3 + hand: the fingers' joint-equality row (MuJoCo's soft-constraint model, see the tool's header)
            assert d.nefc == 1
            got.update(efc_aref=[d.efc_aref[0]], efc_R=[1.0 / d.efc_D[0]], efc_force=[d.efc_force[0]], qfrc_constraint=d.qfrc_constraint[:nv])
            s = {**s, "efc_aref": [s["efc_aref"]], "efc_R": [s["efc_R"]], "efc_force": [s["efc_force"]]}
        else:
            assert d.nefc == 0
        for k, v in got.items():
            worst[k] = max(worst.get(k, 0.0), rel_err(v, s[k]))
    assert max(worst.values()) < REL, worst


def test_golden_vectors_cover_saturated_and_unsaturated_actuators():
    for tag in CASES:
        g = golden(tag)
        assert g["jacobian_self_check"] < 1e-20 and len(g["samples"]) >= 32
        sat = [max(abs(x) for x in s["qfrc_actuator"][:7]) for s in g["samples"]]
        lim = 87.0 if tag.startswith("fr3") else 50.0
        assert any(abs(x - lim) < 1e-12 for x in sat) and any(x < 0.6 * lim for x in sat)


def _hip_one_substep(tag, g):
    from rcs_amd import sim as S
    from rcs_amd.envs import default_sim_gripper_cfg, default_sim_robot_cfg, xarm7_sim_robot_cfg

    n, nv = len(g["samples"]), g["nv"]
    cfg = xarm7_sim_robot_cfg() if tag == "xarm7" else default_sim_robot_cfg("fr3_empty_world")
    cfg.mjcf_scene_path = cfg.kinematic_model_path = CASES[tag]()
    simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=n)
    if P.KERNEL != "auto":
        simu.set_kernel(P.KERNEL)
    robot = S.SimRobot(simu, None, cfg)
    ctrl = np.array([s["ctrl"] for s in g["samples"]])
    if nv == 9:
        grip = S.SimGripper(simu, default_sim_gripper_cfg())
        grip.set_normalized_width(ctrl[:, 7] / 255.0)
    simu.set_qpos(np.array([s["qpos"] for s in g["samples"]]))
    simu.set_qvel(np.array([s["qvel"] for s in g["samples"]]))
    robot.set_joint_position(ctrl[:, :7])
    simu.step(1)  # the commanded targets reach mjData.ctrl inside this launch
    assert np.abs(simu.ctrl - ctrl).max() < 1e-12, (simu.ctrl[:2], ctrl[:2])
    q, v = simu.qpos, simu.qvel
    simu.close()
    return q, v


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(CASES))
def test_hip_substep_matches_first_principles(tag):
    g = golden(tag)
    q, v = _hip_one_substep(tag, g)
    h = g["timestep"]
    v0 = np.array([s["qvel"] for s in g["samples"]])
    v1 = np.array([s["qvel_next"] for s in g["samples"]])
    q1 = np.array([s["qpos_next"] for s in g["samples"]])
    qacc = np.array([s["qacc_implicit"] for s in g["samples"]])
    assert np.abs(v - v1).max() < REL and np.abs(q - q1).max() < REL, (np.abs(v - v1).max(), np.abs(q - q1).max())
    assert np.abs((v - v0) / h - qacc).max() < 1e-7 * max(1.0, np.abs(qacc).max()), np.abs((v - v0) / h - qacc).max()


# ---- the builder-authored scenes (no reference model: nothing else pins what the shared MJCF compiler makes of them)
AUTHORED = {"ur5e": lambda: P.UR5E_SCENE, "arm6": lambda: P.ARM6_SCENE, "so101": lambda: P.SO101_SCENE, "xarm7_pick": lambda: P.XARM7_PICK_SCENE}


def authored():
    return json.load(open(os.path.join(HERE, "golden", "authored_scene_dynamics.json")))


def xarm7_pick_frictionless_scene() -> str:
    """scenes/xarm7_pick_world with frictionloss = 0 on the arm joints (temporary file): the substep is then the smooth system
    plus the fingers' equality row, which the first-principles vectors predict."""
    import shutil
    import tempfile

    path = os.path.join(tempfile.gettempdir(), "rcs_amd_xarm7_pick_frictionless", "scene.xml")
    if not os.path.exists(path):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        xml = open(P.XARM7_PICK_SCENE).read()
        assert 'frictionloss="1"' in xml
        open(path, "w").write(xml.replace('frictionloss="1"', 'frictionloss="0"'))
        for extra in ("collision_vertices.npz", "render_hulls.npz"):
            shutil.copy(os.path.join(os.path.dirname(P.XARM7_PICK_SCENE), extra), os.path.dirname(path))
    return path


@pytest.mark.parametrize("tag", list(AUTHORED))
def test_authored_scenes_oracle_intermediates_match_first_principles(tag):
    """UR5e / arm6 / SO101 / xarm7_pick_world: the scenes this repository authored.  Same derivation, same bars as for the
    reference's robots; arm6 brings joint axes that are not +z and anchors off the link origin (the kernels' general-axis path),
    SO101 and xarm7_pick a gripper with its equality row.  xarm7_pick's arm joints carry dry friction: the smooth quantities are
    compared on the scene itself, the step on its frictionless variant."""
    import rcs_oracle as O
    from rcs_amd.mjcf import compile_mjcf

    g = authored()["models"][tag]
    nv = g["nv"]
    fric = any(f > 0 for f in g["frictionloss"])
    models = {"own": O.make_model(compile_mjcf(AUTHORED[tag]()), False)}
    if fric:
        models["nofric"] = O.make_model(compile_mjcf(xarm7_pick_frictionless_scene()), False)
    assert models["own"].njnt == nv
    L = O.lib()
    worst = {}
    for s in g["samples"]:
        for kind, m in models.items():
            d = O.OrcData()
            L.orc_reset_data(C.byref(m), C.byref(d))
            for i in range(nv):
                d.qpos[i], d.qvel[i] = s["qpos"][i], s["qvel"][i]
            for i, u in enumerate(s["ctrl"]):
                d.ctrl[i] = u
            L.orc_step1(C.byref(m), C.byref(d))
            L.orc_step2(C.byref(m), C.byref(d))
            got = {}
            if kind == "own":
                got["qM"] = np.array(d.qM[:]).reshape(O.MAXV, O.MAXV)[:nv, :nv]
                for k in ("qfrc_bias", "qfrc_gravcomp", "qfrc_passive", "qfrc_actuator", "qfrc_smooth", "qacc_smooth"):
                    got[k] = getattr(d, k)[:nv]
            if kind == "nofric" or not fric:
                got.update(qvel_next=d.qvel[:nv], qpos_next=d.qpos[:nv])
            for k, v in got.items():
                worst[k] = max(worst.get(k, 0.0), rel_err(v, s[k]))
    assert max(worst.values()) < REL and "qvel_next" in worst, worst


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(AUTHORED))
def test_authored_scenes_hip_substep_matches_first_principles(tag):
    from rcs_amd import sim as S
    from rcs_amd import envs as E

    g = authored()["models"][tag]
    n, nv = len(g["samples"]), g["nv"]
    cfg = {"ur5e": E.ur5e_sim_robot_cfg, "arm6": E.arm6_sim_robot_cfg, "so101": E.so101_sim_robot_cfg, "xarm7_pick": E.xarm7_pick_sim_robot_cfg}[tag]()
    if tag == "xarm7_pick":
        cfg.mjcf_scene_path = cfg.kinematic_model_path = xarm7_pick_frictionless_scene()
    # (xarm7_pick_world: random arm poses reach through the floor and the cube -- the vectors are contact-free mechanics, so the
    # robot's contacts are only detected here, not resolved)
    simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=n, resolve_robot_contacts=False if tag == "xarm7_pick" else None)
    if P.KERNEL != "auto":
        simu.set_kernel(P.KERNEL)
    robot = S.SimRobot(simu, None, cfg)
    ctrl = np.array([s["ctrl"] for s in g["samples"]])
    dof = robot.dof
    if nv > dof:
        grip = S.SimGripper(simu, E.so101_sim_gripper_cfg() if tag == "so101" else E.xarm7_pick_sim_gripper_cfg())
        grip.set_normalized_width(ctrl[:, dof] / 255.0)
    simu.set_qpos(np.array([s["qpos"] for s in g["samples"]]))
    simu.set_qvel(np.array([s["qvel"] for s in g["samples"]]))
    robot.set_joint_position(ctrl[:, :dof])
    simu.step(1)
    assert np.abs(simu.ctrl - ctrl).max() < 1e-12
    q, v = simu.qpos, simu.qvel
    simu.close()
    v1 = np.array([s["qvel_next"] for s in g["samples"]])
    q1 = np.array([s["qpos_next"] for s in g["samples"]])
    assert np.abs(v - v1).max() < REL and np.abs(q - q1).max() < REL, (np.abs(v - v1).max(), np.abs(q - q1).max())
