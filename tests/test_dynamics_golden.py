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
