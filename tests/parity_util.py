"""Shared helpers for the parity tests, smoke() and bench.py's checks (uses the oracle as the checker)."""

from __future__ import annotations

import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENE = os.path.join(ROOT, "robot-control-stack_amd", "rcs_amd", "scenes", "fr3_empty_world", "scene.xml")
MAX_JOINT_MOV = float(np.deg2rad(5))


def rpy_distance(a, b) -> float:
    """Angle between the rotations two roll-pitch-yaw triples describe."""
    from rcs_amd.common import Pose

    pa, pb = Pose(rpy_vector=np.asarray(a)), Pose(rpy_vector=np.asarray(b))
    return float((pa * pb.inverse()).total_angle())


def synthetic_actions(n_envs: int, n_steps: int, seed: int = 0, dof: int = 7):
    """SURVEY 8d action tensor: joints ~ U(+-5 deg)^dof f64, gripper ~ U(0,1) f32, one RNG stream per env."""
    joints = np.zeros((n_steps, n_envs, dof))
    grip = np.zeros((n_steps, n_envs), dtype=np.float32)
    for e in range(n_envs):
        rng = np.random.default_rng(seed + e)
        joints[:, e, :] = rng.uniform(-MAX_JOINT_MOV, MAX_JOINT_MOV, size=(n_steps, dof))
        grip[:, e] = rng.uniform(0, 1, size=n_steps).astype(np.float32)
    return joints, grip


def make_vec_env(n_envs: int, async_control: bool, gripper: bool = True, relative: bool = True, control_mode=None, device: int = 0,
                 max_relative_movement=None):
    from rcs_amd import sim
    from rcs_amd.envs import ControlMode, RelativeTo, SimEnvCreator, default_sim_gripper_cfg, default_sim_robot_cfg

    cfg = sim.SimConfig(async_control=async_control, realtime=False, frequency=30)
    mode = control_mode or ControlMode.JOINTS
    if relative and max_relative_movement is None:
        max_relative_movement = MAX_JOINT_MOV
    return SimEnvCreator()(
        mode, default_sim_robot_cfg("fr3_empty_world"), gripper_cfg=default_sim_gripper_cfg() if gripper else None,
        sim_cfg=cfg, max_relative_movement=max_relative_movement if relative else None, relative_to=RelativeTo.LAST_STEP,
        n_envs=n_envs, device=device,
    )


def make_oracle_envs(n_envs: int, async_control: bool, gripper: bool = True, relative: bool = True, mode: str = "joints",
                     max_relative_movement=None):
    from rcs_amd.mjcf import compile_mjcf
    from rcs_env_oracle import OracleEnv

    cm = compile_mjcf(SCENE)
    if relative and max_relative_movement is None:
        max_relative_movement = MAX_JOINT_MOV
    return [OracleEnv(cm, control_mode=mode, gripper=gripper, max_relative_movement=max_relative_movement if relative else None,
                      async_control=async_control) for _ in range(n_envs)]


def run_joint_rollout_parity(n_envs: int = 64, n_steps: int = 3, async_control: bool = True, seed: int = 0, gripper: bool = True,
                             episodes: int = 1):
    """Fused HIP env-step vs the oracle on the same seeded actions; returns max abs differences + flag mismatches."""
    venv = make_vec_env(n_envs, async_control, gripper=gripper)
    oenvs = make_oracle_envs(n_envs, async_control, gripper=gripper)
    joints, grip = synthetic_actions(n_envs, n_steps * episodes, seed)
    rep = {"max_abs_obs": 0.0, "max_abs_qpos": 0.0, "max_abs_qvel": 0.0, "flag_mismatches": 0, "substep_mismatches": 0, "steps": 0}

    def compare(obs, info, oracle_results, substeps=None):
        for e, (oo, oi) in enumerate(oracle_results):
            for k in ("tquat", "joints"):
                rep["max_abs_obs"] = max(rep["max_abs_obs"], float(np.abs(obs[k][e] - oo[k]).max()))
            # xyzrpy: the reference's Euler extraction (yaw in [0, pi], roll near +-pi at a downward-pointing TCP)
            # is discontinuous exactly where the arm lives, so 1e-16 of noise flips the triple; compare the rotation
            rep["max_abs_obs"] = max(rep["max_abs_obs"], float(np.abs(obs["xyzrpy"][e][:3] - oo["xyzrpy"][:3]).max()),
                                     rpy_distance(obs["xyzrpy"][e][3:], oo["xyzrpy"][3:]))
            if gripper:
                rep["flag_mismatches"] += int(float(obs["gripper"][e]) != float(oo["gripper"]))
                rep["max_abs_obs"] = max(rep["max_abs_obs"], abs(float(info["gripper_width"][e]) - oi["gripper_width"]))
            for k in ("collision", "ik_success", "is_sim_converged", "is_grasped"):
                if k in oi and k in info:
                    rep["flag_mismatches"] += int(bool(info[k][e]) != bool(oi[k]))
        q, v = venv.sim.qpos, venv.sim.qvel
        for e, oe in enumerate(oenvs):
            rep["max_abs_qpos"] = max(rep["max_abs_qpos"], float(np.abs(q[e] - oe.sim.qpos).max()))
            rep["max_abs_qvel"] = max(rep["max_abs_qvel"], float(np.abs(v[e] - oe.sim.qvel).max()))
            if substeps is not None and not async_control:
                rep["substep_mismatches"] += int(int(substeps[e]) != int(oe.sim.s.convergence_steps))

    t = 0
    for _ in range(episodes):
        obs, info = venv.reset()
        compare(obs, info, [oe.reset() for oe in oenvs])
        for _ in range(n_steps):
            act = {"joints": joints[t]}
            if gripper:
                act["gripper"] = grip[t]
            obs, _, _, trunc, info = venv.step(act)
            ores = []
            for e, oe in enumerate(oenvs):
                a = {"joints": joints[t, e]}
                if gripper:
                    a["gripper"] = grip[t, e]
                oo, _, _, otr, oi = oe.step(a)
                rep["flag_mismatches"] += int(bool(trunc[e]) != bool(otr))
                ores.append((oo, oi))
            compare(obs, info, ores, info["substeps"])
            t += 1
            rep["steps"] += 1
    venv.close()
    return rep


def cartesian_actions(n_envs: int, n_steps: int, seed: int = 0):
    """xyz ~ U(+-0.05)^3 m, rpy ~ U(+-0.1)^3 rad, gripper toggled every 10 steps."""
    act = np.zeros((n_steps, n_envs, 6))
    grip = np.zeros((n_steps, n_envs), dtype=np.float32)
    for e in range(n_envs):
        rng = np.random.default_rng(seed + e)
        act[:, e, :3] = rng.uniform(-0.05, 0.05, size=(n_steps, 3))
        act[:, e, 3:] = rng.uniform(-0.1, 0.1, size=(n_steps, 3))
        grip[:, e] = ((np.arange(n_steps) // 10) % 2).astype(np.float32)
    return act, grip


def run_cartesian_rollout_parity(n_envs=32, n_steps=4, async_control=True, seed=0, mode="xyzrpy", relative=True, gripper=True):
    """Cartesian control (relative TRPY / TQuat actions -> CLIK -> joint targets): HIP path vs the oracle."""
    from rcs_amd.envs import ControlMode

    cm = ControlMode.CARTESIAN_TRPY if mode == "xyzrpy" else ControlMode.CARTESIAN_TQuat
    mm = (0.2, float(np.deg2rad(45)))
    venv = make_vec_env(n_envs, async_control, gripper=gripper, relative=relative, control_mode=cm, max_relative_movement=mm)
    oenvs = make_oracle_envs(n_envs, async_control, gripper=gripper, relative=relative, mode=mode, max_relative_movement=mm)
    act6, grip = cartesian_actions(n_envs, n_steps, seed)
    rep = {"max_abs_qpos": 0.0, "max_abs_tquat": 0.0, "max_abs_target": 0.0, "flag_mismatches": 0, "steps": 0, "ik_fail": 0}
    obs, info = venv.reset()
    ores = [oe.reset() for oe in oenvs]
    for t in range(n_steps):
        if mode == "xyzrpy":
            a_all = act6[t]
        else:  # relative tquat action: translation + quaternion of the small rpy rotation
            from rcs_amd.common import Pose

            a_all = np.stack([np.concatenate([act6[t, e, :3], Pose(rpy_vector=act6[t, e, 3:]).rotation_q()]) for e in range(n_envs)])
        if not relative:  # absolute target = current pose shifted
            base = obs["xyzrpy"] if mode == "xyzrpy" else obs["tquat"]
            a_all = base.copy()
            a_all[:, :3] += act6[t, :, :3]
        action = {mode: a_all}
        if gripper:
            action["gripper"] = grip[t]
        obs, _, _, trunc, info = venv.step(action)
        st = venv.robot.get_state()
        q = venv.sim.qpos
        for e, oe in enumerate(oenvs):
            a = {mode: a_all[e]}
            if gripper:
                a["gripper"] = grip[t, e]
            oo, _, _, otr, oi = oe.step(a)
            rep["flag_mismatches"] += int(bool(trunc[e]) != bool(otr))
            for k in ("collision", "ik_success", "is_sim_converged"):
                rep["flag_mismatches"] += int(bool(info[k][e]) != bool(oi[k]))
            rep["ik_fail"] += int(not oi["ik_success"])
            rep["max_abs_qpos"] = max(rep["max_abs_qpos"], float(np.abs(q[e] - oe.sim.qpos).max()))
            rep["max_abs_tquat"] = max(rep["max_abs_tquat"], float(np.abs(obs["tquat"][e] - oo["tquat"]).max()))
            rep["max_abs_target"] = max(rep["max_abs_target"], float(np.abs(st.target_angles[e] - np.array(oe.sim.s.target_angles[:7])).max()))
        rep["steps"] += 1
    venv.close()
    return rep
