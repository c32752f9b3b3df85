"""Default configs (reference python/rcs/envs/utils.py:17-38)."""

from __future__ import annotations

from .. import sim


def default_sim_robot_cfg(scene: str = "fr3_empty_world", idx: str = "0") -> sim.SimRobotConfig:
    import rcs_amd

    cfg = sim.SimRobotConfig()
    cfg.robot_type = rcs_amd.scenes[scene].robot_type
    cfg.add_id(idx)
    cfg.mjcf_scene_path = rcs_amd.scenes[scene].mjb
    cfg.kinematic_model_path = rcs_amd.scenes[scene].mjcf_robot
    return cfg


def default_sim_gripper_cfg(idx: str = "0") -> sim.SimGripperConfig:
    cfg = sim.SimGripperConfig()
    cfg.add_id(idx)
    return cfg
