"""One-call construction of the batched environments for the shipped scenes.

The reference builds its environments through ``SimEnvCreator()(control_mode, robot_cfg, gripper_cfg=..., ...)``
(python/rcs/envs/creators.py:79-128; examples/fr3/fr3_env_joint_control.py:34-41); this is that call with the
per-robot default configurations filled in, for benchmarks, examples and tests.
"""

from __future__ import annotations

import numpy as np

from .. import sim
from .base import ControlMode, RelativeTo
from .creators import SimEnvCreator
from .utils import (arm6_sim_robot_cfg, default_sim_gripper_cfg, default_sim_robot_cfg, so101_sim_gripper_cfg, so101_sim_robot_cfg,
                    ur5e_sim_robot_cfg, xarm7_pick_sim_gripper_cfg, xarm7_pick_sim_robot_cfg, xarm7_sim_robot_cfg)

# max_relative_movement of the reference's joint-control example (examples/fr3/fr3_env_joint_control.py:38)
MAX_JOINT_MOV = float(np.deg2rad(5))

ROBOTS = ("fr3", "xarm7", "xarm7_box", "xarm7_pick", "arm6", "ur5e", "so101")


def robot_cfg_for(robot: str) -> sim.SimRobotConfig:
    if robot == "fr3":
        return default_sim_robot_cfg("fr3_empty_world")
    if robot == "xarm7":
        return xarm7_sim_robot_cfg("xarm7_empty_world")
    if robot == "xarm7_box":
        return xarm7_sim_robot_cfg("xarm7_box_world")
    if robot == "xarm7_pick":
        return xarm7_pick_sim_robot_cfg()
    if robot == "arm6":
        return arm6_sim_robot_cfg()
    if robot == "ur5e":
        return ur5e_sim_robot_cfg()
    if robot == "so101":
        return so101_sim_robot_cfg()
    raise ValueError(f"unknown robot {robot!r}: one of {ROBOTS}")


def make_vec_env(n_envs: int, async_control: bool, gripper: bool = True, relative: bool = True, control_mode=None, device: int = 0,
                 max_relative_movement=None, robot: str = "fr3", relative_to: str = "last_step", frequency: int = 30,
                 max_convergence_steps: int = 500, robot_cfg: sim.SimRobotConfig | None = None, resolve_robot_contacts=None):
    """`n_envs` environments of one robot type on GPU `device`.  `robot_cfg` overrides the robot's default configuration
    (its scene decides the kernel archetype); only the FR3, SO101 and xarm7_pick scenes carry a gripper."""
    cfg = sim.SimConfig(async_control=async_control, realtime=False, frequency=frequency, max_convergence_steps=max_convergence_steps)
    mode = control_mode or ControlMode.JOINTS
    if relative and max_relative_movement is None:
        max_relative_movement = MAX_JOINT_MOV
    if not (robot.startswith("fr3") or robot in ("so101", "xarm7_pick")):
        gripper = False
    gripper_cfg = ((so101_sim_gripper_cfg() if robot == "so101" else xarm7_pick_sim_gripper_cfg() if robot == "xarm7_pick" else default_sim_gripper_cfg())
                   if gripper else None)
    return SimEnvCreator()(
        mode, robot_cfg if robot_cfg is not None else robot_cfg_for(robot),
        gripper_cfg=gripper_cfg,
        sim_cfg=cfg, max_relative_movement=max_relative_movement if relative else None,
        relative_to=RelativeTo.LAST_STEP if relative_to == "last_step" else RelativeTo.CONFIGURED_ORIGIN,
        n_envs=n_envs, device=device, resolve_robot_contacts=resolve_robot_contacts,
    )
