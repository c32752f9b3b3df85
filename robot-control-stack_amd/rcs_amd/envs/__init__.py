"""Batched counterpart of ``rcs.envs`` (reference python/rcs/envs)."""

from .base import ControlMode, RelativeTo  # noqa: F401
from .creators import FR3SimplePickUpSimEnvCreator, SimEnvCreator, SimTaskEnvCreator, VecPickCubeEnv, VecSimEnv  # noqa: F401
from .utils import (arm6_sim_robot_cfg, default_mujoco_cameraset_cfg, default_sim_gripper_cfg, default_sim_robot_cfg,  # noqa: F401
                    so101_sim_gripper_cfg, so101_sim_robot_cfg, ur5e_sim_robot_cfg, xarm7_pick_sim_gripper_cfg, xarm7_pick_sim_robot_cfg,
                    xarm7_sim_robot_cfg)
from .factory import MAX_JOINT_MOV, make_vec_env, robot_cfg_for  # noqa: F401
