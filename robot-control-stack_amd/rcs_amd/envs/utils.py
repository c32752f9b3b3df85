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


def xarm7_sim_robot_cfg(scene: str = "xarm7_empty_world") -> sim.SimRobotConfig:
    """The xArm7 configuration of the reference's example (examples/xarm7/xarm7_env_joint_control.py:41-64): plain
    joint / actuator names (no add_id suffix), no collision geoms, no gripper."""
    import rcs_amd

    cfg = sim.SimRobotConfig()
    cfg.actuators = [f"act{i}" for i in range(1, 8)]
    cfg.joints = [f"joint{i}" for i in range(1, 8)]
    cfg.base = "base"
    cfg.robot_type = rcs_amd.common.RobotType.XArm7
    cfg.attachment_site = "attachment_site"
    cfg.arm_collision_geoms = []
    cfg.mjcf_scene_path = rcs_amd.scenes[scene].mjb
    cfg.kinematic_model_path = rcs_amd.scenes[scene].mjcf_robot
    return cfg


def xarm7_pick_sim_robot_cfg() -> sim.SimRobotConfig:
    """The xArm7 of scenes/xarm7_pick_world (BASELINE configs[3]: the arm of the reference's xarm7.xml with the Franka hand on
    its flange, next to the pick-up scene's cube): the xArm7 example's names, the TCP at the fingertips (0.1034 m along the
    tool axis -- the translation of FrankaHandTCPOffset; the hand is mounted without the FR3's 45 degree turn)."""
    import rcs_amd

    cfg = xarm7_sim_robot_cfg("xarm7_pick_world")
    cfg.tcp_offset = rcs_amd.common.Pose(translation=[0.0, 0.0, 0.1034])
    return cfg


def xarm7_pick_sim_gripper_cfg() -> sim.SimGripperConfig:
    """The Franka hand of scenes/xarm7_pick_world: SimGripperConfig's defaults without the add_id suffix and without the camera
    body the FR3's hand carries."""
    cfg = sim.SimGripperConfig()
    cfg.collision_geoms = ["hand_c", "finger_0_left", "finger_0_right"]
    cfg.collision_geoms_fingers = ["finger_0_left", "finger_0_right"]
    return cfg


def arm6_sim_robot_cfg() -> sim.SimRobotConfig:
    """The builder-authored 6-dof arm (scenes/arm6_empty_world: UR5e-class proportions, NOT a vendor model), configured the
    way the reference's xArm7 example configures its robot; joint limits and home pose are robots_meta_config's UR5e entry."""
    import rcs_amd

    cfg = sim.SimRobotConfig()
    cfg.actuators = [f"act{i}" for i in range(1, 7)]
    cfg.joints = ["shoulder_pan", "shoulder_lift", "elbow", "wrist_1", "wrist_2", "wrist_3"]
    cfg.base = "base"
    cfg.robot_type = rcs_amd.common.RobotType.UR5e
    cfg.attachment_site = "attachment_site"
    cfg.arm_collision_geoms = []
    cfg.mjcf_scene_path = rcs_amd.scenes["arm6_empty_world"].mjb
    cfg.kinematic_model_path = rcs_amd.scenes["arm6_empty_world"].mjcf_robot
    return cfg


def ur5e_sim_robot_cfg() -> sim.SimRobotConfig:
    """The builder-authored UR5e-proportioned arm (scenes/ur5e_empty_world: public DH lengths and link masses, NOT a vendor
    file); joint limits and home pose are robots_meta_config's UR5e entry."""
    import rcs_amd

    cfg = sim.SimRobotConfig()
    cfg.actuators = ["shoulder_pan", "shoulder_lift", "elbow", "wrist_1", "wrist_2", "wrist_3"]
    cfg.joints = [f"{a}_joint" for a in cfg.actuators]
    cfg.base = "base"
    cfg.robot_type = rcs_amd.common.RobotType.UR5e
    cfg.attachment_site = "attachment_site"
    cfg.arm_collision_geoms = []
    cfg.mjcf_scene_path = rcs_amd.scenes["ur5e_empty_world"].mjb
    cfg.kinematic_model_path = rcs_amd.scenes["ur5e_empty_world"].mjcf_robot
    return cfg


def so101_sim_robot_cfg() -> sim.SimRobotConfig:
    """The builder-authored SO-101-proportioned 5-dof arm (scenes/so101_empty_world); home pose and limits are
    robots_meta_config's SO101 entry mapped from the servo bus's normalised units to radians
    (common.sim_robots_meta_config).  Its gripper: `so101_sim_gripper_cfg`."""
    import rcs_amd

    cfg = sim.SimRobotConfig()
    cfg.actuators = [f"act{i}" for i in range(1, 6)]
    cfg.joints = ["shoulder_pan", "shoulder_lift", "elbow_flex", "wrist_flex", "wrist_roll"]
    cfg.base = "base"
    cfg.robot_type = rcs_amd.common.RobotType.SO101
    cfg.attachment_site = "attachment_site"
    cfg.arm_collision_geoms = []
    cfg.mjcf_scene_path = rcs_amd.scenes["so101_empty_world"].mjb
    cfg.kinematic_model_path = rcs_amd.scenes["so101_empty_world"].mjcf_robot
    return cfg


def so101_sim_gripper_cfg() -> sim.SimGripperConfig:
    """Two sliding fingers of 30 mm stroke each behind one tendon actuator (ctrl 0 .. 255), no collision geoms."""
    cfg = sim.SimGripperConfig()
    cfg.max_joint_width = 0.03
    cfg.collision_geoms = []
    cfg.collision_geoms_fingers = []
    cfg.joint = "finger_joint1"
    cfg.actuator = "gripper_act"
    return cfg


def default_sim_gripper_cfg(idx: str = "0") -> sim.SimGripperConfig:
    cfg = sim.SimGripperConfig()
    cfg.add_id(idx)
    return cfg


def default_mujoco_cameraset_cfg():
    """Reference python/rcs/envs/utils.py:60-70 (256 x 256: "needed for VLAs")."""
    from ..camera import CameraType, SimCameraConfig

    return {
        "wrist": SimCameraConfig(identifier="wrist_0", type=CameraType.fixed, frame_rate=10, resolution_width=256, resolution_height=256),
        "default_free": SimCameraConfig(identifier="", type=CameraType.default_free, frame_rate=10, resolution_width=256, resolution_height=256),
    }
