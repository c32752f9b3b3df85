"""Single-environment restatement of the reference Gymnasium loop (TEST INFRASTRUCTURE).

Collapses the wrapper stack that ``SimEnvCreator`` builds (reference
python/rcs/envs/creators.py:79-128) into one class whose ``reset`` / ``step``
perform the same side effects in the same order:

    RelativeActionSpace   base.py:365-565
      GripperWrapperSim   envs/sim.py:119-131
        GripperWrapper    base.py:680-735
          RobotSimWrapper envs/sim.py:35-76
            RobotEnv      base.py:191-304

The physics / adapter calls go to the C oracle (oracle/rcs_oracle.py).
"""

from __future__ import annotations

import copy

import numpy as np

import rcs_oracle as O

JOINTS, CARTESIAN_TRPY, CARTESIAN_TQUAT = "joints", "xyzrpy", "tquat"
LAST_STEP, CONFIGURED_ORIGIN = "last_step", "configured_origin"

FR3_Q_HOME = np.array([0.0, -np.pi / 4, 0.0, -3.0 * np.pi / 4, 0.0, np.pi / 2, np.pi / 4])  # Robot.h:29-31
FR3_LOW = np.array([-2.3093, -1.5133, -2.4937, -2.7478, -2.4800, 0.8521, -2.6895])  # Robot.h:36-40
FR3_HIGH = np.array([2.3093, 1.5133, 2.4937, -0.4461, 2.4800, 4.2094, 2.6895])
# robot descriptions: names as the reference configures them, home pose and joint limits from robots_meta_config
FR3 = dict(joints=[f"fr3_joint{i}_0" for i in range(1, 8)], actuators=[f"fr3_joint{i}_0" for i in range(1, 8)],
           site="attachment_site_0", base="base_0", q_home=FR3_Q_HOME, low=FR3_LOW, high=FR3_HIGH,
           gripper_joint="finger_joint1_0", gripper_actuator="actuator8_0")
XARM7 = dict(  # examples/xarm7/xarm7_env_joint_control.py:41-64; include/rcs/Robot.h (XArm7 entry)
    joints=[f"joint{i}" for i in range(1, 8)], actuators=[f"act{i}" for i in range(1, 8)], site="attachment_site", base="base",
    q_home=np.array([0, -45.0 / 180.0 * np.pi, 0, 15.0 / 180.0 * np.pi, 0, -25.0 / 180.0 * np.pi, 0]),
    low=np.array([-2 * np.pi, -2.094395, -2 * np.pi, -3.92699, -2 * np.pi, -np.pi, -2 * np.pi]),
    high=np.array([2 * np.pi, 2.059488, 2 * np.pi, 0.191986, 2 * np.pi, 1.692969, 2 * np.pi]),
    gripper_joint=None, gripper_actuator=None, arm_collision_geoms=[])
# scenes/xarm7_pick_world (builder-authored: the xArm7 with the Franka hand on its flange next to the pick-up cube, BASELINE configs[3])
XARM7_PICK = dict(XARM7, gripper_joint="finger_joint1", gripper_actuator="actuator8",
                  gripper_cfg=dict(collision_geoms=["hand_c", "finger_0_left", "finger_0_right"], collision_geoms_fingers=["finger_0_left", "finger_0_right"]))
ARM6 = dict(  # scenes/arm6_empty_world (builder-authored 6-dof arm); home pose and limits: Robot.h's UR5e entry
    joints=["shoulder_pan", "shoulder_lift", "elbow", "wrist_1", "wrist_2", "wrist_3"], actuators=[f"act{i}" for i in range(1, 7)],
    site="attachment_site", base="base",
    q_home=np.array([-0.4488354, -2.02711196, 1.64630026, -1.18999615, -1.57079762, -2.01963249]),
    low=np.array([-2 * np.pi, -2 * np.pi, -np.pi, -2 * np.pi, -2 * np.pi, -2 * np.pi]),
    high=np.array([2 * np.pi, 2 * np.pi, np.pi, 2 * np.pi, 2 * np.pi, 2 * np.pi]),
    gripper_joint=None, gripper_actuator=None, arm_collision_geoms=[])
UR5E = dict(  # scenes/ur5e_empty_world (builder-authored, public DH lengths); home pose and limits: Robot.h's UR5e entry
    joints=[f"{a}_joint" for a in ("shoulder_pan", "shoulder_lift", "elbow", "wrist_1", "wrist_2", "wrist_3")],
    actuators=["shoulder_pan", "shoulder_lift", "elbow", "wrist_1", "wrist_2", "wrist_3"], site="attachment_site", base="base",
    q_home=ARM6["q_home"], low=ARM6["low"], high=ARM6["high"], gripper_joint=None, gripper_actuator=None, arm_collision_geoms=[])
# scenes/so101_empty_world (builder-authored 5-dof arm + two-finger gripper).  Robot.h's SO101 entry is in the servo bus's
# normalised units (-100 .. 100); a simulation reads it mapped linearly onto the joints' ranges (rcs_amd.common.sim_robots_meta_config)
_SO101_LO = np.array([-1.91986, -1.74533, -1.69, -1.65806, -2.74385])
_SO101_HI = np.array([1.91986, 1.74533, 1.69, 1.65806, 2.84121])
_SO101_HOME_NORM = np.array([-9.40612320177057, -99.66130397967824, 99.9124726477024, 69.96996996996998, -9.095744680851055])  # Robot.h:82-84
SO101 = dict(
    joints=["shoulder_pan", "shoulder_lift", "elbow_flex", "wrist_flex", "wrist_roll"], actuators=[f"act{i}" for i in range(1, 6)],
    site="attachment_site", base="base", q_home=_SO101_LO + (_SO101_HOME_NORM + 100.0) / 200.0 * (_SO101_HI - _SO101_LO),
    low=_SO101_LO.copy(), high=_SO101_HI.copy(), gripper_joint="finger_joint1", gripper_actuator="gripper_act", arm_collision_geoms=[],
    gripper_cfg=dict(max_joint_width=0.03, collision_geoms=[], collision_geoms_fingers=[]))
TRPY_LOW = np.array([-0.855, -0.855, 0.0])  # base.py:31-38
TRPY_HIGH = np.array([0.855, 0.855, 1.188])


class OracleEnv:
    def __init__(self, cm, control_mode=JOINTS, gripper=True, max_relative_movement=None, relative_to=LAST_STEP,
                 async_control=False, frequency=30, max_convergence_steps=500, tcp_offset: O.Pose | None = None,
                 robot: dict | None = None):
        robot = FR3 if robot is None else robot  # SimRobot.h:23-30 + add_id("0")
        gripper = gripper and robot["gripper_joint"] is not None
        self.robot = robot
        self.sim = O.Sim(
            cm, robot["joints"], robot["actuators"], robot["site"], robot["base"], robot["q_home"], tcp_offset,
            gripper_joint=robot["gripper_joint"] if gripper else None,
            gripper_actuator=robot["gripper_actuator"] if gripper else None,
            arm_collision_geoms=robot.get("arm_collision_geoms"), gripper_cfg=robot.get("gripper_cfg"),
        )
        self.sim.set_config(async_control=async_control, frequency=frequency, max_convergence_steps=max_convergence_steps)
        self.timestep = cm.timestep
        self.mode = control_mode
        self.has_gripper = gripper
        self.max_mov = max_relative_movement
        if self.max_mov is not None and control_mode != JOINTS and not isinstance(self.max_mov, tuple):
            self.max_mov = (float(self.max_mov), np.deg2rad(90))  # base.py:381-385
        self.relative_to = relative_to
        self.prev_action = None  # RobotEnv.prev_action, base.py:227 (never cleared, quirk Q2)
        self._last_gripper_cmd = None
        self._origin = None
        self._last_action = None

    # ------------------------------------------------------------------ obs
    def _get_obs(self):  # base.py:246-253
        pose = self.sim.get_cartesian_position()
        return {
            "tquat": np.concatenate([pose.translation(), pose.rotation_q()]),
            "joints": self.sim.get_joint_position(),
            "xyzrpy": pose.xyzrpy(),
        }

    def _set_origin_to_current(self):  # base.py:455-459
        self._origin = self.sim.get_joint_position() if self.mode == JOINTS else self.sim.get_cartesian_position()

    def _gripper_obs(self, obs, info):
        if not self.has_gripper:
            return obs, info
        obs = dict(obs)
        obs["gripper"] = self._last_gripper_cmd if self._last_gripper_cmd is not None else 1  # base.py:710-715
        s = self.sim.s  # envs/sim.py:125-131
        if "collision" not in info or not info["collision"]:
            info["collision"] = bool(s.grp_collision)
        w = self.sim.gripper_get_normalized_width()
        info["gripper_width"] = w
        info["is_grasped"] = 0.01 < w < 0.99
        return obs, info

    # ---------------------------------------------------------------- reset
    def reset(self):
        if self.has_gripper:  # GripperWrapper.reset, base.py:703-708 (before sim.reset, quirk Q1)
            self.sim.gripper_reset()
            self._last_gripper_cmd = None
        self.sim.reset()  # RobotSimWrapper.reset, envs/sim.py:68-76
        self.sim.robot_reset()  # RobotEnv.reset, base.py:290-304
        self.sim.step(1)
        obs, info = self._gripper_obs(self._get_obs(), {})
        if self.max_mov is not None:  # RelativeActionSpace.reset, base.py:461-466
            self._set_origin_to_current()
            self._last_action = None
        return obs, info

    # ----------------------------------------------------------------- step
    def _relative_action(self, action):  # RelativeActionSpace.action, base.py:468-565
        if self.relative_to == LAST_STEP:
            self._set_origin_to_current()
        action = copy.deepcopy(action)
        fresh = self.relative_to == LAST_STEP or self._last_action is None
        if self.mode == JOINTS:
            a = np.asarray(action["joints"], dtype=np.float64)
            if fresh:
                limited = np.clip(a, -self.max_mov, self.max_mov)
            else:
                limited = np.clip(a - self._last_action, -self.max_mov, self.max_mov) + self._last_action
            self._last_action = limited
            action["joints"] = np.clip(self._origin + limited, self.robot["low"], self.robot["high"])
            return action
        key = self.mode
        a = np.asarray(action[key], dtype=np.float64)
        given = O.Pose(translation=a[:3], rpy_vector=a[3:]) if key == CARTESIAN_TRPY else O.Pose(translation=a[:3], quaternion=a[3:])
        if fresh:
            offset = given.limit_translation_length(self.max_mov[0]).limit_rotation_angle(self.max_mov[1])
        else:
            diff = given * self._last_action.inverse()
            offset = diff.limit_translation_length(self.max_mov[0]).limit_rotation_angle(self.max_mov[1]) * self._last_action
        self._last_action = offset
        rot = offset * self._origin
        t = self._origin.translation() + offset.translation()
        if key == CARTESIAN_TRPY:
            unclipped = O.Pose(translation=t, rpy_vector=rot.rotation_rpy())
            action[key] = np.concatenate([np.clip(unclipped.translation(), TRPY_LOW, TRPY_HIGH), unclipped.rotation_rpy()])
        else:
            unclipped = O.Pose(translation=t, quaternion=rot.rotation_q())
            action[key] = np.concatenate([np.clip(unclipped.translation(), TRPY_LOW, TRPY_HIGH), unclipped.rotation_q()])
        return action

    def step(self, action):
        if self.max_mov is not None:
            action = self._relative_action(action)
        else:
            action = copy.deepcopy(action)
        if self.has_gripper:  # GripperWrapper.action, base.py:721-735
            g = np.clip(np.round(action["gripper"]), 0.0, 1.0)
            if g == 0:
                self.sim.gripper_grasp()
            else:
                self.sim.gripper_open()
            self._last_gripper_cmd = g
            del action["gripper"]
        # RobotEnv.step, base.py:255-288
        key = self.mode
        a = np.asarray(action[key], dtype=np.float64)
        changed = self.prev_action is None or not np.allclose(a, self.prev_action[key], atol=1e-3, rtol=0)
        if changed:
            if key == JOINTS:
                self.sim.set_joint_position(a)
            elif key == CARTESIAN_TRPY:
                self.sim.set_cartesian_position(O.Pose(translation=a[:3], rpy_vector=a[3:]))
            else:
                self.sim.set_cartesian_position(O.Pose(translation=a[:3], quaternion=a[3:]))
        self.prev_action = copy.deepcopy(action)
        # RobotSimWrapper.step, envs/sim.py:49-66
        s = self.sim.s
        if s.async_control:
            self.sim.step(round(1 / s.frequency / self.timestep))
        else:
            self.sim.step_until_convergence()
        info = {"collision": bool(s.robot_collision), "ik_success": bool(s.ik_success), "is_sim_converged": self.sim.is_converged()}
        truncated = bool(s.robot_collision) or not bool(s.ik_success)
        obs, info = self._gripper_obs(self._get_obs(), info)
        return obs, 0, False, truncated, info


class OraclePickCubeEnv(OracleEnv):
    """``SimTaskEnvCreator()(...)`` with the default gripper (reference creators.py:131-187): ``RandomCubePos`` sits
    between the RobotSimWrapper and the RobotEnv, ``PickCubeSuccessWrapper`` outermost.

        PickCubeSuccessWrapper   envs/sim.py:386-431
          RelativeActionSpace
            GripperWrapperSim / GripperWrapper
              RobotSimWrapper
                RandomCubePos    envs/sim.py:358-383
                  RobotEnv
    """

    EE_HOME = np.array([0.34169773, 0.00047028, 0.4309004])

    def __init__(self, cm, control_mode=CARTESIAN_TRPY, delta_actions=True, async_control=True, frequency=30,
                 tcp_offset: O.Pose | None = None, include_rotation=True):
        super().__init__(cm, control_mode=control_mode, gripper=True,
                         max_relative_movement=(0.2, np.deg2rad(45)) if delta_actions else None, relative_to=LAST_STEP,
                         async_control=async_control, frequency=frequency, tcp_offset=tcp_offset)
        self.include_rotation = include_rotation

    def reset(self, box_qpos=None):
        # GripperWrapper.reset -> RobotSimWrapper.reset: sim.reset() -> RandomCubePos.reset: RobotEnv.reset,
        # sim.step(1), box_joint qpos := placement -> RobotSimWrapper: sim.step(1), observe
        self.sim.gripper_reset()
        self._last_gripper_cmd = None
        self.sim.reset()
        self.sim.robot_reset()
        self.sim.step(1)
        if box_qpos is None:  # sim.py:371-383 (global numpy generator, this order of draws)
            iso = (self.sim.get_base_pose() * O.Pose(translation=[0.498, 0.0, 0.226], rpy_vector=[0, 0, 0])).translation()
            x = iso[0] + np.random.random() * 0.2 - 0.1
            y = iso[1] + np.random.random() * 0.2 - 0.1
            box_qpos = [x, y, 0.0288 / 2, 2 * np.random.random() - 1 if self.include_rotation else 0, 0, 0, 1]
        self.sim.box_qpos = box_qpos
        self.sim.step(1)
        obs, info = self._gripper_obs(self._get_obs(), {})
        if self.max_mov is not None:
            self._set_origin_to_current()
            self._last_action = None
        return obs, info

    def step(self, action):
        obs, reward, _, truncated, info = super().step(action)
        box = self.sim.box_qpos
        success = bool(box[2] > 0.15 + 0.852 and obs["gripper"] == 0)  # BINARY_GRIPPER_CLOSED
        info["success"] = success
        if success:
            reward = 5
        else:
            tcp_to_obj = np.linalg.norm(box[:3] - self.sim.get_cartesian_position().translation())
            obj_to_goal = np.linalg.norm(box[:3] - self.EE_HOME)
            reward = 1 - np.tanh(5 * tcp_to_obj)
            reward += info["is_grasped"]
            reward += (1 - np.tanh(5 * obj_to_goal)) * info["is_grasped"]
        reward /= 5
        return obs, reward, success, truncated, info
