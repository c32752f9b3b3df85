"""``SimEnvCreator`` for N environments (reference python/rcs/envs/creators.py:43-128).

The reference builds one Gymnasium env out of nested wrappers around one MuJoCo
instance.  Here the same constructor arguments produce a :class:`VecSimEnv`:
``reset()`` / ``step(action)`` have the Gymnasium signatures, but every array
carries a leading ``n_envs`` axis and one call is one kernel launch that performs
the wrappers' side effects, the physics substeps and the observation for all
environments (csrc/sim_kernels.h: k_run).
"""

from __future__ import annotations

import ctypes as C
from typing import Any

import numpy as np

from .. import _lib, common, sim
from .base import ControlMode, RelativeTo

_MODE = {ControlMode.JOINTS: 0, ControlMode.CARTESIAN_TRPY: 1, ControlMode.CARTESIAN_TQuat: 2}
_REL = {None: 0, RelativeTo.LAST_STEP: 1, RelativeTo.CONFIGURED_ORIGIN: 2}
_ACTION_KEY = {ControlMode.JOINTS: "joints", ControlMode.CARTESIAN_TRPY: "xyzrpy", ControlMode.CARTESIAN_TQuat: "tquat"}

DEFAULT_MAX_CART_MOV = 0.5  # reference base.py:366-368
DEFAULT_MAX_CART_ROT = np.deg2rad(90)
DEFAULT_MAX_JOINT_MOV = np.deg2rad(5)


class VecSimEnv:
    """N-environment ``gym.Env`` look-alike returned by :class:`SimEnvCreator`.

    obs dict: ``tquat [N,7]``, ``joints [N,dof]``, ``xyzrpy [N,6]`` (+ ``gripper [N]``);
    info dict: ``collision``, ``ik_success``, ``is_sim_converged`` (+ ``gripper_width``, ``is_grasped``), all ``[N]``;
    ``step`` returns ``(obs, reward [N] = 0, terminated [N] = False, truncated [N], info)``.
    """

    def __init__(self, simulation: sim.Sim, robot: sim.SimRobot, gripper: sim.SimGripper | None,
                 control_mode: ControlMode, max_relative_movement, relative_to: RelativeTo, camera_set=None):
        self.camera_set = camera_set  # CameraSetWrapper(env, camera_set, include_depth=True), base.py:585-677
        self.sim = simulation
        self.robot = robot
        self.gripper = gripper
        self.control_mode = control_mode
        self.n_envs = simulation.n_envs
        self.dof = robot.dof
        self._L = simulation._L
        meta = common.sim_robots_meta_config(robot.get_config().robot_type)
        self._low = np.ascontiguousarray(meta.joint_limits[0][: self.dof], dtype=np.float64)
        self._high = np.ascontiguousarray(meta.joint_limits[1][: self.dof], dtype=np.float64)
        rel = 0
        max_mov = [0.0, 0.0]
        if max_relative_movement is not None:
            rel = _REL[relative_to]
            if control_mode == ControlMode.JOINTS:
                assert isinstance(max_relative_movement, float), "joint-space max_mov must be a float (rad)"
                max_mov = [float(max_relative_movement), 0.0]
            elif isinstance(max_relative_movement, tuple):
                max_mov = [float(max_relative_movement[0]), float(max_relative_movement[1])]
            else:
                max_mov = [float(max_relative_movement), float(DEFAULT_MAX_CART_ROT)]
        self.max_mov = max_mov
        self.relative_to = relative_to if rel else None
        d = _lib.EnvDesc()
        d.control_mode = _MODE[control_mode]
        d.relative_to = rel
        d.max_mov[:] = max_mov
        d.binary_gripper = 1
        d.joint_low = self._low.ctypes.data_as(C.POINTER(C.c_double))
        d.joint_high = self._high.ctypes.data_as(C.POINTER(C.c_double))
        _lib.check(self._L.rcsh_env_configure(simulation._h, C.byref(d)))
        self.obs_width = self._L.rcsh_env_obs_width(simulation._h)
        self.action_width = self._L.rcsh_env_action_width(simulation._h)
        self.action_key = _ACTION_KEY[control_mode]
        # What an environment gets whose geoms are found in a contact this configuration does not resolve (info["contact_unresolved"],
        # sticky until reset; csrc/check_team.h).  The reference resolves every contact in every mode (mj_step2, sim.cpp:112); the
        # lean kernels of scenes without a free body do not, because the capability costs the contact-free rollout ~15 %.
        #   "resolve": (default, round 5) nothing stays unresolved: the Sim resolves robot contacts environment by environment
        #              (Sim(resolve_robot_contacts=None): mode 7 in scenes without a free body), robot <-> robot included.  A Sim that
        #              was created with resolve_robot_contacts=False is switched to that mode as soon as a step reports a contact
        #              (host-array interface only: the `*_dev` entry points never read the flags back);
        #   "flag":    report only -- with a Sim created with resolve_robot_contacts=False the environment steps on unresolved (the arm
        #              passes through the floor / itself) and the caller decides (mask it, reset it, discard the episode).
        self.on_unresolved_contact = "resolve"

    def _after_step(self, info) -> None:
        if self.on_unresolved_contact == "resolve" and not self.sim.resolve_robot_contacts and info[:, 7].any():
            self.sim.enable_contact_resolution()

    # ---- host-array interface (Gymnasium-shaped)
    def _unpack(self, obs, info, gw) -> tuple[dict[str, Any], dict[str, Any]]:
        d = self.dof
        o: dict[str, Any] = {"tquat": obs[:, 0:7].copy(), "joints": obs[:, 7 : 7 + d].copy(), "xyzrpy": obs[:, 7 + d : 13 + d].copy()}
        i: dict[str, Any] = {}
        if self.gripper is not None:
            o["gripper"] = obs[:, 13 + d].copy()
        if self.camera_set is not None:
            # CameraSetWrapper.observation (base.py:633-674), include_depth=True: "rgb" always, "depth" next to it
            frameset = self.camera_set.get_latest_frames()
            if frameset is None:
                o["frames"] = {}
                i["camera_available"] = False
            else:
                o["frames"] = {name: {"rgb": {"data": f.camera.color.data, "intrinsics": f.camera.color.intrinsics,
                                              "extrinsics": f.camera.color.extrinsics},
                                      "depth": {"data": f.camera.depth.data, "intrinsics": f.camera.depth.intrinsics,
                                                "extrinsics": f.camera.depth.extrinsics}} for name, f in frameset.frames.items()}
                i["camera_available"] = True
                if frameset.avg_timestamp is not None:
                    i["frame_timestamp"] = frameset.avg_timestamp
        return o, i

    def reset(self, seed: int | None = None, options: dict | None = None, mask=None):
        n = self.n_envs
        obs = np.zeros((n, self.obs_width))
        info = np.zeros((n, 8), dtype=np.uint8)
        gw = np.zeros(n)
        m = None if mask is None else np.ascontiguousarray(np.asarray(mask).astype(np.uint8))
        if self.camera_set is not None:
            self.camera_set.clear_buffer()  # CameraSetWrapper.reset
        _lib.check(self._L.rcsh_env_reset(self.sim._h, _lib.ptr(m), _lib.ptr(obs), _lib.ptr(info), _lib.ptr(gw)))
        self.sim._collect_frames()
        o, i = self._unpack(obs, info, gw)
        if self.gripper is not None:  # GripperWrapperSim.observation runs on reset too (envs/sim.py:125-131)
            i["collision"] = info[:, 5].astype(bool)
            i["gripper_width"] = gw
            i["is_grasped"] = info[:, 3].astype(bool)
        return o, i

    def step(self, action: dict[str, Any]):
        n = self.n_envs
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(action[self.action_key], dtype=np.float64), (n, self.action_width)))
        g = None
        if self.gripper is not None:
            assert "gripper" in action, "Gripper action not found."
            g = np.ascontiguousarray(np.broadcast_to(np.asarray(action["gripper"], dtype=np.float32), (n,)))
        obs = np.zeros((n, self.obs_width))
        info = np.zeros((n, 8), dtype=np.uint8)
        gw = np.zeros(n)
        sub = np.zeros(n, dtype=np.int32)
        _lib.check(self._L.rcsh_env_step(self.sim._h, _lib.ptr(a), _lib.ptr(g), _lib.ptr(obs), _lib.ptr(info), _lib.ptr(gw), _lib.ptr(sub)))
        self.sim._collect_frames()
        o, i = self._unpack(obs, info, gw)
        i["collision"] = info[:, 0].astype(bool)
        i["ik_success"] = info[:, 1].astype(bool)
        i["is_sim_converged"] = info[:, 2].astype(bool)
        if self.gripper is not None:
            i["gripper_width"] = gw
            i["is_grasped"] = info[:, 3].astype(bool)
        i["substeps"] = sub
        # (no reference counterpart) a contact phase of the environment ran out of contact / link slots since its last reset:
        # its trajectory is no longer what MuJoCo would compute (csrc/contact_team.h: kMaxCon, kMaxActive)
        i["contact_overflow"] = info[:, 6].astype(bool)
        # (no reference counterpart) the environment's geoms were found in a contact this configuration does not resolve, since its
        # last reset (csrc/check_team.h): MuJoCo would have resolved it, so from that step on the trajectory is not MuJoCo's
        i["contact_unresolved"] = info[:, 7].astype(bool)
        self._after_step(info)
        truncated = info[:, 4].astype(bool)
        return o, np.zeros(n), np.zeros(n, dtype=bool), truncated, i

    # ---- device-pointer interface for resident rollouts (pointers are integers / c_void_p)
    def reset_dev(self, obs_ptr, info_ptr=None, gw_ptr=None, mask_ptr=None) -> None:
        _lib.check(self._L.rcsh_env_reset_dev(self.sim._h, C.c_void_p(mask_ptr), C.c_void_p(obs_ptr), C.c_void_p(info_ptr), C.c_void_p(gw_ptr)))

    def step_dev(self, action_ptr, gripper_ptr, obs_ptr, info_ptr=None, gw_ptr=None, substeps_ptr=None) -> None:
        _lib.check(self._L.rcsh_env_step_dev(self.sim._h, C.c_void_p(action_ptr), C.c_void_p(gripper_ptr), C.c_void_p(obs_ptr),
                                             C.c_void_p(info_ptr), C.c_void_p(gw_ptr), C.c_void_p(substeps_ptr)))

    def close(self) -> None:
        self.sim.close()


class VecPickCubeEnv(VecSimEnv):
    """``SimTaskEnvCreator()(...)`` of the reference for N environments: the :class:`VecSimEnv` of the pick-up scene
    with ``RandomCubePos`` under the simulation wrapper and ``PickCubeSuccessWrapper`` on top (reference
    python/rcs/envs/sim.py:358-431, creators.py:131-187).

    ``reset`` places each environment's cube (one draw of x, y and -- with ``include_rotation`` -- the quaternion's w per
    environment from numpy's global generator, in the reference's order; or the poses given as
    ``options={"box_qpos": [N, 7]}``); ``step`` returns the wrapper's shaped reward, ``terminated = success`` and
    ``info["success"]``; the cube pose of the step is ``info["box_qpos"]``.
    """

    EE_HOME = np.array([0.34169773, 0.00047028, 0.4309004])  # PickCubeSuccessWrapper.EE_HOME
    SUCCESS_HEIGHT = 0.15 + 0.852
    ISO_CUBE = np.array([0.498, 0.0, 0.226])  # RandomCubePos.reset, robot coordinates

    def __init__(self, *args, include_rotation: bool = True, random_pos_args: dict | None = None, **kwargs):
        super().__init__(*args, **kwargs)
        assert self.gripper is not None, "PickCubeSuccessWrapper reads the gripper observation"
        self.include_rotation = include_rotation
        # RandomObjectPos instead of RandomCubePos (creators.py:160-167): joint_name, init_object_pose, include_position, include_rotation
        self.random_pos_args = random_pos_args
        if random_pos_args is not None:
            self.sim._free_joint(random_pos_args["joint_name"])  # KeyError for an unknown joint, like mjData.joint()
            self.init_object_pose = random_pos_args["init_object_pose"]
        t = _lib.PickTaskDesc()
        t.ee_home[:] = [float(x) for x in self.EE_HOME]
        t.success_height = float(self.SUCCESS_HEIGHT)
        _lib.check(self._L.rcsh_env_configure_pick_task(self.sim._h, C.byref(t)))
        self.task_width = 9

    def draw_box_qpos(self) -> np.ndarray:
        """RandomCubePos.reset's placement (sim.py:371-383) -- or RandomObjectPos.reset's (sim.py:331-354) -- for every
        environment, drawing from numpy's global generator in the reference's order."""
        if self.random_pos_args is not None:
            return random_object_qpos(self.init_object_pose, self.n_envs, self.random_pos_args.get("include_position", True),
                                      self.random_pos_args.get("include_rotation", False))
        iso = self.robot.to_pose_in_world_coordinates(common.Pose(translation=self.ISO_CUBE, rpy_vector=np.zeros(3))).translation()
        q = np.zeros((self.n_envs, 7))
        for e in range(self.n_envs):
            x = iso[0] + np.random.random() * 0.2 - 0.1
            y = iso[1] + np.random.random() * 0.2 - 0.1
            w = 2 * np.random.random() - 1 if self.include_rotation else 0.0
            q[e] = [x, y, 0.0288 / 2, w, 0, 0, 1]
        return q

    def reset(self, seed: int | None = None, options: dict | None = None, mask=None):
        n = self.n_envs
        if options is not None and "RandomObjectPos.init_object_pose" in options and self.random_pos_args is not None:
            assert isinstance(options["RandomObjectPos.init_object_pose"], common.Pose), "RandomObjectPos.init_object_pose must be a rcs.common.Pose"
            self.init_object_pose = options["RandomObjectPos.init_object_pose"]  # sticks for later resets, as in the reference
        box = None if options is None else options.get("box_qpos")
        box = self.draw_box_qpos() if box is None else np.ascontiguousarray(np.broadcast_to(np.asarray(box, dtype=np.float64), (n, 7)))
        obs = np.zeros((n, self.obs_width))
        info = np.zeros((n, 8), dtype=np.uint8)
        gw = np.zeros(n)
        m = None if mask is None else np.ascontiguousarray(np.asarray(mask).astype(np.uint8))
        if self.camera_set is not None:
            self.camera_set.clear_buffer()
        _lib.check(self._L.rcsh_env_reset_task(self.sim._h, _lib.ptr(m), _lib.ptr(box), _lib.ptr(obs), _lib.ptr(info), _lib.ptr(gw)))
        self.sim._collect_frames()
        o, i = self._unpack(obs, info, gw)
        i["collision"] = info[:, 5].astype(bool)
        i["gripper_width"] = gw
        i["is_grasped"] = info[:, 3].astype(bool)
        return o, i

    def step(self, action: dict[str, Any]):
        n = self.n_envs
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(action[self.action_key], dtype=np.float64), (n, self.action_width)))
        assert "gripper" in action, "Gripper action not found."
        g = np.ascontiguousarray(np.broadcast_to(np.asarray(action["gripper"], dtype=np.float32), (n,)))
        obs = np.zeros((n, self.obs_width))
        info = np.zeros((n, 8), dtype=np.uint8)
        gw = np.zeros(n)
        sub = np.zeros(n, dtype=np.int32)
        task = np.zeros((n, self.task_width))
        _lib.check(self._L.rcsh_env_step_task(self.sim._h, _lib.ptr(a), _lib.ptr(g), _lib.ptr(obs), _lib.ptr(info), _lib.ptr(gw),
                                              _lib.ptr(sub), _lib.ptr(task)))
        self.sim._collect_frames()
        o, i = self._unpack(obs, info, gw)
        i["collision"] = info[:, 0].astype(bool)
        i["ik_success"] = info[:, 1].astype(bool)
        i["is_sim_converged"] = info[:, 2].astype(bool)
        i["gripper_width"] = gw
        i["is_grasped"] = info[:, 3].astype(bool)
        i["substeps"] = sub
        i["contact_overflow"] = info[:, 6].astype(bool)  # (see VecSimEnv.step)
        i["contact_unresolved"] = info[:, 7].astype(bool)
        i["box_qpos"] = task[:, :7].copy()
        success = task[:, 8] != 0
        i["success"] = success
        return o, task[:, 7].copy(), success, info[:, 4].astype(bool), i

    def step_task_dev(self, action_ptr, gripper_ptr, obs_ptr, info_ptr=None, gw_ptr=None, substeps_ptr=None, task_ptr=None) -> None:
        _lib.check(self._L.rcsh_env_step_task_dev(self.sim._h, C.c_void_p(action_ptr), C.c_void_p(gripper_ptr), C.c_void_p(obs_ptr),
                                                  C.c_void_p(info_ptr), C.c_void_p(gw_ptr), C.c_void_p(substeps_ptr), C.c_void_p(task_ptr)))

    def reset_task_dev(self, box_qpos_ptr, obs_ptr, info_ptr=None, gw_ptr=None, mask_ptr=None) -> None:
        _lib.check(self._L.rcsh_env_reset_task_dev(self.sim._h, C.c_void_p(mask_ptr), C.c_void_p(box_qpos_ptr), C.c_void_p(obs_ptr),
                                                   C.c_void_p(info_ptr), C.c_void_p(gw_ptr)))


def random_object_qpos(init_object_pose: common.Pose, n_envs: int, include_position: bool = True, include_rotation: bool = False) -> np.ndarray:
    """RandomObjectPos.reset (reference python/rcs/envs/sim.py:331-354) for n_envs environments: x, y +- 0.1 m around the
    initial pose, z and orientation kept -- with include_rotation the quaternion's w becomes ``2 u - w`` (unnormalised;
    mj_kinematics normalises it).  Draw order per environment: x, y, then w."""
    t, q = init_object_pose.translation(), init_object_pose.rotation_q()  # xyzw
    out = np.zeros((n_envs, 7))
    for e in range(n_envs):
        x = t[0] + np.random.random() * 0.2 - 0.1 if include_position else t[0]
        y = t[1] + np.random.random() * 0.2 - 0.1 if include_position else t[1]
        w = 2 * np.random.random() - q[3] if include_rotation else q[3]
        out[e] = [x, y, t[2], w, q[0], q[1], q[2]]
    return out


class SimEnvCreator:
    def __call__(self, control_mode: ControlMode, robot_cfg: sim.SimRobotConfig, collision_guard: bool = False,
                 gripper_cfg: sim.SimGripperConfig | None = None, sim_cfg: sim.SimConfig | None = None,
                 hand_cfg=None, cameras=None, max_relative_movement: float | tuple[float, float] | None = None,
                 relative_to: RelativeTo = RelativeTo.LAST_STEP, sim_wrapper=None, n_envs: int = 1, device: int = 0,
                 resolve_robot_contacts=None) -> VecSimEnv:
        if hand_cfg is not None or sim_wrapper is not None or collision_guard:
            raise NotImplementedError("hands, sim_wrapper and collision_guard are outside this backend's hot path")
        simulation = sim.Sim(robot_cfg.mjcf_scene_path, sim_cfg, n_envs=n_envs, device=device, resolve_robot_contacts=resolve_robot_contacts)
        robot = sim.SimRobot(simulation, None, robot_cfg)
        gripper = sim.SimGripper(simulation, gripper_cfg) if gripper_cfg is not None else None
        camera_set = None
        if cameras is not None:  # creators.py:92-96
            from ..camera import SimCameraSet

            camera_set = SimCameraSet(simulation, cameras, physical_units=True, render_on_demand=True)
        return VecSimEnv(simulation, robot, gripper, control_mode, max_relative_movement, relative_to, camera_set=camera_set)


class SimTaskEnvCreator:
    """Reference python/rcs/envs/creators.py:131-187: the pick-up task on top of ``SimEnvCreator`` (RandomCubePos +
    PickCubeSuccessWrapper).  ``render_mode`` is accepted for signature compatibility; this backend has no viewer."""

    def __call__(self, robot_cfg: sim.SimRobotConfig, render_mode: str = "rgb_array", control_mode: ControlMode = ControlMode.CARTESIAN_TRPY,
                 delta_actions: bool = True, cameras=None, hand_cfg=None, gripper_cfg: sim.SimGripperConfig | None = None,
                 sim_cfg: sim.SimConfig | None = None, random_pos_args: dict | None = None, n_envs: int = 1, device: int = 0) -> VecPickCubeEnv:
        from .utils import default_sim_gripper_cfg

        if hand_cfg is not None:
            raise NotImplementedError("hands are outside this backend's hot path")
        if random_pos_args is not None:
            missing = [k for k in ("joint_name", "init_object_pose") if k not in random_pos_args]
            if missing:
                raise TypeError(f"RandomObjectPos.__init__() missing required arguments: {missing}")  # partial(RandomObjectPos, **args) would fail here
        if gripper_cfg is None:
            gripper_cfg = default_sim_gripper_cfg()
        simulation = sim.Sim(robot_cfg.mjcf_scene_path, sim_cfg, n_envs=n_envs, device=device)
        robot = sim.SimRobot(simulation, None, robot_cfg)
        gripper = sim.SimGripper(simulation, gripper_cfg)
        camera_set = None
        if cameras:
            from ..camera import SimCameraSet

            camera_set = SimCameraSet(simulation, cameras, physical_units=True, render_on_demand=True)
        return VecPickCubeEnv(simulation, robot, gripper, control_mode,
                              (0.2, float(np.deg2rad(45))) if delta_actions else None, RelativeTo.LAST_STEP, camera_set=camera_set,
                              random_pos_args=random_pos_args)


class FR3SimplePickUpSimEnvCreator:
    """Reference creators.py:190-224 (gym id ``rcs/FR3SimplePickUpSim-v0``): the pick-up scene at 30 Hz async control
    with the reference's TCP offset."""

    def __call__(self, render_mode: str = "rgb_array", control_mode: ControlMode = ControlMode.CARTESIAN_TRPY,
                 resolution: tuple[int, int] | None = None, frame_rate: int = 0, delta_actions: bool = True,
                 cam_list: list[str] | None = None, n_envs: int = 1, device: int = 0) -> VecPickCubeEnv:
        from .utils import default_sim_robot_cfg

        from ..camera import CameraType, SimCameraConfig

        if resolution is None:
            resolution = (256, 256)
        cameras = {cam: SimCameraConfig(identifier=cam, type=CameraType.fixed, resolution_height=resolution[1], resolution_width=resolution[0],
                                        frame_rate=frame_rate) for cam in (cam_list or [])}
        robot_cfg = default_sim_robot_cfg(scene="fr3_simple_pick_up")
        robot_cfg.tcp_offset = common.Pose(translation=np.array([0.0, 0.0, 0.1034]),
                                           rotation=np.array([[0.707, 0.707, 0], [-0.707, 0.707, 0], [0, 0, 1]]))
        sim_cfg = sim.SimConfig()
        sim_cfg.realtime = False
        sim_cfg.async_control = True
        sim_cfg.frequency = 30
        return SimTaskEnvCreator()(robot_cfg, render_mode, control_mode, delta_actions, cameras, sim_cfg=sim_cfg, n_envs=n_envs, device=device)
