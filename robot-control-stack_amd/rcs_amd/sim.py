"""Batched mirror of ``rcs.sim`` (reference python/rcs/sim/sim.py, src/pybind/rcs.cpp:421-527).

``Sim`` / ``SimRobot`` / ``SimGripper`` keep the reference's class, method and
config-field names; every method acts on all ``n_envs`` environments of the
handle (arrays carry a leading environment axis, ``mask`` selects a subset).
The state lives in HBM; these classes only marshal arguments across the C-ABI.
"""

from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field
from os import PathLike
from pathlib import Path

import numpy as np

from . import _lib, common
from .mjcf import Model, compile_mjcf


@dataclass
class SimConfig:  # reference src/sim/sim.h:29-34
    async_control: bool = False
    realtime: bool = False
    frequency: int = 30
    max_convergence_steps: int = 500


def _mask(mask, n):
    if mask is None:
        return None
    m = np.ascontiguousarray(np.asarray(mask).astype(np.uint8))
    assert m.shape == (n,)
    return m


class Sim:
    """``rcs.sim.Sim(mjmdl, cfg)`` for ``n_envs`` independent copies of the scene (reference sim.py:44-62).

    ``mjmdl`` is the scene ``.xml`` path (MuJoCo's private ``.mjb`` binaries cannot be read without MuJoCo).
    ``model`` is the compiled scene (``opt_timestep`` etc.), the stand-in for ``mujoco.MjModel``.
    """

    def __init__(self, mjmdl: str | PathLike, cfg: SimConfig | None = None, n_envs: int = 1, device: int = 0,
                 resolve_robot_contacts: bool | None = None):
        path = Path(mjmdl)
        if path.suffix == ".mjb":
            path = path.with_suffix(".xml")  # scenes are registered by their .mjb name in the reference
        if path.suffix != ".xml":
            raise RuntimeError(f"Filetype {path.suffix} is unknown")
        self.scene_path = str(path)
        self.model: Model = compile_mjcf(str(path))
        self.n_envs = int(n_envs)
        self.device = int(device)
        self._L = _lib.load()
        desc, self._keep = _lib.make_model_desc(self.model)
        self._h = C.c_void_p()
        _lib.check(self._L.rcsh_sim_create(C.byref(desc), self.n_envs, self.device, C.byref(self._h)))
        # collision geoms beyond the contact table's capacity are invisible to geom-geom detection: say so (an obstacle no callback
        # list names would still count for SimRobot::collision_callback in the reference); listing one of them as a collision
        # geom of SimRobot / SimGripper is refused by the library
        self.undetected_collision_geoms = self._contact_table_dropped()
        if self.undetected_collision_geoms:
            import warnings

            names = [self.model.geom_names[g] or f"#{g}" for g in self.undetected_collision_geoms]
            warnings.warn(f"collision geoms {names} exceed the contact table's capacity ({self._contact_table_reason}): their collisions "
                          "with other geoms are NOT detected (the floor test still sees them)", RuntimeWarning, stacklevel=2)
        unchecked = C.c_int32(0)
        _lib.check(self._L.rcsh_sim_contact_check_unchecked_pairs(self._h, C.byref(unchecked)))
        self.unchecked_geom_pairs = int(unchecked.value)
        if self.unchecked_geom_pairs:
            import warnings

            warnings.warn(f"{self.unchecked_geom_pairs} admitted geom pairs exceed what the end-of-launch contact check holds: a contact of "
                          "one of them raises no contact_unresolved flag and is not resolved as a self contact", RuntimeWarning, stacklevel=2)
        # Contacts of the robot's collision geoms (with the floor, with a free body): RESOLVED by default where there is
        # something to manipulate (scenes with a free body: the pick-up task), DETECTED only (collision flags, as the
        # callbacks need them) in scenes without -- there the contact-capable kernel costs the no-contact rollout ~15 %, so it
        # is opt-in: Sim(..., resolve_robot_contacts=True) makes the arm stop on the floor instead of passing through it.
        has_free = bool(getattr(self.model, "free_bodies", []))
        # (an int selects what is resolved and how -- include/rcs_hip.h, rcsh_contact_options: bit 0 robot <-> floor / free body, bit 1
        # robot <-> robot too, bit 2 environment by environment: the lean kernel for the environments that touch nothing)
        # Default (round 5): what MuJoCo does -- every contact of the robot's geoms is resolved.  Scenes without a free body do it
        # environment by environment (mode 7: the lean kernel steps the environments that touch nothing, an environment found in
        # contact has that launch redone by the contact-resolving kernel and stays on it while the contact lasts); scenes with a
        # free body run the whole batch on the contact-resolving kernel as before (mode 1).  False: contacts are detected only
        # (collision flags, info["contact_unresolved"]).
        resolve = (1 if has_free else 7) if resolve_robot_contacts is None else resolve_robot_contacts
        mode = ((1 if has_free else 7) if resolve else 0) if isinstance(resolve, bool) else int(resolve)
        if mode and not mode & 1:
            mode |= 1
        self.resolve_robot_contacts = mode if self.resolves_robot_contacts(self.model) else 0
        box = _lib.make_free_box_desc(self.model, self.resolve_robot_contacts)
        if box is not None:
            _lib.check(self._L.rcsh_sim_add_free_box(self._h, C.byref(box)))
        elif self.resolve_robot_contacts:
            opts = _lib.make_contact_options(self.model, self.resolve_robot_contacts)
            try:
                _lib.check(self._L.rcsh_sim_set_contact_options(self._h, C.byref(opts)))
            except RuntimeError:
                if resolve_robot_contacts is not None:
                    raise
                # the DEFAULT asks for contacts to be resolved; a scene the contact phase cannot hold (more collision geoms than its
                # table) still loads and steps -- its contacts are detected only, as the warning above says
                self.resolve_robot_contacts = 0
        self._cfg = SimConfig()
        if cfg is not None:
            self.set_config(cfg)

    def _contact_table_dropped(self) -> list[int]:
        import numpy as np

        count = C.c_int32(0)
        ids = np.zeros(64, dtype=np.int32)
        reason = C.create_string_buffer(256)
        _lib.check(self._L.rcsh_sim_contact_table_dropped(self._h, ids.ctypes.data_as(C.POINTER(C.c_int32)), 64, C.byref(count), reason, 256))
        self._contact_table_reason = reason.value.decode()
        return [int(g) for g in ids[: min(count.value, 64)]]

    def contact_escalated(self):
        """([N] bool, [N] bool): the environment is on the contact-resolving kernel right now / a contact of its robot geoms has been
        resolved since its last Sim.reset (per-environment escalation: resolve_robot_contacts bit 2)."""
        import numpy as np

        now, ever = np.zeros(self.n_envs, dtype=np.uint8), np.zeros(self.n_envs, dtype=np.uint8)
        _lib.check(self._L.rcsh_sim_contact_escalated(self._h, _lib.ptr(now), _lib.ptr(ever)))
        return now.astype(bool), ever.astype(bool)

    def enable_contact_resolution(self) -> bool:
        """Switch a scene WITHOUT a free body to the contact-resolving kernels from the next launch on (robot <-> floor contacts
        enter the constraint solve; rcsh_sim_set_contact_options).  False where the archetype cannot (see resolves_robot_contacts)."""
        if self.resolve_robot_contacts:
            return True
        if getattr(self.model, "free_bodies", []) or not self.resolves_robot_contacts(self.model):
            return False
        opts = _lib.make_contact_options(self.model, 7)
        _lib.check(self._L.rcsh_sim_set_contact_options(self._h, C.byref(opts)))
        self.resolve_robot_contacts = 7
        return True

    def set_contact_check(self, every: int) -> None:
        """Cadence of the end-of-launch check for contacts nobody resolves: every `every`-th stepping launch (default 1 -- exact
        per env-step; 0: off).  See rcsh_sim_set_contact_check (include/rcs_hip.h) for what a larger cadence trades."""
        _lib.check(self._L.rcsh_sim_set_contact_check(self._h, int(every)))

    def contact_unresolved(self) -> np.ndarray:
        """[N] bool: the environment's geoms were found in a contact this configuration does not resolve, since its last
        Sim.reset (the sticky flag the end-of-launch check sets, csrc/check_team.h; also info["contact_unresolved"])."""
        import numpy as np

        out = np.zeros(self.n_envs, dtype=np.uint8)
        _lib.check(self._L.rcsh_sim_contact_unresolved(self._h, _lib.ptr(out)))
        return out.astype(bool)

    @staticmethod
    def resolves_robot_contacts(model: Model) -> bool:
        """Scenes whose robot-geom contacts CAN enter the constraint solve: the 7-dof arm + two-finger gripper archetype with
        elliptic cones and collision geoms -- FR3 + hand; with dry joint friction (xArm7 + gripper) only next to a free body and
        without a noslip pass (the friction-dof rows would take part in it).  Elsewhere contacts only raise the collision flags."""
        import numpy as np

        if not (model.njnt == 9 and model.nu == 8 and model.cone == "elliptic" and model.ngeom > 1):
            return False
        if np.any(np.asarray(model.arrays["dof_frictionloss"]) > 0):
            return bool(getattr(model, "free_bodies", [])) and model.noslip_iterations == 0
        return True

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._L.rcsh_sim_destroy(h)
            self._h = None

    close = __del__

    # -- reference Sim API (rcs.cpp:493-506)
    def set_config(self, cfg: SimConfig) -> bool:
        for cs in getattr(self, "_rate_camera_sets", ()):
            # the longest launch this configuration produces: step_until_convergence's cap, or one asynchronous env-step
            conv = cfg.max_convergence_steps if cfg.max_convergence_steps > 0 else 2000
            cs.ensure_capacity(max(conv, int(round(1.0 / (max(cfg.frequency, 1) * self.model.timestep)))))
        self._cfg = SimConfig(cfg.async_control, cfg.realtime, cfg.frequency, cfg.max_convergence_steps)
        _lib.check(self._L.rcsh_sim_set_config(self._h, int(cfg.async_control), int(cfg.realtime), int(cfg.frequency),
                                               int(cfg.max_convergence_steps)))
        return True

    def get_config(self) -> SimConfig:
        return SimConfig(self._cfg.async_control, self._cfg.realtime, self._cfg.frequency, self._cfg.max_convergence_steps)

    def step(self, k: int) -> None:
        for cs in getattr(self, "_rate_camera_sets", ()):
            cs.ensure_capacity(int(k))  # (a launch longer than the render schedule was sized for)
        _lib.check(self._L.rcsh_sim_step(self._h, int(k)))
        self._collect_frames()

    def step_until_convergence(self) -> None:
        _lib.check(self._L.rcsh_sim_step_until_convergence(self._h))
        self._collect_frames()

    def _collect_frames(self) -> None:
        """Rendering callbacks (Sim::invoke_rendering_callbacks, sim.cpp:63-81): camera sets with render_on_demand=False render
        the frames that became due inside the launch that just ran (rcs_amd/camera.py).  Called by every host-array stepping
        entry point of Sim and of the environments; users of the `*_dev` entry points call `camera_set.collect()` themselves."""
        for cs in getattr(self, "_rate_camera_sets", ()):
            cs.collect()

    def is_converged(self) -> np.ndarray:
        out = np.zeros(self.n_envs, dtype=np.uint8)
        _lib.check(self._L.rcsh_sim_is_converged(self._h, _lib.ptr(out), None))
        return out.astype(bool)

    def convergence_steps(self) -> np.ndarray:
        out = np.zeros(self.n_envs, dtype=np.uint8)
        steps = np.zeros(self.n_envs, dtype=np.int32)
        _lib.check(self._L.rcsh_sim_is_converged(self._h, _lib.ptr(out), _lib.ptr(steps)))
        return steps

    def reset(self, mask=None) -> None:
        _lib.check(self._L.rcsh_sim_reset(self._h, _lib.ptr(_mask(mask, self.n_envs))))

    def set_stream(self, hip_stream: int | None) -> None:
        """Enqueue all further work on a caller-owned HIP stream (None: back to the handle's own stream)."""
        _lib.check(self._L.rcsh_sim_set_stream(self._h, C.c_void_p(hip_stream)))

    def set_kernel(self, variant: str) -> None:
        """Pin the kernel variant: "auto" / "team" (16 lanes per environment).  "lane", the one-lane kernel of ABI 1, was removed: ValueError."""
        _lib.check(self._L.rcsh_sim_set_kernel(self._h, {"auto": 0, "team": 1, "lane": 2, "team_occ2": 3}[variant]))

    def get_state(self) -> np.ndarray:
        """Opaque snapshot of everything that evolves (physics, callback scheduler, robot / gripper / wrapper state)."""
        blob = np.empty(int(self._L.rcsh_sim_state_bytes(self._h)), dtype=np.uint8)
        _lib.check(self._L.rcsh_sim_get_state(self._h, C.c_void_p(blob.ctypes.data)))
        return blob

    def set_state(self, blob: np.ndarray) -> None:
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        if blob.size != int(self._L.rcsh_sim_state_bytes(self._h)):
            raise ValueError("state blob of a different scene or batch size")
        _lib.check(self._L.rcsh_sim_set_state(self._h, C.c_void_p(blob.ctypes.data)))

    def synchronize(self) -> None:
        _lib.check(self._L.rcsh_sim_synchronize(self._h))

    # -- mjData views (reference python/rcs/envs/sim.py:343-411 reads data.joint(name).qpos)
    def _get(self, fn, width, dtype=np.float64):
        out = np.zeros((self.n_envs, width), dtype=dtype)
        _lib.check(fn(self._h, _lib.ptr(out)))
        return out

    @property
    def qpos(self) -> np.ndarray:
        return self._get(self._L.rcsh_sim_get_qpos, self.model.nq)

    @property
    def qvel(self) -> np.ndarray:
        return self._get(self._L.rcsh_sim_get_qvel, self.model.nv)

    @property
    def ctrl(self) -> np.ndarray:
        return self._get(self._L.rcsh_sim_get_ctrl, self.model.nu)

    @property
    def time(self) -> np.ndarray:
        return self._get(self._L.rcsh_sim_get_time, 1)[:, 0]

    def set_qpos(self, qpos, mask=None) -> None:
        q = np.ascontiguousarray(np.broadcast_to(np.asarray(qpos, dtype=np.float64), (self.n_envs, self.model.nq)))
        _lib.check(self._L.rcsh_sim_set_qpos(self._h, _lib.ptr(q), _lib.ptr(_mask(mask, self.n_envs))))

    # ``sim.data.joint(name).qpos`` of a free joint (reference python/rcs/envs/sim.py:379-383,399-412)
    def _free_joint(self, name: str) -> dict:
        for fb in getattr(self.model, "free_bodies", []):
            if fb["joint_name"] == name:
                return fb
        raise KeyError(f"Invalid name '{name}'. Valid free joints: {[fb['joint_name'] for fb in getattr(self.model, 'free_bodies', [])]}")

    def free_joint_qpos(self, name: str) -> np.ndarray:
        """[n_envs, 7]: x y z qw qx qy qz."""
        self._free_joint(name)
        return self._get(self._L.rcsh_sim_get_free_qpos, 7)

    def free_joint_qvel(self, name: str) -> np.ndarray:
        self._free_joint(name)
        return self._get(self._L.rcsh_sim_get_free_qvel, 6)

    def set_free_joint_qpos(self, name: str, qpos, mask=None) -> None:
        self._free_joint(name)
        q = np.ascontiguousarray(np.broadcast_to(np.asarray(qpos, dtype=np.float64), (self.n_envs, 7)))
        _lib.check(self._L.rcsh_sim_set_free_qpos(self._h, _lib.ptr(q), _lib.ptr(_mask(mask, self.n_envs))))

    def set_free_joint_qvel(self, name: str, qvel, mask=None) -> None:
        self._free_joint(name)
        v = np.ascontiguousarray(np.broadcast_to(np.asarray(qvel, dtype=np.float64), (self.n_envs, 6)))
        _lib.check(self._L.rcsh_sim_set_free_qvel(self._h, _lib.ptr(v), _lib.ptr(_mask(mask, self.n_envs))))

    def set_qvel(self, qvel, mask=None) -> None:
        q = np.ascontiguousarray(np.broadcast_to(np.asarray(qvel, dtype=np.float64), (self.n_envs, self.model.nv)))
        _lib.check(self._L.rcsh_sim_set_qvel(self._h, _lib.ptr(q), _lib.ptr(_mask(mask, self.n_envs))))


@dataclass
class SimRobotConfig(common.RobotConfig):  # reference src/sim/SimRobot.h:14-47
    joint_rotational_tolerance: float = 0.05 * (math.pi / 180.0)
    seconds_between_callbacks: float = 0.1
    trajectory_trace: bool = False
    arm_collision_geoms: list[str] = field(default_factory=lambda: [f"fr3_link{i}_collision" for i in range(8)])
    joints: list[str] = field(default_factory=lambda: [f"fr3_joint{i}" for i in range(1, 8)])
    actuators: list[str] = field(default_factory=lambda: [f"fr3_joint{i}" for i in range(1, 8)])
    base: str = "base"
    mjcf_scene_path: str = "assets/scenes/fr3_empty_world/scene.xml"

    def add_id(self, id: str) -> None:  # noqa: A002  (reference name)
        self.arm_collision_geoms = [f"{s}_{id}" for s in self.arm_collision_geoms]
        self.joints = [f"{s}_{id}" for s in self.joints]
        self.actuators = [f"{s}_{id}" for s in self.actuators]
        self.attachment_site = f"{self.attachment_site}_{id}"
        self.base = f"{self.base}_{id}"


@dataclass
class SimRobotState:  # SimRobot.h:49-57, one entry per environment
    previous_angles: np.ndarray
    target_angles: np.ndarray
    inverse_tcp_offset: common.Pose
    ik_success: np.ndarray
    collision: np.ndarray
    is_moving: np.ndarray
    is_arrived: np.ndarray


def _lookup(model: Model, kind: str, name: str, label: str) -> int:
    i = model.name2id(kind, name)
    if i < 0:
        raise RuntimeError(f"No {label} named {name}")  # SimRobot.cpp:57-93
    return i


class DeviceKinematics:
    """``rcs.common.Kinematics`` (rcs.cpp:289-295) on the simulated chain: batched CLIK / FK kernels (csrc/ik.h).

    The reference builds a separate pinocchio model from the robot's MJCF (``common.Pin``, creators.py:81-85); here the
    IK frame and chain are the scene's own link tables, so no second model is loaded.
    """

    def __init__(self, sim: "Sim", dof: int):
        self.sim, self.dof = sim, dof

    def _tcp(self, tcp_offset):
        if tcp_offset is None:
            return None
        v = tcp_offset.as_vec7() if isinstance(tcp_offset, common.Pose) else np.asarray(tcp_offset, dtype=np.float64)
        return np.ascontiguousarray(v.reshape(7))

    def inverse(self, pose, q0, tcp_offset=None):
        """pose [N,7] (or Pose), q0 [N,dof] -> (q [N,nq], success [N], iterations [N]); failed rows of q are unspecified."""
        n = self.sim.n_envs
        if isinstance(pose, common.Pose):
            pose = pose.as_vec7()
        p = np.ascontiguousarray(np.broadcast_to(np.asarray(pose, dtype=np.float64), (n, 7)))
        q0 = np.ascontiguousarray(np.broadcast_to(np.asarray(q0, dtype=np.float64)[..., : self.dof], (n, self.dof)))
        q = np.zeros((n, self.sim.model.nq))
        ok = np.zeros(n, dtype=np.uint8)
        it = np.zeros(n, dtype=np.int32)
        _lib.check(self.sim._L.rcsh_ik_inverse(self.sim._h, _lib.ptr(p), _lib.ptr(q0), _lib.ptr(self._tcp(tcp_offset)), _lib.ptr(q),
                                               _lib.ptr(ok), _lib.ptr(it)))
        return q, ok.astype(bool), it

    def forward(self, q0, tcp_offset=None) -> np.ndarray:
        n = self.sim.n_envs
        q0 = np.ascontiguousarray(np.broadcast_to(np.asarray(q0, dtype=np.float64)[..., : self.dof], (n, self.dof)))
        out = np.zeros((n, 7))
        _lib.check(self.sim._L.rcsh_ik_forward(self.sim._h, _lib.ptr(q0), _lib.ptr(self._tcp(tcp_offset)), _lib.ptr(out)))
        return out


class SimRobot:
    """``rcs.sim.SimRobot(sim, ik, cfg, register_convergence_callback=True)`` (rcs.cpp:516-527)."""

    def __init__(self, sim: Sim, ik, cfg: SimRobotConfig, register_convergence_callback: bool = True):
        self.sim = sim
        self._ik = ik
        self._cfg = cfg
        self._L = sim._L
        m = sim.model
        cgeoms = np.array([_lookup(m, "geom", g, "geom") for g in cfg.arm_collision_geoms], dtype=np.int32)
        site = _lookup(m, "site", cfg.attachment_site, "site")
        base = _lookup(m, "body", cfg.base, "body")
        joints = np.array([_lookup(m, "jnt", j, "joint") for j in cfg.joints], dtype=np.int32)
        acts = np.array([_lookup(m, "actuator", a, "actuator") for a in cfg.actuators], dtype=np.int32)
        meta = common.sim_robots_meta_config(cfg.robot_type)
        self.dof = len(joints)
        q_home = np.ascontiguousarray(meta.q_home[: self.dof], dtype=np.float64)
        d = _lib.RobotDesc()
        d.dof = self.dof
        d.joint_ids = joints.ctypes.data_as(C.POINTER(C.c_int32))
        d.actuator_ids = acts.ctypes.data_as(C.POINTER(C.c_int32))
        d.attachment_site, d.base_body = site, base
        d.q_home = q_home.ctypes.data_as(C.POINTER(C.c_double))
        d.tcp_offset[:] = [float(x) for x in cfg.tcp_offset.as_vec7()]
        d.joint_rotational_tolerance = cfg.joint_rotational_tolerance
        d.seconds_between_callbacks = cfg.seconds_between_callbacks
        d.register_convergence_callback = int(register_convergence_callback)
        d.n_collision_geoms = len(cgeoms)
        d.collision_geom_ids = cgeoms.ctypes.data_as(C.POINTER(C.c_int32))
        _lib.check(self._L.rcsh_sim_add_robot(sim._h, C.byref(d)))
        if self._ik is None:
            self._ik = DeviceKinematics(sim, self.dof)

    @property
    def n_envs(self) -> int:
        return self.sim.n_envs

    def _q(self, q):
        return np.ascontiguousarray(np.broadcast_to(np.asarray(q, dtype=np.float64)[..., : self.dof], (self.n_envs, self.dof)))

    def get_config(self) -> SimRobotConfig:
        return self._cfg

    def get_state(self) -> SimRobotState:
        n = self.n_envs
        ik, col, mov, arr = (np.zeros(n, dtype=np.uint8) for _ in range(4))
        prev, tgt = np.zeros((n, self.dof)), np.zeros((n, self.dof))
        _lib.check(self._L.rcsh_robot_get_state(self.sim._h, _lib.ptr(ik), _lib.ptr(col), _lib.ptr(mov), _lib.ptr(arr),
                                                _lib.ptr(prev), _lib.ptr(tgt)))
        return SimRobotState(prev, tgt, self._cfg.tcp_offset.inverse(), ik.astype(bool), col.astype(bool), mov.astype(bool),
                             arr.astype(bool))

    def get_cartesian_position(self) -> np.ndarray:
        """[n_envs, 7] x y z qx qy qz qw, robot frame, TCP offset applied (SimRobot.cpp:114-121)."""
        out = np.zeros((self.n_envs, 7))
        _lib.check(self._L.rcsh_robot_get_cartesian_position(self.sim._h, _lib.ptr(out)))
        return out

    def get_cartesian_pose(self, env: int = 0) -> common.Pose:
        v = self.get_cartesian_position()[env]
        return common.Pose(translation=v[:3], quaternion=v[3:])

    def set_joint_position(self, q, mask=None) -> None:
        _lib.check(self._L.rcsh_robot_set_joint_position(self.sim._h, _lib.ptr(self._q(q)), _lib.ptr(_mask(mask, self.n_envs))))

    def get_joint_position(self) -> np.ndarray:
        out = np.zeros((self.n_envs, self.dof))
        _lib.check(self._L.rcsh_robot_get_joint_position(self.sim._h, _lib.ptr(out)))
        return out

    def move_home(self, mask=None) -> None:
        _lib.check(self._L.rcsh_robot_move_home(self.sim._h, _lib.ptr(_mask(mask, self.n_envs))))

    def reset(self, mask=None) -> None:
        _lib.check(self._L.rcsh_robot_reset(self.sim._h, _lib.ptr(_mask(mask, self.n_envs))))

    def close(self) -> None:
        pass

    def set_cartesian_position(self, pose, mask=None) -> None:
        if isinstance(pose, common.Pose):
            pose = pose.as_vec7()
        p = np.ascontiguousarray(np.broadcast_to(np.asarray(pose, dtype=np.float64), (self.n_envs, 7)))
        _lib.check(self._L.rcsh_robot_set_cartesian_position(self.sim._h, _lib.ptr(p), _lib.ptr(_mask(mask, self.n_envs))))

    def get_ik(self):
        return self._ik

    def get_base_pose_in_world_coordinates(self) -> common.Pose:
        out = np.zeros((self.n_envs, 7))
        _lib.check(self._L.rcsh_robot_get_base_pose(self.sim._h, _lib.ptr(out)))
        return common.Pose(translation=out[0, :3], quaternion=out[0, 3:])

    def to_pose_in_robot_coordinates(self, pose_in_world_coordinates: common.Pose) -> common.Pose:
        return self.get_base_pose_in_world_coordinates().inverse() * pose_in_world_coordinates

    def to_pose_in_world_coordinates(self, pose_in_robot_coordinates: common.Pose) -> common.Pose:
        return self.get_base_pose_in_world_coordinates() * pose_in_robot_coordinates

    def set_joints_hard(self, q, mask=None) -> None:
        _lib.check(self._L.rcsh_robot_set_joints_hard(self.sim._h, _lib.ptr(self._q(q)), _lib.ptr(_mask(mask, self.n_envs))))


@dataclass
class SimGripperConfig:  # reference src/sim/SimGripper.h:15-45
    epsilon_inner: float = 0.005
    epsilon_outer: float = 0.005
    seconds_between_callbacks: float = 0.05
    max_actuator_width: float = 255
    min_actuator_width: float = 0
    max_joint_width: float = 0.04
    min_joint_width: float = 0.0
    ignored_collision_geoms: list[str] = field(default_factory=list)
    collision_geoms: list[str] = field(default_factory=lambda: ["hand_c", "d435i_collision", "finger_0_left", "finger_0_right"])
    collision_geoms_fingers: list[str] = field(default_factory=lambda: ["finger_0_left", "finger_0_right"])
    joint: str = "finger_joint1"
    actuator: str = "actuator8"

    def add_id(self, id: str) -> None:  # noqa: A002
        self.collision_geoms = [f"{s}_{id}" for s in self.collision_geoms]
        self.collision_geoms_fingers = [f"{s}_{id}" for s in self.collision_geoms_fingers]
        self.ignored_collision_geoms = [f"{s}_{id}" for s in self.ignored_collision_geoms]
        self.joint = f"{self.joint}_{id}"
        self.actuator = f"{self.actuator}_{id}"


@dataclass
class SimGripperState:  # SimGripper.h:47-52, one entry per environment
    last_commanded_width: np.ndarray
    is_moving: np.ndarray
    last_width: np.ndarray
    collision: np.ndarray


class SimGripper:
    """``rcs.sim.SimGripper(sim, cfg)`` (rcs.cpp:508-515)."""

    def __init__(self, sim: Sim, cfg: SimGripperConfig):
        self.sim = sim
        self._cfg = cfg
        self._L = sim._L
        m = sim.model
        act = _lookup(m, "actuator", cfg.actuator, "actuator")
        jnt = _lookup(m, "jnt", cfg.joint, "joint")
        ids = lambda names: np.array([_lookup(m, "geom", g, "geom") for g in names] or [0], dtype=np.int32)  # noqa: E731
        cg, cf, ig = ids(cfg.collision_geoms), ids(cfg.collision_geoms_fingers), ids(cfg.ignored_collision_geoms)
        i32p = C.POINTER(C.c_int32)
        d = _lib.GripperDesc(jnt, act, cfg.epsilon_inner, cfg.epsilon_outer, cfg.seconds_between_callbacks,
                             cfg.max_actuator_width, cfg.min_actuator_width, cfg.max_joint_width, cfg.min_joint_width,
                             len(cfg.collision_geoms), len(cfg.collision_geoms_fingers), len(cfg.ignored_collision_geoms),
                             cg.ctypes.data_as(i32p), cf.ctypes.data_as(i32p), ig.ctypes.data_as(i32p))
        _lib.check(self._L.rcsh_sim_add_gripper(sim._h, C.byref(d)))

    @property
    def n_envs(self) -> int:
        return self.sim.n_envs

    def get_config(self) -> SimGripperConfig:
        return self._cfg

    def get_state(self) -> SimGripperState:
        n = self.n_envs
        lc, lw = np.zeros(n), np.zeros(n)
        mv, col = np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.uint8)
        _lib.check(self._L.rcsh_gripper_get_state(self.sim._h, _lib.ptr(lc), _lib.ptr(mv), _lib.ptr(lw), _lib.ptr(col)))
        return SimGripperState(lc, mv.astype(bool), lw, col.astype(bool))

    def set_normalized_width(self, width, force: float = 0.0, mask=None) -> None:
        w = np.ascontiguousarray(np.broadcast_to(np.asarray(width, dtype=np.float64), (self.n_envs,)))
        _lib.check(self._L.rcsh_gripper_set_normalized_width(self.sim._h, _lib.ptr(w), float(force), _lib.ptr(_mask(mask, self.n_envs))))

    def get_normalized_width(self) -> np.ndarray:
        out = np.zeros(self.n_envs)
        _lib.check(self._L.rcsh_gripper_get_normalized_width(self.sim._h, _lib.ptr(out)))
        return out

    def is_grasped(self) -> np.ndarray:
        out = np.zeros(self.n_envs, dtype=np.uint8)
        _lib.check(self._L.rcsh_gripper_is_grasped(self.sim._h, _lib.ptr(out)))
        return out.astype(bool)

    def grasp(self, mask=None) -> None:
        self.shut(mask)

    def open(self, mask=None) -> None:
        self.set_normalized_width(1.0, mask=mask)

    def shut(self, mask=None) -> None:
        self.set_normalized_width(0.0, mask=mask)

    def reset(self, mask=None) -> None:
        _lib.check(self._L.rcsh_gripper_reset(self.sim._h, _lib.ptr(_mask(mask, self.n_envs))))

    def close(self) -> None:
        pass
