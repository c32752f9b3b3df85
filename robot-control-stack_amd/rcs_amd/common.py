"""Host-side mirror of ``rcs._core.common`` (reference src/pybind/rcs.cpp:186-347).

Same class / method / keyword names as the reference bindings: ``Pose``, ``RPY``,
``RobotType``, ``RobotPlatform``, ``RobotMetaConfig``, ``robots_meta_config``,
``RobotConfig``, ``FrankaHandTCPOffset``.  Single poses are host values (the
reference's are too); the batched device-side pose math lives in csrc/pose.h.
"""

from __future__ import annotations

import enum
import math
from dataclasses import dataclass, field

import numpy as np


class RobotType(enum.IntEnum):  # reference include/rcs/Robot.h:22
    FR3 = 0
    UR5e = 1
    SO101 = 2
    XArm7 = 3


class RobotPlatform(enum.IntEnum):  # Robot.h:23
    SIMULATION = 0
    HARDWARE = 1


@dataclass
class RobotMetaConfig:  # Robot.h:16-20
    q_home: np.ndarray
    dof: int
    joint_limits: np.ndarray  # [2, dof]: low row, high row


_META = {  # Robot.h:24-95
    RobotType.FR3: RobotMetaConfig(
        np.array([0.0, -math.pi / 4, 0.0, -3.0 * math.pi / 4, 0.0, math.pi / 2, math.pi / 4]),
        7,
        np.array([[-2.3093, -1.5133, -2.4937, -2.7478, -2.4800, 0.8521, -2.6895],
                  [2.3093, 1.5133, 2.4937, -0.4461, 2.4800, 4.2094, 2.6895]]),
    ),
    RobotType.UR5e: RobotMetaConfig(
        np.array([-0.4488354, -2.02711196, 1.64630026, -1.18999615, -1.57079762, -2.01963249]),
        6,
        np.array([[-2 * math.pi, -2 * math.pi, -math.pi, -2 * math.pi, -2 * math.pi, -2 * math.pi],
                  [2 * math.pi, 2 * math.pi, math.pi, 2 * math.pi, 2 * math.pi, 2 * math.pi]]),
    ),
    RobotType.XArm7: RobotMetaConfig(
        np.array([0, -45.0 / 180.0 * math.pi, 0, 15.0 / 180.0 * math.pi, 0, -25.0 / 180.0 * math.pi, 0]),
        7,
        np.array([[-2 * math.pi, -2.094395, -2 * math.pi, -3.92699, -2 * math.pi, -math.pi, -2 * math.pi],
                  [2 * math.pi, 2.059488, 2 * math.pi, 0.191986, 2 * math.pi, 1.692969, 2 * math.pi]]),
    ),
    RobotType.SO101: RobotMetaConfig(
        np.array([-9.40612320177057, -99.66130397967824, 99.9124726477024, 69.96996996996998, -9.095744680851055]),
        5,
        np.array([[-100.0] * 5, [100.0] * 5]),
    ),
}


def robots_meta_config(robot_type: RobotType) -> RobotMetaConfig:
    """``common.robots_meta_config(robot_type)`` (rcs.cpp:320-326)."""
    return _META[RobotType(robot_type)]


# Joint ranges [rad] of the simulated SO101 stand-in (scenes/so101_empty_world).  The reference's SO101 entry is in the
# units of the real arm's servo bus, normalised to -100 .. 100 per joint (extensions/rcs_so101 drives hardware only; the
# reference has no simulated SO101), which a simulation in radians cannot use as is.
SO101_SIM_JOINT_RANGES = np.array([[-1.91986, -1.74533, -1.69, -1.65806, -2.74385],
                                   [1.91986, 1.74533, 1.69, 1.65806, 2.84121]])


def sim_robots_meta_config(robot_type: RobotType) -> RobotMetaConfig:
    """`robots_meta_config` as a simulation reads it: identical for every robot whose entry is in radians; the SO101's
    normalised home pose and limits are mapped linearly onto the simulated joints' ranges (-100 -> lower, 100 -> upper)."""
    meta = robots_meta_config(robot_type)
    if RobotType(robot_type) != RobotType.SO101:
        return meta
    lo, hi = SO101_SIM_JOINT_RANGES
    to_rad = lambda x: lo + (np.asarray(x) + 100.0) / 200.0 * (hi - lo)  # noqa: E731
    return RobotMetaConfig(to_rad(meta.q_home), meta.dof, np.stack([to_rad(meta.joint_limits[0]), to_rad(meta.joint_limits[1])]))


def IdentityTranslation() -> np.ndarray:
    return np.zeros(3)


def IdentityRotMatrix() -> np.ndarray:
    return np.eye(3)


def IdentityRotQuatVec() -> np.ndarray:
    return np.array([0.0, 0.0, 0.0, 1.0])


def FrankaHandTCPOffset() -> np.ndarray:  # reference src/rcs/Pose.cpp:11-15
    return np.array([[0.707, 0.707, 0, 0], [-0.707, 0.707, 0, 0], [0, 0, 1, 0.1034], [0, 0, 0, 1]], dtype=np.float64)


# ---- quaternion helpers, coefficient order x y z w (reference Pose.cpp:115)


def _qmul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by + ay * bw + az * bx - ax * bz,
        aw * bz + az * bw + ax * by - ay * bx,
        aw * bw - ax * bx - ay * by - az * bz,
    ])


def _qnorm(q):
    q = np.asarray(q, dtype=np.float64)
    return q / math.sqrt(float(q @ q))


def _qrot(q, v):
    u = np.asarray(q[:3])
    uv = 2.0 * np.cross(u, v)
    return v + q[3] * uv + np.cross(u, uv)


def _q2m(q):
    x, y, z, w = q
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([[1 - (tyy + tzz), txy - twz, txz + twy],
                     [txy + twz, 1 - (txx + tzz), tyz - twx],
                     [txz - twy, tyz + twx, 1 - (txx + tyy)]])


def _m2q(m):
    m = np.asarray(m, dtype=np.float64)
    t = m[0, 0] + m[1, 1] + m[2, 2]
    q = np.zeros(4)
    if t > 0:
        t = math.sqrt(t + 1.0)
        q[3] = 0.5 * t
        t = 0.5 / t
        q[0] = (m[2, 1] - m[1, 2]) * t
        q[1] = (m[0, 2] - m[2, 0]) * t
        q[2] = (m[1, 0] - m[0, 1]) * t
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j = (i + 1) % 3
        k = (j + 1) % 3
        t = math.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
        q[i] = 0.5 * t
        t = 0.5 / t
        q[3] = (m[k, j] - m[j, k]) * t
        q[j] = (m[j, i] + m[i, j]) * t
        q[k] = (m[k, i] + m[i, k]) * t
    return q


def _angular_distance(a, b):
    d = _qmul(a, np.array([-b[0], -b[1], -b[2], b[3]]))
    return 2.0 * math.atan2(math.sqrt(d[0] ** 2 + d[1] ** 2 + d[2] ** 2), abs(d[3]))


def _slerp(a, t, b):
    one = 1.0 - np.finfo(np.float64).eps
    d = float(np.dot(a, b))
    if abs(d) >= one:
        s0, s1 = 1.0 - t, t
    else:
        theta = math.acos(abs(d))
        st = math.sin(theta)
        s0 = math.sin((1.0 - t) * theta) / st
        s1 = math.sin(t * theta) / st
    if d < 0:
        s1 = -s1
    return s0 * np.asarray(a) + s1 * np.asarray(b)


def _polar_rotation(m):
    """Orthogonal polar factor of a 3x3 matrix (what an affine transform's rotation() is)."""
    u, _, vt = np.linalg.svd(np.asarray(m, dtype=np.float64))
    r = u @ vt
    if np.linalg.det(r) < 0:
        u[:, -1] = -u[:, -1]
        r = u @ vt
    return r


class RPY:
    """Extrinsic x/y/z = roll/pitch/yaw (reference include/rcs/Pose.h:23-65)."""

    def __init__(self, roll: float = 0.0, pitch: float = 0.0, yaw: float = 0.0, rpy=None):
        if rpy is not None:
            roll, pitch, yaw = (float(x) for x in np.asarray(rpy).reshape(3))
        elif isinstance(roll, (np.ndarray, list, tuple)):
            roll, pitch, yaw = (float(x) for x in np.asarray(roll).reshape(3))
        self.roll, self.pitch, self.yaw = float(roll), float(pitch), float(yaw)

    def as_quaternion_vector(self) -> np.ndarray:
        qz = np.array([0, 0, math.sin(self.yaw / 2), math.cos(self.yaw / 2)])
        qy = np.array([0, math.sin(self.pitch / 2), 0, math.cos(self.pitch / 2)])
        qx = np.array([math.sin(self.roll / 2), 0, 0, math.cos(self.roll / 2)])
        return _qmul(_qmul(qz, qy), qx)

    def rotation_matrix(self) -> np.ndarray:
        return _q2m(self.as_quaternion_vector())

    def as_vector(self) -> np.ndarray:
        return np.array([self.roll, self.pitch, self.yaw])

    def is_close(self, other: "RPY", eps: float = 1e-8) -> bool:
        return float(np.abs(self.as_vector() - other.as_vector()).sum()) < eps

    def __add__(self, other: "RPY") -> "RPY":
        return RPY(self.roll + other.roll, self.pitch + other.pitch, self.yaw + other.yaw)

    def __str__(self) -> str:
        return f"RPY({self.roll:.6f}, {self.pitch:.6f}, {self.yaw:.6f})"


class Pose:
    """Immutable SE(3) value, API of ``rcs.common.Pose`` (rcs.cpp:247-288, src/rcs/Pose.cpp)."""

    __slots__ = ("_t", "_q")

    def __init__(self, *args, translation=None, quaternion=None, rpy_vector=None, rotation=None, pose_matrix=None,
                 rpy=None, pose=None):
        if len(args) == 1 and not any(x is not None for x in (translation, quaternion, rpy_vector, rotation, pose_matrix, rpy, pose)):
            a = args[0]
            if isinstance(a, Pose):
                pose = a
            elif isinstance(a, RPY):
                rpy = a
            else:
                a = np.asarray(a, dtype=np.float64)
                if a.shape == (4, 4):
                    pose_matrix = a
                elif a.shape == (3, 3):
                    rotation = a
                elif a.size == 4:
                    quaternion = a
                else:
                    translation = a
        self._t = np.zeros(3)
        self._q = np.array([0.0, 0.0, 0.0, 1.0])
        if pose is not None:
            self._t, self._q = pose._t.copy(), pose._q.copy()
            return
        if pose_matrix is not None:  # Pose.cpp:33-38
            m = np.asarray(pose_matrix, dtype=np.float64).reshape(4, 4)
            self._t = m[:3, 3].copy()
            self._q = _qnorm(_m2q(_polar_rotation(m[:3, :3])))
            return
        if translation is not None:
            self._t = np.asarray(translation, dtype=np.float64).reshape(3).copy()
        if rotation is not None:  # Pose.cpp:40-45,101-104
            self._q = _m2q(np.asarray(rotation, dtype=np.float64).reshape(3, 3))
            if translation is not None:
                self._q = _qnorm(self._q)
        elif quaternion is not None:  # Pose.cpp:47-52
            self._q = _qnorm(np.asarray(quaternion, dtype=np.float64).reshape(4))
        elif rpy_vector is not None:  # Pose.cpp:68-73
            self._q = _qnorm(RPY(rpy=rpy_vector).as_quaternion_vector())
        elif rpy is not None:  # Pose.cpp:61-66
            self._q = _qnorm(rpy.as_quaternion_vector())

    @classmethod
    def _raw(cls, q, t) -> "Pose":
        p = cls()
        p._q = _qnorm(q)
        p._t = np.asarray(t, dtype=np.float64).copy()
        return p

    def translation(self) -> np.ndarray:
        return self._t.copy()

    def rotation_q(self) -> np.ndarray:
        return self._q.copy()

    def rotation_m(self) -> np.ndarray:
        return _q2m(self._q)

    def pose_matrix(self) -> np.ndarray:
        m = np.eye(4)
        m[:3, :3] = _q2m(self._q)
        m[:3, 3] = self._t
        return m

    def rotation_rpy(self) -> RPY:  # Pose.cpp:133-138 (Euler extraction with yaw in [0, pi], quirk Q13)
        m = _q2m(self._q)
        yaw = math.atan2(m[1, 0], m[0, 0])
        c2 = math.sqrt(m[2, 2] * m[2, 2] + m[2, 1] * m[2, 1])
        if yaw < 0:
            yaw += math.pi
            pitch = math.atan2(-m[2, 0], -c2)
        else:
            pitch = math.atan2(-m[2, 0], c2)
        s1, c1 = math.sin(yaw), math.cos(yaw)
        roll = math.atan2(s1 * m[0, 2] - c1 * m[1, 2], c1 * m[1, 1] - s1 * m[0, 1])
        return RPY(roll, pitch, yaw)

    def xyzrpy(self) -> np.ndarray:
        return np.concatenate([self._t, self.rotation_rpy().as_vector()])

    def interpolate(self, dest_pose: "Pose", progress: float) -> "Pose":
        progress = min(progress, 1.0)
        return Pose._raw(_slerp(self._q, progress, dest_pose._q), self._t + (dest_pose._t - self._t) * progress)

    def inverse(self) -> "Pose":
        qc = np.array([-self._q[0], -self._q[1], -self._q[2], self._q[3]])
        return Pose._raw(qc, -_qrot(qc, self._t))

    def total_angle(self) -> float:
        return _angular_distance(self._q, IdentityRotQuatVec())

    def limit_rotation_angle(self, max_angle: float) -> "Pose":
        cur = self.total_angle()
        if cur > max_angle and max_angle >= 0:
            return Pose._raw(_slerp(IdentityRotQuatVec(), max_angle / cur, self._q), self._t)
        return Pose(pose=self)

    def limit_translation_length(self, max_length: float) -> "Pose":
        n = float(np.linalg.norm(self._t))
        if n > max_length and max_length >= 0:
            return Pose._raw(self._q, self._t / n * max_length)
        return Pose(pose=self)

    def is_close(self, other: "Pose", eps_r: float = 1e-8, eps_t: float = 1e-8) -> bool:
        return float(np.abs(self._t - other._t).sum()) < eps_t and _angular_distance(self._q, other._q) < eps_r

    def __mul__(self, other: "Pose") -> "Pose":
        return Pose._raw(_qmul(self._q, other._q), _qrot(self._q, other._t) + self._t)

    def __str__(self) -> str:
        r = self.rotation_rpy()
        return f"{self.pose_matrix()}\nroll: {r.roll}\tpitch: {r.pitch}\tyaw: {r.yaw}"

    def as_vec7(self) -> np.ndarray:
        """x y z qx qy qz qw -- the wire format of the C-ABI."""
        return np.concatenate([self._t, self._q])


@dataclass
class RobotConfig:  # Robot.h:97-104
    robot_type: RobotType = RobotType.FR3
    robot_platform: RobotPlatform = RobotPlatform.SIMULATION
    tcp_offset: Pose = field(default_factory=Pose)
    attachment_site: str = "attachment_site"
    kinematic_model_path: str = "assets/scenes/fr3_empty_world/robot.xml"
