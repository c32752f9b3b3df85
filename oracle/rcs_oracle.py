"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE -- see oracle/rcs_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "librcs_oracle.so")

MAXBODY, MAXV, MAXU, MAXEQ, MAXTENDON, MAXWRAP, MAXSITE, MAXARM = 32, 16, 16, 4, 4, 8, 32, 8
MAXGEOM, MAXCON, MAXCGEOM = 32, 256, 16
MAXEFC = MAXEQ + 3 * MAXV + 3 * MAXCON
NVT = MAXV + 6
BODY_BOX = -2
D = C.c_double
I = C.c_int


def build(force: bool = False) -> str:
    """Compile oracle/*.c into oracle/_build/librcs_oracle.so (gcc, seconds)."""
    srcs = [os.path.join(_HERE, f) for f in ("rcs_physics.c", "rcs_pose_ik.c", "rcs_sim.c", "rcs_object.c", "rcs_contact.c", "rcs_oracle.h", "Makefile")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _SO


class OrcBox(C.Structure):
    _fields_ = [
        ("present", I), ("qpos0", D * 7), ("mass", D), ("inertia", D * 3), ("size", D * 3), ("friction", D * 3), ("geom_friction", D * 3),
        ("solref", D * 2), ("solimp", D * 5), ("plane_z", D), ("impratio", D), ("noslip_tolerance", D),
        ("noslip_iterations", I), ("nv_total", I), ("meaninertia", D),
    ]


class OrcBoxData(C.Structure):
    _fields_ = [
        ("qpos", D * 7), ("qvel", D * 6), ("qacc", D * 6), ("qacc_warmstart", D * 6),
        ("qfrc_smooth", D * 6), ("qacc_smooth", D * 6), ("xpos", D * 3), ("xquat", D * 4), ("ncon", I), ("zone", I * 4),
        ("con_dist", D * 4), ("con_pos", D * 3 * 4), ("con_mu", D * 4),
        ("J", D * 6 * 12), ("aref", D * 12), ("D", D * 12), ("R", D * 12), ("force", D * 12),
        ("newton_iter", I), ("noslip_iter", I),
    ]


class OrcModel(C.Structure):
    _fields_ = [
        ("nbody", I), ("njnt", I), ("nu", I), ("ntendon", I), ("nwrap", I), ("neq", I), ("nsite", I),
        ("timestep", D), ("gravity", D * 3),
        ("body_parentid", I * MAXBODY), ("body_rootid", I * MAXBODY), ("body_jntadr", I * MAXBODY),
        ("body_pos", D * 3 * MAXBODY), ("body_quat", D * 4 * MAXBODY), ("body_ipos", D * 3 * MAXBODY),
        ("body_iquat", D * 4 * MAXBODY), ("body_mass", D * MAXBODY), ("body_inertia", D * 3 * MAXBODY),
        ("body_gravcomp", D * MAXBODY),
        ("jnt_type", I * MAXV), ("jnt_bodyid", I * MAXV), ("jnt_pos", D * 3 * MAXV), ("jnt_axis", D * 3 * MAXV),
        ("jnt_limited", I * MAXV), ("jnt_range", D * 2 * MAXV), ("jnt_margin", D * MAXV),
        ("jnt_solref", D * 2 * MAXV), ("jnt_solimp", D * 5 * MAXV),
        ("jnt_actfrclimited", I * MAXV), ("jnt_actfrcrange", D * 2 * MAXV), ("jnt_actgravcomp", I * MAXV),
        ("dof_armature", D * MAXV), ("dof_damping", D * MAXV), ("qpos0", D * MAXV),
        ("tendon_adr", I * MAXTENDON), ("tendon_num", I * MAXTENDON), ("wrap_objid", I * MAXWRAP), ("wrap_prm", D * MAXWRAP),
        ("eq_obj1id", I * MAXEQ), ("eq_obj2id", I * MAXEQ), ("eq_active0", I * MAXEQ),
        ("eq_data", D * 5 * MAXEQ), ("eq_solref", D * 2 * MAXEQ), ("eq_solimp", D * 5 * MAXEQ),
        ("actuator_trntype", I * MAXU), ("actuator_trnid", I * MAXU), ("actuator_gear", D * MAXU),
        ("actuator_gainprm", D * 3 * MAXU), ("actuator_biasprm", D * 3 * MAXU), ("actuator_biastype", I * MAXU),
        ("actuator_ctrllimited", I * MAXU), ("actuator_ctrlrange", D * 2 * MAXU),
        ("actuator_forcelimited", I * MAXU), ("actuator_forcerange", D * 2 * MAXU),
        ("site_bodyid", I * MAXSITE), ("site_pos", D * 3 * MAXSITE), ("site_quat", D * 4 * MAXSITE),
        ("ngeom", I), ("geom_type", I * MAXGEOM), ("geom_bodyid", I * MAXGEOM),
        ("geom_contype", I * MAXGEOM), ("geom_conaffinity", I * MAXGEOM), ("geom_vertadr", I * MAXGEOM), ("geom_vertnum", I * MAXGEOM),
        ("geom_pos", D * 3 * MAXGEOM), ("geom_quat", D * 4 * MAXGEOM), ("geom_size", D * 3 * MAXGEOM),
        ("geom_friction", D * 3 * MAXGEOM),
        ("mesh_vert", C.POINTER(D)), ("body_weldid", I * MAXBODY), ("resolve_contacts", I),
        ("dof_invweight0", D * MAXV), ("body_invweight0", D * MAXBODY), ("geom_aabb", D * 6 * MAXGEOM), ("geom_rbound", D * MAXGEOM), ("geom_center", D * 3 * MAXGEOM),
        ("dof_frictionloss", D * MAXV), ("dof_solref", D * 2 * MAXV), ("dof_solimp", D * 5 * MAXV),
        ("box", OrcBox),
    ]


MAXSELF = 256


class OrcContact(C.Structure):
    _fields_ = [("geom", I * 2), ("body", I * 2), ("pos", D * 3), ("frame", D * 9), ("dist", D), ("mu", D), ("efc_address", I), ("zone", I)]


class OrcData(C.Structure):
    _fields_ = [
        ("time", D), ("qpos", D * MAXV), ("qvel", D * MAXV), ("ctrl", D * MAXU), ("qacc", D * MAXV), ("qacc_warmstart", D * MAXV),
        ("xpos", D * 3 * MAXBODY), ("xquat", D * 4 * MAXBODY), ("xmat", D * 9 * MAXBODY),
        ("xipos", D * 3 * MAXBODY), ("ximat", D * 9 * MAXBODY),
        ("xanchor", D * 3 * MAXV), ("xaxis", D * 3 * MAXV),
        ("site_xpos", D * 3 * MAXSITE), ("site_xmat", D * 9 * MAXSITE),
        ("subtree_com", D * 3 * MAXBODY), ("cinert", D * 10 * MAXBODY), ("cdof", D * 6 * MAXV),
        ("qM", D * MAXV * MAXV), ("ten_length", D * MAXTENDON), ("actuator_length", D * MAXU),
        ("cvel", D * 6 * MAXBODY), ("cdof_dot", D * 6 * MAXV), ("actuator_velocity", D * MAXU),
        ("qfrc_bias", D * MAXV), ("qfrc_passive", D * MAXV), ("qfrc_gravcomp", D * MAXV),
        ("actuator_force", D * MAXU), ("qfrc_actuator", D * MAXV), ("qfrc_smooth", D * MAXV), ("qacc_smooth", D * MAXV),
        ("nefc", I), ("ncon", I), ("efc_type", I * MAXEFC), ("efc_J", D * NVT * MAXEFC),
        ("efc_pos", D * MAXEFC), ("efc_margin", D * MAXEFC), ("efc_vel", D * MAXEFC),
        ("efc_D", D * MAXEFC), ("efc_aref", D * MAXEFC), ("efc_force", D * MAXEFC),
        ("efc_K", D * MAXEFC), ("efc_B", D * MAXEFC), ("efc_I", D * MAXEFC), ("efc_frictionloss", D * MAXEFC),
        ("efc_R", D * MAXEFC), ("efc_mu", D * MAXEFC),
        ("qfrc_constraint", D * NVT), ("solver_niter", I), ("noslip_niter", I), ("contact_geom", I * 2 * MAXCON),
        ("contact", OrcContact * MAXCON), ("coupled", I), ("nself", I), ("self_geom", I * 2 * MAXSELF),
        ("sep_n", I), ("sep_pair", I * 2 * 8), ("sep_dir", D * 3 * 8), ("self_depth", D * MAXSELF), ("pen_seen", D),
        ("box", OrcBoxData),
    ]


class OrcPose(C.Structure):
    _fields_ = [("t", D * 3), ("q", D * 4)]

    @staticmethod
    def identity() -> "OrcPose":
        p = OrcPose()
        p.q[3] = 1.0
        return p

    def translation(self) -> np.ndarray:
        return np.array(self.t[:])

    def rotation_q(self) -> np.ndarray:
        return np.array(self.q[:])


class OrcIk(C.Structure):
    _fields_ = [("m", C.POINTER(OrcModel)), ("site", I)]


class OrcSim(C.Structure):
    _fields_ = [
        ("m", C.POINTER(OrcModel)), ("d", OrcData),
        ("async_control", I), ("realtime", I), ("frequency", I), ("max_convergence_steps", I),
        ("convergence_steps", C.c_long), ("converged", I),
        ("has_robot", I), ("robot_conv_registered", I), ("has_gripper", I),
        ("cb_last", D * 2), ("any_last", D * 2), ("all_last", D * 2), ("any_ret", I * 2), ("all_ret", I * 2),
        ("arm_n", I), ("arm_jnt", I * MAXARM), ("arm_act", I * MAXARM), ("attachment_site", I), ("base_body", I),
        ("joint_rotational_tolerance", D), ("robot_period", D), ("tcp_offset", OrcPose), ("q_home", D * MAXARM),
        ("previous_angles", D * MAXARM), ("target_angles", D * MAXARM),
        ("ik_success", I), ("robot_collision", I), ("is_moving", I), ("is_arrived", I),
        ("ik", OrcIk), ("last_ik_iterations", I), ("arm_ncgeom", I), ("arm_cgeom", I * MAXCGEOM),
        ("grp_jnt", I), ("grp_act", I), ("grp_period", D),
        ("max_actuator_width", D), ("min_actuator_width", D), ("max_joint_width", D), ("min_joint_width", D),
        ("epsilon_inner", D), ("epsilon_outer", D), ("last_commanded_width", D), ("last_width", D),
        ("grp_is_moving", I), ("grp_collision", I),
        ("grp_ncgeom", I), ("grp_cgeom", I * MAXCGEOM), ("grp_ncfgeom", I), ("grp_cfgeom", I * MAXCGEOM),
        ("grp_nignored", I), ("grp_ignored", I * MAXCGEOM),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_sizeof_model.restype = C.c_ulong
        L.orc_sizeof_sim.restype = C.c_ulong
        assert L.orc_sizeof_model() == C.sizeof(OrcModel), (L.orc_sizeof_model(), C.sizeof(OrcModel))
        assert L.orc_sizeof_sim() == C.sizeof(OrcSim), (L.orc_sizeof_sim(), C.sizeof(OrcSim))
        L.orc_pose_total_angle.restype = D
        L.orc_gripper_get_normalized_width.restype = D
        L.orc_gripper_set_normalized_width.argtypes = [C.c_void_p, D, D]
        L.orc_pose_limit_rotation_angle.argtypes = [C.c_void_p, D, C.c_void_p]
        L.orc_pose_limit_translation_length.argtypes = [C.c_void_p, D, C.c_void_p]
        L.orc_pose_interpolate.argtypes = [C.c_void_p, C.c_void_p, D, C.c_void_p]
        L.orc_pose_is_close.argtypes = [C.c_void_p, C.c_void_p, D, D]
        L.orc_sim_step.argtypes = [C.c_void_p, C.c_long]
        _lib = L
    return _lib


def _fill(dst, src):
    a = np.ascontiguousarray(src)
    flat = a.reshape(-1)
    n = len(flat)
    if n == 0:
        return
    ptr = C.cast(dst, C.POINTER(D if a.dtype == np.float64 else I))
    for i in range(n):
        ptr[i] = flat[i].item()


def make_model(cm, resolve_contacts: bool = True) -> OrcModel:
    """Fill an ``orc_model`` from a compiled scene (``rcs_amd.mjcf.Model``) and run set0.  `resolve_contacts` False: contacts
    of robot geoms are detected (collision flags) but exert no force."""
    if cm.nq != cm.nv or cm.nq != cm.njnt:
        raise ValueError("oracle supports hinge/slide joints only")
    if cm.nbody > MAXBODY or cm.nv > MAXV or cm.nu > MAXU or cm.nsite > MAXSITE or cm.neq > MAXEQ or cm.ntendon > MAXTENDON or cm.nwrap > MAXWRAP:
        raise ValueError("scene exceeds oracle table sizes")
    m = OrcModel()
    m.nbody, m.njnt, m.nu = cm.nbody, cm.njnt, cm.nu
    m.ntendon, m.nwrap, m.neq, m.nsite = cm.ntendon, cm.nwrap, cm.neq, cm.nsite
    m.timestep = cm.timestep
    _fill(m.gravity, cm.gravity)
    f64 = lambda k: np.asarray(cm.arrays[k], dtype=np.float64)  # noqa: E731
    i32 = lambda k: np.asarray(cm.arrays[k], dtype=np.int32)  # noqa: E731
    for name in ("body_pos", "body_quat", "body_ipos", "body_iquat", "body_mass", "body_inertia", "body_gravcomp",
                 "jnt_pos", "jnt_axis", "jnt_range", "jnt_margin", "jnt_solref", "jnt_solimp", "jnt_actfrcrange",
                 "dof_armature", "dof_damping", "qpos0", "wrap_prm", "eq_data", "eq_solref", "eq_solimp",
                 "actuator_gear", "actuator_gainprm", "actuator_biasprm", "actuator_ctrlrange", "actuator_forcerange",
                 "site_pos", "site_quat"):
        _fill(getattr(m, name), f64(name))
    for name in ("body_rootid", "body_jntadr", "jnt_type", "jnt_bodyid", "jnt_limited", "jnt_actfrclimited",
                 "jnt_actgravcomp", "tendon_adr", "tendon_num", "wrap_objid", "eq_obj1id", "eq_obj2id", "eq_active0",
                 "actuator_trntype", "actuator_trnid", "actuator_biastype", "actuator_ctrllimited",
                 "actuator_forcelimited", "site_bodyid"):
        _fill(getattr(m, name), i32(name))
    _fill(m.body_parentid, i32("body_parentid"))
    _fill(m.body_weldid, i32("body_weldid"))
    if cm.ngeom > MAXGEOM:
        raise ValueError("scene exceeds oracle geom table size")
    m.ngeom = cm.ngeom
    for name in ("geom_type", "geom_bodyid", "geom_contype", "geom_conaffinity", "geom_vertadr", "geom_vertnum"):
        _fill(getattr(m, name), i32(name))
    for name in ("geom_pos", "geom_quat", "geom_size", "geom_friction"):
        _fill(getattr(m, name), f64(name))
    m.resolve_contacts = int(resolve_contacts)
    # solver options and the default contact parameters (every geom of the RCS scenes keeps the default solref / solimp)
    m.box.impratio = cm.impratio
    m.box.noslip_iterations = cm.noslip_iterations if cm.cone == "elliptic" else 0
    m.box.noslip_tolerance = 1e-6  # mjOption default
    _fill(m.box.solref, np.array([0.02, 1.0]))
    _fill(m.box.solimp, np.array([0.9, 0.95, 0.001, 0.5, 2.0]))
    verts = np.ascontiguousarray(cm.arrays["mesh_vert"], dtype=np.float64).reshape(-1)
    if verts.size == 0:
        verts = np.zeros(3)
    m._verts = verts  # keep alive
    m.mesh_vert = verts.ctypes.data_as(C.POINTER(D))
    for name in ("dof_frictionloss", "dof_solref", "dof_solimp"):
        _fill(getattr(m, name), f64(name))
    free = getattr(cm, "free_bodies", [])
    if len(free) > 1:
        raise ValueError("oracle supports one free box")
    if free:
        fb = free[0]
        m.box.present = 1
        _fill(m.box.geom_friction, np.asarray(fb.get("geom_friction", fb["friction"]), dtype=np.float64))
        for name in ("qpos0", "inertia", "size", "friction", "solref", "solimp"):
            _fill(getattr(m.box, name), np.asarray(fb[name], dtype=np.float64))
        m.box.mass, m.box.plane_z = fb["mass"], fb["plane_z"]
        m.box.impratio = cm.impratio
        m.box.noslip_iterations = cm.noslip_iterations if cm.cone == "elliptic" else 0
        m.box.noslip_tolerance = 1e-6  # mjOption default
        if cm.cone != "elliptic":
            raise ValueError("oracle contacts: elliptic cones only")
    lib().orc_set0(C.byref(m))
    return m


def _np(arr, n=None):
    a = np.ctypeslib.as_array(arr).copy()
    return a if n is None else a[:n]


class Pose:
    """Oracle-side ``rcs.common.Pose`` (xyzw quaternion), thin wrapper over orc_pose_*."""

    def __init__(self, translation=None, quaternion=None, rpy_vector=None, rotation=None, pose_matrix=None, _raw=None):
        L = lib()
        self.p = OrcPose.identity()
        if _raw is not None:
            self.p = _raw
            return
        t = (D * 3)(*(np.zeros(3) if translation is None else np.asarray(translation, dtype=np.float64).reshape(3)))
        if pose_matrix is not None:
            mm = (D * 16)(*np.asarray(pose_matrix, dtype=np.float64).reshape(16))
            L.orc_pose_from_matrix4(mm, C.byref(self.p))
        elif quaternion is not None:
            q = (D * 4)(*np.asarray(quaternion, dtype=np.float64).reshape(4))
            L.orc_pose_from_quat_t(q, t, C.byref(self.p))
        elif rpy_vector is not None:
            r = (D * 3)(*np.asarray(rpy_vector, dtype=np.float64).reshape(3))
            L.orc_pose_from_rpy_t(r, t, C.byref(self.p))
        elif rotation is not None:
            r9 = (D * 9)(*np.asarray(rotation, dtype=np.float64).reshape(9))
            L.orc_pose_from_rotm_t(r9, t, C.byref(self.p))
        elif translation is not None:
            self.p.t[:] = list(t)

    def translation(self):
        return self.p.translation()

    def rotation_q(self):
        return self.p.rotation_q()

    def rotation_m(self):
        r = (D * 9)()
        lib().orc_pose_rotation_m(C.byref(self.p), r)
        return np.array(r[:]).reshape(3, 3)

    def pose_matrix(self):
        r = (D * 16)()
        lib().orc_pose_matrix(C.byref(self.p), r)
        return np.array(r[:]).reshape(4, 4)

    def rotation_rpy(self):
        r = (D * 3)()
        lib().orc_pose_rpy(C.byref(self.p), r)
        return np.array(r[:])

    def xyzrpy(self):
        r = (D * 6)()
        lib().orc_pose_xyzrpy(C.byref(self.p), r)
        return np.array(r[:])

    def __mul__(self, other: "Pose") -> "Pose":
        out = OrcPose()
        lib().orc_pose_mul(C.byref(self.p), C.byref(other.p), C.byref(out))
        return Pose(_raw=out)

    def inverse(self) -> "Pose":
        out = OrcPose()
        lib().orc_pose_inverse(C.byref(self.p), C.byref(out))
        return Pose(_raw=out)

    def total_angle(self) -> float:
        return lib().orc_pose_total_angle(C.byref(self.p))

    def limit_rotation_angle(self, max_angle: float) -> "Pose":
        out = OrcPose()
        lib().orc_pose_limit_rotation_angle(C.byref(self.p), float(max_angle), C.byref(out))
        return Pose(_raw=out)

    def limit_translation_length(self, max_length: float) -> "Pose":
        out = OrcPose()
        lib().orc_pose_limit_translation_length(C.byref(self.p), float(max_length), C.byref(out))
        return Pose(_raw=out)

    def interpolate(self, dest: "Pose", progress: float) -> "Pose":
        out = OrcPose()
        lib().orc_pose_interpolate(C.byref(self.p), C.byref(dest.p), float(progress), C.byref(out))
        return Pose(_raw=out)

    def is_close(self, other: "Pose", eps_r: float = 1e-8, eps_t: float = 1e-8) -> bool:
        return bool(lib().orc_pose_is_close(C.byref(self.p), C.byref(other.p), float(eps_r), float(eps_t)))


def franka_hand_tcp_offset() -> Pose:
    out = OrcPose()
    lib().orc_franka_hand_tcp_offset(C.byref(out))
    return Pose(_raw=out)


# Contacts of the robot's geoms enter the constraint solve (False: they only raise the collision flags).  None = the HIP
# backend's default (resolves_contacts below); an explicit bool mirrors rcs_amd.sim.Sim(resolve_robot_contacts=...).
DEFAULT_RESOLVE_CONTACTS = None


def can_resolve_contacts(cm) -> bool:
    """What the HIP backend can resolve: the 7-dof arm + two-finger gripper archetype with elliptic cones -- FR3 + hand, and
    (round 3) the same archetype with dry joint friction (xArm7 + gripper) where the scene asks for no noslip pass (the
    friction-dof rows would take part in it)."""
    if cm.cone != "elliptic":
        return False
    if np.any(np.asarray(cm.arrays["dof_frictionloss"]) > 0) and cm.noslip_iterations > 0:
        return False
    return bool(cm.njnt == 9 and cm.nu == 8 and cm.ngeom > 1)


def resolves_contacts(cm) -> int:
    """The backend's DEFAULT (orc_model.resolve_contacts: bit 0 robot <-> floor / free body, bit 1 robot <-> robot): every contact
    resolved, as MuJoCo does, in scenes without a free body (round 5: environment by environment on the HIP side); in scenes with
    a free body robot <-> floor / free body (the whole batch on the contact-resolving kernels); nothing where the archetype cannot."""
    if not can_resolve_contacts(cm):
        return 0
    if getattr(cm, "free_bodies", []):
        return 1
    return 3 if not np.any(np.asarray(cm.arrays["dof_frictionloss"]) > 0) else 0


class Sim:
    """One oracle environment: Sim + SimRobot (+ SimGripper), restating the reference objects."""

    def __init__(self, cm, robot_joints, robot_actuators, attachment_site, base, q_home, tcp_offset: Pose | None = None,
                 gripper_joint: str | None = None, gripper_actuator: str | None = None,
                 register_convergence_callback: bool = True, idx: str = "0", arm_collision_geoms: list[str] | None = None,
                 resolve_contacts: bool | None = None, gripper_cfg: dict | None = None):
        """`gripper_cfg`: SimGripperConfig fields that differ from the reference's defaults (SimGripper.h:15-45):
        `max_joint_width`, `collision_geoms`, `collision_geoms_fingers` (full geom names)."""
        L = lib()
        self.cm = cm
        if resolve_contacts is None:
            resolve_contacts = resolves_contacts(cm) if DEFAULT_RESOLVE_CONTACTS is None else DEFAULT_RESOLVE_CONTACTS
        if resolve_contacts is True:  # "everything the backend resolves in this scene" (rcs_amd.sim.Sim(resolve_robot_contacts=True))
            resolve_contacts = resolves_contacts(cm) or 1
        self.model = make_model(cm, resolve_contacts)
        self.s = OrcSim()
        L.orc_sim_init(C.byref(self.s), C.byref(self.model))
        n = len(robot_joints)
        j = (I * n)(*[self._id("jnt", x, "joint") for x in robot_joints])
        a = (I * n)(*[self._id("actuator", x, "actuator") for x in robot_actuators])
        site = self._id("site", attachment_site, "site")
        base_id = self._id("body", base, "body")
        qh = (D * n)(*q_home)
        off = (tcp_offset or Pose()).p
        L.orc_sim_add_robot(C.byref(self.s), n, j, a, site, base_id, qh, C.byref(off), int(register_convergence_callback))
        if gripper_joint is not None:
            L.orc_sim_add_gripper(C.byref(self.s), self._id("jnt", gripper_joint, "joint"),
                                  self._id("actuator", gripper_actuator, "actuator"))
            if gripper_cfg and "max_joint_width" in gripper_cfg:
                self.s.max_joint_width = float(gripper_cfg["max_joint_width"])
                L.orc_gripper_reset(C.byref(self.s))
        self.n = n
        # SimRobotConfig.arm_collision_geoms / SimGripperConfig collision geom lists (SimRobot.h:19-22, SimGripper.h:24-29)
        if arm_collision_geoms is None:
            arm_collision_geoms = [f"fr3_link{i}_collision_{idx}" for i in range(8)]
        arm_g = [self._id("geom", g, "geom") for g in arm_collision_geoms]
        L.orc_sim_set_robot_cgeoms(C.byref(self.s), len(arm_g), (I * max(len(arm_g), 1))(*arm_g))
        if gripper_joint is not None:
            names = (gripper_cfg or {}).get("collision_geoms", [f"{g}_{idx}" for g in ("hand_c", "d435i_collision", "finger_0_left", "finger_0_right")])
            fnames = (gripper_cfg or {}).get("collision_geoms_fingers", [f"{g}_{idx}" for g in ("finger_0_left", "finger_0_right")])
            cg = [self._id("geom", g, "geom") for g in names]
            cf = [self._id("geom", g, "geom") for g in fnames]
            L.orc_sim_set_gripper_cgeoms(C.byref(self.s), len(cg), (I * max(len(cg), 1))(*cg), len(cf), (I * max(len(cf), 1))(*cf), 0, (I * 1)(0))

    def _id(self, kind, name, label):
        i = self.cm.name2id(kind, name)
        if i < 0:
            raise RuntimeError(f"No {label} named {name}")
        return i

    # Sim
    def set_config(self, async_control=False, realtime=False, frequency=30, max_convergence_steps=500):
        self.s.async_control, self.s.realtime = int(async_control), int(realtime)
        self.s.frequency, self.s.max_convergence_steps = frequency, max_convergence_steps

    def step(self, k: int):
        lib().orc_sim_step(C.byref(self.s), int(k))

    def step_until_convergence(self):
        lib().orc_sim_step_until_convergence(C.byref(self.s))

    def is_converged(self) -> bool:
        return bool(self.s.converged)

    def reset(self):
        lib().orc_sim_reset(C.byref(self.s))

    # mjData.joint("box_joint").qpos / .qvel of the scene's free box
    @property
    def box_qpos(self) -> np.ndarray:
        return np.array(self.s.d.box.qpos[:])

    @box_qpos.setter
    def box_qpos(self, q):
        self.s.d.box.qpos[:] = [float(x) for x in q]

    @property
    def box_qvel(self) -> np.ndarray:
        return np.array(self.s.d.box.qvel[:])

    @box_qvel.setter
    def box_qvel(self, v):
        self.s.d.box.qvel[:] = [float(x) for x in v]

    # SimRobot
    def set_joint_position(self, q):
        lib().orc_robot_set_joint_position(C.byref(self.s), (D * self.n)(*np.asarray(q, dtype=np.float64)[: self.n]))

    def get_joint_position(self) -> np.ndarray:
        q = (D * self.n)()
        lib().orc_robot_get_joint_position(C.byref(self.s), q)
        return np.array(q[:])

    def get_cartesian_position(self) -> Pose:
        out = OrcPose()
        lib().orc_robot_get_cartesian_position(C.byref(self.s), C.byref(out))
        return Pose(_raw=out)

    def get_base_pose(self) -> Pose:
        out = OrcPose()
        lib().orc_robot_get_base_pose(C.byref(self.s), C.byref(out))
        return Pose(_raw=out)

    def set_cartesian_position(self, pose: Pose):
        lib().orc_robot_set_cartesian_position(C.byref(self.s), C.byref(pose.p))

    # common.Kinematics (Pin) on the robot's chain
    def ik_inverse(self, pose: Pose, q0, tcp_offset: Pose | None = None):
        """Pin::inverse: (q[model.nq] or None, iterations)."""
        q0 = np.ascontiguousarray(q0, dtype=np.float64)
        out = (D * MAXV)()
        it = I(0)
        off = (tcp_offset or Pose()).p
        ok = lib().orc_ik_inverse(C.byref(self.s.ik), C.byref(pose.p), (D * len(q0))(*q0), len(q0), C.byref(off), out, C.byref(it))
        return (np.array(out[: self.model.njnt]) if ok else None), int(it.value)

    def ik_forward(self, q0, tcp_offset: Pose | None = None) -> Pose:
        q0 = np.ascontiguousarray(q0, dtype=np.float64)
        out = OrcPose()
        off = (tcp_offset or Pose()).p
        lib().orc_ik_forward(C.byref(self.s.ik), (D * len(q0))(*q0), len(q0), C.byref(off), C.byref(out))
        return Pose(_raw=out)

    def set_joints_hard(self, q):
        lib().orc_robot_set_joints_hard(C.byref(self.s), (D * self.n)(*np.asarray(q, dtype=np.float64)))

    def robot_reset(self):
        lib().orc_robot_reset(C.byref(self.s))

    def move_home(self):
        lib().orc_robot_move_home(C.byref(self.s))

    # SimGripper
    def gripper_set_normalized_width(self, w: float, force: float = 0.0):
        if lib().orc_gripper_set_normalized_width(C.byref(self.s), float(w), float(force)):
            raise ValueError("width must be between 0 and 1, force must be positive")

    def gripper_get_normalized_width(self) -> float:
        return lib().orc_gripper_get_normalized_width(C.byref(self.s))

    def gripper_is_grasped(self) -> bool:
        return bool(lib().orc_gripper_is_grasped(C.byref(self.s)))

    def gripper_reset(self):
        lib().orc_gripper_reset(C.byref(self.s))

    def gripper_grasp(self):
        self.gripper_set_normalized_width(0.0)

    def gripper_open(self):
        self.gripper_set_normalized_width(1.0)

    # raw state views
    @property
    def qpos(self):
        return _np(self.s.d.qpos, self.model.njnt)

    @property
    def qvel(self):
        return _np(self.s.d.qvel, self.model.njnt)

    @property
    def ctrl(self):
        return _np(self.s.d.ctrl, self.model.nu)

    @property
    def time(self):
        return self.s.d.time
