"""ctypes binding of librcs_hip.so (the C-ABI declared in include/rcs_hip.h).

The library is the only execution path: there is no CPU fallback.  If the shared
object is missing, or no gfx950 device is visible when a Sim is created, the
error is raised to the caller.
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RCSH_LIB") or os.path.join(_HERE, "librcs_hip.so")  # (RCSH_LIB: a development build, for A/B measurements)

RCSH_OK, RCSH_ERR_ARG, RCSH_ERR_NAME, RCSH_ERR_MODEL, RCSH_ERR_DEVICE, RCSH_ERR_STATE = range(6)

_I32P = C.POINTER(C.c_int32)
_F64P = C.POINTER(C.c_double)
_U8P = C.POINTER(C.c_uint8)
_F32P = C.POINTER(C.c_float)


class ModelDesc(C.Structure):
    _INT_FIELDS = (
        "body_parentid", "body_jntadr", "body_jntnum", "jnt_type", "jnt_bodyid", "jnt_limited", "jnt_actfrclimited",
        "jnt_actgravcomp", "tendon_adr", "tendon_num", "wrap_objid", "eq_obj1id", "eq_obj2id", "eq_active0",
        "actuator_trntype", "actuator_trnid", "actuator_biastype", "actuator_ctrllimited", "actuator_forcelimited",
        "site_bodyid",
    )
    _fields_ = [
        ("nbody", C.c_int32), ("njnt", C.c_int32), ("nu", C.c_int32), ("ntendon", C.c_int32), ("nwrap", C.c_int32),
        ("neq", C.c_int32), ("nsite", C.c_int32),
        ("timestep", C.c_double), ("gravity", C.c_double * 3),
        ("body_parentid", _I32P), ("body_jntadr", _I32P), ("body_jntnum", _I32P),
        ("body_pos", _F64P), ("body_quat", _F64P), ("body_ipos", _F64P), ("body_iquat", _F64P),
        ("body_mass", _F64P), ("body_inertia", _F64P), ("body_gravcomp", _F64P),
        ("jnt_type", _I32P), ("jnt_bodyid", _I32P), ("jnt_pos", _F64P), ("jnt_axis", _F64P),
        ("jnt_limited", _I32P), ("jnt_range", _F64P), ("jnt_margin", _F64P), ("jnt_solref", _F64P), ("jnt_solimp", _F64P),
        ("jnt_actfrclimited", _I32P), ("jnt_actfrcrange", _F64P), ("jnt_actgravcomp", _I32P),
        ("dof_armature", _F64P), ("dof_damping", _F64P), ("dof_frictionloss", _F64P), ("qpos0", _F64P),
        ("tendon_adr", _I32P), ("tendon_num", _I32P), ("wrap_objid", _I32P), ("wrap_prm", _F64P),
        ("eq_obj1id", _I32P), ("eq_obj2id", _I32P), ("eq_active0", _I32P),
        ("eq_data", _F64P), ("eq_solref", _F64P), ("eq_solimp", _F64P),
        ("actuator_trntype", _I32P), ("actuator_trnid", _I32P), ("actuator_gear", _F64P),
        ("actuator_gainprm", _F64P), ("actuator_biasprm", _F64P), ("actuator_biastype", _I32P),
        ("actuator_ctrllimited", _I32P), ("actuator_ctrlrange", _F64P),
        ("actuator_forcelimited", _I32P), ("actuator_forcerange", _F64P),
        ("site_bodyid", _I32P), ("site_pos", _F64P), ("site_quat", _F64P),
        ("ngeom", C.c_int32), ("nmeshvert", C.c_int32),
        ("geom_type", _I32P), ("geom_bodyid", _I32P), ("geom_contype", _I32P), ("geom_conaffinity", _I32P),
        ("geom_pos", _F64P), ("geom_quat", _F64P), ("geom_size", _F64P),
        ("geom_vertadr", _I32P), ("geom_vertnum", _I32P), ("mesh_vert", _F64P),
     ("dof_solref", _F64P), ("dof_solimp", _F64P), ("geom_friction", _F64P),
    ]


class FreeBoxDesc(C.Structure):
    _fields_ = [
        ("qpos0", C.c_double * 7), ("mass", C.c_double), ("inertia", C.c_double * 3), ("size", C.c_double * 3),
        ("friction", C.c_double * 3), ("solref", C.c_double * 2), ("solimp", C.c_double * 5), ("plane_z", C.c_double),
        ("impratio", C.c_double), ("noslip_tolerance", C.c_double), ("noslip_iterations", C.c_int32),
        ("cone_elliptic", C.c_int32), ("geom_friction", C.c_double * 3), ("floor_friction", C.c_double * 3),
        ("resolve_robot_contacts", C.c_int32), ("reserved", C.c_int32),
    ]


class ContactOptions(C.Structure):
    _fields_ = [
        ("impratio", C.c_double), ("noslip_tolerance", C.c_double), ("noslip_iterations", C.c_int32), ("cone_elliptic", C.c_int32),
        ("solref", C.c_double * 2), ("solimp", C.c_double * 5), ("resolve_robot_contacts", C.c_int32), ("reserved", C.c_int32),
    ]


def make_contact_options(cm, resolve_robot_contacts: bool = True) -> ContactOptions:
    """rcsh_contact_options of a compiled scene: mjModel.opt's solver options + the default contact solref / solimp."""
    o = ContactOptions()
    o.impratio, o.noslip_tolerance, o.noslip_iterations = float(cm.impratio), 1e-6, int(cm.noslip_iterations)
    o.cone_elliptic = int(cm.cone == "elliptic")
    o.solref[:] = [0.02, 1.0]
    o.solimp[:] = [0.9, 0.95, 0.001, 0.5, 2.0]
    o.resolve_robot_contacts = int(resolve_robot_contacts)
    return o


def make_free_box_desc(cm, resolve_robot_contacts: bool = True) -> FreeBoxDesc | None:
    """rcsh_free_box_desc of a compiled scene's free body (None: the scene has none)."""
    free = getattr(cm, "free_bodies", [])
    if not free:
        return None
    if len(free) > 1:
        raise RuntimeError("scenes with more than one free body are not supported")
    fb = free[0]
    d = FreeBoxDesc()
    for name in ("qpos0", "inertia", "size", "friction", "solref", "solimp"):
        getattr(d, name)[:] = [float(x) for x in fb[name]]
    d.mass, d.plane_z = float(fb["mass"]), float(fb["plane_z"])
    d.impratio, d.noslip_iterations = float(cm.impratio), int(cm.noslip_iterations)
    d.noslip_tolerance = 1e-6  # mjOption default; the MJCF subset has no attribute for it
    d.cone_elliptic = int(cm.cone == "elliptic")
    d.geom_friction[:] = [float(x) for x in fb.get("geom_friction", fb["friction"])]
    d.floor_friction[:] = [float(x) for x in fb.get("floor_friction", (1.0, 0.005, 0.0001))]
    d.resolve_robot_contacts = int(resolve_robot_contacts)
    return d


class RenderSceneDesc(C.Structure):
    _fields_ = [
        ("nshape", C.c_int32), ("nplanes", C.c_int32), ("shape", _I32P), ("link", _I32P), ("pos", _F64P), ("rot", _F64P),
        ("size", _F64P), ("plane_adr", _I32P), ("plane_num", _I32P), ("sphere", _F64P), ("planes", _F64P),
        ("znear", C.c_double), ("zfar", C.c_double),
    ]


class RenderColours(C.Structure):
    _fields_ = [("colour", _F64P), ("headlight_ambient", C.c_double * 3), ("headlight_diffuse", C.c_double * 3),
                ("light_dir", C.c_double * 3), ("light_diffuse", C.c_double * 3), ("sky_rgb1", C.c_double * 3), ("sky_rgb2", C.c_double * 3)]


class CameraDesc(C.Structure):
    _fields_ = [("link", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("pos", C.c_double * 3), ("rot", C.c_double * 9),
                ("fovy_deg", C.c_double)]


class PickTaskDesc(C.Structure):
    _fields_ = [("ee_home", C.c_double * 3), ("success_height", C.c_double)]


class RobotDesc(C.Structure):
    _fields_ = [
        ("dof", C.c_int32), ("joint_ids", _I32P), ("actuator_ids", _I32P),
        ("attachment_site", C.c_int32), ("base_body", C.c_int32), ("q_home", _F64P),
        ("tcp_offset", C.c_double * 7), ("joint_rotational_tolerance", C.c_double),
        ("seconds_between_callbacks", C.c_double), ("register_convergence_callback", C.c_int32),
        ("n_collision_geoms", C.c_int32), ("collision_geom_ids", _I32P),
    ]


class GripperDesc(C.Structure):
    _fields_ = [
        ("joint_id", C.c_int32), ("actuator_id", C.c_int32),
        ("epsilon_inner", C.c_double), ("epsilon_outer", C.c_double), ("seconds_between_callbacks", C.c_double),
        ("max_actuator_width", C.c_double), ("min_actuator_width", C.c_double),
        ("max_joint_width", C.c_double), ("min_joint_width", C.c_double),
        ("n_collision_geoms", C.c_int32), ("n_finger_geoms", C.c_int32), ("n_ignored_geoms", C.c_int32),
        ("collision_geom_ids", _I32P), ("finger_geom_ids", _I32P), ("ignored_geom_ids", _I32P),
    ]


class EnvDesc(C.Structure):
    _fields_ = [
        ("control_mode", C.c_int32), ("relative_to", C.c_int32), ("max_mov", C.c_double * 2),
        ("binary_gripper", C.c_int32), ("joint_low", _F64P), ("joint_high", _F64P),
    ]


# every symbol include/rcs_hip.h declares; load() fails if one is missing
EXPORTS = (
    "rcsh_last_error", "rcsh_abi_version", "rcsh_device_count", "rcsh_sim_create", "rcsh_sim_destroy",
    "rcsh_sim_num_envs", "rcsh_sim_synchronize", "rcsh_sim_stream", "rcsh_sim_set_stream", "rcsh_sim_wait_for", "rcsh_sim_set_kernel", "rcsh_sim_set_config", "rcsh_sim_get_config",
    "rcsh_sim_step", "rcsh_sim_step_until_convergence", "rcsh_sim_is_converged", "rcsh_sim_reset",
    "rcsh_sim_add_robot", "rcsh_robot_set_joint_position", "rcsh_robot_get_joint_position",
    "rcsh_robot_get_cartesian_position", "rcsh_robot_get_base_pose", "rcsh_robot_set_cartesian_position",
    "rcsh_robot_set_joints_hard", "rcsh_robot_reset", "rcsh_robot_move_home", "rcsh_robot_get_state",
    "rcsh_ik_inverse", "rcsh_ik_forward", "rcsh_sim_add_gripper", "rcsh_gripper_set_normalized_width",
    "rcsh_gripper_get_normalized_width", "rcsh_gripper_is_grasped", "rcsh_gripper_reset", "rcsh_gripper_get_state",
    "rcsh_sim_get_qpos", "rcsh_sim_get_qvel", "rcsh_sim_get_ctrl", "rcsh_sim_get_time", "rcsh_sim_set_qpos",
    "rcsh_sim_set_qvel", "rcsh_sim_add_free_box", "rcsh_sim_set_contact_options", "rcsh_sim_reset_free_box", "rcsh_sim_get_free_qpos", "rcsh_sim_get_free_qvel",
    "rcsh_sim_set_free_qpos", "rcsh_sim_set_free_qvel", "rcsh_sim_nq", "rcsh_sim_nu", "rcsh_sim_state_bytes", "rcsh_sim_get_state", "rcsh_sim_set_state", "rcsh_env_configure", "rcsh_env_obs_width",
    "rcsh_env_action_width", "rcsh_env_reset", "rcsh_env_step", "rcsh_env_reset_dev", "rcsh_env_step_dev",
    "rcsh_sim_set_render_scene", "rcsh_sim_add_camera", "rcsh_hull_edges", "rcsh_camera_render", "rcsh_camera_render_dev",
    "rcsh_sim_set_render_colours", "rcsh_camera_render_rgb", "rcsh_camera_render_rgb_dev", "rcsh_sim_set_render_f64",
    "rcsh_sim_set_render_schedule", "rcsh_render_pending", "rcsh_render_dropped", "rcsh_camera_render_snapshot",
    "rcsh_env_configure_pick_task", "rcsh_env_reset_task", "rcsh_env_step_task", "rcsh_env_reset_task_dev", "rcsh_env_step_task_dev",
    "rcsh_dev_alloc", "rcsh_dev_free", "rcsh_dev_upload", "rcsh_dev_download", "rcsh_prof_enable", "rcsh_prof_read",
    "rcsh_debug_dump_model",
    "rcsh_comm_get_unique_id", "rcsh_comm_init", "rcsh_comm_rank", "rcsh_env_allgather_obs_dev", "rcsh_comm_allgather_dev", "rcsh_comm_wait",
    "rcsh_comm_destroy", "rcsh_comm_copy_create", "rcsh_comm_copy_connect", "rcsh_comm_copy_recv_buffer",
    "rcsh_sim_contact_table_dropped",
    "rcsh_sim_contact_check_unchecked_pairs",
    "rcsh_sim_contact_unresolved",
    "rcsh_sim_set_contact_check",
    "rcsh_sim_contact_escalated",
)

_lib = None


def load() -> C.CDLL:
    """Load librcs_hip.so and check that every declared entry point is exported."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
            "rcs_amd has no CPU execution path."
        )
    L = C.CDLL(LIB_PATH)
    missing = [s for s in EXPORTS if not hasattr(L, s)]
    if missing:
        raise ImportError(f"librcs_hip.so does not export: {missing}")
    L.rcsh_last_error.restype = C.c_char_p
    L.rcsh_sim_stream.restype = C.c_void_p
    L.rcsh_sim_stream.argtypes = [C.c_void_p]
    L.rcsh_sim_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.rcsh_sim_wait_for.argtypes = [C.c_void_p, C.c_void_p]
    L.rcsh_comm_get_unique_id.argtypes = [C.c_char_p]
    L.rcsh_comm_init.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32]
    L.rcsh_env_allgather_obs_dev.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    L.rcsh_comm_allgather_dev.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t]
    L.rcsh_comm_wait.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    L.rcsh_comm_destroy.argtypes = [C.c_void_p]
    L.rcsh_comm_copy_create.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_size_t, C.c_char_p]
    L.rcsh_comm_copy_connect.argtypes = [C.c_void_p, C.c_char_p]
    L.rcsh_comm_copy_recv_buffer.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
    L.rcsh_sim_set_kernel.argtypes = [C.c_void_p, C.c_int32]
    L.rcsh_sim_state_bytes.restype = C.c_size_t
    L.rcsh_sim_state_bytes.argtypes = [C.c_void_p]
    L.rcsh_sim_get_state.argtypes = [C.c_void_p, C.c_void_p]
    L.rcsh_sim_set_state.argtypes = [C.c_void_p, C.c_void_p]
    L.rcsh_sim_create.argtypes = [C.POINTER(ModelDesc), C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    L.rcsh_sim_add_free_box.argtypes = [C.c_void_p, C.POINTER(FreeBoxDesc)]
    L.rcsh_sim_set_contact_options.argtypes = [C.c_void_p, C.POINTER(ContactOptions)]
    L.rcsh_sim_contact_escalated.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.rcsh_sim_reset_free_box.argtypes = [C.c_void_p]
    L.rcsh_sim_contact_check_unchecked_pairs.argtypes = [C.c_void_p, _I32P]
    L.rcsh_sim_contact_table_dropped.argtypes = [C.c_void_p, _I32P, C.c_int32, _I32P, C.c_char_p, C.c_size_t]
    for fn in (L.rcsh_sim_get_free_qpos, L.rcsh_sim_get_free_qvel):
        fn.argtypes = [C.c_void_p, C.c_void_p]
    for fn in (L.rcsh_sim_set_free_qpos, L.rcsh_sim_set_free_qvel):
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.rcsh_sim_set_render_scene.argtypes = [C.c_void_p, C.POINTER(RenderSceneDesc)]
    L.rcsh_hull_edges.argtypes = [_F64P, C.c_int32, C.c_int32, _I32P, _F64P, C.POINTER(C.c_int32), _F64P]
    L.rcsh_sim_add_camera.argtypes = [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(C.c_int32)]
    L.rcsh_camera_render.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.rcsh_camera_render_dev.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.rcsh_sim_set_render_schedule.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
    L.rcsh_render_pending.argtypes = [C.c_void_p, C.c_void_p]
    L.rcsh_render_dropped.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    L.rcsh_camera_render_snapshot.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.rcsh_sim_set_render_colours.argtypes = [C.c_void_p, C.POINTER(RenderColours)]
    L.rcsh_camera_render_rgb.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.rcsh_sim_set_render_f64.argtypes = [C.c_void_p, C.c_int32]
    L.rcsh_camera_render_rgb_dev.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.rcsh_env_configure_pick_task.argtypes = [C.c_void_p, C.POINTER(PickTaskDesc)]
    L.rcsh_env_reset_task.argtypes = [C.c_void_p] + [C.c_void_p] * 5
    L.rcsh_env_step_task.argtypes = [C.c_void_p] + [C.c_void_p] * 7
    L.rcsh_env_reset_task_dev.argtypes = [C.c_void_p] + [C.c_void_p] * 5
    L.rcsh_env_step_task_dev.argtypes = [C.c_void_p] + [C.c_void_p] * 7
    L.rcsh_sim_destroy.argtypes = [C.c_void_p]
    L.rcsh_sim_destroy.restype = None
    L.rcsh_sim_step.argtypes = [C.c_void_p, C.c_int64]
    L.rcsh_gripper_set_normalized_width.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
    L.rcsh_dev_alloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    L.rcsh_dev_free.argtypes = [C.c_void_p, C.c_void_p]
    L.rcsh_debug_dump_model.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.rcsh_dev_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.rcsh_dev_download.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    _lib = L
    return L


def check(rc: int) -> None:
    """Map C-ABI error codes onto the exception types the reference raises."""
    if rc == RCSH_OK:
        return
    msg = load().rcsh_last_error().decode()
    if rc == RCSH_ERR_ARG:
        raise ValueError(msg)
    if rc in (RCSH_ERR_NAME, RCSH_ERR_MODEL, RCSH_ERR_STATE, RCSH_ERR_DEVICE):
        raise RuntimeError(msg)
    raise RuntimeError(f"rcs_hip error {rc}: {msg}")


def ptr(a: np.ndarray | None):
    """void* of a C-contiguous numpy array (None -> NULL).  The returned object holds a reference to the array (numpy's
    `data_as` contract), so `ptr(temporary)` as a call argument keeps the temporary alive until the call has returned -- a bare
    `c_void_p(a.ctypes.data)` does not: the C side then reads freed memory (found by tests/test_dynamics_golden.py, round 3)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def make_model_desc(cm) -> tuple[ModelDesc, list[np.ndarray]]:
    """Marshal a compiled scene (rcs_amd.mjcf.Model) into rcsh_model_desc; returns keep-alive arrays too."""
    if cm.nq != cm.nv or cm.nv != cm.njnt:
        raise RuntimeError("scene has joints outside the supported set (hinge / slide)")
    d = ModelDesc()
    d.nbody, d.njnt, d.nu = cm.nbody, cm.njnt, cm.nu
    d.ntendon, d.nwrap, d.neq, d.nsite = cm.ntendon, cm.nwrap, cm.neq, cm.nsite
    d.ngeom, d.nmeshvert = cm.ngeom, int(cm.arrays["mesh_vert"].shape[0])
    d.timestep = cm.timestep
    d.gravity[:] = [float(x) for x in cm.gravity]
    keep: list[np.ndarray] = []
    for name, ctype in ModelDesc._fields_:
        if ctype not in (_I32P, _F64P):
            continue
        dtype = np.int32 if ctype is _I32P else np.float64
        arr = np.ascontiguousarray(cm.arrays[name], dtype=dtype).reshape(-1)
        if arr.size == 0:
            arr = np.zeros(1, dtype=dtype)
        keep.append(arr)
        setattr(d, name, arr.ctypes.data_as(ctype))
    return d, keep
