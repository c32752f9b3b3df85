"""``rcs_hip``: the reference-side extension package of the MI355X batched simulation backend.

Laid out like the reference's hardware extensions (reference extensions/rcs_fr3: a Python package around a compiled pybind11
``_core``).  ``rcs_hip._core.sim`` exposes ``Sim`` / ``SimConfig`` / ``SimRobot`` / ``SimRobotConfig`` / ``SimRobotState`` /
``SimGripper`` / ``SimGripperConfig`` / ``SimGripperState`` / ``SimCameraSet`` / ``SimCameraConfig`` / ``FrameSet`` / ``CameraType`` with the reference's method names (python/rcs/_core/sim.pyi) and a
leading environment axis on every array.  Build: ``python __graft_entry__.py`` (g++ + pybind11, links ``librcs_hip.so``).
"""

from __future__ import annotations

import numpy as np


def model_tables(cm) -> dict:
    """The ``model`` argument of ``rcs_hip._core.sim.Sim``: mjModel-named tables of a compiled scene (``rcs_amd.mjcf.Model``;
    with MuJoCo present the same names can be read off an ``mjModel``)."""
    d = {k: np.ascontiguousarray(v) for k, v in cm.arrays.items()}
    for k in ("nbody", "njnt", "nu", "ntendon", "nwrap", "neq", "nsite", "ngeom"):
        d[k] = int(getattr(cm, k))
    d["nmeshvert"] = int(cm.arrays["mesh_vert"].shape[0])
    d["timestep"] = float(cm.timestep)
    d["gravity"] = np.asarray(cm.gravity, dtype=np.float64)
    d["names"] = {"joint": list(cm.jnt_names), "actuator": list(cm.actuator_names), "body": list(cm.body_names), "site": list(cm.site_names),
                  "geom": list(cm.geom_names)}
    return d


def free_box_tables(cm, resolve_robot_contacts: bool = True) -> dict | None:
    """The ``free_box`` argument: constants of the scene's free body (``None`` if it has none)."""
    free = getattr(cm, "free_bodies", [])
    if not free:
        return None
    fb = free[0]
    d = {k: np.asarray(fb[k], dtype=np.float64) for k in ("qpos0", "inertia", "size", "friction", "solref", "solimp")}
    d["geom_friction"] = np.asarray(fb.get("geom_friction", fb["friction"]), dtype=np.float64)
    d["floor_friction"] = np.asarray(fb.get("floor_friction", (1.0, 0.005, 0.0001)), dtype=np.float64)
    d.update(mass=float(fb["mass"]), plane_z=float(fb["plane_z"]), impratio=float(cm.impratio), noslip_tolerance=1e-6,
             noslip_iterations=int(cm.noslip_iterations), cone_elliptic=int(cm.cone == "elliptic"), resolve_robot_contacts=int(resolve_robot_contacts))
    return d


def render_tables(cm, scene_dir: str) -> dict:
    """The argument of ``Sim.set_render_scene``: the shapes the ray-caster draws, their colours and lights, and the pose of every
    camera a ``SimCameraConfig`` can name -- the scene's MJCF cameras by name, ``""`` for ``CameraType.default_free`` and
    ``"<free>"`` for ``CameraType.free`` (``rcs_amd.render``)."""
    from rcs_amd import render

    rs = render.build_render_scene(cm, scene_dir)
    d = {k: np.ascontiguousarray(getattr(rs, k)) for k in ("shape", "link", "pos", "rot", "size", "plane_adr", "plane_num", "sphere", "planes", "colour",
                                                           "headlight_ambient", "headlight_diffuse", "light_dir", "light_diffuse", "sky_rgb1", "sky_rgb2")}
    d.update(nshape=len(rs.shape), nplanes=len(rs.planes), znear=float(rs.znear), zfar=float(rs.zfar))
    cams = {name: render.camera_in_link(cm, name) for name in cm.cam_names}
    cams["<free>"] = render.free_camera(cm)
    if cm.stat_extent is not None and cm.stat_center is not None:
        cams[""] = render.default_free_camera(cm)
    d["cameras"] = {k: (int(v[0]), np.asarray(v[1], dtype=np.float64), np.asarray(v[2], dtype=np.float64), float(v[3])) for k, v in cams.items()}
    return d


def pin_tables(path: str, frame_id: str, device: int = 0, urdf: bool = False) -> dict:
    """What ``rcs_hip._core.common.Pin(path, frame_id, urdf)`` loads: the model tables of the file at `path` plus the ids of the
    serial chain that carries the frame `frame_id` -- its hinge joints from the root outwards, their actuators, the site, the root
    body the chain hangs on (the frame ``Pin`` works in; DESIGN.md section 5 on pinocchio's root-frame semantics).  ``urdf=True``
    (the reference's default, src/pybind/rcs.cpp:296-300): a URDF, whose links are frames by name (``rcs_amd.urdf``); an empty
    `frame_id` then means the chain's last link (what ``RoboticsLibraryIK`` calls operational frame 0).  Otherwise an MJCF file
    and `frame_id` names a site."""
    from rcs_amd.mjcf import compile_mjcf

    if urdf:
        from rcs_amd.urdf import compile_urdf

        cm, info = compile_urdf(path)
        if not frame_id:
            if len(info["leaves"]) != 1:
                raise RuntimeError(f"{path}: the URDF is not a serial chain (leaf links {info['leaves']}): name the frame")
            frame_id = info["leaves"][0]
    else:
        cm = compile_mjcf(path)
    site = cm.name2id("site", frame_id)
    if site < 0:
        raise RuntimeError(f"No {'link' if urdf else 'site'} named {frame_id}")
    parent = np.asarray(cm.arrays["body_parentid"])
    jnt_body = np.asarray(cm.arrays["jnt_bodyid"])
    jnt_type = np.asarray(cm.arrays["jnt_type"])
    chain, b = [], int(np.asarray(cm.arrays["site_bodyid"])[site])
    base = b
    while b > 0:
        js = [int(j) for j in np.nonzero(jnt_body == b)[0] if jnt_type[j] == 3]
        chain = js + chain
        if js:
            base = int(parent[b])
        b = int(parent[b])
    trn = np.asarray(cm.arrays["actuator_trnid"]).reshape(cm.nu, -1)[:, 0]
    trntype = np.asarray(cm.arrays["actuator_trntype"])
    acts = []
    for j in chain:
        a = [u for u in range(cm.nu) if trntype[u] == 0 and trn[u] == j]
        if not a:
            raise RuntimeError(f"joint {cm.jnt_names[j]} of the chain has no actuator: the backend's archetypes are actuated arms")
        acts.append(a[0])
    while base > 0 and len(np.nonzero(jnt_body == base)[0]) == 0 and parent[base] > 0:
        base = int(parent[base])  # the robot's root body: the first static body under the world
    return {"model": model_tables(cm), "joints": np.asarray(chain, dtype=np.int32), "actuators": np.asarray(acts, dtype=np.int32), "site": int(site),
            "base": int(base), "device": int(device)}
