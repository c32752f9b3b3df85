"""Scene description for the depth ray-caster (csrc/render.h), built from a compiled scene.

The reference renders with MuJoCo's OpenGL rasteriser through ``SimCameraSet`` (src/sim/camera.cpp:86-140): every geom
of the visible groups (0-2: the floor, free objects, the robot's *visual* meshes) ends up in the depth buffer.  This
backend casts one ray per pixel against analytic shapes instead: the floor plane, boxes (the free cube), capsules (the
wrist camera's body, drawn as its collision capsule), and the convex hulls of the robot's *collision* meshes (``render_hulls.npz``; the visual OBJ meshes -- 59 files
for the FR3 -- are not used, so the robot's silhouette is that of its collision hulls, a few millimetres fatter).  The
camera model, the depth encoding and everything above the pixels follow the reference.

Geoms and cameras are expressed in the frame of the *link* they ride on (the body that carries the joint; welded
children fold into it), which is what the kernels track per environment.
"""

from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

from .mjcf import GEOM_BOX, GEOM_CAPSULE, GEOM_MESH, GEOM_PLANE, Model, find_data_file, quat_mul, quat_to_mat

LINK_WORLD, LINK_FREE_BODY = -1, -2
SHAPE_PLANE, SHAPE_BOX, SHAPE_HULL, SHAPE_CAPSULE = 0, 1, 2, 3


def _compose(pa, qa, pb, qb):
    """Pose b given in frame a, with a given in some frame: returns b in that frame (positions, wxyz quaternions)."""
    return pa + quat_to_mat(qa) @ pb, quat_mul(qa, qb)


def _pose_in_link(cm: Model, body: int, pos, quat):
    """(link, pos, quat): the pose relative to the link (jointed body) that `body` is welded to; link -1 = world."""
    A = cm.arrays
    weld = int(A["body_weldid"][body])
    p, q = np.asarray(pos, dtype=np.float64), np.asarray(quat, dtype=np.float64)
    b = body
    while b != weld:
        p, q = _compose(A["body_pos"][b], A["body_quat"][b], p, q)
        b = int(A["body_parentid"][b])
    if weld == 0:
        return LINK_WORLD, p, q
    return int(A["body_jntadr"][weld]), p, q


@dataclass
class RenderScene:
    """Flat tables of the ray-caster: one row per shape, hull planes concatenated."""

    shape: np.ndarray      # [ng] SHAPE_*
    link: np.ndarray       # [ng] link index, LINK_WORLD or LINK_FREE_BODY
    pos: np.ndarray        # [ng, 3] shape frame in its link's frame
    rot: np.ndarray        # [ng, 9] row-major
    size: np.ndarray       # [ng, 3] box half extents; hulls: half extents of the bounding box centred at sphere[:3]; capsules (axis z): (r, r, r + half length)
    plane_adr: np.ndarray  # [ng] first row of `planes` (hulls)
    plane_num: np.ndarray  # [ng]
    sphere: np.ndarray     # [ng, 4] bounding sphere: centre (shape frame), radius; radius < 0: unbounded (plane)
    planes: np.ndarray     # [np, 4] n . x <= d, shape frame
    names: list[str] = field(default_factory=list)
    znear: float = 0.01
    zfar: float = 50.0
    # colour frames: per shape rgb (3), second checker colour (3), edge of a checker square, checker flag; the lighting
    colour: np.ndarray = field(default_factory=lambda: np.zeros((0, 8)))
    headlight_ambient: np.ndarray = field(default_factory=lambda: np.zeros(3))
    headlight_diffuse: np.ndarray = field(default_factory=lambda: np.zeros(3))
    light_dir: np.ndarray = field(default_factory=lambda: np.array([0.0, 0.0, -1.0]))
    light_diffuse: np.ndarray = field(default_factory=lambda: np.zeros(3))
    sky_rgb1: np.ndarray = field(default_factory=lambda: np.zeros(3))
    sky_rgb2: np.ndarray = field(default_factory=lambda: np.zeros(3))


def build_render_scene(cm: Model, scene_dir: str) -> RenderScene:
    A = cm.arrays
    hull_file = find_data_file([scene_dir, *getattr(cm, "data_dirs", [])], "render_hulls.npz")
    hulls = dict(np.load(hull_file)) if hull_file else {}
    rows, planes, names, colours = [], [], [], []

    def colour_of(rgba, checker=None):
        if checker is not None:
            return [*checker[0], *checker[1], float(checker[2]), 1.0]
        return [*np.asarray(rgba, dtype=np.float64)[:3], 0.0, 0.0, 0.0, 1.0, 0.0]

    def add(shape, link, p, q, size=(0, 0, 0), pl=None, sphere=(0, 0, 0, -1.0), name="", colour=None):
        adr = sum(len(x) for x in planes)
        colours.append(colour if colour is not None else colour_of((0.5, 0.5, 0.5, 1.0)))
        if pl is not None:
            planes.append(pl)
        rows.append((shape, link, p, quat_to_mat(q).reshape(9), np.asarray(size, dtype=np.float64), adr, 0 if pl is None else len(pl),
                     np.asarray(sphere, dtype=np.float64)))
        names.append(name)

    for g in range(cm.ngeom):
        t = int(A["geom_type"][g])
        link, p, q = _pose_in_link(cm, int(A["geom_bodyid"][g]), A["geom_pos"][g], A["geom_quat"][g])
        gcol = colour_of(A["geom_rgba"][g], cm.geom_checker[g] if t == GEOM_PLANE else None)
        if t == GEOM_PLANE:
            add(SHAPE_PLANE, link, p, q, name=cm.geom_names[g], colour=gcol)
        elif t == GEOM_BOX:
            s = A["geom_size"][g]
            add(SHAPE_BOX, link, p, q, size=s, sphere=(0, 0, 0, float(np.linalg.norm(s))), name=cm.geom_names[g], colour=gcol)
        elif t == GEOM_CAPSULE:
            r, hl = float(A["geom_size"][g][0]), float(A["geom_size"][g][1])
            add(SHAPE_CAPSULE, link, p, q, size=(r, r, r + hl), sphere=(0, 0, 0, r + hl), name=cm.geom_names[g], colour=gcol)
        elif t == GEOM_MESH and cm.geom_mesh[g] in hulls:
            pl = hulls[cm.geom_mesh[g]]
            v = A["mesh_vert"][A["geom_vertadr"][g]: A["geom_vertadr"][g] + A["geom_vertnum"][g]]
            c = 0.5 * (v.min(axis=0) + v.max(axis=0))
            # bounding sphere about the centre of the hull's bounding box; `size` = half extents of that box
            add(SHAPE_HULL, link, p, q, size=0.5 * (v.max(axis=0) - v.min(axis=0)), pl=pl, sphere=(*c, float(np.linalg.norm(v - c, axis=1).max())),
                name=cm.geom_names[g], colour=gcol)
        # other geom types (and meshes without hull data) are not drawn
    for fb in getattr(cm, "free_bodies", []):
        s = fb["size"]
        add(SHAPE_BOX, LINK_FREE_BODY, np.zeros(3), np.array([1.0, 0, 0, 0]), size=s, sphere=(0, 0, 0, float(np.linalg.norm(s))), name=fb["geom_name"],
            colour=colour_of(fb["rgba"]))
    if cm.stat_extent is None:
        raise RuntimeError("rendering needs <statistic extent=...> in the scene (near / far clip planes are fractions of it)")
    col = lambda i, dt: np.ascontiguousarray(np.array([r[i] for r in rows], dtype=dt))  # noqa: E731
    return RenderScene(shape=col(0, np.int32), link=col(1, np.int32), pos=col(2, np.float64), rot=col(3, np.float64), size=col(4, np.float64),
                       plane_adr=col(5, np.int32), plane_num=col(6, np.int32), sphere=col(7, np.float64),
                       planes=np.ascontiguousarray(np.concatenate(planes) if planes else np.zeros((1, 4))), names=names,
                       znear=cm.vis_znear * cm.stat_extent, zfar=cm.vis_zfar * cm.stat_extent,
                       colour=np.ascontiguousarray(np.array(colours, dtype=np.float64).reshape(-1, 8)),
                       headlight_ambient=np.asarray(cm.vis_headlight_ambient, dtype=np.float64), headlight_diffuse=np.asarray(cm.vis_headlight_diffuse, dtype=np.float64),
                       light_dir=np.asarray(cm.lights[0][0] if cm.lights else (0.0, 0.0, -1.0), dtype=np.float64),
                       light_diffuse=np.asarray(cm.lights[0][1] if cm.lights else (0.0, 0.0, 0.0), dtype=np.float64),
                       sky_rgb1=np.asarray(cm.skybox[0] if cm.skybox else (0.0, 0.0, 0.0), dtype=np.float64),
                       sky_rgb2=np.asarray(cm.skybox[1] if cm.skybox else (0.0, 0.0, 0.0), dtype=np.float64))


def camera_in_link(cm: Model, name: str):
    """(link, pos, rot[9], fovy_deg) of the MJCF camera `name`."""
    cid = cm.name2id("cam", name)
    if cid < 0:
        raise RuntimeError(f"No camera named {name}")
    A = cm.arrays
    link, p, q = _pose_in_link(cm, int(A["cam_bodyid"][cid]), A["cam_pos"][cid], A["cam_quat"][cid])
    return link, p, quat_to_mat(q).reshape(9), float(A["cam_fovy"][cid])


def free_camera(cm: Model):
    """(link, pos, rot[9], fovy_deg) of an untouched mjvCamera of type mjCAMERA_FREE (what SimCameraSet builds for
    CameraType.free, camera.cpp:36-47: mjv_defaultCamera, then only `type` and `fixedcamid` are set): it looks at the world
    origin from 2 m away, azimuth 90, elevation -45 degrees."""
    az, el = np.deg2rad(90.0), np.deg2rad(-45.0)
    forward = np.array([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)])
    up = np.array([-np.sin(el) * np.cos(az), -np.sin(el) * np.sin(az), np.cos(el)])
    right = np.cross(forward, up)
    rot = np.stack([right, up, -forward], axis=1)
    return LINK_WORLD, -2.0 * forward, rot.reshape(9), float(cm.vis_fovy)


def default_free_camera(cm: Model):
    """(link, pos, rot[9], fovy_deg) of MuJoCo's default free camera (mjv_defaultFreeCamera + mjv_updateCamera): it looks
    at stat.center from 1.5 x stat.extent away, along the direction given by vis.global.azimuth / elevation.
    (MuJoCo's conventions restated from memory -- "verify": forward = (cos el cos az, cos el sin az, sin el),
    up = (-sin el cos az, -sin el sin az, cos el); the camera frame has -z forward and +y up.)"""
    if cm.stat_extent is None or cm.stat_center is None:
        raise RuntimeError("the default free camera needs <statistic center=... extent=...> in the scene")
    az, el = np.deg2rad(cm.vis_azimuth), np.deg2rad(cm.vis_elevation)
    forward = np.array([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)])
    up = np.array([-np.sin(el) * np.cos(az), -np.sin(el) * np.sin(az), np.cos(el)])
    right = np.cross(forward, up)
    rot = np.stack([right, up, -forward], axis=1)  # columns: camera x, y, z in the world
    pos = np.asarray(cm.stat_center) - 1.5 * cm.stat_extent * forward
    return LINK_WORLD, pos, rot.reshape(9), float(cm.vis_fovy)
