"""URDF chain reader for the kinematics classes (``Pin(path, frame_id, urdf=True)``, ``RoboticsLibraryIK(urdf_path)``).

The reference's default ``Pin`` constructor parses a URDF with pinocchio (``pinocchio::urdf::buildModel``, reference
src/rcs/Kinematics.cpp:12-19; src/pybind/rcs.cpp:296-300: ``Pin(path, frame_id="fr3_link8", urdf=True)``) and the reference ships
``assets/fr3/urdf/fr3.urdf``.  This backend compiles ONE scene description, the MJCF subset of ``rcs_amd.mjcf``; a URDF is
therefore rewritten as MJCF text -- the kinematic tree with every ``<origin xyz rpy>`` as a body placement, ``<axis>`` / ``<limit>``
on revolute and prismatic joints, fixed joints as welded bodies -- and goes through the same compiler, finaliser and kernels as
any other scene.  Every link gets a site at its origin that carries the link's name, so that a link name is a valid ``frame_id``
(pinocchio adds a BODY frame per link).

What a URDF does not say is filled in so that the result is a complete model: links without ``<inertial>`` get a nominal 1 kg /
0.01 kg m^2 (kinematics reads none of it), every moving joint gets a position servo (the kernels' archetypes are actuated arms).
``<mimic>``, ``<transmission>``, meshes and collision geometry are ignored: the kinematics classes read the chain only.
"""

from __future__ import annotations

import math
import os
import xml.etree.ElementTree as ET


def _floats(text: str | None, n: int, default: float = 0.0) -> list[float]:
    if not text:
        return [default] * n
    v = [float(x) for x in text.split()]
    if len(v) != n:
        raise RuntimeError(f"URDF: expected {n} numbers, got {text!r}")
    return v


def _rpy_to_quat(rpy) -> list[float]:
    """URDF ``rpy``: rotations about the FIXED x, y, z axes in that order, R = Rz(yaw) Ry(pitch) Rx(roll); quaternion w x y z."""
    r, p, y = (0.5 * a for a in rpy)
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    return [cr * cp * cy + sr * sp * sy, sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy]


def _fmt(v) -> str:
    return " ".join(repr(float(x)) for x in v)


def urdf_to_mjcf(path: str) -> tuple[str, dict]:
    """MJCF text of the URDF at `path` and what was read: ``{"links": [...], "joints": [moving joints, tree order], "root": name,
    "leaves": [links without children]}``."""
    root = ET.parse(path).getroot()
    if root.tag != "robot":
        raise RuntimeError(f"{path}: not a URDF (root element <{root.tag}>)")
    links = {ln.get("name"): ln for ln in root.findall("link")}
    children: dict[str, list] = {name: [] for name in links}
    has_parent = set()
    for j in root.findall("joint"):
        parent, child = j.find("parent").get("link"), j.find("child").get("link")
        if parent not in links or child not in links:
            raise RuntimeError(f"URDF joint {j.get('name')}: unknown link")
        children[parent].append(j)
        has_parent.add(child)
    roots = [n for n in links if n not in has_parent]
    if len(roots) != 1:
        raise RuntimeError(f"{path}: a URDF has exactly one root link, found {roots}")
    moving: list[str] = []
    actuators: list[str] = []
    out: list[str] = []

    def inertial(ln, indent: str) -> str:
        ine = ln.find("inertial")
        if ine is None or ine.find("mass") is None:
            return f'{indent}<inertial pos="0 0 0" mass="1" diaginertia="0.01 0.01 0.01"/>'
        org = ine.find("origin")
        pos = _floats(org.get("xyz") if org is not None else None, 3)
        quat = _rpy_to_quat(_floats(org.get("rpy") if org is not None else None, 3))
        mass = float(ine.find("mass").get("value"))
        it = ine.find("inertia")
        if it is None:
            full = [0.01, 0.01, 0.01, 0.0, 0.0, 0.0]
        else:
            full = [float(it.get(k, "0")) for k in ("ixx", "iyy", "izz", "ixy", "ixz", "iyz")]
        return f'{indent}<inertial pos="{_fmt(pos)}" quat="{_fmt(quat)}" mass="{max(mass, 1e-6)!r}" fullinertia="{_fmt(full)}"/>'

    def quat_mul(a, b):
        return [a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]]

    def quat_rot(q, v):
        w, x, y, z = q
        return [(1 - 2 * (y * y + z * z)) * v[0] + 2 * (x * y - w * z) * v[1] + 2 * (x * z + w * y) * v[2],
                2 * (x * y + w * z) * v[0] + (1 - 2 * (x * x + z * z)) * v[1] + 2 * (y * z - w * x) * v[2],
                2 * (x * z - w * y) * v[0] + 2 * (y * z + w * x) * v[1] + (1 - 2 * (x * x + y * y)) * v[2]]

    def origin_of(joint):
        org = joint.find("origin") if joint is not None else None
        return _floats(org.get("xyz") if org is not None else None, 3), _rpy_to_quat(_floats(org.get("rpy") if org is not None else None, 3))

    def welded(name: str, pos, quat, indent: str) -> None:
        """Links hanging on `name` by FIXED joints ride on the same body: their frames become sites of it (placement composed),
        their own children hang on it with the composed placement.  (`pos`, `quat`: the link's frame in the body's.)"""
        out.append(f'{indent}<site name="{name}" pos="{_fmt(pos)}" quat="{_fmt(quat)}"/>')
        for j in children[name]:
            jp, jq = origin_of(j)
            cp = [a + b for a, b in zip(pos, quat_rot(quat, jp))]
            cq = quat_mul(quat, jq)
            child = j.find("child").get("link")
            if j.get("type") == "fixed":
                welded(child, cp, cq, indent)
            else:
                emit(child, j, cp, cq, indent)

    def emit(name: str, joint, pos, quat, indent: str) -> None:
        out.append(f'{indent}<body name="{name}" pos="{_fmt(pos)}" quat="{_fmt(quat)}">')
        out.append(inertial(links[name], indent + "  "))
        if joint is not None:
            jtype = joint.get("type")
            if jtype in ("revolute", "continuous", "prismatic"):
                axis = _floats(joint.find("axis").get("xyz") if joint.find("axis") is not None else "1 0 0", 3)
                lim = joint.find("limit")
                kind = "slide" if jtype == "prismatic" else "hinge"
                rng = ""
                lo, hi = -2 * math.pi, 2 * math.pi
                if jtype != "continuous" and lim is not None and lim.get("lower") is not None and lim.get("upper") is not None:
                    lo, hi = float(lim.get("lower")), float(lim.get("upper"))
                    rng = f' range="{lo!r} {hi!r}"'
                out.append(f'{indent}  <joint name="{joint.get("name")}" type="{kind}" axis="{_fmt(axis)}"{rng} armature="0.1"/>')
                moving.append(joint.get("name"))
                actuators.append(f'    <general name="act_{joint.get("name")}" joint="{joint.get("name")}" biastype="affine" gainprm="1000" biasprm="0 -1000 -100" '
                                 f'ctrlrange="{lo!r} {hi!r}"/>')
            else:
                raise RuntimeError(f"URDF joint {joint.get('name')}: type {jtype!r} is outside the supported set (revolute, continuous, prismatic, fixed)")
        welded(name, [0.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0], indent + "  ")
        out.append(f"{indent}</body>")

    emit(roots[0], None, [0.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0], "    ")
    text = "\n".join([
        f'<mujoco model="{root.get("name", "urdf")}">',
        '  <compiler angle="radian" autolimits="true"/>',
        '  <option integrator="implicitfast"/>',
        "  <worldbody>",
        *out,
        "  </worldbody>",
        "  <actuator>",
        *actuators,
        "  </actuator>",
        "</mujoco>",
        "",
    ])
    leaves = [n for n in links if not children[n]]
    return text, {"links": list(links), "joints": moving, "root": roots[0], "leaves": leaves}


def compile_urdf(path: str):
    """``rcs_amd.mjcf.Model`` of the URDF at `path` (through its MJCF rewrite) and the reader's summary."""
    import tempfile

    from .mjcf import compile_mjcf

    text, info = urdf_to_mjcf(path)
    with tempfile.TemporaryDirectory() as d:
        f = os.path.join(d, "urdf_as_mjcf.xml")
        with open(f, "w") as fh:
            fh.write(text)
        cm = compile_mjcf(f)
    return cm, info
