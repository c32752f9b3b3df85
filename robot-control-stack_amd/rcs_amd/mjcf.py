"""MJCF-subset compiler: scene XML -> flat constant tables.

Stands where the reference calls ``mujoco.MjModel.from_xml_path`` (reference:
python/rcs/sim/sim.py:44-55).  MuJoCo itself is a third-party dependency of the
reference (mujoco==3.2.6, pyproject.toml:23) that is not available here, so the
subset of its model compiler that the RCS scenes exercise is restated from the
published MJCF semantics:

* ``<include>``, ``<compiler angle/eulerseq/autolimits>``, ``<option>``
* nested ``<default>`` classes, ``childclass``, per-element ``class``; one dummy
  actuator per class that every actuator shortcut writes into
* bodies (pos / quat / euler / xyaxes / zaxis), ``<inertial>`` or inertia
  inferred from primitive geoms, hinge / slide / free joints, geoms, sites,
  cameras
* fixed tendons, joint equalities, position / motor / general actuators

The output (:class:`Model`) is a bag of numpy arrays named after the ``mjModel``
fields they correspond to; ``rcs_amd._lib`` marshals it across the C-ABI.
"""

from __future__ import annotations

import math
import os
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field

import numpy as np

# joint / geom / actuator enums (values follow mjtJoint, mjtGeom, ... so that
# tables read like mjModel dumps)
JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = 0, 1, 2, 3
GEOM_PLANE, GEOM_HFIELD, GEOM_SPHERE, GEOM_CAPSULE, GEOM_ELLIPSOID, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH = range(8)
GEOM_TYPES = {
    "plane": GEOM_PLANE,
    "hfield": GEOM_HFIELD,
    "sphere": GEOM_SPHERE,
    "capsule": GEOM_CAPSULE,
    "ellipsoid": GEOM_ELLIPSOID,
    "cylinder": GEOM_CYLINDER,
    "box": GEOM_BOX,
    "mesh": GEOM_MESH,
}
TRN_JOINT, TRN_TENDON = 0, 3
BIAS_NONE, BIAS_AFFINE = 0, 1
GAIN_FIXED = 0
EQ_JOINT = 2

DEFAULT_SOLREF = (0.02, 1.0)
DEFAULT_SOLIMP = (0.9, 0.95, 0.001, 0.5, 2.0)


class MjcfError(ValueError):
    pass


# --------------------------------------------------------------------------- math


def _floats(s: str | None, n: int | None = None, default=None) -> np.ndarray | None:
    if s is None:
        return None if default is None else np.array(default, dtype=np.float64)
    v = np.array([float(x) for x in s.split()], dtype=np.float64)
    if n is not None and len(v) != n:
        # MuJoCo pads short vectors with the defaults (e.g. friction="1" -> 3 numbers)
        if default is not None and len(v) < n:
            full = np.array(default, dtype=np.float64)
            full[: len(v)] = v
            return full
        raise MjcfError(f"expected {n} numbers, got {s!r}")
    return v


def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array(
        [
            aw * bw - ax * bx - ay * by - az * bz,
            aw * bx + ax * bw + ay * bz - az * by,
            aw * by - ax * bz + ay * bw + az * bx,
            aw * bz + ax * by - ay * bx + az * bw,
        ]
    )


def quat_normalize(q):
    q = np.asarray(q, dtype=np.float64)
    n = np.linalg.norm(q)
    if n < 1e-15:
        return np.array([1.0, 0, 0, 0])
    return q / n


def quat_to_mat(q):
    w, x, y, z = q
    return np.array(
        [
            [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
            [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
            [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z],
        ]
    )


def mat_to_quat(m):
    """Rotation matrix -> wxyz quaternion (largest-component branch, w >= 0 not forced)."""
    m = np.asarray(m, dtype=np.float64)
    t = np.trace(m)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = [0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s]
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = math.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
        q = [(m[2, 1] - m[1, 2]) / s, 0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s]
    elif m[1, 1] > m[2, 2]:
        s = math.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
        q = [(m[0, 2] - m[2, 0]) / s, (m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s]
    else:
        s = math.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
        q = [(m[1, 0] - m[0, 1]) / s, (m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s]
    return quat_normalize(q)


def _axis_angle_quat(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    s = math.sin(angle / 2)
    return np.array([math.cos(angle / 2), axis[0] * s, axis[1] * s, axis[2] * s])


# ------------------------------------------------------------------------ defaults


class _Defaults:
    """One node of the ``<default>`` class tree.

    ``attrs[tag]`` holds the accumulated attribute dict of that element type.  All
    actuator shortcuts share the pseudo-tag ``"actuator"`` (MuJoCo keeps a single
    dummy actuator per class; a ``<position>`` default therefore also sets
    ``biastype="affine"`` for a ``<general>`` of the same class).
    """

    def __init__(self, name: str, parent: "_Defaults | None"):
        self.name = name
        self.parent = parent
        self.attrs: dict[str, dict[str, str]] = {}
        if parent is not None:
            self.attrs = {k: dict(v) for k, v in parent.attrs.items()}


_ACTUATOR_TAGS = ("general", "motor", "position", "velocity")


def _apply_actuator_shortcut(tag: str, given: dict[str, str], base: dict[str, str]) -> dict[str, str]:
    """Fold an actuator element into the canonical ``general`` attribute dict."""
    out = dict(base)
    g = dict(given)
    gain = _floats(out.get("gainprm"), 3, (1, 0, 0))
    bias = _floats(out.get("biasprm"), 3, (0, 0, 0))
    if tag == "motor":
        out["gaintype"], out["biastype"] = "fixed", "none"
    elif tag == "position":
        out["gaintype"], out["biastype"] = "fixed", "affine"
        kp = float(g.pop("kp")) if "kp" in g else gain[0]
        gain[0] = kp
        bias[1] = -kp
        if "kv" in g:
            bias[2] = -float(g.pop("kv"))
        out["gainprm"] = " ".join(repr(float(x)) for x in gain)
        out["biasprm"] = " ".join(repr(float(x)) for x in bias)
    elif tag == "velocity":
        out["gaintype"], out["biastype"] = "fixed", "affine"
        kv = float(g.pop("kv")) if "kv" in g else gain[0]
        gain[0] = kv
        bias = np.array([0.0, 0.0, -kv])
        out["gainprm"] = " ".join(repr(float(x)) for x in gain)
        out["biasprm"] = " ".join(repr(float(x)) for x in bias)
    out.update(g)
    return out


# --------------------------------------------------------------------------- model


@dataclass
class Model:
    """Flat constant tables of one compiled scene (field names follow ``mjModel``)."""

    # options
    timestep: float = 0.002
    gravity: np.ndarray = field(default_factory=lambda: np.array([0.0, 0.0, -9.81]))
    integrator: str = "Euler"
    cone: str = "pyramidal"
    impratio: float = 1.0
    noslip_iterations: int = 0
    # rendering: mjModel.stat.extent (None: not given -- MuJoCo would derive it from the model's bounding box),
    # mjModel.vis.map.znear / zfar (fractions of extent)
    stat_extent: float | None = None
    stat_center: np.ndarray | None = None
    vis_znear: float = 0.01
    vis_zfar: float = 50.0
    # mjModel.vis.global: orientation of the default free camera, its field of view
    vis_azimuth: float = 90.0
    vis_elevation: float = -45.0
    vis_fovy: float = 45.0
    # colour rendering: mjModel.vis.headlight, the scene's directional lights (direction, diffuse), the skybox gradient
    # (rgb1 at the zenith, rgb2 at the nadir; None: no skybox, the background is black) -- rcs_amd/render.py
    vis_headlight_ambient: np.ndarray = field(default_factory=lambda: np.array([0.1, 0.1, 0.1]))
    vis_headlight_diffuse: np.ndarray = field(default_factory=lambda: np.array([0.4, 0.4, 0.4]))
    lights: list = field(default_factory=list)
    skybox: tuple | None = None
    # sizes
    nbody: int = 0
    njnt: int = 0
    nq: int = 0
    nv: int = 0
    nu: int = 0
    ngeom: int = 0
    nsite: int = 0
    ncam: int = 0
    ntendon: int = 0
    nwrap: int = 0
    neq: int = 0
    # names
    body_names: list[str] = field(default_factory=list)
    jnt_names: list[str] = field(default_factory=list)
    geom_names: list[str] = field(default_factory=list)
    site_names: list[str] = field(default_factory=list)
    cam_names: list[str] = field(default_factory=list)
    tendon_names: list[str] = field(default_factory=list)
    actuator_names: list[str] = field(default_factory=list)
    arrays: dict[str, np.ndarray] = field(default_factory=dict)

    def __getattr__(self, item):
        arrays = self.__dict__.get("arrays", {})
        if item in arrays:
            return arrays[item]
        raise AttributeError(item)

    def name2id(self, kind: str, name: str) -> int:
        names = getattr(self, f"{kind}_names")
        try:
            return names.index(name)
        except ValueError:
            return -1


# ------------------------------------------------------------------------ compiler


def find_data_file(dirs: list[str], name: str) -> str | None:
    """First of `dirs` (the scene's own directory, then those of the files it includes) that holds `name`."""
    for d in dirs:
        f = os.path.join(d, name)
        if os.path.exists(f):
            return f
    return None


class _Compiler:
    def __init__(self, path: str):
        self.data_dirs = [os.path.dirname(os.path.abspath(path))]
        self.path = os.path.abspath(path)
        self.root = self._load(self.path)
        self.angle_deg = True  # MuJoCo default: degrees
        self.eulerseq = "xyz"
        self.autolimits = True
        self.defaults: dict[str, _Defaults] = {"main": _Defaults("main", None)}
        # accumulators
        self.bodies: list[dict] = []
        self.joints: list[dict] = []
        self.geoms: list[dict] = []
        self.sites: list[dict] = []
        self.cams: list[dict] = []
        self.tendons: list[dict] = []
        self.equalities: list[dict] = []
        self.actuators: list[dict] = []
        self.option: dict[str, str] = {}

    # ---- loading with <include> expansion
    def _load(self, path: str) -> ET.Element:
        root = ET.parse(path).getroot()
        if root.tag != "mujoco":
            raise MjcfError(f"{path}: root element must be <mujoco>")
        self._expand_includes(root, os.path.dirname(path))
        return root

    def _expand_includes(self, elem: ET.Element, base: str):
        i = 0
        children = list(elem)
        for child in children:
            if child.tag == "include":
                inc_path = os.path.join(base, child.attrib["file"])
                self.data_dirs.append(os.path.dirname(os.path.abspath(inc_path)))  # mesh-derived tables may live beside an included file
                inc_root = ET.parse(inc_path).getroot()
                self._expand_includes(inc_root, os.path.dirname(inc_path))
                idx = list(elem).index(child)
                elem.remove(child)
                for k, sub in enumerate(list(inc_root)):
                    elem.insert(idx + k, sub)
            else:
                self._expand_includes(child, base)
            i += 1

    # ---- helpers
    def _angle(self, v):
        return np.deg2rad(v) if self.angle_deg else v

    def _orientation(self, a: dict[str, str]) -> np.ndarray:
        if "quat" in a:
            return quat_normalize(_floats(a["quat"], 4))
        if "euler" in a:
            e = self._angle(_floats(a["euler"], 3))
            q = np.array([1.0, 0, 0, 0])
            for ch, ang in zip(self.eulerseq, e):
                axis = {"x": (1, 0, 0), "y": (0, 1, 0), "z": (0, 0, 1)}[ch.lower()]
                r = _axis_angle_quat(axis, ang)
                q = quat_mul(q, r) if ch.islower() else quat_mul(r, q)
            return quat_normalize(q)
        if "axisangle" in a:
            v = _floats(a["axisangle"], 4)
            ax = v[:3] / np.linalg.norm(v[:3])
            return quat_normalize(_axis_angle_quat(ax, float(self._angle(v[3]))))
        if "xyaxes" in a:
            v = _floats(a["xyaxes"], 6)
            x = v[:3] / np.linalg.norm(v[:3])
            y = v[3:] - x * np.dot(x, v[3:])
            y = y / np.linalg.norm(y)
            z = np.cross(x, y)
            return mat_to_quat(np.stack([x, y, z], axis=1))
        if "zaxis" in a:
            z = _floats(a["zaxis"], 3)
            z = z / np.linalg.norm(z)
            src = np.array([0.0, 0, 1])
            ax = np.cross(src, z)
            s = np.linalg.norm(ax)
            ang = math.atan2(s, float(np.dot(src, z)))
            if s < 1e-10:
                ax = np.array([1.0, 0, 0])
            else:
                ax = ax / s
            return quat_normalize(_axis_angle_quat(ax, ang))
        return np.array([1.0, 0, 0, 0])

    def _resolve(self, tag: str, elem: ET.Element, childclass: str | None) -> dict[str, str]:
        cls = elem.attrib.get("class", childclass or "main")
        if cls not in self.defaults:
            raise MjcfError(f"unknown default class {cls!r}")
        d = self.defaults[cls]
        given = {k: v for k, v in elem.attrib.items() if k != "class"}
        if tag in _ACTUATOR_TAGS:
            return _apply_actuator_shortcut(tag, given, d.attrs.get("actuator", {}))
        out = dict(d.attrs.get(tag, {}))
        out.update(given)
        return out

    def _geom_colour(self, ga: dict) -> dict:
        """rgba of a geom (its own, else its material's, else MuJoCo's grey) and, for a material with a builtin checker
        texture, the two colours and the edge length of the squares (texuniform: `texrepeat` tiles of 2 x 2 squares per unit)."""
        rgba = _floats(ga.get("rgba"), 4, (0.5, 0.5, 0.5, 1.0))
        checker = None
        mat = self._materials.get(ga.get("material", ""))
        if ga.get("material") and mat is None:
            raise MjcfError(f"unknown material {ga.get('material')!r}")
        if mat is not None:
            if "rgba" not in ga:
                rgba = _floats(mat.get("rgba"), 4, (1.0, 1.0, 1.0, 1.0))
            tex = self._textures.get(mat.get("texture", ""))
            if tex is not None and tex.get("builtin") == "checker":
                rep = _floats(mat.get("texrepeat"), 2, (1.0, 1.0))
                checker = (_floats(tex.get("rgb1"), 3, (0.8, 0.8, 0.8)), _floats(tex.get("rgb2"), 3, (0.5, 0.5, 0.5)), 0.5 / float(rep[0]))
        return dict(rgba=rgba, checker=checker)

    def _parse_assets(self):
        self._materials: dict[str, dict] = {}
        self._textures: dict[str, dict] = {}
        for asset in self.root.findall("asset"):
            for t in asset.findall("texture"):
                self._textures[t.attrib.get("name", "__" + t.attrib.get("type", "2d"))] = dict(t.attrib)
            for mt in asset.findall("material"):
                self._materials[mt.attrib["name"]] = dict(mt.attrib)

    # ---- sections
    def _parse_compiler(self):
        for c in self.root.findall("compiler"):
            if "angle" in c.attrib:
                self.angle_deg = c.attrib["angle"] == "degree"
            if "eulerseq" in c.attrib:
                self.eulerseq = c.attrib["eulerseq"]
            if "autolimits" in c.attrib:
                self.autolimits = c.attrib["autolimits"] == "true"
        for o in self.root.findall("option"):
            self.option.update(o.attrib)

    def _walk_defaults(self, elem: ET.Element, node: _Defaults):
        own = [c for c in elem if c.tag != "default"]
        nested = [c for c in elem if c.tag == "default"]
        for child in own:
            if child.tag in _ACTUATOR_TAGS:
                node.attrs["actuator"] = _apply_actuator_shortcut(
                    child.tag, dict(child.attrib), node.attrs.get("actuator", {})
                )
            else:
                node.attrs.setdefault(child.tag, {}).update(child.attrib)
        for child in nested:
            name = child.attrib.get("class")
            if name is None:
                raise MjcfError("nested <default> needs a class name")
            sub = _Defaults(name, node)
            self.defaults[name] = sub
            self._walk_defaults(child, sub)

    def _parse_body(self, elem: ET.Element, parent: int, childclass: str | None):
        a = elem.attrib
        bid = len(self.bodies)
        if elem.tag == "worldbody":
            body = dict(name="world", parent=0, pos=np.zeros(3), quat=np.array([1.0, 0, 0, 0]), gravcomp=0.0)
        else:
            childclass = a.get("childclass", childclass)
            body = dict(
                name=a.get("name", f"body{bid}"),
                parent=parent,
                pos=_floats(a.get("pos"), 3, (0, 0, 0)),
                quat=self._orientation(a),
                gravcomp=float(a.get("gravcomp", 0)),
            )
        body.update(inertial=None, geoms=[], joints=[])
        self.bodies.append(body)
        for child in elem:
            if child.tag == "body":
                self._parse_body(child, bid, childclass)
            elif child.tag == "inertial":
                ia = child.attrib
                inert = dict(
                    pos=_floats(ia.get("pos"), 3, (0, 0, 0)),
                    mass=float(ia["mass"]),
                )
                if "fullinertia" in ia:
                    f = _floats(ia["fullinertia"], 6)
                    full = np.array([[f[0], f[3], f[4]], [f[3], f[1], f[5]], [f[4], f[5], f[2]]])
                    w, vecs = np.linalg.eigh(full)
                    if np.linalg.det(vecs) < 0:
                        vecs[:, 2] = -vecs[:, 2]
                    inert["quat"] = mat_to_quat(vecs)
                    inert["diag"] = w
                else:
                    inert["quat"] = self._orientation(ia)
                    inert["diag"] = _floats(ia["diaginertia"], 3)
                body["inertial"] = inert
            elif child.tag in ("joint", "freejoint"):
                ja = self._resolve("joint", child, childclass) if child.tag == "joint" else dict(child.attrib, type="free")
                jtype = {"hinge": JNT_HINGE, "slide": JNT_SLIDE, "free": JNT_FREE, "ball": JNT_BALL}[ja.get("type", "hinge")]
                axis = _floats(ja.get("axis"), 3, (0, 0, 1))
                axis = axis / np.linalg.norm(axis)
                rng = _floats(ja.get("range"), 2, (0, 0))
                if jtype == JNT_HINGE:
                    rng = self._angle(rng)
                lim_attr = ja.get("limited", "auto")
                limited = (lim_attr == "true") or (lim_attr == "auto" and self.autolimits and "range" in ja)
                afr = _floats(ja.get("actuatorfrcrange"), 2, (0, 0))
                afl_attr = ja.get("actuatorfrclimited", "auto")
                afl = (afl_attr == "true") or (afl_attr == "auto" and self.autolimits and "actuatorfrcrange" in ja)
                ref = float(ja.get("ref", 0))
                if jtype == JNT_HINGE:
                    ref = float(self._angle(ref))
                joint = dict(
                    name=ja.get("name", f"joint{len(self.joints)}"),
                    type=jtype,
                    body=bid,
                    pos=_floats(ja.get("pos"), 3, (0, 0, 0)),
                    axis=axis,
                    limited=limited,
                    range=rng,
                    ref=ref,
                    armature=float(ja.get("armature", 0)),
                    damping=float(ja.get("damping", 0)),
                    frictionloss=float(ja.get("frictionloss", 0)),
                    stiffness=float(ja.get("stiffness", 0)),
                    actfrclimited=afl,
                    actfrcrange=afr,
                    actgravcomp=ja.get("actuatorgravcomp", "false") == "true",
                    solref=_floats(ja.get("solreflimit"), 2, DEFAULT_SOLREF),
                    solimp=_floats(ja.get("solimplimit"), 5, DEFAULT_SOLIMP),
                    solreffriction=_floats(ja.get("solreffriction"), 2, DEFAULT_SOLREF),
                    solimpfriction=_floats(ja.get("solimpfriction"), 5, DEFAULT_SOLIMP),
                    margin=float(ja.get("margin", 0)),
                )
                body["joints"].append(len(self.joints))
                self.joints.append(joint)
            elif child.tag == "geom":
                ga = self._resolve("geom", child, childclass)
                gtype = GEOM_TYPES[ga.get("type", "sphere")]
                size = _floats(ga.get("size"), None, (0, 0, 0))
                size3 = np.zeros(3)
                size3[: len(size)] = size
                pos = _floats(ga.get("pos"), 3, (0, 0, 0))
                quat = self._orientation(ga)
                if "fromto" in ga:
                    ft = _floats(ga["fromto"], 6)
                    p0, p1 = ft[:3], ft[3:]
                    pos = 0.5 * (p0 + p1)
                    d = p1 - p0
                    ln = np.linalg.norm(d)
                    quat = self._orientation({"zaxis": " ".join(map(repr, map(float, d / ln)))})
                    size3[1] = 0.5 * ln
                geom = dict(
                    name=ga.get("name", ""),
                    type=gtype,
                    body=bid,
                    pos=pos,
                    quat=quat,
                    size=size3,
                    contype=int(ga.get("contype", 1)),
                    conaffinity=int(ga.get("conaffinity", 1)),
                    condim=int(ga.get("condim", 3)),
                    group=int(ga.get("group", 0)),
                    priority=int(ga.get("priority", 0)),
                    friction=_floats(ga.get("friction"), 3, (1, 0.005, 0.0001)),
                    solref=_floats(ga.get("solref"), 2, DEFAULT_SOLREF),
                    solimp=_floats(ga.get("solimp"), 5, DEFAULT_SOLIMP),
                    margin=float(ga.get("margin", 0)),
                    gap=float(ga.get("gap", 0)),
                    mesh=ga.get("mesh", ""),
                    mass=float(ga["mass"]) if "mass" in ga else None,
                    density=float(ga.get("density", 1000)),
                    **self._geom_colour(ga),
                )
                body["geoms"].append(len(self.geoms))
                self.geoms.append(geom)
            elif child.tag == "site":
                sa = self._resolve("site", child, childclass)
                self.sites.append(
                    dict(name=sa.get("name", ""), body=bid, pos=_floats(sa.get("pos"), 3, (0, 0, 0)), quat=self._orientation(sa))
                )
            elif child.tag == "camera":
                ca = self._resolve("camera", child, childclass)
                res = _floats(ca.get("resolution"), 2, (1, 1))
                self.cams.append(
                    dict(
                        name=ca.get("name", ""),
                        body=bid,
                        pos=_floats(ca.get("pos"), 3, (0, 0, 0)),
                        quat=self._orientation(ca),
                        fovy=float(ca.get("fovy", 45.0)),
                        resolution=res,
                    )
                )
            # lights and anything visual are not on the Sim.step() path

    def _parse_rest(self):
        jid = {j["name"]: i for i, j in enumerate(self.joints)}
        for ten in self.root.findall("tendon"):
            for fx in ten.findall("fixed"):
                wraps = []
                for j in fx.findall("joint"):
                    if j.attrib["joint"] not in jid:
                        raise MjcfError(f"tendon references unknown joint {j.attrib['joint']!r}")
                    wraps.append((jid[j.attrib["joint"]], float(j.attrib["coef"])))
                self.tendons.append(dict(name=fx.attrib.get("name", ""), wraps=wraps))
            if ten.findall("spatial"):
                raise MjcfError("spatial tendons are outside the supported MJCF subset")
        for eq in self.root.findall("equality"):
            for e in eq:
                if e.tag != "joint":
                    raise MjcfError(f"<equality><{e.tag}> is outside the supported MJCF subset")
                ea = self._resolve("equality", e, None)
                j2 = ea.get("joint2")
                self.equalities.append(
                    dict(
                        type=EQ_JOINT,
                        obj1=jid[ea["joint1"]],
                        obj2=jid[j2] if j2 is not None else -1,
                        polycoef=_floats(ea.get("polycoef"), 5, (0, 1, 0, 0, 0)),
                        solref=_floats(ea.get("solref"), 2, DEFAULT_SOLREF),
                        solimp=_floats(ea.get("solimp"), 5, DEFAULT_SOLIMP),
                        active=ea.get("active", "true") == "true",
                    )
                )
        tid = {t["name"]: i for i, t in enumerate(self.tendons)}
        for act in self.root.findall("actuator"):
            for e in act:
                if e.tag not in _ACTUATOR_TAGS:
                    raise MjcfError(f"<actuator><{e.tag}> is outside the supported MJCF subset")
                aa = self._resolve(e.tag, e, None)
                if aa.get("dyntype", "none") != "none":
                    raise MjcfError("actuator dynamics (dyntype) are outside the supported MJCF subset")
                if "joint" in aa:
                    trntype, trnid = TRN_JOINT, jid[aa["joint"]]
                elif "tendon" in aa:
                    trntype, trnid = TRN_TENDON, tid[aa["tendon"]]
                else:
                    raise MjcfError("actuator needs a joint or tendon transmission")
                gaintype = aa.get("gaintype", "fixed")
                biastype = aa.get("biastype", "none")
                if gaintype != "fixed" or biastype not in ("none", "affine"):
                    raise MjcfError("only fixed gain / none|affine bias actuators are supported")
                ctrlrange = _floats(aa.get("ctrlrange"), 2, (0, 0))
                has_ctrlrange = "ctrlrange" in aa
                inherit = float(aa.get("inheritrange", 0))
                if not has_ctrlrange and inherit > 0 and trntype == TRN_JOINT:
                    # position shortcut: ctrlrange := joint range scaled about its mean
                    r = self.joints[trnid]["range"]
                    mean, half = 0.5 * (r[0] + r[1]), 0.5 * (r[1] - r[0]) * inherit
                    ctrlrange = np.array([mean - half, mean + half])
                    has_ctrlrange = True
                cl_attr = aa.get("ctrllimited", "auto")
                ctrllimited = (cl_attr == "true") or (cl_attr == "auto" and self.autolimits and has_ctrlrange)
                fl_attr = aa.get("forcelimited", "auto")
                forcelimited = (fl_attr == "true") or (fl_attr == "auto" and self.autolimits and "forcerange" in aa)
                self.actuators.append(
                    dict(
                        name=aa.get("name", ""),
                        trntype=trntype,
                        trnid=trnid,
                        gear=_floats(aa.get("gear"), 6, (1, 0, 0, 0, 0, 0))[0],
                        gainprm=_floats(aa.get("gainprm"), 3, (1, 0, 0)),
                        biasprm=_floats(aa.get("biasprm"), 3, (0, 0, 0)),
                        biastype=BIAS_AFFINE if biastype == "affine" else BIAS_NONE,
                        ctrllimited=ctrllimited,
                        ctrlrange=ctrlrange,
                        forcelimited=forcelimited,
                        forcerange=_floats(aa.get("forcerange"), 2, (0, 0)),
                    )
                )

    # ---- inertia inferred from primitive geoms (bodies without <inertial>)
    @staticmethod
    def _geom_mass_inertia(g: dict):
        t, s = g["type"], g["size"]
        if t == GEOM_SPHERE:
            vol = 4.0 / 3.0 * math.pi * s[0] ** 3
            unit = np.full(3, 0.4 * s[0] ** 2)
        elif t == GEOM_BOX:
            vol = 8 * s[0] * s[1] * s[2]
            unit = np.array([s[1] ** 2 + s[2] ** 2, s[0] ** 2 + s[2] ** 2, s[0] ** 2 + s[1] ** 2]) / 3.0
        elif t == GEOM_CYLINDER:
            r, h = s[0], 2 * s[1]
            vol = math.pi * r * r * h
            unit = np.array([(3 * r * r + h * h) / 12.0, (3 * r * r + h * h) / 12.0, r * r / 2.0])
        elif t == GEOM_CAPSULE:
            r, h = s[0], 2 * s[1]
            vc, vs = math.pi * r * r * h, 4.0 / 3.0 * math.pi * r**3
            vol = vc + vs
            # cylinder + two hemispheres (parallel axis for the caps)
            ic_ax = 0.5 * r * r * vc
            ic_tr = vc * (3 * r * r + h * h) / 12.0
            is_ax = 0.4 * r * r * vs
            is_tr = vs * (0.4 * r * r + 0.375 * r * h + 0.25 * h * h)
            unit = np.array([ic_tr + is_tr, ic_tr + is_tr, ic_ax + is_ax]) / vol
        elif t == GEOM_ELLIPSOID:
            vol = 4.0 / 3.0 * math.pi * s[0] * s[1] * s[2]
            unit = np.array([s[1] ** 2 + s[2] ** 2, s[0] ** 2 + s[2] ** 2, s[0] ** 2 + s[1] ** 2]) / 5.0
        else:
            return None
        mass = g["mass"] if g["mass"] is not None else g["density"] * vol
        return mass, unit * mass

    def _infer_inertial(self, body: dict):
        total, com, parts = 0.0, np.zeros(3), []
        for gi in body["geoms"]:
            g = self.geoms[gi]
            if g["mass"] is not None and g["mass"] == 0:
                continue
            if g["mass"] is None and g["density"] == 0:
                continue
            if g["type"] in (GEOM_PLANE, GEOM_HFIELD):
                continue
            mi = self._geom_mass_inertia(g)
            if mi is None:
                # mesh geoms need the mesh volume; static bodies do not care, moving ones must
                # carry an explicit <inertial> in this subset
                body["_needs_mesh_inertia"] = True
                continue
            m, diag = mi
            if m <= 0:
                continue
            parts.append((m, g["pos"], quat_to_mat(g["quat"]), diag))
            total += m
            com += m * g["pos"]
        if total <= 0:
            return dict(pos=np.zeros(3), quat=np.array([1.0, 0, 0, 0]), mass=0.0, diag=np.zeros(3))
        com /= total
        full = np.zeros((3, 3))
        for m, p, R, diag in parts:
            d = p - com
            full += R @ np.diag(diag) @ R.T + m * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
        w, vecs = np.linalg.eigh(full)
        # MuJoCo orders principal inertias descending
        order = np.argsort(-w)
        w, vecs = w[order], vecs[:, order]
        if np.linalg.det(vecs) < 0:
            vecs[:, 2] = -vecs[:, 2]
        return dict(pos=com, quat=mat_to_quat(vecs), mass=total, diag=w)

    # ---- assemble
    def compile(self) -> Model:
        self._parse_compiler()
        self._parse_assets()
        for top in self.root.findall("default"):
            self._walk_defaults(top, self.defaults["main"])
        worlds = self.root.findall("worldbody")
        if not worlds:
            raise MjcfError("no <worldbody>")
        # several <worldbody> sections (from includes) merge into body 0
        self._parse_body(worlds[0], 0, None)
        for extra in worlds[1:]:
            # re-parent children of later worldbody sections onto the existing world body
            saved = self.bodies[0]
            tmp_start = len(self.bodies)
            self._parse_body(extra, 0, None)
            dup = self.bodies.pop(tmp_start)  # the duplicate "world" entry
            saved["geoms"] += dup["geoms"]
            for b in self.bodies[tmp_start:]:
                b["parent"] = b["parent"] - 1 if b["parent"] > tmp_start else (0 if b["parent"] == tmp_start else b["parent"])
            for coll in (self.joints, self.geoms, self.sites, self.cams):
                for item in coll:
                    if item["body"] == tmp_start:
                        item["body"] = 0
                    elif item["body"] > tmp_start:
                        item["body"] -= 1
        self._split_free_bodies()
        self._sort_bodies_depth_first()
        self._parse_rest()
        m = self._emit()
        m.free_bodies = self.free_bodies  # type: ignore[attr-defined]
        m.data_dirs = list(self.data_dirs)  # type: ignore[attr-defined]
        return m

    def _split_free_bodies(self):
        """Take free-floating objects (a child of the world with one free joint and one box geom, e.g. the cube of
        ``fr3_simple_pick_up``) out of the articulated tables: they are simulated as separate rigid bodies that
        touch the floor plane.  They must come last in the file, as in the reference's scenes."""
        self.free_bodies: list[dict] = []
        while len(self.bodies) > 1:
            b = self.bodies[-1]
            if not (b["joints"] and self.joints[b["joints"][0]]["type"] == JNT_FREE):
                break
            bid = len(self.bodies) - 1
            if b["parent"] != 0 or len(b["joints"]) != 1 or b["joints"][0] != len(self.joints) - 1:
                raise MjcfError(f"free body {b['name']!r}: only top-level bodies with a single free joint are supported")
            gids = b["geoms"]
            if len(gids) != 1 or self.geoms[gids[0]]["type"] != GEOM_BOX or gids[0] != len(self.geoms) - 1:
                raise MjcfError(f"free body {b['name']!r}: exactly one box geom is supported")
            if any(x["body"] == bid for x in self.sites + self.cams):
                raise MjcfError(f"free body {b['name']!r}: sites / cameras on free bodies are not supported")
            g = self.geoms[gids[0]]
            if np.any(g["pos"] != 0) or np.any(quat_normalize(g["quat"]) != np.array([1.0, 0, 0, 0])):
                raise MjcfError(f"free body {b['name']!r}: the box geom must sit at the body frame")
            inert = b["inertial"] if b["inertial"] is not None else None
            if inert is not None:
                raise MjcfError(f"free body {b['name']!r}: explicit <inertial> is not supported")
            mass, diag = self._geom_mass_inertia(g)
            planes = [p for p in self.geoms if p["type"] == GEOM_PLANE and p["body"] == 0]
            if len(planes) != 1:
                raise MjcfError("free bodies need exactly one floor plane in the world body")
            pl = planes[0]
            if np.any(quat_normalize(pl["quat"]) != np.array([1.0, 0, 0, 0])):
                raise MjcfError("the floor plane must be horizontal")
            if pl["priority"] != g["priority"] or pl["condim"] != 3 or g["condim"] != 3:
                raise MjcfError("floor / box contact: equal priority and condim 3 are supported")
            if pl["margin"] or pl["gap"] or g["margin"] or g["gap"]:
                raise MjcfError("floor / box contact: margin and gap are not supported")
            j = self.joints[b["joints"][0]]
            self.free_bodies.insert(0, dict(
                name=b["name"], joint_name=j["name"], geom_name=g["name"],
                qpos0=np.concatenate([b["pos"], quat_normalize(b["quat"])]),
                mass=float(mass), inertia=np.asarray(diag, dtype=np.float64), size=g["size"].copy(),
                # mj_contactParam, equal priority: friction element-wise max, solref / solimp mixed 50:50 (solmix 1:1)
                friction=np.maximum(pl["friction"], g["friction"]), geom_friction=g["friction"].copy(), floor_friction=pl["friction"].copy(), rgba=g["rgba"].copy(),
                solref=0.5 * (pl["solref"] + g["solref"]), solimp=0.5 * (pl["solimp"] + g["solimp"]),
                plane_z=float(pl["pos"][2]),
            ))
            self.bodies.pop()
            self.joints.pop()
            self.geoms.pop()

    def _sort_bodies_depth_first(self):
        # _parse_body already emits parents before children in depth-first order, which is the
        # mjModel ordering; nothing to do except assert it.
        for i, b in enumerate(self.bodies):
            if i and b["parent"] >= i:
                raise MjcfError("body order is not parent-first")

    def _emit(self) -> Model:
        o = self.option
        m = Model()
        m.timestep = float(o.get("timestep", 0.002))
        m.gravity = _floats(o.get("gravity"), 3, (0, 0, -9.81))
        m.integrator = o.get("integrator", "Euler")
        m.cone = o.get("cone", "pyramidal")
        m.impratio = float(o.get("impratio", 1))
        m.noslip_iterations = int(o.get("noslip_iterations", 0))
        for st in self.root.findall("statistic"):
            if "extent" in st.attrib:
                m.stat_extent = float(st.attrib["extent"])
            if "center" in st.attrib:
                m.stat_center = _floats(st.attrib["center"], 3)
        for vis in self.root.findall("visual"):
            for gl in vis.findall("global"):
                m.vis_azimuth = float(gl.attrib.get("azimuth", m.vis_azimuth))
                m.vis_elevation = float(gl.attrib.get("elevation", m.vis_elevation))
                m.vis_fovy = float(gl.attrib.get("fovy", m.vis_fovy))
            for mp in vis.findall("map"):
                m.vis_znear = float(mp.attrib.get("znear", m.vis_znear))
                m.vis_zfar = float(mp.attrib.get("zfar", m.vis_zfar))
            for hl in vis.findall("headlight"):
                m.vis_headlight_ambient = _floats(hl.attrib.get("ambient"), 3, m.vis_headlight_ambient)
                m.vis_headlight_diffuse = _floats(hl.attrib.get("diffuse"), 3, m.vis_headlight_diffuse)
        for wb in self.root.findall("worldbody"):
            for li in wb.findall("light"):  # lights fixed in the world (what the RCS scenes have)
                if li.attrib.get("directional", "false") == "true":
                    d = _floats(li.attrib.get("dir"), 3, (0, 0, -1))
                    m.lights.append((d / np.linalg.norm(d), _floats(li.attrib.get("diffuse"), 3, (0.7, 0.7, 0.7))))
        for tex in self._textures.values():
            if tex.get("type") == "skybox" and tex.get("builtin") == "gradient":
                m.skybox = (_floats(tex.get("rgb1"), 3, (0.8, 0.8, 0.8)), _floats(tex.get("rgb2"), 3, (0.5, 0.5, 0.5)))

        nb = len(self.bodies)
        A: dict[str, np.ndarray] = {}
        A["body_parentid"] = np.array([b["parent"] for b in self.bodies], dtype=np.int32)
        A["body_pos"] = np.array([b["pos"] for b in self.bodies], dtype=np.float64).reshape(nb, 3)
        A["body_quat"] = np.array([b["quat"] for b in self.bodies], dtype=np.float64).reshape(nb, 4)
        A["body_gravcomp"] = np.array([b["gravcomp"] for b in self.bodies], dtype=np.float64)
        ipos, iquat, mass, inertia = [], [], [], []
        for i, b in enumerate(self.bodies):
            inert = b["inertial"] if b["inertial"] is not None else self._infer_inertial(b)
            if b.get("_needs_mesh_inertia") and b["inertial"] is None and b["joints"]:
                raise MjcfError(f"body {b['name']!r}: mesh-derived inertia needs an explicit <inertial>")
            ipos.append(inert["pos"])
            iquat.append(quat_normalize(inert["quat"]))
            mass.append(inert["mass"])
            inertia.append(inert["diag"])
        A["body_ipos"] = np.array(ipos, dtype=np.float64).reshape(nb, 3)
        A["body_iquat"] = np.array(iquat, dtype=np.float64).reshape(nb, 4)
        A["body_mass"] = np.array(mass, dtype=np.float64)
        A["body_inertia"] = np.array(inertia, dtype=np.float64).reshape(nb, 3)
        # weld / root ids
        rootid = np.zeros(nb, dtype=np.int32)
        weldid = np.zeros(nb, dtype=np.int32)
        for i, b in enumerate(self.bodies):
            if i == 0:
                continue
            p = b["parent"]
            rootid[i] = i if p == 0 else rootid[p]
            weldid[i] = i if b["joints"] else weldid[p]
        A["body_rootid"] = rootid
        A["body_weldid"] = weldid

        # joints: qpos / dof addressing
        nj = len(self.joints)
        qadr, dadr, nq, nv = [], [], 0, 0
        for j in self.joints:
            qadr.append(nq)
            dadr.append(nv)
            nq += {JNT_FREE: 7, JNT_BALL: 4}.get(j["type"], 1)
            nv += {JNT_FREE: 6, JNT_BALL: 3}.get(j["type"], 1)
        A["jnt_type"] = np.array([j["type"] for j in self.joints], dtype=np.int32)
        A["jnt_bodyid"] = np.array([j["body"] for j in self.joints], dtype=np.int32)
        A["jnt_qposadr"] = np.array(qadr, dtype=np.int32)
        A["jnt_dofadr"] = np.array(dadr, dtype=np.int32)
        A["jnt_pos"] = np.array([j["pos"] for j in self.joints], dtype=np.float64).reshape(nj, 3)
        A["jnt_axis"] = np.array([j["axis"] for j in self.joints], dtype=np.float64).reshape(nj, 3)
        A["jnt_limited"] = np.array([j["limited"] for j in self.joints], dtype=np.int32)
        A["jnt_range"] = np.array([j["range"] for j in self.joints], dtype=np.float64).reshape(nj, 2)
        A["jnt_margin"] = np.array([j["margin"] for j in self.joints], dtype=np.float64)
        A["jnt_solref"] = np.array([j["solref"] for j in self.joints], dtype=np.float64).reshape(nj, 2)
        A["jnt_solimp"] = np.array([j["solimp"] for j in self.joints], dtype=np.float64).reshape(nj, 5)
        A["jnt_actfrclimited"] = np.array([j["actfrclimited"] for j in self.joints], dtype=np.int32)
        A["jnt_actfrcrange"] = np.array([j["actfrcrange"] for j in self.joints], dtype=np.float64).reshape(nj, 2)
        A["jnt_actgravcomp"] = np.array([j["actgravcomp"] for j in self.joints], dtype=np.int32)
        # per-dof copies (1-dof joints: identical indexing; free joints expand)
        dof_jnt, dof_body, arm, damp, floss, fsolref, fsolimp = [], [], [], [], [], [], []
        for ji, j in enumerate(self.joints):
            n = {JNT_FREE: 6, JNT_BALL: 3}.get(j["type"], 1)
            for _ in range(n):
                dof_jnt.append(ji)
                dof_body.append(j["body"])
                arm.append(j["armature"])
                damp.append(j["damping"])
                floss.append(j["frictionloss"])
                fsolref.append(j["solreffriction"])
                fsolimp.append(j["solimpfriction"])
        A["dof_jntid"] = np.array(dof_jnt, dtype=np.int32)
        A["dof_bodyid"] = np.array(dof_body, dtype=np.int32)
        A["dof_armature"] = np.array(arm, dtype=np.float64)
        A["dof_damping"] = np.array(damp, dtype=np.float64)
        A["dof_frictionloss"] = np.array(floss, dtype=np.float64)
        A["dof_solref"] = np.array(fsolref, dtype=np.float64).reshape(len(floss), 2)
        A["dof_solimp"] = np.array(fsolimp, dtype=np.float64).reshape(len(floss), 5)
        qpos0 = np.zeros(nq)
        for j, qa in zip(self.joints, qadr):
            if j["type"] == JNT_FREE:
                b = self.bodies[j["body"]]
                qpos0[qa : qa + 3] = b["pos"]
                qpos0[qa + 3 : qa + 7] = b["quat"]
            elif j["type"] == JNT_BALL:
                qpos0[qa : qa + 4] = [1, 0, 0, 0]
            else:
                qpos0[qa] = j["ref"]
        A["qpos0"] = qpos0
        # body -> joint addressing
        A["body_jntnum"] = np.array([len(b["joints"]) for b in self.bodies], dtype=np.int32)
        A["body_jntadr"] = np.array([b["joints"][0] if b["joints"] else -1 for b in self.bodies], dtype=np.int32)

        ng = len(self.geoms)
        A["geom_type"] = np.array([g["type"] for g in self.geoms], dtype=np.int32)
        A["geom_bodyid"] = np.array([g["body"] for g in self.geoms], dtype=np.int32)
        A["geom_contype"] = np.array([g["contype"] for g in self.geoms], dtype=np.int32)
        A["geom_conaffinity"] = np.array([g["conaffinity"] for g in self.geoms], dtype=np.int32)
        A["geom_condim"] = np.array([g["condim"] for g in self.geoms], dtype=np.int32)
        A["geom_priority"] = np.array([g["priority"] for g in self.geoms], dtype=np.int32)
        A["geom_pos"] = np.array([g["pos"] for g in self.geoms], dtype=np.float64).reshape(ng, 3)
        A["geom_quat"] = np.array([g["quat"] for g in self.geoms], dtype=np.float64).reshape(ng, 4)
        A["geom_size"] = np.array([g["size"] for g in self.geoms], dtype=np.float64).reshape(ng, 3)
        A["geom_friction"] = np.array([g["friction"] for g in self.geoms], dtype=np.float64).reshape(ng, 3)
        A["geom_solref"] = np.array([g["solref"] for g in self.geoms], dtype=np.float64).reshape(ng, 2)
        A["geom_solimp"] = np.array([g["solimp"] for g in self.geoms], dtype=np.float64).reshape(ng, 5)
        A["geom_margin"] = np.array([g["margin"] for g in self.geoms], dtype=np.float64)
        A["geom_gap"] = np.array([g["gap"] for g in self.geoms], dtype=np.float64)
        A["geom_group"] = np.array([g["group"] for g in self.geoms], dtype=np.int32)
        m.arrays = A
        m.geom_mesh = [g["mesh"] for g in self.geoms]  # type: ignore[attr-defined]
        A["geom_rgba"] = np.array([g["rgba"] for g in self.geoms], dtype=np.float64).reshape(ng, 4)
        m.geom_checker = [g["checker"] for g in self.geoms]  # type: ignore[attr-defined]
        # collision vertex sets of mesh geoms (hull vertices, numbers only; tools/make_collision_vertices.py)
        vadr, vnum, verts = [], [], []
        vfile = find_data_file(self.data_dirs, "collision_vertices.npz")
        table = dict(np.load(vfile)) if vfile else {}
        for g in self.geoms:
            v = table.get(g["mesh"]) if g["type"] == GEOM_MESH else None
            vadr.append(sum(len(x) for x in verts))
            vnum.append(0 if v is None else len(v))
            if v is not None:
                verts.append(np.asarray(v, dtype=np.float64))
        A["geom_vertadr"] = np.array(vadr, dtype=np.int32)
        A["geom_vertnum"] = np.array(vnum, dtype=np.int32)
        A["mesh_vert"] = np.concatenate(verts).reshape(-1, 3) if verts else np.zeros((0, 3))

        ns = len(self.sites)
        A["site_bodyid"] = np.array([s["body"] for s in self.sites], dtype=np.int32)
        A["site_pos"] = np.array([s["pos"] for s in self.sites], dtype=np.float64).reshape(ns, 3)
        A["site_quat"] = np.array([s["quat"] for s in self.sites], dtype=np.float64).reshape(ns, 4)
        nc = len(self.cams)
        A["cam_bodyid"] = np.array([c["body"] for c in self.cams], dtype=np.int32)
        A["cam_pos"] = np.array([c["pos"] for c in self.cams], dtype=np.float64).reshape(nc, 3)
        A["cam_quat"] = np.array([c["quat"] for c in self.cams], dtype=np.float64).reshape(nc, 4)
        A["cam_fovy"] = np.array([c["fovy"] for c in self.cams], dtype=np.float64)

        # fixed tendons
        tadr, tnum, wobj, wprm = [], [], [], []
        for t in self.tendons:
            tadr.append(len(wobj))
            tnum.append(len(t["wraps"]))
            for jidx, coef in t["wraps"]:
                wobj.append(jidx)
                wprm.append(coef)
        A["tendon_adr"] = np.array(tadr, dtype=np.int32)
        A["tendon_num"] = np.array(tnum, dtype=np.int32)
        A["wrap_objid"] = np.array(wobj, dtype=np.int32)
        A["wrap_prm"] = np.array(wprm, dtype=np.float64)

        ne = len(self.equalities)
        A["eq_type"] = np.array([e["type"] for e in self.equalities], dtype=np.int32)
        A["eq_obj1id"] = np.array([e["obj1"] for e in self.equalities], dtype=np.int32)
        A["eq_obj2id"] = np.array([e["obj2"] for e in self.equalities], dtype=np.int32)
        A["eq_active0"] = np.array([e["active"] for e in self.equalities], dtype=np.int32)
        A["eq_data"] = np.array([e["polycoef"] for e in self.equalities], dtype=np.float64).reshape(ne, 5)
        A["eq_solref"] = np.array([e["solref"] for e in self.equalities], dtype=np.float64).reshape(ne, 2)
        A["eq_solimp"] = np.array([e["solimp"] for e in self.equalities], dtype=np.float64).reshape(ne, 5)

        nu = len(self.actuators)
        A["actuator_trntype"] = np.array([a["trntype"] for a in self.actuators], dtype=np.int32)
        A["actuator_trnid"] = np.array([a["trnid"] for a in self.actuators], dtype=np.int32)
        A["actuator_gear"] = np.array([a["gear"] for a in self.actuators], dtype=np.float64)
        A["actuator_gainprm"] = np.array([a["gainprm"] for a in self.actuators], dtype=np.float64).reshape(nu, 3)
        A["actuator_biasprm"] = np.array([a["biasprm"] for a in self.actuators], dtype=np.float64).reshape(nu, 3)
        A["actuator_biastype"] = np.array([a["biastype"] for a in self.actuators], dtype=np.int32)
        A["actuator_ctrllimited"] = np.array([a["ctrllimited"] for a in self.actuators], dtype=np.int32)
        A["actuator_ctrlrange"] = np.array([a["ctrlrange"] for a in self.actuators], dtype=np.float64).reshape(nu, 2)
        A["actuator_forcelimited"] = np.array([a["forcelimited"] for a in self.actuators], dtype=np.int32)
        A["actuator_forcerange"] = np.array([a["forcerange"] for a in self.actuators], dtype=np.float64).reshape(nu, 2)

        m.nbody, m.njnt, m.nq, m.nv, m.nu = nb, nj, nq, nv, nu
        m.ngeom, m.nsite, m.ncam = ng, ns, nc
        m.ntendon, m.nwrap, m.neq = len(self.tendons), len(wobj), ne
        m.body_names = [b["name"] for b in self.bodies]
        m.jnt_names = [j["name"] for j in self.joints]
        m.geom_names = [g["name"] for g in self.geoms]
        m.site_names = [s["name"] for s in self.sites]
        m.cam_names = [c["name"] for c in self.cams]
        m.tendon_names = [t["name"] for t in self.tendons]
        m.actuator_names = [a["name"] for a in self.actuators]
        return m


def compile_mjcf(path: str | os.PathLike) -> Model:
    """Compile the scene at ``path`` (``.xml``) into flat tables."""
    path = os.fspath(path)
    if not path.endswith(".xml"):
        raise MjcfError(f"Filetype {os.path.splitext(path)[1]} is unknown (only MJCF .xml is supported)")
    return _Compiler(path).compile()
