"""Generate tests/golden/reference_models.json: the physical constants of the reference's robot descriptions, read from the
reference's OWN MJCF files by a small independent reader (xml.etree + explicit default-class resolution; it shares no
code with rcs_amd/mjcf.py, which both the HIP backend and the oracle consume).  tests/test_oracle_pins.py compares the
compiled tables of this repository's authored scenes against it, so a wrong inertia, range, gain or default-class
resolution in the shared compiler -- invisible to kernel-vs-oracle parity -- shows up there.

    python tools/make_reference_model_fixture.py      (run where /root/reference exists)

Sources: assets/fr3/mjcf/fr3_0.xml + fr3_common.xml, assets/xarm7/mjcf/xarm7.xml, assets/scenes/fr3_simple_pick_up/scene.xml.
Numbers only: names, masses, frames, inertials, joint ranges / armature / damping / frictionloss / actuatorfrcrange, actuator
gains, tendon / equality parameters, primitive geom sizes and friction, solver options.
"""
import json
import os
import xml.etree.ElementTree as ET

REF = "/root/reference/assets"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_models.json")


def floats(s):
    return [float(x) for x in s.split()] if s is not None else None


class Defaults:
    """<default> tree: class name -> {tag: attributes}, children inheriting from their parents."""

    def __init__(self, root):
        self.cls = {}
        top = root.find("default")
        if top is not None:
            self._walk(top, {}, top.get("class", "main"))

    def _walk(self, node, inherited, name):
        mine = {k: dict(v) for k, v in inherited.items()}
        for child in node:
            if child.tag != "default":
                mine.setdefault(child.tag, {}).update(child.attrib)
        self.cls[name] = mine
        for child in node:
            if child.tag == "default":
                self._walk(child, mine, child.get("class"))

    def resolve(self, tag, elem, childclass):
        cls = elem.get("class", childclass)
        base = dict(self.cls.get(cls, self.cls.get("main", {})).get(tag, {})) if cls or "main" in self.cls else {}
        base.update(elem.attrib)
        return base


def read_robot(path):
    root = ET.parse(path).getroot()
    d = Defaults(root)
    out = {"bodies": [], "actuators": [], "equality": [], "tendons": []}

    def body(elem, parent, childclass):
        cc = elem.get("childclass", childclass)
        b = {"name": elem.get("name"), "parent": parent, "pos": floats(elem.get("pos")) or [0, 0, 0], "quat": floats(elem.get("quat")),
             "euler": floats(elem.get("euler")), "gravcomp": float(elem.get("gravcomp", 0)), "joints": [], "geoms": []}
        inert = elem.find("inertial")
        if inert is not None:
            b["inertial"] = {"mass": float(inert.get("mass")), "pos": floats(inert.get("pos")), "quat": floats(inert.get("quat")),
                             "diaginertia": floats(inert.get("diaginertia"))}
        for j in elem.findall("joint"):
            a = d.resolve("joint", j, cc)
            b["joints"].append({"name": a.get("name"), "type": a.get("type", "hinge"), "axis": floats(a.get("axis")) or [0, 0, 1], "range": floats(a.get("range")),
                                "pos": floats(a.get("pos")),
                                "armature": float(a.get("armature", 0)), "damping": float(a.get("damping", 0)), "frictionloss": float(a.get("frictionloss", 0)),
                                "actuatorfrcrange": floats(a.get("actuatorfrcrange")), "actuatorgravcomp": a.get("actuatorgravcomp", "false") == "true"})
        for g in elem.findall("geom"):
            a = d.resolve("geom", g, cc)
            if a.get("contype") == "0" and a.get("conaffinity") == "0":
                continue  # visual
            b["geoms"].append({"name": a.get("name"), "type": a.get("type", "sphere"), "size": floats(a.get("size")), "pos": floats(a.get("pos")),
                               "friction": floats(a.get("friction")), "mesh": a.get("mesh"), "mass": a.get("mass")})
        out["bodies"].append(b)
        for c in elem.findall("body"):
            body(c, b["name"], cc)

    for top in root.find("worldbody").findall("body"):
        body(top, "world", None)
    act = root.find("actuator")
    if act is not None:
        for a in act:
            r = d.resolve(a.tag, a, None)
            out["actuators"].append({"tag": a.tag, **{k: (floats(v) if k in ("gainprm", "biasprm", "forcerange", "ctrlrange") else v) for k, v in r.items()}})
    eq = root.find("equality")
    if eq is not None:
        for e in eq:
            out["equality"].append({"tag": e.tag, "joint1": e.get("joint1"), "joint2": e.get("joint2"), "solref": floats(e.get("solref")), "solimp": floats(e.get("solimp"))})
    ten = root.find("tendon")
    if ten is not None:
        for t in ten:
            out["tendons"].append({"name": t.get("name"), "joints": [(j.get("joint"), float(j.get("coef"))) for j in t.findall("joint")]})
    comp, opt = root.find("compiler"), root.find("option")
    out["compiler"] = dict(comp.attrib) if comp is not None else {}
    out["option"] = dict(opt.attrib) if opt is not None else {}
    return out


def main():
    api = {"fr3": read_robot(os.path.join(REF, "fr3/mjcf/fr3_0.xml")), "xarm7": read_robot(os.path.join(REF, "xarm7/mjcf/xarm7.xml"))}
    common = ET.parse(os.path.join(REF, "fr3/mjcf/fr3_common.xml")).getroot()
    api["fr3"]["option"] = dict(common.find("option").attrib)
    api["fr3"]["compiler"] = dict(common.find("compiler").attrib)
    scene = ET.parse(os.path.join(REF, "scenes/fr3_simple_pick_up/scene.xml")).getroot()
    box = [b for b in scene.find("worldbody").findall("body") if b.get("name") == "box_geom"][0]
    g = box.find("geom")
    api["pick_up_box"] = {"pos": floats(box.get("pos")), "quat": floats(box.get("quat")), "size": floats(g.get("size")), "friction": floats(g.get("friction")),
                          "density": float(g.get("density")), "joint": box.find("joint").get("type")}
    floor = [x for x in scene.find("worldbody").findall("geom") if x.get("name") == "floor"][0]
    api["pick_up_floor"] = {"type": floor.get("type"), "friction": floats(floor.get("friction"))}
    json.dump(api, open(OUT, "w"), indent=1, sort_keys=True)
    print("wrote", OUT, {k: len(v.get("bodies", [])) for k, v in api.items() if isinstance(v, dict)})


if __name__ == "__main__":
    main()
