"""Generate robot-control-stack_amd/rcs_amd/scenes/fr3_empty_world/fr3.urdf: the KINEMATIC content of the reference's
assets/fr3/urdf/fr3.urdf -- links by name, joints with origin / axis / limits -- without its collision primitives (the kinematics
classes read the chain only; the simulation's geometry comes from the MJCF scene).  The numbers are the robot's constants and must
equal the reference's.  Run where /root/reference exists:  python tools/make_fr3_urdf.py
"""
import os
import xml.etree.ElementTree as ET

SRC = "/root/reference/assets/fr3/urdf/fr3.urdf"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "robot-control-stack_amd", "rcs_amd", "scenes", "fr3_empty_world", "fr3.urdf")


def main():
    root = ET.parse(SRC).getroot()
    lines = ['<?xml version="1.0" ?>',
             "<!-- kinematic chain of the reference's assets/fr3/urdf/fr3.urdf (links, joint origins / axes / limits); made by tools/make_fr3_urdf.py -->",
             f'<robot name="{root.get("name")}">']
    for el in root:
        if el.tag == "link":
            lines.append(f'  <link name="{el.get("name")}"/>')
        elif el.tag == "joint":
            lines.append(f'  <joint name="{el.get("name")}" type="{el.get("type")}">')
            for sub in el:
                if sub.tag in ("origin", "parent", "child", "axis", "limit"):
                    attrs = " ".join(f'{k}="{v}"' for k, v in sub.attrib.items())
                    lines.append(f"    <{sub.tag} {attrs}/>")
            lines.append("  </joint>")
    lines.append("</robot>")
    open(OUT, "w").write("\n".join(lines) + "\n")
    print("wrote", OUT)


if __name__ == "__main__":
    main()
