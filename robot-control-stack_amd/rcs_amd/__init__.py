"""rcs_amd -- MI355X-native batched simulation backend behind the RCS ``rcs.sim`` / ``rcs.envs`` surface.

Mirrors the reference package layout (reference python/rcs/__init__.py): ``common``, ``sim``,
``envs`` and the ``scenes`` registry.
"""

from __future__ import annotations

import os
from dataclasses import dataclass

from . import _lib, camera, common, mjcf, render, sim  # noqa: F401
from . import envs  # noqa: F401,E402

__version__ = "0.1.0"

_SCENES_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scenes")


@dataclass(kw_only=True)
class Scene:  # reference python/rcs/__init__.py:17-31
    mjb: str
    mjcf_scene: str
    mjcf_robot: str
    urdf: str | None = None
    robot_type: common.RobotType


def _scene(name: str, robot_type: common.RobotType) -> Scene:
    d = os.path.join(_SCENES_DIR, name)
    xml = os.path.join(d, "scene.xml")
    # the reference registers the MuJoCo-compiled scene.mjb; this backend compiles the .xml itself,
    # so `mjb` aliases the xml (Sim() maps a .mjb suffix back to .xml)
    return Scene(mjb=xml, mjcf_scene=xml, mjcf_robot=xml, urdf=None, robot_type=robot_type)


scenes: dict[str, Scene] = {
    "fr3_empty_world": _scene("fr3_empty_world", common.RobotType.FR3),
    "fr3_simple_pick_up": _scene("fr3_simple_pick_up", common.RobotType.FR3),
    "xarm7_empty_world": _scene("xarm7_empty_world", common.RobotType.XArm7),
    "xarm7_box_world": _scene("xarm7_box_world", common.RobotType.XArm7),
    "xarm7_pick_world": _scene("xarm7_pick_world", common.RobotType.XArm7),
    "arm6_empty_world": _scene("arm6_empty_world", common.RobotType.UR5e),
    "ur5e_empty_world": _scene("ur5e_empty_world", common.RobotType.UR5e),
    "so101_empty_world": _scene("so101_empty_world", common.RobotType.SO101),
}

__all__ = ["__version__", "common", "sim", "envs", "scenes", "mjcf", "camera", "render"]
