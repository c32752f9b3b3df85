"""Enums of the reference Gym layer (reference python/rcs/envs/base.py:167-171,344-347)."""

from enum import Enum, auto


class ControlMode(Enum):
    JOINTS = auto()
    CARTESIAN_TRPY = auto()
    CARTESIAN_TQuat = auto()


class RelativeTo(Enum):
    LAST_STEP = auto()
    CONFIGURED_ORIGIN = auto()
