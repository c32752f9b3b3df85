"""dev tool: where a substep of the pinch goes (every environment holds the cube; library built with tools/build_timing.sh --
workgroup 0's cycle counters between the marks, tools/esc_timing.py's names).    python tools/grasp_timing.py [n_envs]"""
import ctypes as C, dataclasses, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "robot-control-stack_amd"), os.path.join(ROOT, "tests")]
import rcs_amd._lib as _lib
_lib.LIB_PATH = os.path.join(ROOT, "robot-control-stack_amd", "rcs_amd", "librcs_hip_timing.so")
from rcs_amd import common
from rcs_amd import sim as S
from rcs_amd.envs import default_sim_gripper_cfg, default_sim_robot_cfg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg, gcfg = dataclasses.replace(default_sim_robot_cfg("fr3_simple_pick_up"), tcp_offset=common.Pose(common.FrankaHandTCPOffset())), default_sim_gripper_cfg()
CUBE = np.array([0.44, 0.1])
simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=n)
robot = S.SimRobot(simu, None, cfg)
grip = S.SimGripper(simu, gcfg)
qb = np.tile(np.array([CUBE[0], CUBE[1], 0.0288, 0, 0, 0, 1.0]), (n, 1))
qb[:, 3], qb[:, 6] = np.cos(np.pi / 2), np.sin(np.pi / 2)
simu.reset(); robot.reset(); grip.reset()
simu.set_free_joint_qpos("box_joint", qb)
simu.step(1)
home = np.asarray(robot.get_cartesian_position())[0, 3:]
L = simu._L
out = (C.c_ulonglong * 96)()
def read():
    L.rcsh_debug_team_cycles96(out, 1)
    return np.array(out[:], dtype=np.float64)
NAMES = {0: "pos stage", 1: "1", 2: "2", 3: "3", 4: "4", 15: "15", 5: "5", 6: "6", 7: "7", 8: "8", 9: "loop tail", 10: "epilogue+check", 11: "11", 12: "prologue12", 13: "13", 14: "14",
         16: "box 16", 17: "box 17", 18: "box 18", 19: "19", 20: "20", 21: "21", 22: "22", 23: "23", 24: "before collide", 37: "link frames", 38: "lane per geom", 39: "hulls wavefront", 25: "floor/fastpath", 26: "compaction", 27: "rows/qacc_smooth/M", 28: "newton pre", 55: "x update", 48: "rows+grad",
         49: "stiffness", 50: "Hessian", 51: "row loads", 52: "LDL+solves", 53: "pre linesearch", 54: "linesearch", 30: "forces/Y/K", 31: "noslip rest", 40: "ns rel", 41: "ns owner", 44: "ns slots", 32: "results", 61: "61", 62: "62", 63: "63"}
def stage(tag, k, mv=None, g=None):
    if mv is not None:
        robot.set_cartesian_position(np.tile(np.concatenate([mv, home]), (n, 1)))
    if g is not None:
        (grip.shut if g == 0 else grip.open)()
    simu.qpos
    base = read()
    simu.step(k)
    z = simu.free_joint_qpos("box_joint")[:, 2]
    d = read() - base
    tot = sum(d[i] for i in NAMES)
    print(f"{tag}: cube z {z.min():.3f}..{z.max():.3f}; workgroup 0: contact phases {d[33]:.0f} ({d[33] / k:.1f} per substep), coupled {d[34]:.0f}, Newton iterations {d[29]:.0f}, line search evaluations {d[35]:.0f}, noslip sweeps {d[36]:.0f} / contact updates {d[45]:.0f}")
    print("   marked cycles per substep %.0f: " % (tot / k) + ", ".join(f"{NAMES[i]} {d[i] / k:.0f}" for i in NAMES if d[i] > 0))
at = lambda z: [CUBE[0], CUBE[1], z]  # noqa: E731
stage("above", 500, mv=at(0.2))
stage("down", 700, mv=at(0.035))
stage("closing", 250, g=0)
stage("lifting", 600, mv=at(0.3))
stage("held", 200)
simu.close()
