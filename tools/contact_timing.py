"""dev tool: cycle counts of the contact phase (contact_team.h) during a scripted pinch (library built with tools/build_timing.sh)."""
import ctypes as C, dataclasses, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "robot-control-stack_amd"), os.path.join(ROOT, "tests")]
import rcs_amd._lib as lib
lib.LIB_PATH = os.path.join(ROOT, "robot-control-stack_amd", "rcs_amd", "librcs_hip_timing.so")
import numpy as np
from rcs_amd import common
from rcs_amd import sim as S
from rcs_amd.envs import default_sim_gripper_cfg, default_sim_robot_cfg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = dataclasses.replace(default_sim_robot_cfg("fr3_simple_pick_up"), tcp_offset=common.Pose(common.FrankaHandTCPOffset()))
simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=n)
robot = S.SimRobot(simu, None, cfg)
grip = S.SimGripper(simu, default_sim_gripper_cfg())
simu.reset(); robot.reset(); grip.reset(); simu.step(1)
home = robot.get_cartesian_position()
q = np.asarray(home)[0, 3:]
out = (C.c_ulonglong * 64)()
# TEAM_MARK(i) adds the cycles since the previous mark to slot i; TEAM_COUNT(i) counts
COLLIDE = ((37, "link frames"), (38, "lane per geom"), (39, "hulls, whole wavefront"), (25, "floor corners, fast-path test"))
NEWTON = ((55, "x update (+ start)"), (48, "rows + gradient"), (49, "stiffness sums"), (50, "Hessian"), (51, "row loads"), (52, "LDL + solves"),
          (53, "before line search"), (54, "line search"))
NOSLIP = ((40, "rel across lanes"), (41, "owner"), (44, "slot lanes"))
def stage(tag, k, mv=None, g=None):
    if mv is not None:
        robot.set_cartesian_position(np.tile(np.concatenate([mv, q]), (n, 1)))
    if g is not None:
        (grip.shut if g == 0 else grip.open)()
    simu._L.rcsh_debug_team_cycles64(out); base = np.array(out[:], dtype=np.float64)
    simu.step(k)
    simu._L.rcsh_debug_team_cycles64(out); a = np.array(out[:], dtype=np.float64) - base
    calls, coupled = max(a[33], 1), max(a[34], 1)
    print(f"{tag}: {k} substeps, contact phases {a[33]:.0f} (coupled {a[34]:.0f}); box z {simu.free_joint_qpos('box_joint')[0, 2]:.3f}")
    if a[33] == 0:
        return
    print(f"    collision {sum(a[i] for i, _ in COLLIDE) / calls:8.0f} cycles per phase: " + ", ".join(f"{nm} {a[i] / calls:.0f}" for i, nm in COLLIDE))
    if a[34] == 0:
        return
    newton = a[28] + sum(a[i] for i, _ in NEWTON)
    noslip = a[31] + sum(a[i] for i, _ in NOSLIP)
    total = a[26] + a[27] + newton + a[30] + noslip + a[32]
    print(f"    coupled solve {total / coupled:8.0f} cycles per coupled phase: compaction {a[26] / coupled:.0f}, rows / qacc_smooth / M factor {a[27] / coupled:.0f}, "
          f"Newton {newton / coupled:.0f}, forces / Y / K blocks {a[30] / coupled:.0f}, noslip {noslip / coupled:.0f}, results {a[32] / coupled:.0f}")
    ni, nc = max(a[29], 1), max(a[45], 1)
    print(f"    Newton: {a[29] / coupled:.2f} iterations, {a[35] / coupled:.2f} line-search evaluations per coupled phase; per iteration {sum(a[i] for i, _ in NEWTON) / ni:.0f} cycles: "
          + ", ".join(f"{nm} {a[i] / ni:.0f}" for i, nm in NEWTON))
    print(f"    noslip: {a[36] / coupled:.2f} sweeps of {a[45] / max(a[36], 1):.1f} contacts per coupled phase; per contact update {sum(a[i] for i, _ in NOSLIP) / nc:.0f} cycles: "
          + ", ".join(f"{nm} {a[i] / nc:.0f}" for i, nm in NOSLIP))
stage("above", 400, mv=[0.44, 0.1, 0.2])
stage("down", 600, mv=[0.44, 0.1, 0.035])
stage("closing", 100, g=0)
stage("closed", 100)
stage("lifting", 300, mv=[0.44, 0.1, 0.3])
stage("held", 100)
