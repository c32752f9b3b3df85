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
out = (C.c_ulonglong * 48)()
names = {24: "before (since last mark)", 37: "  link frames", 38: "  lane per geom", 39: "  hulls, cooperative", 25: "frames + narrow phase (rest)", 26: "compaction", 27: "rows, qacc_smooth, M factor", 28: "Newton", 30: "forces, Y, A blocks",
         31: "noslip sweeps", 32: "results"}
def stage(tag, k, mv=None, g=None):
    if mv is not None:
        robot.set_cartesian_position(np.tile(np.concatenate([mv, q]), (n, 1)))
    if g is not None:
        (grip.shut if g == 0 else grip.open)()
    simu._L.rcsh_debug_team_cycles48(out); base = np.array(out[:], dtype=np.float64)
    simu.step(k)
    simu._L.rcsh_debug_team_cycles48(out); a = np.array(out[:], dtype=np.float64) - base
    calls, coupled = max(a[33], 1), max(a[34], 1)
    print(f"{tag}: {k} substeps, contact phases {a[33]:.0f} (coupled {a[34]:.0f}); box z {simu.free_joint_qpos('box_joint')[0, 2]:.3f}")
    if a[33] > 0:
        for i, nm in names.items():
            print(f"    {nm:30s} {a[i] / (coupled if 26 < i < 37 else calls):10.0f} cycles per {'coupled ' if 26 < i < 37 else ''}phase")
        nc = max(a[45], 1)
        print(f"    noslip per contact update ({a[45] / max(a[36], 1):.1f} contacts per sweep): loop head {a[40] / nc:.0f}, owner {a[41] / nc:.0f}, dx {a[42] / nc:.0f}, body_spatial {a[43] / nc:.0f}, accumulate {a[44] / nc:.0f} cycles")
        print(f"    Newton iterations {a[29] / coupled:.2f}, line-search evaluations {a[35] / coupled:.2f}, noslip sweeps {a[36] / coupled:.2f} per coupled phase")
stage("above", 400, mv=[0.44, 0.1, 0.2])
stage("down", 600, mv=[0.44, 0.1, 0.035])
stage("closing", 100, g=0)
stage("closed", 100)
stage("lifting", 300, mv=[0.44, 0.1, 0.3])
stage("held", 100)
