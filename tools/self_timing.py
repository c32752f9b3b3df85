"""dev tool: cycle counts of the self-collision test of the DET kernels (library built with tools/build_timing.sh):
step_until_convergence of fr3_empty_world, per call of self_collision_pairs (workgroup 0)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "robot-control-stack_amd"), os.path.join(ROOT, "tests")]
import rcs_amd._lib as lib
lib.LIB_PATH = os.path.join(ROOT, "robot-control-stack_amd", "rcs_amd", "librcs_hip_timing.so")
import numpy as np
from rcs_amd import sim as S
from rcs_amd.envs import default_sim_gripper_cfg, default_sim_robot_cfg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = default_sim_robot_cfg("fr3_empty_world")
simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=n)
robot = S.SimRobot(simu, None, cfg)
grip = S.SimGripper(simu, default_sim_gripper_cfg())
simu.step(1)
out = (C.c_ulonglong * 48)()
rng = np.random.default_rng(0)
q = np.asarray(robot.get_joint_position()) + rng.uniform(-0.09, 0.09, (n, 7))
simu._L.rcsh_debug_team_cycles48(out); base = np.array(out[:], dtype=np.float64)
robot.set_joint_position(q)
simu.step_until_convergence()
simu._L.rcsh_debug_team_cycles48(out); a = np.array(out[:], dtype=np.float64) - base
calls = max(a[47], 1)
print(f"substeps {simu.convergence_steps()[:4]}, calls of the pair test (workgroup 0): {a[47]:.0f}, narrow-phase candidates {a[46]:.0f}")
print(f"  before (since last mark)      {a[42] / calls:10.0f}")
print(f"  slack test + spheres          {a[40] / calls:10.0f}")
print(f"  oriented boxes                {a[43] / calls:10.0f}   (ran in {a[41]:.0f} of {a[47]:.0f} calls)")
print(f"  staging  (per candidate)      {a[44] / max(a[46], 1):10.0f}")
print(f"  frames + shapes (per cand.)   {a[37] / max(a[46], 1):10.0f}")
print(f"  remembered direction          {a[38] / max(a[46], 1):10.0f}   (settled {a[39]:.0f} of {a[46]:.0f})")
print(f"  portal refinement (per cand.) {a[45] / max(a[46], 1):10.0f}")
if len(sys.argv) > 2:
    print("all slots (cycles, workgroup 0):", {i: int(a[i]) for i in range(48) if a[i]}, "sum of marks", int(sum(a[i] for i in range(48) if i not in (29, 33, 34, 35, 36, 39, 41, 46, 47))))
