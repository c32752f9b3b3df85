"""dev tool: what the contact phase costs when nothing touches -- the pick-up scene stepped with the robot at home and the cube on
the floor beside it, contacts resolved (k_run_team<..., BOX, CON>) against detected only (k_run_team<..., BOX>); and the empty world."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "robot-control-stack_amd")]
import numpy as np
from rcs_amd import sim as S
from rcs_amd.envs import default_sim_gripper_cfg, default_sim_robot_cfg

n, k = 4096, 1700
for scene, resolve in (("fr3_empty_world", None), ("fr3_simple_pick_up", False), ("fr3_simple_pick_up", True)):
    cfg = default_sim_robot_cfg(scene)
    simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=n, resolve_robot_contacts=resolve)
    S.SimRobot(simu, None, cfg)
    S.SimGripper(simu, default_sim_gripper_cfg())
    simu.step(200); simu.qpos
    t0 = time.perf_counter(); simu.step(k); simu.qpos; dt = time.perf_counter() - t0
    print(f"{scene:20s} resolve={resolve}: {dt / k * 1e6:7.2f} us per substep of {n} environments = {dt / k * 2.4e9 / 1:9.0f} cycles at 2.4 GHz; {n * k / dt / 1e6:7.1f} M substeps/s", flush=True)
    simu.close()
