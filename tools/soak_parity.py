"""dev tool: longer / wider runs of the free-box and pick-task parity checks than the test suite affords."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "robot-control-stack_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import parity_util as pu
OFF = int(os.environ.get("SOAK_SEED_OFFSET", "0"))  # other seeds, same checks
for seed in range(OFF, OFF + 4):
    rep = pu.run_free_box_parity(n_envs=64, n_calls=20, k=25, seed=seed)
    print("box", seed, {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in rep.items() if k not in ("final_z", "zones")})
for seed in range(OFF, OFF + 3):
    rep = pu.run_pick_task_parity(n_envs=32, n_steps=15, seed=seed, episodes=2)
    print("task", seed, {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in rep.items()})
rep = pu.run_depth_render_parity(n_envs=8, width=96, height=64, seed=5 + OFF, n_calls=4)
print("depth", {k: v for k, v in rep.items() if k != "sample"})
import time
for kernel in ("team",):
    pu.KERNEL = kernel
    t0 = time.time()
    rep = pu.run_joint_rollout_parity(n_envs=192, n_steps=40, async_control=True, seed=100 + OFF)
    print(kernel, "joints async", {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in rep.items()}, f"{time.time() - t0:.0f}s")
    rep = pu.run_joint_rollout_parity(n_envs=64, n_steps=6, async_control=False, seed=101 + OFF)
    print(kernel, "joints conv ", {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in rep.items()})
    rep = pu.run_cartesian_rollout_parity(n_envs=96, n_steps=12, async_control=True, seed=102 + OFF, mode="xyzrpy")
    print(kernel, "cartesian   ", {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in rep.items()})
pu.KERNEL = "team"
rep = pu.run_joint_rollout_parity(n_envs=96, n_steps=20, async_control=True, seed=103 + OFF, robot="xarm7")
print("xarm7 async", {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in rep.items()})
rep = pu.run_joint_rollout_parity(n_envs=96, n_steps=20, async_control=True, seed=104 + OFF, robot="arm6")
print("arm6 async", {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in rep.items()})
for robot in ("ur5e", "so101"):
    rep = pu.run_joint_rollout_parity(n_envs=96, n_steps=20, async_control=True, seed=105 + OFF, robot=robot)
    print(robot, "async", {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in rep.items()})
    rep = pu.run_joint_rollout_parity(n_envs=48, n_steps=5, async_control=False, seed=106 + OFF, robot=robot)
    print(robot, "conv ", {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in rep.items()})
for seed in (2 + OFF, 3 + OFF):
    rep = pu.run_self_collision_parity(n_envs=96, seed=seed)
    print("self collision", seed, rep)
for seed in (1 + OFF, 2 + OFF):
    rep = pu.run_grasp_parity(n_envs=8, seed=seed) if "seed" in pu.run_grasp_parity.__code__.co_varnames else pu.run_grasp_parity(n_envs=8)
    print("grasp", seed, {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in rep.items() if not hasattr(v, "shape")})
rep = pu.run_rate_driven_camera_parity(n_envs=8, seed=9 + OFF)
print("rate-driven cameras", {k: v for k, v in rep.items() if k != "debug"})
# ---- round 3: the cube thrown at the base over more seeds (support ties broken by rule), the xArm7 pick-up over more seeds,
# the first-principles vectors through the kernel
for seed in [x + OFF for x in (5, 11, 12, 13, 14, 15, 21, 22)]:
    rep = pu.run_cube_against_base_parity(seed=seed)
    print("cube vs base seed", seed, {"worst_env_pos_err": f"{rep['env_pos_err'].max():.2e}", "max_abs_quat": f"{rep['max_abs_quat']:.2e}", "base_contact_envs": rep["base_contact_envs"], "max_ncon": rep["max_ncon"]})
for seed in range(OFF, OFF + 4):
    rep = pu.run_xarm7_pick_parity(n_envs=6, seed=seed)
    print("xarm7 pick seed", seed, {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in rep.items() if k != "stages"},
          "lifted z", [round(float(z), 4) for z in rep["stages"]["held"]["box_z"]])
print("elapsed", time.time() - t0 if "t0" in dir() else "")
