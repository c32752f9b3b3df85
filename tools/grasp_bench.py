"""dev tool: throughput of the contact-rich regime -- every environment of the batch pinches, lifts and holds the cube at once
(fine-grained API, physics substeps; the pick-up task's random actions of `bench.py --task pick_up` rarely touch it).

    python tools/grasp_bench.py [n_envs] [--oracle] [--xarm7]     (--oracle: time the CPU restatement on the same script, one environment;
                                                                 --xarm7: scenes/xarm7_pick_world -- BASELINE configs[3] -- instead of the FR3)
"""
import dataclasses, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "robot-control-stack_amd"), os.path.join(ROOT, "tests")]
import rcs_amd._lib as _lib
if os.environ.get("RCSH_LIB"): _lib.LIB_PATH = os.environ["RCSH_LIB"]
from rcs_amd import common
from rcs_amd import sim as S
from rcs_amd.envs import default_sim_gripper_cfg, default_sim_robot_cfg

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if args else 4096
XARM7 = "--xarm7" in sys.argv
if XARM7:
    from rcs_amd.envs import xarm7_pick_sim_gripper_cfg, xarm7_pick_sim_robot_cfg
    cfg, gcfg = xarm7_pick_sim_robot_cfg(), xarm7_pick_sim_gripper_cfg()
    CUBE, BASE_Z = np.array([0.40, 0.0]), 0.12  # (targets are in the robot frame: the xArm7's base sits 0.12 m above the floor)
else:
    cfg, gcfg = dataclasses.replace(default_sim_robot_cfg("fr3_simple_pick_up"), tcp_offset=common.Pose(common.FrankaHandTCPOffset())), default_sim_gripper_cfg()
    CUBE, BASE_Z = np.array([0.44, 0.1]), 0.0
simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=n)
robot = S.SimRobot(simu, None, cfg)
grip = S.SimGripper(simu, gcfg)
rng = np.random.default_rng(0)
qb = np.tile(np.array([CUBE[0], CUBE[1], 0.0288, 0, 0, 0, 1.0]), (n, 1))
qb[1:, 0] += rng.uniform(-0.004, 0.004, n - 1)
qb[1:, 1] += rng.uniform(-0.004, 0.004, n - 1)
yaw = np.zeros(n); yaw[1:] = rng.uniform(-0.1, 0.1, n - 1)
qb[:, 3], qb[:, 6] = np.cos((np.pi + yaw) / 2), np.sin((np.pi + yaw) / 2)
simu.reset(); robot.reset(); grip.reset()
simu.set_free_joint_qpos("box_joint", qb)
simu.step(1)
home = common.Pose(rotation=np.diag([1.0, -1.0, -1.0])).rotation_q() if XARM7 else np.asarray(robot.get_cartesian_position())[0, 3:]
rows = []
def stage(tag, k, mv=None, g=None):
    if mv is not None:
        robot.set_cartesian_position(np.tile(np.concatenate([mv, home]), (n, 1)))
    if g is not None:
        (grip.shut if g == 0 else grip.open)()
    simu.qpos  # (drain)
    t0 = time.perf_counter(); simu.step(k); z = simu.free_joint_qpos("box_joint")[:, 2]; dt = time.perf_counter() - t0
    rows.append((tag, k, dt))
    print(f"{tag:8s} {k:4d} substeps  {dt * 1e3:8.1f} ms  {n * k / dt / 1e6:7.2f} M substeps/s = {n * k / 17 / dt / 1e3:8.1f} k env-steps/s at 17 substeps;  cube z {z.min():.3f} .. {z.max():.3f}", flush=True)
at = lambda z: [CUBE[0], CUBE[1], z - BASE_Z]  # noqa: E731
stage("above", 500, mv=at(0.2))
stage("down", 700, mv=at(0.035))
stage("closing", 250, g=0)
stage("lifting", 600, mv=at(0.3))
stage("held", 200)
stage("released", 300, g=1)
simu.close()

if "--oracle" in sys.argv and not XARM7:
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import rcs_oracle as O
    from rcs_amd.mjcf import compile_mjcf
    from rcs_env_oracle import FR3_Q_HOME
    cm = compile_mjcf(cfg.mjcf_scene_path.replace(".mjb", ".xml"))
    arm = [f"fr3_joint{i}_0" for i in range(1, 8)]
    o = O.Sim(cm, arm, arm, "attachment_site_0", "base_0", FR3_Q_HOME, O.franka_hand_tcp_offset(), "finger_joint1_0", "actuator8_0", resolve_contacts=True)
    o.reset(); o.robot_reset(); o.gripper_reset(); o.box_qpos = qb[0]; o.step(1)
    hp = o.get_cartesian_position()
    def ostage(tag, k, mv=None, g=None):
        if mv is not None:
            o.set_cartesian_position(O.Pose(translation=np.array(mv), quaternion=hp.rotation_q()))
        if g is not None:
            (o.gripper_grasp if g == 0 else o.gripper_open)()
        t0 = time.perf_counter(); o.step(k); dt = time.perf_counter() - t0
        print(f"oracle {tag:8s} {k:4d} substeps {dt * 1e3:8.1f} ms  {k / dt / 1e3:7.1f} k substeps/s on one core", flush=True)
    ostage("above", 400, mv=[0.44, 0.1, 0.2]); ostage("down", 600, mv=[0.44, 0.1, 0.035]); ostage("closing", 200, g=0)
    ostage("lifting", 500, mv=[0.44, 0.1, 0.3]); ostage("held", 200); ostage("released", 300, g=1)
