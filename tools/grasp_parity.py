"""dev tool: scripted pinch-and-lift of the pick-up scene's cube, HIP kernel vs oracle, stage by stage.

    python tools/grasp_parity.py [n_envs]
"""
import os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "robot-control-stack_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import rcs_oracle as O
from rcs_amd import sim as S
from rcs_amd.common import Pose
from rcs_amd.envs import default_sim_gripper_cfg, default_sim_robot_cfg
from rcs_amd.mjcf import compile_mjcf
from rcs_env_oracle import FR3_Q_HOME

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
import dataclasses
from rcs_amd import common
cfg = dataclasses.replace(default_sim_robot_cfg("fr3_simple_pick_up"), tcp_offset=common.Pose(common.FrankaHandTCPOffset()))
simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=n)
robot = S.SimRobot(simu, None, cfg)
grip = S.SimGripper(simu, default_sim_gripper_cfg())
cm = compile_mjcf(cfg.mjcf_scene_path.replace(".mjb", ".xml"))
arm = [f"fr3_joint{i}_0" for i in range(1, 8)]
os_ = [O.Sim(cm, arm, arm, "attachment_site_0", "base_0", FR3_Q_HOME, O.franka_hand_tcp_offset(), "finger_joint1_0", "actuator8_0", resolve_contacts=True) for _ in range(n)]
rng = np.random.default_rng(0)
# cube placements: a few mm / degrees off the gripper's axis
qb = np.tile(np.array([0.44, 0.1, 0.0288, 0, 0, 0, 1.0]), (n, 1))
qb[1:, 0] += rng.uniform(-0.004, 0.004, n - 1)
qb[1:, 1] += rng.uniform(-0.004, 0.004, n - 1)
yaw = np.zeros(n); yaw[1:] = rng.uniform(-0.1, 0.1, n - 1)
qb[:, 3] = np.cos((np.pi + yaw) / 2); qb[:, 6] = np.sin((np.pi + yaw) / 2)

simu.reset(); robot.reset(); grip.reset()
for o in os_:
    o.reset(); o.robot_reset(); o.gripper_reset()
simu.set_free_joint_qpos("box_joint", qb)
for e, o in enumerate(os_):
    o.box_qpos = qb[e]
simu.step(1); [o.step(1) for o in os_]

def compare(tag, k):
    t0 = time.time(); simu.step(k); tk = time.time() - t0
    t0 = time.time(); [o.step(k) for o in os_]; to = time.time() - t0
    q, v, bq, bv = simu.qpos, simu.qvel, simu.free_joint_qpos("box_joint"), simu.free_joint_qvel("box_joint")
    dq = max(np.abs(q[e] - np.asarray(o.qpos)).max() for e, o in enumerate(os_))
    dv = max(np.abs(v[e] - np.asarray(o.qvel)).max() for e, o in enumerate(os_))
    db = max(np.abs(bq[e] - o.box_qpos).max() for e, o in enumerate(os_))
    dbv = max(np.abs(bv[e] - o.box_qvel).max() for e, o in enumerate(os_))
    d = os_[0].s.d
    print(f"{tag:10s} k={k:4d} dq={dq:.2e} dv={dv:.2e} dbox={db:.2e} dboxv={dbv:.2e} | oracle: ncon={d.ncon} coupled={d.coupled} newton={d.solver_niter} noslip={d.noslip_niter} "
          f"box z={[round(float(o.box_qpos[2]), 4) for o in os_]} kernel z={np.round(bq[:, 2], 4)} | kernel {tk*1e3:.1f} ms oracle {to*1e3:.0f} ms", flush=True)

home = os_[0].get_cartesian_position()
def move(xyz):
    robot.set_cartesian_position(np.tile(np.concatenate([xyz, home.rotation_q()]), (n, 1)))
    for o in os_:
        o.set_cartesian_position(O.Pose(translation=np.array(xyz), quaternion=home.rotation_q()))

grip.open(); [o.gripper_open() for o in os_]
compare("settle", 100)
move([0.44, 0.1, 0.20]); compare("above", 500)
move([0.44, 0.1, 0.035])
for i in range(7):
    compare(f"down{i}", 100)
grip.shut(); [o.gripper_grasp() for o in os_]
for i in range(6):
    compare(f"close{i}", 50)
move([0.44, 0.1, 0.30])
for i in range(8):
    compare(f"lift{i}", 100)
grip.open(); [o.gripper_open() for o in os_]
for i in range(4):
    compare(f"release{i}", 100)
simu.close()
