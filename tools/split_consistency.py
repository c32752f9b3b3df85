"""dev tool: the same scripted pinch at batch scale with the stepping cut into launches of different lengths, kernel against
kernel.  The coupled solve's starting point inside a launch differs from a fresh launch's, so equal results for every
environment say that every solve of every environment reached its minimiser (no stalled or capped Newton solve anywhere).

    python tools/split_consistency.py [n_envs] [chunk_a] [chunk_b]
"""
import dataclasses, os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "robot-control-stack_amd"), os.path.join(ROOT, "tests")]
from rcs_amd import common
from rcs_amd import sim as S
from rcs_amd.envs import default_sim_gripper_cfg, default_sim_robot_cfg


def pinch(n, chunk, seed=0, spread=0.004, yaw=0.1):
    cfg = dataclasses.replace(default_sim_robot_cfg("fr3_simple_pick_up"), tcp_offset=common.Pose(common.FrankaHandTCPOffset()))
    simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=n)
    robot = S.SimRobot(simu, None, cfg)
    grip = S.SimGripper(simu, default_sim_gripper_cfg())
    rng = np.random.default_rng(seed)
    qb = np.tile(np.array([0.44, 0.1, 0.0288, 0, 0, 0, 1.0]), (n, 1))
    qb[1:, 0] += rng.uniform(-spread, spread, n - 1)
    qb[1:, 1] += rng.uniform(-spread, spread, n - 1)
    y = np.zeros(n); y[1:] = rng.uniform(-yaw, yaw, n - 1)
    qb[:, 3], qb[:, 6] = np.cos((np.pi + y) / 2), np.sin((np.pi + y) / 2)
    simu.reset(); robot.reset(); grip.reset()
    simu.set_free_joint_qpos("box_joint", qb)
    simu.step(1)
    home = np.asarray(robot.get_cartesian_position())[0, 3:]

    def run(k):
        for i in range(0, k, chunk):
            simu.step(min(chunk, k - i))

    out = {}
    robot.set_cartesian_position(np.tile(np.concatenate([[0.44, 0.1, 0.2], home]), (n, 1))); run(400)
    robot.set_cartesian_position(np.tile(np.concatenate([[0.44, 0.1, 0.035], home]), (n, 1))); run(600)
    grip.shut(); run(200)
    out["closed"] = (simu.qpos.copy(), simu.free_joint_qpos("box_joint").copy())
    robot.set_cartesian_position(np.tile(np.concatenate([[0.44, 0.1, 0.3], home]), (n, 1))); run(500)
    out["lifted"] = (simu.qpos.copy(), simu.free_joint_qpos("box_joint").copy())
    grip.open(); run(300)
    out["released"] = (simu.qpos.copy(), simu.free_joint_qpos("box_joint").copy())
    simu.close()
    return out


def compare(n=4096, ca=17, cb=100, **kw):
    a, b = pinch(n, ca, **kw), pinch(n, cb, **kw)
    rep = {}
    for tag in a:
        dq = np.abs(a[tag][0] - b[tag][0]).max(axis=1)
        db = np.abs(a[tag][1] - b[tag][1]).max(axis=1)
        rep[tag] = {"max_dq": float(dq.max()), "max_dbox": float(db.max()), "envs_over_1e-8": int(((dq > 1e-8) | (db > 1e-8)).sum()),
                    "worst_env": int(np.argmax(np.maximum(dq, db))), "box_z": (float(a[tag][1][:, 2].min()), float(a[tag][1][:, 2].max()))}
    return rep


def newton_stats(n, chunk, **kw):
    """(library built with tools/build_timing.sh) outcomes of the coupled Newton solves of one run, all environments"""
    import ctypes as C
    import rcs_amd._lib as lib
    lib.LIB_PATH = os.path.join(ROOT, "robot-control-stack_amd", "rcs_amd", "librcs_hip_timing.so")
    L = lib.load()
    out = (C.c_ulonglong * 64)()
    L.rcsh_debug_team_cycles64(out); base = list(out)
    pinch(n, chunk, **kw)
    L.rcsh_debug_team_cycles64(out)
    d = [out[i] - base[i] for i in range(64)]
    return {"solves": d[60], "worst_iterations": out[56], "over_20_iterations": d[57], "ran_into_the_cap": d[58], "left_on_a_non_descent_direction": d[59]}


if __name__ == "__main__":
    if "--newton-stats" in sys.argv:
        sys.argv.remove("--newton-stats")
        a = [float(x) for x in sys.argv[1:]]
        print(newton_stats(int(a[0]) if a else 4096, 17, seed=int(a[1]) if len(a) > 1 else 0, spread=a[2] if len(a) > 2 else 0.004, yaw=a[3] if len(a) > 3 else 0.1))
        sys.exit(0)
    args = [int(x) for x in sys.argv[1:]]
    n, ca, cb = (args + [4096, 17, 100])[:3] if len(args) < 3 else args[:3]
    for tag, r in compare(n, ca, cb).items():
        print(tag, r)
