"""dev tool: CLIK iteration statistics of the Cartesian bench workload (SURVEY 8d config 3 actions)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "robot-control-stack_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
from parity_util import make_vec_env, cartesian_actions
from rcs_amd.envs import ControlMode
from rcs_amd.common import Pose
import rcs_oracle as O

n, T = 1024, 12
for async_control in (True, False):
    venv = make_vec_env(n, async_control, gripper=True, relative=True, control_mode=ControlMode.CARTESIAN_TRPY,
                        max_relative_movement=(0.2, float(np.deg2rad(45))))
    act, grip = cartesian_actions(n, T, 0)
    obs, info = venv.reset()
    ik = venv.robot.get_ik()
    tcp = O.franka_hand_tcp_offset()
    tcp7 = np.concatenate([tcp.translation(), tcp.rotation_q()])
    for t in range(T):
        # the target the env will hand to the IK: last commanded pose shifted by the action
        obs, _, _, trunc, info = venv.step({"xyzrpy": act[t], "gripper": grip[t]})
        if t in (0, 3, 7, 11):
            q0 = venv.robot.get_joint_position()
            cur = venv.robot.get_cartesian_position()  # [n,7]
            tgt = cur.copy(); tgt[:, :3] += act[t, :, :3]
            q, ok, iters = ik.inverse(tgt, q0, tcp7)
            print(f"async={async_control} step {t}: ik_success(env) {info['ik_success'].mean():.3f}  probe: ok {ok.mean():.3f} "
                  f"iters mean {iters.mean():.0f} median {np.median(iters):.0f} p90 {np.percentile(iters, 90):.0f} max {iters.max()}")
    venv.close()
