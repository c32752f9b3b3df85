import os, sys, numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [os.path.join(ROOT, "robot-control-stack_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
from parity_util import make_vec_env, make_oracle_envs, synthetic_actions
n, T = 96, 6
venv = make_vec_env(n, True); oenvs = make_oracle_envs(n, True)
j, g = synthetic_actions(n, T, 1)
venv.reset(); [o.reset() for o in oenvs]
np.set_printoptions(precision=12, linewidth=200)
for t in range(T):
    venv.step({"joints": j[t], "gripper": g[t]})
    for e, o in enumerate(oenvs): o.step({"joints": j[t, e], "gripper": g[t, e]})
    q = venv.sim.qpos; v = venv.sim.qvel
    d = np.array([np.abs(q[e] - oenvs[e].sim.qpos).max() for e in range(n)])
    bad = np.argsort(-d)[:3]
    print("step", t, "max", d.max(), "envs", bad, d[bad])
    for e in bad[:1]:
        print("  hip q", q[e][6:], "v", v[e][6:], "ctrl", venv.sim.ctrl[e][7], "grip act", g[:t+1, e].round())
        print("  orc q", oenvs[e].sim.qpos[6:], "v", oenvs[e].sim.qvel[6:])
