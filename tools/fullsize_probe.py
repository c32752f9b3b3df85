import sys, os, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [os.path.join(ROOT, "robot-control-stack_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import parity_util as pu
for n, k in ((1024, 3), (4096, 3)):
    t0 = time.time()
    rep = pu.run_joint_rollout_parity(n_envs=n, n_steps=k, async_control=True, seed=7)
    print(n, k, {a: (f"{v:.2e}" if isinstance(v, float) else v) for a, v in rep.items()}, f"{time.time() - t0:.1f}s", flush=True)
