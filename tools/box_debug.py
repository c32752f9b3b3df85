import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, "robot-control-stack_amd")
import parity_util as pu
print(pu.run_pick_task_parity(n_envs=16, n_steps=6, seed=1, episodes=2))
