import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, "robot-control-stack_amd")
import parity_util as pu
for kick in (False, True):
    rep = pu.run_free_box_parity(n_envs=32, n_calls=12, k=25, seed=3, kick=kick)
    print(kick, {k: v for k, v in rep.items() if k != "final_z"})
    print(rep["final_z"].round(5))
