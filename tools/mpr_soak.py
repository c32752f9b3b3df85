"""dev tool: the cube thrown at the robot's base (hull-vs-box portal refinement) over several seeds, per-environment errors."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "robot-control-stack_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import parity_util as pu

for seed in [int(x) for x in sys.argv[1:]] or [5, 11, 12, 13, 14, 15]:
    rep = pu.run_cube_against_base_parity(seed=seed)
    err = np.sort(rep["env_pos_err"])[::-1]
    print("cube vs base seed", seed, "contact envs", rep["base_contact_envs"], "max_ncon", rep["max_ncon"], "quat", f"{rep['max_abs_quat']:.2e}",
          "env_pos_err (desc)", " ".join(f"{x:.1e}" for x in err))
