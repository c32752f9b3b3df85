"""dev tool: how many Newton iterations the ORACLE's coupled solve takes on the headline workload with contacts resolved
    ORC_NEWTON_TRACE=1 python tools/newton_cap_probe.py [n_envs] [n_steps] [seed]   (per-iteration trace of the solves that run into the cap, stderr)"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("robot-control-stack_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import parity_util as PU
import rcs_oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
O.DEFAULT_RESOLVE_CONTACTS = True
oenvs = PU.make_oracle_envs(n, True)
joints, grip = PU.synthetic_actions(n, steps, seed)
lib = C.CDLL(O._SO)
stats = (C.c_longlong * 4).in_dll(lib, "orc_newton_stats")
t0 = time.time()
for e, oe in enumerate(oenvs):
    before = list(stats)
    oe.reset()
    for t in range(steps):
        oe.step({"joints": joints[t, e], "gripper": grip[t, e]})
    d = [a - b for a, b in zip(list(stats), before)]
    if d[0]:
        print(f"env {e}: solves {d[0]}, iterations per solve {d[1] / d[0]:.2f}, capped {d[2]}, over 20 {d[3]}")
print(f"total: solves {stats[0]}, iterations per solve {stats[1] / max(stats[0], 1):.2f}, capped {stats[2]}, over 20 {stats[3]}  ({time.time() - t0:.0f} s)")
