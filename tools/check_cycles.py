"""dev tool (GPU): cycles of the end-of-launch check's phases (workgroup 0), headline rollout.  Library built with -DRCSH_CHECK_DEBUG.
    RCSH_LIB=.../librcs_hip_chkdbg.so [RCSH_CHECK_CERTIFY=0] python tools/check_cycles.py [n_envs] [skip] [steps]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("robot-control-stack_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
from rcs_amd import _lib
if os.environ.get("RCSH_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["RCSH_LIB"])
import parity_util as PU

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 20
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
venv = PU.make_vec_env(n, True)
L = venv._L
joints, grip = PU.synthetic_actions(n, skip + steps, 0)
venv.reset()
cyc = (C.c_ulonglong * 16)()
out = (C.c_int * 64)()
for t in range(skip):
    venv.step({"joints": joints[t], "gripper": grip[t]})
L.rcsh_debug_check_cycles(cyc, 1)
L.rcsh_debug_check(out, 1)
for t in range(skip, skip + steps):
    venv.step({"joints": joints[t], "gripper": grip[t]})
L.rcsh_debug_check_cycles(cyc, 1)
L.rcsh_debug_check(out, 1)
o = list(out)
names_ = ["FK", "frames+boxes+tables", "plane", "spheres", "OBB", "narrow", "sep store"]
print("certify", os.environ.get("RCSH_CHECK_CERTIFY", "1"), "cycles per check (workgroup 0, lean + contact-resolving launches):", {nm: int(cyc[i] / steps) for i, nm in enumerate(names_)}, "total", int(sum(cyc[:7]) / steps))
print(f"  slack: due pairs + links per env-step {o[44]/n/steps:.2f} (links {o[45]/n/steps:.3f}); wavefronts with nothing due {o[42]} of {o[43]}")
print(f"  per env-step: sphere survivors {o[32]/n/steps:.2f}, pairs to narrow {o[33]/n/steps:.3f}, certificate fails {o[1]/n/steps:.4f}, flagged {o[34]/steps:.2f} of {o[37]/steps:.0f} checked")
