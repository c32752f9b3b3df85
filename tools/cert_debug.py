"""dev tool (GPU): which geom pairs fail the certifying end-of-launch check on the headline rollout, and how many environments sit on the
contact-resolving launch step by step.  Needs a library built with -DRCSH_CHECK_DEBUG:
    RCSH_LIB=robot-control-stack_amd/rcs_amd/librcs_hip_chkdbg.so python tools/cert_debug.py [n_envs] [n_steps] [seed]"""
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
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
venv = PU.make_vec_env(n, os.environ.get("RCSH_ASYNC", "1") != "0")
L = venv._L
out = (C.c_int * 64)()
pairs = (C.c_int32 * 2048)()
npair, nb = C.c_int32(0), C.c_int32(0)
L.rcsh_debug_check_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
L.rcsh_debug_check_pairs(venv.sim._h, pairs, 1024, C.byref(npair), C.byref(nb))
names = venv.sim.model.geom_names
joints, grip = PU.synthetic_actions(n, steps, seed)
venv.reset()
L.rcsh_debug_check(out, 1)
hist = {}
for t in range(steps):
    venv.step({"joints": joints[t], "gripper": grip[t]})
    now, ever = venv.sim.contact_escalated()
    L.rcsh_debug_check(out, 1)
    o = list(out)
    for k in range(min(o[1], 28)):
        i = o[2 + k]
        key = (names[pairs[2 * i]], names[pairs[2 * i + 1]]) if 0 <= i < npair.value else ("?", str(i))
        hist[key] = hist.get(key, 0) + 1
    if o[1] and t < 12:
        f = (C.c_double * 128)()
        L.rcsh_debug_check_f(f)
        for k in range(min(o[1], 8)):
            i = int(f[4 * k])
            print("     fail:", names[pairs[2 * i]], names[pairs[2 * i + 1]], "margin %.4f gap end %.5f gap start %.5f" % (f[4 * k + 1], f[4 * k + 2], f[4 * k + 3]))
    if t < 40 or t % 50 == 49:
        print(f"step {t}: on the contact-resolving launch {int(now.sum())}, contact resolved ever {int(ever.sum())}; check: floor fails {o[0]}, pair fails {o[1]}, "
              f"sphere survivors/env {o[32]/n:.2f}, to narrow/env {o[33]/n:.3f}, full MPR {o[38]}, flagged {o[34]} of {o[37]}")
print("pairs that failed the certificate (first 28 of every step):")
for k, v in sorted(hist.items(), key=lambda kv: -kv[1])[:30]:
    print("  ", v, k)
