"""dev tool (GPU): which test of the end-of-launch contact check fires.  Needs a library built with -DRCSH_CHECK_DEBUG (RCSH_LIB=...)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("robot-control-stack_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
from rcs_amd import _lib
if os.environ.get("RCSH_LIB"):
    _lib.LIB_PATH = os.environ["RCSH_LIB"]
import parity_util as PU

n = 4
venv = PU.make_vec_env(n, True)
L = venv._L
out = (C.c_int * 64)()
print("after construction:", venv.sim.contact_unresolved())
L.rcsh_debug_check(out, 1)
print("  counters", list(out)[:3], list(out)[32:38])
venv.reset()
print("after reset:", venv.sim.contact_unresolved())
L.rcsh_debug_check(out, 1)
print("  counters", list(out)[:3], list(out)[32:38])
pairs = (C.c_int32 * 2048)()
npair, nb = C.c_int32(0), C.c_int32(0)
L.rcsh_debug_check_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
L.rcsh_debug_check_pairs(venv.sim._h, pairs, 1024, C.byref(npair), C.byref(nb))
print("pairs", npair.value, "body pairs", nb.value)
names = venv.sim.model.geom_names
for t in range(3):
    _, _, _, _, info = venv.step({"joints": np.zeros((n, 7)), "gripper": np.ones(n)})
    L.rcsh_debug_check(out, 1)
    o = list(out)
    print("step", t, "flag", info["contact_unresolved"], "plane hits", o[0], "pair hits", o[1], "sphere survivors", o[32], "geom pairs to MPR", o[33], "leader hits", o[34], "r.flags bit", o[35], "pre_flags bit", o[36], "leaders", o[37], "full MPR", o[38], "not prefetched", o[39])
    for k in range(min(o[1], 28)):
        i = o[2 + k]
        print("   pair", i, names[pairs[2 * i]], names[pairs[2 * i + 1]])

cyc = (C.c_ulonglong * 16)()
L.rcsh_debug_check_cycles(cyc, 1)
N = 50
for t in range(N):
    venv.step({"joints": np.random.default_rng(t).uniform(-0.08, 0.08, (n, 7)), "gripper": np.ones(n) * (t % 2)})
L.rcsh_debug_check_cycles(cyc, 1)
names_ = ["FK", "frames+boxes", "plane", "spheres", "OBB", "narrow", "sep store"]
print("cycles per check (workgroup 0):", {nm: int(cyc[i] / N) for i, nm in enumerate(names_)}, "total", int(sum(cyc[:7]) / N))

# statistics of a random walk (the headline's actions) on a bigger batch
venv.close()
n = 1024
venv = PU.make_vec_env(n, True)
venv.reset()
joints, grip = PU.synthetic_actions(n, 300, 0)
L.rcsh_debug_check(out, 1)
for t in range(300):
    venv.step({"joints": joints[t], "gripper": grip[t]})
    if t % 50 == 49:
        L.rcsh_debug_check(out, 1)
        o = list(out)
        print(f"steps {t-49}..{t}: per env-step: sphere survivors {o[32]/n/50:.2f}, pairs to narrow {o[33]/n/50:.2f}, full MPR {o[38]/n/50:.3f} (no slot {o[40]/n/50:.3f}, direction failed {o[41]/n/50:.3f}), hits {o[1]/n/50:.4f}; per wave-step not prefetched {o[39]/(n/4)/50:.3f}")
