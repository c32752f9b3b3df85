"""dev tool (GPU, -DRCSH_CHECK_DEBUG library via RCSH_LIB): what the contact check flags, step by step, on the headline workload."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("robot-control-stack_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
from rcs_amd import _lib
if os.environ.get("RCSH_LIB"):
    _lib.LIB_PATH = os.environ["RCSH_LIB"]
import parity_util as PU
n, seed, upto = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
watch = [int(x) for x in sys.argv[4:]]
venv = PU.make_vec_env(n, True)
L = venv._L
joints, grip = PU.synthetic_actions(n, upto, seed)
venv.reset()
out = (C.c_int * 64)()
pairs = (C.c_int32 * 2048)()
npair, nb = C.c_int32(0), C.c_int32(0)
L.rcsh_debug_check_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
L.rcsh_debug_check_pairs(venv.sim._h, pairs, 1024, C.byref(npair), C.byref(nb))
names = venv.sim.model.geom_names
prev = np.zeros(n, dtype=bool)
L.rcsh_debug_check(out, 1)
for t in range(upto):
    _, _, _, _, info = venv.step({"joints": joints[t], "gripper": grip[t]})
    fl = np.asarray(info["contact_unresolved"], dtype=bool)
    new = np.nonzero(fl & ~prev)[0]
    L.rcsh_debug_check(out, 1)
    o = list(out)
    if len(new) and (not watch or set(new) & set(watch)):
        print("step", t, "newly flagged", list(new), "plane hits", o[0], "pair hits", o[1], [(names[pairs[2 * i]], names[pairs[2 * i + 1]]) for i in o[2:2 + min(o[1], 8)]])
    prev = fl
