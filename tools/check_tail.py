"""dev tool (GPU): how long the end-of-launch check takes per wavefront -- mean and the longest one, and what the longest one did.
Library built with -DRCSH_CHECK_TAIL.    RCSH_LIB=.../librcs_hip_tail.so python tools/check_tail.py [n_envs] [skip] [steps]"""
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
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
venv = PU.make_vec_env(n, True)
L = venv._L
joints, grip = PU.synthetic_actions(n, skip + steps, 0)
venv.reset()
pairs_ = (C.c_int32 * 2048)(); np_, nb_ = C.c_int32(0), C.c_int32(0)
out = (C.c_ulonglong * 16)()
for t in range(skip):
    venv.step({"joints": joints[t], "gripper": grip[t]})
L.rcsh_debug_check_tail(out, 1)
hist = (C.c_ulonglong * 64)()
L.rcsh_debug_check_hist(hist, 1)
worst = []
for t in range(skip, skip + steps):
    venv.step({"joints": joints[t], "gripper": grip[t]})
    L.rcsh_debug_check_tail(out, 1)
    o = list(out)
    worst.append((o[2], o[0] / max(o[1], 1), o[3], o[1], o[4:9], o[9:13], o[13:16]))
w = np.array([x[0] for x in worst]); m = np.array([x[1] for x in worst])
print(f"per launch: mean wavefront {m.mean():.0f} cycles, longest wavefront mean {w.mean():.0f} (max {w.max()}); wavefronts leaving at the slack test {np.mean([x[2] / x[3] for x in worst]):.2f}")
for x in worst[:12]:
    print("  longest %d cycles: narrow rounds %d, Gilbert runs %d, start-frame queries %d, refinements %d, box rounds %d; staging %d cycles (%d rounds not prefetched), narrow phase %d, before it %d; remembered direction's test %d cycles, shapes and slots before it %d, the taking teams' part of the rounds %d" % (x[0], *x[4], *x[5], *x[6]))

L.rcsh_debug_check_hist(hist, 1)
h = np.array(hist[:], dtype=np.float64) / steps
print("wavefronts per launch by total cycles (8k bins):      ", " ".join("%d" % round(x) for x in h[:16]))
print("wavefronts per launch by narrow-phase cycles (4k bins):", " ".join("%d" % round(x) for x in h[16:32]))
print("wavefronts per launch by cycles before it (4k bins):   ", " ".join("%d" % round(x) for x in h[32:48]))
print("environments of the lean launch per launch by narrow-phase rounds (0..15+):", " ".join("%.1f" % x for x in h[48:64]))
