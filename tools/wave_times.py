"""dev tool (GPU): the lean launch wavefront by wavefront -- when each started, left the substep loop, entered the end-of-launch check
and ended (100 MHz clock).  Library built with -DRCSH_WAVE_TIMES.    RCSH_LIB=.../librcs_hip_wt.so python tools/wave_times.py [n_envs] [skip]"""
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
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 30
venv = PU.make_vec_env(n, True)
L = venv._L
joints, grip = PU.synthetic_actions(n, skip + 8, 0)
venv.reset()
buf = (C.c_ulonglong * (4 * 4096))()
for t in range(skip + 8):
    venv.step({"joints": joints[t], "gripper": grip[t]})
    if t < skip:
        continue
    L.rcsh_debug_wave_times(buf)
    w = np.array(buf[:], dtype=np.float64).reshape(4, 4096)[:, : n // 4] * 0.01  # microseconds
    t0 = w[0].min()
    start, loop, epi, chk = w[0] - t0, w[1] - w[0], w[2] - w[1], w[3] - w[2]
    end = w[3] - t0
    pc = lambda x: " ".join("%.1f" % v for v in np.percentile(x, [0, 50, 90, 99, 100]))  # noqa: E731
    print(f"launch {t}: span {end.max():.1f} us; start [{pc(start)}]; substep loop [{pc(loop)}]; epilogue [{pc(epi)}]; check [{pc(chk)}]; end [{pc(end)}]  (min / median / 90 % / 99 % / max)")
    last = np.argsort(end)[-5:][::-1]
    print("   last to end: " + "; ".join(f"wg {i}: start {start[i]:.1f} loop {loop[i]:.1f} epilogue {epi[i]:.1f} check {chk[i]:.1f}" for i in last))
