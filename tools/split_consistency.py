"""dev tool: the same scripted pinch at batch scale with the stepping cut into launches of different lengths, kernel against
kernel.  The coupled solve's starting point inside a launch differs from a fresh launch's, so equal results for every
environment say that every solve of every environment reached its minimiser (no stalled or capped Newton solve anywhere).

    python tools/split_consistency.py [n_envs] [chunk_a] [chunk_b]
"""
import os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "robot-control-stack_amd"), os.path.join(ROOT, "tests")]


from parity_util import batch_pinch as pinch, run_split_consistency as compare  # noqa: E402


def newton_stats(n, chunk, **kw):
    """(library built with tools/build_timing.sh) outcomes of the coupled Newton solves of one run, all environments"""
    import ctypes as C
    import rcs_amd._lib as lib
    lib.LIB_PATH = os.path.join(ROOT, "robot-control-stack_amd", "rcs_amd", "librcs_hip_timing.so")
    L = lib.load()
    out = (C.c_ulonglong * 64)()
    L.rcsh_debug_team_cycles64(out); base = list(out)
    pinch(n, chunk, **kw)
    L.rcsh_debug_team_cycles64(out)
    d = [out[i] - base[i] for i in range(64)]
    return {"solves": d[60], "worst_iterations": out[56], "over_20_iterations": d[57], "ran_into_the_cap": d[58], "left_on_a_non_descent_direction": d[59]}


if __name__ == "__main__":
    if "--newton-stats" in sys.argv:
        sys.argv.remove("--newton-stats")
        a = [float(x) for x in sys.argv[1:]]
        print(newton_stats(int(a[0]) if a else 4096, 17, seed=int(a[1]) if len(a) > 1 else 0, spread=a[2] if len(a) > 2 else 0.004, yaw=a[3] if len(a) > 3 else 0.1))
        sys.exit(0)
    args = [int(x) for x in sys.argv[1:]]
    n, ca, cb = (args + [4096, 17, 100])[:3] if len(args) < 3 else args[:3]
    for tag, r in compare(n, ca, cb).items():
        print(tag, r)
