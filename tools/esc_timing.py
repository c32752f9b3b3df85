"""dev tool: where an escalated environment's contact-resolving launch spends its cycles (library built with tools/build_timing.sh).
    python tools/esc_timing.py [n_envs] [n_steps] [window]
Only workgroup 0 records: of the lean launch (slot 0-15 phases of its four environments) and of the contact-resolving launch (the
first escalated environment).  Windows without an escalated environment give the lean kernel's own numbers to subtract."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "robot-control-stack_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import rcs_amd._lib as lib
lib.LIB_PATH = os.path.join(ROOT, "robot-control-stack_amd", "rcs_amd", "librcs_hip_timing.so")
import numpy as np
import parity_util as PU

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
win = int(sys.argv[3]) if len(sys.argv) > 3 else 50
venv = PU.make_vec_env(n, os.environ.get("ESC_TIMING_ASYNC", "1") != "0")  # (ESC_TIMING_ASYNC=0: step_until_convergence)
joints, grip = PU.synthetic_actions(n, steps, int(os.environ.get("ESC_TIMING_SEED", "0")))  # (seed e: environment e of the 4096-environment rollout first)
if os.environ.get("ESC_TIMING_GRIP"):  # e.g. 0: every gripper commanded shut (pads pressed together: the many-contact case in workgroup 0)
    grip[:] = float(os.environ["ESC_TIMING_GRIP"])
venv.reset()
L = venv.sim._L
out = (C.c_ulonglong * 96)()
def read():
    L.rcsh_debug_team_cycles96(out, 1)
    return np.array(out[:], dtype=np.float64)
NAMES = {0: "pos stage", 1: "1", 2: "2", 3: "3", 4: "4", 15: "15", 5: "5", 6: "6", 7: "7", 8: "8", 9: "loop tail", 10: "epilogue+check", 11: "11", 12: "prologue12", 13: "13", 14: "14",
         16: "collide before self", 17: "self broad", 18: "self narrow rest", 19: "self box-box", 20: "hull staging", 61: "Gilbert", 62: "portal refinement", 63: "hull records", 24: "before collide", 37: "link frames", 38: "lane per geom", 39: "hulls wavefront", 25: "floor/fastpath", 26: "compaction", 27: "rows/qacc_smooth/M", 28: "newton pre", 55: "x update", 48: "rows+grad",
         49: "stiffness", 50: "Hessian", 51: "row loads", 52: "LDL+solves", 53: "pre linesearch", 54: "linesearch", 30: "forces/Y/K", 31: "noslip rest", 40: "ns rel", 41: "ns owner", 44: "ns slots", 32: "results"}
try:  # the geom pairs of the self-contact stage (index -> geoms), for the slack test's examples
    import ctypes as _C
    g01 = (_C.c_int32 * (2 * 256))(); npair_ = _C.c_int32(); nb_ = _C.c_int32()
    L.rcsh_debug_check_pairs(venv.sim._h, g01, 256, _C.byref(npair_), _C.byref(nb_))
    PAIRS = [(g01[2 * i], g01[2 * i + 1]) for i in range(npair_.value)]
    print("pairs:", len(PAIRS))
except Exception as exc:  # noqa: BLE001
    PAIRS = []
    print("no pair table:", exc)
base = read()
t0 = time.time()
for t in range(steps):
    venv.step({"joints": joints[t], "gripper": grip[t]})
    if (t + 1) % win == 0:
        a = read(); d = a - base; base = a
        now, ever = venv.sim.contact_escalated()
        dt = (time.time() - t0) / win; t0 = time.time()
        print(f"steps {t + 1 - win}..{t}: escalated now {int(now.sum())}, ever {int(ever.sum())}; host {dt * 1e3:.2f} ms/step; contact phases (wg0) {d[33]:.0f}, coupled {d[34]:.0f}, newton its {d[29]:.0f}, ls evals {d[35]:.0f}, noslip sweeps {d[36]:.0f} contacts {d[45]:.0f}")
        print(f"    Newton (all workgroups): solves {d[60]:.0f}, worst iteration count so far {a[56]:.0f}, solves over 20 iterations {d[57]:.0f}, capped at 100 {d[58]:.0f}, left on a non-descent direction {d[59]:.0f}")
        print(f"    self stage (wg0): box-box narrow {d[21]:.0f}, hull narrow {d[22]:.0f} of which full MPR {d[23]:.0f}")
        if d[65]:
            print(f"    contact-resolving launches, all workgroups: {d[65] / win:.1f} workgroups with work per step, mean {d[64] / d[65]:.0f} cycles each, worst {a[66]:.0f}; "
                  f"contact phases {d[71]:.0f}, coupled {d[72]:.0f} ({d[70] / max(d[72], 1):.2f} contacts each, most {a[68]:.0f}; {d[69]:.0f} on the tree formulation), "
                  f"Newton iterations per solve {d[67] / max(d[60], 1):.2f}")
        if d[81] or d[76]:
            print(f"    per coupled phase, all workgroups: few contacts ({d[81]:.0f}): collide {d[78] / max(d[81], 1):.0f}, Newton {d[79] / max(d[81], 1):.0f}, noslip {d[80] / max(d[81], 1):.0f} cycles; "
                  f"many ({d[76]:.0f}): collide {d[73] / max(d[76], 1):.0f}, Newton {d[74] / max(d[76], 1):.0f}, noslip {d[75] / max(d[76], 1):.0f}; quiet collision passes {d[82] / max(d[71] - d[72], 1):.0f}")
        if d[92]:
            print(f"    slack test at the top of the collision pass: {d[92]:.0f} passes, {d[93]:.0f} with a geom pair due, {d[94]:.0f} with a geom due against the floor; last examples: pair {int(a[95]) % 1000 - 1} {PAIRS[int(a[95]) % 1000 - 1] if 0 < int(a[95]) % 1000 <= len(PAIRS) else ''}, floor geom {int(a[95]) // 1000 - 1}")
        if d[92]:
            sd = (C.c_double * 16)(); L.rcsh_debug_slack(sd)
            gg = int(sd[2])
            print(f"      example: env {int(sd[4])} pair {int(sd[0])} geoms {gg & 255},{(gg >> 8) & 255} common ancestor {((gg >> 16) & 255) - 1}: remaining gap after this substep's motion {sd[1]:.3e}, stored by the last pass {sd[3]:.3e}")
        if a[66]:
            print(f"    worst workgroup of the window (environment {a[91]:.0f}): {a[66]:.0f} cycles; contact phases {a[86]:.0f}, coupled {a[87]:.0f} with {a[88] / max(a[87], 1):.1f} contacts each "
                  f"({a[89]:.0f} on the tree formulation); collide {a[83]:.0f}, Newton {a[84]:.0f}, noslip {a[85]:.0f}, quiet collision passes {a[90]:.0f}")
        tot = sum(d[i] for i in NAMES)
        print("    total marked cycles per step %.0f: " % (tot / win) + ", ".join(f"{NAMES[i]} {d[i] / win:.0f}" for i in NAMES if d[i] > 0))
