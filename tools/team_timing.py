"""dev tool: per-phase cycle counts of the team kernel (library built with -DRCSH_PHASE_TIMING)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "robot-control-stack_amd"), os.path.join(ROOT, "tests")]
os.environ["RCSH_KERNEL"] = "team"
import rcs_amd._lib as lib
lib.LIB_PATH = os.path.join(ROOT, "robot-control-stack_amd", "rcs_amd", "librcs_hip_timing.so")
import numpy as np
from parity_util import make_vec_env, synthetic_actions
n, T = 4096, 20
ASYNC = os.environ.get("ASYNC", "1") == "1"
env = make_vec_env(n, ASYNC, robot=(sys.argv[1] if len(sys.argv) > 1 else "fr3"))
env.sim.set_contact_check(int(os.environ.get("CHECK_EVERY", "0")))  # (the end-of-launch contact check would sit in the epilogue mark)
j, g = synthetic_actions(64 if os.environ.get("TILED", "1") == "1" else n, T, 0, dof=env.dof)
if j.shape[1] != n: j = np.tile(j, (1, n // 64, 1)); g = np.tile(g, (1, n // 64))
env.reset()
out = (C.c_ulonglong * 24)()
env._L.rcsh_debug_team_cycles(out)
base = np.array(out[:], dtype=np.float64)
nsub = 0
for t in range(T):
    info = env.step({"joints": j[t], "gripper": g[t]})[4]
    nsub += int(info["substeps"][0:4].max())
env._L.rcsh_debug_team_cycles(out)
a = (np.array(out[:], dtype=np.float64) - base)
names = ["local frame", "frame scan", "axis+vel+acc scans", "inertia+wrench+leaf scans", "M row (S exchange)", "actuation+rows", "factor slot",
         "implicit solve+integrate", "callbacks+sync", "leader post + loop sync", "epilogue", "prologue"]
sub = nsub  # (substeps workgroup 0 ran: its slowest team's)
print("team kernel, block 0 lane 0, cycles per substep:")
for i in range(10): print(f"  {names[i]:28s} {a[i] / sub:9.0f}")
print(f"  {'substep total':28s} {a[:10].sum() / sub:9.0f}")
print(f"  (of actuation+rows: through the gripper / coupling block {a[15] / sub:.0f})")
print(f"per launch: tables -> LDS {a[12] / T:.0f}  load_env {a[13] / T:.0f}  env_prologue {a[14] / T:.0f}  rest of prologue {a[11] / T:.0f}  epilogue {a[10] / T:.0f}  substeps {a[:10].sum() / T:.0f}")
out64 = (C.c_ulonglong * 64)()
env._L.rcsh_debug_team_cycles64(out64)
if out64[61]:
    print(f"dry-friction candidates: {out64[62] / out64[61]:.3f} of the team-substeps found a self-consistent zone set in the slot; newton_rows ran in {out64[63] / (out64[61] / 4):.3f} of the wavefront-substeps")
if out64[44]:
    print(f"newton_rows: {out64[44]} team-calls, {out64[45] / out64[44]:.2f} iterations each, worst {out64[46]}, over 4 iterations {out64[47]}, at the cap {out64[43]}")
if out64[42]:
    print(f"block 0: newton_rows entered {out64[42]} times in {sub} substeps, {out64[40] / out64[42]:.0f} cycles per call")
if out64[49] + out64[50] + out64[51]:
    tt = out64[49] + out64[50] + out64[51]; ww = out64[52] + out64[53] + out64[54]
    print(f"slot rounds used, team-substeps: 1: {out64[49] / tt:.3f}  2: {out64[50] / tt:.3f}  3: {out64[51] / tt:.3f};  passes run, wavefront-substeps: 1: {out64[52] / ww:.3f}  2: {out64[53] / ww:.3f}  3: {out64[54] / ww:.3f}")
if out64[55] + out64[56] + out64[57] + out64[58]:
    print(f"late winners: rows differing from the first base: 0: {out64[55]}  1: {out64[56]}  2: {out64[57]}  3+: {out64[58]}")
if out64[61]:
    print(f"first-round winners: base {out64[24]}, a near move {out64[25]}, a far move {out64[26]}")
