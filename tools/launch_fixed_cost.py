"""dev tool (GPU): the per-launch fixed cost of the fused env-step launch -- resident env.step launches at several control
frequencies (= substeps per env-step), fitted to a + b * substeps.  RCSH_LIB: a development build; CHECK_EVERY: contact check cadence."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "robot-control-stack_amd"), os.path.join(ROOT, "tests")]
import numpy as np
from rcs_amd import _lib
if os.environ.get("RCSH_LIB"):
    _lib.LIB_PATH = os.environ["RCSH_LIB"]
from parity_util import make_vec_env
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
res = {}
for freq in (500, 100, 50, 30, 15):
    env = make_vec_env(n, True, frequency=freq)
    env.sim.set_contact_check(int(os.environ.get("CHECK_EVERY", "0")))
    L, h = env._L, env.sim._h
    def dev(nbytes, host=None):
        p = C.c_void_p(); _lib.check(L.rcsh_dev_alloc(h, nbytes, C.byref(p)))
        if host is not None: _lib.check(L.rcsh_dev_upload(h, p, host.ctypes.data_as(C.c_void_p), host.nbytes))
        return p.value
    rng = np.random.default_rng(0)
    T = 64
    act = np.ascontiguousarray((rng.random((T, n, 7)) * 2 - 1) * 0.0873)
    grip = np.ascontiguousarray(rng.random((T, n), dtype=np.float32))
    a_d, g_d = dev(act.nbytes, act), dev(grip.nbytes, grip)
    obs, info, gw, sub = dev(8 * n * 21), dev(8 * n), dev(8 * n), dev(4 * n)
    env.reset_dev(obs, info, gw)
    best = 1e9
    for rep in range(4):
        for t in range(40): env.step_dev(a_d + (t % T) * act[0].nbytes, g_d + (t % T) * grip[0].nbytes, obs, info, gw, sub)
        env.sim.synchronize()
        t0 = time.perf_counter(); N = 400
        for t in range(N): env.step_dev(a_d + (t % T) * act[0].nbytes, g_d + (t % T) * grip[0].nbytes, obs, info, gw, sub)
        env.sim.synchronize()
        best = min(best, (time.perf_counter() - t0) / N * 1e6)
    s = np.zeros(n, dtype=np.int32); _lib.check(L.rcsh_dev_download(h, s.ctypes.data_as(C.c_void_p), C.c_void_p(sub), s.nbytes))
    res[int(s[0])] = best
    env.close()
ks = np.array(sorted(res)); ts = np.array([res[k] for k in ks])
b, a = np.polyfit(ks, ts, 1)
print({k: round(v, 2) for k, v in res.items()}, f"us per env-step launch by substeps; fit: {a:.2f} us fixed + {b:.3f} us per substep")
