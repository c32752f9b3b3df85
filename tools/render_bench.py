"""dev tool: time of one depth frame / one colour frame per environment (ray-casting kernels only), per camera and resolution."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "robot-control-stack_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import ctypes as C
import numpy as np
from rcs_amd import _lib
if os.environ.get("RCSH_LIB"): _lib.LIB_PATH = os.environ["RCSH_LIB"]  # (a development build of the library)
np.random.seed(0)  # RandomCubePos draws from numpy's global generator: the same cubes every run
from rcs_amd.camera import SimCameraConfig, SimCameraSet
from rcs_amd.envs import FR3SimplePickUpSimEnvCreator

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = FR3SimplePickUpSimEnvCreator()(n_envs=n)
L, h = env._L, env.sim._h


def dev_alloc(nbytes):
    p = C.c_void_p()
    _lib.check(L.rcsh_dev_alloc(h, nbytes, C.byref(p)))
    return p.value


def dev_get(ptr, shape, dtype):
    out = np.zeros(shape, dtype=dtype)
    _lib.check(L.rcsh_dev_download(h, out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), out.nbytes))
    return out


# (a camera set must exist BEFORE the stepping whose frames it renders: the kernels keep the qpos of the last position stage
# only while a render scene is attached)
SimCameraSet(env.sim, {"w": SimCameraConfig(identifier="wrist_0", resolution_width=8, resolution_height=8)}, physical_units=True)
env.reset()
rng = np.random.default_rng(0)
for _ in range(3):
    env.step({"xyzrpy": rng.uniform(-0.05, 0.05, (n, 6)), "gripper": rng.uniform(0, 1, n)})
for res in ((64, 64), (128, 128), (256, 256)):
    cs = SimCameraSet(env.sim, {c: SimCameraConfig(identifier=c, resolution_width=res[0], resolution_height=res[1]) for c in ("wrist_0", "bird_eye_cam")}, physical_units=True)
    for c in cs.camera_names:
        out = dev_alloc(n * res[1] * res[0] * 2)
        cs.render_depth_mm_dev(c, out); env.sim.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            cs.render_depth_mm_dev(c, out)
        env.sim.synchronize()
        dt = (time.perf_counter() - t0) / 5
        rgb = dev_alloc(n * res[1] * res[0] * 3)
        cs.render_rgb_dev(c, rgb); env.sim.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            cs.render_rgb_dev(c, rgb)
        env.sim.synchronize()
        dt_rgb = (time.perf_counter() - t0) / 5
        mm = dev_get(out, (n, res[1], res[0]), np.uint16)
        px = dev_get(rgb, (n, res[1], res[0], 3), np.uint8)
        print(f"{c:14s} {res[0]}x{res[1]}: depth {dt * 1e3:8.3f} ms per batch of {n} frames ({n * res[0] * res[1] / dt / 1e9:6.2f} G rays/s, nearest {int(mm.min())} mm, "
              f"checksum {int(mm.astype(np.uint64).sum())}); colour {dt_rgb * 1e3:8.3f} ms, mean rgb {[round(float(x), 1) for x in px.reshape(-1, 3).mean(axis=0)]}")
        _lib.check(L.rcsh_dev_free(h, C.c_void_p(out))); _lib.check(L.rcsh_dev_free(h, C.c_void_p(rgb)))
