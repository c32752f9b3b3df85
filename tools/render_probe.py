"""dev tool (GPU): ray-casting time on fr3_empty_world (no contact kernels needed: works with a -DRCSH_DEV_ONLY_FR3 -DRCSH_DEV_NO_CONTACT_KERNELS
library, RCSH_LIB=...), bird's-eye and wrist camera, 4096 x 256 x 256 depth."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "robot-control-stack_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import ctypes as C
import numpy as np
from rcs_amd import _lib
if os.environ.get("RCSH_LIB"): _lib.LIB_PATH = os.environ["RCSH_LIB"]
from rcs_amd import sim as S
from rcs_amd.camera import SimCameraConfig, SimCameraSet
from rcs_amd.envs import default_sim_gripper_cfg, default_sim_robot_cfg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
res = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg = default_sim_robot_cfg("fr3_empty_world")
simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(async_control=True), n_envs=n)
robot = S.SimRobot(simu, None, cfg)
S.SimGripper(simu, default_sim_gripper_cfg())
cs = SimCameraSet(simu, {c: SimCameraConfig(identifier=c, frame_rate=0, resolution_width=res, resolution_height=res) for c in ("wrist_0", "bird_eye_cam")},
                  physical_units=True, render_on_demand=True)
rng = np.random.default_rng(0)
q = np.array([0.0, -0.785, 0.0, -2.356, 0.0, 1.571, 0.785]) + rng.uniform(-0.4, 0.4, (n, 7))
robot.set_joints_hard(q)
simu.step(2)
L, h = simu._L, simu._h
for c in cs.camera_names:
    p = C.c_void_p()
    _lib.check(L.rcsh_dev_alloc(h, n * res * res * 2, C.byref(p)))
    cs.render_depth_mm_dev(c, p.value); simu.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        cs.render_depth_mm_dev(c, p.value)
    simu.synchronize()
    dt = (time.perf_counter() - t0) / 5
    out = np.zeros((n, res, res), dtype=np.uint16)
    _lib.check(L.rcsh_dev_download(h, out.ctypes.data_as(C.c_void_p), p, out.nbytes))
    print(f"{c:14s} {res}x{res}: {dt * 1e3:7.3f} ms per {n} frames ({n * res * res / dt / 1e9:6.1f} G rays/s) checksum {int(out.astype(np.uint64).sum())} nearest {int(out.min())} mm")
    _lib.check(L.rcsh_dev_free(h, p))
simu.close()
