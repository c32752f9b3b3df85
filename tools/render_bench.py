"""dev tool: time of one depth frame / one colour frame per environment (ray-casting kernels only), per camera and resolution."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "robot-control-stack_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import torch
from rcs_amd.camera import SimCameraConfig, SimCameraSet
from rcs_amd.envs import FR3SimplePickUpSimEnvCreator

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = FR3SimplePickUpSimEnvCreator()(n_envs=n)
env.sim.set_stream(torch.cuda.current_stream().cuda_stream)
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
        out = torch.zeros((n, res[1], res[0]), device="cuda", dtype=torch.uint16)
        cs.render_depth_mm_dev(c, out.data_ptr()); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            cs.render_depth_mm_dev(c, out.data_ptr())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        rgb = torch.zeros((n, res[1], res[0], 3), device="cuda", dtype=torch.uint8)
        cs.render_rgb_dev(c, rgb.data_ptr()); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            cs.render_rgb_dev(c, rgb.data_ptr())
        torch.cuda.synchronize()
        dt_rgb = (time.perf_counter() - t0) / 5
        print(f"{c:14s} {res[0]}x{res[1]}: depth {dt * 1e3:8.3f} ms per batch of {n} frames ({n * res[0] * res[1] / dt / 1e9:6.2f} G rays/s, nearest {int(out.to(torch.int32).min())} mm); "
              f"colour {dt_rgb * 1e3:8.3f} ms, mean rgb {[round(float(x), 1) for x in rgb.to(torch.float32).mean(dim=(0, 1, 2))]}")
