"""dev tool: does replaying the env-step launches from a HIP graph shorten the gap between dependent launches?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "robot-control-stack_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import torch
from parity_util import MAX_JOINT_MOV, make_vec_env

n, K = 4096, 100
env = make_vec_env(n, async_control=True, gripper=True, relative=True)
side = torch.cuda.Stream()
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
joints = (torch.rand((K, n, 7), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1) * MAX_JOINT_MOV
grip = torch.rand((K, n), generator=gen, device="cuda", dtype=torch.float32)
obs = torch.zeros((n, env.obs_width), device="cuda", dtype=torch.float64)
info = torch.zeros((n, 8), device="cuda", dtype=torch.uint8)
gw = torch.zeros((n,), device="cuda", dtype=torch.float64)
sub = torch.zeros((n,), device="cuda", dtype=torch.int32)

def steps():
    for t in range(K):
        env.step_dev(joints[t].data_ptr(), grip[t].data_ptr(), obs.data_ptr(), info.data_ptr(), gw.data_ptr(), sub.data_ptr())

env.sim.set_stream(torch.cuda.current_stream().cuda_stream)
env.reset_dev(obs.data_ptr(), info.data_ptr(), gw.data_ptr())
steps(); torch.cuda.synchronize()
t0 = time.perf_counter(); steps(); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"stream launches: {(t1 - t0) / K * 1e3:.5f} ms/step")
g = torch.cuda.CUDAGraph()
env.sim.set_stream(side.cuda_stream)  # (synchronises: before the capture starts)
with torch.cuda.graph(g, stream=side):
    steps()
env.sim.set_stream(torch.cuda.current_stream().cuda_stream)
g.replay(); torch.cuda.synchronize()
t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"graph replay:    {(t1 - t0) / K * 1e3:.5f} ms/step")
