"""Dump the finalised FR3 DevModel (GPU box only: creating a Sim needs a device). Usage: python tools/dump_model.py out.bin"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "robot-control-stack_amd"), os.path.join(ROOT, "tests")]
from parity_util import make_vec_env  # noqa: E402

env = make_vec_env(64, True)
L, h = env._L, env.sim._h
size = C.c_size_t(0)
L.rcsh_debug_dump_model(h, None, 0, C.byref(size))
buf = C.create_string_buffer(size.value)
L.rcsh_debug_dump_model(h, buf, size.value, C.byref(size))
open(sys.argv[1], "wb").write(buf.raw)
print("DevModel bytes:", size.value)
