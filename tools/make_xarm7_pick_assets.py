"""Mesh-derived tables of scenes/xarm7_pick_world: the hand and finger entries of the FR3 scene's tables (the gripper of that
scene IS the Franka hand; tools/make_collision_vertices.py and tools/make_render_hulls.py made them from the reference's
meshes), copied under the same mesh names.  python tools/make_xarm7_pick_assets.py"""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SC = os.path.join(ROOT, "robot-control-stack_amd", "rcs_amd", "scenes")
for name in ("collision_vertices.npz", "render_hulls.npz"):
    src = np.load(os.path.join(SC, "fr3_empty_world", name))
    np.savez(os.path.join(SC, "xarm7_pick_world", name), **{k: src[k] for k in ("franka_hand_coll", "finger_coll")})
    print("wrote", name)
