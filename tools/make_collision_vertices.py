"""Generate the collision vertex sets of a scene's mesh geoms (run where the reference assets are present).

    python tools/make_collision_vertices.py

MuJoCo collides the CONVEX HULL of a mesh geom; for contacts against a plane only the hull's vertices matter
(the deepest point of a convex body along any direction is a hull vertex).  This script reads the reference's
collision meshes (binary STL / OBJ, reference assets/fr3/stl, assets/grippers/franka_hand), keeps the hull vertices
and writes them -- numbers only -- to rcs_amd/scenes/<scene>/collision_vertices.npz, keyed by MJCF mesh name.
Meshes missing from the reference checkout (.MISSING_LARGE_BLOBS: the camera mount) get no entry; their geoms
then never report contacts (documented in DESIGN.md).
"""
import os
import struct

import numpy as np
from scipy.spatial import ConvexHull

REF = "/root/reference/assets"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "robot-control-stack_amd", "rcs_amd", "scenes",
                   "fr3_empty_world", "collision_vertices.npz")

MESHES = {  # MJCF mesh name -> file (reference assets/fr3/mjcf/fr3_common.xml:47-56, fr3_0.xml:145,157)
    **{f"fr3_link{i}_coll": f"fr3/stl/fr3_link{i}.stl" for i in range(8)},
    "franka_hand_coll": "grippers/franka_hand/stl/franka_hand.stl",
    "finger_coll": "grippers/franka_hand/obj/finger_0.obj",
    "camera_mount_coll": "cameras/real_sense/stl/Panda_RealSenseD435_Camera_Mount.stl",
}


def read_stl(path):
    d = open(path, "rb").read()
    n = struct.unpack("<I", d[80:84])[0]
    a = np.frombuffer(d[84 : 84 + 50 * n], dtype=np.dtype([("n", "<3f4"), ("v", "<9f4"), ("a", "<u2")]))
    return a["v"].reshape(-1, 3).astype(np.float64)


def read_obj(path):
    return np.array([[float(x) for x in ln.split()[1:4]] for ln in open(path) if ln.startswith("v ")], dtype=np.float64)


def main():
    out = {}
    for name, rel in MESHES.items():
        path = os.path.join(REF, rel)
        if not os.path.exists(path):
            cands = [p for p in (os.path.join(REF, "fr3/obj", os.path.basename(rel)), os.path.join(REF, "scenes/fr3_empty_world/assets", os.path.basename(rel))) if os.path.exists(p)]
            if not cands:
                print(f"{name}: {rel} missing from the checkout -> no collision vertices")
                continue
            path = cands[0]
        v = read_stl(path) if path.endswith(".stl") else read_obj(path)
        v = np.unique(v, axis=0)
        hull = ConvexHull(v)
        hv = v[np.sort(hull.vertices)]
        out[name] = hv
        print(f"{name}: {len(v)} unique vertices -> {len(hv)} hull vertices, bbox {hv.min(0).round(4)} .. {hv.max(0).round(4)}")
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
