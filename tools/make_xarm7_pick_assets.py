"""Mesh-derived tables of scenes/xarm7_pick_world.  python tools/make_xarm7_pick_assets.py  (then tools/make_render_hulls.py)

* the hand and finger entries of the FR3 scene's tables (the gripper of that scene IS the Franka hand;
  tools/make_collision_vertices.py made them from the reference's meshes), copied under the same mesh names;
* the xArm7's own collision hulls: the reference's xarm7.xml gives every moving link one convex collision mesh
  (assets/xarm7/mjcf/xarm7.xml:103-156, class "collision", files assets/xarm7/stl/link<i>_convex.stl, end_tool_convex.stl).
  Those hulls have 350-470 vertices; the contact phase stages a hull's vertices in LDS and takes at most 152 (the size of the
  FR3's collision meshes), so each hull is thinned to an INNER hull of at most MAXV of its own vertices, chosen greedily -- start
  from the six axis extremes, keep adding the vertex that sticks out furthest from the hull of those chosen so far.  The
  deviation (printed; a few tenths of a millimetre) is how far the furthest dropped vertex lies outside the kept hull.
  Numbers only -- vertex coordinates -- go into collision_vertices.npz, keyed by the mesh names scene.xml uses.
"""
import os
import sys

import numpy as np
from scipy.spatial import ConvexHull

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SC = os.path.join(ROOT, "robot-control-stack_amd", "rcs_amd", "scenes")
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_collision_vertices import read_stl  # noqa: E402

REF = "/root/reference/assets/xarm7/stl"
MAXV = 152
LINKS = {**{f"xarm7_link{i}_coll": f"link{i}_convex.stl" for i in range(1, 7)}, "xarm7_end_tool_coll": "end_tool_convex.stl"}


def thin_hull(v, maxv):
    hv = v[np.sort(ConvexHull(v).vertices)]
    if len(hv) <= maxv:
        return hv, 0.0
    keep = sorted({int(np.argmax(s * hv[:, k])) for k in range(3) for s in (1.0, -1.0)})
    out = 0.0
    while True:
        eq = ConvexHull(hv[keep]).equations
        d = (hv @ eq[:, :3].T + eq[:, 3]).max(axis=1)  # > 0: outside the hull of the kept vertices
        d[keep] = -1.0
        i = int(np.argmax(d))
        if len(keep) >= maxv or d[i] <= 0:
            out = max(float(d[i]), 0.0)
            break
        keep.append(i)
    return hv[np.sort(keep)], out


def main():
    src = dict(np.load(os.path.join(SC, "fr3_empty_world", "collision_vertices.npz")))
    dst_path = os.path.join(SC, "xarm7_pick_world", "collision_vertices.npz")
    out = {k: src[k] for k in ("franka_hand_coll", "finger_coll")}
    if os.path.isdir(REF):
        for name, f in LINKS.items():
            v = np.unique(read_stl(os.path.join(REF, f)), axis=0)
            out[name], dev = thin_hull(v, MAXV)
            print(f"{name}: {len(v)} vertices -> {len(out[name])} kept, furthest dropped vertex {1000 * dev:.2f} mm outside")
    else:  # (the reference checkout is absent: keep the link hulls already in the table)
        old = dict(np.load(dst_path))
        out.update({k: old[k] for k in LINKS if k in old})
    np.savez_compressed(dst_path, **out)
    print("wrote", dst_path, os.path.getsize(dst_path), "bytes")


if __name__ == "__main__":
    main()
