"""Generate the face planes of the convex hulls the depth renderer casts rays against.

    python tools/make_render_hulls.py

Input: rcs_amd/scenes/<scene>/collision_vertices.npz (hull vertices of the scene's collision meshes, written by
tools/make_collision_vertices.py).  Output: rcs_amd/scenes/<scene>/render_hulls.npz, keyed by MJCF mesh name:
[m, 4] rows (nx, ny, nz, d) with unit outward normals, the drawn body being { x : n . x <= d for every row }.

The collision hulls carry 200-300 facets; a ray pays for every one of them.  Facets whose normals lie within 8 degrees
of each other (largest facets first) are replaced by ONE supporting plane with their area-weighted normal: ~100 planes
per link, a polytope that contains the hull and exceeds it by at most ~3 mm at its corners (printed below) -- well
inside the difference between the collision hulls and the visual meshes the reference draws.
"""
import os

import numpy as np
from scipy.spatial import ConvexHull, HalfspaceIntersection

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = os.path.join(ROOT, "robot-control-stack_amd", "rcs_amd", "scenes")


ANGLE_DEG = 8.0


def planes_of(verts: np.ndarray):
    h = ConvexHull(verts)
    n = h.equations[:, :3]
    tri = verts[h.simplices]
    area = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
    used = np.zeros(len(n), dtype=bool)
    out = []
    for i in np.argsort(-area, kind="stable"):
        if used[i]:
            continue
        grp = ~used & (n @ n[i] >= np.cos(np.deg2rad(ANGLE_DEG)))
        used |= grp
        nb = (n[grp] * area[grp, None]).sum(axis=0)
        nb /= np.linalg.norm(nb)
        out.append(np.append(nb, (verts @ nb).max()))  # supporting plane: the hull stays inside
    pl = np.array(out)
    corners = HalfspaceIntersection(np.concatenate([pl[:, :3], -pl[:, 3:4]], axis=1), verts.mean(axis=0)).intersections
    excess = float((corners @ h.equations[:, :3].T + h.equations[:, 3]).max())
    return pl, excess


for scene in sorted(os.listdir(SCENES)):
    src = os.path.join(SCENES, scene, "collision_vertices.npz")
    if not os.path.exists(src):
        continue
    res = {k: planes_of(v) for k, v in np.load(src).items()}
    np.savez_compressed(os.path.join(SCENES, scene, "render_hulls.npz"), **{k: pl for k, (pl, _) in res.items()})
    print(scene, {k: (len(pl), f"+{1000 * ex:.1f} mm") for k, (pl, ex) in res.items()})
