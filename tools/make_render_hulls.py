"""Generate the face planes of the convex hulls the depth renderer casts rays against.

    python tools/make_render_hulls.py

Input: rcs_amd/scenes/<scene>/collision_vertices.npz (hull vertices of the scene's collision meshes, written by
tools/make_collision_vertices.py).  Output: rcs_amd/scenes/<scene>/render_hulls.npz, keyed by MJCF mesh name:
[m, 4] rows (nx, ny, nz, d) with unit outward normals, the hull being { x : n . x <= d for every row }.  Coplanar
facets of the triangulated hull are merged.
"""
import os

import numpy as np
from scipy.spatial import ConvexHull

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = os.path.join(ROOT, "robot-control-stack_amd", "rcs_amd", "scenes")


def planes_of(verts: np.ndarray) -> np.ndarray:
    eq = ConvexHull(verts).equations  # n . x + b <= 0
    eq = eq[np.lexsort(np.round(eq, 9).T[::-1])]
    keep = [0] + [i for i in range(1, len(eq)) if np.abs(eq[i] - eq[i - 1]).max() > 1e-9]
    eq = eq[keep]
    return np.concatenate([eq[:, :3], -eq[:, 3:4]], axis=1)


for scene in sorted(os.listdir(SCENES)):
    src = os.path.join(SCENES, scene, "collision_vertices.npz")
    if not os.path.exists(src):
        continue
    out = {k: planes_of(v) for k, v in np.load(src).items()}
    np.savez_compressed(os.path.join(SCENES, scene, "render_hulls.npz"), **out)
    print(scene, {k: len(v) for k, v in out.items()})
