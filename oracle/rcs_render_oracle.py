"""CPU restatement of the depth ray-caster (TEST INFRASTRUCTURE; see oracle/rcs_oracle.h for the rules).

numpy, one ray per pixel, the same shapes and the same formulas as csrc/render.h; the frames come from the oracle's
mjData restatement (xpos / xmat of the last position stage).  What is restated from the reference -- the camera model
of MuJoCo, the OpenGL depth encoding, python/rcs/camera/sim.py:57-115 (row flip, metres in float32, uint16
millimetres, intrinsics, extrinsics) -- is cited in csrc/render.h and rcs_amd/camera.py.  PARITY UNPINNED for the
pixels themselves: the reference's come from MuJoCo's OpenGL rasteriser over the visual meshes.
"""

from __future__ import annotations

import numpy as np

SHAPE_PLANE, SHAPE_BOX, SHAPE_HULL, SHAPE_CAPSULE = 0, 1, 2, 3
last_hit_shape = None


def oracle_frames(osim, cm) -> dict:
    """link index -> (R [3,3], p [3]) from the oracle's mjData: robot links, -1 world, -2 the free box."""
    d = osim.s.d
    xpos = np.ctypeslib.as_array(d.xpos)
    xmat = np.ctypeslib.as_array(d.xmat)
    frames = {-1: (np.eye(3), np.zeros(3))}
    for j in range(cm.njnt):
        b = int(cm.arrays["jnt_bodyid"][j])
        frames[j] = (xmat[b].reshape(3, 3).copy(), xpos[b].copy())
    if getattr(cm, "free_bodies", []):
        w, x, y, z = d.box.xquat[:]
        R = np.array([[w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])
        frames[-2] = (R, np.array(d.box.xpos[:]))
    return frames


def render_depth(rs, cam, frames, colour=False):
    """rs: rcs_amd.render.RenderScene; cam: (link, pos, rot9, fovy_deg, W, H).  Returns (depth_gl [H,W] float32 rows
    bottom-up, depth_mm [H,W] uint16 rows top-down, cam_R, cam_p) and, with `colour`, rgb [H,W,3] uint8 rows bottom-up:
    the colour of the shape entered first, flat-shaded on the entry face (csrc/render.h, COLOR)."""
    link, cpos, crot, fovy, W, H = cam
    Rl, pl = frames[link]
    cR, cp = Rl @ np.asarray(crot).reshape(3, 3), Rl @ np.asarray(cpos) + pl
    ty = np.tan(fovy * np.pi / 360.0)
    tx = ty * W / H
    col, row = np.meshgrid(np.arange(W), np.arange(H))
    dc = np.stack([(2.0 * (col + 0.5) / W - 1.0) * tx, (2.0 * (row + 0.5) / H - 1.0) * ty, -np.ones((H, W))], axis=-1)
    d = dc @ cR.T
    dd = (d * d).sum(-1)
    best = np.full((H, W), rs.zfar)
    hit = np.zeros((H, W), dtype=bool)
    hit_g = np.full((H, W), -1)
    hit_n = np.zeros((H, W, 3))  # entry face normal, shape frame
    for g in range(len(rs.shape)):
        Rg, pg = frames[int(rs.link[g])]
        R, p = Rg @ rs.rot[g].reshape(3, 3), Rg @ rs.pos[g] + pg
        live = np.ones((H, W), dtype=bool)
        if rs.sphere[g][3] >= 0:
            oc = R @ rs.sphere[g][:3] + p - cp
            b = d @ oc
            live = ~((oc @ oc) * dd - b * b > rs.sphere[g][3] ** 2 * dd)
        lo = R.T @ (cp - p)
        ld = d @ R
        t0 = np.full((H, W), rs.znear)
        t1 = best.copy()
        if rs.shape[g] == SHAPE_PLANE:
            ok = live & (ld[..., 2] < 0) & (lo[2] > 0)
            with np.errstate(divide="ignore", invalid="ignore"):
                t = -lo[2] / ld[..., 2]
            ok &= (t >= t0) & (t < t1)
            best = np.where(ok, t, best)
            hit |= ok
            hit_g = np.where(ok, g, hit_g)
            hit_n = np.where(ok[..., None], np.array([0.0, 0.0, 1.0]), hit_n)
            continue
        ok = live.copy()
        face_n = np.zeros((H, W, 3))
        if rs.shape[g] == SHAPE_CAPSULE:
            # axis z of the shape frame, radius size[0], half length size[2] - size[0]: the first point of the ray on the wall
            # between the caps or on the outer half of a cap sphere
            r, hl = rs.size[g][0], rs.size[g][2] - rs.size[g][0]
            with np.errstate(divide="ignore", invalid="ignore"):
                a = ld[..., 0] ** 2 + ld[..., 1] ** 2
                bq = lo[0] * ld[..., 0] + lo[1] * ld[..., 1]
                cq = lo[0] ** 2 + lo[1] ** 2 - r * r
                disc = bq * bq - a * cq
                t = (-bq - np.sqrt(disc)) / a
                te = np.where((a > 0) & (disc >= 0) & (np.abs(lo[2] + t * ld[..., 2]) <= hl), t, np.inf)
                A = a + ld[..., 2] ** 2
                for zc in (-hl, hl):
                    oz = lo[2] - zc
                    B, Cq = bq + oz * ld[..., 2], cq + oz * oz
                    ds = B * B - A * Cq
                    t = (-B - np.sqrt(ds)) / A
                    zr = oz + t * ld[..., 2]
                    te = np.where((ds >= 0) & ((zr >= 0) if zc > 0 else (zr <= 0)) & (t < te), t, te)
            ok &= np.isfinite(te) & (te > rs.znear) & (te < best)
            hp = lo + np.where(ok, te, 0.0)[..., None] * ld
            face_n = hp.copy()
            face_n[..., 2] = hp[..., 2] - np.clip(hp[..., 2], -hl, hl)
            best = np.where(ok, te, best)
            hit |= ok
            hit_g = np.where(ok, g, hit_g)
            hit_n = np.where(ok[..., None], face_n, hit_n)
            continue
        with np.errstate(divide="ignore", invalid="ignore"):
            if rs.shape[g] == SHAPE_BOX:
                for k in range(3):
                    zero = ld[..., k] == 0
                    inv = 1.0 / ld[..., k]
                    ta, tb = (-rs.size[g][k] - lo[k]) * inv, (rs.size[g][k] - lo[k]) * inv
                    ta, tb = np.minimum(ta, tb), np.maximum(ta, tb)
                    t0n, t1n = np.maximum(t0, ta), np.minimum(t1, tb)
                    axis = np.zeros(3)
                    axis[k] = 1.0
                    face_n = np.where((~(zero | ~ok) & (ta > t0))[..., None], np.where(ld[..., k] > 0, -1.0, 1.0)[..., None] * axis, face_n)
                    t0, t1 = np.where(zero | ~ok, t0, t0n), np.where(zero | ~ok, t1, t1n)
                    ok &= np.where(zero, abs(lo[k]) <= rs.size[g][k], t0 <= t1)
            else:
                for pl4 in rs.planes[rs.plane_adr[g]: rs.plane_adr[g] + rs.plane_num[g]]:
                    nd = ld @ pl4[:3]
                    no = pl4[3] - pl4[:3] @ lo
                    zero = nd == 0
                    t = no / nd
                    t0n = np.where(nd < 0, np.maximum(t0, t), t0)
                    t1n = np.where(nd > 0, np.minimum(t1, t), t1)
                    face_n = np.where((~(zero | ~ok) & (nd < 0) & (t > t0))[..., None], pl4[:3], face_n)
                    t0, t1 = np.where(zero | ~ok, t0, t0n), np.where(zero | ~ok, t1, t1n)
                    ok &= np.where(zero, no >= 0, t0 <= t1)
        ok &= (t0 > rs.znear) & (t0 < best)
        best = np.where(ok, t0, best)
        hit |= ok
        hit_g = np.where(ok, g, hit_g)
        hit_n = np.where(ok[..., None], face_n, hit_n)
    global last_hit_shape
    last_hit_shape = hit_g  # [H, W] index of the shape each ray enters first (-1: none), rows bottom-up: for the tests' bookkeeping
    inv_near, inv_far = 1.0 / rs.znear, 1.0 / rs.zfar
    dgl = np.where(hit, (inv_near - 1.0 / best) / (inv_near - inv_far), 1.0).astype(np.float32)
    # python/rcs/camera/sim.py:57-86 on the buffer mjr_readPixels returned
    frame = dgl[::-1]
    z = np.float32(rs.znear) / (np.float32(1) - frame * np.float32(1 - rs.znear / rs.zfar))
    mm = (z * np.float32(1000)).astype(np.uint16)
    if not colour:
        return dgl, mm, cR, cp
    inv_len = 1.0 / np.sqrt(dd)
    f = 0.5 * (d[..., 2] * inv_len + 1.0)
    out = rs.sky_rgb2 + f[..., None] * (rs.sky_rgb1 - rs.sky_rgb2)
    ldir = rs.light_dir / np.linalg.norm(rs.light_dir)
    for g in range(len(rs.shape)):
        m = hit & (hit_g == g)
        if not m.any():
            continue
        Rg, pg = frames[int(rs.link[g])]
        R, p = Rg @ rs.rot[g].reshape(3, 3), Rg @ rs.pos[g] + pg
        nw = hit_n @ R.T
        nn = 1.0 / np.sqrt((nw * nw).sum(-1))
        kv = np.maximum(-(nw * d).sum(-1) * nn * inv_len, 0.0)
        kl = np.maximum(-(nw @ ldir) * nn, 0.0)
        base = np.broadcast_to(rs.colour[g][:3], (H, W, 3))
        if rs.colour[g][7] != 0:
            hp = (cp + best[..., None] * d - p) @ R
            second = ((np.floor(hp[..., 0] / rs.colour[g][6]).astype(np.int64) + np.floor(hp[..., 1] / rs.colour[g][6]).astype(np.int64)) & 1) != 0
            base = np.where(second[..., None], rs.colour[g][3:6], rs.colour[g][:3])
        shaded = base * (rs.headlight_ambient + rs.headlight_diffuse * kv[..., None] + rs.light_diffuse * kl[..., None])
        out = np.where(m[..., None], shaded, out)
    rgb = (np.clip(out, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)
    return dgl, mm, cR, cp, rgb
