"""dev tool (CPU): which admitted geom pairs of the FR3 are structurally near, and how often a path-length certificate would fail.
Rolls the oracle's headline rollout for a few environments and, per env-step and pair, takes the exact hull-hull distance (numpy GJK).
    python tools/near_pairs_probe.py [n_envs] [n_steps] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("robot-control-stack_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import parity_util as PU  # noqa: E402
import rcs_oracle as RO  # noqa: E402


def closest_on_simplex(P):
    """closest point to the origin on conv(P) (P: k x 3, k <= 4); returns (point, kept rows)"""
    k = len(P)
    best = None
    for mask in range(1, 1 << k):
        idx = [i for i in range(k) if mask >> i & 1]
        Q = P[idx]
        m = len(idx)
        if m == 1:
            lam = np.array([1.0])
        else:
            A = (Q[1:] - Q[0]).T
            try:
                x = np.linalg.lstsq(A, -Q[0], rcond=None)[0]
            except np.linalg.LinAlgError:
                continue
            lam = np.concatenate([[1 - x.sum()], x])
            if (lam < -1e-12).any():
                continue
        v = lam @ Q
        d = v @ v
        if best is None or d < best[0] - 1e-18:
            best = (d, v, idx)
    return best[1], best[2]


def gjk_distance(VA, VB, iters=64):
    def sup(d):
        return VA[np.argmax(VA @ d)] - VB[np.argmin(VB @ d)]
    v = sup(np.array([1.0, 0, 0]))
    S = v[None, :]
    for _ in range(iters):
        n2 = v @ v
        if n2 < 1e-20:
            return 0.0
        w = sup(-v)
        if n2 - v @ w < 1e-12 * max(n2, 1e-12) + 1e-14:
            break
        S = np.vstack([S, w])
        v, keep = closest_on_simplex(S)
        S = S[keep]
        if len(S) == 4:
            return 0.0
    return float(np.sqrt(v @ v))


def geom_world_verts(cm, m, d, g):
    t = m.geom_type[g]
    b = m.geom_bodyid[g]
    R = np.array(d.xmat[b]).reshape(3, 3)
    p = np.array(d.xpos[b])
    gq = np.array(m.geom_quat[g])
    from rcs_amd.mjcf import quat_to_mat
    Rg = R @ quat_to_mat(gq)
    pg = p + R @ np.array(m.geom_pos[g])
    if t == 7:
        adr, num = m.geom_vertadr[g], m.geom_vertnum[g]
        if num == 0:
            return None
        V = np.ctypeslib.as_array(m.mesh_vert, shape=(adr + num, 3))[adr:adr + num]
    elif t == 6:
        s = np.array(m.geom_size[g])
        V = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]) * s
    else:
        return None
    return V @ Rg.T + pg


n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
oenvs = PU.make_oracle_envs(n, True)
cm = oenvs[0].sim.cm
joints, grip = PU.synthetic_actions(n, steps, seed)
for oe in oenvs:
    oe.reset()
m = oenvs[0].sim.s.m.contents if hasattr(oenvs[0].sim.s.m, "contents") else oenvs[0].sim.s.m
ng = m.ngeom
# admitted pairs: different weld bodies, not parent-child (unless parent welded to world), masks
pairs = []
for i in range(ng):
    for j in range(i + 1, ng):
        if m.geom_type[i] == 0 or m.geom_type[j] == 0:
            continue
        bi, bj = m.geom_bodyid[i], m.geom_bodyid[j]
        wi, wj = m.body_weldid[bi], m.body_weldid[bj]
        if wi == wj:
            continue
        pi, pj = m.body_weldid[m.body_parentid[wi]], m.body_weldid[m.body_parentid[wj]]
        if (pi == wj and wj != 0) or (pj == wi and wi != 0):
            continue
        if not ((m.geom_contype[i] & m.geom_conaffinity[j]) or (m.geom_contype[j] & m.geom_conaffinity[i])):
            continue
        pairs.append((i, j))
print("admitted pairs", len(pairs))
names = cm.geom_names
mind = {}
hist = {}
for t in range(steps):
    for e, oe in enumerate(oenvs):
        oe.step({"joints": joints[t, e], "gripper": grip[t, e]})
        if t % 5:
            continue
        d = oe.sim.s.d
        W = {}
        for (i, j) in pairs:
            for g in (i, j):
                if g not in W:
                    W[g] = geom_world_verts(cm, m, d, g)
            if W[i] is None or W[j] is None:
                continue
            ca, cb = W[i].mean(0), W[j].mean(0)
            ra, rb = np.linalg.norm(W[i] - ca, axis=1).max(), np.linalg.norm(W[j] - cb, axis=1).max()
            if np.linalg.norm(ca - cb) - ra - rb > 0.03:
                continue
            dist = gjk_distance(W[i], W[j])
            key = (names[i], names[j])
            mind[key] = min(mind.get(key, 1e9), dist)
            h = hist.setdefault(key, [0, 0, 0, 0])
            h[0] += 1
            if dist < 0.01:
                h[1] += 1
            if dist < 0.003:
                h[2] += 1
            if dist < 0.001:
                h[3] += 1
tot = n * ((steps + 4) // 5)
print("samples", tot)
for key, v in sorted(mind.items(), key=lambda kv: kv[1]):
    h = hist[key]
    print("%-40s %-40s min %.5f  <1cm %.3f  <3mm %.3f  <1mm %.3f" % (key[0], key[1], v, h[1] / tot, h[2] / tot, h[3] / tot))
