"""dev tool (CPU): how often a path-length certificate of "no contact inside the launch" would fail on the headline rollout.
Per env-step and admitted pair: exact gap at the step's start (d0) and end (d1) -- numpy GJK -- against the travel bound
M = sum over the joints between the two links of lever x |dq|.  Reports the share of env-steps in which any pair fails
  (a) d1 > M            (the round-5 certifying check, exact gaps)
  (b) d0 + d1 > M       (the path-length form)
and the same for the floor with isotropic / vertical levers.
    python tools/cert_rate_probe.py [n_envs] [n_steps] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("robot-control-stack_amd", "oracle", "tests", "tools"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import parity_util as PU  # noqa: E402
from rcs_amd.mjcf import quat_to_mat  # noqa: E402

sys.argv, argv = sys.argv[:1] + ["0", "0"], sys.argv
import importlib.util  # noqa: E402
spec = importlib.util.spec_from_file_location("npp", os.path.join(ROOT, "tools", "near_pairs_probe.py"))
src = open(os.path.join(ROOT, "tools", "near_pairs_probe.py")).read().split("\nn = int(sys.argv[1])")[0]
ns = {"__file__": os.path.join(ROOT, "tools", "near_pairs_probe.py")}
exec(compile(src, "npp", "exec"), ns)
gjk_distance, geom_world_verts = ns["gjk_distance"], ns["geom_world_verts"]

n = int(argv[1]) if len(argv) > 1 else 4
steps = int(argv[2]) if len(argv) > 2 else 200
seed = int(argv[3]) if len(argv) > 3 else 0
oenvs = PU.make_oracle_envs(n, True)
cm = oenvs[0].sim.cm
joints, grip = PU.synthetic_actions(n, steps, seed)
for oe in oenvs:
    oe.reset()
m = oenvs[0].sim.s.m.contents if hasattr(oenvs[0].sim.s.m, "contents") else oenvs[0].sim.s.m
ng, nj = m.ngeom, m.njnt
names = cm.geom_names
floor = [g for g in range(ng) if m.geom_type[g] == 0]
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
        if (names[i], names[j]) == ("fr3_link0_collision_0", "fr3_link1_collision_0"):
            continue  # (the host proves this pair apart in every pose)
        pairs.append((i, j))


def anc_joints(b):
    out = set()
    while b > 0:
        for j in range(nj):
            if m.jnt_bodyid[j] == b:
                out.add(j)
        b = m.body_parentid[b]
    return out


ganc = {g: anc_joints(m.geom_bodyid[g]) for g in range(ng)}
stats = dict(steps=0, a=0, b=0, fl_iso=0, fl_dir=0, fl_dir_sum=0, any_b=0)
which_a, which_b = {}, {}
for e, oe in enumerate(oenvs):
    prevW, prevq, prevd = None, None, {}
    for t in range(steps):
        oe.step({"joints": joints[t, e], "gripper": grip[t, e]})
        d = oe.sim.s.d
        q = np.array(d.qpos[:nj])
        W = {g: geom_world_verts(cm, m, d, g) for g in range(ng)}
        anchor = np.array([d.xanchor[j] for j in range(nj)])
        axis = np.array([d.xaxis[j] for j in range(nj)])
        dist = {}
        if prevq is not None:
            dq = np.abs(q - prevq) * 1.2  # (path >= net; a little slack)
            stats["steps"] += 1
            fa = fb = False
            for (i, j) in pairs:
                if W[i] is None or W[j] is None:
                    continue
                between = ganc[i] ^ ganc[j]
                M = 0.0
                for jj in between:
                    g_down = j if jj in ganc[j] else i
                    lev = 1.0 if m.jnt_type[jj] == 2 else np.linalg.norm(W[g_down] - anchor[jj], axis=1).max() * 1.02
                    M += lev * dq[jj]
                ca, cb = W[i].mean(0), W[j].mean(0)
                ra, rb = np.linalg.norm(W[i] - ca, axis=1).max(), np.linalg.norm(W[j] - cb, axis=1).max()
                if np.linalg.norm(ca - cb) - ra - rb > M:
                    dist[(i, j)] = np.linalg.norm(ca - cb) - ra - rb
                    continue
                d1 = gjk_distance(W[i], W[j])
                dist[(i, j)] = d1
                d0 = prevd.get((i, j), 0.0)
                if d1 <= M:
                    fa = True
                    which_a[(names[i], names[j])] = which_a.get((names[i], names[j]), 0) + 1
                if d0 + d1 <= M:
                    fb = True
                    which_b[(names[i], names[j])] = which_b.get((names[i], names[j]), 0) + 1
            stats["a"] += fa
            stats["b"] += fb
            # floor: every non-plane geom not welded to the world
            f_iso = f_dir = f_dir_sum = False
            for g in range(ng):
                if W[g] is None or m.body_weldid[m.geom_bodyid[g]] == 0:
                    continue
                h1 = W[g][:, 2].min()
                h0 = prevW[g][:, 2].min()
                Mi = Md = 0.0
                for jj in ganc[g]:
                    r = W[g] - anchor[jj]
                    if m.jnt_type[jj] == 2:
                        Mi += dq[jj]
                        Md += abs(axis[jj][2]) * dq[jj]
                    else:
                        Mi += np.linalg.norm(r, axis=1).max() * 1.02 * dq[jj]
                        vz = np.abs(np.cross(axis[jj], r)[:, 2]).max()
                        Md += (vz * 1.1 + 0.05 * np.linalg.norm(r, axis=1).max()) * dq[jj]
                f_iso |= h1 <= Mi
                f_dir |= h1 <= Md
                f_dir_sum |= h0 + h1 <= Md
            stats["fl_iso"] += f_iso
            stats["fl_dir"] += f_dir
            stats["fl_dir_sum"] += f_dir_sum
            stats["any_b"] += fb or f_dir_sum
        prevW, prevq, prevd = W, q, dist
    print("env", e, stats, flush=True)
print("pairs failing (a):", sorted(which_a.items(), key=lambda kv: -kv[1])[:12])
print("pairs failing (b):", sorted(which_b.items(), key=lambda kv: -kv[1])[:12])
