"""Independent first-principles dynamics of the FR3 (+ hand) and xArm7 chains -> tests/golden/{fr3,xarm7}_dynamics.json.

What this is: a Lagrangian derivation that shares NO code and NO formulation with the CPU oracle (oracle/rcs_physics.c:
composite-rigid-body + recursive Newton-Euler about subtree centres of mass, restated from MuJoCo's pipeline) or with the HIP
kernels (csrc/dyn_team.h: world-origin Pluecker scans).  Input: tests/golden/reference_models.json only -- the constants an
independent reader took from the reference's own MJCF files (tools/make_reference_model_fixture.py) -- never rcs_amd/mjcf.py.

Method (mpmath, 40 significant digits, so that every finite difference below is exact to ~1e-24):

  * forward kinematics by products of homogeneous transforms  T_body = T_parent . [pos, quat] . [joint motion];
  * kinetic energy  T = 1/2 sum_b ( m_b |c_b'|^2 + w_b^T I_b w_b ) + 1/2 sum_j armature_j q_j'^2  with the centre-of-mass
    Jacobians obtained by DIFFERENTIATING the forward kinematics (central differences in 40-digit arithmetic), the angular
    Jacobians read off  R' R^T  the same way  =>  M(q) = sum_b m_b Jv^T Jv + Jw^T (R I R^T) Jw + diag(armature);
  * Coriolis / centrifugal forces from the Christoffel symbols of M(q):  c_i = sum_jk (dM_ij/dq_k - 1/2 dM_jk/dq_i) q_j' q_k';
  * gravity from the potential  V = -sum_b m_b g.c_b :  G = dV/dq;   bias = c + G   (MuJoCo's qfrc_bias convention:
    M qacc + qfrc_bias = applied forces);
  * gravity compensation: the force  -gravcomp_b m_b g  at each body's centre of mass, through that body's Jv;
  * actuator / passive forces from the MJCF reference text of the elements (position: kp (ctrl - q) - kv q'; general with
    affine bias: gain ctrl + b0 + b1 length + b2 velocity; ctrlrange / forcerange / actuatorfrcrange clamps; fixed tendon =
    sum coef q; joint damping; `actuatorgravcomp` moves the compensation from qfrc_passive to qfrc_actuator, inside the
    actuatorfrcrange clamp);
  * one implicitfast step of the smooth system,  (M - h dF/dv) a+ = F,  v+ = v + h a+  (dF/dv: joint damping, the actuators'
    velocity gains, symmetric) -- for the models WITHOUT constraint rows (the arms without hand / friction rows) this is the
    whole substep and pins the kernels' result; for FR3 + hand the joint-equality row is added with MuJoCo's documented soft
    constraint (solref / solimp -> K, B, impedance, R = (1 - d)/d . (invweight0_1 + invweight0_2)): the one part of the
    golden file that is still a statement about MuJoCo's constraint model rather than about mechanics, marked as such.

One documented stand-in: body d435i_0's inertial comes from a mesh blob that is missing from the reference checkout; the same
solid 90 x 25 x 25 mm box the authored scene declares is used (scene.xml header).

    python tools/derive_fr3_dynamics.py            (about a minute)
"""
import json
import os

import mpmath as mp
import numpy as np

mp.mp.dps = 40
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODELS = os.path.join(ROOT, "tests", "golden", "reference_models.json")
GRAVITY = mp.matrix([0, 0, mp.mpf("-9.81")])  # MuJoCo's default option gravity; neither robot file overrides it
TIMESTEP = mp.mpf("0.002")  # MuJoCo's default option timestep; neither robot file overrides it
D435I_STANDIN = {"mass": 0.072, "pos": [0, 0, -0.0125], "quat": None, "diaginertia": [7.5e-06, 5.235e-05, 5.235e-05]}


def mpf(x):
    return mp.mpf(repr(float(x)))  # the decimal the MJCF holds, as the double a parser makes of it


def quat_R(q):
    w, x, y, z = [mpf(v) for v in q]
    n = mp.sqrt(w * w + x * x + y * y + z * z)
    w, x, y, z = w / n, x / n, y / n, z / n
    return mp.matrix([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def axis_R(axis, angle):
    """Rodrigues."""
    a = mp.matrix(axis)
    a = a / mp.norm(a)
    K = mp.matrix([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return mp.eye(3) + mp.sin(angle) * K + (1 - mp.cos(angle)) * (K * K)


def euler_R(e):
    """MuJoCo's default eulerseq "xyz": intrinsic rotations about x, then y, then z."""
    return axis_R([1, 0, 0], mpf(e[0])) * axis_R([0, 1, 0], mpf(e[1])) * axis_R([0, 0, 1], mpf(e[2]))


class Chain:
    def __init__(self, robot, drop=()):
        """`drop`: names of bodies removed together with their subtrees (the arm without its hand)."""
        self.bodies, self.dofs = [], []
        names = {}
        dropped = set()
        for b in robot["bodies"]:
            if b["name"] in drop or b["parent"] in dropped or any(j["type"] == "free" for j in b["joints"]):
                dropped.add(b["name"])
                continue
            R0 = quat_R(b["quat"]) if b["quat"] else euler_R(b["euler"]) if b["euler"] else mp.eye(3)
            inert = b.get("inertial") or (D435I_STANDIN if b["name"] == "d435i_0" else None)
            rec = {"name": b["name"], "parent": names.get(b["parent"], -1), "pos": mp.matrix([mpf(v) for v in b["pos"]]), "R0": R0, "dof": -1,
                   "gravcomp": mpf(b["gravcomp"]), "mass": mp.mpf(0), "ipos": mp.zeros(3, 1), "iR": mp.eye(3), "inertia": mp.zeros(3, 1)}
            if inert:
                rec.update(mass=mpf(inert["mass"]), ipos=mp.matrix([mpf(v) for v in inert["pos"]]), inertia=mp.matrix([mpf(v) for v in inert["diaginertia"]]),
                           iR=quat_R(inert["quat"]) if inert.get("quat") else mp.eye(3))
            assert len(b["joints"]) <= 1
            for j in b["joints"]:
                rec["dof"] = len(self.dofs)
                self.dofs.append({**j, "body": len(self.bodies)})
            names[b["name"]] = len(self.bodies)
            self.bodies.append(rec)
        self.nv = len(self.dofs)
        self.armature = [mpf(j["armature"]) for j in self.dofs]
        self.damping = [mpf(j["damping"]) for j in self.dofs]

    def fk(self, q):
        """World rotation and origin of every body."""
        out = []
        for b in self.bodies:
            Rp, pp = out[b["parent"]] if b["parent"] >= 0 else (mp.eye(3), mp.zeros(3, 1))
            R, p = Rp * b["R0"], pp + Rp * b["pos"]
            if b["dof"] >= 0:
                j = self.dofs[b["dof"]]
                ax = [mpf(v) for v in j["axis"]]
                if j["type"] == "hinge":
                    Q = axis_R(ax, q[b["dof"]])
                    if j.get("pos"):  # anchor off the body origin (the builder-authored arm6 scene): rotate about the axis THROUGH it
                        a0 = mp.matrix([mpf(v) for v in j["pos"]])
                        p = p + R * (a0 - Q * a0)
                    R = R * Q
                else:
                    a = mp.matrix(ax)
                    p = p + R * (a / mp.norm(a)) * q[b["dof"]]
            out.append((R, p))
        return out

    def coms(self, q):
        return [(R, p + R * b["ipos"]) for b, (R, p) in zip(self.bodies, self.fk(q))]

    def jacobians(self, q, h=mp.mpf(10) ** -14):
        """Jv[b] (3 x nv) and Jw[b] (3 x nv) by differentiating the forward kinematics."""
        base = self.coms(q)
        Jv = [mp.zeros(3, self.nv) for _ in self.bodies]
        Jw = [mp.zeros(3, self.nv) for _ in self.bodies]
        for k in range(self.nv):
            qp, qm = list(q), list(q)
            qp[k] += h
            qm[k] -= h
            P, Mi = self.coms(qp), self.coms(qm)
            for b in range(len(self.bodies)):
                dc = (P[b][1] - Mi[b][1]) / (2 * h)
                W = ((P[b][0] - Mi[b][0]) / (2 * h)) * base[b][0].T  # R' R^T = [w]x
                for r in range(3):
                    Jv[b][r, k] = dc[r]
                Jw[b][0, k], Jw[b][1, k], Jw[b][2, k] = W[2, 1], W[0, 2], W[1, 0]
        return base, Jv, Jw

    def mass_matrix(self, q):
        base, Jv, Jw = self.jacobians(q)
        M = mp.zeros(self.nv, self.nv)
        for b, rec in enumerate(self.bodies):
            if rec["mass"] == 0:
                continue
            Ri = base[b][0] * rec["iR"]
            Iw = Ri * mp.diag(list(rec["inertia"])) * Ri.T
            M += rec["mass"] * (Jv[b].T * Jv[b]) + Jw[b].T * Iw * Jw[b]
        for i in range(self.nv):
            M[i, i] += self.armature[i]
        return M

    def potential(self, q):
        return -sum(rec["mass"] * (GRAVITY.T * c)[0] for rec, (_, c) in zip(self.bodies, self.coms(q)))

    def bias(self, q, v, h=mp.mpf(10) ** -10):
        nv = self.nv
        dM = []
        G = mp.zeros(nv, 1)
        for k in range(nv):
            qp, qm = list(q), list(q)
            qp[k] += h
            qm[k] -= h
            dM.append((self.mass_matrix(qp) - self.mass_matrix(qm)) / (2 * h))
            G[k] = (self.potential(qp) - self.potential(qm)) / (2 * h)
        c = mp.zeros(nv, 1)
        for i in range(nv):
            c[i] = sum((dM[k][i, j] - dM[i][j, k] / 2) * v[j] * v[k] for j in range(nv) for k in range(nv))
        return c + G, c, G

    def gravcomp(self, q):
        _, Jv, _ = self.jacobians(q)
        f = mp.zeros(self.nv, 1)
        for b, rec in enumerate(self.bodies):
            if rec["gravcomp"] != 0 and rec["mass"] != 0:
                f += Jv[b].T * (-rec["gravcomp"] * rec["mass"] * GRAVITY)
        return f


def clamp(x, lo, hi):
    return min(max(x, lo), hi)


def actuation(robot, chain, q, v, ctrl):
    """(qfrc from the actuators alone, d qfrc / d v as a matrix, per-actuator force)."""
    nv = chain.nv
    names = [j["name"] for j in chain.dofs]
    tendons = {t["name"]: t["joints"] for t in robot["tendons"]}
    qfrc, dv, forces = mp.zeros(nv, 1), mp.zeros(nv, nv), []
    acts = [a for a in robot["actuators"] if a.get("joint") in names or (a.get("tendon") and all(j in names for j, _ in tendons[a["tendon"]]))]
    for a, u in zip(acts, ctrl):
        moment = mp.zeros(nv, 1)
        if a.get("joint"):
            moment[names.index(a["joint"])] = 1
        else:
            for jn, coef in tendons[a["tendon"]]:
                moment[names.index(jn)] = mpf(coef)
        length, velocity = sum(moment[i] * q[i] for i in range(nv)), sum(moment[i] * v[i] for i in range(nv))
        if a["tag"] == "position":
            kp, kv = mpf(a["kp"]), mpf(a.get("kv", 0))
            gain, b0, b1, b2 = kp, mp.mpf(0), -kp, -kv
            j = chain.dofs[names.index(a["joint"])]
            crange = [mpf(x) for x in j["range"]] if a.get("inheritrange") else None  # inheritrange = 1: ctrlrange := the joint's range
            frange = None
        else:
            gain = mpf(a["gainprm"][0])
            b0, b1, b2 = [mpf(x) for x in (a["biasprm"] + [0, 0, 0])[:3]]  # affine bias (the FR3 gripper inherits it from its class)
            crange = [mpf(x) for x in a["ctrlrange"]] if a.get("ctrlrange") else None
            frange = [mpf(x) for x in a["forcerange"]] if a.get("forcerange") else None
        u = clamp(mpf(u), *crange) if crange else mpf(u)
        f = gain * u + b0 + b1 * length + b2 * velocity
        saturated = False
        if frange:
            saturated = not (frange[0] < f < frange[1])
            f = clamp(f, *frange)
        forces.append(f)
        qfrc += moment * f
        if not saturated:
            dv += moment * moment.T * b2
    return qfrc, dv, forces


def impedance(solimp, r):
    d0, dmax, width, mid, power = [mpf(x) for x in (list(solimp) + [0.9, 0.95, 0.001, 0.5, 2][len(solimp):])]
    x = abs(r) / width
    if x >= 1:
        return dmax
    if x == 0:
        return d0
    if power == 1:
        y = x
    elif x <= mid:
        y = x ** power / mid ** (power - 1)
    else:
        y = 1 - (1 - x) ** power / (1 - mid) ** (power - 1)
    return d0 + y * (dmax - d0)


def sample(robot, chain, rng, with_equality, gentle=False):
    """`gentle`: small tracking errors and velocities, so that no actuator force reaches a clamp (the other half of the samples
    saturates `actuatorfrcrange` / `forcerange` on most joints)."""
    nv = chain.nv
    lo = np.array([j["range"][0] if j["range"] else -3.0 for j in chain.dofs])
    hi = np.array([j["range"][1] if j["range"] else 3.0 for j in chain.dofs])
    lo, hi = np.maximum(lo, -3.0), np.minimum(hi, 3.0)
    span = hi - lo
    qn = lo + span * rng.uniform(0.15, 0.85, nv)  # away from the joint limits: no limit row
    vn = rng.uniform(-1.0, 1.0, nv) * (0.01 if gentle else 1.0)
    names = [j["name"] for j in chain.dofs]
    slide = np.array([j["type"] == "slide" for j in chain.dofs])
    vn[slide] *= 0.05
    tendons_ = {t["name"]: t["joints"] for t in robot["tendons"]}
    ctrl = []
    for a in robot["actuators"]:  # (the same selection and order as actuation() below)
        if a.get("joint") in names:
            ctrl.append(float(qn[names.index(a["joint"])] + rng.uniform(-0.15, 0.15) * (0.01 if gentle else 1.0)))
        elif a.get("tendon") and all(j in names for j, _ in tendons_[a["tendon"]]):
            ctrl.append(float(rng.uniform(0, 255)))
    q, v = [mpf(x) for x in qn], [mpf(x) for x in vn]
    M = chain.mass_matrix(q)
    bias, cor, grav = chain.bias(q, v)
    gc = chain.gravcomp(q)
    act, dact, forces = actuation(robot, chain, q, v, ctrl)
    passive = mp.matrix([-chain.damping[i] * v[i] for i in range(nv)])
    qfrc_actuator = act.copy()
    dF = dact.copy()
    for i, j in enumerate(chain.dofs):
        dF[i, i] -= chain.damping[i]
        if j["actuatorgravcomp"]:
            qfrc_actuator[i] += gc[i]
        else:
            passive[i] += gc[i]
        if j["actuatorfrcrange"]:
            qfrc_actuator[i] = clamp(qfrc_actuator[i], *[mpf(x) for x in j["actuatorfrcrange"]])
    smooth = passive - bias + qfrc_actuator
    qacc_smooth = mp.lu_solve(M, smooth)
    h = TIMESTEP
    out = {"qpos": qn.tolist(), "qvel": vn.tolist(), "ctrl": [float(c) for c in ctrl]}
    constraint = mp.zeros(nv, 1)
    if with_equality:
        e = robot["equality"][0]
        i1, i2 = names.index(e["joint1"]), names.index(e["joint2"])
        # diagonal of the inverse inertia at qpos0 = 0 (mjModel.dof_invweight0 of a single-dof joint)
        Minv0 = M0_inverse(chain)
        diag_approx = Minv0[i1, i1] + Minv0[i2, i2]
        r = q[i1] - q[i2]
        J = mp.zeros(1, nv)
        J[0, i1], J[0, i2] = 1, -1
        tc, dr = [mpf(x) for x in e["solref"]]
        tc = max(tc, 2 * h)
        imp = impedance(e["solimp"], r)
        dmax = mpf(e["solimp"][1])
        K, B = 1 / (dmax * dmax * tc * tc * dr * dr), 2 / (dmax * tc)
        aref = -B * (J * mp.matrix(v))[0] - K * imp * r
        R = (1 - imp) / imp * diag_approx
        Dd = 1 / R
        qacc = mp.lu_solve(M + J.T * J * Dd, M * qacc_smooth + J.T * (Dd * aref))
        force = -Dd * ((J * qacc)[0] - aref)
        constraint = J.T * force
        out.update(efc_aref=float(aref), efc_R=float(R), efc_force=float(force), qacc_constrained=[float(x) for x in qacc])
    qacc_next = mp.lu_solve(M - h * dF, smooth + constraint)
    v_next = [v[i] + h * qacc_next[i] for i in range(nv)]
    q_next = [q[i] + h * v_next[i] for i in range(nv)]
    f = lambda m: [float(x) for x in m]  # noqa: E731
    out.update(qM=[[float(M[i, j]) for j in range(nv)] for i in range(nv)], qfrc_bias=f(bias), coriolis=f(cor), gravity=f(grav), qfrc_gravcomp=f(gc),
               qfrc_passive=f(passive), qfrc_actuator=f(qfrc_actuator), actuator_force=f(forces), qfrc_smooth=f(smooth), qacc_smooth=f(qacc_smooth),
               qfrc_constraint=f(constraint), qacc_implicit=f(qacc_next), qvel_next=f(v_next), qpos_next=f(q_next))
    return out


_M0 = {}


def M0_inverse(chain):
    if id(chain) not in _M0:
        _M0[id(chain)] = mp.inverse(chain.mass_matrix([mp.mpf(0)] * chain.nv))
    return _M0[id(chain)]


def self_check(chain, rng):
    """The differentiated Jacobians against the textbook geometric ones (hinge: a x (c - p), a; slide: a, 0)."""
    q = [mpf(x) for x in rng.uniform(-1, 1, chain.nv)]
    frames = chain.fk(q)
    base, Jv, Jw = chain.jacobians(q)
    worst = mp.mpf(0)
    for b, rec in enumerate(chain.bodies):
        anc, k = [], b
        while k >= 0:
            anc.append(k)
            k = chain.bodies[k]["parent"]
        for j, dof in enumerate(chain.dofs):
            col_v, col_w = mp.zeros(3, 1), mp.zeros(3, 1)
            if dof["body"] in anc:
                Rj, pj = frames[dof["body"]]
                a = mp.matrix([mpf(x) for x in dof["axis"]])
                a = Rj * (a / mp.norm(a))
                if dof.get("pos"):
                    pj = pj + Rj * mp.matrix([mpf(x) for x in dof["pos"]])
                if dof["type"] == "hinge":
                    d = base[b][1] - pj
                    col_v = mp.matrix([a[1] * d[2] - a[2] * d[1], a[2] * d[0] - a[0] * d[2], a[0] * d[1] - a[1] * d[0]])
                    col_w = a
                else:
                    col_v = a
            for r in range(3):
                worst = max(worst, abs(Jv[b][r, j] - col_v[r]), abs(Jw[b][r, j] - col_w[r]))
    assert worst < mp.mpf(10) ** -22, worst
    return float(worst)


def main(n_samples=32):
    models = json.load(open(MODELS))
    jobs = [("fr3", "fr3", (), True, "FR3 + Franka hand (9 dof, joint-equality row between the fingers)"),
            ("fr3_arm", "fr3", ("hand_0",), False, "FR3 without its hand (7 dof, no constraint row): the substep is the smooth system alone"),
            ("xarm7", "xarm7", (), False, "xArm7 (7 dof); the smooth system -- its dry-friction rows are not part of these vectors")]
    for tag, key, drop, eq, what in jobs:
        chain = Chain(models[key], drop)
        rng = np.random.default_rng({"fr3": 11, "fr3_arm": 12, "xarm7": 13}[tag])
        err = self_check(chain, rng)
        out = {"what": what, "source": "tools/derive_fr3_dynamics.py over tests/golden/reference_models.json (mpmath, 40 digits)",
               "nv": chain.nv, "dof_names": [j["name"] for j in chain.dofs], "timestep": float(TIMESTEP), "gravity": [0, 0, -9.81],
               "jacobian_self_check": err, "dropped_bodies": list(drop), "samples": [sample(models[key], chain, rng, eq, gentle=bool(i % 2)) for i in range(n_samples)]}
        path = os.path.join(ROOT, "tests", "golden", f"{tag}_dynamics.json")
        json.dump(out, open(path, "w"), indent=0)
        print("wrote", path, "nv", chain.nv, "jacobian check", err)


def main_authored(n_samples=16):
    """The builder-authored scenes -- no reference model exists for them, so nothing pins what the shared MJCF compiler makes of
    them.  The same derivation over THEIR files, read by the same independent reader
    (tools/make_reference_model_fixture.read_robot): what oracle and kernels compute for these robots is then held to first
    principles too -- general joint axes and off-origin anchors (arm6), the 5-dof arm with its gripper (so101), the UR5e
    proportions, the xArm7 with the Franka hand (xarm7_pick_world; one-substep prediction for its variant without dry friction)."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from make_reference_model_fixture import read_robot

    scenes = os.path.join(ROOT, "robot-control-stack_amd", "rcs_amd", "scenes")
    out = {"source": "tools/derive_fr3_dynamics.py main_authored() over the repository's own scene files (independent reader)", "timestep": float(TIMESTEP),
           "gravity": [0, 0, -9.81], "models": {}}
    for tag, scene, eq in (("ur5e", "ur5e_empty_world", False), ("arm6", "arm6_empty_world", False), ("so101", "so101_empty_world", True),
                           ("xarm7_pick", "xarm7_pick_world", True)):
        robot = read_robot(os.path.join(scenes, scene, "scene.xml"))
        chain = Chain(robot)
        rng = np.random.default_rng(sum(map(ord, tag)))
        err = self_check(chain, rng)
        samples = [sample(robot, chain, rng, eq, gentle=bool(i % 2)) for i in range(n_samples)]
        out["models"][tag] = {"scene": scene, "nv": chain.nv, "dof_names": [j["name"] for j in chain.dofs], "jacobian_self_check": err,
                              "frictionloss": [float(j["frictionloss"]) for j in chain.dofs], "samples": samples}
        print(tag, "nv", chain.nv, "jacobian check", err)
    path = os.path.join(ROOT, "tests", "golden", "authored_scene_dynamics.json")
    json.dump(out, open(path, "w"), indent=0)
    print("wrote", path)


if __name__ == "__main__":
    import sys

    if "--authored" in sys.argv:
        main_authored()
    else:
        main()
