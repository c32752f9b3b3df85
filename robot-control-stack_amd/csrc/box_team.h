// box_team.h -- one free rigid box on the floor plane, stepped inside the team kernel (team.h: 16 lanes per
// environment).  This is the `box_geom` body of the reference's pick-up scene
// (assets/scenes/fr3_simple_pick_up/scene.xml:30-33; solver options assets/fr3/mjcf/fr3_common.xml:3: elliptic
// friction cones, impratio 20, 5 noslip iterations), which RandomCubePos places and PickCubeSuccessWrapper reads
// (python/rcs/envs/sim.py:358-431).
//
// While no robot geom touches the box (contact_team.h otherwise: `coupled`) it shares no constraint row with the robot, so its
// 6 degrees of freedom are a separate block of the constrained forward dynamics: free-joint kinematics, plane-box collision (at most four
// corner contacts), one elliptic-cone contact of three rows per corner, Newton on the primal cost with an exact line
// search, the noslip post-pass (Gauss-Seidel over the contacts' friction rows without regularisation; deliberately
// only a few sweeps, as in MuJoCo), semi-implicit Euler with quaternion integration.
//
// Lane roles: the box state (19 doubles) sits in the team's LDS block and every lane of the team carries a copy in
// registers; lane t owns contact t & 3 -- its three Jacobian rows, its cone zone, its force -- and the four contacts'
// contributions to cost, gradient, Hessian and joint force are added across the 4-lane quad with two DPP quad_perm
// steps, so all 16 lanes hold identical sums and factor the same 6x6 Hessian redundantly (the instructions are
// issued for the wave anyway).  The noslip sweep is sequential by nature: contact c's owner computes, quad_perm
// broadcasts.
#pragma once
#include "dyn_team.h"
#include "ik.h"
#include "team.h"

namespace rcsh {

// constants of the box and of its contact with the floor, host-prepared (rcs_hip.hip: rcsh_sim_add_free_box)
struct BoxCfg {
  int32_t present, noslip_iterations;
  double qpos0[7];  // x y z, qw qx qy qz
  double mass, inertia[3], inv_mass, inv_inertia[3];
  double size[3];   // half extents
  double fr;        // sliding friction coefficient of the (floor, box) pair (condim 3: both tangential directions)
  double geom_mu;   // the box geom's own sliding friction (mixed with a robot geom's per contact)
  int32_t resolve, pad0;  // contacts of robot geoms enter the constraint solve (contact_team.h)
  double K, B;      // stiffness / damping of the reference acceleration (solref)
  Imp imp;          // solimp
  double inv_impratio;
  double plane_z;
  double scale;     // 1 / (meaninertia * nv) of the whole scene: MuJoCo's solver-statistics scaling
  double noslip_tolerance;
};

// per-team LDS block: qpos 7, qvel 6, qacc_warmstart 6, then the 12 x 6 contact Jacobian for the noslip pass
// (and the pose the last position stage saw: what the renderer draws)
// (kBoxA: the box's acceleration out of a coupled robot + box solve; kBoxX: the minimiser the last coupled Newton solve found
// over the robot's and the box's dofs -- a third candidate for the next solve's starting point, contact_team.h)
constexpr int kBoxQ = 0, kBoxV = 7, kBoxW = 13, kBoxPre = 19, kBoxX = 26, kBoxState = 41, kBoxJ = 41, kBoxA = kBoxJ + 72, kBoxLds = kBoxA + 6;

#if defined(__HIP__)

template <int CTRL>
RCSH_D double quad_perm(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, lo32(x), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, hi32(x), CTRL, 0xf, 0xf, true);
  return mk64(hi, lo);
}
// sum over the 4 lanes of the quad, the same bits on all of them
RCSH_D double quad_sum(double x) {
  x += quad_perm<0xB1>(x);  // [1,0,3,2]
  x += quad_perm<0x4E>(x);  // [2,3,0,1]
  return x;
}
// value of lane C of the quad
template <int C>
RCSH_D double quad_bcast(double x) { return quad_perm<C * 0x55>(x); }

// one contact: rows in the contact frame (normal, two tangents), regularisers, reference accelerations
struct BoxContact {
  double J[3][6], aref[3], D[3], R[3], mu;
};

// elliptic-cone cost of one contact at jar = J qacc - aref: force f = -dcost/djar and Hessian Hc (lower triangle
// 00 10 11 20 21 22).  Zones: top (separating, no force), bottom (sticking: quadratic in every row), middle
// (sliding: 0.5 Dm (N - mu T)^2).
RCSH_D double box_cone(const BoxContact& c, double fr, const double* jar, double* f, double* Hc) {
  const double mu = c.mu;
  const double U0 = jar[0] * mu, U1 = jar[1] * fr, U2 = jar[2] * fr;
  const double N = U0, T = sqrt(U1 * U1 + U2 * U2);
#pragma unroll
  for (int k = 0; k < 6; ++k) Hc[k] = 0.0;
  f[0] = f[1] = f[2] = 0.0;
  if (N >= mu * T) return 0.0;
  if (mu * N + T <= 0) {
    double cost = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      cost += 0.5 * c.D[k] * jar[k] * jar[k];
      f[k] = -c.D[k] * jar[k];
    }
    Hc[0] = c.D[0]; Hc[2] = c.D[1]; Hc[5] = c.D[2];
    return cost;
  }
  const double Dm = c.D[0] * fast_rcp(mu * mu * (1 + mu * mu));
  const double NmT = N - mu * T;
  const double iT = fast_rcp(T), u1 = U1 * iT, u2 = U2 * iT;
  f[0] = -mu * (Dm * NmT);
  f[1] = fr * (Dm * NmT * mu * u1);
  f[2] = fr * (Dm * NmT * mu * u2);
  const double k = -Dm * NmT * mu * iT;  // > 0
  Hc[0] = mu * mu * Dm;
  Hc[1] = mu * fr * (-Dm * mu * u1);
  Hc[3] = mu * fr * (-Dm * mu * u2);
  Hc[2] = fr * fr * (Dm * mu * mu * u1 * u1 + k * (1 - u1 * u1));
  Hc[4] = fr * fr * (Dm * mu * mu * u2 * u1 + k * (-u2 * u1));
  Hc[5] = fr * fr * (Dm * mu * mu * u2 * u2 + k * (1 - u2 * u2));
  return 0.5 * Dm * NmT * NmT;
}

RCSH_D void box_jar(const BoxContact& c, const double* x, double* jar) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    double s = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) s += c.J[k][j] * x[j];
    jar[k] = s - c.aref[k];
  }
}

// total primal cost at x (Gauss term + the four contacts)
RCSH_D double box_cost(const BoxCfg& b, const BoxContact& c, const double* Md, const double* xs, const double* x) {
  double jar[3], f[3], Hc[6];
  box_jar(c, x, jar);
  double cost = quad_sum(box_cone(c, b.fr, jar, f, Hc));
#pragma unroll
  for (int j = 0; j < 6; ++j) cost += 0.5 * Md[j] * (x[j] - xs[j]) * (x[j] - xs[j]);
  return cost;
}

// mju_QCQP2: min 0.5 x'Ax + x'b  s.t.  (x0^2 + x1^2) / d^2 <= r^2 (both friction coefficients equal d)
RCSH_D bool box_qcqp2(double* res, double A00, double A01, double A11, double b0, double b1, double d, double r) {
  const double s1 = b0 * d, s2 = b1 * d;
  const double S11 = A00 * d * d, S22 = A11 * d * d, S12 = A01 * d * d;
  double la = 0, v1 = 0, v2 = 0;
  for (int iter = 0; iter < 20; ++iter) {
    TEAM_COUNT(23)
    const double det = (S11 + la) * (S22 + la) - S12 * S12;
    if (det < 1e-10) {
      res[0] = res[1] = 0;
      return false;
    }
    const double di = fast_rcp(det), P11 = (S22 + la) * di, P22 = (S11 + la) * di, P12 = -S12 * di;
    v1 = -P11 * s1 - P12 * s2;
    v2 = -P12 * s1 - P22 * s2;
    const double val = v1 * v1 + v2 * v2 - r * r;
    if (val < 1e-10) break;
    const double deriv = -2 * (P11 * v1 * v1 + 2 * P12 * v1 * v2 + P22 * v2 * v2);
    const double delta = -val * fast_rcp(deriv);
    if (delta < 1e-10) break;
    la += delta;
  }
  res[0] = v1 * d;
  res[1] = v2 * d;
  return la != 0;
}

// One substep of the box.  bs: the team's LDS block (kBoxLds doubles); improvement0: 0.5 f^2 R summed over the
// robot's non-equality constraint rows (MuJoCo's first noslip sweep counts every such row of the scene).
// coupled (team-uniform): the step's constraints were solved together with the robot's (contact_team.h), the box's
// acceleration is in bs[kBoxA..]: only the position stage's bookkeeping and the integration are left.
RCSH_D void box_substep(const BoxCfg& b, double* bs, const double* gravity, double h, double improvement0, int t, bool coupled) {
  constexpr double kMin = 1e-15;
  double p[3], q[4], v[6], warm[6];
#pragma unroll
  for (int k = 0; k < 3; ++k) p[k] = bs[kBoxQ + k];
#pragma unroll
  for (int k = 0; k < 4; ++k) q[k] = bs[kBoxQ + 3 + k];
#pragma unroll
  for (int k = 0; k < 6; ++k) { v[k] = bs[kBoxV + k]; warm[k] = bs[kBoxW + k]; }
  TEAM_MARK(9)
  // ---- mj_kinematics: normalise the quaternion in qpos, frame of the box
  {
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n < kMin) { q[0] = 1; q[1] = q[2] = q[3] = 0; }
    else if (fabs(n - 1) > kMin) {
      const double s = 1 / n;
#pragma unroll
      for (int k = 0; k < 4; ++k) q[k] *= s;
    }
  }
  if (t == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) bs[kBoxPre + k] = p[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) bs[kBoxPre + 3 + k] = q[k];
  }
  double R[9];
  {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = w * w + x * x - y * y - z * z; R[4] = w * w - x * x + y * y - z * z; R[8] = w * w - x * x - y * y + z * z;
    R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y); R[3] = 2 * (x * y + w * z);
    R[5] = 2 * (y * z - w * x); R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x);
  }
  // ---- mjc_PlaneBox: lane t tests corner t & 7 (height of the corner over the plane); the first four penetrating
  // corners in corner order are the contacts
  const double dist = p[2] - b.plane_z;
  auto corner_z = [&](int i) {
    return R[6] * (i & 1 ? b.size[0] : -b.size[0]) + R[7] * (i & 2 ? b.size[1] : -b.size[1]) + R[8] * (i & 4 ? b.size[2] : -b.size[2]);
  };
  uint32_t hits;
  {
    const double ld = corner_z(t & 7);
    hits = team_ballot(!(dist + ld > 0 || ld > 0)) & 0xffu;
  }
  const int ncon = min(4, (int)__popc(hits));
  // ---- the lane's contact: slot t & 3
  BoxContact c;
  {
    const int slot = t & 3;
    uint32_t mk = hits;
    for (int k = 0; k < slot; ++k) mk &= mk - 1;
    const bool has = mk != 0;
    const int ci = has ? __ffs(mk) - 1 : 0;
    const double vec[3] = {ci & 1 ? b.size[0] : -b.size[0], ci & 2 ? b.size[1] : -b.size[1], ci & 4 ? b.size[2] : -b.size[2]};
    double corner[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) corner[r] = R[3 * r] * vec[0] + R[3 * r + 1] * vec[1] + R[3 * r + 2] * vec[2];
    const double cdist = dist + corner[2];
    // contact point relative to the box origin: corner - normal * cdist / 2
    const double rr[3] = {corner[0], corner[1], corner[2] - 0.5 * cdist};
    // frame: normal (0,0,1), tangents (0,1,0) and (-1,0,0) (mju_makeFrame); row k: [a_k, R^T (r x a_k)]
    const double fr3[3][3] = {{0, 0, 1}, {0, 1, 0}, {-1, 0, 0}};
    const double imp = impedance(b.imp, cdist, 0.0);
    double R0 = (1 - imp) / imp * b.inv_mass;
    if (R0 < kMin) R0 = kMin;
    const double R1 = R0 * b.inv_impratio;
    const double Rk[3] = {R0, R1, R1 * b.fr * b.fr / (b.fr * b.fr)};
    c.mu = b.fr * sqrt(R1 / R0);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      double rxa[3];
      cross3(rr, fr3[k], rxa);
      double vel = 0;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        c.J[k][j] = has ? fr3[k][j] : 0.0;
        c.J[k][3 + j] = has ? R[j] * rxa[0] + R[3 + j] * rxa[1] + R[6 + j] * rxa[2] : 0.0;
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) vel += c.J[k][j] * v[j];
      c.R[k] = has ? Rk[k] : 0.0;
      c.D[k] = has ? 1 / Rk[k] : 0.0;
      c.aref[k] = has ? -b.B * vel - (k == 0 ? b.K * imp * cdist : 0.0) : 0.0;
    }
    if (t < 4) {
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < 6; ++j) bs[kBoxJ + 6 * (3 * t + k) + j] = c.J[k][j];
    }
  }
  // ---- mj_fwdAcceleration: gravity and the gyroscopic torque (angular velocity in the body frame)
  const double Md[6] = {b.mass, b.mass, b.mass, b.inertia[0], b.inertia[1], b.inertia[2]};
  const double Mi[6] = {b.inv_mass, b.inv_mass, b.inv_mass, b.inv_inertia[0], b.inv_inertia[1], b.inv_inertia[2]};
  double xs[6];
  {
    const double* w = v + 3;
    const double Iw[3] = {b.inertia[0] * w[0], b.inertia[1] * w[1], b.inertia[2] * w[2]};
    double gyro[3];
    cross3(w, Iw, gyro);
#pragma unroll
    for (int j = 0; j < 3; ++j) { xs[j] = gravity[j]; xs[3 + j] = -gyro[j] * b.inv_inertia[j]; }
  }
  double x[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) x[j] = xs[j];
  TEAM_MARK(16)
  if (coupled) {
#pragma unroll
    for (int j = 0; j < 6; ++j) x[j] = bs[kBoxA + j];
  } else if (ncon > 0) {
    // ---- Newton on the primal cost; warm start = the cheaper of qacc_warmstart and qacc_smooth
    if (!(box_cost(b, c, Md, xs, xs) < box_cost(b, c, Md, xs, warm))) {
#pragma unroll
      for (int j = 0; j < 6; ++j) x[j] = warm[j];
    }
    double f[3];
    bool f_at_x = false;  // f holds the forces at the current x (the loop leaves through one of its two breaks: right after computing them)
    TEAM_MARK(17)
    for (int it = 0; it < 50; ++it) {
      TEAM_COUNT(21)
      double jar[3], Hc[6], grad[6], H[36];
      box_jar(c, x, jar);
      box_cone(c, b.fr, jar, f, Hc);
      f_at_x = true;
      // gradient first: the Hessian is only assembled when another step follows
      double g2 = 0;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        grad[i] = Md[i] * (x[i] - xs[i]) - quad_sum(c.J[0][i] * f[0] + c.J[1][i] * f[1] + c.J[2][i] * f[2]);
        g2 += grad[i] * grad[i];
      }
      if (b.scale * sqrt(g2) < 1e-13) break;
      {
        // the lane's J' Hc J, then the quad sums
        double W[3][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          W[0][j] = Hc[0] * c.J[0][j] + Hc[1] * c.J[1][j] + Hc[3] * c.J[2][j];
          W[1][j] = Hc[1] * c.J[0][j] + Hc[2] * c.J[1][j] + Hc[4] * c.J[2][j];
          W[2][j] = Hc[3] * c.J[0][j] + Hc[4] * c.J[1][j] + Hc[5] * c.J[2][j];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int j = 0; j <= i; ++j)
            H[6 * i + j] = quad_sum(c.J[0][i] * W[0][j] + c.J[1][i] * W[1][j] + c.J[2][i] * W[2][j]) + (i == j ? Md[i] : 0.0);
      }
      double d[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) d[j] = -grad[j];
      ldl6_solve(H, d);
      double dphi0 = 0, dMd = 0;
#pragma unroll
      for (int j = 0; j < 6; ++j) { dphi0 += grad[j] * d[j]; dMd += Md[j] * d[j] * d[j]; }
      if (!(dphi0 < 0)) break;
      // line search: root of phi'(a) by safeguarded 1-D Newton, to 1e-3 (the outer loop's gradient test sets the accuracy
      // of the solution; MuJoCo's own line search stops at ls_tolerance = 0.01); a = 1 is exact unless a contact changes zone
      double jd[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        double s = 0;
#pragma unroll
        for (int j = 0; j < 6; ++j) s += c.J[k][j] * d[j];
        jd[k] = s;
      }
      double lo = 0, hi = -1, a = 1, best = 1, dx = 1e300, dxold = 1e300;
      for (int ls = 0; ls < 20; ++ls) {
        TEAM_COUNT(22)
        double ja[3], fa[3], Ha[6];
#pragma unroll
        for (int k = 0; k < 3; ++k) ja[k] = jar[k] + a * jd[k];
        box_cone(c, b.fr, ja, fa, Ha);
        double dphi = -(jd[0] * fa[0] + jd[1] * fa[1] + jd[2] * fa[2]);
        double ddphi = jd[0] * (Ha[0] * jd[0] + Ha[1] * jd[1] + Ha[3] * jd[2]) + jd[1] * (Ha[1] * jd[0] + Ha[2] * jd[1] + Ha[4] * jd[2]) +
                       jd[2] * (Ha[3] * jd[0] + Ha[4] * jd[1] + Ha[5] * jd[2]);
        dphi = quad_sum(dphi);
        ddphi = quad_sum(ddphi) + dMd;
#pragma unroll
        for (int j = 0; j < 6; ++j) dphi += Md[j] * (x[j] + a * d[j] - xs[j]) * d[j];
        best = a;
        if (fabs(dphi) <= 1e-3 * fabs(dphi0)) break;
        if (dphi < 0) lo = a; else hi = a;
        double an = a - dphi * fast_rcp(ddphi);
        // Newton on phi' with the bracket as the safeguard (the rtsafe rule): bisect when the step leaves the bracket or does
        // not at least halve the step before last -- phi' is piecewise smooth, between two pieces Newton alone can cycle
        if (hi > 0 && (!(an > lo && an < hi) || fabs(2 * dphi) > fabs(dxold * ddphi))) an = 0.5 * (lo + hi);
        if (hi < 0 && !(an > lo)) an = 2 * a;
        if (fabs(an - a) <= 1e-3 * a) break;
        dxold = dx;
        dx = an - a;
        a = an;
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) x[j] += best * d[j];
      f_at_x = false;
    }
    if (!f_at_x) {  // (only when the loop ran into its cap)
      double jar[3], Hc[6];
      box_jar(c, x, jar);
      box_cone(c, b.fr, jar, f, Hc);  // forces at the solution
    }
    TEAM_MARK(18)
    if (b.noslip_iterations > 0) {
      // ---- mj_solNoSlip on the friction rows.  The lane's rows of A = J M^-1 J' (no regulariser) and of
      // b = J qacc_smooth - aref; all 12 forces on every lane.
      // (only the two friction rows are updated, the normal forces stay: row 0 of A and the normal forces' part of the
      // residual are constants of the pass)
      double A[3][12], bb[3], force[12];
#pragma unroll
      for (int k = 1; k < 3; ++k) {
        double jm[6], s = 0;
#pragma unroll
        for (int j = 0; j < 6; ++j) { jm[j] = c.J[k][j] * Mi[j]; s += c.J[k][j] * xs[j]; }
        bb[k] = s - c.aref[k];
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          double a = 0;
#pragma unroll
          for (int j = 0; j < 6; ++j) a += jm[j] * bs[kBoxJ + 6 * i + j];
          A[k][i] = a;
        }
      }
      // gather: force[3 s + k] of slot s from that slot's lane
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        force[k] = quad_bcast<0>(f[k]); force[3 + k] = quad_bcast<1>(f[k]);
        force[6 + k] = quad_bcast<2>(f[k]); force[9 + k] = quad_bcast<3>(f[k]);
      }
#pragma unroll
      for (int k = 1; k < 3; ++k) bb[k] += A[k][0] * force[0] + A[k][3] * force[3] + A[k][6] * force[6] + A[k][9] * force[9];
      int iter = 0;
      while (iter < b.noslip_iterations) {
        double improvement = 0;
        if (iter == 0)
          improvement = improvement0 + quad_sum(0.5 * (f[0] * f[0] * c.R[0] + f[1] * f[1] * c.R[1] + f[2] * f[2] * c.R[2]));
        auto sweep = [&](auto slot_c) {
          constexpr int C = decltype(slot_c)::value;
          if (C >= ncon) return;
          // the owner of contact C (lane C of every quad) runs the update on its rows; the others wait
          double res[3] = {0, 0, 0}, old[3], nf[3] = {0, 0, 0}, change = 0;
          if ((t & 3) == C) {
#pragma unroll
          for (int k = 1; k < 3; ++k) {
            double s = bb[k];
#pragma unroll
            for (int j = 0; j < 4; ++j) s += A[k][3 * j + 1] * force[3 * j + 1] + A[k][3 * j + 2] * force[3 * j + 2];
            res[k] = s;
          }
          old[0] = nf[0] = force[3 * C]; old[1] = force[3 * C + 1]; old[2] = force[3 * C + 2];
          if (old[0] < kMin) {
            nf[0] = nf[1] = nf[2] = 0;
          } else {
            const double A11 = A[1][3 * C + 1], A12 = A[1][3 * C + 2], A22 = A[2][3 * C + 2];
            const double b1 = res[1] - A11 * old[1] - A12 * old[2], b2 = res[2] - A[2][3 * C + 1] * old[1] - A22 * old[2];
            double vv[2];
            if (box_qcqp2(vv, A11, A12, A22, b1, b2, b.fr, old[0])) {
              double s = (vv[0] * vv[0] + vv[1] * vv[1]) / (b.fr * b.fr);
              s = sqrt(old[0] * old[0] * fast_rcp(s > kMin ? s : kMin));
              vv[0] *= s; vv[1] *= s;
            }
            nf[1] = vv[0]; nf[2] = vv[1];
          }
          // costChange(): a step that raises the dual cost is undone
          const double dl[3] = {0.0, nf[1] - old[1], nf[2] - old[2]};  // (the normal force is not touched)
#pragma unroll
          for (int k = 1; k < 3; ++k) {
#pragma unroll
            for (int l = 1; l < 3; ++l) change += 0.5 * dl[k] * A[k][3 * C + l] * dl[l];
            change += dl[k] * res[k];
          }
          if (change > 1e-10) { nf[0] = old[0]; nf[1] = old[1]; nf[2] = old[2]; change = 0; }
          }
          improvement -= quad_bcast<C>(change);
#pragma unroll
          for (int k = 0; k < 3; ++k) force[3 * C + k] = quad_bcast<C>(nf[k]);
        };
        sweep(std::integral_constant<int, 0>{});
        sweep(std::integral_constant<int, 1>{});
        sweep(std::integral_constant<int, 2>{});
        sweep(std::integral_constant<int, 3>{});
        improvement *= b.scale;
        ++iter;
        if (improvement < b.noslip_tolerance) break;
      }
      // dualFinish: qacc = qacc_smooth + M^-1 J' force
      const int sl = t & 3;
      double fo[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) fo[k] = sl == 0 ? force[k] : (sl == 1 ? force[3 + k] : (sl == 2 ? force[6 + k] : force[9 + k]));
#pragma unroll
      for (int j = 0; j < 6; ++j) x[j] = xs[j] + Mi[j] * quad_sum(c.J[0][j] * fo[0] + c.J[1][j] * fo[1] + c.J[2][j] * fo[2]);
    }
  }
  TEAM_MARK(19)
  // ---- integrate: velocity, position, quaternion (mju_quatIntegrate with the new angular velocity)
#pragma unroll
  for (int j = 0; j < 6; ++j) v[j] += h * x[j];
#pragma unroll
  for (int j = 0; j < 3; ++j) p[j] += h * v[j];
  {
    const double wn = sqrt(v[3] * v[3] + v[4] * v[4] + v[5] * v[5]);
    if (wn >= kMin) {
      const double ang = h * wn, s = sin(0.5 * ang) / wn, co = cos(0.5 * ang);
      const double r[4] = {co, v[3] * s, v[4] * s, v[5] * s};
      const double o[4] = {q[0] * r[0] - q[1] * r[1] - q[2] * r[2] - q[3] * r[3], q[0] * r[1] + q[1] * r[0] + q[2] * r[3] - q[3] * r[2],
                           q[0] * r[2] - q[1] * r[3] + q[2] * r[0] + q[3] * r[1], q[0] * r[3] + q[1] * r[2] - q[2] * r[1] + q[3] * r[0]};
      const double n = sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
      const double sc = fabs(n - 1) > kMin ? 1 / n : 1.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) q[k] = o[k] * sc;
    }
  }
  TEAM_MARK(20)
  if (t == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) bs[kBoxQ + k] = p[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) bs[kBoxQ + 3 + k] = q[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) { bs[kBoxV + k] = v[k]; bs[kBoxW + k] = x[k]; }
  }
}

#endif  // __HIP__

}  // namespace rcsh
