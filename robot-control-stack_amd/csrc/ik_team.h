// ik_team.h -- the CLIK of ik.h (reference Pin::inverse, src/rcs/Kinematics.cpp:28-68) computed by a team of 16
// lanes per environment, lane t = arm joint t (team.h).
//
// Per iteration the forward kinematics of the arm is one 3-round scan of frame compositions, every lane builds
// its own Jacobian column, the columns are exchanged through the team's LDS block and every lane forms and solves
// the same 6x6 damped normal equations; lane t then moves joint t.  The SE(3) log / Jlog of the placement error is
// evaluated redundantly by all lanes (it is inherently serial and the instructions are issued for the wave
// anyway).  A wavefront iterates until its four environments are done, so one slow or failing target (1000
// iterations) holds up three neighbours, not sixty-three.
#pragma once
#include "dyn_team.h"
#include "ik.h"

namespace rcsh {

#if defined(__HIP__)

// LDS block of one team for the IK: the lanes' Jacobian columns
template <class T>
struct IkTeamBlock {
  double J[T::NARM][6];
  double A[22];  // the lower triangle of J J' + damping, an entry (or two) per lane
};

// desired site placement in world coordinates for a TCP target in robot coordinates: base * (target * tcp^-1)
RCSH_D void clik_desired(const DevModelHead& m, const Pose& target, const Pose& tcp, double* Rd, double* td) {
  Pose tinv, des_r, base, des;
  pose_inverse(tcp, tinv);
  pose_mul(target, tinv, des_r);
  const double bq[4] = {m.base_quat[1], m.base_quat[2], m.base_quat[3], m.base_quat[0]};
  pose_from_quat(bq, m.base_pos, base);
  pose_mul(base, des_r, des);
  quat_to_mat(des.q, Rd);
  td[0] = des.t[0]; td[1] = des.t[1]; td[2] = des.t[2];
}

// err = log6(R, p) and Jlog = Jlog6 of the INVERSE placement (R^T, -R^T p), as Pin::inverse needs them.  One
// so3 log serves both: log(R^T) = -log(R), same angle, so the angle's acos / sin / cos are evaluated once
// (ik.h's se3_log + se3_jlog evaluate them three times).
RCSH_D void se3_log_and_jlog_inverse(const double* R, const double* p, double* err, double* Jlog) {
  // so3 log as ik.h's so3_log, with the angle's sine taken from the one fast_sincos below instead of a libm call
  const double pi = 3.141592653589793238462643383279502884;
  const double tr = R[0] + R[4] + R[8];
  const double t = tr >= 3.0 ? 0.0 : (tr <= -1.0 ? pi : acos((tr - 1.0) / 2.0)), t2 = t * t;
  const bool small = t < kTaylor;
  double st = 0.0, ct = 1.0;
  if (!small) fast_sincos(t, &st, &ct);
  const double tinv = small ? 0.0 : fast_rcp(t), t2inv = tinv * tinv;
  double w[3];
  if (t >= pi - 1e-2) {
    const double cphi = -(tr - 1.0) / 2.0;
    const double beta = t2 / (1.0 + cphi);
    const double t0 = (R[0] + cphi) * beta, t1 = (R[4] + cphi) * beta, t2b = (R[8] + cphi) * beta;
    w[0] = (R[7] > R[5] ? 1.0 : -1.0) * (t0 > 0 ? sqrt(t0) : 0);
    w[1] = (R[2] > R[6] ? 1.0 : -1.0) * (t1 > 0 ? sqrt(t1) : 0);
    w[2] = (R[3] > R[1] ? 1.0 : -1.0) * (t2b > 0 ? sqrt(t2b) : 0);
  } else {
    const double k = 0.5 * (small ? 1.0 : t * fast_rcp(st));
    w[0] = k * (R[7] - R[5]);
    w[1] = k * (R[2] - R[6]);
    w[2] = k * (R[3] - R[1]);
  }
  const double i1c = small ? 0.0 : fast_rcp(1.0 - ct);  // 1 / (1 - cos t)
  // ---- log6
  {
    const double alpha = small ? 1.0 - t2 / 12.0 - t2 * t2 / 720.0 : 0.5 * t * st * i1c;
    const double beta = small ? 1.0 / 12.0 + t2 / 720.0 : t2inv - 0.5 * st * tinv * i1c;
    double wxp[3];
    cross3(w, p, wxp);
    const double wp = dot3(w, p);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      err[i] = alpha * p[i] - 0.5 * wxp[i] + beta * wp * w[i];
      err[3 + i] = w[i];
    }
  }
  // ---- Jlog6 at (R^T, -R^T p): rotation vector -w, same angle
  const double wi[3] = {-w[0], -w[1], -w[2]};
  double pt[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) pt[r] = -(R[r] * p[0] + R[3 + r] * p[1] + R[6 + r] * p[2]);
  double A[9];
  {
    const double s1c = st * i1c;
    const double alpha = small ? 1.0 / 12.0 + t2 / 720.0 : t2inv - 0.5 * s1c * tinv;
    const double diag = small ? 0.5 * (2.0 - t2 / 6.0) : 0.5 * t * s1c;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) A[3 * r + c] = alpha * wi[r] * wi[c] + (r == c ? diag : 0.0);
    A[1] -= 0.5 * wi[2]; A[2] += 0.5 * wi[1];
    A[3] += 0.5 * wi[2]; A[5] -= 0.5 * wi[0];
    A[6] -= 0.5 * wi[1]; A[7] += 0.5 * wi[0];
  }
  const double i22 = 0.5 * i1c;
  const double beta = small ? 1.0 / 12.0 + t2 / 720.0 : t2inv - st * tinv * i22;
  const double bdot = small ? 1.0 / 360.0 : -2.0 * t2inv * t2inv + (1.0 + st * tinv) * t2inv * i22;
  const double wp = dot3(wi, pt);
  double v3[3], Cm[9];
#pragma unroll
  for (int i = 0; i < 3; ++i) v3[i] = (bdot * wp) * wi[i] - (t2 * bdot + 2.0 * beta) * pt[i];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) Cm[3 * r + c] = v3[r] * wi[c] + beta * wi[r] * pt[c] + (r == c ? wp * beta : 0.0);
  Cm[1] -= 0.5 * pt[2]; Cm[2] += 0.5 * pt[1];
  Cm[3] += 0.5 * pt[2]; Cm[5] -= 0.5 * pt[0];
  Cm[6] -= 0.5 * pt[1]; Cm[7] += 0.5 * pt[0];
  double B[9];
  mulmm(Cm, A, B);
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      Jlog[6 * r + c] = A[3 * r + c];
      Jlog[6 * r + 3 + c] = B[3 * r + c];
      Jlog[6 * (3 + r) + c] = 0.0;
      Jlog[6 * (3 + r) + 3 + c] = A[3 * r + c];
    }
}

// Every lane of the wave calls this.  `run`: this team has a target (uniform within the team); Rd / td: desired
// world placement of the site (same values on all lanes of the team); q: lane t's joint angle, in / out.
// Returns success (uniform within the team); *iterations as Pin::inverse counts them.
template <class T>
RCSH_D bool clik_team(const DevModelHead& m, const LinkRec* links, IkTeamBlock<T>& blk, int t, bool run, const double* Rd,
                      const double* td, double& q, int* iterations) {
  constexpr int NA = T::NARM;
  const bool joint = t < NA;
  const int tl = joint ? t : NA - 1;
  KinK kk;
  kk.load(links[tl]);
  double site_rot[9], site_pos[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) site_rot[k] = m.site_rot[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) site_pos[k] = m.site_pos[k];
  const int site_lane = (threadIdx.x & ~(kTeamLanes - 1)) + m.site_link;
  // the lane's one or two entries of the normal equations' lower triangle (21 entries, row-major; 16 lanes)
  int er[2] = {0, 0}, ec[2] = {0, 0};
  static_assert(kTeamLanes == 16, "21 entries over 16 lanes: two trips");
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int k = t + kTeamLanes * u < 21 ? t + kTeamLanes * u : 0;
#pragma unroll
    for (int r = 0; r < 6; ++r)
      if (k >= r * (r + 1) / 2) { er[u] = r; ec[u] = k - r * (r + 1) / 2; }
  }
  bool active = run, success = false;
  int it = 0;
  for (int i = 0; __ballot(active) != 0; ++i) {
    // arm chain: local frames, 3 compose rounds (lanes 0..7 as a chain)
    double R[9], p[3];
    link_local_frame(kk, q, R, p);
    compose_round<1, 0xf>(R, p);
    compose_round<2, 0xf>(R, p);
    compose_round<4, 0xf>(R, p);
    double ax[3], an[3];
    mulmv(R, kk.axis, ax);
    mulmv(R, kk.jpos, an);
    an[0] += p[0]; an[1] += p[1]; an[2] += p[2];
    // site placement: computed from every lane's frame, taken from the lane of the site's link
    double Rs[9], ps[3];
    mulmm(R, site_rot, Rs);
    mulmv(R, site_pos, ps);
#pragma unroll
    for (int k = 0; k < 9; ++k) Rs[k] = lane_get(Rs[k], site_lane);
#pragma unroll
    for (int k = 0; k < 3; ++k) ps[k] = lane_get(ps[k] + p[k], site_lane);
    // iMd = frame^-1 * desired
    double Ri[9], pi[3];
    const double dp[3] = {td[0] - ps[0], td[1] - ps[1], td[2] - ps[2]};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c) Ri[3 * r + c] = Rs[r] * Rd[c] + Rs[3 + r] * Rd[3 + c] + Rs[6 + r] * Rd[6 + c];
      pi[r] = Rs[r] * dp[0] + Rs[3 + r] * dp[1] + Rs[6 + r] * dp[2];
    }
    double err[6], Jlog[36];
    se3_log_and_jlog_inverse(Ri, pi, err, Jlog);
    if (active) {
      it = i;
      if (sqrt(dot6(err, err)) < kIkEps) { success = true; active = false; }
      else if (i >= kIkMaxIter) active = false;
    }
    // the lane's column of J <- -Jlog * J_local
    double JJ[6];
    {
      const double rr[3] = {ps[0] - an[0], ps[1] - an[1], ps[2] - an[2]};
      double lin[3], col[6];
      cross3(ax, rr, lin);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        col[k] = Rs[k] * lin[0] + Rs[3 + k] * lin[1] + Rs[6 + k] * lin[2];
        col[3 + k] = Rs[k] * ax[0] + Rs[3 + k] * ax[1] + Rs[6 + k] * ax[2];
      }
      // (Jlog = [[A, B], [0, A]]: the zero block's nine products, which IEEE arithmetic would not let the compiler drop, are left out --
      // each added an exact zero)
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        double s = 0;
#pragma unroll
        for (int k = r < 3 ? 0 : 3; k < 6; ++k) s += Jlog[6 * r + k] * col[k];
        JJ[r] = -s;
      }
    }
    if (joint) {
#pragma unroll
      for (int r = 0; r < 6; ++r) blk.J[tl][r] = JJ[r];
    }
    team_sync();
    // J J' + damping: 21 sums of NARM products.  Every lane formed all of them through round 6's first half (147 multiply-adds and 42
    // LDS reads a lane and iteration, a seventh of the iteration's instructions); now a lane forms its one or two entries -- the same
    // products summed in the same order -- and the triangle goes round through LDS.
    {
      double e0 = 0.0, e1 = 0.0;
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        e0 += blk.J[j][er[0]] * blk.J[j][ec[0]];
        e1 += blk.J[j][er[1]] * blk.J[j][ec[1]];
      }
      if (er[0] == ec[0]) e0 += kIkDamp;
      if (er[1] == ec[1]) e1 += kIkDamp;
      blk.A[t] = e0;
      if (t + kTeamLanes < 21) blk.A[t + kTeamLanes] = e1;
    }
    team_sync();
    double JJt[36];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c <= r; ++c) JJt[6 * r + c] = blk.A[r * (r + 1) / 2 + c];
    double y[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) y[k] = err[k];
    ldl6_solve(JJt, y);  // reads the lower triangle only
    if (active && joint) q += -dot6(JJ, y) * kIkDt;
  }
  if (iterations) *iterations = it;
  return success;
}

#endif  // __HIP__

}  // namespace rcsh
