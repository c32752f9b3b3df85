// ik.h -- damped-least-squares closed-loop IK on the arm chain, one thread per environment.
//
// Same iteration as the reference's Pin::inverse (reference src/rcs/Kinematics.cpp:28-68; constants
// include/rcs/Kinematics.h:32-35): err = log6(frame^-1 * target), J <- -Jlog6 * J_local,
// v = -J^T (J J^T + 1e-6 I)^-1 err, q += 0.1 v, until |err| < 1e-4 or 1000 iterations.  The frame is the
// attachment site; its world placement and LOCAL 6 x n Jacobian come from the link tables (model.h), so the
// kinematic model is by construction the simulated one (reference quirk Q14 relies on the two agreeing).
#pragma once
#include "dyn.h"
#include "pose.h"

namespace rcsh {

constexpr double kIkEps = 1e-4;
constexpr int kIkMaxIter = 1000;
constexpr double kIkDt = 1e-1;
constexpr double kIkDamp = 1e-6;
constexpr double kTaylor = 1.220703125e-04;  // eps^(1/4): series below, closed form above

// rotation matrix (row-major) -> axis * angle; returns the angle
RCSH_HD double so3_log(const double* R, double* w) {
  const double pi = 3.141592653589793238462643383279502884;
  const double tr = R[0] + R[4] + R[8];
  double theta;
  if (tr >= 3.0) theta = 0;
  else if (tr <= -1.0) theta = pi;
  else theta = acos((tr - 1.0) / 2.0);
  if (theta >= pi - 1e-2) {
    const double cphi = -(tr - 1.0) / 2.0;
    const double beta = theta * theta / (1.0 + cphi);
    const double t0 = (R[0] + cphi) * beta, t1 = (R[4] + cphi) * beta, t2 = (R[8] + cphi) * beta;
    w[0] = (R[7] > R[5] ? 1.0 : -1.0) * (t0 > 0 ? sqrt(t0) : 0);
    w[1] = (R[2] > R[6] ? 1.0 : -1.0) * (t1 > 0 ? sqrt(t1) : 0);
    w[2] = (R[3] > R[1] ? 1.0 : -1.0) * (t2 > 0 ? sqrt(t2) : 0);
  } else {
    const double t = (theta > kTaylor ? theta / sin(theta) : 1.0) / 2.0;
    w[0] = t * (R[7] - R[5]);
    w[1] = t * (R[2] - R[6]);
    w[2] = t * (R[3] - R[1]);
  }
  return theta;
}

// SE(3) log: (R, p) -> [v; w]
RCSH_HD void se3_log(const double* R, const double* p, double* out) {
  double w[3];
  const double t = so3_log(R, w), t2 = t * t;
  double alpha, beta;
  if (t < kTaylor) {
    alpha = 1.0 - t2 / 12.0 - t2 * t2 / 720.0;
    beta = 1.0 / 12.0 + t2 / 720.0;
  } else {
    double st, ct;
    sincos(t, &st, &ct);
    alpha = t * st / (2.0 * (1.0 - ct));
    beta = 1.0 / t2 - st / (2.0 * t * (1.0 - ct));
  }
  double wxp[3];
  cross3(w, p, wxp);
  const double wp = dot3(w, p);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    out[i] = alpha * p[i] - 0.5 * wxp[i] + beta * wp * w[i];
    out[3 + i] = w[i];
  }
}

// Jacobian of the SE(3) log at (R, p): 6x6 row-major
RCSH_HD void se3_jlog(const double* R, const double* p, double* Jlog) {
  double w[3];
  const double t = so3_log(R, w), t2 = t * t;
  double A[9];
  {
    double alpha, diag;
    if (t < kTaylor) {
      alpha = 1.0 / 12.0 + t2 / 720.0;
      diag = 0.5 * (2.0 - t2 / 6.0);
    } else {
      double st, ct;
      sincos(t, &st, &ct);
      const double s1c = st / (1.0 - ct);
      alpha = 1.0 / t2 - s1c / (2.0 * t);
      diag = 0.5 * t * s1c;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) A[3 * r + c] = alpha * w[r] * w[c] + (r == c ? diag : 0.0);
    A[1] -= 0.5 * w[2]; A[2] += 0.5 * w[1];
    A[3] += 0.5 * w[2]; A[5] -= 0.5 * w[0];
    A[6] -= 0.5 * w[1]; A[7] += 0.5 * w[0];
  }
  double beta, bdot;
  if (t < kTaylor) {
    beta = 1.0 / 12.0 + t2 / 720.0;
    bdot = 1.0 / 360.0;
  } else {
    double st, ct;
    sincos(t, &st, &ct);
    const double tinv = 1.0 / t, t2inv = tinv * tinv, i22 = 1.0 / (2.0 * (1.0 - ct));
    beta = t2inv - st * tinv * i22;
    bdot = -2.0 * t2inv * t2inv + (1.0 + st * tinv) * t2inv * i22;
  }
  const double wp = dot3(w, p);
  double v3[3], Cm[9];
#pragma unroll
  for (int i = 0; i < 3; ++i) v3[i] = (bdot * wp) * w[i] - (t2 * bdot + 2.0 * beta) * p[i];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) Cm[3 * r + c] = v3[r] * w[c] + beta * w[r] * p[c] + (r == c ? wp * beta : 0.0);
  Cm[1] -= 0.5 * p[2]; Cm[2] += 0.5 * p[1];
  Cm[3] += 0.5 * p[2]; Cm[5] -= 0.5 * p[0];
  Cm[6] -= 0.5 * p[1]; Cm[7] += 0.5 * p[0];
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

// world placement of the attachment site for arm angles q, plus (optionally) world joint axes and anchors
template <class T>
RCSH_HD void site_fk(const DevModel& m, const double* q, double* Rs, double* ps, double (*ax)[3], double (*anchor)[3]) {
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {0, 0, 0};
#pragma unroll
  for (int i = 0; i < T::NARM; ++i) {
    double o[3], R0[9], a[3];
    mulmv(R, m.pos0[i], o);
    o[0] += p[0]; o[1] += p[1]; o[2] += p[2];
    mulmm(R, m.rot0[i], R0);
    mulmv(R0, m.axis[i], a);
    double s, c;
    fast_sincos(q[i] - m.qpos0[i], &s, &c);
    const double* u = m.axis[i];
    const double t = 1.0 - c;
    const double Q[9] = {c + t * u[0] * u[0],        t * u[0] * u[1] - s * u[2], t * u[0] * u[2] + s * u[1],
                         t * u[0] * u[1] + s * u[2], c + t * u[1] * u[1],        t * u[1] * u[2] - s * u[0],
                         t * u[0] * u[2] - s * u[1], t * u[1] * u[2] + s * u[0], c + t * u[2] * u[2]};
    double an[3], rj[3];
    mulmv(R0, m.jpos[i], an);
    an[0] += o[0]; an[1] += o[1]; an[2] += o[2];
    mulmm(R0, Q, R);
    mulmv(R, m.jpos[i], rj);
    p[0] = an[0] - rj[0]; p[1] = an[1] - rj[1]; p[2] = an[2] - rj[2];
    if (ax) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { ax[i][k] = a[k]; anchor[i][k] = an[k]; }
    }
    if (i == m.site_link) {
      mulmv(R, m.site_pos, ps);
      ps[0] += p[0]; ps[1] += p[1]; ps[2] += p[2];
      mulmm(R, m.site_rot, Rs);
    }
  }
}

// 6x6 SPD solve, LDL^T in place (A row-major full), x <- A^-1 x
RCSH_HD void ldl6_solve(double* A, double* x) {
  double L[36], D[6], Dinv[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double s = A[6 * j + j];
#pragma unroll
    for (int k = 0; k < j; ++k) s -= L[6 * j + k] * L[6 * j + k] * D[k];
    D[j] = s;
    const double inv = fast_rcp(s);
    Dinv[j] = inv;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double t = A[6 * i + j];
#pragma unroll
      for (int k = 0; k < j; ++k) t -= L[6 * i + k] * L[6 * j + k] * D[k];
      L[6 * i + j] = t * inv;
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int k = 0; k < i; ++k) x[i] -= L[6 * i + k] * x[k];
#pragma unroll
  for (int i = 0; i < 6; ++i) x[i] *= Dinv[i];
#pragma unroll
  for (int i = 5; i >= 0; --i)
#pragma unroll
    for (int k = i + 1; k < 6; ++k) x[i] -= L[6 * k + i] * x[k];
}

// Pin::inverse.  `target` is the desired TCP pose in ROBOT coordinates, `tcp` the TCP offset, q is in/out
// (arm angles; the model's remaining dofs are reported as 0 by the callers, reference quirk Q7).
template <class T>
RCSH_HD bool clik(const DevModel& m, const Pose& target, const Pose& tcp, double* q, int* iterations) {
  // desired site placement in world coordinates: base * (target * tcp^-1)
  Pose tinv, des_r, base, des;
  pose_inverse(tcp, tinv);
  pose_mul(target, tinv, des_r);
  const double bq[4] = {m.base_quat[1], m.base_quat[2], m.base_quat[3], m.base_quat[0]};
  pose_from_quat(bq, m.base_pos, base);
  pose_mul(base, des_r, des);
  double Rd[9];
  quat_to_mat(des.q, Rd);
  bool success = false;
  int it = 0;
  for (int i = 0;; ++i) {
    double Rs[9], ps[3], ax[T::NARM][3], an[T::NARM][3];
    site_fk<T>(m, q, Rs, ps, ax, an);
    // iMd = frame^-1 * desired
    double Ri[9], pi[3];
    const double dp[3] = {des.t[0] - ps[0], des.t[1] - ps[1], des.t[2] - ps[2]};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c) Ri[3 * r + c] = Rs[r] * Rd[c] + Rs[3 + r] * Rd[3 + c] + Rs[6 + r] * Rd[6 + c];
      pi[r] = Rs[r] * dp[0] + Rs[3 + r] * dp[1] + Rs[6 + r] * dp[2];
    }
    double err[6];
    se3_log(Ri, pi, err);
    it = i;
    if (sqrt(dot6(err, err)) < kIkEps) { success = true; break; }
    if (i >= kIkMaxIter) break;
    // Jlog6 at iMd^-1
    double Rt[9], pt[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) Rt[3 * r + c] = Ri[3 * c + r];
#pragma unroll
    for (int r = 0; r < 3; ++r) pt[r] = -(Rt[3 * r] * pi[0] + Rt[3 * r + 1] * pi[1] + Rt[3 * r + 2] * pi[2]);
    double Jlog[36];
    se3_jlog(Rt, pt, Jlog);
    // J <- -Jlog * J_local, column by column; accumulate J J^T
    double JJ[T::NARM][6], JJt[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) JJt[k] = 0.0;
#pragma unroll
    for (int j = 0; j < T::NARM; ++j) {
      const double rr[3] = {ps[0] - an[j][0], ps[1] - an[j][1], ps[2] - an[j][2]};
      double lin[3];
      cross3(ax[j], rr, lin);
      double col[6];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        col[k] = Rs[k] * lin[0] + Rs[3 + k] * lin[1] + Rs[6 + k] * lin[2];
        col[3 + k] = Rs[k] * ax[j][0] + Rs[3 + k] * ax[j][1] + Rs[6 + k] * ax[j][2];
      }
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) s += Jlog[6 * r + k] * col[k];
        JJ[j][r] = -s;
      }
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) JJt[6 * r + c] += JJ[j][r] * JJ[j][c];
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) JJt[6 * r + r] += kIkDamp;
    double y[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) y[k] = err[k];
    ldl6_solve(JJt, y);
#pragma unroll
    for (int j = 0; j < T::NARM; ++j) q[j] += -dot6(JJ[j], y) * kIkDt;
  }
  if (iterations) *iterations = it;
  return success;
}

}  // namespace rcsh
