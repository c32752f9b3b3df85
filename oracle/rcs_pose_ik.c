/*
 * rcs_pose_ik.c -- TEST INFRASTRUCTURE (see rcs_oracle.h).
 *
 * Pose / RPY restated from reference src/rcs/Pose.cpp + include/rcs/Pose.h, with the
 * Eigen 3.4 operations those files call written out (Eigen is a third-party
 * header library absent from /root/reference; its Quaternion / eulerAngles /
 * slerp algorithms are restated from the published sources).
 * Pin CLIK restated from reference src/rcs/Kinematics.cpp:28-82, with the
 * pinocchio 3.7.0 primitives it calls (log6, Jlog6, LOCAL frame Jacobian)
 * written out from their published formulas -- parity unpinned beyond the
 * reference's own tolerance tests.
 *
 * Quaternion storage is Eigen coeffs() order: x, y, z, w.
 */
#include <math.h>
#include <string.h>

#include "rcs_oracle.h"

#define QX 0
#define QY 1
#define QZ 2
#define QW 3

/* ---------------------------------------------------------------- Eigen pieces */
static void eq_normalize(double* q) { /* Quaternion::normalize(): coeffs /= norm */
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
static void eq_mul(double* r, const double* a, const double* b) { /* Hamilton product */
  double w = a[QW] * b[QW] - a[QX] * b[QX] - a[QY] * b[QY] - a[QZ] * b[QZ];
  double x = a[QW] * b[QX] + a[QX] * b[QW] + a[QY] * b[QZ] - a[QZ] * b[QY];
  double y = a[QW] * b[QY] + a[QY] * b[QW] + a[QZ] * b[QX] - a[QX] * b[QZ];
  double z = a[QW] * b[QZ] + a[QZ] * b[QW] + a[QX] * b[QY] - a[QY] * b[QX];
  r[QX] = x; r[QY] = y; r[QZ] = z; r[QW] = w;
}
static void eq_rotate(double* r, const double* q, const double* v) { /* _transformVector */
  double uv[3] = {q[QY] * v[2] - q[QZ] * v[1], q[QZ] * v[0] - q[QX] * v[2], q[QX] * v[1] - q[QY] * v[0]};
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  double c[3] = {q[QY] * uv[2] - q[QZ] * uv[1], q[QZ] * uv[0] - q[QX] * uv[2], q[QX] * uv[1] - q[QY] * uv[0]};
  r[0] = v[0] + q[QW] * uv[0] + c[0];
  r[1] = v[1] + q[QW] * uv[1] + c[1];
  r[2] = v[2] + q[QW] * uv[2] + c[2];
}
static void eq_to_matrix(double* m, const double* q) { /* toRotationMatrix, row-major out */
  double tx = 2 * q[QX], ty = 2 * q[QY], tz = 2 * q[QZ];
  double twx = tx * q[QW], twy = ty * q[QW], twz = tz * q[QW];
  double txx = tx * q[QX], txy = ty * q[QX], txz = tz * q[QX];
  double tyy = ty * q[QY], tyz = tz * q[QY], tzz = tz * q[QZ];
  m[0] = 1 - (tyy + tzz); m[1] = txy - twz; m[2] = txz + twy;
  m[3] = txy + twz; m[4] = 1 - (txx + tzz); m[5] = tyz - twx;
  m[6] = txz - twy; m[7] = tyz + twx; m[8] = 1 - (txx + tyy);
}
static void eq_from_matrix(double* q, const double* m) { /* Quaternion(Matrix3), row-major in */
  double t = m[0] + m[4] + m[8];
  if (t > 0) {
    t = sqrt(t + 1.0);
    q[QW] = 0.5 * t;
    t = 0.5 / t;
    q[QX] = (m[7] - m[5]) * t;
    q[QY] = (m[2] - m[6]) * t;
    q[QZ] = (m[3] - m[1]) * t;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[4 * i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[QW] = (m[3 * k + j] - m[3 * j + k]) * t;
    q[j] = (m[3 * j + i] + m[3 * i + j]) * t;
    q[k] = (m[3 * k + i] + m[3 * i + k]) * t;
  }
}
static void eq_slerp(double* r, const double* a, double t, const double* b) {
  const double one = 1.0 - 2.220446049250313e-16;
  double d = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
  double absd = fabs(d), s0, s1;
  if (absd >= one) {
    s0 = 1.0 - t; s1 = t;
  } else {
    double theta = acos(absd), st = sin(theta);
    s0 = sin((1.0 - t) * theta) / st;
    s1 = sin(t * theta) / st;
  }
  if (d < 0) s1 = -s1;
  for (int i = 0; i < 4; i++) r[i] = s0 * a[i] + s1 * b[i];
}
static double eq_angular_distance(const double* a, const double* b) {
  double bc[4] = {-b[QX], -b[QY], -b[QZ], b[QW]}, d[4];
  eq_mul(d, a, bc);
  return 2 * atan2(sqrt(d[QX] * d[QX] + d[QY] * d[QY] + d[QZ] * d[QZ]), fabs(d[QW]));
}
/* RPY::as_quaternion (Pose.h:43-48): AngleAxis(yaw,Z) * AngleAxis(pitch,Y) * AngleAxis(roll,X) */
static void rpy_to_quat(double* q, const double* rpy) {
  double qz[4] = {0, 0, sin(0.5 * rpy[2]), cos(0.5 * rpy[2])};
  double qy[4] = {0, sin(0.5 * rpy[1]), 0, cos(0.5 * rpy[1])};
  double qx[4] = {sin(0.5 * rpy[0]), 0, 0, cos(0.5 * rpy[0])};
  double t[4];
  eq_mul(t, qz, qy);
  eq_mul(q, t, qx);
}
/* rotation part of an affine matrix as Eigen::Transform<Affine>::rotation() defines it: the orthogonal
   polar factor (Eigen computes it via SVD; here by Newton iteration, same fixed point) */
static void polar_rotation(double* r, const double* m) {
  double a[9];
  memcpy(a, m, sizeof(a));
  for (int it = 0; it < 60; it++) {
    double det = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
    double inv_t[9]; /* inverse transpose = cofactor / det */
    inv_t[0] = (a[4] * a[8] - a[5] * a[7]) / det; inv_t[1] = (a[5] * a[6] - a[3] * a[8]) / det; inv_t[2] = (a[3] * a[7] - a[4] * a[6]) / det;
    inv_t[3] = (a[2] * a[7] - a[1] * a[8]) / det; inv_t[4] = (a[0] * a[8] - a[2] * a[6]) / det; inv_t[5] = (a[1] * a[6] - a[0] * a[7]) / det;
    inv_t[6] = (a[1] * a[5] - a[2] * a[4]) / det; inv_t[7] = (a[2] * a[3] - a[0] * a[5]) / det; inv_t[8] = (a[0] * a[4] - a[1] * a[3]) / det;
    double diff = 0;
    for (int i = 0; i < 9; i++) {
      double n = 0.5 * (a[i] + inv_t[i]);
      diff += fabs(n - a[i]);
      a[i] = n;
    }
    if (diff == 0) break;
    if (diff < 1e-16) break;
  }
  memcpy(r, a, sizeof(a));
}

/* ------------------------------------------------------------------- Pose API */
void orc_pose_identity(orc_pose* p) { /* Pose.cpp:19-22 */
  p->t[0] = p->t[1] = p->t[2] = 0;
  p->q[QX] = p->q[QY] = p->q[QZ] = 0; p->q[QW] = 1;
}
void orc_pose_from_matrix4(const double* m, orc_pose* out) { /* Pose.cpp:33-38 */
  double lin[9] = {m[0], m[1], m[2], m[4], m[5], m[6], m[8], m[9], m[10]}, r[9];
  out->t[0] = m[3]; out->t[1] = m[7]; out->t[2] = m[11];
  polar_rotation(r, lin);
  eq_from_matrix(out->q, r);
  eq_normalize(out->q);
}
void orc_pose_from_rotm_t(const double* r, const double* t, orc_pose* out) { /* Pose.cpp:40-45 */
  memcpy(out->t, t, sizeof(out->t));
  eq_from_matrix(out->q, r);
  eq_normalize(out->q);
}
void orc_pose_from_quat_t(const double* q, const double* t, orc_pose* out) { /* Pose.cpp:47-52 */
  memcpy(out->t, t, sizeof(out->t));
  memcpy(out->q, q, sizeof(out->q));
  eq_normalize(out->q);
}
void orc_pose_from_rpy_t(const double* rpy, const double* t, orc_pose* out) { /* Pose.cpp:61-73 */
  memcpy(out->t, t, sizeof(out->t));
  rpy_to_quat(out->q, rpy);
  eq_normalize(out->q);
}
void orc_pose_rotation_m(const orc_pose* p, double* r) { eq_to_matrix(r, p->q); } /* Pose.cpp:111-113 */
void orc_pose_matrix(const orc_pose* p, double* m) { /* Pose.cpp:119-127 */
  double r[9];
  eq_to_matrix(r, p->q);
  m[0] = r[0]; m[1] = r[1]; m[2] = r[2]; m[3] = p->t[0];
  m[4] = r[3]; m[5] = r[4]; m[6] = r[5]; m[7] = p->t[1];
  m[8] = r[6]; m[9] = r[7]; m[10] = r[8]; m[11] = p->t[2];
  m[12] = m[13] = m[14] = 0; m[15] = 1;
}
void orc_pose_rpy(const orc_pose* p, double* rpy) { /* Pose.cpp:133-138: eulerAngles(2,1,0) */
  double m[9];
  eq_to_matrix(m, p->q);
  double r0 = atan2(m[3], m[0]); /* yaw */
  double c2 = sqrt(m[8] * m[8] + m[7] * m[7]);
  double r1;
  if (r0 < 0) {
    r0 += M_PI;
    r1 = atan2(-m[6], -c2);
  } else {
    r1 = atan2(-m[6], c2);
  }
  double s1 = sin(r0), c1 = cos(r0);
  double r2 = atan2(s1 * m[2] - c1 * m[5], c1 * m[4] - s1 * m[1]);
  rpy[0] = r2; rpy[1] = r1; rpy[2] = r0;
}
void orc_pose_xyzrpy(const orc_pose* p, double* out) { /* Pose.cpp:156-161 */
  memcpy(out, p->t, 3 * sizeof(double));
  orc_pose_rpy(p, out + 3);
}
void orc_pose_mul(const orc_pose* a, const orc_pose* b, orc_pose* out) { /* Pose.cpp:173-178 */
  orc_pose r;
  eq_rotate(r.t, a->q, b->t);
  r.t[0] += a->t[0]; r.t[1] += a->t[1]; r.t[2] += a->t[2];
  eq_mul(r.q, a->q, b->q);
  eq_normalize(r.q);
  *out = r;
}
void orc_pose_inverse(const orc_pose* a, orc_pose* out) { /* Pose.cpp:203-206 */
  orc_pose r;
  r.q[QX] = -a->q[QX]; r.q[QY] = -a->q[QY]; r.q[QZ] = -a->q[QZ]; r.q[QW] = a->q[QW];
  eq_rotate(r.t, r.q, a->t);
  r.t[0] = -r.t[0]; r.t[1] = -r.t[1]; r.t[2] = -r.t[2];
  eq_normalize(r.q);
  *out = r;
}
double orc_pose_total_angle(const orc_pose* a) { /* Pose.cpp:180-182 */
  const double id[4] = {0, 0, 0, 1};
  return eq_angular_distance(a->q, id);
}
void orc_pose_limit_rotation_angle(const orc_pose* a, double max_angle, orc_pose* out) { /* Pose.cpp:184-192 */
  double cur = orc_pose_total_angle(a);
  orc_pose r = *a;
  if (cur > max_angle && max_angle >= 0) {
    const double id[4] = {0, 0, 0, 1};
    eq_slerp(r.q, id, max_angle / cur, a->q);
    eq_normalize(r.q);
  }
  *out = r;
}
void orc_pose_limit_translation_length(const orc_pose* a, double max_length, orc_pose* out) { /* Pose.cpp:193-201 */
  orc_pose r = *a;
  double n = sqrt(a->t[0] * a->t[0] + a->t[1] * a->t[1] + a->t[2] * a->t[2]);
  if (n > max_length && max_length >= 0) {
    for (int i = 0; i < 3; i++) r.t[i] = a->t[i] / n * max_length;
    eq_normalize(r.q);
  }
  *out = r;
}
void orc_pose_interpolate(const orc_pose* a, const orc_pose* dest, double progress, orc_pose* out) { /* Pose.cpp:140-154 */
  orc_pose r;
  if (progress > 1) progress = 1;
  for (int i = 0; i < 3; i++) r.t[i] = a->t[i] + (dest->t[i] - a->t[i]) * progress;
  eq_slerp(r.q, a->q, progress, dest->q);
  eq_normalize(r.q);
  *out = r;
}
int orc_pose_is_close(const orc_pose* a, const orc_pose* b, double eps_r, double eps_t) { /* Pose.cpp:208-211 */
  double l1 = fabs(a->t[0] - b->t[0]) + fabs(a->t[1] - b->t[1]) + fabs(a->t[2] - b->t[2]);
  return l1 < eps_t && eq_angular_distance(a->q, b->q) < eps_r;
}
void orc_franka_hand_tcp_offset(orc_pose* out) { /* Pose.cpp:11-15 fed to Pose(Matrix4d) */
  const double m[16] = {0.707, 0.707, 0, 0, -0.707, 0.707, 0, 0, 0, 0, 1, 0.1034, 0, 0, 0, 1};
  orc_pose_from_matrix4(m, out);
}

/* ---------------------------------------------------------------------- CLIK */

/* world placement (R row-major, p) of the IK frame and its LOCAL 6 x nv Jacobian [linear; angular] */
/* pinocchio places the ROOT body of the parsed tree at the universe (urdf / mjcf visitors' addRootJoint: identity
   placement), so a root body that the file offsets from its world -- xArm7's `base` sits 0.12 m up -- has that offset
   dropped: the IK frame is the root body's frame, the frame get_cartesian_position reports in.  (FR3: root body at
   the origin, no-op.)  A Jacobian expressed in the site frame does not change. */
static void ik_to_root_frame(const orc_model* m, const orc_data* d, int site, double* R, double* p) {
  int root = m->site_bodyid[site];
  while (root > 0 && m->body_parentid[root] > 0) root = m->body_parentid[root];
  if (root <= 0) return;
  const double* Rb = d->xmat[root];
  const double* pb = d->xpos[root];
  double Rn[9], dp[3] = {p[0] - pb[0], p[1] - pb[1], p[2] - pb[2]}, pn[3];
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) Rn[3 * r + c] = Rb[r] * R[c] + Rb[3 + r] * R[3 + c] + Rb[6 + r] * R[6 + c];
    pn[r] = Rb[r] * dp[0] + Rb[3 + r] * dp[1] + Rb[6 + r] * dp[2];
  }
  memcpy(R, Rn, sizeof(Rn));
  memcpy(p, pn, sizeof(pn));
}

static void ik_fk(const orc_ik* ik, const double* q, double* R, double* p, double J[6][ORC_MAXV]) {
  const orc_model* m = ik->m;
  orc_data d;
  memset(&d, 0, sizeof(d));
  memcpy(d.qpos, q, sizeof(double) * m->njnt);
  orc_kinematics(m, &d);
  memcpy(R, d.site_xmat[ik->site], 9 * sizeof(double));
  memcpy(p, d.site_xpos[ik->site], 3 * sizeof(double));
  if (!J) { ik_to_root_frame(m, &d, ik->site, R, p); return; }
  for (int r = 0; r < 6; r++) memset(J[r], 0, sizeof(J[r]));
  int b = m->site_bodyid[ik->site];
  while (b > 0) {
    int j = m->body_jntadr[b];
    if (j >= 0) {
      double lin[3], ang[3] = {0, 0, 0};
      const double* ax = d.xaxis[j];
      if (m->jnt_type[j] == ORC_JNT_SLIDE) {
        memcpy(lin, ax, sizeof(lin));
      } else {
        double r[3] = {p[0] - d.xanchor[j][0], p[1] - d.xanchor[j][1], p[2] - d.xanchor[j][2]};
        lin[0] = ax[1] * r[2] - ax[2] * r[1];
        lin[1] = ax[2] * r[0] - ax[0] * r[2];
        lin[2] = ax[0] * r[1] - ax[1] * r[0];
        memcpy(ang, ax, sizeof(ang));
      }
      /* express in the frame: R^T * v */
      for (int k = 0; k < 3; k++) {
        J[k][j] = R[k] * lin[0] + R[3 + k] * lin[1] + R[6 + k] * lin[2];
        J[3 + k][j] = R[k] * ang[0] + R[3 + k] * ang[1] + R[6 + k] * ang[2];
      }
    }
    b = m->body_parentid[b];
  }
  ik_to_root_frame(m, &d, ik->site, R, p);
}

#define TAYLOR_PREC 1.220703125e-04 /* eps^(1/4), pinocchio's TaylorSeriesExpansion precision<3> */
/* pinocchio log3: rotation matrix -> axis*angle, returns theta */
static double pin_log3(const double* R, double* w) {
  const double PI_value = M_PI;
  double tr = R[0] + R[4] + R[8];
  double theta;
  if (tr >= 3.0) theta = 0;
  else if (tr <= -1.0) theta = PI_value;
  else theta = acos((tr - 1.0) / 2.0);
  if (theta >= PI_value - 1e-2) {
    /* near pi: explicit formula from the diagonal */
    double cphi = -(tr - 1.0) / 2.0;
    double beta = theta * theta / (1.0 + cphi);
    double t0 = (R[0] + cphi) * beta, t1 = (R[4] + cphi) * beta, t2 = (R[8] + cphi) * beta;
    w[0] = (R[7] > R[5] ? 1.0 : -1.0) * (t0 > 0 ? sqrt(t0) : 0);
    w[1] = (R[2] > R[6] ? 1.0 : -1.0) * (t1 > 0 ? sqrt(t1) : 0);
    w[2] = (R[3] > R[1] ? 1.0 : -1.0) * (t2 > 0 ? sqrt(t2) : 0);
  } else {
    double t = (theta > TAYLOR_PREC ? theta / sin(theta) : 1.0) / 2.0;
    w[0] = t * (R[7] - R[5]);
    w[1] = t * (R[2] - R[6]);
    w[2] = t * (R[3] - R[1]);
  }
  return theta;
}
/* pinocchio log6: SE3 (R,p) -> [v; w] */
static void pin_log6(const double* R, const double* p, double* out) {
  double w[3];
  double t = pin_log3(R, w), t2 = t * t;
  double alpha, beta;
  if (t < TAYLOR_PREC) {
    alpha = 1.0 - t2 / 12.0 - t2 * t2 / 720.0;
    beta = 1.0 / 12.0 + t2 / 720.0;
  } else {
    double st = sin(t), ct = cos(t);
    alpha = t * st / (2.0 * (1.0 - ct));
    beta = 1.0 / t2 - st / (2.0 * t * (1.0 - ct));
  }
  double wxp[3] = {w[1] * p[2] - w[2] * p[1], w[2] * p[0] - w[0] * p[2], w[0] * p[1] - w[1] * p[0]};
  double wp = w[0] * p[0] + w[1] * p[1] + w[2] * p[2];
  for (int i = 0; i < 3; i++) {
    out[i] = alpha * p[i] - 0.5 * wxp[i] + beta * wp * w[i];
    out[3 + i] = w[i];
  }
}
/* pinocchio Jlog3 / Jlog6 */
static void pin_jlog3(double theta, const double* w, double A[3][3]) {
  double alpha, diag;
  if (theta < TAYLOR_PREC) {
    alpha = 1.0 / 12.0 + theta * theta / 720.0;
    diag = 0.5 * (2.0 - theta * theta / 6.0);
  } else {
    double st = sin(theta), ct = cos(theta), s1c = st / (1.0 - ct);
    alpha = 1.0 / (theta * theta) - s1c / (2.0 * theta);
    diag = 0.5 * theta * s1c;
  }
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) A[r][c] = alpha * w[r] * w[c];
  for (int r = 0; r < 3; r++) A[r][r] += diag;
  /* addSkew(0.5 * w) */
  A[0][1] -= 0.5 * w[2]; A[0][2] += 0.5 * w[1];
  A[1][0] += 0.5 * w[2]; A[1][2] -= 0.5 * w[0];
  A[2][0] -= 0.5 * w[1]; A[2][1] += 0.5 * w[0];
}
static void pin_jlog6(const double* R, const double* p, double Jlog[6][6]) {
  double w[3], A[3][3];
  double t = pin_log3(R, w), t2 = t * t;
  pin_jlog3(t, w, A);
  double beta, bdot;
  if (t < TAYLOR_PREC) {
    beta = 1.0 / 12.0 + t2 / 720.0;
    bdot = 1.0 / 360.0;
  } else {
    double tinv = 1.0 / t, t2inv = tinv * tinv, st = sin(t), ct = cos(t), i22 = 1.0 / (2.0 * (1.0 - ct));
    beta = t2inv - st * tinv * i22;
    bdot = -2.0 * t2inv * t2inv + (1.0 + st * tinv) * t2inv * i22;
  }
  double wp = w[0] * p[0] + w[1] * p[1] + w[2] * p[2];
  double v3[3], C[3][3];
  for (int i = 0; i < 3; i++) v3[i] = (bdot * wp) * w[i] - (t2 * bdot + 2.0 * beta) * p[i];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) C[r][c] = v3[r] * w[c] + beta * w[r] * p[c];
  for (int r = 0; r < 3; r++) C[r][r] += wp * beta;
  /* addSkew(0.5 * p) */
  C[0][1] -= 0.5 * p[2]; C[0][2] += 0.5 * p[1];
  C[1][0] += 0.5 * p[2]; C[1][2] -= 0.5 * p[0];
  C[2][0] -= 0.5 * p[1]; C[2][1] += 0.5 * p[0];
  memset(Jlog, 0, sizeof(double) * 36);
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      double b = 0;
      for (int k = 0; k < 3; k++) b += C[r][k] * A[k][c];
      Jlog[r][c] = A[r][c];
      Jlog[r][3 + c] = b;
      Jlog[3 + r][3 + c] = A[r][c];
    }
}
/* 6x6 LDLT solve (Eigen ldlt() pivots; the damped JJ^T is SPD, plain LDL^T reaches the same solution) */
static void ldlt6_solve(double A[6][6], double* x) {
  double L[6][6] = {{0}}, D[6];
  for (int j = 0; j < 6; j++) {
    double s = A[j][j];
    for (int k = 0; k < j; k++) s -= L[j][k] * L[j][k] * D[k];
    D[j] = s;
    L[j][j] = 1;
    for (int i = j + 1; i < 6; i++) {
      double t = A[i][j];
      for (int k = 0; k < j; k++) t -= L[i][k] * L[j][k] * D[k];
      L[i][j] = t / s;
    }
  }
  for (int i = 0; i < 6; i++) for (int k = 0; k < i; k++) x[i] -= L[i][k] * x[k];
  for (int i = 0; i < 6; i++) x[i] /= D[i];
  for (int i = 5; i >= 0; i--) for (int k = i + 1; k < 6; k++) x[i] -= L[k][i] * x[k];
}

/* Pin::inverse, Kinematics.cpp:28-68.  Returns 1 on success (q_out has model.nq entries). */
int orc_ik_inverse(const orc_ik* ik, const orc_pose* pose, const double* q0, int nq0, const orc_pose* tcp_offset,
                   double* q_out, int* iterations) {
  const double eps = 1e-4, DT = 1e-1, damp = 1e-6; /* Kinematics.h:32-35 */
  const int IT_MAX = 1000;
  const orc_model* m = ik->m;
  int nv = m->njnt;
  orc_pose inv_off, des;
  orc_pose_inverse(tcp_offset, &inv_off);
  orc_pose_mul(pose, &inv_off, &des);
  double Rd[9];
  eq_to_matrix(Rd, des.q);
  double q[ORC_MAXV];
  memset(q, 0, sizeof(q));
  for (int i = 0; i < nq0 && i < nv; i++) q[i] = q0[i];
  int success = 0, it = 0;
  for (int i = 0;; i++) {
    double R[9], p[3], J[6][ORC_MAXV];
    ik_fk(ik, q, R, p, J);
    /* iMd = oMf^-1 * oMdes */
    double Ri[9], pi[3], dp[3] = {des.t[0] - p[0], des.t[1] - p[1], des.t[2] - p[2]};
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) Ri[3 * r + c] = R[r] * Rd[c] + R[3 + r] * Rd[3 + c] + R[6 + r] * Rd[6 + c];
      pi[r] = R[r] * dp[0] + R[3 + r] * dp[1] + R[6 + r] * dp[2];
    }
    double err[6];
    pin_log6(Ri, pi, err);
    double en = 0;
    for (int k = 0; k < 6; k++) en += err[k] * err[k];
    it = i;
    if (sqrt(en) < eps) { success = 1; break; }
    if (i >= IT_MAX) { success = 0; break; }
    /* Jlog6(iMd.inverse()) */
    double Rt[9], pt[3];
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) Rt[3 * r + c] = Ri[3 * c + r];
    }
    for (int r = 0; r < 3; r++) pt[r] = -(Rt[3 * r] * pi[0] + Rt[3 * r + 1] * pi[1] + Rt[3 * r + 2] * pi[2]);
    double Jlog[6][6], JJ[6][ORC_MAXV];
    pin_jlog6(Rt, pt, Jlog);
    for (int r = 0; r < 6; r++)
      for (int c = 0; c < nv; c++) {
        double s = 0;
        for (int k = 0; k < 6; k++) s += Jlog[r][k] * J[k][c];
        JJ[r][c] = -s;
      }
    double JJt[6][6], y[6];
    for (int r = 0; r < 6; r++)
      for (int c = 0; c < 6; c++) {
        double s = 0;
        for (int k = 0; k < nv; k++) s += JJ[r][k] * JJ[c][k];
        JJt[r][c] = s;
      }
    for (int r = 0; r < 6; r++) JJt[r][r] += damp;
    memcpy(y, err, sizeof(y));
    ldlt6_solve(JJt, y);
    for (int c = 0; c < nv; c++) {
      double v = 0;
      for (int r = 0; r < 6; r++) v += JJ[r][c] * y[r];
      q[c] += (-v) * DT;
    }
  }
  if (iterations) *iterations = it;
  if (success) memcpy(q_out, q, sizeof(double) * nv);
  return success;
}

/* Pin::forward, Kinematics.cpp:70-81 (returns pose * tcp_offset.inverse(), quirk Q7) */
void orc_ik_forward(const orc_ik* ik, const double* q0, int nq0, const orc_pose* tcp_offset, orc_pose* out) {
  double q[ORC_MAXV], R[9], p[3];
  memset(q, 0, sizeof(q));
  for (int i = 0; i < nq0 && i < ik->m->njnt; i++) q[i] = q0[i];
  ik_fk(ik, q, R, p, 0);
  orc_pose f, inv_off;
  orc_pose_from_rotm_t(R, p, &f);
  orc_pose_inverse(tcp_offset, &inv_off);
  orc_pose_mul(&f, &inv_off, out);
}
