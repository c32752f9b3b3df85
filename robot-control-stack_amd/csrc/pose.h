// pose.h -- SE(3) helpers for the observation / action path, one thread per environment.
//
// Value semantics follow the reference's immutable Pose (reference include/rcs/Pose.h,
// src/rcs/Pose.cpp): quaternion + translation, quaternion coefficient order x y z w
// (Pose.cpp:115), every constructor re-normalises, RPY = extrinsic xyz read back with the
// yaw-in-[0,pi] branch the reference inherits from its Euler-angle extraction
// (Pose.cpp:133-138).
#pragma once
#include "dyn.h"

namespace rcsh {

struct Pose {
  double t[3];
  double q[4];  // x y z w
};

RCSH_HD void quat_normalize(double* q) {
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
RCSH_HD void quat_mul(const double* a, const double* b, double* r) {
  const double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  const double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  const double y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  const double z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z; r[3] = w;
}
RCSH_HD void quat_rotate(const double* q, const double* v, double* r) {
  double uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  const double c0 = q[1] * uv[2] - q[2] * uv[1], c1 = q[2] * uv[0] - q[0] * uv[2], c2 = q[0] * uv[1] - q[1] * uv[0];
  r[0] = v[0] + q[3] * uv[0] + c0;
  r[1] = v[1] + q[3] * uv[1] + c1;
  r[2] = v[2] + q[3] * uv[2] + c2;
}
RCSH_HD void quat_to_mat(const double* q, double* m) {
  const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
  const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  m[0] = 1 - (tyy + tzz); m[1] = txy - twz; m[2] = txz + twy;
  m[3] = txy + twz; m[4] = 1 - (txx + tzz); m[5] = tyz - twx;
  m[6] = txz - twy; m[7] = tyz + twx; m[8] = 1 - (txx + tyy);
}
// rotation matrix (row-major) -> quaternion; trace branch, else the largest-diagonal branch
RCSH_HD void mat_to_quat(const double* m, double* q) {
  double t = m[0] + m[4] + m[8];
  if (t > 0) {
    t = sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (m[7] - m[5]) * t;
    q[1] = (m[2] - m[6]) * t;
    q[2] = (m[3] - m[1]) * t;
  } else {
    // i = argmax of the diagonal with the comparison order (1 vs 0, then 2 vs winner)
    const bool i1 = m[4] > m[0];
    const double dwin = i1 ? m[4] : m[0];
    const bool i2 = m[8] > dwin;
    if (i2) {
      t = sqrt(m[8] - m[0] - m[4] + 1.0);
      q[2] = 0.5 * t; t = 0.5 / t;
      q[3] = (m[3] - m[1]) * t; q[0] = (m[2] + m[6]) * t; q[1] = (m[5] + m[7]) * t;
    } else if (i1) {
      t = sqrt(m[4] - m[8] - m[0] + 1.0);
      q[1] = 0.5 * t; t = 0.5 / t;
      q[3] = (m[2] - m[6]) * t; q[2] = (m[7] + m[5]) * t; q[0] = (m[1] + m[3]) * t;
    } else {
      t = sqrt(m[0] - m[4] - m[8] + 1.0);
      q[0] = 0.5 * t; t = 0.5 / t;
      q[3] = (m[7] - m[5]) * t; q[1] = (m[3] + m[1]) * t; q[2] = (m[6] + m[2]) * t;
    }
  }
}
RCSH_HD void pose_from_mat(const double* R, const double* t, Pose& out) {
  out.t[0] = t[0]; out.t[1] = t[1]; out.t[2] = t[2];
  mat_to_quat(R, out.q);
  quat_normalize(out.q);
}
RCSH_HD void pose_from_quat(const double* q, const double* t, Pose& out) {
  out.t[0] = t[0]; out.t[1] = t[1]; out.t[2] = t[2];
  out.q[0] = q[0]; out.q[1] = q[1]; out.q[2] = q[2]; out.q[3] = q[3];
  quat_normalize(out.q);
}
// roll-pitch-yaw -> quaternion: qz(yaw) * qy(pitch) * qx(roll)
RCSH_HD void pose_from_rpy(const double* rpy, const double* t, Pose& out) {
  double sz, cz, sy, cy, sx, cx;
  sincos(0.5 * rpy[2], &sz, &cz);
  sincos(0.5 * rpy[1], &sy, &cy);
  sincos(0.5 * rpy[0], &sx, &cx);
  const double qz[4] = {0, 0, sz, cz}, qy[4] = {0, sy, 0, cy}, qx[4] = {sx, 0, 0, cx};
  double tmp[4];
  quat_mul(qz, qy, tmp);
  quat_mul(tmp, qx, out.q);
  quat_normalize(out.q);
  out.t[0] = t[0]; out.t[1] = t[1]; out.t[2] = t[2];
}
RCSH_HD void pose_mul(const Pose& a, const Pose& b, Pose& out) {
  Pose r;
  quat_rotate(a.q, b.t, r.t);
  r.t[0] += a.t[0]; r.t[1] += a.t[1]; r.t[2] += a.t[2];
  quat_mul(a.q, b.q, r.q);
  quat_normalize(r.q);
  out = r;
}
RCSH_HD void pose_inverse(const Pose& a, Pose& out) {
  Pose r;
  r.q[0] = -a.q[0]; r.q[1] = -a.q[1]; r.q[2] = -a.q[2]; r.q[3] = a.q[3];
  quat_rotate(r.q, a.t, r.t);
  r.t[0] = -r.t[0]; r.t[1] = -r.t[1]; r.t[2] = -r.t[2];
  quat_normalize(r.q);
  out = r;
}
RCSH_HD void pose_rpy(const Pose& p, double* rpy) {
  double m[9];
  quat_to_mat(p.q, m);
  double yaw = atan2(m[3], m[0]);
  const double c2 = sqrt(m[8] * m[8] + m[7] * m[7]);
  double pitch;
  if (yaw < 0) {
    yaw += 3.141592653589793238462643383279502884;
    pitch = atan2(-m[6], -c2);
  } else {
    pitch = atan2(-m[6], c2);
  }
  double s1, c1;
  sincos(yaw, &s1, &c1);
  rpy[0] = atan2(s1 * m[2] - c1 * m[5], c1 * m[4] - s1 * m[1]);
  rpy[1] = pitch;
  rpy[2] = yaw;
}
RCSH_HD double quat_angular_distance(const double* a, const double* b) {
  const double bc[4] = {-b[0], -b[1], -b[2], b[3]};
  double d[4];
  quat_mul(a, bc, d);
  return 2 * atan2(sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), fabs(d[3]));
}
RCSH_HD void quat_slerp(const double* a, double t, const double* b, double* r) {
  const double one = 1.0 - 2.220446049250313e-16;
  const double d = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
  const double absd = fabs(d);
  double s0, s1;
  if (absd >= one) {
    s0 = 1.0 - t; s1 = t;
  } else {
    const double theta = acos(absd), st = sin(theta);
    s0 = sin((1.0 - t) * theta) / st;
    s1 = sin(t * theta) / st;
  }
  if (d < 0) s1 = -s1;
  for (int i = 0; i < 4; ++i) r[i] = s0 * a[i] + s1 * b[i];
}
RCSH_HD void pose_limit_rotation_angle(const Pose& a, double max_angle, Pose& out) {
  const double id[4] = {0, 0, 0, 1};
  const double cur = quat_angular_distance(a.q, id);
  Pose r = a;
  if (cur > max_angle && max_angle >= 0) {
    quat_slerp(id, max_angle / cur, a.q, r.q);
    quat_normalize(r.q);
  }
  out = r;
}
RCSH_HD void pose_limit_translation_length(const Pose& a, double max_length, Pose& out) {
  Pose r = a;
  const double n = sqrt(a.t[0] * a.t[0] + a.t[1] * a.t[1] + a.t[2] * a.t[2]);
  if (n > max_length && max_length >= 0) {
    for (int i = 0; i < 3; ++i) r.t[i] = a.t[i] / n * max_length;
    quat_normalize(r.q);
  }
  out = r;
}

}  // namespace rcsh
