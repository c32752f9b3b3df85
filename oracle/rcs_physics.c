/*
 * rcs_physics.c -- TEST INFRASTRUCTURE (see rcs_oracle.h).
 *
 * Restatement of what mj_step1 / mj_step2 / mj_resetData compute for the RCS
 * scenes (call sites: reference src/sim/sim.cpp:110,112,118).  The algorithm is
 * MuJoCo 3.2.6's published forward-dynamics pipeline; MuJoCo's source is not in
 * /root/reference, so every stage below names the MuJoCo routine it restates
 * ("mj_kinematics", "mj_crb", ...) instead of a reference file:line, and the
 * whole file is PARITY UNPINNED except through the reference's own test pins.
 *
 * Scope: kinematic trees of hinge / slide joints, fixed tendons, joint
 * equalities, joint limits, affine actuators, gravity compensation,
 * implicitfast.  Contacts: rcs_contact.c (robot geoms), rcs_object.c (the free box on the floor).
 */
#include <math.h>
#include <string.h>

#include "rcs_oracle.h"

#define MINVAL 1e-15
#define MINIMP 0.0001
#define MAXIMP 0.9999

/* ------------------------------------------------------------------ small math */
static void v3_copy(double* r, const double* a) { r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; }
static void v3_zero(double* r) { r[0] = r[1] = r[2] = 0; }
static void v3_add(double* r, const double* a, const double* b) { r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; }
static void v3_sub(double* r, const double* a, const double* b) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; }
static void v3_addscl(double* r, const double* a, double s) { r[0] += a[0] * s; r[1] += a[1] * s; r[2] += a[2] * s; }
static double v3_dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void v3_cross(double* r, const double* a, const double* b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static void m3_mulvec(double* r, const double* m, const double* v) {
  double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  double y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  double z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
/* wxyz quaternions (MuJoCo convention) */
static void q_mul(double* r, const double* a, const double* b) {
  double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
static void q_normalize(double* q) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
static void q_to_mat(double* m, const double* q) {
  double q00 = q[0] * q[0], q11 = q[1] * q[1], q22 = q[2] * q[2], q33 = q[3] * q[3];
  double q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
  double q12 = q[1] * q[2], q13 = q[1] * q[3], q23 = q[2] * q[3];
  m[0] = q00 + q11 - q22 - q33; m[4] = q00 - q11 + q22 - q33; m[8] = q00 - q11 - q22 + q33;
  m[1] = 2 * (q12 - q03); m[2] = 2 * (q13 + q02);
  m[3] = 2 * (q12 + q03); m[5] = 2 * (q23 - q01);
  m[6] = 2 * (q13 - q02); m[7] = 2 * (q23 + q01);
}
static void q_rotvec(double* r, const double* v, const double* q) {
  double m[9];
  q_to_mat(m, q);
  m3_mulvec(r, m, v);
}
static void q_axis_angle(double* q, const double* axis, double angle) {
  double s = sin(angle * 0.5);
  q[0] = cos(angle * 0.5); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}

/* spatial algebra: vectors are [angular(3); linear(3)] like MuJoCo's c-frame quantities */
static void cross_motion(double* r, const double* vel, const double* v) {
  double a[3], b[3];
  v3_cross(r, vel, v);
  v3_cross(a, vel, v + 3);
  v3_cross(b, vel + 3, v);
  v3_add(r + 3, a, b);
}
static void cross_force(double* r, const double* vel, const double* f) {
  double a[3], b[3];
  v3_cross(a, vel, f);
  v3_cross(b, vel + 3, f + 3);
  v3_add(r, a, b);
  v3_cross(r + 3, vel, f + 3);
}
/* 10-number inertia (xx yy zz xy xz yz, m*d(3), m) times motion vector */
static void mul_inert_vec(double* r, const double* i, const double* v) {
  r[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  r[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  r[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  r[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  r[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  r[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}

/* dense SPD solve (n <= ORC_MAXV) via Cholesky; returns 0 on success */
static int chol_factor(double L[ORC_MAXV][ORC_MAXV], int n) {
  for (int j = 0; j < n; j++) {
    double s = L[j][j];
    for (int k = 0; k < j; k++) s -= L[j][k] * L[j][k];
    if (s <= 0) return 1;
    L[j][j] = sqrt(s);
    for (int i = j + 1; i < n; i++) {
      double t = L[i][j];
      for (int k = 0; k < j; k++) t -= L[i][k] * L[j][k];
      L[i][j] = t / L[j][j];
    }
  }
  return 0;
}
static void chol_solve(const double L[ORC_MAXV][ORC_MAXV], int n, double* x) {
  for (int i = 0; i < n; i++) {
    double t = x[i];
    for (int k = 0; k < i; k++) t -= L[i][k] * x[k];
    x[i] = t / L[i][i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double t = x[i];
    for (int k = i + 1; k < n; k++) t -= L[k][i] * x[k];
    x[i] = t / L[i][i];
  }
}

/* ------------------------------------------------------------- position stage */

/* mj_kinematics: body frames, joint anchors/axes, inertial frames, site frames */
void orc_kinematics(const orc_model* m, orc_data* d) {
  v3_zero(d->xpos[0]);
  d->xquat[0][0] = 1; d->xquat[0][1] = d->xquat[0][2] = d->xquat[0][3] = 0;
  q_to_mat(d->xmat[0], d->xquat[0]);
  v3_zero(d->xipos[0]);
  q_to_mat(d->ximat[0], d->xquat[0]);
  for (int i = 1; i < m->nbody; i++) {
    int p = m->body_parentid[i];
    double xpos[3], xquat[4], vec[3];
    /* frame at qpos0 relative to parent */
    m3_mulvec(vec, d->xmat[p], m->body_pos[i]);
    v3_add(xpos, vec, d->xpos[p]);
    q_mul(xquat, d->xquat[p], m->body_quat[i]);
    int j = m->body_jntadr[i];
    if (j >= 0) {
      double q = d->qpos[j] - m->qpos0[j];
      q_rotvec(d->xaxis[j], m->jnt_axis[j], xquat);
      q_rotvec(vec, m->jnt_pos[j], xquat);
      v3_add(d->xanchor[j], vec, xpos);
      if (m->jnt_type[j] == ORC_JNT_SLIDE) {
        v3_addscl(xpos, d->xaxis[j], q);
      } else { /* hinge: rotate about the anchor */
        double qloc[4], tmp[4];
        q_axis_angle(qloc, m->jnt_axis[j], q);
        q_mul(tmp, xquat, qloc);
        memcpy(xquat, tmp, sizeof(tmp));
        q_rotvec(vec, m->jnt_pos[j], xquat);
        v3_sub(xpos, d->xanchor[j], vec);
      }
    }
    q_normalize(xquat);
    v3_copy(d->xpos[i], xpos);
    memcpy(d->xquat[i], xquat, sizeof(xquat));
    q_to_mat(d->xmat[i], xquat);
    /* inertial frame */
    double iq[4];
    m3_mulvec(vec, d->xmat[i], m->body_ipos[i]);
    v3_add(d->xipos[i], vec, xpos);
    q_mul(iq, xquat, m->body_iquat[i]);
    q_to_mat(d->ximat[i], iq);
  }
  for (int s = 0; s < m->nsite; s++) {
    int b = m->site_bodyid[s];
    double vec[3], sq[4];
    m3_mulvec(vec, d->xmat[b], m->site_pos[s]);
    v3_add(d->site_xpos[s], vec, d->xpos[b]);
    q_mul(sq, d->xquat[b], m->site_quat[s]);
    q_to_mat(d->site_xmat[s], sq);
  }
}

/* mj_comPos: subtree COMs, body inertias and dof axes about the tree root's COM */
static void com_pos(const orc_model* m, orc_data* d) {
  double submass[ORC_MAXBODY];
  for (int i = 0; i < m->nbody; i++) {
    submass[i] = m->body_mass[i];
    for (int k = 0; k < 3; k++) d->subtree_com[i][k] = d->xipos[i][k] * m->body_mass[i];
  }
  for (int i = m->nbody - 1; i > 0; i--) {
    int p = m->body_parentid[i];
    submass[p] += submass[i];
    for (int k = 0; k < 3; k++) d->subtree_com[p][k] += d->subtree_com[i][k];
  }
  for (int i = 0; i < m->nbody; i++) {
    if (submass[i] < MINVAL) v3_copy(d->subtree_com[i], d->xipos[i]);
    else for (int k = 0; k < 3; k++) d->subtree_com[i][k] /= submass[i];
  }
  memset(d->cinert[0], 0, sizeof(d->cinert[0]));
  for (int i = 1; i < m->nbody; i++) {
    double off[3], tmp[9];
    const double* R = d->ximat[i];
    const double* I = m->body_inertia[i];
    double mass = m->body_mass[i];
    v3_sub(off, d->xipos[i], d->subtree_com[m->body_rootid[i]]);
    /* R diag(I) R^T */
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) tmp[3 * r + c] = R[3 * r] * I[0] * R[3 * c] + R[3 * r + 1] * I[1] * R[3 * c + 1] + R[3 * r + 2] * I[2] * R[3 * c + 2];
    double* ci = d->cinert[i];
    ci[0] = tmp[0] + mass * (off[1] * off[1] + off[2] * off[2]);
    ci[1] = tmp[4] + mass * (off[0] * off[0] + off[2] * off[2]);
    ci[2] = tmp[8] + mass * (off[0] * off[0] + off[1] * off[1]);
    ci[3] = tmp[1] - mass * off[0] * off[1];
    ci[4] = tmp[2] - mass * off[0] * off[2];
    ci[5] = tmp[5] - mass * off[1] * off[2];
    ci[6] = mass * off[0]; ci[7] = mass * off[1]; ci[8] = mass * off[2];
    ci[9] = mass;
  }
  for (int j = 0; j < m->njnt; j++) {
    double off[3];
    v3_sub(off, d->subtree_com[m->body_rootid[m->jnt_bodyid[j]]], d->xanchor[j]);
    if (m->jnt_type[j] == ORC_JNT_SLIDE) {
      v3_zero(d->cdof[j]);
      v3_copy(d->cdof[j] + 3, d->xaxis[j]);
    } else {
      v3_copy(d->cdof[j], d->xaxis[j]);
      v3_cross(d->cdof[j] + 3, d->xaxis[j], off);
    }
  }
}

/* mj_crb: composite rigid body mass matrix (dense), + armature */
static void crb(const orc_model* m, orc_data* d) {
  double crbI[ORC_MAXBODY][10];
  memcpy(crbI, d->cinert, sizeof(crbI));
  for (int i = m->nbody - 1; i > 0; i--) {
    int p = m->body_parentid[i];
    if (p > 0) for (int k = 0; k < 10; k++) crbI[p][k] += crbI[i][k];
  }
  memset(d->qM, 0, sizeof(d->qM));
  for (int i = 0; i < m->njnt; i++) {
    double buf[6];
    mul_inert_vec(buf, crbI[m->jnt_bodyid[i]], d->cdof[i]);
    /* walk up the dof tree: ancestors of dof i are the joints of ancestor bodies */
    int b = m->jnt_bodyid[i];
    int j = i;
    while (j >= 0) {
      double v = 0;
      for (int k = 0; k < 6; k++) v += d->cdof[j][k] * buf[k];
      d->qM[i][j] = d->qM[j][i] = v;
      /* parent dof */
      b = m->body_parentid[m->jnt_bodyid[j]];
      while (b > 0 && m->body_jntadr[b] < 0) b = m->body_parentid[b];
      j = b > 0 ? m->body_jntadr[b] : -1;
    }
    d->qM[i][i] += m->dof_armature[i];
  }
}

void orc_mass_matrix(const orc_model* m, orc_data* d) {
  orc_kinematics(m, d);
  com_pos(m, d);
  crb(m, d);
}

/* mj_tendon (fixed) + mj_transmission */
static void tendon_and_transmission(const orc_model* m, orc_data* d) {
  for (int t = 0; t < m->ntendon; t++) {
    double len = 0;
    for (int w = 0; w < m->tendon_num[t]; w++) {
      int k = m->tendon_adr[t] + w;
      len += m->wrap_prm[k] * d->qpos[m->wrap_objid[k]];
    }
    d->ten_length[t] = len;
  }
  for (int u = 0; u < m->nu; u++) {
    double g = m->actuator_gear[u];
    if (m->actuator_trntype[u] == ORC_TRN_JOINT) d->actuator_length[u] = g * d->qpos[m->actuator_trnid[u]];
    else d->actuator_length[u] = g * d->ten_length[m->actuator_trnid[u]];
  }
}

/* actuator_moment row u (dense over dofs) */
static void actuator_moment(const orc_model* m, int u, double* row) {
  memset(row, 0, sizeof(double) * ORC_MAXV);
  double g = m->actuator_gear[u];
  if (m->actuator_trntype[u] == ORC_TRN_JOINT) {
    row[m->actuator_trnid[u]] = g;
  } else {
    int t = m->actuator_trnid[u];
    for (int w = 0; w < m->tendon_num[t]; w++) {
      int k = m->tendon_adr[t] + w;
      row[m->wrap_objid[k]] += g * m->wrap_prm[k];
    }
  }
}

/* solimp -> impedance at |pos - margin| (MuJoCo getimpedance) */
double orc_impedance(const double* solimp_in, double pos, double margin);
static double impedance(const double* solimp_in, double pos, double margin) { return orc_impedance(solimp_in, pos, margin); }
double orc_impedance(const double* solimp_in, double pos, double margin) {
  double s[5];
  memcpy(s, solimp_in, sizeof(s));
  /* mj_assignImp range clamps */
  if (s[0] < MINIMP) s[0] = MINIMP; if (s[0] > MAXIMP) s[0] = MAXIMP;
  if (s[1] < MINIMP) s[1] = MINIMP; if (s[1] > MAXIMP) s[1] = MAXIMP;
  if (s[2] < 0) s[2] = 0;
  if (s[3] < MINIMP) s[3] = MINIMP; if (s[3] > MAXIMP) s[3] = MAXIMP;
  if (s[4] < 1) s[4] = 1;
  if (s[0] == s[1] || s[2] <= MINVAL) return 0.5 * (s[0] + s[1]);
  double x = (pos - margin) / s[2];
  if (x < 0) x = -x;
  if (x >= 1) return s[1];
  if (x <= 0) return s[0];
  double y;
  if (s[4] == 1) y = x;
  else if (x <= s[3]) y = pow(x, s[4]) / pow(s[3], s[4] - 1);
  else y = 1 - pow(1 - x, s[4]) / pow(1 - s[3], s[4] - 1);
  return s[0] + y * (s[1] - s[0]);
}

/* mj_makeConstraint (equality + limit rows) and mj_makeImpedance for position-dependent parts */
static void make_constraint(const orc_model* m, orc_data* d) {
  double *K = d->efc_K, *B = d->efc_B, *I = d->efc_I;
  int n = 0;
  for (int e = 0; e < m->neq; e++) {
    if (!m->eq_active0[e]) continue;
    int j1 = m->eq_obj1id[e], j2 = m->eq_obj2id[e];
    const double* c = m->eq_data[e];
    memset(d->efc_J[n], 0, sizeof(d->efc_J[n]));
    double pos, diag;
    if (j2 >= 0) {
      double dif = d->qpos[j2] - m->qpos0[j2];
      double poly = c[0] + c[1] * dif + c[2] * dif * dif + c[3] * dif * dif * dif + c[4] * dif * dif * dif * dif;
      double deriv = c[1] + 2 * c[2] * dif + 3 * c[3] * dif * dif + 4 * c[4] * dif * dif * dif;
      pos = d->qpos[j1] - m->qpos0[j1] - poly;
      d->efc_J[n][j1] = 1;
      d->efc_J[n][j2] = -deriv;
      diag = m->dof_invweight0[j1] + m->dof_invweight0[j2];
    } else {
      pos = d->qpos[j1] - m->qpos0[j1] - c[0];
      d->efc_J[n][j1] = 1;
      diag = m->dof_invweight0[j1];
    }
    d->efc_type[n] = ORC_EFC_EQUALITY;
    d->efc_pos[n] = pos;
    d->efc_margin[n] = 0;
    /* impedance, stiffness, damping */
    double imp = impedance(m->eq_solimp[e], pos, 0);
    double dmax = m->eq_solimp[e][1];
    if (dmax < MINIMP) dmax = MINIMP; if (dmax > MAXIMP) dmax = MAXIMP;
    double tc = m->eq_solref[e][0], dr = m->eq_solref[e][1];
    if (tc > 0) { /* refsafe */
      if (tc < 2 * m->timestep) tc = 2 * m->timestep;
      double kd = dmax * dmax * tc * tc * dr * dr, bd = dmax * tc;
      K[n] = 1 / (kd > MINVAL ? kd : MINVAL);
      B[n] = 2 / (bd > MINVAL ? bd : MINVAL);
    } else {
      K[n] = -tc / (dmax * dmax);
      B[n] = -dr / dmax;
    }
    I[n] = imp;
    double R = (1 - imp) / imp * diag;
    if (R < MINVAL) R = MINVAL;
    d->efc_D[n] = 1 / R;
    n++;
  }
  /* mj_instantiateFriction: one row per dof with frictionloss > 0; J = e_j, pos = 0.  mj_makeImpedance sets the
     stiffness of friction rows to zero, so aref = -B * velocity; the impedance is solimp's value at distance 0 */
  for (int j = 0; j < m->njnt; j++) {
    if (m->dof_frictionloss[j] <= 0) continue;
    memset(d->efc_J[n], 0, sizeof(d->efc_J[n]));
    d->efc_J[n][j] = 1;
    d->efc_type[n] = ORC_EFC_FRICTION;
    d->efc_pos[n] = 0;
    d->efc_margin[n] = 0;
    d->efc_frictionloss[n] = m->dof_frictionloss[j];
    double imp = impedance(m->dof_solimp[j], 0, 0);
    double dmax = m->dof_solimp[j][1];
    if (dmax < MINIMP) dmax = MINIMP; if (dmax > MAXIMP) dmax = MAXIMP;
    double tc = m->dof_solref[j][0], dr = m->dof_solref[j][1];
    if (tc > 0) {
      if (tc < 2 * m->timestep) tc = 2 * m->timestep;
      double bd = dmax * tc;
      B[n] = 2 / (bd > MINVAL ? bd : MINVAL);
    } else {
      B[n] = -dr / dmax;
    }
    K[n] = 0;
    I[n] = imp;
    double R = (1 - imp) / imp * m->dof_invweight0[j];
    if (R < MINVAL) R = MINVAL;
    d->efc_D[n] = 1 / R;
    n++;
  }
  for (int j = 0; j < m->njnt; j++) {
    if (!m->jnt_limited[j]) continue;
    for (int side = 0; side < 2; side++) {
      double dist = side == 0 ? d->qpos[j] - m->jnt_range[j][0] : m->jnt_range[j][1] - d->qpos[j];
      if (dist >= m->jnt_margin[j]) continue;
      memset(d->efc_J[n], 0, sizeof(d->efc_J[n]));
      d->efc_J[n][j] = side == 0 ? 1 : -1;
      d->efc_type[n] = ORC_EFC_LIMIT;
      d->efc_pos[n] = dist;
      d->efc_margin[n] = m->jnt_margin[j];
      double imp = impedance(m->jnt_solimp[j], dist, m->jnt_margin[j]);
      double dmax = m->jnt_solimp[j][1];
      if (dmax < MINIMP) dmax = MINIMP; if (dmax > MAXIMP) dmax = MAXIMP;
      double tc = m->jnt_solref[j][0], dr = m->jnt_solref[j][1];
      if (tc > 0) {
        if (tc < 2 * m->timestep) tc = 2 * m->timestep;
        double kd = dmax * dmax * tc * tc * dr * dr, bd = dmax * tc;
        K[n] = 1 / (kd > MINVAL ? kd : MINVAL);
        B[n] = 2 / (bd > MINVAL ? bd : MINVAL);
      } else {
        K[n] = -tc / (dmax * dmax);
        B[n] = -dr / dmax;
      }
      I[n] = imp;
      double R = (1 - imp) / imp * m->dof_invweight0[j];
      if (R < MINVAL) R = MINVAL;
      d->efc_D[n] = 1 / R;
      n++;
    }
  }
  d->nefc = n;
}

/* ------------------------------------------------------------- velocity stage */

/* mj_comVel */
static void com_vel(const orc_model* m, orc_data* d) {
  memset(d->cvel[0], 0, sizeof(d->cvel[0]));
  for (int i = 1; i < m->nbody; i++) {
    double cvel[6];
    memcpy(cvel, d->cvel[m->body_parentid[i]], sizeof(cvel));
    int j = m->body_jntadr[i];
    if (j >= 0) {
      cross_motion(d->cdof_dot[j], cvel, d->cdof[j]);
      for (int k = 0; k < 6; k++) cvel[k] += d->cdof[j][k] * d->qvel[j];
    }
    memcpy(d->cvel[i], cvel, sizeof(cvel));
  }
}

/* mj_rne with flg_acc = 0: Coriolis, centrifugal and gravity */
static void rne_bias(const orc_model* m, orc_data* d) {
  double cacc[ORC_MAXBODY][6], cfrc[ORC_MAXBODY][6];
  memset(cacc, 0, sizeof(cacc));
  memset(cfrc, 0, sizeof(cfrc));
  for (int k = 0; k < 3; k++) cacc[0][3 + k] = -m->gravity[k];
  for (int i = 1; i < m->nbody; i++) {
    memcpy(cacc[i], cacc[m->body_parentid[i]], sizeof(cacc[i]));
    int j = m->body_jntadr[i];
    if (j >= 0) for (int k = 0; k < 6; k++) cacc[i][k] += d->cdof_dot[j][k] * d->qvel[j];
    double t1[6], t2[6];
    mul_inert_vec(t1, d->cinert[i], cacc[i]);
    mul_inert_vec(t2, d->cinert[i], d->cvel[i]);
    cross_force(cfrc[i], d->cvel[i], t2);
    for (int k = 0; k < 6; k++) cfrc[i][k] += t1[k];
  }
  for (int i = m->nbody - 1; i > 0; i--) {
    int p = m->body_parentid[i];
    if (p > 0) for (int k = 0; k < 6; k++) cfrc[p][k] += cfrc[i][k];
  }
  for (int j = 0; j < m->njnt; j++) {
    double v = 0;
    for (int k = 0; k < 6; k++) v += d->cdof[j][k] * cfrc[m->jnt_bodyid[j]][k];
    d->qfrc_bias[j] = v;
  }
}

/* mj_passive: joint damping + gravity compensation (split by jnt_actgravcomp) */
static void passive(const orc_model* m, orc_data* d) {
  for (int j = 0; j < m->njnt; j++) {
    d->qfrc_passive[j] = -m->dof_damping[j] * d->qvel[j];
    d->qfrc_gravcomp[j] = 0;
  }
  for (int i = 1; i < m->nbody; i++) {
    if (m->body_gravcomp[i] == 0) continue;
    double f[3];
    for (int k = 0; k < 3; k++) f[k] = -m->gravity[k] * m->body_mass[i] * m->body_gravcomp[i];
    /* mj_applyFT(force at xipos): J^T f over the ancestors' dofs */
    int b = i;
    while (b > 0) {
      int j = m->body_jntadr[b];
      if (j >= 0) {
        double col[3];
        if (m->jnt_type[j] == ORC_JNT_SLIDE) v3_copy(col, d->xaxis[j]);
        else {
          double r[3];
          v3_sub(r, d->xipos[i], d->xanchor[j]);
          v3_cross(col, d->xaxis[j], r);
        }
        d->qfrc_gravcomp[j] += v3_dot(col, f);
      }
      b = m->body_parentid[b];
    }
  }
  for (int j = 0; j < m->njnt; j++)
    if (!m->jnt_actgravcomp[j]) d->qfrc_passive[j] += d->qfrc_gravcomp[j];
}

void orc_step1(const orc_model* m, orc_data* d) {
  /* mj_fwdPosition */
  orc_kinematics(m, d);
  com_pos(m, d);
  tendon_and_transmission(m, d);
  crb(m, d);
  if (m->box.present) orc_box_step1(&m->box, &d->box, m->timestep); /* box frame, floor contacts, its rows */
  orc_collide(m, d); /* mj_collision: d->contact; d->coupled when a robot geom touches something and contacts are resolved */
  make_constraint(m, d);
  if (d->coupled) orc_make_coupled_rows(m, d);
  /* mj_fwdVelocity */
  for (int u = 0; u < m->nu; u++) {
    double row[ORC_MAXV], v = 0;
    actuator_moment(m, u, row);
    for (int j = 0; j < m->njnt; j++) v += row[j] * d->qvel[j];
    d->actuator_velocity[u] = v;
  }
  com_vel(m, d);
  passive(m, d);
  rne_bias(m, d);
  /* mj_referenceConstraint */
  for (int i = 0; i < d->nefc; i++) {
    double v = 0;
    for (int j = 0; j < m->njnt; j++) v += d->efc_J[i][j] * d->qvel[j];
    if (m->box.present) for (int k = 0; k < 6; k++) v += d->efc_J[i][m->njnt + k] * d->box.qvel[k];
    d->efc_vel[i] = v;
    d->efc_aref[i] = -d->efc_K[i] * d->efc_I[i] * (d->efc_pos[i] - d->efc_margin[i]) - d->efc_B[i] * v;
  }
}

/* --------------------------------------------------- acceleration / constraint */

static double clip(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* mj_fwdActuation: affine actuators, force limits, actuator-side gravity compensation, joint-level clamp */
static void fwd_actuation(const orc_model* m, orc_data* d) {
  memset(d->qfrc_actuator, 0, sizeof(d->qfrc_actuator));
  for (int u = 0; u < m->nu; u++) {
    double ctrl = d->ctrl[u];
    if (m->actuator_ctrllimited[u]) ctrl = clip(ctrl, m->actuator_ctrlrange[u][0], m->actuator_ctrlrange[u][1]);
    double gain = m->actuator_gainprm[u][0];
    double bias = 0;
    if (m->actuator_biastype[u])
      bias = m->actuator_biasprm[u][0] + m->actuator_biasprm[u][1] * d->actuator_length[u] + m->actuator_biasprm[u][2] * d->actuator_velocity[u];
    double f = gain * ctrl + bias;
    if (m->actuator_forcelimited[u]) f = clip(f, m->actuator_forcerange[u][0], m->actuator_forcerange[u][1]);
    d->actuator_force[u] = f;
    double row[ORC_MAXV];
    actuator_moment(m, u, row);
    for (int j = 0; j < m->njnt; j++) d->qfrc_actuator[j] += row[j] * f;
  }
  for (int j = 0; j < m->njnt; j++)
    if (m->jnt_actgravcomp[j]) d->qfrc_actuator[j] += d->qfrc_gravcomp[j];
  for (int j = 0; j < m->njnt; j++)
    if (m->jnt_actfrclimited[j]) d->qfrc_actuator[j] = clip(d->qfrc_actuator[j], m->jnt_actfrcrange[j][0], m->jnt_actfrcrange[j][1]);
}

/* Primal constraint solve: qacc = argmin 1/2 |qacc - qacc_smooth|_M^2 + sum_i s_i(J_i qacc - aref_i)
 * with s_i quadratic (equality), one-sided quadratic (limit) or Huber (dry joint friction: quadratic inside
 * |J qacc - aref| < R * frictionloss, linear outside).  Newton on the piecewise-quadratic
 * cost: a full step that leaves the active set unchanged lands on the exact minimiser; otherwise an
 * exact line search along the Newton direction is taken and the step repeated.  (MuJoCo's Newton
 * solver stops at tolerance 1e-8 of the same problem.) */
/* Row states (mj_constraintUpdate): 0 off (satisfied limit), 1 quadratic, 2 linear-negative (friction row pushing
 * with +frictionloss), 3 linear-positive (-frictionloss).  `dir` breaks ties on a zone boundary during the line search
 * (which side the iterate is about to enter); 0 applies MuJoCo's own comparisons. */
enum { ST_OFF = 0, ST_QUAD = 1, ST_LINNEG = 2, ST_LINPOS = 3 };
static int row_state(const orc_data* d, int i, double jar, double dir) {
  if (d->efc_type[i] == ORC_EFC_EQUALITY) return ST_QUAD;
  if (d->efc_type[i] == ORC_EFC_LIMIT) return (jar < 0 || (jar == 0 && dir < 0)) ? ST_QUAD : ST_OFF;
  double rf = d->efc_frictionloss[i] / d->efc_D[i];
  if (jar < -rf || (jar == -rf && dir <= 0)) return ST_LINNEG;
  if (jar > rf || (jar == rf && dir >= 0)) return ST_LINPOS;
  return ST_QUAD;
}
static double row_force(const orc_data* d, int i, double jar, int state) {
  if (state == ST_QUAD) return -d->efc_D[i] * jar;
  if (state == ST_LINNEG) return d->efc_frictionloss[i];
  if (state == ST_LINPOS) return -d->efc_frictionloss[i];
  return 0;
}
static void efc_residual(const orc_model* m, const orc_data* d, const double* x, double* jar, int* state) {
  for (int i = 0; i < d->nefc; i++) {
    double v = -d->efc_aref[i];
    for (int j = 0; j < m->njnt; j++) v += d->efc_J[i][j] * x[j];
    jar[i] = v;
    state[i] = row_state(d, i, v, 0);
  }
}

static void solve_constraints(const orc_model* m, orc_data* d) {
  int nv = m->njnt, ne = d->nefc;
  double L[ORC_MAXV][ORC_MAXV];
  memcpy(L, d->qM, sizeof(L));
  chol_factor(L, nv);
  memcpy(d->qacc_smooth, d->qfrc_smooth, sizeof(double) * nv);
  chol_solve(L, nv, d->qacc_smooth);
  memcpy(d->qacc, d->qacc_smooth, sizeof(double) * nv);
  memset(d->qfrc_constraint, 0, sizeof(d->qfrc_constraint));
  d->solver_niter = 0;
  if (ne == 0) return;
  double jar[ORC_MAXEFC];
  int state[ORC_MAXEFC];
  for (int iter = 0; iter < 50; iter++) {
    efc_residual(m, d, d->qacc, jar, state);
    /* Newton step of the cost with the rows frozen in their current zones: quadratic rows enter the Hessian,
       linear rows push with a constant force */
    double H[ORC_MAXV][ORC_MAXV], g[ORC_MAXV], dq[ORC_MAXV];
    memcpy(H, d->qM, sizeof(H));
    for (int r = 0; r < nv; r++) {
      double v = -d->qfrc_smooth[r];
      for (int c = 0; c < nv; c++) v += d->qM[r][c] * d->qacc[c];
      g[r] = v;
    }
    for (int i = 0; i < ne; i++) {
      if (state[i] == ST_OFF) continue;
      double f = row_force(d, i, jar[i], state[i]);
      for (int r = 0; r < nv; r++) {
        if (d->efc_J[i][r] == 0) continue;
        g[r] -= f * d->efc_J[i][r];
        if (state[i] == ST_QUAD)
          for (int c = 0; c < nv; c++) H[r][c] += d->efc_D[i] * d->efc_J[i][r] * d->efc_J[i][c];
      }
    }
    for (int r = 0; r < nv; r++) dq[r] = -g[r];
    chol_factor(H, nv);
    chol_solve(H, nv, dq);
    d->solver_niter = iter + 1;
    /* full step */
    double xt[ORC_MAXV], jar_t[ORC_MAXEFC];
    int state_t[ORC_MAXEFC], same = 1;
    for (int r = 0; r < nv; r++) xt[r] = d->qacc[r] + dq[r];
    efc_residual(m, d, xt, jar_t, state_t);
    for (int i = 0; i < ne; i++) if (state_t[i] != state[i]) same = 0;
    if (same) { memcpy(d->qacc, xt, sizeof(double) * nv); break; }
    /* exact line search along dq: phi'(a) = p0 + a p1 - sum_i f_i(a) jd_i is piecewise linear; walk its pieces */
    double jd[ORC_MAXEFC], p0 = 0, p1 = 0;
    for (int r = 0; r < nv; r++) {
      double md = 0, gm = -d->qfrc_smooth[r];
      for (int c = 0; c < nv; c++) { md += d->qM[r][c] * dq[c]; gm += d->qM[r][c] * d->qacc[c]; }
      p0 += gm * dq[r];
      p1 += md * dq[r];
    }
    for (int i = 0; i < ne; i++) {
      double v = 0;
      for (int j = 0; j < nv; j++) v += d->efc_J[i][j] * dq[j];
      jd[i] = v;
    }
    int st[ORC_MAXEFC];
    for (int i = 0; i < ne; i++) st[i] = row_state(d, i, jar[i], jd[i]);
    double alpha = 0;
    for (int guard = 0; guard < 3 * ne + 2; guard++) {
      double c0 = p0, c1 = p1, a_next = INFINITY;
      for (int i = 0; i < ne; i++) {
        if (st[i] == ST_QUAD) { c0 += d->efc_D[i] * jar[i] * jd[i]; c1 += d->efc_D[i] * jd[i] * jd[i]; }
        else if (st[i] != ST_OFF) c0 -= row_force(d, i, 0, st[i]) * jd[i];
        if (d->efc_type[i] == ORC_EFC_EQUALITY || jd[i] == 0) continue;
        /* the zone boundary this row crosses next while moving in direction jd */
        double bound;
        if (d->efc_type[i] == ORC_EFC_LIMIT) bound = 0;
        else {
          double rf = d->efc_frictionloss[i] / d->efc_D[i];
          if (jd[i] > 0) { if (st[i] == ST_LINPOS) continue; bound = st[i] == ST_LINNEG ? -rf : rf; }
          else { if (st[i] == ST_LINNEG) continue; bound = st[i] == ST_LINPOS ? rf : -rf; }
        }
        double ab = (bound - jar[i]) / jd[i];
        if (ab > alpha && ab < a_next) a_next = ab;
      }
      double a_star = -c0 / c1;
      if (a_star <= a_next) { if (a_star > alpha) alpha = a_star; break; }
      alpha = a_next;
      for (int i = 0; i < ne; i++) {
        if (d->efc_type[i] == ORC_EFC_EQUALITY || jd[i] == 0) continue;
        if (d->efc_type[i] == ORC_EFC_LIMIT) {
          if (-jar[i] / jd[i] == a_next) st[i] = st[i] == ST_QUAD ? ST_OFF : ST_QUAD;
        } else {
          double rf = d->efc_frictionloss[i] / d->efc_D[i];
          if (jd[i] > 0) {
            if (st[i] == ST_LINNEG && (-rf - jar[i]) / jd[i] == a_next) st[i] = ST_QUAD;
            else if (st[i] == ST_QUAD && (rf - jar[i]) / jd[i] == a_next) st[i] = ST_LINPOS;
          } else {
            if (st[i] == ST_LINPOS && (rf - jar[i]) / jd[i] == a_next) st[i] = ST_QUAD;
            else if (st[i] == ST_QUAD && (-rf - jar[i]) / jd[i] == a_next) st[i] = ST_LINNEG;
          }
        }
      }
    }
    for (int r = 0; r < nv; r++) d->qacc[r] += alpha * dq[r];
  }
  efc_residual(m, d, d->qacc, jar, state);
  for (int i = 0; i < ne; i++) {
    d->efc_force[i] = row_force(d, i, jar[i], state[i]);
    for (int j = 0; j < nv; j++) d->qfrc_constraint[j] += d->efc_J[i][j] * d->efc_force[i];
  }
}

void orc_step2(const orc_model* m, orc_data* d) {
  int nv = m->njnt;
  fwd_actuation(m, d);
  /* mj_fwdAcceleration */
  for (int j = 0; j < nv; j++) d->qfrc_smooth[j] = d->qfrc_passive[j] - d->qfrc_bias[j] + d->qfrc_actuator[j];
  if (d->coupled) {
    /* mj_fwdConstraint over robot + box in one problem (rcs_contact.c): Newton, then the noslip pass over all rows */
    if (m->box.present) orc_box_smooth(&m->box, &d->box, m->gravity);
    orc_solve_coupled(m, d);
    memcpy(d->qacc_warmstart, d->qacc, sizeof(double) * nv);
    if (m->box.present) orc_box_integrate(&m->box, &d->box, m->timestep);
  } else {
    /* mj_fwdConstraint (noslip has no friction rows to act on without contacts / frictionloss) */
    solve_constraints(m, d);
    memcpy(d->qacc_warmstart, d->qacc, sizeof(double) * nv);
    if (m->box.present) {
      /* the box's block of mj_fwdConstraint + integration; the first noslip sweep's improvement counts the cost
         0.5 f^2 R of every non-equality row of the scene, i.e. the robot's too */
      double imp0 = 0;
      for (int i = 0; i < d->nefc; i++)
        if (d->efc_type[i] != ORC_EFC_EQUALITY) imp0 += 0.5 * d->efc_force[i] * d->efc_force[i] / d->efc_D[i];
      orc_box_step2(&m->box, &d->box, m->gravity, m->timestep, imp0);
    }
  }
  /* mj_implicit, implicitfast: (M - h dF/dv) qacc = qfrc_smooth + qfrc_constraint with
     dF/dv = -damping (passive) + moment^T bias_vel moment (actuators not clamped by forcerange) */
  double A[ORC_MAXV][ORC_MAXV], rhs[ORC_MAXV];
  double h = m->timestep;
  memcpy(A, d->qM, sizeof(A));
  for (int j = 0; j < nv; j++) A[j][j] += h * m->dof_damping[j];
  for (int u = 0; u < m->nu; u++) {
    if (!m->actuator_biastype[u]) continue;
    double bv = m->actuator_biasprm[u][2];
    if (bv == 0) continue;
    if (m->actuator_forcelimited[u] &&
        (d->actuator_force[u] <= m->actuator_forcerange[u][0] || d->actuator_force[u] >= m->actuator_forcerange[u][1]))
      continue;
    double row[ORC_MAXV];
    actuator_moment(m, u, row);
    for (int r = 0; r < nv; r++)
      for (int c = 0; c < nv; c++) A[r][c] -= h * row[r] * bv * row[c];
  }
  for (int j = 0; j < nv; j++) rhs[j] = d->qfrc_smooth[j] + d->qfrc_constraint[j];
  chol_factor(A, nv);
  chol_solve(A, nv, rhs);
  /* mj_advance */
  for (int j = 0; j < nv; j++) d->qvel[j] += h * rhs[j];
  for (int j = 0; j < nv; j++) d->qpos[j] += h * d->qvel[j];
  d->time += h;
}

/* mj_resetData */
void orc_reset_data(const orc_model* m, orc_data* d) {
  memset(d, 0, sizeof(*d));
  for (int j = 0; j < m->njnt; j++) d->qpos[j] = m->qpos0[j];
  if (m->box.present) orc_box_reset(&m->box, &d->box);
}

/* mj_setConst / set0: dof_invweight0 = diag(M(qpos0)^-1) */
void orc_set0(orc_model* m) {
  orc_data d;
  orc_reset_data(m, &d);
  orc_mass_matrix(m, &d);
  double L[ORC_MAXV][ORC_MAXV];
  memcpy(L, d.qM, sizeof(L));
  chol_factor(L, m->njnt);
  for (int j = 0; j < m->njnt; j++) {
    double e[ORC_MAXV];
    memset(e, 0, sizeof(e));
    e[j] = 1;
    chol_solve(L, m->njnt, e);
    m->dof_invweight0[j] = e[j];
  }
  /* geom_aabb: bounding box in the geom frame (mesh: of its vertices) */
  for (int g = 0; g < m->ngeom; g++) {
    double* bb = m->geom_aabb[g];
    const double* sz = m->geom_size[g];
    memset(bb, 0, 6 * sizeof(double));
    if (m->geom_type[g] == 7 && m->geom_vertnum[g] > 0) {
      double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
      for (int v = 0; v < m->geom_vertnum[g]; v++)
        for (int k = 0; k < 3; k++) {
          const double w = m->mesh_vert[3 * (m->geom_vertadr[g] + v) + k];
          if (w < lo[k]) lo[k] = w;
          if (w > hi[k]) hi[k] = w;
        }
      for (int k = 0; k < 3; k++) { bb[k] = 0.5 * (lo[k] + hi[k]); bb[3 + k] = 0.5 * (hi[k] - lo[k]); }
    } else if (m->geom_type[g] == 6) { bb[3] = sz[0]; bb[4] = sz[1]; bb[5] = sz[2]; }
    else if (m->geom_type[g] == 3) { bb[3] = bb[4] = sz[0]; bb[5] = sz[0] + sz[1]; }
    else if (m->geom_type[g] == 2) { bb[3] = bb[4] = bb[5] = sz[0]; }
    /* geom_rbound: about the geom's own origin (mesh: farthest vertex) */
    double rb = sz[0];
    if (m->geom_type[g] == 7) {
      rb = 0;
      for (int v = 0; v < m->geom_vertnum[g]; v++) {
        const double* w = m->mesh_vert + 3 * (m->geom_vertadr[g] + v);
        const double r2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
        if (r2 > rb) rb = r2;
      }
      rb = sqrt(rb);
    } else if (m->geom_type[g] == 6) rb = sqrt(sz[0] * sz[0] + sz[1] * sz[1] + sz[2] * sz[2]);
    else if (m->geom_type[g] == 3) rb = sz[0] + sz[1];
    m->geom_rbound[g] = rb;
    m->geom_center[g][0] = m->geom_center[g][1] = m->geom_center[g][2] = 0;
    if (m->geom_type[g] == 7 && m->geom_vertnum[g] > 0) {
      double c[3] = {0, 0, 0};
      for (int v = 0; v < m->geom_vertnum[g]; v++)
        for (int k = 0; k < 3; k++) c[k] += m->mesh_vert[3 * (m->geom_vertadr[g] + v) + k];
      for (int k = 0; k < 3; k++) m->geom_center[g][k] = c[k] / m->geom_vertnum[g];
    }
  }
  /* body_invweight0 (translational): mean diagonal of J M^-1 J' for the body's centre-of-mass Jacobian at qpos0 */
  for (int b = 0; b < m->nbody; b++) {
    m->body_invweight0[b] = 0;
    if (m->body_weldid[b] == 0) continue; /* static */
    double tr = 0;
    for (int k = 0; k < 3; k++) {
      double Jr[ORC_MAXV], y[ORC_MAXV], f[3] = {0, 0, 0};
      f[k] = 1;
      memset(Jr, 0, sizeof(Jr));
      int bb = b;
      while (bb > 0) {
        int j = m->body_jntadr[bb];
        if (j >= 0) {
          double col[3];
          if (m->jnt_type[j] == ORC_JNT_SLIDE) v3_copy(col, d.xaxis[j]);
          else {
            double r[3];
            v3_sub(r, d.xipos[b], d.xanchor[j]);
            v3_cross(col, d.xaxis[j], r);
          }
          Jr[j] = v3_dot(col, f);
        }
        bb = m->body_parentid[bb];
      }
      memcpy(y, Jr, sizeof(y));
      chol_solve(L, m->njnt, y);
      for (int j = 0; j < m->njnt; j++) tr += Jr[j] * y[j];
    }
    m->body_invweight0[b] = tr / 3;
  }
  { /* mj_setConst: stat.meaninertia = mean diagonal of M(qpos0) over all dofs */
    double s = 0;
    for (int j = 0; j < m->njnt; j++) s += d.qM[j][j];
    m->box.nv_total = m->njnt;
    if (m->box.present) {
      s += 3 * m->box.mass + m->box.inertia[0] + m->box.inertia[1] + m->box.inertia[2];
      m->box.nv_total += 6;
    }
    m->box.meaninertia = s / m->box.nv_total;
  }
}
