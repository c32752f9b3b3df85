/*
 * rcs_object.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see rcs_oracle.h).
 *
 * One free rigid box on the floor plane: the `box_geom` body of the reference's pick-up scene
 * (assets/scenes/fr3_simple_pick_up/scene.xml:30-33, options assets/fr3/mjcf/fr3_common.xml:3: elliptic cones,
 * impratio 20, 5 noslip iterations), moved by `RandomCubePos` and read by `PickCubeSuccessWrapper`
 * (python/rcs/envs/sim.py:358-431).
 *
 * PARITY UNPINNED: everything here restates MuJoCo 3.2.6 (absent from /root/reference and from this image) from its
 * published description -- free-joint kinematics and integration, the plane-box collider, contact parameter mixing,
 * the constraint impedance / reference acceleration, the elliptic-cone primal cost with its three zones, and the
 * noslip post-pass (dual PGS on the friction dimensions without regularisation).  While no robot geom touches the box it shares no
 * constraint row with the robot, so its block of the constrained problem separates exactly and is solved on its own
 * (orc_box_step2; with robot-box contacts the coupled problem of rcs_contact.c takes over: orc_box_smooth / orc_box_integrate); the solver iterates the same strictly convex cost MuJoCo's Newton
 * does, to a tighter tolerance.  The noslip pass is NOT iterated to convergence in MuJoCo (5 sweeps), so it is
 * restated sweep by sweep, including the early exit on the scaled cost improvement.
 */
#include <math.h>
#include <string.h>

#include "rcs_oracle.h"

#define MINVAL 1e-15

static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void cross3(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

/* wxyz quaternion -> row-major rotation matrix (mju_quat2Mat) */
static void quat2mat(const double* q, double* R) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = w * w + x * x - y * y - z * z;
  R[4] = w * w - x * x + y * y - z * z;
  R[8] = w * w - x * x - y * y + z * z;
  R[1] = 2 * (x * y - w * z);
  R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z);
  R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y);
  R[7] = 2 * (y * z + w * x);
}

/* mju_normalize4 */
static void normalize4(double* q) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) {
    q[0] = 1;
    q[1] = q[2] = q[3] = 0;
  } else if (fabs(n - 1) > MINVAL) {
    double s = 1 / n;
    for (int k = 0; k < 4; k++) q[k] *= s;
  }
}

void orc_box_reset(const orc_box* b, orc_box_data* d) {
  memset(d, 0, sizeof(*d));
  memcpy(d->qpos, b->qpos0, sizeof(d->qpos));
  memcpy(d->xpos, b->qpos0, sizeof(d->xpos));
  memcpy(d->xquat, b->qpos0 + 3, sizeof(d->xquat));
}

/* getimpedance() of mj_makeImpedance: impedance at constraint violation |pos - margin| */
static double impedance(const double* solimp, double x) {
  double dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
  if (dmin < 0.0001) dmin = 0.0001; if (dmin > 0.9999) dmin = 0.9999;
  if (dmax < 0.0001) dmax = 0.0001; if (dmax > 0.9999) dmax = 0.9999;
  if (width < 0) width = 0;
  if (mid < 0.0001) mid = 0.0001; if (mid > 0.9999) mid = 0.9999;
  if (power < 1) power = 1;
  if (dmin == dmax || width <= MINVAL) return 0.5 * (dmin + dmax);
  x = fabs(x) / width;
  if (x >= 1) return dmax;
  if (x <= 0) return dmin;
  double y;
  if (power == 1) y = x;
  else if (x <= mid) y = pow(x / mid, power) * mid;
  else y = 1 - pow((1 - x) / (1 - mid), power) * (1 - mid);
  return dmin + y * (dmax - dmin);
}

/* position + velocity stage of the box: mj_kinematics (free joint), mjc_PlaneBox, mj_instantiateContact,
   mj_makeImpedance, mj_referenceConstraint */
void orc_box_step1(const orc_box* b, orc_box_data* d, double timestep) {
  normalize4(d->qpos + 3);
  memcpy(d->xpos, d->qpos, sizeof(d->xpos));
  memcpy(d->xquat, d->qpos + 3, sizeof(d->xquat));
  double R[9];
  quat2mat(d->qpos + 3, R);
  const double* p = d->qpos;
  /* mjc_PlaneBox: plane normal +z through (0, 0, plane_z); corners in MuJoCo's enumeration order, at most 4 */
  const double nrm[3] = {0, 0, 1};
  double dist = p[2] - b->plane_z;
  d->ncon = 0;
  for (int i = 0; i < 8 && d->ncon < 4; i++) {
    double vec[3] = {(i & 1 ? b->size[0] : -b->size[0]), (i & 2 ? b->size[1] : -b->size[1]),
                     (i & 4 ? b->size[2] : -b->size[2])};
    double corner[3];
    for (int r = 0; r < 3; r++) corner[r] = R[3 * r] * vec[0] + R[3 * r + 1] * vec[1] + R[3 * r + 2] * vec[2];
    double ldist = dot3(nrm, corner);
    if (dist + ldist > 0 /* margin */ || ldist > 0) continue;
    int c = d->ncon++;
    d->con_dist[c] = dist + ldist;
    for (int r = 0; r < 3; r++) d->con_pos[c][r] = corner[r] + p[r] - nrm[r] * d->con_dist[c] * 0.5;
  }
  /* contact frame of normal (0,0,1) by mju_makeFrame: y = (0,1,0), z = x cross y = (-1,0,0) */
  const double frame[3][3] = {{0, 0, 1}, {0, 1, 0}, {-1, 0, 0}};
  /* solref -> stiffness / damping (mj_makeImpedance; refsafe: time constant at least 2 timesteps) */
  double tc = b->solref[0], dr = b->solref[1];
  if (tc < 2 * timestep) tc = 2 * timestep;
  double dmax = b->solimp[1];
  if (dmax < 0.0001) dmax = 0.0001; if (dmax > 0.9999) dmax = 0.9999;
  double K = 1 / (dmax * dmax * tc * tc * dr * dr), B = 2 / (dmax * tc);
  double invweight = 1 / b->mass; /* body_invweight0[box][0]; the plane's body (world) has 0 */
  for (int c = 0; c < d->ncon; c++) {
    double r[3] = {d->con_pos[c][0] - p[0], d->con_pos[c][1] - p[1], d->con_pos[c][2] - p[2]};
    for (int k = 0; k < 3; k++) {
      double* J = d->J[3 * c + k];
      double rxa[3];
      cross3(r, frame[k], rxa);
      for (int j = 0; j < 3; j++) J[j] = frame[k][j];
      for (int j = 0; j < 3; j++) J[3 + j] = R[j] * rxa[0] + R[3 + j] * rxa[1] + R[6 + j] * rxa[2]; /* R^T (r x a) */
    }
    double imp = impedance(b->solimp, d->con_dist[c]);
    double R0 = (1 - imp) / imp * invweight;
    if (R0 < MINVAL) R0 = MINVAL;
    double R1 = R0 / (b->impratio > MINVAL ? b->impratio : MINVAL);
    d->R[3 * c] = R0;
    d->R[3 * c + 1] = R1;
    d->R[3 * c + 2] = R1 * b->friction[0] * b->friction[0] / (b->friction[0] * b->friction[0]); /* condim 3: mu2 = mu1 */
    d->con_mu[c] = b->friction[0] * sqrt(R1 / R0);
    for (int k = 0; k < 3; k++) {
      d->D[3 * c + k] = 1 / d->R[3 * c + k];
      double vel = 0;
      for (int j = 0; j < 6; j++) vel += d->J[3 * c + k][j] * d->qvel[j];
      d->aref[3 * c + k] = -B * vel - (k == 0 ? K * imp * d->con_dist[c] : 0.0);
    }
  }
}

/* ---- primal cost of the box block: Gauss term + elliptic-cone contact cost (mj_constraintUpdate) */
typedef struct {
  double cost, grad[6], H[6][6];
} primal;

static void contact_cost(const orc_box* b, const orc_box_data* d, int c, const double* jar, double* cost, double* f,
                         double Hc[3][3], int* zone) {
  const double mu = d->con_mu[c], fr[2] = {b->friction[0], b->friction[0]};
  const double* D = d->D + 3 * c;
  double U[3] = {jar[0] * mu, jar[1] * fr[0], jar[2] * fr[1]};
  double N = U[0], T = sqrt(U[1] * U[1] + U[2] * U[2]);
  memset(Hc, 0, 9 * sizeof(double));
  f[0] = f[1] = f[2] = 0;
  *cost = 0;
  if (N >= mu * T) { /* top zone: separating, no force */
    *zone = 0;
  } else if (mu * N + T <= 0) { /* bottom zone: quadratic in every row */
    *zone = 2;
    for (int k = 0; k < 3; k++) {
      *cost += 0.5 * D[k] * jar[k] * jar[k];
      f[k] = -D[k] * jar[k];
      Hc[k][k] = D[k];
    }
  } else { /* middle zone: 0.5 * Dm * (N - mu T)^2 */
    *zone = 1;
    const double s[3] = {mu, fr[0], fr[1]};
    double Dm = D[0] / (mu * mu * (1 + mu * mu));
    double NmT = N - mu * T;
    *cost = 0.5 * Dm * NmT * NmT;
    double u[2] = {U[1] / T, U[2] / T};
    double gU[3] = {Dm * NmT, -Dm * NmT * mu * u[0], -Dm * NmT * mu * u[1]};
    for (int k = 0; k < 3; k++) f[k] = -s[k] * gU[k];
    double HU[3][3];
    HU[0][0] = Dm;
    for (int j = 0; j < 2; j++) {
      HU[0][1 + j] = HU[1 + j][0] = -Dm * mu * u[j];
      for (int k = 0; k < 2; k++)
        HU[1 + j][1 + k] = Dm * mu * mu * u[j] * u[k] - Dm * NmT * mu * ((j == k ? 1.0 : 0.0) - u[j] * u[k]) / T;
    }
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++) Hc[j][k] = s[j] * s[k] * HU[j][k];
  }
}

static void primal_eval(const orc_box* b, orc_box_data* d, const double* x, primal* P, int want_H) {
  const double Md[6] = {b->mass, b->mass, b->mass, b->inertia[0], b->inertia[1], b->inertia[2]};
  P->cost = 0;
  if (want_H) memset(P->H, 0, sizeof(P->H));
  for (int j = 0; j < 6; j++) {
    double dx = x[j] - d->qacc_smooth[j];
    P->grad[j] = Md[j] * dx;
    P->cost += 0.5 * Md[j] * dx * dx;
    if (want_H) P->H[j][j] = Md[j];
  }
  for (int c = 0; c < d->ncon; c++) {
    double jar[3], cost, f[3], Hc[3][3];
    for (int k = 0; k < 3; k++) {
      double s = 0;
      for (int j = 0; j < 6; j++) s += d->J[3 * c + k][j] * x[j];
      jar[k] = s - d->aref[3 * c + k];
    }
    contact_cost(b, d, c, jar, &cost, f, Hc, &d->zone[c]);
    P->cost += cost;
    for (int k = 0; k < 3; k++) {
      d->force[3 * c + k] = f[k];
      for (int j = 0; j < 6; j++) P->grad[j] -= d->J[3 * c + k][j] * f[k];
    }
    if (want_H)
      for (int k = 0; k < 3; k++)
        for (int l = 0; l < 3; l++) {
          if (Hc[k][l] == 0) continue;
          for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) P->H[i][j] += d->J[3 * c + k][i] * Hc[k][l] * d->J[3 * c + l][j];
        }
  }
}

/* in-place Cholesky solve of a 6x6 SPD system */
static void chol6_solve(double A[6][6], double* x) {
  for (int j = 0; j < 6; j++) {
    for (int k = 0; k < j; k++) A[j][j] -= A[j][k] * A[j][k];
    A[j][j] = sqrt(A[j][j]);
    for (int i = j + 1; i < 6; i++) {
      for (int k = 0; k < j; k++) A[i][j] -= A[i][k] * A[j][k];
      A[i][j] /= A[j][j];
    }
  }
  for (int i = 0; i < 6; i++) {
    for (int k = 0; k < i; k++) x[i] -= A[i][k] * x[k];
    x[i] /= A[i][i];
  }
  for (int i = 5; i >= 0; i--) {
    for (int k = i + 1; k < 6; k++) x[i] -= A[k][i] * x[k];
    x[i] /= A[i][i];
  }
}

/* Newton with a safeguarded line search (1-D Newton on phi', to 1e-3: the outer gradient test sets the accuracy) on the
   strictly convex primal cost */
static void box_newton(const orc_box* b, orc_box_data* d) {
  primal P, Q;
  double x[6], xs[6];
  /* warm start: the better of qacc_warmstart and qacc_smooth (mj_fwdConstraint / warmstart()) */
  memcpy(x, d->qacc_warmstart, sizeof(x));
  memcpy(xs, d->qacc_smooth, sizeof(xs));
  primal_eval(b, d, xs, &Q, 0);
  primal_eval(b, d, x, &P, 0);
  if (Q.cost < P.cost) memcpy(x, xs, sizeof(x));
  const double scale = 1 / (b->meaninertia * (b->nv_total > 1 ? b->nv_total : 1));
  int it = 0;
  for (; it < 50; it++) {
    primal_eval(b, d, x, &P, 1);
    double g2 = 0;
    for (int j = 0; j < 6; j++) g2 += P.grad[j] * P.grad[j];
    if (scale * sqrt(g2) < 1e-13) break;
    double p[6];
    for (int j = 0; j < 6; j++) p[j] = -P.grad[j];
    chol6_solve(P.H, p);
    /* line search: root of phi'(a) = grad(x + a p) . p in a bracket [lo, hi], 1-D Newton steps with bisection as
       the fallback; a = 1 is the exact minimiser whenever no contact changes zone along the step */
    double lo = 0, hi = -1, a = 1, dphi0 = 0, dx = 1e300, dxold = 1e300;
    for (int j = 0; j < 6; j++) dphi0 += P.grad[j] * p[j];
    if (!(dphi0 < 0)) break;
    double best = 1;
    for (int ls = 0; ls < 20; ls++) {
      double xa[6];
      for (int j = 0; j < 6; j++) xa[j] = x[j] + a * p[j];
      primal_eval(b, d, xa, &Q, 1);
      double dphi = 0, ddphi = 0;
      for (int j = 0; j < 6; j++) {
        dphi += Q.grad[j] * p[j];
        for (int k = 0; k < 6; k++) ddphi += p[j] * Q.H[j][k] * p[k];
      }
      best = a;
      if (fabs(dphi) <= 1e-3 * fabs(dphi0)) break;
      if (dphi < 0) lo = a; else hi = a;
      double an = a - dphi / ddphi;
      /* Newton on phi' with the bracket as the safeguard (the rtsafe rule): bisect when the step leaves the bracket or does not
         at least halve the step before last -- phi' is piecewise smooth, between two pieces Newton alone can cycle */
      if (hi > 0 && (!(an > lo && an < hi) || fabs(2 * dphi) > fabs(dxold * ddphi))) an = 0.5 * (lo + hi);
      if (hi < 0 && !(an > lo)) an = 2 * a;
      if (fabs(an - a) <= 1e-3 * a) break;
      dxold = dx;
      dx = an - a;
      a = an;
    }
    for (int j = 0; j < 6; j++) x[j] += best * p[j];
  }
  primal_eval(b, d, x, &P, 0); /* forces at the solution */
  memcpy(d->qacc, x, sizeof(x));
  d->newton_iter = it;
}

/* mju_QCQP2: min 0.5 x'Ax + x'b  s.t.  sum (x_i / d_i)^2 <= r^2 */
static int qcqp2(double* res, const double* Ain, const double* bin, const double* dd, double r) {
  double b1 = bin[0] * dd[0], b2 = bin[1] * dd[1];
  double A11 = Ain[0] * dd[0] * dd[0], A22 = Ain[3] * dd[1] * dd[1], A12 = Ain[1] * dd[0] * dd[1];
  double la = 0, v1 = 0, v2 = 0;
  for (int iter = 0; iter < 20; iter++) {
    double det = (A11 + la) * (A22 + la) - A12 * A12;
    if (det < 1e-10) {
      res[0] = res[1] = 0;
      return 0;
    }
    double di = 1 / det, P11 = (A22 + la) * di, P22 = (A11 + la) * di, P12 = -A12 * di;
    v1 = -P11 * b1 - P12 * b2;
    v2 = -P12 * b1 - P22 * b2;
    double val = v1 * v1 + v2 * v2 - r * r;
    if (val < 1e-10) break;
    double deriv = -2 * (P11 * v1 * v1 + 2 * P12 * v1 * v2 + P22 * v2 * v2);
    double delta = -val / deriv;
    if (delta < 1e-10) break;
    la += delta;
  }
  res[0] = v1 * dd[0];
  res[1] = v2 * dd[1];
  return la != 0;
}

/* mj_solNoSlip restricted to the box's contact rows.  improvement0: 0.5 * force^2 * R summed over the scene's other
   non-equality rows (the robot's limit rows), which MuJoCo adds to the first sweep's improvement. */
static void box_noslip(const orc_box* b, orc_box_data* d, double improvement0) {
  const int n = 3 * d->ncon;
  const double Mi[6] = {1 / b->mass, 1 / b->mass, 1 / b->mass, 1 / b->inertia[0], 1 / b->inertia[1], 1 / b->inertia[2]};
  double A[12][12], bb[12];
  for (int i = 0; i < n; i++) {
    for (int k = 0; k < n; k++) {
      double s = 0;
      for (int j = 0; j < 6; j++) s += d->J[i][j] * Mi[j] * d->J[k][j];
      A[i][k] = s;
    }
    double s = 0;
    for (int j = 0; j < 6; j++) s += d->J[i][j] * d->qacc_smooth[j];
    bb[i] = s - d->aref[i];
  }
  const double scale = 1 / (b->meaninertia * (b->nv_total > 1 ? b->nv_total : 1));
  const double fr[2] = {b->friction[0], b->friction[0]};
  double* force = d->force;
  int iter = 0;
  while (iter < b->noslip_iterations) {
    double improvement = 0;
    if (iter == 0) {
      improvement = improvement0;
      for (int i = 0; i < n; i++) improvement += 0.5 * force[i] * force[i] * d->R[i];
    }
    for (int c = 0; c < d->ncon; c++) {
      const int i = 3 * c;
      double res[3], old[3];
      for (int k = 0; k < 3; k++) {
        double s = bb[i + k];
        for (int j = 0; j < n; j++) s += A[i + k][j] * force[j];
        res[k] = s;
        old[k] = force[i + k];
      }
      if (force[i] < MINVAL) {
        force[i] = force[i + 1] = force[i + 2] = 0;
      } else {
        double Ac[4] = {A[i + 1][i + 1], A[i + 1][i + 2], A[i + 2][i + 1], A[i + 2][i + 2]};
        double bc[2] = {res[1] - Ac[0] * old[1] - Ac[1] * old[2], res[2] - Ac[2] * old[1] - Ac[3] * old[2]};
        double v[2];
        int active = qcqp2(v, Ac, bc, fr, force[i]);
        if (active) {
          double s = v[0] * v[0] / (fr[0] * fr[0]) + v[1] * v[1] / (fr[1] * fr[1]);
          s = sqrt(force[i] * force[i] / (s > MINVAL ? s : MINVAL));
          v[0] *= s;
          v[1] *= s;
        }
        force[i + 1] = v[0];
        force[i + 2] = v[1];
      }
      /* costChange(): 0.5 delta' A delta + delta' res; a step that raises the dual cost is undone */
      double dl[3] = {force[i] - old[0], force[i + 1] - old[1], force[i + 2] - old[2]};
      double change = 0;
      for (int k = 0; k < 3; k++) {
        for (int l = 0; l < 3; l++) change += 0.5 * dl[k] * A[i + k][i + l] * dl[l];
        change += dl[k] * res[k];
      }
      if (change > 1e-10) {
        for (int k = 0; k < 3; k++) force[i + k] = old[k];
        change = 0;
      }
      improvement -= change;
    }
    improvement *= scale;
    iter++;
    if (improvement < b->noslip_tolerance) break;
  }
  d->noslip_iter = iter;
  /* dualFinish: qacc = qacc_smooth + M^-1 J' force */
  for (int j = 0; j < 6; j++) {
    double s = 0;
    for (int i = 0; i < n; i++) s += d->J[i][j] * force[i];
    d->qacc[j] = d->qacc_smooth[j] + Mi[j] * s;
  }
}

/* mj_fwdAcceleration of the box: gravity and the gyroscopic bias (angular velocity in the body frame) */
void orc_box_smooth(const orc_box* b, orc_box_data* d, const double* gravity) {
  const double* w = d->qvel + 3;
  double Iw[3] = {b->inertia[0] * w[0], b->inertia[1] * w[1], b->inertia[2] * w[2]}, gyro[3];
  cross3(w, Iw, gyro);
  for (int j = 0; j < 3; j++) {
    d->qfrc_smooth[j] = b->mass * gravity[j];
    d->qfrc_smooth[3 + j] = -gyro[j];
    d->qacc_smooth[j] = gravity[j];
    d->qacc_smooth[3 + j] = -gyro[j] / b->inertia[j];
  }
}

/* integration of the box with the acceleration in d->qacc: implicitfast has no velocity-dependent force to treat
   implicitly here (plain semi-implicit Euler), mj_integratePos with mju_quatIntegrate */
void orc_box_integrate(const orc_box* b, orc_box_data* d, double h) {
  (void)b;
  const double* w = d->qvel + 3;
  memcpy(d->qacc_warmstart, d->qacc, sizeof(d->qacc));
  for (int j = 0; j < 6; j++) d->qvel[j] += h * d->qacc[j];
  for (int j = 0; j < 3; j++) d->qpos[j] += h * d->qvel[j];
  double ax[3] = {w[0], w[1], w[2]};
  double nrm = sqrt(dot3(ax, ax));
  double q[4] = {d->qpos[3], d->qpos[4], d->qpos[5], d->qpos[6]};
  normalize4(q);
  if (nrm >= MINVAL) {
    double ang = h * nrm, s = sin(0.5 * ang) / nrm, c = cos(0.5 * ang);
    double r[4] = {c, ax[0] * s, ax[1] * s, ax[2] * s};
    double o[4] = {q[0] * r[0] - q[1] * r[1] - q[2] * r[2] - q[3] * r[3], q[0] * r[1] + q[1] * r[0] + q[2] * r[3] - q[3] * r[2],
                   q[0] * r[2] - q[1] * r[3] + q[2] * r[0] + q[3] * r[1], q[0] * r[3] + q[1] * r[2] - q[2] * r[1] + q[3] * r[0]};
    memcpy(q, o, sizeof(q));
    normalize4(q);
  }
  memcpy(d->qpos + 3, q, sizeof(q));
}

/* acceleration stage + integration of the box on its own (no contact with the robot in this step): mj_fwdAcceleration,
   mj_fwdConstraint restricted to the box's block, integration */
void orc_box_step2(const orc_box* b, orc_box_data* d, const double* gravity, double h, double improvement0) {
  orc_box_smooth(b, d, gravity);
  if (d->ncon == 0) {
    memcpy(d->qacc, d->qacc_smooth, sizeof(d->qacc));
    d->newton_iter = d->noslip_iter = 0;
  } else {
    box_newton(b, d);
    if (b->noslip_iterations > 0) box_noslip(b, d, improvement0);
  }
  orc_box_integrate(b, d, h);
}
