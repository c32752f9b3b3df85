/*
 * rcs_contact.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see rcs_oracle.h).
 *
 * Contacts of the robot's collision geoms -- with the floor plane and with the free box of the pick-up scene -- and the
 * constraint problem that couples the robot's joints with the box's six degrees of freedom.  This is what makes a grasp
 * possible: finger pads (boxes, reference assets/fr3/mjcf/fr3_0.xml:145-162, friction 2) and the finger / hand / link
 * hulls against the cube (assets/scenes/fr3_simple_pick_up/scene.xml:30-33), read by SimGripper::collision_callback /
 * SimRobot::collision_callback (src/sim/SimGripper.cpp:108-130, src/sim/SimRobot.cpp:172-182) and by
 * PickCubeSuccessWrapper (python/rcs/envs/sim.py:396-431).
 *
 * PARITY UNPINNED: MuJoCo 3.2.6 is absent from /root/reference and from this image; everything below restates its
 * PUBLISHED algorithms, not its source:
 *   - box-box: separating-axis test over the 15 axes, face contact = the incident face clipped against the reference
 *     face's side planes (up to 8 points), edge contact = the closest points of the two edges; contact points half way
 *     between the surfaces (mjc_BoxBox's documented output: up to 8 contacts, midpoint positions);
 *   - convex-convex (hull / capsule against the box): Minkowski Portal Refinement as published by G. Snethen
 *     (XenoCollide) and used by MuJoCo through libccd (ccd_tolerance 1e-6, ccd_iterations 50): one contact per pair;
 *   - plane-convex: the support vertex along the plane normal, plus up to three more support vertices along directions
 *     tilted by 1e-3 about the normal (my reading of mjc_PlaneConvex's "up to 3 more contacts"); plane-box / plane-capsule
 *     as in MuJoCo's primitives (box corners in corner order, at most 4; the capsule's two end spheres);
 *   - contact parameters, impedance / reference acceleration, elliptic cones and the noslip pass exactly as rcs_object.c
 *     restates them for the box alone, over ALL rows of the scene: Newton on the primal cost of the coupled system to its
 *     minimiser (MuJoCo: tolerance 1e-8), then mj_solNoSlip sweep by sweep.
 * Closed-form pins (tests/test_host_logic.py): a pinched cube holds iff mu N >= m g, the static penetration of a contact,
 * known-answer box-box / hull-box configurations.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rcs_oracle.h"

#define MINVAL 1e-15
#define MPR_TOL 1e-6
#define MPR_ITER 50

/* ------------------------------------------------------------------ small math */
static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void cross3(const double* a, const double* b, double* o) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
static void sub3(const double* a, const double* b, double* o) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
static void copy3(double* o, const double* a) { o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; }
static double norm3(const double* a) { return sqrt(dot3(a, a)); }
static void mat_vec(const double* R, const double* v, double* o) { /* o = R v, row-major */
  double x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2], y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2], z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static void matT_vec(const double* R, const double* v, double* o) { /* o = R^T v */
  double x = R[0] * v[0] + R[3] * v[1] + R[6] * v[2], y = R[1] * v[0] + R[4] * v[1] + R[7] * v[2], z = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static void quat2mat(const double* q, double* R) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = w * w + x * x - y * y - z * z; R[4] = w * w - x * x + y * y - z * z; R[8] = w * w - x * x - y * y + z * z;
  R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y); R[3] = 2 * (x * y + w * z);
  R[5] = 2 * (y * z - w * x); R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x);
}
static void quat_mul(const double* a, const double* b, double* r) {
  double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}

/* mju_makeFrame: complete a right-handed orthonormal frame from its (unit) first row */
static void make_frame(double* f) {
  double *x = f, *y = f + 3, *z = f + 6;
  if (x[1] > -0.5 && x[1] < 0.5) { y[0] = 0; y[1] = 1; y[2] = 0; }
  else { y[0] = 0; y[1] = 0; y[2] = 1; }
  double s = dot3(x, y);
  for (int k = 0; k < 3; k++) y[k] -= s * x[k];
  s = 1 / norm3(y);
  for (int k = 0; k < 3; k++) y[k] *= s;
  cross3(x, y, z);
}

/* ------------------------------------------------------------------ box - box */

/* Sutherland-Hodgman: clip polygon (n points of 2 coordinates + carried 3rd) against  sign * p[axis] <= lim */
static int clip_poly(double (*in)[3], int n, int axis, double sign, double lim, double (*out)[3]) {
  int m = 0;
  for (int i = 0; i < n; i++) {
    const double* a = in[i];
    const double* b = in[(i + 1) % n];
    double da = sign * a[axis] - lim, db = sign * b[axis] - lim;
    if (da <= 0) { copy3(out[m], a); m++; }
    if ((da < 0 && db > 0) || (da > 0 && db < 0)) {
      double s = da / (da - db);
      for (int k = 0; k < 3; k++) out[m][k] = a[k] + s * (b[k] - a[k]);
      m++;
    }
  }
  return m;
}

int orc_box_box(const double* p1, const double* R1, const double* s1, const double* p2, const double* R2, const double* s2,
                double* pos, double* normal, double* dist) {
  double d[3], t[3], R[3][3], Q[3][3];
  sub3(p2, p1, d);
  matT_vec(R1, d, t);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      R[i][j] = R1[i] * R2[j] + R1[3 + i] * R2[3 + j] + R1[6 + i] * R2[6 + j]; /* A_i . B_j */
      Q[i][j] = fabs(R[i][j]);
    }
  /* separating-axis test; penetration along every axis, the shallowest wins (faces preferred over edges by 5 %) */
  double best = INFINITY;
  int code = -1;
  for (int i = 0; i < 3; i++) {
    double pen = s1[i] + s2[0] * Q[i][0] + s2[1] * Q[i][1] + s2[2] * Q[i][2] - fabs(t[i]);
    if (pen < 0) return 0;
    if (pen < best) { best = pen; code = i; }
  }
  double tb[3];
  for (int j = 0; j < 3; j++) {
    tb[j] = t[0] * R[0][j] + t[1] * R[1][j] + t[2] * R[2][j];
    double pen = s2[j] + s1[0] * Q[0][j] + s1[1] * Q[1][j] + s1[2] * Q[2][j] - fabs(tb[j]);
    if (pen < 0) return 0;
    if (pen < best) { best = pen; code = 3 + j; }
  }
  double ebest = INFINITY;
  int ecode = -1;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double l2 = 1 - R[i][j] * R[i][j];
      if (l2 < 1e-6) continue; /* (nearly) parallel edges: the face axes cover this direction */
      int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      double ra = s1[i1] * Q[i2][j] + s1[i2] * Q[i1][j];
      double rb = s2[j1] * Q[i][j2] + s2[j2] * Q[i][j1];
      double tl = fabs(t[i2] * R[i1][j] - t[i1] * R[i2][j]);
      double pen = (ra + rb - tl) / sqrt(l2);
      if (pen < 0) return 0;
      if (pen < ebest) { ebest = pen; ecode = 3 * i + j; }
    }
  if (ecode >= 0 && ebest * 1.05 < best) {
    /* edge - edge: one contact at the closest points of the two edges */
    int i = ecode / 3, j = ecode % 3;
    double A[3] = {R1[i], R1[3 + i], R1[6 + i]}, B[3] = {R2[j], R2[3 + j], R2[6 + j]}, n[3];
    cross3(A, B, n);
    double l = norm3(n);
    for (int k = 0; k < 3; k++) n[k] /= l;
    if (dot3(n, d) < 0) for (int k = 0; k < 3; k++) n[k] = -n[k];
    double pa[3], pb[3];
    copy3(pa, p1);
    copy3(pb, p2);
    for (int k = 0; k < 3; k++) {
      if (k != i) {
        double Ak[3] = {R1[k], R1[3 + k], R1[6 + k]};
        double sg = dot3(n, Ak) > 0 ? s1[k] : -s1[k];
        for (int c = 0; c < 3; c++) pa[c] += sg * Ak[c];
      }
      if (k != j) {
        double Bk[3] = {R2[k], R2[3 + k], R2[6 + k]};
        double sg = dot3(n, Bk) > 0 ? -s2[k] : s2[k];
        for (int c = 0; c < 3; c++) pb[c] += sg * Bk[c];
      }
    }
    double p[3];
    sub3(pb, pa, p);
    double uaub = dot3(A, B), q1 = dot3(A, p), q2 = -dot3(B, p), dd = 1 - uaub * uaub, al = 0, be = 0;
    if (dd > 1e-4) { al = (q1 + uaub * q2) / dd; be = (uaub * q1 + q2) / dd; }
    for (int c = 0; c < 3; c++) {
      pos[c] = 0.5 * ((pa[c] + al * A[c]) + (pb[c] + be * B[c]));
      normal[c] = n[c];
    }
    dist[0] = -ebest;
    return 1;
  }
  /* face contact: reference box = owner of the winning axis; clip the other box's most anti-parallel face against the
     side planes of the reference face, keep what lies below the reference face */
  const int ref1 = code < 3;
  const int a = ref1 ? code : code - 3;
  const double *Rr = ref1 ? R1 : R2, *pr = ref1 ? p1 : p2, *sr = ref1 ? s1 : s2;
  const double *Ri = ref1 ? R2 : R1, *pi = ref1 ? p2 : p1, *si = ref1 ? s2 : s1;
  double ci[3];
  sub3(pi, pr, ci);
  double Ar[3] = {Rr[a], Rr[3 + a], Rr[6 + a]};
  const double sgn = dot3(ci, Ar) >= 0 ? 1.0 : -1.0;
  double n[3] = {sgn * Ar[0], sgn * Ar[1], sgn * Ar[2]};
  int b = 0;
  double bestdot = -1;
  for (int k = 0; k < 3; k++) {
    double Bk[3] = {Ri[k], Ri[3 + k], Ri[6 + k]};
    double v = fabs(dot3(n, Bk));
    if (v > bestdot) { bestdot = v; b = k; }
  }
  double Bb[3] = {Ri[b], Ri[3 + b], Ri[6 + b]};
  const double sb = dot3(n, Bb) > 0 ? -si[b] : si[b];
  const int u = (b + 1) % 3, v = (b + 2) % 3, a1 = (a + 1) % 3, a2 = (a + 2) % 3;
  double Bu[3] = {Ri[u], Ri[3 + u], Ri[6 + u]}, Bv[3] = {Ri[v], Ri[3 + v], Ri[6 + v]};
  static const double su[4] = {1, -1, -1, 1}, sv[4] = {1, 1, -1, -1};
  double poly[2][16][3];
  for (int q = 0; q < 4; q++) {
    double w[3], x[3];
    for (int c = 0; c < 3; c++) w[c] = pi[c] + sb * Bb[c] + su[q] * si[u] * Bu[c] + sv[q] * si[v] * Bv[c] - pr[c];
    matT_vec(Rr, w, x); /* reference-frame coordinates */
    copy3(poly[0][q], x);
  }
  int np = 4;
  np = clip_poly(poly[0], np, a1, 1.0, sr[a1], poly[1]);
  np = clip_poly(poly[1], np, a1, -1.0, sr[a1], poly[0]);
  np = clip_poly(poly[0], np, a2, 1.0, sr[a2], poly[1]);
  np = clip_poly(poly[1], np, a2, -1.0, sr[a2], poly[0]);
  int nc = 0;
  for (int q = 0; q < np && nc < 8; q++) {
    const double depth = sr[a] - sgn * poly[0][q][a];
    if (depth < 0) continue;
    double w[3];
    mat_vec(Rr, poly[0][q], w);
    for (int c = 0; c < 3; c++) {
      pos[3 * nc + c] = w[c] + pr[c] + n[c] * 0.5 * depth;
      normal[3 * nc + c] = ref1 ? n[c] : -n[c];
    }
    dist[nc] = -depth;
    nc++;
  }
  return nc;
}

/* ------------------------------------------------------------------ Minkowski Portal Refinement */

enum { SH_HULL = 0, SH_BOX = 1, SH_CAPSULE = 2 };
typedef struct {
  int type;
  const double *p, *R;   /* world frame of the geom */
  const double* size;    /* box half extents; capsule radius, half length (axis z) */
  const double* verts;   /* hull vertices, geom frame */
  int nvert;
  double center[3];      /* an interior point, world */
} shape;

/* Support ties.  The portal refinement keeps asking for the support of A - B along the normal of a triangle of three of its
 * vertices; as soon as two of them share a vertex of one shape, that normal is perpendicular to an EDGE of the other shape by
 * construction, and the two ends of that edge tie to the last bit.  libccd (and MuJoCo through it) breaks such ties by the sign
 * of a number that is zero up to round-off -- the outcome, a contact normal that can differ by tens of degrees, then depends
 * on the compiler and on 1e-16 of the inputs (two runs of THIS file on inputs 1e-14 apart disagreed in a quarter of the
 * cube-against-base throws).  Here ties are broken by rule instead: vertices within 1e-10 m of the largest projection count
 * as tied and the lowest index wins; a box or capsule axis whose direction component is above -1e-10 takes its positive end.
 * Either outcome is one MuJoCo could have produced; the support value changes by at most 1e-10 m against MPR's 1e-6 tolerance. */
#define SUPPORT_TIE 1e-10
/* the FIRST vertex whose projection on l is within SUPPORT_TIE of the largest (-1: no vertices) */
static int hull_support_index(const double* verts, int nvert, const double* l) {
  double bestv = -INFINITY;
  for (int i = 0; i < nvert; i++) {
    double v = dot3(verts + 3 * i, l);
    if (v > bestv) bestv = v;
  }
  for (int i = 0; i < nvert; i++)
    if (dot3(verts + 3 * i, l) >= bestv - SUPPORT_TIE) return i;
  return -1;
}
static void support(const shape* s, const double* dir, double* out) {
  double l[3], w[3] = {0, 0, 0};
  matT_vec(s->R, dir, l);
  if (s->type == SH_HULL) {
    const int bi = hull_support_index(s->verts, s->nvert, l);
    copy3(w, s->verts + 3 * (bi < 0 ? 0 : bi));
  } else if (s->type == SH_BOX) {
    for (int k = 0; k < 3; k++) w[k] = l[k] >= -SUPPORT_TIE ? s->size[k] : -s->size[k];
  } else {
    double nl = norm3(l);
    w[2] = l[2] >= -SUPPORT_TIE ? s->size[1] : -s->size[1];
    if (nl > MINVAL) for (int k = 0; k < 3; k++) w[k] += s->size[0] * l[k] / nl;
  }
  mat_vec(s->R, w, out);
  out[0] += s->p[0]; out[1] += s->p[1]; out[2] += s->p[2];
}

typedef struct { double v[3], v1[3], v2[3]; } mpr_pt; /* point of the Minkowski difference and its two witnesses */

static void mpr_support(const shape* a, const shape* b, const double* dir, mpr_pt* o) {
  double nd[3] = {-dir[0], -dir[1], -dir[2]};
  support(a, dir, o->v1);
  support(b, nd, o->v2);
  sub3(o->v1, o->v2, o->v);
}
static void portal_dir(const mpr_pt* p1, const mpr_pt* p2, const mpr_pt* p3, double* dir) {
  double e1[3], e2[3];
  sub3(p2->v, p1->v, e1);
  sub3(p3->v, p1->v, e2);
  cross3(e1, e2, dir);
  double l = norm3(dir);
  if (l > MINVAL) { dir[0] /= l; dir[1] /= l; dir[2] /= l; }
}
static void expand_portal(mpr_pt* p0, mpr_pt* p1, mpr_pt* p2, mpr_pt* p3, const mpr_pt* p4) {
  double c[3];
  cross3(p4->v, p0->v, c);
  if (dot3(p1->v, c) > 0) {
    if (dot3(p2->v, c) > 0) *p1 = *p4; else *p3 = *p4;
  } else {
    if (dot3(p3->v, c) > 0) *p2 = *p4; else *p1 = *p4;
  }
}
/* squared distance of the origin from triangle (a, b, c) and the closest point */
static double origin_tri_dist2(const double* a, const double* b, const double* c, double* witness) {
  double ab[3], ac[3], ap[3] = {-a[0], -a[1], -a[2]};
  sub3(b, a, ab);
  sub3(c, a, ac);
  double d1 = dot3(ab, ap), d2 = dot3(ac, ap);
  double s, t;
  if (d1 <= 0 && d2 <= 0) { s = 0; t = 0; }
  else {
    double bp[3] = {-b[0], -b[1], -b[2]}, d3 = dot3(ab, bp), d4 = dot3(ac, bp);
    double cp[3] = {-c[0], -c[1], -c[2]}, d5 = dot3(ab, cp), d6 = dot3(ac, cp);
    double vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
    if (d3 >= 0 && d4 <= d3) { s = 1; t = 0; }
    else if (vc <= 0 && d1 >= 0 && d3 <= 0) { s = d1 / (d1 - d3); t = 0; }
    else if (d6 >= 0 && d5 <= d6) { s = 0; t = 1; }
    else if (vb <= 0 && d2 >= 0 && d6 <= 0) { s = 0; t = d2 / (d2 - d6); }
    else if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) { t = (d4 - d3) / ((d4 - d3) + (d5 - d6)); s = 1 - t; }
    else { double den = 1 / (va + vb + vc); s = vb * den; t = vc * den; }
  }
  for (int k = 0; k < 3; k++) witness[k] = a[k] + s * ab[k] + t * ac[k];
  return dot3(witness, witness);
}

/* penetration of two convex shapes: 1 if they intersect; depth, direction (shape a -> shape b) and contact position */
static int mpr_penetration(const shape* A, const shape* B, double* depth, double* dir_out, double* pos) {
  mpr_pt p0, p1, p2, p3, p4;
  double dir[3], va[3], vb[3];
  /* portal discovery */
  sub3(A->center, B->center, p0.v);
  copy3(p0.v1, A->center);
  copy3(p0.v2, B->center);
  if (fabs(p0.v[0]) < MINVAL && fabs(p0.v[1]) < MINVAL && fabs(p0.v[2]) < MINVAL) p0.v[0] = 1e-5;
  double l = norm3(p0.v);
  for (int k = 0; k < 3; k++) dir[k] = -p0.v[k] / l;
  mpr_support(A, B, dir, &p1);
  if (dot3(p1.v, dir) <= 0) { copy3(dir_out, dir); return 0; }
  cross3(p0.v, p1.v, dir);
  l = norm3(dir);
  if (l < 1e-12) {
    /* origin on the ray v0 -> v1: touching (v1 is the origin) or the segment contains it; penetration along the ray */
    double dv[3] = {p1.v[0], p1.v[1], p1.v[2]};
    *depth = norm3(dv);
    for (int k = 0; k < 3; k++) { dir_out[k] = *depth > MINVAL ? dv[k] / *depth : -p0.v[k] / norm3(p0.v); pos[k] = 0.5 * (p1.v1[k] + p1.v2[k]); }
    return 1;
  }
  for (int k = 0; k < 3; k++) dir[k] /= l;
  mpr_support(A, B, dir, &p2);
  if (dot3(p2.v, dir) <= 0) { copy3(dir_out, dir); return 0; }
  sub3(p1.v, p0.v, va);
  sub3(p2.v, p0.v, vb);
  cross3(va, vb, dir);
  l = norm3(dir);
  for (int k = 0; k < 3; k++) dir[k] /= l;
  if (dot3(dir, p0.v) > 0) { /* the portal must face away from v0 */
    mpr_pt tmp = p1; p1 = p2; p2 = tmp;
    for (int k = 0; k < 3; k++) dir[k] = -dir[k];
  }
  for (int guard = 0;; guard++) {
    if (guard > 100) { dir_out[0] = dir_out[1] = dir_out[2] = 0; return 0; }
    mpr_support(A, B, dir, &p3);
    if (dot3(p3.v, dir) <= 0) { copy3(dir_out, dir); return 0; }
    int cont = 0;
    cross3(p1.v, p3.v, va);
    if (dot3(va, p0.v) < -MINVAL) { p2 = p3; cont = 1; }
    if (!cont) {
      cross3(p3.v, p2.v, va);
      if (dot3(va, p0.v) < -MINVAL) { p1 = p3; cont = 1; }
    }
    if (!cont) break;
    sub3(p1.v, p0.v, va);
    sub3(p2.v, p0.v, vb);
    cross3(va, vb, dir);
    l = norm3(dir);
    for (int k = 0; k < 3; k++) dir[k] /= l;
  }
  /* portal refinement: push the portal outwards until the origin is inside */
  for (int it = 0;; it++) {
    portal_dir(&p1, &p2, &p3, dir);
    if (dot3(dir, p1.v) >= 0) break; /* the portal encloses the origin */
    mpr_support(A, B, dir, &p4);
    double dv4 = dot3(p4.v, dir);
    double dmax = fmax(dot3(p1.v, dir), fmax(dot3(p2.v, dir), dot3(p3.v, dir)));
    if (dv4 < 0 || dv4 - dmax <= MPR_TOL || it > MPR_ITER) { const double keep = dv4 < 0 ? 1.0 : 0.0; for (int k = 0; k < 3; k++) dir_out[k] = keep * dir[k]; return 0; }
    expand_portal(&p0, &p1, &p2, &p3, &p4);
  }
  /* penetration: refine the portal towards the surface of the difference */
  for (int it = 0;; it++) {
    portal_dir(&p1, &p2, &p3, dir);
    mpr_support(A, B, dir, &p4);
    double dv4 = dot3(p4.v, dir);
    double dmax = fmax(dot3(p1.v, dir), fmax(dot3(p2.v, dir), dot3(p3.v, dir)));
    if (dv4 - dmax <= MPR_TOL || it > MPR_ITER) {
      double w[3];
      double d2 = origin_tri_dist2(p1.v, p2.v, p3.v, w);
      *depth = sqrt(d2);
      if (*depth > MINVAL) for (int k = 0; k < 3; k++) dir_out[k] = w[k] / *depth;
      else copy3(dir_out, dir);
      /* position: barycentric combination of the witnesses */
      double b[4], c[3];
      cross3(p1.v, p2.v, c); b[0] = dot3(c, p3.v);
      cross3(p3.v, p2.v, c); b[1] = dot3(c, p0.v);
      cross3(p0.v, p1.v, c); b[2] = dot3(c, p3.v);
      cross3(p2.v, p1.v, c); b[3] = dot3(c, p0.v);
      double sum = b[0] + b[1] + b[2] + b[3];
      if (sum <= 0) {
        b[0] = 0;
        cross3(p2.v, p3.v, c); b[1] = dot3(c, dir);
        cross3(p3.v, p1.v, c); b[2] = dot3(c, dir);
        cross3(p1.v, p2.v, c); b[3] = dot3(c, dir);
        sum = b[1] + b[2] + b[3];
      }
      const mpr_pt* pp[4] = {&p0, &p1, &p2, &p3};
      for (int k = 0; k < 3; k++) {
        double a1 = 0, a2 = 0;
        for (int q = 0; q < 4; q++) { a1 += b[q] * pp[q]->v1[k]; a2 += b[q] * pp[q]->v2[k]; }
        pos[k] = 0.5 * (a1 + a2) / sum;
      }
      return 1;
    }
    expand_portal(&p0, &p1, &p2, &p3, &p4);
  }
}

static void hull_center(const double* verts, int nvert, const double* p, const double* R, double* out) {
  double c[3] = {0, 0, 0};
  for (int i = 0; i < nvert; i++) { c[0] += verts[3 * i]; c[1] += verts[3 * i + 1]; c[2] += verts[3 * i + 2]; }
  for (int k = 0; k < 3; k++) c[k] /= nvert > 0 ? nvert : 1;
  mat_vec(R, c, out);
  out[0] += p[0]; out[1] += p[1]; out[2] += p[2];
}

int orc_mpr_hull_box(const double* verts, int nvert, const double* ph, const double* Rh, const double* pb, const double* Rb,
                     const double* sb, double* pos, double* normal, double* dist) {
  shape H = {SH_HULL, ph, Rh, 0, verts, nvert, {0, 0, 0}}, Bx = {SH_BOX, pb, Rb, sb, 0, 0, {pb[0], pb[1], pb[2]}};
  hull_center(verts, nvert, ph, Rh, H.center);
  double depth;
  if (!mpr_penetration(&H, &Bx, &depth, normal, pos)) return 0;
  *dist = -depth;
  return 1;
}

/* ------------------------------------------------------------------ collision driver */
static void geom_frame_compute(const orc_model* m, const orc_data* d, int g, double* R, double* p);
/* world frames of the geoms of the position stage being collided (orc_collide fills them once; its helpers read them) */
typedef struct { double R[ORC_MAXGEOM][9], p[ORC_MAXGEOM][3]; } geom_frames;
static const geom_frames* g_frames = 0;
static void geom_frame(const orc_model* m, const orc_data* d, int g, double* R, double* p) {
  if (g_frames) { memcpy(R, g_frames->R[g], 9 * sizeof(double)); memcpy(p, g_frames->p[g], 3 * sizeof(double)); return; }
  geom_frame_compute(m, d, g, R, p);
}

static void geom_frame_compute(const orc_model* m, const orc_data* d, int g, double* R, double* p) {
  int b = m->geom_bodyid[g];
  double q[4], v[3];
  quat_mul(d->xquat[b], m->geom_quat[g], q);
  quat2mat(q, R);
  mat_vec(d->xmat[b], m->geom_pos[g], v);
  p[0] = v[0] + d->xpos[b][0]; p[1] = v[1] + d->xpos[b][1]; p[2] = v[2] + d->xpos[b][2];
}
static double geom_rbound(const orc_model* m, int g) { return m->geom_rbound[g]; }
/* MuJoCo's contact list has no fixed bound and neither has this restatement's (ORC_MAXCON is storage, far above what the scenes
   produce).  orc_set_contact_cap(): a test may ask for the first `cap` contacts of MuJoCo's order only, to see what a backend with a
   bounded contact phase does when it overflows -- never the default. */
static int g_contact_cap = ORC_MAXCON;
static int g_resolve_cap = 0;
void orc_set_contact_cap(int cap) { g_resolve_cap = cap; }
static orc_contact* add_contact(orc_data* d, int g1, int g2, int b1, int b2, const double* pos, const double* n, double dist, double mu) {
  if (d->ncon >= g_contact_cap) return 0;
  orc_contact* c = &d->contact[d->ncon];
  c->geom[0] = g1; c->geom[1] = g2; c->body[0] = b1; c->body[1] = b2;
  copy3(c->pos, pos);
  copy3(c->frame, n);
  make_frame(c->frame);
  c->dist = dist;
  c->mu = mu;
  c->efc_address = -1;
  c->zone = 0;
  d->contact_geom[d->ncon][0] = g1;
  d->contact_geom[d->ncon][1] = g2;
  d->ncon++;
  return c;
}

/* mj_collision for the RCS scenes: (floor, robot geom) pairs, (floor, free box), (robot geom, free box), in MuJoCo's
   order of body pairs.  Pair filters as MuJoCo applies them: bodies welded together never collide, contype / conaffinity
   masks must match, parent-child pairs are skipped unless the parent is the world.  Pairs of two robot geoms -- self
   collision, the robot against geoms welded to the world -- are detected (orc_data.self_geom) but carry no rows. */
static int shape_of(const orc_model* m, int g, const double* gp, const double* gR, shape* S) {
  const int t = m->geom_type[g];
  if (t != 7 && t != 6 && t != 3) return 0;
  if (t == 7 && m->geom_vertnum[g] == 0) return 0; /* mesh blob missing from the checkout */
  S->type = t == 7 ? SH_HULL : t == 6 ? SH_BOX : SH_CAPSULE;
  S->p = gp; S->R = gR; S->size = m->geom_size[g];
  S->verts = m->mesh_vert + 3 * m->geom_vertadr[g];
  S->nvert = m->geom_vertnum[g];
  copy3(S->center, gp);
  if (t == 7) { mat_vec(gR, m->geom_center[g], S->center); for (int k = 0; k < 3; k++) S->center[k] += gp[k]; }
  return 1;
}
/* oriented bounding boxes (centre, axes = columns of R, half extents): 1 if one of the 15 candidate axes separates them
   (broad phase, as mj_collision's bounding-volume test: conservative) */
static int obb_disjoint(const double* Ra, const double* ca, const double* ha, const double* Rb, const double* cb, const double* hb) {
  double C[3][3], A[3][3], d[3], tv[3];
  sub3(cb, ca, d);
  matT_vec(Ra, d, tv);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      C[i][j] = Ra[i] * Rb[j] + Ra[3 + i] * Rb[3 + j] + Ra[6 + i] * Rb[6 + j];
      A[i][j] = fabs(C[i][j]) + 1e-9;
    }
  for (int i = 0; i < 3; i++)
    if (fabs(tv[i]) > ha[i] + hb[0] * A[i][0] + hb[1] * A[i][1] + hb[2] * A[i][2]) return 1;
  for (int j = 0; j < 3; j++) {
    const double tw = tv[0] * C[0][j] + tv[1] * C[1][j] + tv[2] * C[2][j];
    if (fabs(tw) > hb[j] + ha[0] * A[0][j] + ha[1] * A[1][j] + ha[2] * A[2][j]) return 1;
  }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      const double ra = ha[i1] * A[i2][j] + ha[i2] * A[i1][j], rb = hb[j1] * A[i][j2] + hb[j2] * A[i][j1];
      if (fabs(tv[i2] * C[i1][j] - tv[i1] * C[i2][j]) > ra + rb) return 1;
    }
  return 0;
}
int orc_dbg_mpr_calls = 0;
int orc_dbg_sphere_pass = 0, orc_dbg_pairs = 0, orc_dbg_skip_self = 0;
int orc_dbg_mpr_pairs[512];
/* geom-geom pairs of the robot (every convex pair through mjc_Convex's MPR: only "do they overlap" is read from it) */
/* The geom pairs MuJoCo's filters let collide are a property of the model: listed once per model (a process steps one
   model, or a few in turn; the key is the model's address plus a checksum of what the filters read). */
static const orc_model* g_pairs_model = 0;
static long g_pairs_sum = 0;
static int g_npairs = 0;
static short g_pairs[ORC_MAXGEOM * ORC_MAXGEOM / 2][2];
static long pair_checksum(const orc_model* m) {
  long s = m->ngeom * 131 + m->nbody;
  for (int g = 0; g < m->ngeom; g++) s = s * 31 + m->geom_bodyid[g] * 7 + m->geom_type[g] * 3 + m->geom_contype[g] + 2 * m->geom_conaffinity[g] + m->geom_vertnum[g];
  for (int b = 0; b < m->nbody; b++) s = s * 17 + m->body_weldid[b] + 5 * m->body_parentid[b];
  return s;
}
static void list_pairs(const orc_model* m) {
  g_npairs = 0;
  for (int g1 = 0; g1 < m->ngeom; g1++) {
    if (m->geom_type[g1] == 0) continue;
    for (int g2 = g1 + 1; g2 < m->ngeom; g2++) {
      if (m->geom_type[g2] == 0) continue;
      const int b1 = m->geom_bodyid[g1], b2 = m->geom_bodyid[g2];
      const int w1 = m->body_weldid[b1], w2 = m->body_weldid[b2];
      if (w1 == w2) continue;
      if (!((m->geom_contype[g1] & m->geom_conaffinity[g2]) || (m->geom_contype[g2] & m->geom_conaffinity[g1]))) continue;
      const int pw1 = m->body_weldid[m->body_parentid[w1]], pw2 = m->body_weldid[m->body_parentid[w2]];
      if (w1 && w2 && (w1 == pw2 || w2 == pw1)) continue;
      g_pairs[g_npairs][0] = (short)g1; g_pairs[g_npairs][1] = (short)g2;
      g_npairs++;
    }
  }
  g_pairs_model = m;
  g_pairs_sum = pair_checksum(m);
}
#define SELF_TOUCH 1e-9 /* csrc/check_team.h: kCheckTouch */
static void self_collide(const orc_model* m, orc_data* d) {
  d->nself = 0;
  if (orc_dbg_skip_self) return;
  if (g_pairs_model != m || g_pairs_sum != pair_checksum(m)) list_pairs(m);
  /* world frames and bounding-box centres of the geoms, once per position stage */
  double R[ORC_MAXGEOM][9], p[ORC_MAXGEOM][3], c[ORC_MAXGEOM][3], rad[ORC_MAXGEOM];
  for (int g = 0; g < m->ngeom; g++) {
    if (m->geom_type[g] == 0) continue;
    geom_frame(m, d, g, R[g], p[g]);
    mat_vec(R[g], m->geom_aabb[g], c[g]);
    for (int k = 0; k < 3; k++) c[g][k] += p[g][k];
    rad[g] = norm3(m->geom_aabb[g] + 3);
  }
  for (int i = 0; i < g_npairs; i++) {
    const int g1 = g_pairs[i][0], g2 = g_pairs[i][1];
    {
      double x[3];
      sub3(c[g1], c[g2], x);
      const double rs = rad[g1] + rad[g2];
      orc_dbg_pairs++;
      if (dot3(x, x) > rs * rs) continue;
      orc_dbg_sphere_pass++;
      if (obb_disjoint(R[g1], c[g1], m->geom_aabb[g1] + 3, R[g2], c[g2], m->geom_aabb[g2] + 3)) continue;
      /* MuJoCo orders a pair by geom type, then by id */
      const int swap = m->geom_type[g1] > m->geom_type[g2];
      const int ga = swap ? g2 : g1, gb = swap ? g1 : g2;
      shape A, B;
      if (!shape_of(m, ga, p[ga], R[ga], &A) || !shape_of(m, gb, p[gb], R[gb], &B)) continue;
      double depth, dir[3], pos[3];
      /* the remembered separating direction of this pair, if any: support of A - B along it still negative -> apart */
      int slot = -1;
      for (int k = 0; k < d->sep_n; k++) if (d->sep_pair[k][0] == ga && d->sep_pair[k][1] == gb) slot = k;
      if (slot >= 0) {
        double dw[3];
        mpr_pt q;
        mat_vec(R[ga], d->sep_dir[slot], dw);
        mpr_support(&A, &B, dw, &q);
        if (dot3(q.v, dw) < 0) continue;
      }
      if (orc_dbg_mpr_calls < 256) { orc_dbg_mpr_pairs[2 * orc_dbg_mpr_calls] = ga; orc_dbg_mpr_pairs[2 * orc_dbg_mpr_calls + 1] = gb; }
      orc_dbg_mpr_calls++;
      if ((m->resolve_contacts & 2) && m->geom_type[ga] == 6 && m->geom_type[gb] == 6) {
        /* two boxes (the fingertip pads of the two fingers): mjc_BoxBox, up to 8 points; the pair's place in d->contact is
           settled by sort_contacts (MuJoCo: by body pair, then by geom) */
        double bpos[24], bnrm[24], bdist[8];
        const int nc = orc_box_box(p[ga], R[ga], m->geom_size[ga], p[gb], R[gb], m->geom_size[gb], bpos, bnrm, bdist);
        const double mu = fmax(m->geom_friction[ga][0], m->geom_friction[gb][0]);
        depth = 0;
        /* (SELF_TOUCH: a point that touches exactly is no contact -- MuJoCo gives rows to dist < 0 only, and the sign of a distance of
           0.0 is round-off: the two fingers' pads meet with a gap of exactly 0 at finger qpos 0, i.e. after every reset) */
        for (int c = 0; c < nc; c++) {
          if (!(bdist[c] < -SELF_TOUCH)) continue;
          add_contact(d, ga, gb, m->geom_bodyid[ga], m->geom_bodyid[gb], bpos + 3 * c, bnrm + 3 * c, bdist[c], mu);
          depth = fmax(depth, -bdist[c]);
        }
        if (!(depth > 0)) continue;
        if (d->nself < ORC_MAXSELF) { d->self_geom[d->nself][0] = ga; d->self_geom[d->nself][1] = gb; d->self_depth[d->nself] = depth; d->nself++; }
        continue;
      }
      if (!mpr_penetration(&A, &B, &depth, dir, pos)) {
        if (dot3(dir, dir) > 0.5) { /* a unit separating direction came back */
          if (slot < 0) { slot = d->sep_n < ORC_MAXSEP ? d->sep_n++ : (ga + gb) % ORC_MAXSEP; d->sep_pair[slot][0] = ga; d->sep_pair[slot][1] = gb; }
          matT_vec(R[ga], dir, d->sep_dir[slot]);
        }
        continue;
      }
      if (d->nself < ORC_MAXSELF) { d->self_geom[d->nself][0] = ga; d->self_geom[d->nself][1] = gb; d->self_depth[d->nself] = depth; d->nself++; }
      /* self contacts resolved: the pair's contact (mjc_Convex: one point; normal from geom[0] to geom[1]) joins d->contact */
      if ((m->resolve_contacts & 2) && depth > SELF_TOUCH)
        add_contact(d, ga, gb, m->geom_bodyid[ga], m->geom_bodyid[gb], pos, dir, -depth, fmax(m->geom_friction[ga][0], m->geom_friction[gb][0]));
    }
  }
}

/* mjData.contact is ordered by body pair (lower body id first; mj_collision walks the broad phase's sorted body pairs), within a
   body pair by the geoms of the first body, then of the second.  The free box is the scene's last body and geom.  Contacts of one
   geom pair keep the order their collider produced them in (stable sort). */
static void contact_key(const orc_model* m, const orc_contact* c, long* key) {
  long b[2], g[2];
  for (int s = 0; s < 2; s++) {
    b[s] = c->body[s] == ORC_BODY_BOX ? m->nbody : c->body[s];
    g[s] = c->geom[s];
  }
  const int sw = b[0] > b[1] || (b[0] == b[1] && g[0] > g[1]);
  key[0] = b[sw]; key[1] = b[!sw]; key[2] = g[sw]; key[3] = g[!sw];
}
static void sort_contacts(const orc_model* m, orc_data* d) {
  for (int i = 1; i < d->ncon; i++) {
    orc_contact c = d->contact[i];
    long kc[4];
    contact_key(m, &c, kc);
    int j = i - 1;
    for (; j >= 0; j--) {
      long kj[4];
      contact_key(m, &d->contact[j], kj);
      int gt = 0;
      for (int q = 0; q < 4; q++) {
        if (kj[q] != kc[q]) { gt = kj[q] > kc[q]; break; }
      }
      if (!gt) break;
      d->contact[j + 1] = d->contact[j];
    }
    d->contact[j + 1] = c;
  }
  for (int i = 0; i < d->ncon; i++) { d->contact_geom[i][0] = d->contact[i].geom[0]; d->contact_geom[i][1] = d->contact[i].geom[1]; }
}

void orc_collide(const orc_model* m, orc_data* d) {
  d->ncon = 0;
  d->coupled = 0;
  geom_frames fr;
  for (int g = 0; g < m->ngeom; g++) geom_frame_compute(m, d, g, fr.R[g], fr.p[g]);
  g_frames = &fr;
  g_contact_cap = ORC_MAXCON;
  self_collide(m, d);
  const int gbox = m->ngeom;
  int robot_contacts = d->ncon; /* (self contacts, where they are resolved) */
  /* ---- floor plane against the robot's geoms */
  for (int pg = 0; pg < m->ngeom; pg++) {
    if (m->geom_type[pg] != 0) continue;
    const int pb = m->geom_bodyid[pg];
    double pR[9], pp[3];
    geom_frame(m, d, pg, pR, pp);
    const double n[3] = {pR[2], pR[5], pR[8]};
    for (int g = 0; g < m->ngeom; g++) {
      if (g == pg || m->geom_type[g] == 0) continue;
      const int b = m->geom_bodyid[g];
      if (m->body_weldid[b] == m->body_weldid[pb]) continue;
      if (!((m->geom_contype[g] & m->geom_conaffinity[pg]) || (m->geom_contype[pg] & m->geom_conaffinity[g]))) continue;
      {
        int w1 = m->body_weldid[pb], w2 = m->body_weldid[b];
        int pw1 = m->body_weldid[m->body_parentid[w1]], pw2 = m->body_weldid[m->body_parentid[w2]];
        if ((w1 && w1 == pw2) || (w2 && w2 == pw1)) continue;
      }
      double gR[9], gp[3], x[3];
      geom_frame(m, d, g, gR, gp);
      sub3(gp, pp, x);
      if (dot3(n, x) - geom_rbound(m, g) > 0) continue; /* broad phase */
      const double mu = fmax(m->geom_friction[pg][0], m->geom_friction[g][0]);
      const double* sz = m->geom_size[g];
      const int before = d->ncon;
      if (m->geom_type[g] == 7) {
        /* support vertex along -n, then up to three more along directions tilted about -n */
        const int nv = m->geom_vertnum[g];
        const double* V = m->mesh_vert + 3 * m->geom_vertadr[g];
        double fr[9] = {n[0], n[1], n[2]};
        make_frame(fr);
        int chosen[4], nch = 0;
        for (int q = 0; q < 4; q++) {
          double dir[3];
          if (q == 0) { dir[0] = -n[0]; dir[1] = -n[1]; dir[2] = -n[2]; }
          else {
            const double ang = 2 * M_PI * (q - 1) / 3, cs = 1e-3 * cos(ang), sn = 1e-3 * sin(ang);
            for (int k = 0; k < 3; k++) dir[k] = -n[k] + cs * fr[3 + k] + sn * fr[6 + k];
          }
          double dl[3];
          matT_vec(gR, dir, dl);
          const int bi = hull_support_index(V, nv, dl);
          if (bi < 0) break;
          int dup = 0;
          for (int k = 0; k < nch; k++) dup = dup || chosen[k] == bi;
          if (dup) continue;
          double w[3], xw[3];
          mat_vec(gR, V + 3 * bi, w);
          for (int k = 0; k < 3; k++) xw[k] = w[k] + gp[k];
          sub3(xw, pp, x);
          const double dist = dot3(n, x);
          if (dist >= 0) { if (q == 0) break; else continue; }
          chosen[nch++] = bi;
          double cpos[3] = {xw[0] - n[0] * dist * 0.5, xw[1] - n[1] * dist * 0.5, xw[2] - n[2] * dist * 0.5};
          add_contact(d, pg, g, pb, b, cpos, n, dist, mu);
        }
      } else if (m->geom_type[g] == 6) {
        /* mjc_PlaneBox: corners in corner order, at most four */
        int cnt = 0;
        sub3(gp, pp, x);
        const double cdist = dot3(n, x);
        for (int c = 0; c < 8 && cnt < 4; c++) {
          double loc[3] = {(c & 1 ? sz[0] : -sz[0]), (c & 2 ? sz[1] : -sz[1]), (c & 4 ? sz[2] : -sz[2])}, w[3];
          mat_vec(gR, loc, w);
          const double ld = dot3(n, w);
          if (cdist + ld > 0 || ld > 0) continue;
          const double dist = cdist + ld;
          double cpos[3] = {w[0] + gp[0] - n[0] * dist * 0.5, w[1] + gp[1] - n[1] * dist * 0.5, w[2] + gp[2] - n[2] * dist * 0.5};
          add_contact(d, pg, g, pb, b, cpos, n, dist, mu);
          cnt++;
        }
      } else if (m->geom_type[g] == 3 || m->geom_type[g] == 2) {
        /* mjc_PlaneCapsule: the two end spheres (a sphere: one) */
        for (int e = 0; e < (m->geom_type[g] == 3 ? 2 : 1); e++) {
          double loc[3] = {0, 0, m->geom_type[g] == 3 ? (e ? -sz[1] : sz[1]) : 0}, w[3], c[3];
          mat_vec(gR, loc, w);
          for (int k = 0; k < 3; k++) c[k] = w[k] + gp[k];
          sub3(c, pp, x);
          const double dist = dot3(n, x) - sz[0];
          if (dist >= 0) continue;
          double cpos[3] = {c[0] - n[0] * (sz[0] + dist * 0.5), c[1] - n[1] * (sz[0] + dist * 0.5), c[2] - n[2] * (sz[0] + dist * 0.5)};
          add_contact(d, pg, g, pb, b, cpos, n, dist, mu);
        }
      }
      robot_contacts += d->ncon - before;
    }
    /* ---- the floor against the free box: what orc_box_step1 found (mjc_PlaneBox) */
    if (m->box.present) {
      for (int c = 0; c < d->box.ncon; c++)
        add_contact(d, pg, gbox, pb, ORC_BODY_BOX, d->box.con_pos[c], n, d->box.con_dist[c], m->box.friction[0]);
    }
  }
  /* ---- the robot's geoms against the free box */
  if (m->box.present) {
    double bR[9];
    quat2mat(d->box.xquat, bR);
    const double* bp = d->box.xpos;
    const double* bs = m->box.size;
    const double brb = sqrt(dot3(bs, bs));
    for (int g = 0; g < m->ngeom; g++) {
      if (m->geom_type[g] == 0) continue;
      if (!(m->geom_contype[g] & 1) && !(m->geom_conaffinity[g] & 1)) continue; /* the box geom: contype = conaffinity = 1 */
      if (m->geom_type[g] == 7 && m->geom_vertnum[g] == 0) continue;            /* mesh blob missing from the checkout */
      const int b = m->geom_bodyid[g];
      double gR[9], gp[3], x[3];
      geom_frame(m, d, g, gR, gp);
      sub3(gp, bp, x);
      const double rsum = geom_rbound(m, g) + brb;
      if (dot3(x, x) > rsum * rsum) continue; /* broad phase: bounding spheres */
      const double mu = fmax(m->geom_friction[g][0], m->box.geom_friction[0]);
      const int before = d->ncon;
      if (m->geom_type[g] == 6) {
        /* same geom type: lower geom id first -> the robot's box is geom 1, the normal points from it to the free box */
        double pos[24], nrm[24], dist[8];
        const int nc = orc_box_box(gp, gR, m->geom_size[g], bp, bR, bs, pos, nrm, dist);
        for (int c = 0; c < nc; c++) add_contact(d, g, gbox, b, ORC_BODY_BOX, pos + 3 * c, nrm + 3 * c, dist[c], mu);
      } else {
        shape S = {m->geom_type[g] == 7 ? SH_HULL : SH_CAPSULE, gp, gR, m->geom_size[g], m->mesh_vert + 3 * m->geom_vertadr[g],
                   m->geom_vertnum[g], {gp[0], gp[1], gp[2]}};
        shape Bx = {SH_BOX, bp, bR, bs, 0, 0, {bp[0], bp[1], bp[2]}};
        if (S.type == SH_HULL) { mat_vec(gR, m->geom_center[g], S.center); for (int k = 0; k < 3; k++) S.center[k] += gp[k]; }
        double depth, dir[3], pos[3];
        if (m->geom_type[g] == 7) {
          /* MuJoCo orders a pair by geom type: box (6) before mesh (7) -> the free box is geom 1 */
          if (mpr_penetration(&Bx, &S, &depth, dir, pos)) add_contact(d, gbox, g, ORC_BODY_BOX, b, pos, dir, -depth, mu);
        } else {
          if (mpr_penetration(&S, &Bx, &depth, dir, pos)) add_contact(d, g, gbox, b, ORC_BODY_BOX, pos, dir, -depth, mu);
        }
      }
      robot_contacts += d->ncon - before;
    }
  }
  if (m->resolve_contacts & 2) sort_contacts(m, d);
  if (m->resolve_contacts && g_resolve_cap > 0 && d->ncon > g_resolve_cap) d->ncon = g_resolve_cap; /* (orc_set_contact_cap: tests only) */
  d->coupled = m->resolve_contacts && robot_contacts > 0;
  for (int i = 0; i < d->ncon; i++)
    if ((d->contact[i].geom[0] < m->ngeom && m->geom_type[d->contact[i].geom[0]] != 0) || (d->contact[i].geom[1] < m->ngeom && m->geom_type[d->contact[i].geom[1]] != 0))
      d->pen_seen = fmax(d->pen_seen, -d->contact[i].dist);
  for (int i = 0; i < d->nself; i++) d->pen_seen = fmax(d->pen_seen, d->self_depth[i]);
  g_frames = 0;
}

/* ------------------------------------------------------------------ rows of the coupled problem */

/* translational Jacobian row of a point on a body along direction f (robot bodies: over the joints; the free box: over
   its 6 dofs -- world linear velocity, body-frame angular velocity), ADDED to J with weight w */
static void add_point_jac(const orc_model* m, const orc_data* d, int body, const double* x, const double* f, double w, double* J) {
  if (body == ORC_BODY_BOX) {
    double R[9], r[3], rxf[3], loc[3];
    quat2mat(d->box.xquat, R);
    sub3(x, d->box.xpos, r);
    cross3(r, f, rxf);
    matT_vec(R, rxf, loc);
    for (int k = 0; k < 3; k++) { J[m->njnt + k] += w * f[k]; J[m->njnt + 3 + k] += w * loc[k]; }
    return;
  }
  int b = body;
  while (b > 0) {
    int j = m->body_jntadr[b];
    if (j >= 0) {
      double col[3];
      if (m->jnt_type[j] == ORC_JNT_SLIDE) copy3(col, d->xaxis[j]);
      else {
        double r[3];
        sub3(x, d->xanchor[j], r);
        cross3(d->xaxis[j], r, col);
      }
      J[j] += w * dot3(col, f);
    }
    b = m->body_parentid[b];
  }
}

/* mj_instantiateContact + mj_makeImpedance for the contacts of orc_collide: three rows per contact (elliptic cone,
   condim 3) appended behind the robot's equality / friction / limit rows */
void orc_make_coupled_rows(const orc_model* m, orc_data* d) {
  int n = d->nefc;
  const double* solref = m->box.solref;
  const double* solimp = m->box.solimp;
  double tc = solref[0], dr = solref[1];
  if (tc < 2 * m->timestep) tc = 2 * m->timestep;
  double dmax = solimp[1];
  if (dmax < 0.0001) dmax = 0.0001; if (dmax > 0.9999) dmax = 0.9999;
  const double K = 1 / (dmax * dmax * tc * tc * dr * dr), B = 2 / (dmax * tc);
  for (int c = 0; c < d->ncon; c++) {
    orc_contact* con = &d->contact[c];
    con->efc_address = n;
    double iw = 0;
    for (int s = 0; s < 2; s++) iw += con->body[s] == ORC_BODY_BOX ? 1 / m->box.mass : m->body_invweight0[con->body[s]];
    const double imp = orc_impedance(solimp, con->dist, 0);
    double R0 = (1 - imp) / imp * iw;
    if (R0 < MINVAL) R0 = MINVAL;
    const double R1 = R0 / (m->box.impratio > MINVAL ? m->box.impratio : MINVAL);
    for (int k = 0; k < 3; k++) {
      memset(d->efc_J[n + k], 0, sizeof(d->efc_J[n + k]));
      add_point_jac(m, d, con->body[1], con->pos, con->frame + 3 * k, 1.0, d->efc_J[n + k]);
      add_point_jac(m, d, con->body[0], con->pos, con->frame + 3 * k, -1.0, d->efc_J[n + k]);
      d->efc_type[n + k] = k == 0 ? ORC_EFC_CONTACT : ORC_EFC_CONTACT_T;
      d->efc_pos[n + k] = k == 0 ? con->dist : 0;
      d->efc_margin[n + k] = 0;
      d->efc_K[n + k] = k == 0 ? K : 0; /* friction rows: zero position, only the damping term survives */
      d->efc_B[n + k] = B;
      d->efc_I[n + k] = imp;
      d->efc_R[n + k] = k == 0 ? R0 : R1; /* condim 3: both friction coefficients equal, R2 = R1 */
      d->efc_D[n + k] = 1 / d->efc_R[n + k];
      d->efc_frictionloss[n + k] = 0;
      d->efc_mu[n + k] = k == 0 ? con->mu * sqrt(R1 / R0) : con->mu; /* normal row: regularised mu; friction rows: the pair's coefficient */
    }
    n += 3;
  }
  d->nefc = n;
}

/* ------------------------------------------------------------------ the coupled solve */

typedef struct {
  int nv, nr; /* total dofs, robot dofs */
  double M[ORC_NVT][ORC_NVT], a0[ORC_NVT], f0[ORC_NVT];
} csys;

static int chol_n(double L[ORC_NVT][ORC_NVT], int n) {
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
static void chol_solve_n(const double L[ORC_NVT][ORC_NVT], int n, double* x) {
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

/* elliptic cone of one contact (mj_constraintUpdate): cost, force f = -dcost/djar, Hessian; zones 0 top / 1 middle / 2 bottom */
static double cone_cost(const double* D, double mu, double fr, const double* jar, double* f, double Hc[3][3], int* zone) {
  const double U[3] = {jar[0] * mu, jar[1] * fr, jar[2] * fr};
  const double N = U[0], T = sqrt(U[1] * U[1] + U[2] * U[2]);
  memset(Hc, 0, 9 * sizeof(double));
  f[0] = f[1] = f[2] = 0;
  if (N >= mu * T) { *zone = 0; return 0; }
  if (mu * N + T <= 0) {
    *zone = 2;
    double cost = 0;
    for (int k = 0; k < 3; k++) { cost += 0.5 * D[k] * jar[k] * jar[k]; f[k] = -D[k] * jar[k]; Hc[k][k] = D[k]; }
    return cost;
  }
  *zone = 1;
  const double s[3] = {mu, fr, fr};
  const double Dm = D[0] / (mu * mu * (1 + mu * mu)), NmT = N - mu * T;
  const double u[2] = {U[1] / T, U[2] / T};
  const double gU[3] = {Dm * NmT, -Dm * NmT * mu * u[0], -Dm * NmT * mu * u[1]};
  for (int k = 0; k < 3; k++) f[k] = -s[k] * gU[k];
  double HU[3][3];
  HU[0][0] = Dm;
  for (int j = 0; j < 2; j++) {
    HU[0][1 + j] = HU[1 + j][0] = -Dm * mu * u[j];
    for (int k = 0; k < 2; k++) HU[1 + j][1 + k] = Dm * mu * mu * u[j] * u[k] - Dm * NmT * mu * ((j == k ? 1.0 : 0.0) - u[j] * u[k]) / T;
  }
  for (int j = 0; j < 3; j++)
    for (int k = 0; k < 3; k++) Hc[j][k] = s[j] * s[k] * HU[j][k];
  return 0.5 * Dm * NmT * NmT;
}

/* primal cost at x: Gauss term + every row's cost; optionally gradient and Hessian; forces left in d->efc_force */
static double primal(orc_data* d, const csys* S, const double* x, double* grad, double H[ORC_NVT][ORC_NVT]) {
  const int nv = S->nv;
  double cost = 0;
  for (int r = 0; r < nv; r++) {
    double g = -S->f0[r];
    for (int c = 0; c < nv; c++) g += S->M[r][c] * x[c];
    if (grad) grad[r] = g;
    cost += 0.5 * (x[r] - S->a0[r]) * g;
  }
  if (H) memcpy(H, S->M, sizeof(S->M));
  for (int i = 0; i < d->nefc;) {
    const int type = d->efc_type[i];
    if (type == ORC_EFC_CONTACT) {
      double jar[3], f[3], Hc[3][3];
      int zone;
      for (int k = 0; k < 3; k++) {
        double s = -d->efc_aref[i + k];
        for (int j = 0; j < nv; j++) s += d->efc_J[i + k][j] * x[j];
        jar[k] = s;
      }
      cost += cone_cost(d->efc_D + i, d->efc_mu[i], d->efc_mu[i + 1], jar, f, Hc, &zone);
      for (int k = 0; k < 3; k++) {
        d->efc_force[i + k] = f[k];
        if (grad) for (int j = 0; j < nv; j++) grad[j] -= d->efc_J[i + k][j] * f[k];
      }
      if (H && zone)
        for (int k = 0; k < 3; k++)
          for (int l = 0; l < 3; l++) {
            if (Hc[k][l] == 0) continue;
            for (int r = 0; r < nv; r++) {
              const double a = d->efc_J[i + k][r] * Hc[k][l];
              if (a == 0) continue;
              for (int c = 0; c < nv; c++) H[r][c] += a * d->efc_J[i + l][c];
            }
          }
      for (int c = 0; c < d->ncon; c++) if (d->contact[c].efc_address == i) d->contact[c].zone = zone;
      i += 3;
      continue;
    }
    double jar = -d->efc_aref[i];
    for (int j = 0; j < nv; j++) jar += d->efc_J[i][j] * x[j];
    double f = 0, hd = 0;
    if (type == ORC_EFC_EQUALITY || (type == ORC_EFC_LIMIT && jar < 0)) {
      cost += 0.5 * d->efc_D[i] * jar * jar; f = -d->efc_D[i] * jar; hd = d->efc_D[i];
    } else if (type == ORC_EFC_FRICTION) {
      const double fl = d->efc_frictionloss[i], rf = fl / d->efc_D[i];
      if (jar <= -rf) { cost += -0.5 * rf * fl - fl * jar; f = fl; }
      else if (jar >= rf) { cost += -0.5 * rf * fl + fl * jar; f = -fl; }
      else { cost += 0.5 * d->efc_D[i] * jar * jar; f = -d->efc_D[i] * jar; hd = d->efc_D[i]; }
    }
    d->efc_force[i] = f;
    if (f != 0 && grad) for (int j = 0; j < nv; j++) grad[j] -= d->efc_J[i][j] * f;
    if (hd != 0 && H)
      for (int r = 0; r < nv; r++) {
        const double a = d->efc_J[i][r] * hd;
        if (a == 0) continue;
        for (int c = 0; c < nv; c++) H[r][c] += a * d->efc_J[i][c];
      }
    i++;
  }
  return cost;
}

#define ORC_NEWTON_REL 2e-12 /* csrc/contact_team.h: kNewtonRel */
long long orc_newton_stats[4]; /* solves, iterations, capped at 100, over 20 (development statistics) */
/* mju_QCQP2 (as in rcs_object.c) */
static int qcqp2(double* res, const double* Ain, const double* bin, const double* dd, double r) {
  double b1 = bin[0] * dd[0], b2 = bin[1] * dd[1];
  double A11 = Ain[0] * dd[0] * dd[0], A22 = Ain[3] * dd[1] * dd[1], A12 = Ain[1] * dd[0] * dd[1];
  double la = 0, v1 = 0, v2 = 0;
  for (int iter = 0; iter < 20; iter++) {
    double det = (A11 + la) * (A22 + la) - A12 * A12;
    if (det < 1e-10) { res[0] = res[1] = 0; return 0; }
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

/* mj_fwdConstraint for the coupled system: Newton on the primal cost to its minimiser, then mj_solNoSlip.
   In: d->qfrc_smooth (robot), d->box.qfrc_smooth / qacc_smooth, the rows.  Out: d->qacc, d->box.qacc, efc_force,
   d->qfrc_constraint (robot dofs, then the box's). */
void orc_solve_coupled(const orc_model* m, orc_data* d) {
  static csys S; /* (single-threaded test infrastructure) */
  static double H[ORC_NVT][ORC_NVT], Hq[ORC_NVT][ORC_NVT];
  const int nr = m->njnt, nb = m->box.present ? 6 : 0, nv = nr + nb;
  S.nv = nv; S.nr = nr;
  memset(S.M, 0, sizeof(S.M));
  for (int r = 0; r < nr; r++)
    for (int c = 0; c < nr; c++) S.M[r][c] = d->qM[r][c];
  const double Mb[6] = {m->box.mass, m->box.mass, m->box.mass, m->box.inertia[0], m->box.inertia[1], m->box.inertia[2]};
  for (int k = 0; k < nb; k++) S.M[nr + k][nr + k] = Mb[k];
  /* qacc_smooth = M^-1 qfrc_smooth */
  static double L[ORC_NVT][ORC_NVT];
  memcpy(L, S.M, sizeof(L));
  chol_n(L, nv);
  for (int r = 0; r < nr; r++) S.f0[r] = d->qfrc_smooth[r];
  for (int k = 0; k < nb; k++) S.f0[nr + k] = d->box.qfrc_smooth[k];
  memcpy(S.a0, S.f0, sizeof(S.a0));
  chol_solve_n(L, nv, S.a0);
  for (int r = 0; r < nr; r++) d->qacc_smooth[r] = S.a0[r];
  /* warm start: the cheaper of qacc_warmstart and qacc_smooth */
  double x[ORC_NVT], xw[ORC_NVT], grad[ORC_NVT], p[ORC_NVT], gq[ORC_NVT];
  for (int r = 0; r < nr; r++) xw[r] = d->qacc_warmstart[r];
  for (int k = 0; k < nb; k++) xw[nr + k] = d->box.qacc_warmstart[k];
  memcpy(x, S.a0, sizeof(x));
  if (primal(d, &S, xw, 0, 0) < primal(d, &S, x, 0, 0)) memcpy(x, xw, sizeof(x));
  const double scale = 1 / (m->box.meaninertia * (nv > 1 ? nv : 1));
  int it = 0;
  static double trace_g[100], trace_a[100], trace_c[100];
  for (; it < 100; it++) {
    const double cost_it = primal(d, &S, x, grad, H);
    double g2 = 0, q2 = 0;
    for (int j = 0; j < nv; j++) g2 += grad[j] * grad[j];
    /* Converged: the gradient below the absolute bar -- or at its round-off floor.  The gradient is the difference of the Gauss term
       and the contacts' generalised force, each of the size of that force: with a contact force of 1e3 N its floor is 1e-13 of
       that, above the absolute bar, and the loop would spin at a fixed point until its cap (2 % of the headline workload's solves,
       measured round 5: tools/newton_cap_probe.py; the relative gradient falls from O(1) to <= 6.4e-13 in ONE step once the zones
       are right).  q2: |J' f|^2 over the contact rows. */
    for (int j = 0; j < nv; j++) {
      double q = 0;
      for (int i = 0; i < d->nefc; i++) if (d->efc_type[i] == ORC_EFC_CONTACT) q += d->efc_J[i][j] * d->efc_force[i];
      q2 += q * q;
    }
    trace_g[it] = scale * sqrt(g2); trace_c[it] = cost_it; trace_a[it] = sqrt(g2) / (sqrt(q2) + 1e-300);
    if (scale * sqrt(g2) < 1e-12 || g2 <= ORC_NEWTON_REL * ORC_NEWTON_REL * q2) break;
    for (int j = 0; j < nv; j++) p[j] = -grad[j];
    if (chol_n(H, nv)) break;
    chol_solve_n(H, nv, p);
    double dphi0 = 0;
    for (int j = 0; j < nv; j++) dphi0 += grad[j] * p[j];
    if (!(dphi0 < 0)) break;
    /* line search: root of phi'(a) by safeguarded 1-D Newton; a = 1 is exact while no row changes zone */
    double lo = 0, hi = -1, a = 1, best = 1, dx = 1e300, dxold = 1e300;
    for (int ls = 0; ls < 30; ls++) {
      double xa[ORC_NVT];
      for (int j = 0; j < nv; j++) xa[j] = x[j] + a * p[j];
      primal(d, &S, xa, gq, Hq);
      double dphi = 0, ddphi = 0;
      for (int j = 0; j < nv; j++) {
        dphi += gq[j] * p[j];
        for (int k = 0; k < nv; k++) ddphi += p[j] * Hq[j][k] * p[k];
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
    /* ... or the step no longer moves the iterate (a fixed point of the iteration in floating point) */
    int moved = 0;
    for (int j = 0; j < nv; j++) { const double xn = x[j] + best * p[j]; moved |= xn != x[j]; x[j] = xn; }
    if (!moved) { it++; break; }
  }
  orc_newton_stats[0] += 1; orc_newton_stats[1] += it; if (it >= 100) orc_newton_stats[2] += 1; if (it > 20) orc_newton_stats[3] += 1;
  if (it >= 100 && getenv("ORC_NEWTON_TRACE")) {
    fprintf(stderr, "newton capped: ncon %d nefc %d\n", d->ncon, d->nefc);
    for (int k = 0; k < 100; k += (k < 20 ? 1 : 10)) fprintf(stderr, "   it %d  scale|g| %.3e  cost %.17g  |g|/|qfrc_contact| %.3e\n", k, trace_g[k], trace_c[k], trace_a[k]);
  }
  d->solver_niter = it;
  primal(d, &S, x, 0, 0); /* forces (and cone zones) at the solution */
  const int ne = d->nefc;
  d->noslip_niter = 0;
  int nfric = 0;
  for (int i = 0; i < ne; i++) nfric += d->efc_type[i] == ORC_EFC_CONTACT;
  if (m->box.noslip_iterations > 0 && nfric > 0) {
    /* ---- mj_solNoSlip: dual Gauss-Seidel over the contacts' friction rows WITHOUT regularisation; every row of the scene
       is a row of A = J M^-1 J', only the friction rows' forces move */
    double* A = (double*)malloc(sizeof(double) * ne * ne);
    double* MiJ = (double*)malloc(sizeof(double) * ne * nv);
    double* bb = (double*)malloc(sizeof(double) * ne);
    double* force = d->efc_force;
    for (int i = 0; i < ne; i++) {
      double col[ORC_NVT];
      for (int j = 0; j < nv; j++) col[j] = d->efc_J[i][j];
      chol_solve_n(L, nv, col);
      for (int j = 0; j < nv; j++) MiJ[i * nv + j] = col[j];
      double s = -d->efc_aref[i];
      for (int j = 0; j < nv; j++) s += d->efc_J[i][j] * S.a0[j];
      bb[i] = s;
    }
    for (int i = 0; i < ne; i++)
      for (int k = 0; k < ne; k++) {
        double s = 0;
        for (int j = 0; j < nv; j++) s += d->efc_J[i][j] * MiJ[k * nv + j];
        A[i * ne + k] = s;
      }
    int iter = 0;
    while (iter < m->box.noslip_iterations) {
      double improvement = 0;
      if (iter == 0)
        for (int i = 0; i < ne; i++)
          if (d->efc_type[i] != ORC_EFC_EQUALITY) improvement += 0.5 * force[i] * force[i] / d->efc_D[i];
      for (int i = 0; i < ne; i++) {
        if (d->efc_type[i] != ORC_EFC_CONTACT) continue;
        const double fr[2] = {d->efc_mu[i + 1], d->efc_mu[i + 2]};
        double res[3], old[3];
        for (int k = 0; k < 3; k++) {
          double s = bb[i + k];
          for (int j = 0; j < ne; j++) s += A[(i + k) * ne + j] * force[j];
          res[k] = s;
          old[k] = force[i + k];
        }
        if (force[i] < MINVAL) {
          force[i] = force[i + 1] = force[i + 2] = 0;
        } else {
          double Ac[4] = {A[(i + 1) * ne + i + 1], A[(i + 1) * ne + i + 2], A[(i + 2) * ne + i + 1], A[(i + 2) * ne + i + 2]};
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
        double dl[3] = {force[i] - old[0], force[i + 1] - old[1], force[i + 2] - old[2]}, change = 0;
        for (int k = 0; k < 3; k++) {
          for (int l = 0; l < 3; l++) change += 0.5 * dl[k] * A[(i + k) * ne + i + l] * dl[l];
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
      if (improvement < m->box.noslip_tolerance) break;
    }
    d->noslip_niter = iter;
    free(A); free(MiJ); free(bb);
    /* dualFinish: qacc = qacc_smooth + M^-1 J' force */
    double q[ORC_NVT];
    memset(q, 0, sizeof(q));
    for (int i = 0; i < ne; i++)
      for (int j = 0; j < nv; j++) q[j] += d->efc_J[i][j] * force[i];
    chol_solve_n(L, nv, q);
    for (int j = 0; j < nv; j++) x[j] = S.a0[j] + q[j];
  }
  memset(d->qfrc_constraint, 0, sizeof(d->qfrc_constraint));
  for (int i = 0; i < ne; i++)
    for (int j = 0; j < nv; j++) d->qfrc_constraint[j] += d->efc_J[i][j] * d->efc_force[i];
  for (int r = 0; r < nr; r++) d->qacc[r] = x[r];
  for (int k = 0; k < nb; k++) d->box.qacc[k] = x[nr + k];
}
