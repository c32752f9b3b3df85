// contact_team.h -- contacts of the robot's collision geoms (with the floor, with the free box) and the constraint
// problem that couples the robot's joints with the box's six degrees of freedom: what makes a grasp hold.
// Reference: finger pads assets/fr3/mjcf/fr3_0.xml:145-162 against the cube assets/scenes/fr3_simple_pick_up/scene.xml:30-33,
// read by SimGripper::collision_callback / SimRobot::collision_callback (src/sim/SimGripper.cpp:108-130,
// src/sim/SimRobot.cpp:172-182) and PickCubeSuccessWrapper (python/rcs/envs/sim.py:396-431).  Physics: MuJoCo's
// collision + soft-constraint pipeline for these scenes (box-box by separating axes + face clipping, convex pairs by
// Minkowski portal refinement, elliptic cones, Newton on the primal cost, the noslip pass), restated -- see DESIGN.md.
//
// THE WAVE GANGS UP ON ONE ENVIRONMENT.  A team's 16 lanes own an environment in the fast path (dyn_team.h); contacts
// are rare (a gripper near the cube, an arm on the floor) and their working set -- up to 48 contacts of 3 rows, a
// 15 x 15 Hessian -- does not fit four times into the LDS a workgroup may use without evicting its neighbours from the
// CU.  So when a team's broad phase fires, ALL 64 lanes of the wavefront work on that one environment, teams taking
// turns:
//   * collision: lane g tests collision geom g (pad boxes by SAT + clipping, hulls / the capsule by MPR, everything
//     against the floor plane), contacts are compacted into MuJoCo's order with a prefix over the lanes;
//   * lane c then OWNS contact c for the rest of the solve: its frame, its 3 x 6 map G_c from spatial motion about the
//     world origin to contact-frame velocity, reference accelerations, regularisers and force live in its registers;
//   * the contact Jacobian is never formed.  J_c = G_c (S_B - S_A) with S_b the 6 x nv spatial Jacobian of body b, so
//     J x = G_c (U_B - U_A) needs the bodies' spatial accelerations U_b = S_b x (9 links + the box: one lane each, the
//     robot's by summing motion axes down the chain), J' f is a wrench per body pushed through the same leaf-to-root
//     sums as the bias forces, and J' D J is a 6 x 6 contact stiffness per body PAIR folded into the mass matrix's own
//     recursion (composite stiffness next to composite inertia): H = M + S'(K) S;
//   * the 15 x 15 Hessian is assembled in LDS by 15 lanes, then every lane factors it (LDL', fully unrolled, registers)
//     and solves -- redundancy is free, the instructions are issued for the wave anyway;
//   * the noslip pass is Gauss-Seidel over the contacts in order: the owner lane solves its 2 x 2 friction QCQP, the
//     change of its wrench is pushed into the bodies' accelerations through Y_b = M^-1 S_b' (held for the few links in
//     contact), 9 + 1 lanes updating one body each.
#pragma once
#include "box_team.h"
#include "contact_types.h"

namespace rcsh {

#if defined(__HIP__)


// The contact phase is split into three non-inlined functions (collision, Newton, noslip) that hand their state over in
// LDS: each gets a register allocation of its own instead of one that must hold everything at once.
#define RCSH_CONTACT_FN __device__ __noinline__

// ---- wave-level helpers (64 lanes)
RCSH_D double wave_sum(double x) {
  x = quad_sum(x);
  x += row_rotate<4>(x);
  x += row_rotate<8>(x);
  return (wave_read(x, 0) + wave_read(x, 16)) + (wave_read(x, 32) + wave_read(x, 48));
}
RCSH_D int wave_lane() { return threadIdx.x & 63; }

// CAP: contacts per environment this arena has records for (contact_types.h: kMaxCon with a free box in the scene -- that kernel's LDS
// must let four workgroups share a CU --, kMaxConNoBox without one: the kernel of per-environment escalation)
template <class T, int CAP = kMaxCon>
struct ContactArena {
  static constexpr int kCap = CAP;
  static constexpr int NL = T::NL, NB = T::NL + 2, NV = T::NL + 6;
  static constexpr int kBox = NL, kWorld = NL + 1;
  double V[NB][6];         // spatial velocity of the bodies about the world origin [angular; linear]
  double U[NB][6];         // spatial acceleration for the current iterate
  double Up[NB][6];        // ... for the search direction
  double W[NB][6];         // wrench on every body
  double X[16], A0[16], P[16], Gd[16];
  double H[NV * (NV + 1) / 2];
  static constexpr int kAcc = 2 * kMaxActive + 1 + kMaxPairs;
  double KA[kAcc][21];     // contact stiffness: (link a, box) pairs, (world, link a) pairs, (world, box), (link, link) pairs
  double rec[CAP][14];     // contact records (con_load): what the phases hand to each other
  double stage[64][8];     // scratch: clipping polygons of the box-box collider / per-contact wrenches / stiffness batches / Y
  int32_t cb[64];          // bodies of contact c: A | B << 8 | class bits << 16
  int32_t cnt[64][2];      // per geom lane (a geom table has at most kMaxCGeom <= 32 entries): plane contacts, box contacts; the upper half
                           // holds the contacts' sort keys (keyp)
  int32_t act[kMaxActive + 1]; // links in contact
  int32_t pairs[kMaxPairs];    // pairs of links in contact with each other (self contact): la | lb << 8, la < lb
  int32_t nact, ncon, pad[2];
  int32_t npairs, pad3;
  // what the collision phase may use as scratch (nothing of the solve is alive then): V .. KA, contiguous
  // ... except its last 12 NL doubles: the world frames of the links, R(9) p(3), which only the collision phase reads
  static constexpr int kScratch = 4 * NB * 6 + 4 * 16 + NV * (NV + 1) / 2 + kAcc * 21 - 12 * NL;
  __device__ double* scratch() { return &V[0][0]; }
  __device__ double (*frames())[12] { return reinterpret_cast<double (*)[12]>(&V[0][0] + kScratch); }
  // contact c's place in mjData.contact: body pair, geom pair, index within the pair (contact_collide's merge)
  __device__ int32_t* keyp() { return &cnt[32][0]; }
  static_assert(kMaxCGeom <= 32 && CAP <= 64, "sort keys share the counters' array; a lane per contact");
};

// is joint j an ancestor-or-self joint of link i?
template <class T>
RCSH_D bool is_anc(int j, int i) {
  if (T::GRIP && i == T::NARM + 1 && j == T::NARM) return false;
  return j <= i;
}

// bit j: joint j is an ancestor-or-self joint of link i (i < 0, welded to the world: none)
template <class T>
RCSH_D uint32_t anc_mask(int i) {
  if (i < 0) return 0u;
  uint32_t m = (2u << i) - 1u;
  if (T::GRIP && i == T::NARM + 1) m &= ~(1u << T::NARM);
  return m;
}

// mju_makeFrame
RCSH_D void make_frame(const double* n, double* t1, double* t2) {
  double y[3] = {0, 0, 0};
  if (n[1] > -0.5 && n[1] < 0.5) y[1] = 1; else y[2] = 1;
  const double s = dot3(n, y);
#pragma unroll
  for (int k = 0; k < 3; ++k) y[k] -= s * n[k];
  const double il = 1.0 / sqrt(dot3(y, y));
#pragma unroll
  for (int k = 0; k < 3; ++k) t1[k] = y[k] * il;
  cross3(n, t1, t2);
}
RCSH_D void mulTv(const double* R, const double* v, double* o) {
  const double x = R[0] * v[0] + R[3] * v[1] + R[6] * v[2], y = R[1] * v[0] + R[4] * v[1] + R[7] * v[2], z = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}

// The smallest penetration of two boxes along their six face axes -- dev_box_box's own first six tests, same expressions: negative:
// apart.  Every contact point dev_box_box returns is at most this deep (face case: depths are measured along the face axis of
// that smallest penetration; edge case: chosen only when its penetration is smaller still), so two boxes whose faces just touch --
// the two fingers' pads of a closed gripper, in every substep -- are settled by their lane without the serial collider.
RCSH_D double box_box_face_pen(const double* p1, const double* R1, const double* s1, const double* p2, const double* R2, const double* s2) {
  double d[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]}, t[3], R[3][3], Q[3][3];
  mulTv(R1, d, t);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      R[i][j] = R1[i] * R2[j] + R1[3 + i] * R2[3 + j] + R1[6 + i] * R2[6 + j];
      Q[i][j] = fabs(R[i][j]);
    }
  double best = INFINITY;
  for (int i = 0; i < 3; ++i) {
    const double pen = s1[i] + s2[0] * Q[i][0] + s2[1] * Q[i][1] + s2[2] * Q[i][2] - fabs(t[i]);
    best = pen < best ? pen : best;
  }
  for (int j = 0; j < 3; ++j) {
    const double tb = t[0] * R[0][j] + t[1] * R[1][j] + t[2] * R[2][j];
    const double pen = s2[j] + s1[0] * Q[0][j] + s1[1] * Q[1][j] + s1[2] * Q[2][j] - fabs(tb);
    best = pen < best ? pen : best;
  }
  return best;
}

// ------------------------------------------------------------------ box - box (oracle: orc_box_box)
RCSH_CONTACT_FN int dev_box_box(const double* p1, const double* R1, const double* s1, const double* p2, const double* R2, const double* s2,
                                       double* pos /* [8][3] */, double* nrm /* [3] */, double* dist /* [8] */,
                                       double* poly_lds /* 48 doubles of LDS for the clipping polygons; null: overlap test only (returns 0 / 1) */) {
  double d[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]}, t[3], R[3][3], Q[3][3];
  mulTv(R1, d, t);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      R[i][j] = R1[i] * R2[j] + R1[3 + i] * R2[3 + j] + R1[6 + i] * R2[6 + j];
      Q[i][j] = fabs(R[i][j]);
    }
  double best = INFINITY;
  int code = -1;
  for (int i = 0; i < 3; ++i) {
    const double pen = s1[i] + s2[0] * Q[i][0] + s2[1] * Q[i][1] + s2[2] * Q[i][2] - fabs(t[i]);
    if (pen < 0) return 0;
    if (pen < best) { best = pen; code = i; }
  }
  for (int j = 0; j < 3; ++j) {
    const double tb = t[0] * R[0][j] + t[1] * R[1][j] + t[2] * R[2][j];
    const double pen = s2[j] + s1[0] * Q[0][j] + s1[1] * Q[1][j] + s1[2] * Q[2][j] - fabs(tb);
    if (pen < 0) return 0;
    if (pen < best) { best = pen; code = 3 + j; }
  }
  double ebest = INFINITY;
  int ecode = -1;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const double l2 = 1 - R[i][j] * R[i][j];
      if (l2 < 1e-6) continue;
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      const double ra = s1[i1] * Q[i2][j] + s1[i2] * Q[i1][j];
      const double rb = s2[j1] * Q[i][j2] + s2[j2] * Q[i][j1];
      const double tl = fabs(t[i2] * R[i1][j] - t[i1] * R[i2][j]);
      const double pen = (ra + rb - tl) / sqrt(l2);
      if (pen < 0) return 0;
      if (pen < ebest) { ebest = pen; ecode = 3 * i + j; }
    }
  if (!poly_lds) return 1;
  if (ecode >= 0 && ebest * 1.05 < best) {
    const int i = ecode / 3, j = ecode % 3;
    const double A[3] = {R1[i], R1[3 + i], R1[6 + i]}, B[3] = {R2[j], R2[3 + j], R2[6 + j]};
    double n[3];
    cross3(A, B, n);
    const double l = sqrt(dot3(n, n));
    for (int k = 0; k < 3; ++k) n[k] /= l;
    if (dot3(n, d) < 0) for (int k = 0; k < 3; ++k) n[k] = -n[k];
    double pa[3] = {p1[0], p1[1], p1[2]}, pb[3] = {p2[0], p2[1], p2[2]};
    for (int k = 0; k < 3; ++k) {
      if (k != i) {
        const double Ak[3] = {R1[k], R1[3 + k], R1[6 + k]};
        const double sg = dot3(n, Ak) > 0 ? s1[k] : -s1[k];
        for (int c = 0; c < 3; ++c) pa[c] += sg * Ak[c];
      }
      if (k != j) {
        const double Bk[3] = {R2[k], R2[3 + k], R2[6 + k]};
        const double sg = dot3(n, Bk) > 0 ? -s2[k] : s2[k];
        for (int c = 0; c < 3; ++c) pb[c] += sg * Bk[c];
      }
    }
    const double p[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
    const double uaub = dot3(A, B), q1 = dot3(A, p), q2 = -dot3(B, p), dd = 1 - uaub * uaub;
    double al = 0, be = 0;
    if (dd > 1e-4) { al = (q1 + uaub * q2) / dd; be = (uaub * q1 + q2) / dd; }
    for (int c = 0; c < 3; ++c) {
      pos[c] = 0.5 * ((pa[c] + al * A[c]) + (pb[c] + be * B[c]));
      nrm[c] = n[c];
    }
    dist[0] = -ebest;
    return 1;
  }
  const bool ref1 = code < 3;
  const int a = ref1 ? code : code - 3;
  const double *Rr = ref1 ? R1 : R2, *pr = ref1 ? p1 : p2, *sr = ref1 ? s1 : s2;
  const double *Ri = ref1 ? R2 : R1, *pi = ref1 ? p2 : p1, *si = ref1 ? s2 : s1;
  const double ci[3] = {pi[0] - pr[0], pi[1] - pr[1], pi[2] - pr[2]};
  const double Ar[3] = {Rr[a], Rr[3 + a], Rr[6 + a]};
  const double sgn = dot3(ci, Ar) >= 0 ? 1.0 : -1.0;
  const double n[3] = {sgn * Ar[0], sgn * Ar[1], sgn * Ar[2]};
  int b = 0;
  double bestdot = -1;
  for (int k = 0; k < 3; ++k) {
    const double Bk[3] = {Ri[k], Ri[3 + k], Ri[6 + k]};
    const double v = fabs(dot3(n, Bk));
    if (v > bestdot) { bestdot = v; b = k; }
  }
  const double Bb[3] = {Ri[b], Ri[3 + b], Ri[6 + b]};
  const double sb = dot3(n, Bb) > 0 ? -si[b] : si[b];
  const int u = (b + 1) % 3, v = (b + 2) % 3, a1 = (a + 1) % 3, a2 = (a + 2) % 3;
  const double Bu[3] = {Ri[u], Ri[3 + u], Ri[6 + u]}, Bv[3] = {Ri[v], Ri[3 + v], Ri[6 + v]};
  double (*poly)[8][3] = reinterpret_cast<double (*)[8][3]>(in_lds(poly_lds));  // (at most 8 vertices: a quad clipped by 4 half planes)
  for (int q = 0; q < 4; ++q) {
    const double su = (q == 0 || q == 3) ? 1.0 : -1.0, sv = q < 2 ? 1.0 : -1.0;
    double w[3];
    for (int c = 0; c < 3; ++c) w[c] = pi[c] + sb * Bb[c] + su * si[u] * Bu[c] + sv * si[v] * Bv[c] - pr[c];
    mulTv(Rr, w, poly[0][q]);
  }
  int np = 4, cur = 0;
  for (int pass = 0; pass < 4; ++pass) {
    const int axis = pass < 2 ? a1 : a2;
    const double sign = (pass & 1) ? -1.0 : 1.0, lim = sr[axis];
    int m = 0;
    for (int i = 0; i < np; ++i) {
      const double* pa_ = poly[cur][i];
      const double* pb_ = poly[cur][(i + 1) % np];
      const double da = sign * pa_[axis] - lim, db = sign * pb_[axis] - lim;
      if (da <= 0) { for (int c = 0; c < 3; ++c) poly[cur ^ 1][m][c] = pa_[c]; ++m; }
      if ((da < 0 && db > 0) || (da > 0 && db < 0)) {
        const double s = da / (da - db);
        for (int c = 0; c < 3; ++c) poly[cur ^ 1][m][c] = pa_[c] + s * (pb_[c] - pa_[c]);
        ++m;
      }
    }
    np = m;
    cur ^= 1;
  }
  int nc = 0;
  for (int q = 0; q < np && nc < 8; ++q) {
    const double depth = sr[a] - sgn * poly[cur][q][a];
    if (depth < 0) continue;
    double w[3];
    mulmv(Rr, poly[cur][q], w);
    for (int c = 0; c < 3; ++c) pos[3 * nc + c] = w[c] + pr[c] + n[c] * 0.5 * depth;
    dist[nc] = -depth;
    ++nc;
  }
  for (int c = 0; c < 3; ++c) nrm[c] = ref1 ? n[c] : -n[c];
  return nc;
}

// ------------------------------------------------------------------ Minkowski portal refinement (oracle: mpr_penetration)
// Everything between here and `dev_mpr` is compiled WITHOUT multiply-add contraction and with vector helpers of its own that
// round every product, as the oracle does (gcc -ffp-contract=off) and as libccd built by any C compiler does: the refinement is
// a chain of sign tests on cross / dot products of nearly coplanar support points, and which portal it ends on -- a jump of
// the contact normal by degrees -- must not depend on whether a product was rounded before its sum.
#pragma clang fp contract(off)
RCSH_D double nofma_dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
RCSH_D void nofma_cross3(const double* a, const double* b, double* r) {
  const double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
RCSH_D void nofma_mulmv(const double* A, const double* v, double* r) {
  const double x = A[0] * v[0] + A[1] * v[1] + A[2] * v[2], y = A[3] * v[0] + A[4] * v[1] + A[5] * v[2], z = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
RCSH_D void nofma_mulTv(const double* R, const double* v, double* o) {
  const double x = R[0] * v[0] + R[3] * v[1] + R[6] * v[2], y = R[1] * v[0] + R[4] * v[1] + R[7] * v[2], z = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
#define dot3 nofma_dot3
#define cross3 nofma_cross3
#define mulmv nofma_mulmv
#define mulTv nofma_mulTv
// (frame and size by VALUE: as pointers to the caller's local arrays they pinned those arrays -- 24 doubles per pair -- in private
// memory, the only scratch the flag-only kernels had)
struct Shape {
  int type;              // 0 hull, 1 box, 2 capsule
  double p[3], R[9];     // world frame
  double size[3];
  const double* verts;
  int nvert;
  double center[3];
};
RCSH_D Shape make_shape(int type, const double* p, const double* R, const double* size, const double* verts, int nvert) {
  Shape s;
  s.type = type;
#pragma unroll
  for (int k = 0; k < 3; ++k) { s.p[k] = p[k]; s.size[k] = size[k]; s.center[k] = p[k]; }
#pragma unroll
  for (int k = 0; k < 9; ++k) s.R[k] = R[k];
  s.verts = verts;
  s.nvert = nvert;
  return s;
}
// Support ties are broken by RULE, not by round-off (oracle: SUPPORT_TIE, where the reason is written down): vertices within
// kSupportTie of the largest projection count as tied and the lowest index wins; a box / capsule axis whose direction
// component is above -kSupportTie takes its positive end.
constexpr double kSupportTie = 1e-10;
// Hull support.  TEAM = false: the calling lane scans all vertices (global memory), twice -- the largest projection, then the
// first vertex within the tie band of it.  TEAM = true: the 16 lanes of a team compute the same query on vertices staged in
// LDS, scan every 16th vertex each, agree on the largest projection with four row rotations, then each lane looks for its
// first vertex inside the band and the team takes the lowest index -- a lane scanning global memory alone pays a full round
// trip per vertex, its wavefront having nothing else to run.
template <bool TEAM, bool ONE = false>
RCSH_D int hull_support_index(const double* verts_, int nvert, const double* l) {
  double bestv = -INFINITY;
  int bi = 0;
  if (!TEAM) {
    const double* verts = verts_;
    for (int i = 0; i < nvert; ++i) {
      const double v = verts[3 * i] * l[0] + verts[3 * i + 1] * l[1] + verts[3 * i + 2] * l[2];
      if (v > bestv) bestv = v;
    }
    for (int i = 0; i < nvert; ++i) {
      const double v = verts[3 * i] * l[0] + verts[3 * i + 1] * l[1] + verts[3 * i + 2] * l[2];
      if (v >= bestv - kSupportTie) { bi = i; break; }
    }
    return bi;
  }
  const double* verts = in_lds(verts_);
  const int t = threadIdx.x & (kTeamLanes - 1);
  if constexpr (ONE) {
    // ONE pass over the vertices: the lane's projections stay in registers (a hull has at most kHullMaxVerts vertices: twelve a lane),
    // and the second question -- the first vertex inside the tie band -- is asked of them, not of LDS again (through round 6's first
    // half the scan ran twice: two thirds of a support query's instructions, and the support queries are four fifths of a portal
    // refinement's 43k cycles).  The same products, rounded the same way: the same index.
    constexpr int kTrips = (kHullMaxVerts + 4 * kTeamLanes - 1) / (4 * kTeamLanes);
    double pv[kTrips][4];
#pragma unroll
    for (int r = 0; r < kTrips; ++r) {
      const int i0 = t + 4 * kTeamLanes * r;
#pragma unroll
      for (int u = 0; u < 4; ++u) pv[r][u] = -INFINITY;
      if (4 * kTeamLanes * r < nvert) {
        // four vertices per trip: their reads go out together
        double x[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * kTeamLanes < nvert ? i0 + u * kTeamLanes : 0;
          x[u][0] = verts[3 * i]; x[u][1] = verts[3 * i + 1]; x[u][2] = verts[3 * i + 2];
        }
        sched_fence();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const double v = x[u][0] * l[0] + x[u][1] * l[1] + x[u][2] * l[2];
          pv[r][u] = i0 + u * kTeamLanes < nvert ? v : -INFINITY;
          if (pv[r][u] > bestv) bestv = pv[r][u];
        }
      }
    }
    bestv = fmax(bestv, row_rotate<8>(bestv));
    bestv = fmax(bestv, row_rotate<4>(bestv));
    bestv = fmax(bestv, row_rotate<2>(bestv));
    bestv = fmax(bestv, row_rotate<1>(bestv));
    const double band = bestv - kSupportTie;
    bi = 0x7fffffff;  // (a lane without a vertex in the band: the largest index, never wins)
#pragma unroll
    for (int r = kTrips - 1; r >= 0; --r)
#pragma unroll
      for (int u = 3; u >= 0; --u)
        if (pv[r][u] >= band) bi = t + 4 * kTeamLanes * r + u * kTeamLanes;  // descending: the lane's lowest index is kept
  } else {
    for (int i0 = t; i0 < nvert; i0 += 4 * kTeamLanes) {
      // four vertices per trip: their reads go out together
      double x[4][3];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * kTeamLanes < nvert ? i0 + u * kTeamLanes : i0;
        x[u][0] = verts[3 * i]; x[u][1] = verts[3 * i + 1]; x[u][2] = verts[3 * i + 2];
      }
      sched_fence();
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double v = x[u][0] * l[0] + x[u][1] * l[1] + x[u][2] * l[2];  // (a clamped duplicate of vertex i0 changes no maximum)
        if (v > bestv) bestv = v;
      }
    }
    bestv = fmax(bestv, row_rotate<8>(bestv));
    bestv = fmax(bestv, row_rotate<4>(bestv));
    bestv = fmax(bestv, row_rotate<2>(bestv));
    bestv = fmax(bestv, row_rotate<1>(bestv));
    const double band = bestv - kSupportTie;
    bi = 0x7fffffff;  // (a lane without a vertex in the band: the largest index, never wins)
    for (int i0 = t; i0 < nvert && bi == 0x7fffffff; i0 += 4 * kTeamLanes) {
      double x[4][3];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * kTeamLanes < nvert ? i0 + u * kTeamLanes : i0;
        x[u][0] = verts[3 * i]; x[u][1] = verts[3 * i + 1]; x[u][2] = verts[3 * i + 2];
      }
      sched_fence();
#pragma unroll
      for (int u = 3; u >= 0; --u) {
        const int i = i0 + u * kTeamLanes;
        const double v = x[u][0] * l[0] + x[u][1] * l[1] + x[u][2] * l[2];
        if (i < nvert && v >= band) bi = i;  // descending u: the lowest index of the trip is kept
      }
    }
  }
#define RCSH_ROT_MIN(N)                                                            \
  {                                                                                \
    const int oi = __builtin_amdgcn_update_dpp(0, bi, 0x120 + N, 0xf, 0xf, true);  \
    bi = oi < bi ? oi : bi;                                                        \
  }
  RCSH_ROT_MIN(8) RCSH_ROT_MIN(4) RCSH_ROT_MIN(2) RCSH_ROT_MIN(1)
#undef RCSH_ROT_MIN
  return bi;
}
template <bool TEAM = false, bool ONE = false>
RCSH_D void shape_support(const Shape& s, const double* dir, double* out) {
  double l[3], w[3] = {0, 0, 0};
  mulTv(s.R, dir, l);
  if (s.type == 0) {
    const int bi = hull_support_index<TEAM, ONE>(s.verts, s.nvert, l);
    const double* vv = TEAM ? in_lds(s.verts) : s.verts;
    w[0] = vv[3 * bi]; w[1] = vv[3 * bi + 1]; w[2] = vv[3 * bi + 2];
  } else if (s.type == 1) {
    for (int k = 0; k < 3; ++k) w[k] = l[k] >= -kSupportTie ? s.size[k] : -s.size[k];
  } else {
    const double nl = sqrt(dot3(l, l));
    w[2] = l[2] >= -kSupportTie ? s.size[1] : -s.size[1];
    if (nl > kMinVal) for (int k = 0; k < 3; ++k) w[k] += s.size[0] * l[k] / nl;
  }
  mulmv(s.R, w, out);
  out[0] += s.p[0]; out[1] += s.p[1]; out[2] += s.p[2];
}
struct MprPt { double v[3], v1[3], v2[3]; };
template <bool TEAM = false, bool ONE = false>
RCSH_D void mpr_support(const Shape& a, const Shape& b, const double* dir, MprPt& o) {
  const double nd[3] = {-dir[0], -dir[1], -dir[2]};
  shape_support<TEAM, ONE>(a, dir, o.v1);
  shape_support<TEAM, ONE>(b, nd, o.v2);
  for (int k = 0; k < 3; ++k) o.v[k] = o.v1[k] - o.v2[k];
}
// Gilbert's iteration for "are the two shapes apart": C = A - B is convex, and a unit direction d with the support of C along it
// negative proves 0 outside C, the shapes at least that support's size apart.  From a point x of C: d = -x / |x| (towards the
// origin); s = the support point of C along d; d . s < -margin: proven (dir, gap); else x moves to the point of the segment
// [x, s] nearest the origin and the step repeats.  One support query a step, where a full refinement (MPR) costs ten to twenty --
// and the direction it ends on has a larger gap than the one MPR stops at, so the remembered direction survives more motion and
// the slack credited to the pair is larger.  Used as an accelerator only: a failure (touching, overlapping, or just slow -- the
// iteration zigzags near contact) falls through to MPR, and `margin` keeps its verdicts away from the band of MPR's own tolerance.
// `extra`: support queries spent AFTER the first proof on a better one (a larger gap: the slack a caller credits the pair with -- the
// first direction that separates often proves a millimetre where the shapes are centimetres apart, and the pair is back a substep later).
template <bool TEAM, bool ONE = false>
RCSH_D bool gilbert_apart(const Shape& A, const Shape& B, const double* x0, int iters, double margin, double* dir, double* gap, int extra = 0) {
  double x[3] = {x0[0], x0[1], x0[2]};
  bool proven = false;
  for (int it = 0; it < iters; ++it) {
    const double n2 = dot3(x, x);
    if (!(n2 > 1e-16)) return proven;
    const double inv = 1.0 / sqrt(n2);
    const double d[3] = {-x[0] * inv, -x[1] * inv, -x[2] * inv};
    MprPt s;
    mpr_support<TEAM, ONE>(A, B, d, s);
    const double h = dot3(s.v, d);
    if (h < -margin) {
      const bool better = !proven || -h > *gap, much = !proven || -h > 1.2 * *gap;
      if (better) { dir[0] = d[0]; dir[1] = d[1]; dir[2] = d[2]; *gap = -h; }
      proven = true;
      if (extra <= 0 || !much) return true;
      --extra;
    } else if (proven) return true;
    const double e[3] = {s.v[0] - x[0], s.v[1] - x[1], s.v[2] - x[2]};
    const double ee = dot3(e, e);
    if (!(ee > 1e-24)) return proven;
    double tt = -dot3(x, e) / ee;
    tt = tt < 0 ? 0.0 : (tt > 1 ? 1.0 : tt);
    x[0] += tt * e[0]; x[1] += tt * e[1]; x[2] += tt * e[2];
  }
  return proven;
}
RCSH_D void portal_dir(const MprPt& p1, const MprPt& p2, const MprPt& p3, double* dir) {
  const double e1[3] = {p2.v[0] - p1.v[0], p2.v[1] - p1.v[1], p2.v[2] - p1.v[2]}, e2[3] = {p3.v[0] - p1.v[0], p3.v[1] - p1.v[1], p3.v[2] - p1.v[2]};
  cross3(e1, e2, dir);
  const double l = sqrt(dot3(dir, dir));
  if (l > kMinVal) { dir[0] /= l; dir[1] /= l; dir[2] /= l; }
}
// (which of the portal's three points the new one replaces is decided at run time: written as selects per component -- assigning
// whole structs behind the branches made the compiler address p1 / p2 / p3 through a pointer and keep all of them in scratch)
RCSH_D void expand_portal(const MprPt& p0, MprPt& p1, MprPt& p2, MprPt& p3, const MprPt& p4) {
  double c[3];
  cross3(p4.v, p0.v, c);
  const bool a1 = dot3(p1.v, c) > 0, a2 = dot3(p2.v, c) > 0, a3 = dot3(p3.v, c) > 0;
  const bool r1 = a1 ? a2 : !a3, r3 = a1 && !a2, r2 = !a1 && a3;  // replace p1 / p3 / p2 (exactly one holds)
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    p1.v[k] = r1 ? p4.v[k] : p1.v[k]; p1.v1[k] = r1 ? p4.v1[k] : p1.v1[k]; p1.v2[k] = r1 ? p4.v2[k] : p1.v2[k];
    p2.v[k] = r2 ? p4.v[k] : p2.v[k]; p2.v1[k] = r2 ? p4.v1[k] : p2.v1[k]; p2.v2[k] = r2 ? p4.v2[k] : p2.v2[k];
    p3.v[k] = r3 ? p4.v[k] : p3.v[k]; p3.v1[k] = r3 ? p4.v1[k] : p3.v1[k]; p3.v2[k] = r3 ? p4.v2[k] : p3.v2[k];
  }
}
RCSH_D double origin_tri_dist2(const double* a, const double* b, const double* c, double* witness) {
  const double ab[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, ac[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]}, ap[3] = {-a[0], -a[1], -a[2]};
  const double d1 = dot3(ab, ap), d2 = dot3(ac, ap);
  double s, t;
  if (d1 <= 0 && d2 <= 0) { s = 0; t = 0; }
  else {
    const double bp[3] = {-b[0], -b[1], -b[2]}, d3 = dot3(ab, bp), d4 = dot3(ac, bp);
    const double cp[3] = {-c[0], -c[1], -c[2]}, d5 = dot3(ab, cp), d6 = dot3(ac, cp);
    const double vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
    if (d3 >= 0 && d4 <= d3) { s = 1; t = 0; }
    else if (vc <= 0 && d1 >= 0 && d3 <= 0) { s = d1 / (d1 - d3); t = 0; }
    else if (d6 >= 0 && d5 <= d6) { s = 0; t = 1; }
    else if (vb <= 0 && d2 >= 0 && d6 <= 0) { s = 0; t = d2 / (d2 - d6); }
    else if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) { t = (d4 - d3) / ((d4 - d3) + (d5 - d6)); s = 1 - t; }
    else { const double den = 1 / (va + vb + vc); s = vb * den; t = vc * den; }
  }
  for (int k = 0; k < 3; ++k) witness[k] = a[k] + s * ab[k] + t * ac[k];
  return dot3(witness, witness);
}
// Returns 0 when the shapes do not overlap; dir_out is then a direction along which the support of A - B is not positive
// (a separating direction), or the zero vector where the refinement gave up without one.
// WANT: what the caller reads.  kMprFull: depth, direction, position.  kMprOverlap: nothing but "do they overlap" (the flag-only
// detection): depth and position are not computed -- the third phase of the refinement, which only sharpens them, is skipped --
// and may be null.  kMprDepth: the depth too, not the position (the unresolved-contact check: pos may be null).
constexpr int kMprFull = 0, kMprOverlap = 1, kMprDepth = 2;
template <bool TEAM, int WANT = kMprFull, bool ONE = false>
RCSH_D int mpr_penetration(const Shape& A, const Shape& B, double* depth, double* dir_out, double* pos) {
  constexpr bool OVERLAP_ONLY = WANT == kMprOverlap;
  constexpr double kTol = 1e-6;
  constexpr int kIter = 50;
  MprPt p0, p1, p2, p3, p4;
  double dir[3], va[3], vb[3];
  for (int k = 0; k < 3; ++k) { p0.v[k] = A.center[k] - B.center[k]; p0.v1[k] = A.center[k]; p0.v2[k] = B.center[k]; }
  if (fabs(p0.v[0]) < kMinVal && fabs(p0.v[1]) < kMinVal && fabs(p0.v[2]) < kMinVal) p0.v[0] = 1e-5;
  double l = sqrt(dot3(p0.v, p0.v));
  for (int k = 0; k < 3; ++k) dir[k] = -p0.v[k] / l;
  mpr_support<TEAM, ONE>(A, B, dir, p1);
  // (a separating direction: `depth`, where the caller gave one, takes the support of A - B along it -- minus a lower bound of the distance)
  if (dot3(p1.v, dir) <= 0) { dir_out[0] = dir[0]; dir_out[1] = dir[1]; dir_out[2] = dir[2]; if (WANT == kMprFull && depth) *depth = dot3(p1.v, dir); return 0; }
  cross3(p0.v, p1.v, dir);
  l = sqrt(dot3(dir, dir));
  if (l < 1e-12) {
    if (OVERLAP_ONLY) return 1;
    *depth = sqrt(dot3(p1.v, p1.v));
    if (WANT == kMprDepth) return 1;
    const double l0 = sqrt(dot3(p0.v, p0.v));
    for (int k = 0; k < 3; ++k) { dir_out[k] = *depth > kMinVal ? p1.v[k] / *depth : -p0.v[k] / l0; pos[k] = 0.5 * (p1.v1[k] + p1.v2[k]); }
    return 1;
  }
  for (int k = 0; k < 3; ++k) dir[k] /= l;
  mpr_support<TEAM, ONE>(A, B, dir, p2);
  if (dot3(p2.v, dir) <= 0) { dir_out[0] = dir[0]; dir_out[1] = dir[1]; dir_out[2] = dir[2]; if (WANT == kMprFull && depth) *depth = dot3(p2.v, dir); return 0; }
  for (int k = 0; k < 3; ++k) { va[k] = p1.v[k] - p0.v[k]; vb[k] = p2.v[k] - p0.v[k]; }
  cross3(va, vb, dir);
  l = sqrt(dot3(dir, dir));
  for (int k = 0; k < 3; ++k) dir[k] /= l;
  if (dot3(dir, p0.v) > 0) {
    const MprPt tmp = p1; p1 = p2; p2 = tmp;
    for (int k = 0; k < 3; ++k) dir[k] = -dir[k];
  }
  for (int guard = 0;; ++guard) {
    if (guard > 100) { dir_out[0] = dir_out[1] = dir_out[2] = 0.0; return 0; }
    mpr_support<TEAM, ONE>(A, B, dir, p3);
    if (dot3(p3.v, dir) <= 0) { dir_out[0] = dir[0]; dir_out[1] = dir[1]; dir_out[2] = dir[2]; if (WANT == kMprFull && depth) *depth = dot3(p3.v, dir); return 0; }
    bool cont = false;
    cross3(p1.v, p3.v, va);
    if (dot3(va, p0.v) < -kMinVal) { p2 = p3; cont = true; }
    if (!cont) {
      cross3(p3.v, p2.v, va);
      if (dot3(va, p0.v) < -kMinVal) { p1 = p3; cont = true; }
    }
    if (!cont) break;
    for (int k = 0; k < 3; ++k) { va[k] = p1.v[k] - p0.v[k]; vb[k] = p2.v[k] - p0.v[k]; }
    cross3(va, vb, dir);
    l = sqrt(dot3(dir, dir));
    for (int k = 0; k < 3; ++k) dir[k] /= l;
  }
  for (int it = 0;; ++it) {
    portal_dir(p1, p2, p3, dir);
    if (dot3(dir, p1.v) >= 0) break;
    mpr_support<TEAM, ONE>(A, B, dir, p4);
    const double dv4 = dot3(p4.v, dir);
    const double dmax = fmax(dot3(p1.v, dir), fmax(dot3(p2.v, dir), dot3(p3.v, dir)));
    if (dv4 < 0 || dv4 - dmax <= kTol || it > kIter) {
      const double keep = dv4 < 0 ? 1.0 : 0.0;
      dir_out[0] = keep * dir[0]; dir_out[1] = keep * dir[1]; dir_out[2] = keep * dir[2];
      if (WANT == kMprFull && depth && dv4 < 0) *depth = dv4;
      return 0;
    }
    expand_portal(p0, p1, p2, p3, p4);
  }
  if (OVERLAP_ONLY) return 1;  // (the origin is inside the portal: every exit of the loop below reports a penetration)
  for (int it = 0;; ++it) {
    portal_dir(p1, p2, p3, dir);
    mpr_support<TEAM, ONE>(A, B, dir, p4);
    const double dv4 = dot3(p4.v, dir);
    const double dmax = fmax(dot3(p1.v, dir), fmax(dot3(p2.v, dir), dot3(p3.v, dir)));
    if (dv4 - dmax <= kTol || it > kIter) {
      double w[3];
      const double d2 = origin_tri_dist2(p1.v, p2.v, p3.v, w);
      *depth = sqrt(d2);
      if (WANT == kMprDepth) return 1;
      if (*depth > kMinVal) for (int k = 0; k < 3; ++k) dir_out[k] = w[k] / *depth;
      else for (int k = 0; k < 3; ++k) dir_out[k] = dir[k];
      double bw[4], c[3];
      cross3(p1.v, p2.v, c); bw[0] = dot3(c, p3.v);
      cross3(p3.v, p2.v, c); bw[1] = dot3(c, p0.v);
      cross3(p0.v, p1.v, c); bw[2] = dot3(c, p3.v);
      cross3(p2.v, p1.v, c); bw[3] = dot3(c, p0.v);
      double sum = bw[0] + bw[1] + bw[2] + bw[3];
      if (sum <= 0) {
        bw[0] = 0;
        cross3(p2.v, p3.v, c); bw[1] = dot3(c, dir);
        cross3(p3.v, p1.v, c); bw[2] = dot3(c, dir);
        cross3(p1.v, p2.v, c); bw[3] = dot3(c, dir);
        sum = bw[1] + bw[2] + bw[3];
      }
      for (int k = 0; k < 3; ++k) {
        const double a1 = bw[0] * p0.v1[k] + bw[1] * p1.v1[k] + bw[2] * p2.v1[k] + bw[3] * p3.v1[k];
        const double a2 = bw[0] * p0.v2[k] + bw[1] * p1.v2[k] + bw[2] * p2.v2[k] + bw[3] * p3.v2[k];
        pos[k] = 0.5 * (a1 + a2) / sum;
      }
      return 1;
    }
    expand_portal(p0, p1, p2, p3, p4);
  }
}

RCSH_CONTACT_FN int dev_mpr(const Shape& A, const Shape& B, double* depth, double* dir_out, double* pos) {
  return mpr_penetration<false>(A, B, depth, dir_out, pos);
}
#undef dot3
#undef cross3
#undef mulmv
#undef mulTv
#pragma clang fp contract(fast)

// ---- self collision, flags only (DET instantiations of k_run_team).  Every lane of the wavefront calls this.
// Broad phase: lane t of a team whose collision callback is due takes pairs t, t + 16, ... of the table, three at a time so
// that their records travel together -- bounding spheres, then the geoms' oriented boxes, all from the pair record.
// Narrow phase: a surviving pair (typically one per environment: links 5 and 7 wrap around the same wrist) has its two
// hulls staged in LDS by the whole wavefront, once for all the teams it survived in; each of those teams then runs the
// portal refinement (mjc_Convex for every convex pair: only "do they overlap" is read from it) identically on its 16 lanes,
// which share the vertex scans of the support queries.
// Fall: world frames of the links of the wavefront's four teams, [4][NL][12] in LDS (R row-major, p); geoms welded to the
// world carry their world frame in the table.  stage: LDS room for kSelfStage doubles.
// Returns the class bits of the overlapping pairs of the calling lane's team.
constexpr int kSelfStageVerts = 304;  // both hulls of a pair (host: build_self_pairs checks)
constexpr int kSelfStage = 3 * kSelfStageVerts;
constexpr int kSelfSlots = 4;          // remembered separating directions per team (pair index, direction): a folded arm keeps three or four pairs near
constexpr int kSelfCache = 4 * 4 * kSelfSlots;
constexpr int kSelfTag = 2;            // which pair the stage holds
constexpr int kMaxSelfPairs = 144;     // pairs whose bounding spheres the lean DET kernels keep in LDS (host: build_self_pairs)
constexpr int kSelfSphereWords = 10;   // per pair: c0, c1, r0 + r1 (inflated by the rounding) as float, the two links, the joints between them
RCSH_D void self_geom_world(const ContactGeom& g, const double* F, double* R, double* p) {
  if (g.link < 0) {
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = g.rot[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = g.pos[k];
    return;
  }
  const double* L = F + 12 * g.link;
  double LR[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) LR[k] = L[k];
  mulmm(LR, g.rot, R);
  mulmv(LR, g.pos, p);
  p[0] += L[9]; p[1] += L[10]; p[2] += L[11];
}
// oriented boxes (centre c, axes = columns of R, half extents h): true if a separating axis exists among the 15 candidates
// `gap` (optional): a lower bound of the boxes' distance where one of the six face axes separates them, else 0
RCSH_D bool obb_disjoint(const double* Ra, const double* ca, const double* ha, const double* Rb, const double* cb, const double* hb, double* gap = nullptr) {
  double C[9], A[9], tw[3], tv[3];
  double sep = 0.0;
  const double d[3] = {cb[0] - ca[0], cb[1] - ca[1], cb[2] - ca[2]};
  mulTv(Ra, d, tv);  // centre offset in A's frame
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      C[3 * i + j] = Ra[i] * Rb[j] + Ra[3 + i] * Rb[3 + j] + Ra[6 + i] * Rb[6 + j];  // A_i . B_j
      A[3 * i + j] = fabs(C[3 * i + j]) + 1e-9;
    }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double g = fabs(tv[i]) - (ha[i] + hb[0] * A[3 * i] + hb[1] * A[3 * i + 1] + hb[2] * A[3 * i + 2]);
    sep = g > sep ? g : sep;
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    tw[j] = tv[0] * C[j] + tv[1] * C[3 + j] + tv[2] * C[6 + j];
    const double g = fabs(tw[j]) - (hb[j] + ha[0] * A[j] + ha[1] * A[3 + j] + ha[2] * A[6 + j]);
    sep = g > sep ? g : sep;
  }
  if (gap) *gap = sep;
  if (sep > 0) return true;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      const double ra = ha[i1] * A[3 * i2 + j] + ha[i2] * A[3 * i1 + j];
      const double rb = hb[j1] * A[3 * i + j2] + hb[j2] * A[3 * i + j1];
      const double over = fabs(tv[i2] * C[3 * i1 + j] - tv[i1] * C[3 * i2 + j]) - (ra + rb);
      if (over > 0) {
        if (gap) {
          // the axis A_i x B_j is not a unit vector: |A_i x B_j|^2 = 1 - (A_i . B_j)^2
          const double l2 = 1.0 - C[3 * i + j] * C[3 * i + j];
          if (l2 > 1e-6) *gap = over / sqrt(l2);
        }
        return true;
      }
    }
  return false;
}
// bounding box of a collision geom in its own frame: centre, half extents
RCSH_D void geom_obb(const ContactGeom& g, double* c, double* h) {
  if (g.type == 7) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { c[k] = g.aabb_c[k]; h[k] = g.aabb_h[k]; }
  } else {
    c[0] = c[1] = c[2] = 0.0;
    if (g.type == 6) { h[0] = g.size[0]; h[1] = g.size[1]; h[2] = g.size[2]; }
    else { h[0] = h[1] = g.size[0]; h[2] = g.size[0] + g.size[1]; }
  }
}
RCSH_D void self_box_world(const double* F, int link, const double* c, const double* rot, double* cw, double* Rw) {
  if (link < 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) cw[k] = c[k];
#pragma unroll
    for (int k = 0; k < 9; ++k) Rw[k] = rot[k];
    return;
  }
  const double* L = F + 12 * link;
  double LR[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) LR[k] = L[k];
  mulmv(LR, c, cw);
  cw[0] += L[9]; cw[1] += L[10]; cw[2] += L[11];
  mulmm(LR, rot, Rw);
}
// Fills the LDS copy of the pairs' bounding spheres (single precision, the sum of the radii rounded up: conservative).
RCSH_D void self_sphere_table_fill(const SelfPair* pairs, int npair, float* tab, int narm) {
  for (int i = threadIdx.x; i < npair; i += 64) {
    const SelfPair& pr = pairs[i];
    float* w = tab + kSelfSphereWords * i;
    w[0] = (float)pr.c0[0]; w[1] = (float)pr.c0[1]; w[2] = (float)pr.c0[2];
    w[3] = (float)pr.c1[0]; w[4] = (float)pr.c1[1]; w[5] = (float)pr.c1[2];
    w[6] = (float)(pr.r0 + pr.r1) * 1.000001f + 1e-6f;
    reinterpret_cast<int*>(w)[7] = pr.l0;
    reinterpret_cast<int*>(w)[8] = pr.l1;
    // the joints between the two links, for the slack test: arm joints [a0, a1) as a range of the prefix sums, plus the
    // slides of the fingers among the two links (joint index + 1, 0: none)
    int a0 = 0, a1 = 0, f0 = 0, f1 = 0;
    {
      const int arm = pr.joints & ((1 << narm) - 1);
      if (arm) { a0 = __ffs(arm) - 1; a1 = 32 - __clz(arm); }
      const int fing = pr.joints >> narm;
      if (fing & 1) f0 = narm + 1;
      if (fing & 2) f1 = narm + 2;
    }
    reinterpret_cast<int*>(w)[9] = a0 | (a1 << 8) | (f0 << 16) | (f1 << 24);
  }
}
// What the self-collision test remembers per team for the length of a launch (lean DET kernels only; LDS).  A pair found
// apart by a gap g cannot touch before the joints between its two links have moved the geoms by g: path[j] integrates
// lever[j] * |dq_j| over the calls (prefix[] its running sums along the arm), and due[i] is the value the pair's sum of
// path[] must reach before the pair is looked at again: the sum when its gap was measured, plus the gap.  In a settled or
// slowly moving arm every pair is skipped, every call.
// (kSlackJ entries per team: joints 0 .. NL - 1 and one more prefix sum.  Ten, not kMaxLinks: with twelve the lean DET kernel of the
// 9-joint robots took 41,144 bytes of LDS -- 184 more than let FOUR workgroups share a CU -- and a batch of 4096 environments ran as
// 768 + 256 workgroups, one round after the other: step_until_convergence at HALF its speed.  tests/test_host_logic.py keeps watch.)
constexpr int kSlackJ = 10;
struct SelfSlack {
  double qprev[4][kSlackJ], path[4][kSlackJ], prefix[4][kSlackJ];
  float due[4][kMaxSelfPairs];
};
RCSH_D void self_slack_clear(SelfSlack& ss) {
  for (int k = threadIdx.x; k < 4 * kSlackJ; k += 64) { (&ss.qprev[0][0])[k] = 0.0; (&ss.path[0][0])[k] = 0.0; (&ss.prefix[0][0])[k] = 0.0; }
  for (int k = threadIdx.x; k < 4 * kMaxSelfPairs; k += 64) (&ss.due[0][0])[k] = 0.0f;
}
RCSH_CONTACT_FN uint32_t self_collision_pairs(const ContactGeom* geoms, const double* verts, const SelfPair* pairs, int npair, const double* Fall_,
                                              double* stage_, int nl, bool team_due, bool keep_stage, const float* sph_, SelfSlack* ss_,
                                              const double* lever_, double q_lane) {
  const double* Fall = in_lds(Fall_);
  double* stage = in_lds(stage_);
  SelfSlack* ss = ss_ ? in_lds(ss_) : nullptr;
  const int lane = threadIdx.x & 63, t = lane & (kTeamLanes - 1), team = lane / kTeamLanes;
  const double* F = Fall + 12 * nl * team;
  uint32_t cmask = 0, smask = 0;  // survivors of the broad phase (of its bounding spheres): bit j = pair t + 16 j
  TEAM_MARK(42)
  if (team_due && sph_ && ss) {
    // how far the joints have come since the last call
    double mine_path = 0.0;
    if (t < nl && t < kSlackJ) {
      const double lever = in_lds(lever_)[t];
      mine_path = ss->path[team][t] + lever * fabs(q_lane - ss->qprev[team][t]);
      ss->path[team][t] = mine_path;
      ss->qprev[team][t] = q_lane;
    }
    // running sums along the lanes (arm joints are lanes 0 .. narm - 1): prefix[k] = path[0] + .. + path[k - 1]
    double incl = mine_path;
    incl += row_up<1>(incl);
    incl += row_up<2>(incl);
    incl += row_up<4>(incl);
    if (t + 1 < kSlackJ) ss->prefix[team][t + 1] = incl;
    stage_fence();
  }
  if (team_due && sph_) {
    // the pairs' bounding spheres from their LDS table -- of the pairs whose slack is used up (with a slack cache; all without)
    const float* sph = in_lds(sph_);
    uint32_t amask = 0;
    if (ss) {
      // the pairs' slack: the gap measured last against what the joints between the two links have moved since.  The reads
      // of all the lane's pairs go out together (a dependent chain per pair would cost an LDS round trip each, three deep).
      constexpr int kMaxPer = (kMaxSelfPairs + kTeamLanes - 1) / kTeamLanes;
      int jw[kMaxPer];
#pragma unroll
      for (int j = 0; j < kMaxPer; ++j) {
        const int i = t + kTeamLanes * j;
        jw[j] = reinterpret_cast<const int*>(sph + kSelfSphereWords * (i < npair ? i : t))[9];
      }
      sched_fence();
      double hi[kMaxPer], lo[kMaxPer], p0[kMaxPer], p1[kMaxPer];
      float due[kMaxPer];
#pragma unroll
      for (int j = 0; j < kMaxPer; ++j) {
        const int i = t + kTeamLanes * j, f0 = (jw[j] >> 16) & 0xff, f1 = (jw[j] >> 24) & 0xff;
        hi[j] = ss->prefix[team][(jw[j] >> 8) & 0xff];
        lo[j] = ss->prefix[team][jw[j] & 0xff];
        p0[j] = f0 ? ss->path[team][f0 - 1] : 0.0;
        p1[j] = f1 ? ss->path[team][f1 - 1] : 0.0;
        due[j] = ss->due[team][i < npair ? i : t];
      }
      sched_fence();
#pragma unroll
      for (int j = 0; j < kMaxPer; ++j) {
        const int i = t + kTeamLanes * j;
        const float moved = (float)((hi[j] - lo[j]) + p0[j] + p1[j]) * 1.000001f + 1e-6f;  // (rounded up)
        if (i < npair && !(moved < due[j])) {
          ss->due[team][i] = moved;  // (gap 0) until one of the stages below measures a new one
          amask |= 1u << j;
        }
      }
    } else {
      for (int j = 0, i = t; i < npair; ++j, i += kTeamLanes) amask |= 1u << j;
    }
    // one pair per lane and round: the wavefront runs the sphere test as often as its busiest lane has pairs to look at
    while (__ballot(amask != 0)) {
      if (!amask) continue;
      const int j = __ffs((int)amask) - 1, i = t + kTeamLanes * j;
      amask &= amask - 1;
      const float* w = sph + kSelfSphereWords * i;
      const double c0[3] = {w[0], w[1], w[2]}, c1[3] = {w[3], w[4], w[5]}, rs = w[6];
      const int l0 = reinterpret_cast<const int*>(w)[7], l1 = reinterpret_cast<const int*>(w)[8];
      double s0[3], s1[3];
      if (l0 >= 0) { const double* L = F + 12 * l0; mulmv(L, c0, s0); s0[0] += L[9]; s0[1] += L[10]; s0[2] += L[11]; }
      else { s0[0] = c0[0]; s0[1] = c0[1]; s0[2] = c0[2]; }
      if (l1 >= 0) { const double* L = F + 12 * l1; mulmv(L, c1, s1); s1[0] += L[9]; s1[1] += L[10]; s1[2] += L[11]; }
      else { s1[0] = c1[0]; s1[1] = c1[1]; s1[2] = c1[2]; }
      const double d[3] = {s0[0] - s1[0], s0[1] - s1[1], s0[2] - s1[2]};
      const double d2 = dot3(d, d);
      if (d2 <= rs * rs) smask |= 1u << j;
      else if (ss) ss->due[team][i] += (float)(sqrt(d2) - rs) * 0.999999f - 2e-5f;  // (single-precision centres: rounded down)
    }
  } else if (team_due) {
    constexpr int kBatch = 3;
    for (int j0 = 0; t + kTeamLanes * j0 < npair; j0 += kBatch) {
      // the spheres of up to three pairs: one round trip
      double c0[kBatch][3], c1[kBatch][3], rs[kBatch];
      int l0[kBatch], l1[kBatch];
      bool have[kBatch];
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        const int i = t + kTeamLanes * (j0 + u);
        have[u] = i < npair;
        const SelfPair& pr = pairs[have[u] ? i : t];
#pragma unroll
        for (int k = 0; k < 3; ++k) { c0[u][k] = pr.c0[k]; c1[u][k] = pr.c1[k]; }
        rs[u] = pr.r0 + pr.r1;
        l0[u] = pr.l0; l1[u] = pr.l1;
      }
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        if (!have[u]) continue;
        double s0[3], s1[3];
        if (l0[u] >= 0) { const double* L = F + 12 * l0[u]; mulmv(L, c0[u], s0); s0[0] += L[9]; s0[1] += L[10]; s0[2] += L[11]; }
        else { s0[0] = c0[u][0]; s0[1] = c0[u][1]; s0[2] = c0[u][2]; }
        if (l1[u] >= 0) { const double* L = F + 12 * l1[u]; mulmv(L, c1[u], s1); s1[0] += L[9]; s1[1] += L[10]; s1[2] += L[11]; }
        else { s1[0] = c1[u][0]; s1[1] = c1[u][1]; s1[2] = c1[u][2]; }
        const double d[3] = {s0[0] - s1[0], s0[1] - s1[1], s0[2] - s1[2]};
        if (dot3(d, d) <= rs[u] * rs[u]) smask |= 1u << (j0 + u);
      }
    }
  }
  TEAM_MARK(40)
#if defined(RCSH_SELF_SKIP) && RCSH_SELF_SKIP >= 2
  smask = 0;  // (measurement: no box tests, no refinement)
#endif
  if (__ballot(smask != 0)) { TEAM_COUNT(41) }
  // the oriented boxes of the spheres' survivors (a handful per environment): one per lane and round, so that the wavefront
  // runs the box test once or twice instead of in every iteration of the loop above
  while (__ballot(smask != 0)) {
    if (smask) {
      const int j = __ffs((int)smask) - 1;
      smask &= smask - 1;
      const SelfPair& pr = pairs[t + kTeamLanes * j];
      double Ra[9], Rb[9], ca[3], cb[3];
      self_box_world(F, pr.l0, pr.c0, pr.rot0, ca, Ra);
      self_box_world(F, pr.l1, pr.c1, pr.rot1, cb, Rb);
      double sep = 0.0;
      if (!obb_disjoint(Ra, ca, pr.h0, Rb, cb, pr.h1, &sep)) cmask |= 1u << j;
      else if (ss) ss->due[team][t + kTeamLanes * j] += (float)sep * 0.999999f - 2e-5f;
    }
  }
  uint32_t mine = 0;
#if defined(RCSH_SELF_SKIP) && RCSH_SELF_SKIP >= 1
  cmask = 0;  // (measurement: no refinement)
#endif
  TEAM_MARK(43)
  TEAM_COUNT(47)
  for (uint64_t pending = __ballot(cmask != 0); pending; pending = __ballot(cmask != 0)) {
    TEAM_COUNT(46)
    const int src = __ffsll((long long)pending) - 1;  // wave-uniform
    const uint32_t smask = (uint32_t)__builtin_amdgcn_readlane((int)cmask, src);
    const int j = __ffs((int)smask) - 1, t0 = src & (kTeamLanes - 1);
    // the teams this pair survived in (their lane t0 holds bit j) take it together
    const bool holder = t == t0 && ((cmask >> j) & 1u);
    const uint32_t take = team_ballot(holder);  // 0 or 1 << t0 for the caller's team
    if (holder) cmask &= ~(1u << j);
    const SelfPair& pr = pairs[t0 + kTeamLanes * j];
    const ContactGeom& a = geoms[pr.g0];
    const ContactGeom& b = geoms[pr.g1];
    // both hulls' vertices -> LDS: independent loads, one memory round trip for the wavefront.  The pair staged last stays
    // (the same pair keeps coming back substep after substep); the tag is only trusted where this function owns the
    // memory for the whole launch (`keep_stage`: not inside the contact arena).
    const int na = 3 * a.vert_num, nb = 3 * b.vert_num;
    const int pidx = t0 + kTeamLanes * j;
    double* tag = stage + kSelfStage + kSelfCache;
    if (!(keep_stage && tag[0] == (double)(pidx + 1))) {
      const double* va = verts + 3 * (size_t)a.vert_adr;
      const double* vb = verts + 3 * (size_t)b.vert_adr;
      for (int k = lane; k < na; k += 64) stage[k] = va[k];
      for (int k = lane; k < nb; k += 64) stage[na + k] = vb[k];
      if (lane == 0) tag[0] = (double)(pidx + 1);
      stage_fence();  // (LDS traffic of one wavefront is ordered)
    }
    TEAM_MARK(44)
    if (take) {
      double Ra[9], pa[3], Rb[9], pb[3];
      self_geom_world(a, F, Ra, pa);
      self_geom_world(b, F, Rb, pb);
      Shape A = make_shape(a.type == 7 ? 0 : a.type == 6 ? 1 : 2, pa, Ra, a.size, stage, a.vert_num);
      Shape B = make_shape(b.type == 7 ? 0 : b.type == 6 ? 1 : 2, pb, Rb, b.size, stage + na, b.vert_num);
      if (a.type == 7) { mulmv(Ra, a.center, A.center); A.center[0] += pa[0]; A.center[1] += pa[1]; A.center[2] += pa[2]; }
      else { A.center[0] = pa[0]; A.center[1] = pa[1]; A.center[2] = pa[2]; }
      if (b.type == 7) { mulmv(Rb, b.center, B.center); B.center[0] += pb[0]; B.center[1] += pb[1]; B.center[2] += pb[2]; }
      else { B.center[0] = pb[0]; B.center[1] = pb[1]; B.center[2] = pb[2]; }
      // A pair that keeps surviving the broad phase (links 5 and 7) is separated by nearly the same plane substep after
      // substep: the direction the last refinement ended on is remembered in the frame of geom 0's link, and one support
      // query along it settles the pair while it still separates.  (Whatever the slot holds, a direction with negative
      // support proves the hulls apart; anything else falls through to the full refinement.)
      TEAM_MARK(37)
      // (the slot that holds this pair, else the one its index maps to.  Two slots a team thrashed once an arm had wandered into a folded
      // pose -- step_until_convergence over 300 env-steps without a reset went from 3.6 to 6.0 ms per step, all of it full refinements of
      // pairs whose direction had just been evicted)
      double* slots = stage + kSelfStage + 4 * kSelfSlots * team;
      int held = -1;
#pragma unroll
      for (int k = 0; k < kSelfSlots; ++k) held = slots[4 * k] == (double)pidx ? k : held;
      double* slot = slots + 4 * (held >= 0 ? held : (pidx & (kSelfSlots - 1)));
      bool apart = false;
      double LR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      if (pr.l0 >= 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) LR[k] = F[12 * pr.l0 + k];
      }
      double x0[3] = {A.center[0] - B.center[0], A.center[1] - B.center[1], A.center[2] - B.center[2]};  // (a point of A - B)
      if (held >= 0) {
        const double dl[3] = {slot[1], slot[2], slot[3]};
        double dw[3];
        mulmv(LR, dl, dw);
        MprPt q;
        mpr_support<true>(A, B, dw, q);
        const double along = dot3(q.v, dw);  // support of A - B along the (unit) direction: minus a lower bound of the distance
        apart = along < 0;
        if (apart && ss && t == t0) ss->due[team][pidx] += (float)(-along) * 0.999999f - 2e-5f;
        x0[0] = q.v[0]; x0[1] = q.v[1]; x0[2] = q.v[2];
      }
      TEAM_MARK(38)
      if (apart) { TEAM_COUNT(39) }
#ifndef RCSH_NO_GILBERT
      if (!apart) {
        // no remembered direction, or it separates no longer: a few support queries towards a better one before the full refinement
        double dg[3], gap = 0.0;
        if (gilbert_apart<true>(A, B, x0, 5, 1e-5, dg, &gap)) {
          apart = true;
          double dl[3];
          mulTv(LR, dg, dl);
          slot[0] = (double)pidx; slot[1] = dl[0]; slot[2] = dl[1]; slot[3] = dl[2];
          if (ss && t == t0) ss->due[team][pidx] += (float)gap * 0.999999f - 2e-5f;
        }
      }
#endif
      if (!apart) {
        double dir[3];
        if (mpr_penetration<true, kMprOverlap>(A, B, nullptr, dir, nullptr)) mine |= (uint32_t)pr.cls;
        else {
          double dl[3];
          mulTv(LR, dir, dl);
          slot[0] = (double)pidx; slot[1] = dl[0]; slot[2] = dl[1]; slot[3] = dl[2];
        }
      }
    }
    stage_fence();
    TEAM_MARK(45)
  }
  return mine;
}

// elliptic cone of one contact at jar (mj_constraintUpdate): force f = -dcost/djar, Hessian (00 10 11 20 21 22), cost
RCSH_D double cone_eval(const double* D, double mu, double fr, const double* jar, double* f, double* Hc) {
  const double U0 = jar[0] * mu, U1 = jar[1] * fr, U2 = jar[2] * fr;
  const double N = U0, T = sqrt(U1 * U1 + U2 * U2);
#pragma unroll
  for (int k = 0; k < 6; ++k) Hc[k] = 0.0;
  f[0] = f[1] = f[2] = 0.0;
  if (N >= mu * T) return 0.0;
  if (mu * N + T <= 0) {
    double cost = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) { cost += 0.5 * D[k] * jar[k] * jar[k]; f[k] = -D[k] * jar[k]; }
    Hc[0] = D[0]; Hc[2] = D[1]; Hc[5] = D[2];
    return cost;
  }
  const double Dm = D[0] / (mu * mu * (1 + mu * mu));
  const double NmT = N - mu * T;
  const double u1 = U1 / T, u2 = U2 / T;
  f[0] = -mu * (Dm * NmT);
  f[1] = fr * (Dm * NmT * mu * u1);
  f[2] = fr * (Dm * NmT * mu * u2);
  const double k = -Dm * NmT * mu / T;
  Hc[0] = mu * mu * Dm;
  Hc[1] = mu * fr * (-Dm * mu * u1);
  Hc[3] = mu * fr * (-Dm * mu * u2);
  Hc[2] = fr * fr * (Dm * mu * mu * u1 * u1 + k * (1 - u1 * u1));
  Hc[4] = fr * fr * (Dm * mu * mu * u2 * u1 + k * (-u2 * u1));
  Hc[5] = fr * fr * (Dm * mu * mu * u2 * u2 + k * (1 - u2 * u2));
  return 0.5 * Dm * NmT * NmT;
}

// mju_QCQP2 with separate friction coefficients (oracle: qcqp2)
// det0 / di0: the caller's S11 S22 - S12^2 (multiplier 0) and its reciprocal when it has them (the noslip sweeps, where A
// and d stay the same update after update), NaN otherwise
RCSH_D bool qcqp2_dev(double* res, double A00, double A01, double A11, double b0, double b1, double d0, double d1, double r,
                      double det0 = __builtin_nan(""), double di0 = 0.0) {
  const double s1 = b0 * d0, s2 = b1 * d1;
  const double S11 = A00 * d0 * d0, S22 = A11 * d1 * d1, S12 = A01 * d0 * d1;
  double la = 0, v1 = 0, v2 = 0;
  for (int iter = 0; iter < 20; ++iter) {
    const bool given = iter == 0 && det0 == det0;
    const double det = given ? det0 : (S11 + la) * (S22 + la) - S12 * S12;
    if (det < 1e-10) { res[0] = res[1] = 0; return false; }
    const double di = given ? di0 : 1 / det, P11 = (S22 + la) * di, P22 = (S11 + la) * di, P12 = -S12 * di;
    v1 = -P11 * s1 - P12 * s2;
    v2 = -P12 * s1 - P22 * s2;
    const double val = v1 * v1 + v2 * v2 - r * r;
    if (val < 1e-10) break;
    const double deriv = -2 * (P11 * v1 * v1 + 2 * P12 * v1 * v2 + P22 * v2 * v2);
    const double delta = -val / deriv;
    if (delta < 1e-10) break;
    la += delta;
  }
  res[0] = v1 * d0;
  res[1] = v2 * d1;
  return la != 0;
}

// Spatial quantity S_b x of every body for a vector x over the dofs (LDS, NV entries): links by summing their ancestors'
// motion axes (lane i < NL), the box from its frame (lane NL), the world zero (lane NL + 1).  out: [NB][6] in LDS.
template <class T>
RCSH_D void body_spatial(const StageTeam<T>& st, const double* x, const double* boxR, const double* boxp, double (*out)[6], int lane) {
  constexpr int NL = T::NL;
  if (lane < NL) {
    double u[6] = {0, 0, 0, 0, 0, 0};
    for (int j = 0; j < NL; ++j) {
      if (!is_anc<T>(j, lane)) continue;
      const double xj = x[j];
#pragma unroll
      for (int k = 0; k < 6; ++k) u[k] += st.S(j, k) * xj;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) out[lane][k] = u[k];
  } else if (lane == NL) {
    double w[3], c[3];
    mulmv(boxR, x + NL + 3, w);  // angular part is in the body frame
    cross3(boxp, w, c);
#pragma unroll
    for (int k = 0; k < 3; ++k) { out[NL][k] = w[k]; out[NL][3 + k] = x[NL + k] + c[k]; }
  } else if (lane == NL + 1) {
#pragma unroll
    for (int k = 0; k < 6; ++k) out[NL + 1][k] = 0.0;
  }
}

// ---- the environment's box: frame of the current position stage (mj_kinematics normalises the quaternion) and velocity
RCSH_D void box_frame(const double* bs, double* bp, double* bR, double* bv) {
  double bq[4];
#pragma unroll
  for (int k = 0; k < 3; ++k) bp[k] = bs[kBoxQ + k];
#pragma unroll
  for (int k = 0; k < 4; ++k) bq[k] = bs[kBoxQ + 3 + k];
#pragma unroll
  for (int k = 0; k < 6; ++k) bv[k] = bs[kBoxV + k];
  const double n = sqrt(bq[0] * bq[0] + bq[1] * bq[1] + bq[2] * bq[2] + bq[3] * bq[3]);
  if (n < kMinVal) { bq[0] = 1; bq[1] = bq[2] = bq[3] = 0; }
  else if (fabs(n - 1) > kMinVal) {
    const double s = 1 / n;
#pragma unroll
    for (int k = 0; k < 4; ++k) bq[k] *= s;
  }
  const double w = bq[0], x = bq[1], y = bq[2], z = bq[3];
  bR[0] = w * w + x * x - y * y - z * z; bR[4] = w * w - x * x + y * y - z * z; bR[8] = w * w - x * x - y * y + z * z;
  bR[1] = 2 * (x * y - w * z); bR[2] = 2 * (x * z + w * y); bR[3] = 2 * (x * y + w * z);
  bR[5] = 2 * (y * z - w * x); bR[6] = 2 * (x * z - w * y); bR[7] = 2 * (y * z + w * x);
}
// column k of the box's spatial Jacobian about the world origin: linear dofs (0; e_k), body-frame angular dofs (R e; p x R e)
RCSH_D void box_column(const double* bR, const double* bp, int k, double* col) {
  if (k < 3) { col[0] = col[1] = col[2] = 0; col[3] = k == 0; col[4] = k == 1; col[5] = k == 2; }
  else {
    const double re[3] = {bR[k - 3], bR[3 + k - 3], bR[6 + k - 3]};
    col[0] = re[0]; col[1] = re[1]; col[2] = re[2];
    cross3(bp, re, col + 3);
  }
}

// One contact as its owner lane sees it: the map G from spatial motion about the world origin to contact-frame velocity
// (rows: normal, two tangents), reference accelerations, regularisers, cone parameters, the two bodies (force +f on B, -f on A).
// Rebuilt from the contact's LDS record by every phase, so that nothing has to survive in registers between the phases.
struct ConLane {
  double G[3][6], aref[3], D[3], Rr[3], mu, fr, f[3];
  int A, B;
  bool on;
};
// record layout (ContactArena::rec): position 3, normal 3, [6] distance -> R0 once the rows exist, [7] friction of the pair,
// [8] inverse weight of the pair -> [8..10] reference accelerations, [11..13] force
template <class AR>
RCSH_D void con_load(const AR& ar, const BoxCfg& b, int lane, int ncon, int world, ConLane& c) {
  c.on = lane < ncon;
  c.A = world; c.B = world;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    c.aref[k] = 0; c.D[k] = 0; c.Rr[k] = 0; c.f[k] = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) c.G[k][j] = 0.0;
  }
  c.mu = 0; c.fr = 0;
  if (!c.on) return;
  const double* r = ar.rec[lane];
  const double pos[3] = {r[0], r[1], r[2]}, n[3] = {r[3], r[4], r[5]};
  double fk[3][3];
  fk[0][0] = n[0]; fk[0][1] = n[1]; fk[0][2] = n[2];
  make_frame(n, fk[1], fk[2]);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    double xf[3];
    cross3(pos, fk[k], xf);
#pragma unroll
    for (int j = 0; j < 3; ++j) { c.G[k][j] = xf[j]; c.G[k][3 + j] = fk[k][j]; }
  }
  const double R0 = r[6], R1 = R0 * b.inv_impratio;
  c.Rr[0] = R0; c.Rr[1] = R1; c.Rr[2] = R1;
  c.fr = r[7];
  c.mu = c.fr * sqrt(R1 / R0);
#pragma unroll
  for (int k = 0; k < 3; ++k) { c.D[k] = 1 / c.Rr[k]; c.aref[k] = r[8 + k]; c.f[k] = r[11 + k]; }
  c.A = ar.cb[lane] & 0xff;
  c.B = (ar.cb[lane] >> 8) & 0xff;
}

// Wrench of every body from the contact forces `fc` of the owner lanes -> ar.W; returns the generalised force J' f of dof
// `lane` (lanes < NV).  Contains barriers: every lane calls it.
// contacts of body lane / 6, as 64-bit lane masks (the same for every call of a phase)
struct BodyMasks { uint64_t asB, asA; };
template <int NBODY>
RCSH_D BodyMasks body_masks(const ConLane& c, int lane) {
  BodyMasks bm{0, 0};
  const int bdy = lane / 6;
#pragma unroll
  for (int bb = 0; bb < NBODY; ++bb) {
    const uint64_t mb = __ballot(c.on && c.B == bb), ma = __ballot(c.on && c.A == bb);
    if (bdy == bb) { bm.asB = mb; bm.asA = ma; }
  }
  return bm;
}
template <class T, class AR>
RCSH_D double contact_qfrc(AR& ar, const StageTeam<T>& st, const ConLane& c, const double* fc, const BodyMasks& bm, const double* bR, const double* bp, int lane) {
  constexpr int NL = T::NL, NV = NL + 6, NB = NL + 2, kBox = NL;
  if (lane < AR::kCap) {
    double* wr = ar.stage[lane];
#pragma unroll
    for (int k = 0; k < 6; ++k) wr[k] = c.on ? c.G[0][k] * fc[0] + c.G[1][k] * fc[1] + c.G[2][k] * fc[2] : 0.0;
  }
  __syncthreads();
  {
    // lane (body, component): sum over the body's contacts in order (bm: the contacts with the body as B / as A)
    const int bdy = lane / 6, k = lane % 6;
    if (bdy < NB - 1) {  // (the world takes no force)
      double s = 0;
      for (uint64_t m = bm.asB | bm.asA; m; m &= m - 1) {
        const int cc = __ffsll((long long)m) - 1;
        const double v = ar.stage[cc][k];
        if (bm.asB >> cc & 1) s += v;
        if (bm.asA >> cc & 1) s -= v;
      }
      ar.W[bdy][k] = s;
    }
  }
  __syncthreads();
  double q = 0;
  if (lane < NL) {
    double w[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < NL; ++i) {
      if (!is_anc<T>(lane, i)) continue;
#pragma unroll
      for (int k = 0; k < 6; ++k) w[k] += ar.W[i][k];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) q += st.S(lane, k) * w[k];
  } else if (lane < NV) {
    double col[6];
    box_column(bR, bp, lane - NL, col);
    q = dot6(col, ar.W[kBox]);
  }
  return q;
}

// Second level of the broad phase, run by the lane of a link whose bounding box reached the floor or the cube: the same
// tests on the bounding boxes of the link's GEOMS (boxes: the geom itself; hulls: the box of its vertices; capsule: its box).
// Only then does the wavefront gang up on the environment.  R, p: world frame of the link.
// `boxs`: the cube's LDS block (its pose and velocity), `box_half`: its half sizes.  Against the cube the geom's box must come
// within the cube's bounding sphere and then overlap the cube itself (separating axes of the two boxes): fingers that
// straddle the cube without touching it leave the wavefront alone.
RCSH_CONTACT_FN bool geom_level_near(const ContactGeom* geoms, int g0, int g1, const double* R, const double* p, const double* pln, double pld,
                                     bool has_plane, const double* boxs, double box_r2, bool has_box, const double* box_half) {
  const double* boxc = boxs + kBoxQ;
  double bp[3], bR[9], bv[6];
  bool have_frame = false;
  for (int g = g0; g < g1; ++g) {
    const ContactGeom& cg = geoms[g];
    double gR[9], c[3], h[3], lc[3];
    mulmm(R, cg.rot, gR);
    if (cg.type == 7) { for (int k = 0; k < 3; ++k) { lc[k] = cg.aabb_c[k]; h[k] = cg.aabb_h[k]; } }
    else if (cg.type == 6) { for (int k = 0; k < 3; ++k) { lc[k] = 0; h[k] = cg.size[k]; } }
    else { lc[0] = lc[1] = lc[2] = 0; h[0] = h[1] = cg.size[0]; h[2] = cg.size[0] + cg.size[1]; }
    double off[3], w[3];
    mulmv(cg.rot, lc, off);
    for (int k = 0; k < 3; ++k) off[k] += cg.pos[k];
    mulmv(R, off, w);
    for (int k = 0; k < 3; ++k) c[k] = w[k] + p[k];
    if (has_plane && cg.plane_ok) {
      double nl[3];
      mulTv(gR, pln, nl);
      if (dot3(pln, c) - pld - (fabs(nl[0]) * h[0] + fabs(nl[1]) * h[1] + fabs(nl[2]) * h[2]) <= 0) return true;
    }
    if (has_box && !(cg.type == 7 && cg.vert_num == 0)) {
      const double d[3] = {boxc[0] - c[0], boxc[1] - c[1], boxc[2] - c[2]};
      double v[3];
      mulTv(gR, d, v);
      const double ex = fmax(fabs(v[0]) - h[0], 0.0), ey = fmax(fabs(v[1]) - h[1], 0.0), ez = fmax(fabs(v[2]) - h[2], 0.0);
      if (ex * ex + ey * ey + ez * ez <= box_r2) {
        if (!have_frame) { box_frame(boxs, bp, bR, bv); have_frame = true; }
        if (!obb_disjoint(gR, c, h, bR, bp, box_half)) return true;
      }
    }
  }
  return false;
}

// ================================================================= phase 1: collision
// Lane g tests collision geom g against the floor and the box; the contacts are compacted into MuJoCo's order -- (floor,
// robot geoms) by geom, (floor, box), (robot geoms, box) by geom -- and left as records in ar.rec / ar.cb.
// Returns bit 0: a robot geom is in contact; bits 8-9: contact classes of this position stage (arm / gripper collision geoms).
// "In contact" for two geoms of the robot: penetrating by more than a nanometre (check_team.h: kCheckTouch, where the reason is
// written down; the oracle's self_collide uses the same bar)
constexpr double kSelfTouch = 1e-9;
constexpr double kNewtonRel = 2e-12;  // oracle: ORC_NEWTON_REL
RCSH_D int contact_key(int b1, int b2, int g1, int g2, int k) { return (b1 << 23) | (b2 << 18) | (g1 << 13) | (g2 << 8) | k; }
constexpr int kKeyBox = 31;  // the free box: the scene's last body, its last geom (reference assets/scenes/fr3_simple_pick_up/scene.xml:30-33)
template <class T, class AR>
RCSH_CONTACT_FN uint32_t contact_collide(const ContactTable& tab_, const CheckTable& ck_, const BoxCfg& b_, const LinkRec* links_, const StageTeam<T>& st_,
                                         const double* bs_, AR& ar_, int env, bool frames_ready = false) {
  constexpr int kMaxCon = AR::kCap;  // (this arena's capacity)
  const ContactTable& tab = *in_lds(&tab_);
  const CheckTable& ck = *in_lds(&ck_);
  const BoxCfg& b = *in_lds(&b_);
  const LinkRec* links = in_lds(links_);
  const StageTeam<T> st{in_lds(st_.base)};
  const double* bs = in_lds(bs_);
  AR& ar = *in_lds(&ar_);
  constexpr int NL = T::NL;
  constexpr int kBox = NL, kWorld = NL + 1;
  const int lane = wave_lane();
  double (*arF)[12] = ar.frames();
  int32_t* arkey = ar.keyp();
  TEAM_MARK(24)
  TEAM_COUNT(33)
  const bool self_on = (b.resolve & 2) && ck.npair > 0;  // (wave-uniform) contacts between two geoms of the robot are resolved
  // ---- which geom pairs (and which geoms against the floor) have to be looked at AT ALL in this substep: a pair found apart by a gap g
  // cannot touch before the joints BETWEEN its two links have moved the geoms by g, a geom found above the floor by h not before the
  // joints on its chain have moved it by h (ContactTable::self_lever turns joint motion into a bound on that; the lean DET kernels'
  // SelfSlack, here per environment in memory because the state has to outlive the launch).  rem[i] / remf: what is left of pair i's gap
  // / of the lane's geom's height -- lower bounds of the distances NOW; <= 0: look.  An escalated environment that touches nothing (six
  // of seven, at any substep of the headline rollout) has every gap left and leaves here: no link frames, no geom tests -- 39.5k cycles
  // of every substep before.  Exact: only what is proven apart is skipped (oracle self_collide / plane tests look at everything in every
  // substep and find the same contacts).
  constexpr int kDp = 12 * kMaxCGeom;  // this substep's joint motion, behind the world boxes
  static_assert(AR::kScratch >= kDp + 12, "joint motion fits behind the world boxes");
  float rem[3] = {0.0f, 0.0f, 0.0f}, remf = 0.0f;
  CheckEntry ent[3] = {};
  uint32_t am = 0;  // bit j: pair lane + 64 j has used its slack up
  float* const remg = self_on && ck.slack ? ck.slack + (size_t)env * kSlackStride : nullptr;
  if (self_on) {
    double* wb = ar.scratch();
    if (remg) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int i = lane + 64 * j;
        rem[j] = i < ck.npair ? remg[i] : 1.0f;
        ent[j] = ck.ent[i < ck.npair ? i : 0];
      }
      remf = lane < tab.ngeom ? remg[kSlackFloor + lane] : 1.0f;
      double* qpg = reinterpret_cast<double*>(remg + kMaxCheckPairs);
      if (lane < NL) {
        const double q = st.q(lane), qp = qpg[lane];
        qpg[lane] = q;
        wb[kDp + lane] = fabs(q - qp);
      }
      // moved[l][c]: how far this substep's joint motion can have moved a point of a geom ON link l relative to the frame of link c (an
      // ancestor-or-self of l; c = -1: the world) -- the joints of l's chain below c, each with ITS lever for link l (CheckTable::lev).
      // A pair is charged moved[la][c] + moved[lb][c], c the deepest common ancestor of its two links; a geom above the floor moved[l][-1].
      constexpr int kMv = NL * (NL + 1);
      double* mv = &ar.stage[0][0];
      static_assert(sizeof(ar.stage) >= sizeof(double) * kMv && kMv <= 128, "the table fits the stage area, two entries a lane");
      float lv[2][NL];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = lane + 64 * u, l = e < kMv ? e / (NL + 1) : 0;
#pragma unroll
        for (int jn = 0; jn < NL; ++jn) lv[u][jn] = ck.lev[12 * jn + l];
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = lane + 64 * u;
        if (e < kMv) {
          const int l = e / (NL + 1), c = e % (NL + 1) - 1;
          const uint32_t jm = anc_mask<T>(l) & ~anc_mask<T>(c);
          double m = 0.0;
#pragma unroll
          for (int jn = 0; jn < NL; ++jn) m += (jm >> jn) & 1u ? wb[kDp + jn] * (double)lv[u][jn] : 0.0;
          mv[e] = m;
        }
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int g0 = ent[j].geoms & 0xff, g1 = (ent[j].geoms >> 8) & 0xff, c = (int)((ent[j].geoms >> 16) & 0xff);
        const int la = ck.glink[g0], lb = ck.glink[g1];
        const double moved = (la >= 0 ? mv[la * (NL + 1) + c] : 0.0) + (lb >= 0 ? mv[lb * (NL + 1) + c] : 0.0);
        // (rounded up, and by more than the subtraction's own rounding: rem stays a lower bound)
        if (moved > 0.0) rem[j] -= (float)moved * 1.00001f + 2.5e-7f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 3; ++j) ent[j] = ck.ent[lane + 64 * j < ck.npair ? lane + 64 * j : 0];
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) am |= (lane + 64 * j < ck.npair && !(rem[j] > 0.0f)) ? 1u << j : 0u;
    if (remg && tab.has_plane && !b.present) {
      // the lane's geom against the floor: its height above it less its bounding radius, less what its chain has moved it since
      const bool mine = lane < tab.ngeom;
      {
        const int lk = mine ? ck.glink[lane] : -1;
        const double moved = lk >= 0 ? (&ar.stage[0][0])[lk * (NL + 1)] : 0.0;  // (moved[l][-1]: against the world)
        if (moved > 0.0) remf -= (float)moved * 1.00001f + 2.5e-7f;
      }
#ifdef RCSH_PHASE_TIMING
      {  // why an escalated environment's pass is not quiet: passes / with a pair due / with a geom due against the floor / an example
        const uint64_t wa = __ballot(am != 0), wf = __ballot(mine && !(remf > 0.0f));
        if (lane == 0) {
          atomicAdd(&g_team_cycles[92], 1ull);
          if (wa) atomicAdd(&g_team_cycles[93], 1ull);
          if (wf) atomicAdd(&g_team_cycles[94], 1ull);
        }
        if (wa && lane == __ffsll((long long)wa) - 1) {
          const int jb = __ffs((int)am) - 1;
          g_slack_dbg[0] = lane + 64 * jb; g_slack_dbg[1] = jb == 0 ? rem[0] : (jb == 1 ? rem[1] : rem[2]);
          g_slack_dbg[2] = (double)(jb == 0 ? ent[0].geoms : (jb == 1 ? ent[1].geoms : ent[2].geoms));
          g_slack_dbg[3] = remg[lane + 64 * jb];  // (what the last pass left)
          g_slack_dbg[4] = env;
        }
        if (wa && lane == __ffsll((long long)wa) - 1) g_team_cycles[95] = (g_team_cycles[95] / 1000ull) * 1000ull + (unsigned long long)(lane + 64 * (__ffs((int)am) - 1) + 1);
        if (wf && lane == __ffsll((long long)wf) - 1) g_team_cycles[95] = (g_team_cycles[95] % 1000ull) + 1000ull * (unsigned long long)(lane + 1);
      }
#endif
      if (__ballot(am != 0 || (mine && !(remf > 0.0f))) == 0) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
          if (lane + 64 * j < ck.npair) remg[lane + 64 * j] = rem[j];
        if (mine) remg[kSlackFloor + lane] = remf;
        TEAM_MARK(25)
        return 0u;
      }
    }
    __syncthreads();  // (the scratch area goes to the stages below)
  }
  // ---- link frames at the pre-step configuration: every lane its own joint's local frame, composed down the chain by the position
  // stage's scan (three or four rounds across the lanes; a lane walking the chain to its link alone took 6k cycles of every pass)
  // (frames_ready, wave-uniform: the substep's position stage has put them there already -- the same frames from the same functions --:
  // the contact-resolving launch of per-environment escalation, whose wavefront has one environment and nothing else in the arena
  // between the position stage and this pass; sim_kernels.h)
  if (!frames_ready) {
    const int tl = lane < NL ? lane : NL - 1;
    KinK kk;
    kk.load(links[tl]);
    double R[9], p[3];
    link_local_frame(kk, st.q(tl), R, p);
    scan_frames<T>(R, p);
    if (lane < NL) {
#pragma unroll
      for (int k = 0; k < 9; ++k) arF[lane][k] = R[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) arF[lane][9 + k] = p[k];
    }
  }
  double bp[3], bR[9], bv[6];
  box_frame(bs, bp, bR, bv);
  __syncthreads();
  TEAM_MARK(37)  // (slots 37-39 here: the contact timing tool; the self-collision marks of the same slots are in kernels it does not launch)
  double ppos[4][3], pdist[4], cpos[8][3], cdist[8], cn[3] = {0, 0, 0};
  int nP = 0, nB = 0, boxfirst = 0;
  bool want_plane = false, want_box = false;  // hull geoms: vertex work left for the cooperative stage
  ContactGeom cg;
  const bool has_geom = lane < tab.ngeom;
  // (a pass that is here for a geom PAIR only -- every geom still has height left above the floor -- skips the floor tests: nothing can
  // touch it, and the heights keep what the slack test left of them)
  const bool floor_due = !remg || b.present || __ballot(has_geom && !(remf > 0.0f)) != 0;
  if (has_geom) {
    cg = tab.geoms[lane];
    double Rl[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pl[3] = {0, 0, 0};
    if (cg.link >= 0) {
#pragma unroll
      for (int k = 0; k < 9; ++k) Rl[k] = arF[cg.link][k];
#pragma unroll
      for (int k = 0; k < 3; ++k) pl[k] = arF[cg.link][9 + k];
    }
    double gR[9], gp[3];
    mulmm(Rl, cg.rot, gR);
    mulmv(Rl, cg.pos, gp);
    gp[0] += pl[0]; gp[1] += pl[1]; gp[2] += pl[2];
    const double* V = tab.verts + 3 * (size_t)cg.vert_adr;
    // ---- floor
    if (floor_due) remf = 1e30f;  // (a geom the floor does not collide with)
    if (tab.has_plane && cg.plane_ok && floor_due) {
      const double n[3] = {tab.plane_n[0], tab.plane_n[1], tab.plane_n[2]};
      const double cdst = dot3(n, gp) - tab.plane_d;
      {
        // the geom's slack against the floor, measured anew: the lowest point of its oriented bounding box (the bounding sphere of
        // link 1's hull reaches the floor in every pose, its box in none)
        double lc3[3], hh3[3], oc3[3], nl3[3];
        geom_obb(cg, lc3, hh3);
        mulmv(gR, lc3, oc3);
        mulTv(gR, n, nl3);
        const double low = cdst + dot3(n, oc3) - (fabs(nl3[0]) * hh3[0] + fabs(nl3[1]) * hh3[1] + fabs(nl3[2]) * hh3[2]);
        remf = fmaxf((float)low * 0.999999f - 1e-6f, 0.0f);
      }
      if (cdst - cg.rbound <= 0) {
        if (cg.type == 7) {
          // (a hull's support vertices: with the whole wavefront, below -- if the hull's bounding box reaches the floor at all: link 1's
          // bounding sphere does in every pose, its box does not, and the staging of its 100-odd vertices cost every substep 11k cycles)
          double oc[3], nl[3];
          mulmv(gR, cg.aabb_c, oc);
          mulTv(gR, n, nl);
          const double ext = fabs(nl[0]) * cg.aabb_h[0] + fabs(nl[1]) * cg.aabb_h[1] + fabs(nl[2]) * cg.aabb_h[2];
          want_plane = cdst + dot3(n, oc) - ext <= 0;
        } else if (cg.type == 6) {
          for (int c = 0; c < 8 && nP < 4; ++c) {
            const double loc[3] = {(c & 1 ? cg.size[0] : -cg.size[0]), (c & 2 ? cg.size[1] : -cg.size[1]), (c & 4 ? cg.size[2] : -cg.size[2])};
            double w[3];
            mulmv(gR, loc, w);
            const double ld = dot3(n, w);
            if (cdst + ld > 0 || ld > 0) continue;
            const double dist = cdst + ld;
            for (int k = 0; k < 3; ++k) ppos[nP][k] = w[k] + gp[k] - n[k] * dist * 0.5;
            pdist[nP] = dist;
            ++nP;
          }
        } else if (cg.type == 3 || cg.type == 2) {
          for (int e = 0; e < (cg.type == 3 ? 2 : 1); ++e) {
            const double loc[3] = {0, 0, cg.type == 3 ? (e ? -cg.size[1] : cg.size[1]) : 0};
            double w[3];
            mulmv(gR, loc, w);
            const double c[3] = {w[0] + gp[0], w[1] + gp[1], w[2] + gp[2]};
            const double dist = dot3(n, c) - tab.plane_d - cg.size[0];
            if (dist >= 0) continue;
            for (int k = 0; k < 3; ++k) ppos[nP][k] = c[k] - n[k] * (cg.size[0] + dist * 0.5);
            pdist[nP] = dist;
            ++nP;
          }
        }
      }
    }
    // ---- the box
    if (!(cg.type == 7 && cg.vert_num == 0)) {
      const double dx[3] = {gp[0] - bp[0], gp[1] - bp[1], gp[2] - bp[2]};
      const double rsum = cg.rbound + sqrt(b.size[0] * b.size[0] + b.size[1] * b.size[1] + b.size[2] * b.size[2]);
      if (dot3(dx, dx) <= rsum * rsum) {
        const double bsz[3] = {b.size[0], b.size[1], b.size[2]};
        if (cg.type == 6) {
          nB = dev_box_box(gp, gR, cg.size, bp, bR, bsz, &cpos[0][0], cn, cdist, &ar.stage[0][0] + 48 * cg.box_slot);
        } else {
          Shape S = make_shape(cg.type == 7 ? 0 : 2, gp, gR, cg.size, V, cg.vert_num);
          Shape Bx = make_shape(1, bp, bR, bsz, nullptr, 0);
          if (cg.type == 7) {
            double c[3];
            mulmv(gR, cg.center, c);
            for (int k = 0; k < 3; ++k) S.center[k] = c[k] + gp[k];
          }
          // the shapes cannot intersect when the hull's bounding box and the cube do not (separating axes of the two boxes)
          bool may = true;
          if (cg.type == 7) {
            double oc[3];
            mulmv(gR, cg.aabb_c, oc);
            oc[0] += gp[0]; oc[1] += gp[1]; oc[2] += gp[2];
            may = dev_box_box(oc, gR, cg.aabb_h, bp, bR, bsz, nullptr, nullptr, nullptr, nullptr) != 0;
          }
          double depth = 0;
          if (may) {
            if (cg.type == 7) want_box = true;  // (the hull's portal refinement: with the whole wavefront, below)
            else nB = dev_mpr(S, Bx, &depth, cn, cpos[0]);
          }
          cdist[0] = -depth;
        }
      }
    }
  }
  // ---- the hulls' vertex work, one hull at a time with the whole wavefront: a lane scanning its hull's vertices in global
  // memory alone pays a memory round trip per vertex (its wavefront has nothing else to run); staged into LDS once, the
  // lanes of every 16-lane team scan every 16th vertex and agree on the winner -- the same vertex the serial scan finds.
  {
    __syncthreads();  // (the pads' clipping polygons in ar.stage are done with)
    TEAM_MARK(38)
    double* hv = &ar.stage[0][0];
    static_assert(sizeof(ar.stage) >= sizeof(double) * 3 * 160, "a hull's vertices fit into the stage");
    for (uint64_t todo = __ballot(has_geom && (want_plane || want_box)); todo; todo &= todo - 1) {
      const int h = __ffsll((long long)todo) - 1;
      const ContactGeom& hg = tab.geoms[h];
      const bool do_plane = (__ballot(want_plane) >> h) & 1, do_box = (__ballot(want_box) >> h) & 1;
      const int nv = hg.vert_num;
      {
        const double* Vh = tab.verts + 3 * (size_t)hg.vert_adr;
        for (int k = lane; k < 3 * nv; k += 64) hv[k] = Vh[k];
        __syncthreads();
      }
      double Rl[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pl[3] = {0, 0, 0};
      if (hg.link >= 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) Rl[k] = arF[hg.link][k];
#pragma unroll
        for (int k = 0; k < 3; ++k) pl[k] = arF[hg.link][9 + k];
      }
      double gR[9], gp[3];
      mulmm(Rl, hg.rot, gR);
      mulmv(Rl, hg.pos, gp);
      gp[0] += pl[0]; gp[1] += pl[1]; gp[2] += pl[2];
      // every lane computes the same; lane h keeps it
      int hP = 0;
      double hpos[4][3], hdist[4];
      if (do_plane) {
        const double n[3] = {tab.plane_n[0], tab.plane_n[1], tab.plane_n[2]};
        double t1[3], t2[3];
        make_frame(n, t1, t2);
        int chosen[4];
        for (int q = 0; q < 4; ++q) {
          double dir[3];
          if (q == 0) { dir[0] = -n[0]; dir[1] = -n[1]; dir[2] = -n[2]; }
          else {
            // cos / sin of 2 pi (q - 1) / 3 as the C library rounds them (the oracle calls it)
            const double kc[3] = {1.0, -0.4999999999999998, -0.5000000000000004}, ks[3] = {0.0, 0.8660254037844387, -0.8660254037844384};
            const double cs = 1e-3 * kc[q - 1], sn = 1e-3 * ks[q - 1];
            for (int k = 0; k < 3; ++k) dir[k] = -n[k] + cs * t1[k] + sn * t2[k];
          }
          double dl[3];
          mulTv(gR, dir, dl);
          if (nv <= 0) break;
          const int bi = hull_support_index<true>(hv, nv, dl);
          bool dup = false;
          for (int k = 0; k < hP; ++k) dup = dup || chosen[k] == bi;
          if (dup) continue;
          const double* hvl = in_lds(hv);
          const double vl[3] = {hvl[3 * bi], hvl[3 * bi + 1], hvl[3 * bi + 2]};
          double w[3];
          mulmv(gR, vl, w);
          const double xw[3] = {w[0] + gp[0], w[1] + gp[1], w[2] + gp[2]};
          const double dist = dot3(n, xw) - tab.plane_d;
          if (dist >= 0) { if (q == 0) break; else continue; }
          chosen[hP] = bi;
          for (int k = 0; k < 3; ++k) hpos[hP][k] = xw[k] - n[k] * dist * 0.5;
          hdist[hP] = dist;
          ++hP;
        }
      }
      int hB = 0;
      double hdepth = 0, hn[3] = {0, 0, 0}, hc[3] = {0, 0, 0};
      if (do_box) {
        const double bsz[3] = {b.size[0], b.size[1], b.size[2]};
        Shape S = make_shape(0, gp, gR, hg.size, hv, nv);
        Shape Bx = make_shape(1, bp, bR, bsz, nullptr, 0);
        double c[3];
        mulmv(gR, hg.center, c);
        for (int k = 0; k < 3; ++k) S.center[k] = c[k] + gp[k];
        hB = mpr_penetration<true>(Bx, S, &hdepth, hn, hc);  // box (type 6) before mesh (type 7)
      }
      if (lane == h) {
        if (do_plane) {
          nP = hP;
          for (int k = 0; k < 4; ++k) { pdist[k] = hdist[k]; ppos[k][0] = hpos[k][0]; ppos[k][1] = hpos[k][1]; ppos[k][2] = hpos[k][2]; }
        }
        if (do_box) {
          nB = hB; boxfirst = 1;
          cdist[0] = -hdepth;
          for (int k = 0; k < 3; ++k) { cn[k] = hn[k]; cpos[0][k] = hc[k]; }
        }
      }
      __syncthreads();
    }
    TEAM_MARK(39)
  }
  // the box against the floor (mjc_PlaneBox: corners in order, at most four)
  int nBP = 0;
  double bppos[4][3], bpdist[4];
  if (tab.has_plane) {
    const double dist = bp[2] - b.plane_z;
    for (int i = 0; i < 8 && nBP < 4; ++i) {
      const double vec[3] = {i & 1 ? b.size[0] : -b.size[0], i & 2 ? b.size[1] : -b.size[1], i & 4 ? b.size[2] : -b.size[2]};
      double corner[3];
      mulmv(bR, vec, corner);
      const double ld = corner[2];
      if (dist + ld > 0 || ld > 0) continue;
      bpdist[nBP] = dist + ld;
      bppos[nBP][0] = corner[0] + bp[0]; bppos[nBP][1] = corner[1] + bp[1]; bppos[nBP][2] = corner[2] + bp[2] - 0.5 * (dist + ld);
      ++nBP;
    }
  }
  if (!self_on && __ballot(nP > 0 || nB > 0) == 0) {  // nothing of the robot touches anything: the fast path keeps the step
    TEAM_MARK(25)
    return 0u;
  }
  if (lane < 32) { ar.cnt[lane][0] = nP; ar.cnt[lane][1] = nB; }
  __syncthreads();
  TEAM_MARK(25)
  int offP = 0, offB = 0, totP = 0, totB = 0;
  for (int g = 0; g < tab.ngeom; ++g) {
    const int a = ar.cnt[g][0], c = ar.cnt[g][1];
    if (g < lane) { offP += a; offB += c; }
    totP += a; totB += c;
  }
  offB += totP + nBP;
  const int nreg_all = totP + nBP + totB;
  const int nreg = nreg_all > kMaxCon ? kMaxCon : nreg_all;
  bool too_many_contacts = nreg_all > kMaxCon;  // (the tail of MuJoCo's contact order is dropped: flagged, kContactOverflow)
  // ---- the robot's geoms against each other (round 5: self contact carries constraint rows, as every entry of mjData.contact does
  // in mj_step2, reference src/sim/sim.cpp:108-115).  Oracle: self_collide (rcs_contact.c) -- bounding spheres, oriented boxes, then
  // mjc_BoxBox for two boxes (the two fingers' pads), MPR for everything else: one point, normal from geom[0] to geom[1] (MuJoCo's
  // order within a pair: by type, then by id -- the pair table's).  The contacts are parked at the END of the record area and
  // merged into MuJoCo's order below.
  int nS = 0;  // (wave-uniform)
  if (self_on) {
    static_assert(AR::kScratch >= 12 * kMaxCGeom && AR::kScratch >= 3 * 152, "world boxes of the geoms / one hull fit the scratch area");
    double* wb = ar.scratch();
    // ---- which pairs have to be looked at: a pair found apart by a gap g cannot touch before the joints BETWEEN its two links have moved
    // the geoms by g (ContactTable::self_lever turns joint motion into a bound on that; the lean DET kernels' SelfSlack, here per
    // environment in memory because the state has to outlive the launch).  rem[i]: what is left of pair i's gap -- a lower bound of
    // the geoms' distance NOW; <= 0: look.  In a settled or slowly moving arm nearly every pair is skipped, every substep: the world
    // boxes, the sphere and box tests and the narrow phase run for the few pairs that are near.  Exact: only pairs proven apart are
    // skipped (oracle self_collide tests every pair in every substep and finds the same contacts).
    TEAM_MARK(16)  // (slots 16-18, 21-23 here: box-less scenes; box_team.h uses them in kernels with a free box)
    uint32_t cm = 0;  // bit j: pair lane + 64 j survived the bounding tests
    uint32_t hot = 0;  // bit j: ... and was in contact the last time it was looked at (rem = -1): Gilbert's iteration would only fail again
    static_assert(kMaxCheckPairs <= 3 * 64, "three pairs per lane");
    if (__ballot(am != 0)) {
    if (has_geom) {
      double Rl[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pl[3] = {0, 0, 0};
      if (cg.link >= 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) Rl[k] = arF[cg.link][k];
#pragma unroll
        for (int k = 0; k < 3; ++k) pl[k] = arF[cg.link][9 + k];
      }
      double lc[3], hh[3], off[3], cw[3], gR[9];
      geom_obb(cg, lc, hh);
      mulmv(cg.rot, lc, off);
      off[0] += cg.pos[0]; off[1] += cg.pos[1]; off[2] += cg.pos[2];
      mulmv(Rl, off, cw);
      mulmm(Rl, cg.rot, gR);
#pragma unroll
      for (int k = 0; k < 3; ++k) wb[12 * lane + k] = cw[k] + pl[k];
#pragma unroll
      for (int k = 0; k < 9; ++k) wb[12 * lane + 3 + k] = gR[k];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (!((am >> j) & 1u)) continue;
      const CheckEntry en = ent[j];
      const int g0 = en.geoms & 0xff, g1 = (en.geoms >> 8) & 0xff;
      double Ra[9], Rb[9], ca[3], cb_[3], ha[3], hb[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) { ca[k] = wb[12 * g0 + k]; cb_[k] = wb[12 * g1 + k]; ha[k] = ck.gh[g0][k]; hb[k] = ck.gh[g1][k]; }
      const double dd[3] = {ca[0] - cb_[0], ca[1] - cb_[1], ca[2] - cb_[2]};
      const double rs = en.rsum, d2 = dot3(dd, dd);
      if (rem[j] <= -0.5f) hot |= 1u << j;
      rem[j] = 0.0f;  // (until one of the tests measures a gap)
      if (d2 > rs * rs) { rem[j] = (float)(sqrt(d2) - rs) * 0.999999f - 1e-6f; continue; }
#pragma unroll
      for (int k = 0; k < 9; ++k) { Ra[k] = wb[12 * g0 + 3 + k]; Rb[k] = wb[12 * g1 + 3 + k]; }
      double sep = 0.0;
      if (obb_disjoint(Ra, ca, ha, Rb, cb_, hb, &sep)) { rem[j] = (float)sep * 0.999999f - 1e-6f; continue; }
      // two boxes (their bounding boxes are the geoms themselves) that do not penetrate by more than the touching bar: no contact
      if (ck.gtype[g0] == 6 && ck.gtype[g1] == 6 && !(box_box_face_pen(ca, Ra, ha, cb_, Rb, hb) > kSelfTouch)) continue;
      cm |= 1u << j;
    }
    }
    __syncthreads();  // (the world boxes are done with: the scratch area stages hulls from here on)
    TEAM_MARK(17)
    // ---- pairs of two BOXES (the fingers' pads against each other: a gripper pressed shut brings two dozen of them, four to eight
    // points each): every lane runs the collider for its own pair -- the serial loop below spent 40k cycles of one lane per pair, 570k
    // per substep of such an environment.  A lane needs 48 doubles of LDS for its clipping polygons: kBoxPool lanes at a time.
    {
      constexpr int kPoolA = AR::kScratch / 48, kPoolB = (64 * 8) / 48, kBoxPool = kPoolA + kPoolB;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int g0 = ent[j].geoms & 0xff, g1 = (ent[j].geoms >> 8) & 0xff;
        bool want = ((cm >> j) & 1u) && ck.gtype[g0] == 6 && ck.gtype[g1] == 6;
        for (uint64_t wm = __ballot(want); wm; wm = __ballot(want)) {
          const int rank = __popcll(wm & ((1ull << lane) - 1ull));
          const bool run = want && rank < kBoxPool;
          int nc = 0;
          double spos[8][3], sn[3] = {0, 0, 0}, sdist[8];
          if (run) {
            TEAM_COUNT(21)
            const ContactGeom& ga = tab.geoms[g0];
            const ContactGeom& gb = tab.geoms[g1];
            double Ra[9], pa[3], Rb[9], pb[3];
            self_geom_world(ga, &arF[0][0], Ra, pa);
            self_geom_world(gb, &arF[0][0], Rb, pb);
            double* poly = rank < kPoolA ? ar.scratch() + 48 * rank : &ar.stage[0][0] + 48 * (rank - kPoolA);
            const int nb = dev_box_box(pa, Ra, ga.size, pb, Rb, gb.size, &spos[0][0], sn, sdist, poly);
            // (a point that touches exactly is no contact: kSelfTouch, oracle SELF_TOUCH)
            for (int k = 0; k < nb; ++k) {
              if (!(sdist[k] < -kSelfTouch)) continue;
              spos[nc][0] = spos[k][0]; spos[nc][1] = spos[k][1]; spos[nc][2] = spos[k][2];
              sdist[nc] = sdist[k];
              ++nc;
            }
          }
          // their places at the end of the record area: a running count over the lanes
          int incl = nc;
#pragma unroll
          for (int d = 1; d < 64; d <<= 1) {
            const int up = __builtin_amdgcn_ds_bpermute(((lane - d) & 63) << 2, incl);
            incl += lane >= d ? up : 0;
          }
          const int total = __builtin_amdgcn_readlane(incl, 63), first = nS + incl - nc;
          if (run && nc > 0) {
            const ContactGeom& ga = tab.geoms[g0];
            const ContactGeom& gb = tab.geoms[g1];
            int c2 = 0;  // what the collision callbacks make of the pair (rcs_hip.hip: list_geom_pairs)
            if ((ga.cls | gb.cls) & 1) c2 |= 1;
            if (!((ga.cls & 4) && (gb.cls & 4)) && ((ga.cls | gb.cls) & 16) && !(gb.cls & 8)) c2 |= 2;
            const int la = ga.link >= 0 ? ga.link : kWorld, lb = gb.link >= 0 ? gb.link : kWorld;
            const bool fwd = ga.body < gb.body;
            for (int k = 0; k < nc; ++k) {
              if (nreg + first + k >= kMaxCon) break;
              const int c = kMaxCon - 1 - (first + k);
              double* r = ar.rec[c];
              r[0] = spos[k][0]; r[1] = spos[k][1]; r[2] = spos[k][2];
              r[3] = sn[0]; r[4] = sn[1]; r[5] = sn[2];
              r[6] = sdist[k];
              r[7] = fmax(ga.mu, gb.mu);
              r[8] = ga.invweight + gb.invweight;
              ar.cb[c] = la | (lb << 8) | (c2 << 16);
              arkey[c] = fwd ? contact_key(ga.body, gb.body, g0 + 1, g1 + 1, k) : contact_key(gb.body, ga.body, g1 + 1, g0 + 1, k);
            }
          }
          int add = total;
          if (nreg + nS + add > kMaxCon) { too_many_contacts = true; add = kMaxCon - nreg - nS; }
          nS += add;
          if (run) { cm &= ~(1u << j); want = false; }
        }
      }
      __syncthreads();  // (the polygons' LDS goes back to the hulls)
    }
    TEAM_MARK(19)
    // ---- everything else: Gilbert's iteration, then the portal refinement.  The refinement is written for a TEAM of 16 lanes (they share
    // the vertex scans of a hull's support queries), and the wavefront has four: up to FOUR pairs are refined side by side, a team each --
    // the lead pair (the first one pending) and pending pairs that need no hull but the lead's, in the lead's places (a hand pressed
    // against link 0 or 1 brings its ten pad boxes and two finger hulls near that link's hull at once: twelve refinements of 40k cycles
    // one after the other were 500k of a substep's 800k).  Which pairs are refined together changes no pair's result.
    for (uint64_t pend = __ballot(cm != 0); pend; pend = __ballot(cm != 0)) {
      const int src = __ffsll((long long)pend) - 1;  // wave-uniform
      const uint32_t sm = (uint32_t)__builtin_amdgcn_readlane((int)cm, src);
      const int j = __ffs((int)sm) - 1;
      if (lane == src) cm &= ~(1u << j);
      const uint32_t eg_lead = j == 0 ? ent[0].geoms : (j == 1 ? ent[1].geoms : ent[2].geoms);
      const uint32_t gg_lead = (uint32_t)__builtin_amdgcn_readlane((int)eg_lead, src);
      const int g0 = gg_lead & 0xff, g1 = (gg_lead >> 8) & 0xff;
      const int ty0 = ck.gtype[g0], ty1 = ck.gtype[g1];
      if (ty0 == 6 && ty1 == 6) {
        // two boxes beyond the parallel stage's pool: one lane
        const ContactGeom& ga = tab.geoms[g0];
        const ContactGeom& gb = tab.geoms[g1];
        double Ra[9], pa[3], Rb[9], pb[3];
        self_geom_world(ga, &arF[0][0], Ra, pa);
        self_geom_world(gb, &arF[0][0], Rb, pb);
        int nc = 0;
        double spos[8][3], sn[3] = {0, 0, 0}, sdist[8];
        TEAM_COUNT(21)
        if (lane == 0) {
          const int nb = dev_box_box(pa, Ra, ga.size, pb, Rb, gb.size, &spos[0][0], sn, sdist, &ar.stage[0][0]);
          // (a point that touches exactly is no contact: kSelfTouch, oracle SELF_TOUCH)
          for (int k = 0; k < nb; ++k) {
            if (!(sdist[k] < -kSelfTouch)) continue;
            spos[nc][0] = spos[k][0]; spos[nc][1] = spos[k][1]; spos[nc][2] = spos[k][2];
            sdist[nc] = sdist[k];
            ++nc;
          }
        }
        nc = __builtin_amdgcn_readfirstlane(nc);
        if (nc > 0 && lane == 0) {
          int c2 = 0;  // what the collision callbacks make of the pair (rcs_hip.hip: list_geom_pairs)
          if ((ga.cls | gb.cls) & 1) c2 |= 1;
          if (!((ga.cls & 4) && (gb.cls & 4)) && ((ga.cls | gb.cls) & 16) && !(gb.cls & 8)) c2 |= 2;
          const int la = ga.link >= 0 ? ga.link : kWorld, lb = gb.link >= 0 ? gb.link : kWorld;
          const bool fwd = ga.body < gb.body;
          for (int k = 0; k < nc; ++k) {
            if (nreg + nS + k >= kMaxCon) break;
            const int c = kMaxCon - 1 - (nS + k);
            double* r = ar.rec[c];
            r[0] = spos[k][0]; r[1] = spos[k][1]; r[2] = spos[k][2];
            r[3] = sn[0]; r[4] = sn[1]; r[5] = sn[2];
            r[6] = sdist[k];
            r[7] = fmax(ga.mu, gb.mu);
            r[8] = ga.invweight + gb.invweight;
            ar.cb[c] = la | (lb << 8) | (c2 << 16);
            arkey[c] = fwd ? contact_key(ga.body, gb.body, g0 + 1, g1 + 1, k) : contact_key(gb.body, ga.body, g1 + 1, g0 + 1, k);
          }
        }
        if (nreg + nS + nc > kMaxCon) { too_many_contacts = true; nc = kMaxCon - nreg - nS; }
        nS += nc;
        continue;
      }
      // the lead's companions: up to three more pending pairs (not two boxes) whose hulls are the lead's, place by place
      int grp_src[4] = {src, -1, -1, -1}, grp_j[4] = {j, 0, 0, 0};
      uint32_t grp_g[4] = {gg_lead, 0u, 0u, 0u};
      int grp_hot[4] = {(__builtin_amdgcn_readlane((int)hot, src) >> j) & 1, 0, 0, 0};
      int ngrp = 1;
      {
        uint32_t ok = 0;
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
          const int c0 = ent[jj].geoms & 0xff, c1 = (ent[jj].geoms >> 8) & 0xff;
          const int u0 = ck.gtype[c0], u1 = ck.gtype[c1];
          const bool fits = (u0 != 7 || c0 == g0) && (u1 != 7 || c1 == g1) && !(u0 == 6 && u1 == 6);
          ok |= ((cm >> jj) & 1u) && fits ? 1u << jj : 0u;
        }
#pragma unroll
        for (int k = 1; k < 4; ++k) {
          const uint64_t w = __ballot(ok != 0);
          if (!w) break;
          const int L = __ffsll((long long)w) - 1;
          const int jj = __ffs(__builtin_amdgcn_readlane((int)ok, L)) - 1;
          const uint32_t eg = jj == 0 ? ent[0].geoms : (jj == 1 ? ent[1].geoms : ent[2].geoms);
          grp_g[k] = (uint32_t)__builtin_amdgcn_readlane((int)eg, L);
          grp_src[k] = L; grp_j[k] = jj;
          grp_hot[k] = (__builtin_amdgcn_readlane((int)hot, L) >> jj) & 1;
          if (lane == L) { ok &= ~(1u << jj); cm &= ~(1u << jj); }
          ngrp = k + 1;
        }
      }
      // the lead's hulls -> LDS (the companions' hulls are the same ones)
      double* hA = ar.scratch();
      double* hB = &ar.stage[0][0];
      {
        const ContactGeom& ga = tab.geoms[g0];
        const ContactGeom& gb = tab.geoms[g1];
        const int na3 = ty0 == 7 ? 3 * ga.vert_num : 0, nb3 = ty1 == 7 ? 3 * gb.vert_num : 0;
        const double* va = tab.verts + 3 * (size_t)ga.vert_adr;
        const double* vb = tab.verts + 3 * (size_t)gb.vert_adr;
        for (int k = lane; k < na3; k += 64) hA[k] = va[k];
        for (int k = lane; k < nb3; k += 64) hB[k] = vb[k];
      }
      __syncthreads();
      TEAM_MARK(20)
      const int team = lane >> 4;
      const bool mine_pair = team < ngrp;
      const uint32_t mg = team == 0 ? grp_g[0] : (team == 1 ? grp_g[1] : (team == 2 ? grp_g[2] : grp_g[3]));
      const int m0 = mine_pair ? (int)(mg & 0xff) : g0, m1 = mine_pair ? (int)((mg >> 8) & 0xff) : g1;
      const ContactGeom& ga = tab.geoms[m0];
      const ContactGeom& gb = tab.geoms[m1];
      int nc = 0;
      double spos0[3] = {0, 0, 0}, sn[3] = {0, 0, 0}, depth = 0.0, gap = 0.0;
      bool apart = false;
      TEAM_COUNT(22)
      if (mine_pair) {
        double Ra[9], pa[3], Rb[9], pb[3];
        self_geom_world(ga, &arF[0][0], Ra, pa);
        self_geom_world(gb, &arF[0][0], Rb, pb);
        const int ta = ga.type, tb = gb.type;
        Shape A = make_shape(ta == 7 ? 0 : ta == 6 ? 1 : 2, pa, Ra, ga.size, hA, ga.vert_num);
        Shape B = make_shape(tb == 7 ? 0 : tb == 6 ? 1 : 2, pb, Rb, gb.size, hB, gb.vert_num);
        if (ta == 7) { mulmv(Ra, ga.center, A.center); A.center[0] += pa[0]; A.center[1] += pa[1]; A.center[2] += pa[2]; }
        if (tb == 7) { mulmv(Rb, gb.center, B.center); B.center[0] += pb[0]; B.center[1] += pb[1]; B.center[2] += pb[2]; }
        // a few steps of Gilbert's iteration settle a pair that is clearly apart (links 5 and 7 wrap around the same wrist and pass the
        // boxes' test in every pose); everything else goes through the full refinement, as in the oracle
        const double x0[3] = {A.center[0] - B.center[0], A.center[1] - B.center[1], A.center[2] - B.center[2]};
        double dg[3];
        // (a pair that was in contact a substep ago goes straight to the refinement: five support queries saved)
        const int hot_mine = team == 0 ? grp_hot[0] : (team == 1 ? grp_hot[1] : (team == 2 ? grp_hot[2] : grp_hot[3]));
        // (more queries after the first proof, for a direction with a larger gap and so more slack, were tried in round 6: an escalated
        // environment is near contact, its pairs are due again whatever the direction proves -- 1.010 against 1.025 M on the 1000-step rollout)
        apart = hot_mine ? false : gilbert_apart<true, true>(A, B, x0, 5, 1e-5, dg, &gap);
        TEAM_MARK(61)
        if (!apart) {
          TEAM_COUNT(23)
          nc = mpr_penetration<true, kMprFull, true>(A, B, &depth, sn, spos0);
          // (apart after all: the refinement's separating direction proves a gap too -- without it the pair came back in every substep)
          if (nc == 0 && depth < 0.0) { apart = true; gap = -depth; }
        }
        if (!(depth > kSelfTouch)) nc = 0;
      }
      TEAM_MARK(62)
      // the pairs' slack (the gap Gilbert's direction proves) goes to the lanes that hold them; the contacts to the record area's end
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (k >= ngrp) break;  // (wave-uniform)
        const bool apart_k = __builtin_amdgcn_readlane((int)apart, 16 * k) != 0;
        const double gap_k = wave_read(gap, 16 * k);
        int nck = __builtin_amdgcn_readlane(nc, 16 * k);
        if ((apart_k || nck > 0) && lane == grp_src[k]) {
          const float gf = apart_k ? (float)gap_k * 0.999999f - 1e-6f : -1.0f;  // (-1: in contact -- due in every substep, and marked)
          const int jk = grp_j[k];
          rem[0] = jk == 0 ? gf : rem[0]; rem[1] = jk == 1 ? gf : rem[1]; rem[2] = jk == 2 ? gf : rem[2];
        }
        if (nck > 0 && lane == 16 * k && nreg + nS < kMaxCon) {
          int c2 = 0;  // what the collision callbacks make of the pair (rcs_hip.hip: list_geom_pairs)
          if ((ga.cls | gb.cls) & 1) c2 |= 1;
          if (!((ga.cls & 4) && (gb.cls & 4)) && ((ga.cls | gb.cls) & 16) && !(gb.cls & 8)) c2 |= 2;
          const int la = ga.link >= 0 ? ga.link : kWorld, lb = gb.link >= 0 ? gb.link : kWorld;
          const bool fwd = ga.body < gb.body;
          const int c = kMaxCon - 1 - nS;
          double* r = ar.rec[c];
          r[0] = spos0[0]; r[1] = spos0[1]; r[2] = spos0[2];
          r[3] = sn[0]; r[4] = sn[1]; r[5] = sn[2];
          r[6] = -depth;
          r[7] = fmax(ga.mu, gb.mu);
          r[8] = ga.invweight + gb.invweight;
          ar.cb[c] = la | (lb << 8) | (c2 << 16);
          arkey[c] = fwd ? contact_key(ga.body, gb.body, m0 + 1, m1 + 1, 0) : contact_key(gb.body, ga.body, m1 + 1, m0 + 1, 0);
        }
        if (nreg + nS + nck > kMaxCon) { too_many_contacts = true; nck = kMaxCon - nreg - nS; }
        nS += nck;
      }
      __syncthreads();  // (the stage is free again)
      TEAM_MARK(63)
    }
    if (remg) {
#pragma unroll
      for (int j = 0; j < 3; ++j)
        if (lane + 64 * j < ck.npair) remg[lane + 64 * j] = rem[j];
      if (has_geom) remg[kSlackFloor + lane] = remf;
    }
    TEAM_MARK(18)
  }
  const int robot_contacts = totP + totB + nS;
  if (robot_contacts == 0) {  // nothing of the robot touches anything: the fast path keeps the step
    return 0u;
  }
  int ncon = nreg + nS;
  if (has_geom) {
    const int lcode = cg.link >= 0 ? cg.link : kWorld;
    for (int k = 0; k < nP; ++k) {
      const int c = offP + k;
      if (c >= kMaxCon) break;
      double* r = ar.rec[c];
      r[0] = ppos[k][0]; r[1] = ppos[k][1]; r[2] = ppos[k][2];
      r[3] = tab.plane_n[0]; r[4] = tab.plane_n[1]; r[5] = tab.plane_n[2];
      r[6] = pdist[k];
      r[7] = fmax(tab.plane_mu, cg.mu);
      r[8] = cg.invweight;
      ar.cb[c] = kWorld | (lcode << 8) | (cg.cls << 16);
      arkey[c] = contact_key(0, cg.body, 0, lane + 1, k);
    }
    for (int k = 0; k < nB; ++k) {
      const int c = offB + k;
      if (c >= kMaxCon) break;
      double* r = ar.rec[c];
      r[0] = cpos[k][0]; r[1] = cpos[k][1]; r[2] = cpos[k][2];
      r[3] = cn[0]; r[4] = cn[1]; r[5] = cn[2];
      r[6] = cdist[k];
      r[7] = fmax(b.geom_mu, cg.mu);
      r[8] = cg.invweight + b.inv_mass;
      ar.cb[c] = (boxfirst ? (kBox | (lcode << 8)) : (lcode | (kBox << 8))) | (cg.cls << 16);
      arkey[c] = contact_key(cg.body, kKeyBox, lane + 1, kKeyBox, k);
    }
  }
  if (lane == 63) {
    for (int k = 0; k < nBP; ++k) {
      const int c = totP + k;
      if (c >= kMaxCon) break;
      double* r = ar.rec[c];
      r[0] = bppos[k][0]; r[1] = bppos[k][1]; r[2] = bppos[k][2];
      r[3] = 0; r[4] = 0; r[5] = 1;
      r[6] = bpdist[k];
      r[7] = b.fr;
      r[8] = b.inv_mass;
      ar.cb[c] = kWorld | (kBox << 8);
      arkey[c] = contact_key(0, kKeyBox, 0, kKeyBox, k);
    }
    ar.ncon = ncon;
    ar.pad[0] = too_many_contacts ? 1 : 0;  // capacity overflow of this phase (contact_newton adds the links' share)
  }
  __syncthreads();
  if (nS > 0) {
    // merge: the self contacts parked at the end of the record area take their places in mjData.contact's order (by body pair,
    // then by geom pair; the three groups written above are in that order among themselves).  Every contact through its lane's registers.
    const int srcc = lane < nreg ? lane : kMaxCon - 1 - (lane - nreg);
    double rr[9];
    int cbv = 0, kv = 0, rank = 0;
    if (lane < ncon) {
#pragma unroll
      for (int k = 0; k < 9; ++k) rr[k] = ar.rec[srcc][k];
      cbv = ar.cb[srcc];
      kv = arkey[srcc];
      for (int c = 0; c < ncon; ++c) rank += arkey[c < nreg ? c : kMaxCon - 1 - (c - nreg)] < kv ? 1 : 0;
    }
    __syncthreads();
    if (lane < ncon) {
#pragma unroll
      for (int k = 0; k < 9; ++k) ar.rec[rank][k] = rr[k];
      ar.cb[rank] = cbv;
    }
    __syncthreads();
  }
  // contact classes of this position stage (what the collision callbacks scan d->contact for).  SimGripper::collision_callback
  // ignores contacts between two finger geoms; none can occur here (no geom-geom pairs of the robot).
  int cls = 0;
  if (lane < ncon) cls = (ar.cb[lane] >> 16) & 0xff;
  const uint32_t hit = (__ballot(cls & 1) ? 1u : 0u) | (__ballot(cls & 2) ? 2u : 0u);
  TEAM_MARK(26)
  return (robot_contacts > 0 ? 1u : 0u) | (hit << 8);
}

// ================================================================= phase 2: rows + Newton on the primal cost
// Lane c owns contact c.  Leaves the minimiser in ar.X, qacc_smooth in ar.A0, the rows' reference accelerations /
// regularisers and the contact forces in the records.
// FRIC: the model has dry joint friction (xArm7): its rows -- one per arm joint, Huber cost: quadratic while
// |qacc_i - aref_i| < R frictionloss, linear with force -+frictionloss outside (oracle primal(): ORC_EFC_FRICTION) -- are rows of
// the coupled problem like the limit rows: a term of the lane's cost / gradient entry, of its diagonal Hessian entry and of
// phi', phi'' along the search line.
template <class T, bool FRIC, class AR>
RCSH_CONTACT_FN void contact_newton(const BoxCfg& b_, const StageTeam<T>& st_, double* bs_, AR& ar_, const double* gravity_,
                                    const LinkRec* links_) {
  const LinkRec* links = in_lds(links_);
  (void)links;
  const BoxCfg& b = *in_lds(&b_);
  const StageTeam<T> st{in_lds(st_.base)};
  double* bs = in_lds(bs_);
  AR& ar = *in_lds(&ar_);
  const double* gravity = in_lds(gravity_);
  constexpr int NL = T::NL, NA = T::NARM, NV = NL + 6;
  constexpr int kBox = NL, kWorld = NL + 1;
  const int lane = wave_lane();
  TEAM_COUNT(34)
  const int ncon = ar.ncon;
  double bp[3], bR[9], bv[6];
  box_frame(bs, bp, bR, bv);
  // ---- spatial velocities of the bodies (qvel parked in ar.X)
  if (lane < NL) ar.X[lane] = st.v(lane);
  else if (lane < NV) ar.X[lane] = bv[lane - NL];
  __syncthreads();
  body_spatial<T>(st, ar.X, bR, bp, ar.V, lane);
  __syncthreads();
  // ---- rows of the lane's contact: regulariser, reference accelerations -> record
  if (lane < ncon) {
    double* r = ar.rec[lane];
    const double pos[3] = {r[0], r[1], r[2]}, n[3] = {r[3], r[4], r[5]};
    const double dist = r[6], iw = r[8];
    const int A_ = ar.cb[lane] & 0xff, B_ = (ar.cb[lane] >> 8) & 0xff;
    double fk[3][3];
    fk[0][0] = n[0]; fk[0][1] = n[1]; fk[0][2] = n[2];
    make_frame(n, fk[1], fk[2]);
    const double imp = impedance(b.imp, dist, 0.0);
    double R0 = (1 - imp) / imp * iw;
    if (R0 < kMinVal) R0 = kMinVal;
    double rel[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) rel[k] = ar.V[B_][k] - ar.V[A_][k];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      double xf[3];
      cross3(pos, fk[k], xf);
      const double vel = xf[0] * rel[0] + xf[1] * rel[1] + xf[2] * rel[2] + fk[k][0] * rel[3] + fk[k][1] * rel[4] + fk[k][2] * rel[5];
      r[8 + k] = -b.B * vel - (k == 0 ? b.K * imp * dist : 0.0);
      r[11 + k] = 0.0;
    }
    r[6] = R0;
  }
  // links in contact (their composite contact stiffness enters the Hessian; the noslip pass keeps M^-1 S' for them)
  // in the order of their first contact, at most kMaxActive: a ballot per link says where it first appears (a contact names
  // at most one link: the other body is the cube or the world), then the earliest ones are picked -- scalar work for the
  // wavefront instead of one lane walking the contact list
  {
    const int myA = lane < ncon ? ar.cb[lane] & 0xff : 0xff, myB = lane < ncon ? (ar.cb[lane] >> 8) & 0xff : 0xff;
    uint64_t first_of[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) first_of[l] = __ballot(myA == l || myB == l);
    uint32_t listed = 0;  // links listed already (a self contact names two links)
    int na = 0;
    for (int k = 0; k < kMaxActive; ++k) {
      int best_l = -1, best_c = 64;
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        const int c0 = first_of[l] && !((listed >> l) & 1u) ? __ffsll((long long)first_of[l]) - 1 : 64;
        if (c0 < best_c) { best_c = c0; best_l = l; }
      }
      if (best_l < 0) break;
      listed |= 1u << best_l;
      if (lane == 0) ar.act[na] = best_l;
      ++na;
    }
    // pairs of links in contact with each other (self contact): each gets a stiffness accumulator of its own
    const int mycode = (myA < NL && myB < NL) ? ((myA < myB ? myA : myB) | ((myA < myB ? myB : myA) << 8)) : -1;
    int np = 0;
    bool pairs_over = false;
    for (uint64_t rem = __ballot(mycode >= 0); rem;) {
      const int code = __builtin_amdgcn_readlane(mycode, __ffsll((long long)rem) - 1);
      if (np < kMaxPairs) { if (lane == 0) ar.pairs[np] = code; ++np; }
      else pairs_over = true;
      rem &= ~__ballot(mycode == code);
    }
    if (lane == 0) {
      ar.nact = na;
      ar.npairs = np;
      // more links in contact than the stiffness fold / the noslip slots hold: the rest keep their contact forces but drop out
      // of the Hessian and of the noslip update -- a different (slower, for noslip: incomplete) iteration than MuJoCo's: flagged
      bool left = pairs_over;
#pragma unroll
      for (int l = 0; l < NL; ++l) left = left || (first_of[l] && !((listed >> l) & 1u));
      if (left) ar.pad[0] |= 2;
    }
  }
  // ---- qacc_smooth: the robot's by its own factorisation (every lane), the box's in closed form
  const double Mb[6] = {b.mass, b.mass, b.mass, b.inertia[0], b.inertia[1], b.inertia[2]};
  {
    double LM[T::NTRI], a0[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) LM[tri(i, j)] = st.M(i, j);
    ldl_factor<NL>(LM);
    // (the factor stays for the noslip pass where the bodies' velocities were: the rows above are done with them)
    static_assert(sizeof(ar.V) >= sizeof(double) * T::NTRI, "the factor of M fits the velocities' area");
    __syncthreads();
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < T::NTRI; ++k) (&ar.V[0][0])[k] = LM[k];
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) a0[i] = st.smooth(i);
    ldl_solve<NL>(LM, a0);
    double xsb[6];
    {
      const double* w = bv + 3;
      const double Iw[3] = {b.inertia[0] * w[0], b.inertia[1] * w[1], b.inertia[2] * w[2]};
      double gyro[3];
      cross3(w, Iw, gyro);
#pragma unroll
      for (int j = 0; j < 3; ++j) { xsb[j] = gravity[j]; xsb[3 + j] = -gyro[j] * b.inv_inertia[j]; }
    }
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < NL; ++i) ar.A0[i] = a0[i];
#pragma unroll
      for (int k = 0; k < 6; ++k) ar.A0[NL + k] = xsb[k];
    }
  }
  __syncthreads();
  ConLane c;
  con_load(ar, b, lane, ncon, kWorld, c);
  const int nact = ar.nact;
  const bool has_eq = T::GRIP && st.eq(0) != 0.0;
  const double eqD = T::GRIP ? st.eq(0) : 0.0, eqAref = T::GRIP ? st.eq(1) : 0.0, eqJ1 = T::GRIP ? st.eq(2) : 0.0;

  // cost of the lane's contact at the bodies' accelerations Ub: force, cone Hessian, jar
  double jar[3] = {0, 0, 0}, Hc[6] = {0, 0, 0, 0, 0, 0}, f[3] = {0, 0, 0};
  auto eval_rows = [&](const double (*Ub)[6], double* jar_out, double* f_out, double* Hc_out) -> double {
    double cost = 0;
    if (c.on) {
      double rel[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) rel[k] = Ub[c.B][k] - Ub[c.A][k];
#pragma unroll
      for (int k = 0; k < 3; ++k) jar_out[k] = dot6(c.G[k], rel) - c.aref[k];
      cost = cone_eval(c.D, c.mu, c.fr, jar_out, f_out, Hc_out);
    }
    return cost;
  };
  // robot rows (limit rows, the finger coupling) and the Gauss term at x (LDS vector): lane t < NV computes its dof's
  // gradient entry without the contacts' part; returns this lane's share of the cost
  auto robot_terms = [&](const double* x, double* grad_out) -> double {
    double cost = 0, g = 0;
    if (lane < NL) {
      double mx = 0;
      for (int j = 0; j < NL; ++j) {
        const double mij = lane >= j ? st.M(lane, j) : st.M(j, lane);
        mx += mij * (x[j] - ar.A0[j]);
      }
      g = mx;
      cost = 0.5 * (x[lane] - ar.A0[lane]) * mx;
      const double sgn = st.limS(lane);
      if (sgn != 0.0) {
        const double r = sgn * x[lane] - st.limA(lane);
        if (r < 0) { const double dd = st.limD(lane); cost += 0.5 * dd * r * r; g += sgn * dd * r; }
      }
      if (T::GRIP && has_eq && (lane == NA || lane == NA + 1)) {
        const double je = x[NA] + eqJ1 * x[NA + 1] - eqAref;
        if (lane == NA) { cost += 0.5 * eqD * je * je; g += eqD * je; }
        else g += eqD * je * eqJ1;
      }
      if constexpr (FRIC) {
        const double fF = links[lane].fl_floss;
        if (fF > 0) {
          const double fD = links[lane].fl_D, fR = links[lane].fl_R, jf = x[lane] - st.fa(lane);
          if (jf <= -fR) { cost += -0.5 * fR * fF - fF * jf; g -= fF; }
          else if (jf >= fR) { cost += -0.5 * fR * fF + fF * jf; g += fF; }
          else { cost += 0.5 * fD * jf * jf; g += fD * jf; }
        }
      }
    } else if (lane < NV) {
      const int k = lane - NL;
      const double dx = x[lane] - ar.A0[lane];
      g = Mb[k] * dx;
      cost = 0.5 * Mb[k] * dx * dx;
    }
    *grad_out = g;
    return cost;
  };

  TEAM_MARK(27)
  // start: the cheaper of the warm start and qacc_smooth
  {
    if (lane < NL) ar.P[lane] = st.xs(lane);
    else if (lane < NV) ar.P[lane] = bs[kBoxW + lane - NL];
    if (lane < NV) ar.X[lane] = ar.A0[lane];
    __syncthreads();
    body_spatial<T>(st, ar.X, bR, bp, ar.U, lane);
    body_spatial<T>(st, ar.P, bR, bp, ar.Up, lane);
    __syncthreads();
    double g, ja[3], fa[3], Ha[6];
    const double c_smooth = wave_sum(eval_rows(ar.U, ja, fa, Ha) + robot_terms(ar.X, &g));
    const double c_warm = wave_sum(eval_rows(ar.Up, ja, fa, Ha) + robot_terms(ar.P, &g));
    // a third candidate: where the previous coupled solve of this environment ended.  MuJoCo's warm start is the previous
    // step's final qacc, i.e. AFTER the noslip pass, which moves the friction forces away from the soft problem's minimiser
    // by about the same amount every step; a steadily held cube starts two to three iterations closer from here.  The cost
    // is strictly convex and the iteration runs to its minimiser: the starting point decides the path, not the result.
    static_assert(NV <= kBoxState - kBoxX, "the kept minimiser fits its slot of the box state");
    if (lane < NV) ar.Gd[lane] = bs[kBoxX + lane];
    __syncthreads();
    body_spatial<T>(st, ar.Gd, bR, bp, ar.W, lane);
    __syncthreads();
    const double c_prev = wave_sum(eval_rows(ar.W, ja, fa, Ha) + robot_terms(ar.Gd, &g));
    if (lane < NV) {
      if (c_warm < c_smooth) ar.X[lane] = ar.P[lane];
      if (c_prev < fmin(c_warm, c_smooth)) ar.X[lane] = ar.Gd[lane];
    }
    __syncthreads();
  }
  // what does not change over the iterations: the contacts of every body (the generalised force) and of every stiffness
  // accumulator a < nact: (link act[a], box); kMaxActive + a: (world, link act[a]); last: (world, box)
  const BodyMasks bmasks = body_masks<NL + 1>(c, lane);
  // ... and kAcc - kMaxPairs + p: the two links of ar.pairs[p] against each other
  constexpr int kAcc = AR::kAcc;
  static_assert(kAcc <= 18, "two accumulators per group of seven lanes");
  const int npairs = ar.npairs;
  uint64_t kmask = 0, kmask2 = 0;  // contacts of accumulator lane / 7, and of accumulator 9 + lane / 7
  {
    int slot = -1;
    if (c.on) {
      if (c.A < NL && c.B < NL) {
        const int code = (c.A < c.B ? c.A : c.B) | ((c.A < c.B ? c.B : c.A) << 8);
        for (int q = 0; q < npairs; ++q) if (ar.pairs[q] == code) slot = 2 * kMaxActive + 1 + q;
      } else {
        const int lk = c.A < NL ? c.A : (c.B < NL ? c.B : -1);
        const bool with_box = c.A == kBox || c.B == kBox;
        if (lk < 0) slot = 2 * kMaxActive;
        else {
          int a = -1;
          for (int k = 0; k < nact; ++k) if (ar.act[k] == lk) a = k;
          slot = a < 0 ? -1 : (with_box ? a : kMaxActive + a);
        }
      }
    }
#pragma unroll
    for (int a = 0; a < kAcc; ++a) {
      const uint64_t m = __ballot(slot == a);
      if (lane / 7 == a) kmask = m;
      if (lane / 7 + 9 == a) kmask2 = m;
    }
  }
  bool at_x = false;  // jar / f / Hc are those of ar.X
  int newton_done = 100;  // iterations the loop took (100: ran into its cap)
  for (int newton_it = 0; newton_it < 100; ++newton_it) {
    TEAM_MARK(55)
    TEAM_COUNT(29)
    // (the bodies' accelerations at x: by the tree pass at the start, along the search direction afterwards)
    if (newton_it == 0) body_spatial<T>(st, ar.X, bR, bp, ar.U, lane);
    __syncthreads();
    eval_rows(ar.U, jar, f, Hc);
    at_x = true;
    double gl;
    robot_terms(ar.X, &gl);
    const double qf = contact_qfrc<T>(ar, st, c, f, bmasks, bR, bp, lane);
    if (lane < NV) { gl -= qf; ar.Gd[lane] = gl; }
    const double g2 = wave_sum(lane < NV ? gl * gl : 0.0);
    // Converged: the gradient below the absolute bar, or at its round-off floor relative to the contacts' generalised force (the
    // gradient is the difference of two terms of that size; a contact force of 1e3 N puts the floor above the absolute bar, and the
    // loop spun at a fixed point until its cap in 2 % of the headline workload's solves: oracle orc_solve_coupled, ORC_NEWTON_REL)
    const double q2 = wave_sum(lane < NV ? qf * qf : 0.0);
    if (b.scale * sqrt(g2) < 1e-12 || g2 <= kNewtonRel * kNewtonRel * q2) { newton_done = newton_it; break; }
    TEAM_MARK(48)
    // ---- contact stiffness K_c = G' Hc G (6 x 6 symmetric, 21 entries), summed per body pair through LDS in three
    // batches of seven entries: accumulator a < nact: (link act[a], box); kMaxActive + a: (world, link act[a]); last: (world, box)
    {
      double Kc[21];
      if (c.on) {
        double Tm[3][6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          Tm[0][k] = Hc[0] * c.G[0][k] + Hc[1] * c.G[1][k] + Hc[3] * c.G[2][k];
          Tm[1][k] = Hc[1] * c.G[0][k] + Hc[2] * c.G[1][k] + Hc[4] * c.G[2][k];
          Tm[2][k] = Hc[3] * c.G[0][k] + Hc[4] * c.G[1][k] + Hc[5] * c.G[2][k];
        }
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
          for (int k = 0; k <= r; ++k) Kc[tri(r, k)] = c.G[0][r] * Tm[0][k] + c.G[1][r] * Tm[1][k] + c.G[2][r] * Tm[2][k];
      } else {
#pragma unroll
        for (int e = 0; e < 21; ++e) Kc[e] = 0.0;
      }
#pragma unroll
      for (int batch = 0; batch < 3; ++batch) {
        __syncthreads();
        if (lane < AR::kCap) {
          double* wr = ar.stage[lane];
#pragma unroll
          for (int e = 0; e < 7; ++e) wr[e] = Kc[7 * batch + e];
        }
        __syncthreads();
        {
          const int a = lane / 7, e = lane % 7;
          if (a < 9) {
            double s = 0;
            for (uint64_t m = kmask; m; m &= m - 1) s += ar.stage[__ffsll((long long)m) - 1][e];
            ar.KA[a][7 * batch + e] = s;
            if (a + 9 < kAcc) {
              double s2 = 0;
              for (uint64_t m = kmask2; m; m &= m - 1) s2 += ar.stage[__ffsll((long long)m) - 1][e];
              ar.KA[a + 9][7 * batch + e] = s2;
            }
          }
        }
      }
      __syncthreads();
    }
    TEAM_MARK(49)
    // ---- Hessian H = M + rows' curvature + S' K S (lower triangle, LDS): lane t < NV writes row t
    if (lane < NL) {
      // composite stiffness below link `lane`: KD over (link, box) and (world, link) pairs, KX over (link, box) pairs
      double KD[21], KX[21];
#pragma unroll
      for (int e = 0; e < 21; ++e) { KD[e] = 0.0; KX[e] = 0.0; }
      for (int a = 0; a < nact; ++a) {
        if (!is_anc<T>(lane, ar.act[a])) continue;
#pragma unroll
        for (int e = 0; e < 21; ++e) { KD[e] += ar.KA[a][e] + ar.KA[kMaxActive + a][e]; KX[e] += ar.KA[a][e]; }
      }
      double Sl[6], y[6], z[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) Sl[k] = st.S(lane, k);
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        double sy = 0, sz = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const int e = r >= k ? tri(r, k) : tri(k, r);
          sy += KD[e] * Sl[k];
          sz += KX[e] * Sl[k];
        }
        y[r] = sy; z[r] = sz;
      }
      // self contact: a pair of links (la, lb) with stiffness K_p contributes (S_lb - S_la)' K_p (S_lb - S_la): the entry of joints
      // i, j is s_i s_j S_i' K_p S_j with s_j = [j moves lb] - [j moves la] -- the joints BETWEEN the two links
      double yp[kMaxPairs][6];
      int sl[kMaxPairs];
#pragma unroll
      for (int q = 0; q < kMaxPairs; ++q) {
        sl[q] = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r) yp[q][r] = 0.0;
        if (q < npairs) {
          const int la = ar.pairs[q] & 0xff, lb = ar.pairs[q] >> 8;
          sl[q] = (is_anc<T>(lane, lb) ? 1 : 0) - (is_anc<T>(lane, la) ? 1 : 0);
          if (sl[q] != 0) {
            const double* Kp = ar.KA[2 * kMaxActive + 1 + q];
#pragma unroll
            for (int r = 0; r < 6; ++r) {
              double sy = 0;
#pragma unroll
              for (int k = 0; k < 6; ++k) sy += Kp[r >= k ? tri(r, k) : tri(k, r)] * Sl[k];
              yp[q][r] = sl[q] * sy;
            }
          }
        }
      }
      for (int j = 0; j <= lane; ++j) {
        double v = st.M(lane, j);
        if (is_anc<T>(j, lane)) {
#pragma unroll
          for (int k = 0; k < 6; ++k) v += st.S(j, k) * y[k];
        }
#pragma unroll
        for (int q = 0; q < kMaxPairs; ++q) {
          if (sl[q] == 0) continue;
          const int la = ar.pairs[q] & 0xff, lb = ar.pairs[q] >> 8;
          const int sj = (is_anc<T>(j, lb) ? 1 : 0) - (is_anc<T>(j, la) ? 1 : 0);
          if (sj == 0) continue;
          double dv = 0;
#pragma unroll
          for (int k = 0; k < 6; ++k) dv += st.S(j, k) * yp[q][k];
          v += sj * dv;
        }
        if (j == lane) {
          const double sgn = st.limS(lane);
          if (sgn != 0.0 && sgn * ar.X[lane] - st.limA(lane) < 0) v += st.limD(lane);
          if constexpr (FRIC) {
            const double fF = links[lane].fl_floss, fR = links[lane].fl_R, jf = ar.X[lane] - st.fa(lane);
            if (fF > 0 && jf > -fR && jf < fR) v += links[lane].fl_D;
          }
        }
        if (T::GRIP && has_eq) {
          if (lane == NA && j == NA) v += eqD;
          if (lane == NA + 1 && j == NA) v += eqD * eqJ1;
          if (lane == NA + 1 && j == NA + 1) v += eqD * eqJ1 * eqJ1;
        }
        ar.H[tri(lane, j)] = v;
      }
      // box rows' robot columns: H[NL + k][lane] = -S_box[:, k] . z
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        double col[6];
        box_column(bR, bp, k, col);
        ar.H[tri(NL + k, lane)] = -dot6(col, z);
      }
    } else if (lane < NV) {
      const int k = lane - NL;
      double Kb[21];
#pragma unroll
      for (int e = 0; e < 21; ++e) Kb[e] = ar.KA[2 * kMaxActive][e];
      for (int a = 0; a < nact; ++a)
#pragma unroll
        for (int e = 0; e < 21; ++e) Kb[e] += ar.KA[a][e];
      double ck[6], y[6];
      box_column(bR, bp, k, ck);
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        double s = 0;
#pragma unroll
        for (int q = 0; q < 6; ++q) s += Kb[r >= q ? tri(r, q) : tri(q, r)] * ck[q];
        y[r] = s;
      }
      for (int l = 0; l <= k; ++l) {
        double cl[6];
        box_column(bR, bp, l, cl);
        ar.H[tri(NL + k, NL + l)] = dot6(cl, y) + (l == k ? Mb[k] : 0.0);
      }
    }
    __syncthreads();
    TEAM_MARK(50)
    // ---- Newton direction p = -H^-1 grad.  Lane i < NV takes row i of the (symmetric) Hessian into registers; LDL'
    // right-looking: step j divides column j by the pivot (every lane its own entry) and subtracts the column's outer
    // product from the rows below, the other rows' entries of column j read across the lanes into scalar registers.  A lane
    // stops at its own step, so its entries right of the diagonal stay what column i was then (L_ki d_i): the backward
    // substitution's coefficients.  Both substitutions are column sweeps with the finished entry read across the lanes.
    double dphi0 = 0;
    {
      const int row = lane < NV ? lane : NV - 1;
      double hr[NV];
#pragma unroll
      for (int k = 0; k < NV; ++k) hr[k] = ar.H[row >= k ? tri(row, k) : tri(k, row)];
      TEAM_MARK(51)
#pragma unroll
      for (int j = 0; j < NV - 1; ++j) {
        const double dj = wave_read(hr[j], j);
        const double lij = lane > j ? hr[j] / dj : 0.0;
#pragma unroll
        for (int k = j + 1; k < NV; ++k) hr[k] -= lij * wave_read(hr[j], k);
      }
      double dg = 0;
#pragma unroll
      for (int k = 0; k < NV; ++k) dg = row == k ? hr[k] : dg;
      const double dinv = 1.0 / dg, gl_ = lane < NV ? ar.Gd[lane] : 0.0;
      double acc = -gl_;
#pragma unroll
      for (int k = 0; k < NV - 1; ++k) {
        const double yk = wave_read(acc, k) * wave_read(dinv, k);
        if (lane > k) acc -= hr[k] * yk;
      }
      acc *= dinv;
#pragma unroll
      for (int k = NV - 1; k >= 1; --k) {
        const double xk = wave_read(acc, k);
        if (lane < k) acc -= hr[k] * dinv * xk;
      }
      dphi0 = wave_sum(lane < NV ? gl_ * acc : 0.0);
      if (lane < NV) ar.P[lane] = acc;
    }
    TEAM_MARK(52)
    if (!(dphi0 < 0)) { newton_done = 1000 + newton_it; break; }
    __syncthreads();
    body_spatial<T>(st, ar.P, bR, bp, ar.Up, lane);
    __syncthreads();
    // ---- line search: root of phi'(a) by safeguarded 1-D Newton (a = 1 is exact while no row changes zone)
    double jd[3] = {0, 0, 0};
    if (c.on) {
      double rel[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) rel[k] = ar.Up[c.B][k] - ar.Up[c.A][k];
#pragma unroll
      for (int k = 0; k < 3; ++k) jd[k] = dot6(c.G[k], rel);
    }
    // Gauss term along the line: phi_M'(a) = gM0 + a pMp
    double gM0l = 0, pMpl = 0;
    if (lane < NL) {
      double mx = 0, mp = 0;
      for (int j = 0; j < NL; ++j) {
        const double mij = lane >= j ? st.M(lane, j) : st.M(j, lane);
        mx += mij * (ar.X[j] - ar.A0[j]);
        mp += mij * ar.P[j];
      }
      gM0l = mx * ar.P[lane]; pMpl = mp * ar.P[lane];
    } else if (lane < NV) {
      const int k = lane - NL;
      gM0l = Mb[k] * (ar.X[lane] - ar.A0[lane]) * ar.P[lane];
      pMpl = Mb[k] * ar.P[lane] * ar.P[lane];
    }
    const double gM0 = wave_sum(gM0l), pMp = wave_sum(pMpl);
    TEAM_MARK(53)
    double lo = 0, hi = -1, a = 1, best = 1, dx = 1e300, dxold = 1e300;
    for (int ls = 0; ls < 30; ++ls) {
      TEAM_COUNT(35)
      double dl = 0, ddl = 0;
      if (c.on) {
        double ja[3], fa[3], Ha[6];
#pragma unroll
        for (int k = 0; k < 3; ++k) ja[k] = jar[k] + a * jd[k];
        cone_eval(c.D, c.mu, c.fr, ja, fa, Ha);
        dl = -(jd[0] * fa[0] + jd[1] * fa[1] + jd[2] * fa[2]);
        ddl = jd[0] * (Ha[0] * jd[0] + Ha[1] * jd[1] + Ha[3] * jd[2]) + jd[1] * (Ha[1] * jd[0] + Ha[2] * jd[1] + Ha[4] * jd[2]) +
              jd[2] * (Ha[3] * jd[0] + Ha[4] * jd[1] + Ha[5] * jd[2]);
      }
      if (lane < NL) {
        const double sgn = st.limS(lane);
        if (sgn != 0.0) {
          const double r = sgn * (ar.X[lane] + a * ar.P[lane]) - st.limA(lane);
          if (r < 0) { const double dd = st.limD(lane), jl = sgn * ar.P[lane]; dl += dd * r * jl; ddl += dd * jl * jl; }
        }
        if (T::GRIP && has_eq && lane == NA) {
          const double je = (ar.X[NA] + a * ar.P[NA]) + eqJ1 * (ar.X[NA + 1] + a * ar.P[NA + 1]) - eqAref;
          const double jde = ar.P[NA] + eqJ1 * ar.P[NA + 1];
          dl += eqD * je * jde; ddl += eqD * jde * jde;
        }
        if constexpr (FRIC) {
          const double fF = links[lane].fl_floss;
          if (fF > 0) {
            const double fD = links[lane].fl_D, fR = links[lane].fl_R, pl = ar.P[lane], jf = ar.X[lane] + a * pl - st.fa(lane);
            if (jf <= -fR) dl -= fF * pl;
            else if (jf >= fR) dl += fF * pl;
            else { dl += fD * jf * pl; ddl += fD * pl * pl; }
          }
        }
      }
      const double dphi = wave_sum(dl) + gM0 + a * pMp;
      const double ddphi = wave_sum(ddl) + pMp;
      best = a;
      if (fabs(dphi) <= 1e-3 * fabs(dphi0)) break;
      if (dphi < 0) lo = a; else hi = a;
      double an = a - dphi / ddphi;
      // Newton on phi' with the bracket as the safeguard (the rtsafe rule): bisect when the step leaves the bracket or does
      // not at least halve the step before last -- phi' is piecewise smooth, between two pieces Newton alone can cycle
      if (hi > 0 && (!(an > lo && an < hi) || fabs(2 * dphi) > fabs(dxold * ddphi))) an = 0.5 * (lo + hi);
      if (hi < 0 && !(an > lo)) an = 2 * a;
      if (fabs(an - a) <= 1e-3 * a) break;
      dxold = dx;
      dx = an - a;
      a = an;
    }
    TEAM_MARK(54)
    __syncthreads();
    bool moved = false;
    if (lane < NV) {
      const double xo = ar.X[lane], xn = xo + best * ar.P[lane];
      moved = xn != xo;
      ar.X[lane] = xn;
    }
    if (lane < NL + 1) {
#pragma unroll
      for (int k = 0; k < 6; ++k) ar.U[lane][k] += best * ar.Up[lane][k];
    }
    at_x = false;
    __syncthreads();
    // (a step that no longer moves the iterate: a fixed point of the iteration in floating point)
    if (!__ballot(moved)) { newton_done = newton_it + 1; break; }
  }
#ifdef RCSH_PHASE_TIMING
  if (lane == 0) {  // (every workgroup) worst iteration count, solves over 20 iterations, capped solves, solves that left on a non-descent direction
    atomicMax(&g_team_cycles[56], (unsigned long long)(newton_done % 1000));
    if (newton_done % 1000 > 20) atomicAdd(&g_team_cycles[57], 1ull);
    if (newton_done == 100) atomicAdd(&g_team_cycles[58], 1ull);
    if (newton_done >= 1000) atomicAdd(&g_team_cycles[59], 1ull);
    atomicAdd(&g_team_cycles[60], 1ull);
    atomicAdd(&g_team_cycles[67], (unsigned long long)(newton_done % 1000));
  }
#endif
  (void)newton_done;
  // forces at the solution -> records (the loop's last evaluation unless it ran into its cap)
  if (!at_x) eval_rows(ar.U, jar, f, Hc);
  if (lane < NV) bs[kBoxX + lane] = ar.X[lane];
  if (c.on) {
    double* r = ar.rec[lane];
    r[11] = f[0]; r[12] = f[1]; r[13] = f[2];
  }
  __syncthreads();
  TEAM_MARK(28)
}

// ================================================================= phase 3: noslip + results
// mj_solNoSlip over the contacts' friction rows (Gauss-Seidel in contact order, no regulariser), then the robot's
// qfrc_constraint -> st.fcon and the box's acceleration -> bs[kBoxA..].  In: the bodies' accelerations ar.U at the Newton
// solution ar.X, the records.
template <class T, class AR>
RCSH_CONTACT_FN void contact_noslip(const BoxCfg& b_, const StageTeam<T>& st_, double* bs_, AR& ar_) {
  const BoxCfg& b = *in_lds(&b_);
  const StageTeam<T> st{in_lds(st_.base)};
  double* bs = in_lds(bs_);
  AR& ar = *in_lds(&ar_);
  constexpr int NL = T::NL, NA = T::NARM, NV = NL + 6, NB = NL + 2;
  constexpr int kBox = NL, kWorld = NL + 1;
  const int lane = wave_lane();
  const int ncon = ar.ncon, nact = ar.nact;
  double bp[3], bR[9], bv[6];
  box_frame(bs, bp, bR, bv);
  const double Mbi[6] = {b.inv_mass, b.inv_mass, b.inv_mass, b.inv_inertia[0], b.inv_inertia[1], b.inv_inertia[2]};
  ConLane c;
  con_load(ar, b, lane, ncon, kWorld, c);
  const bool has_eq = T::GRIP && st.eq(0) != 0.0;
  const double eqD = T::GRIP ? st.eq(0) : 0.0, eqAref = T::GRIP ? st.eq(1) : 0.0, eqJ1 = T::GRIP ? st.eq(2) : 0.0;
  if (b.noslip_iterations > 0) {
    // Y_a = M^-1 S_a' for the links in contact (robot dofs x 6), [a][dof][k] in the stage area: lane (a, k) solves one column
    double (*Y)[NL][6] = reinterpret_cast<double (*)[NL][6]>(&ar.stage[0][0]);
    static_assert(kMaxActive * NL * 6 <= 64 * 8, "Y fits the stage area");
    {
      if (lane < 6 * nact) {
        const int a = lane / 6, k = lane % 6, lk = ar.act[a];
        double LM[T::NTRI], col[NL];  // (M's factor: left in ar.V by contact_newton)
#pragma unroll
        for (int e = 0; e < T::NTRI; ++e) LM[e] = (&ar.V[0][0])[e];
#pragma unroll
        for (int j = 0; j < NL; ++j) col[j] = is_anc<T>(j, lk) ? st.S(j, k) : 0.0;
        ldl_solve<NL>(LM, col);
#pragma unroll
        for (int j = 0; j < NL; ++j) Y[a][j][k] = col[j];
      }
    }
    __syncthreads();
    // The sweeps move wrenches on few bodies -- the links in contact (slots 0..nact-1) and the cube (slot kMaxActive) -- and
    // read only those bodies' accelerations.  Lane 6 s + k keeps component k of the CHANGE of slot s's acceleration in a
    // register; K[s][t] = S_s M^-1 S_t' (6 x 6, from Y) says what a wrench on slot t does to slot s, the cube answers with
    // its own inverse inertia only.  An update is then: the owner reads the two bodies' changes across the lanes, solves
    // its friction rows, hands the wrench change back across the lanes, and the slot lanes add their row of K times it:
    // no barrier and no pass over the kinematic tree per contact.
    // K[s][t] = K[t][s]': the blocks t <= s are kept (block s (s + 1) / 2 + t), the others read transposed
    constexpr int kKBlocks = kMaxActive * (kMaxActive + 1) / 2;
    double (*Kh)[6][6] = reinterpret_cast<double (*)[6][6]>(&ar.rec[0][0]);  // (the records are in the lanes by now)
    double (*Kbox)[6] = reinterpret_cast<double (*)[6]>(&ar.rec[0][0] + kKBlocks * 36);  // the cube's own block
    static_assert(sizeof(ar.rec) >= sizeof(double) * (kKBlocks + 1) * 36, "K fits the records' area");
    auto kval = [&](int s_, int t_, int k_, int m_) -> double { return s_ >= t_ ? Kh[s_ * (s_ + 1) / 2 + t_][k_][m_] : Kh[t_ * (t_ + 1) / 2 + s_][m_][k_]; };
    double kb[6] = {0, 0, 0, 0, 0, 0};  // slot lanes of the cube: their row of S_box M_box^-1 S_box'
    if (lane < 6 * nact) {
      const int sl = lane / 6, k = lane % 6, lk = ar.act[sl];
      for (int t = 0; t <= sl; ++t) {
        double row[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < NL; ++j) {
          if (!is_anc<T>(j, lk)) continue;
          const double sjk = st.S(j, k);
#pragma unroll
          for (int m = 0; m < 6; ++m) row[m] += sjk * Y[t][j][m];
        }
#pragma unroll
        for (int m = 0; m < 6; ++m) Kh[sl * (sl + 1) / 2 + t][k][m] = row[m];
      }
    } else if (lane >= 6 * kMaxActive && lane < 6 * kMaxActive + 6) {
      const int k = lane - 6 * kMaxActive;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        double col[6];
        box_column(bR, bp, i, col);
        const double q = Mbi[i] * col[k];
#pragma unroll
        for (int m = 0; m < 6; ++m) kb[m] += q * col[m];
      }
    }
    auto slot_of = [&](int body) {
      if (body == kBox) return kMaxActive;
      int sl = -1;
      for (int a = 0; a < nact; ++a) if (ar.act[a] == body) sl = a;
      return sl;
    };
    const int my_slots = (slot_of(c.A) & 0xff) | (slot_of(c.B) & 0xff) << 8;  // (0xff: the world, or a link beyond the kept ones)
    if (lane >= 6 * kMaxActive && lane < 6 * kMaxActive + 6) {
#pragma unroll
      for (int m = 0; m < 6; ++m) Kbox[lane - 6 * kMaxActive][m] = kb[m];
    }
    __syncthreads();
    // the 3 x 3 block of A = J M^-1 J' (no regulariser) of the lane's own contact: a wrench w on body B and -w on body A moves
    // the pair's relative acceleration by (K_BB + K_AA) w (a link and the cube, or the world, do not see each other in M^-1)
    double Ac[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int k = 0; k < 3; ++k) Ac[r][k] = 0.0;
    if (c.on) {
      for (int side = 0; side < 2; ++side) {
        const int sl = (my_slots >> (8 * side)) & 0xff;
        if (sl == 0xff) continue;
        const double (*Km)[6] = sl == kMaxActive ? Kbox : Kh[sl * (sl + 1) / 2 + sl];
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
          double rel[6];
#pragma unroll
          for (int k = 0; k < 6; ++k) rel[k] = dot6(Km[k], c.G[kk]);
#pragma unroll
          for (int r = 0; r < 3; ++r) Ac[r][kk] += dot6(c.G[r], rel);
        }
      }
      // two links of the robot (self contact) DO see each other in M^-1: the cross blocks, G (-K_AB - K_BA) G'
      const int sA_ = my_slots & 0xff, sB_ = (my_slots >> 8) & 0xff;
      if (sA_ < kMaxActive && sB_ < kMaxActive) {
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
          double rel[6];
#pragma unroll
          for (int k = 0; k < 6; ++k) {
            double v = 0;
#pragma unroll
            for (int m = 0; m < 6; ++m) v += (kval(sB_, sA_, k, m) + kval(sA_, sB_, k, m)) * c.G[kk][m];
            rel[k] = v;
          }
#pragma unroll
          for (int r = 0; r < 3; ++r) Ac[r][kk] -= dot6(c.G[r], rel);
        }
      }
    }
    // the friction rows' unconstrained solve (the first pass of the QCQP: multiplier 0) needs the inverse of its scaled block only
    const double qS11 = Ac[1][1] * c.fr * c.fr, qS22 = Ac[2][2] * c.fr * c.fr, qS12 = Ac[1][2] * c.fr * c.fr;
    const double qdet = qS11 * qS22 - qS12 * qS12, qdi = 1 / qdet;
    double rel0[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) rel0[k] = ar.U[c.B][k] - ar.U[c.A][k];
    double du = 0;
    __syncthreads();
    TEAM_MARK(30)
    int iter = 0;
    while (iter < b.noslip_iterations) {
      TEAM_COUNT(36)
      double improvement = 0;
      if (iter == 0) {
        double s = c.on ? 0.5 * (c.f[0] * c.f[0] * c.Rr[0] + c.f[1] * c.f[1] * c.Rr[1] + c.f[2] * c.f[2] * c.Rr[2]) : 0.0;
        if (lane < NL) {
          const double sgn = st.limS(lane);
          if (sgn != 0.0) {
            const double r = sgn * ar.X[lane] - st.limA(lane);
            if (r < 0) s += 0.5 * st.limD(lane) * r * r;  // 0.5 f^2 R with f = -D r
          }
        }
        improvement = wave_sum(s);
      }
      for (int cc = 0; cc < ncon; ++cc) {
        TEAM_MARK(40)
        const int slots = wave_read(my_slots, cc), sA = slots & 0xff, sB = slots >> 8;
        double rel[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          double r = rel0[k];
          if (sB != 0xff) r += wave_read(du, 6 * sB + k);
          if (sA != 0xff) r -= wave_read(du, 6 * sA + k);
          rel[k] = r;
        }
        double change = 0, dw[6] = {0, 0, 0, 0, 0, 0};
        int moved = 0;
        if (lane == cc) {
          const double old[3] = {c.f[0], c.f[1], c.f[2]};
          double nf[3] = {old[0], old[1], old[2]};
          if (old[0] < kMinVal) {
            // (a contact the Newton solution left without normal force: all three rows go to zero -- the general update)
            double res[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) res[k] = dot6(c.G[k], rel) - c.aref[k];
            nf[0] = nf[1] = nf[2] = 0;
            const double dl[3] = {-old[0], -old[1], -old[2]};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
#pragma unroll
              for (int l = 0; l < 3; ++l) change += 0.5 * dl[k] * Ac[k][l] * dl[l];
              change += dl[k] * res[k];
            }
            if (change > 1e-10) { nf[0] = old[0]; nf[1] = old[1]; nf[2] = old[2]; change = 0; }
#pragma unroll
            for (int k = 0; k < 6; ++k) dw[k] = c.G[0][k] * (nf[0] - old[0]) + c.G[1][k] * (nf[1] - old[1]) + c.G[2][k] * (nf[2] - old[2]);
          } else {
            // the normal force stays: only the friction rows' residuals, their 2 x 2 block and their part of the wrench are needed
            const double res1 = dot6(c.G[1], rel) - c.aref[1], res2 = dot6(c.G[2], rel) - c.aref[2];
            const double b1 = res1 - Ac[1][1] * old[1] - Ac[1][2] * old[2], b2 = res2 - Ac[2][1] * old[1] - Ac[2][2] * old[2];
            double vv[2];
            if (qcqp2_dev(vv, Ac[1][1], Ac[1][2], Ac[2][2], b1, b2, c.fr, c.fr, old[0], qdet, qdi)) {
              double s = vv[0] * vv[0] / (c.fr * c.fr) + vv[1] * vv[1] / (c.fr * c.fr);
              s = sqrt(old[0] * old[0] / (s > kMinVal ? s : kMinVal));
              vv[0] *= s; vv[1] *= s;
            }
            nf[1] = vv[0]; nf[2] = vv[1];
            const double d1 = nf[1] - old[1], d2 = nf[2] - old[2];
            change = 0.5 * d1 * Ac[1][1] * d1 + 0.5 * d1 * Ac[1][2] * d2 + d1 * res1 + 0.5 * d2 * Ac[2][1] * d1 + 0.5 * d2 * Ac[2][2] * d2 + d2 * res2;
            if (change > 1e-10) { nf[1] = old[1]; nf[2] = old[2]; change = 0; }
#pragma unroll
            for (int k = 0; k < 6; ++k) dw[k] = c.G[1][k] * (nf[1] - old[1]) + c.G[2][k] * (nf[2] - old[2]);
          }
          moved = nf[0] != old[0] || nf[1] != old[1] || nf[2] != old[2];
          c.f[0] = nf[0]; c.f[1] = nf[1]; c.f[2] = nf[2];
        }
        TEAM_MARK(41)
        if (wave_read(moved, cc)) {
          // the wrench change dw on body B, -dw on body A: what it does to the kept bodies' accelerations
          double w[6];
#pragma unroll
          for (int k = 0; k < 6; ++k) w[k] = wave_read(dw[k], cc);
          if (lane < 6 * nact) {
            const int sl = lane / 6, k = lane % 6;
            if (sB < kMaxActive) {
              double acc = 0;
#pragma unroll
              for (int m = 0; m < 6; ++m) acc += kval(sl, sB, k, m) * w[m];
              du += acc;
            }
            if (sA < kMaxActive) {
              double acc = 0;
#pragma unroll
              for (int m = 0; m < 6; ++m) acc += kval(sl, sA, k, m) * w[m];
              du -= acc;
            }
          } else if (lane >= 6 * kMaxActive && lane < 6 * kMaxActive + 6) {
            const double sg = (sB == kMaxActive ? 1.0 : 0.0) - (sA == kMaxActive ? 1.0 : 0.0);
            du += sg * dot6(kb, w);
          }
        }
        TEAM_MARK(44)
        TEAM_COUNT(45)
        improvement -= wave_read(change, cc);
      }
      improvement *= b.scale;
      ++iter;
      if (improvement < b.noslip_tolerance) break;
    }
    __syncthreads();
  }
  TEAM_MARK(31)
  // ---- results: qfrc_constraint of the robot, qacc of the box
  {
    const double qf = contact_qfrc<T>(ar, st, c, c.f, body_masks<NB - 1>(c, lane), bR, bp, lane);
    if (lane < NL) {
      double fc = qf;
      const double sgn = st.limS(lane);
      if (sgn != 0.0) {
        const double r = sgn * ar.X[lane] - st.limA(lane);
        if (r < 0) fc += -sgn * st.limD(lane) * r;
      }
      if (T::GRIP && has_eq && (lane == NA || lane == NA + 1)) {
        const double fe = -eqD * (ar.X[NA] + eqJ1 * ar.X[NA + 1] - eqAref);
        fc += lane == NA ? fe : fe * eqJ1;
      }
      st.fcon(lane) = fc;
      st.xs(lane) = ar.X[lane];
    } else if (lane < NV) {
      const int k = lane - NL;
      bs[kBoxA + k] = ar.A0[lane] + Mbi[k] * qf;
    }
  }
  __syncthreads();
  TEAM_MARK(32)
}

#endif  // __HIP__

}  // namespace rcsh
#include "contact_wide.h"
#include "contact_dense.h"
