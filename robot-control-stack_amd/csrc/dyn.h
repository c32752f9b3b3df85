// dyn.h -- per-environment rigid-body dynamics of one robot archetype, written for
// one GPU thread per environment with every per-link quantity in registers.
//
// What it computes is what mj_step1 / mj_step2 compute for the RCS scenes
// (reference call sites src/sim/sim.cpp:110,112): kinematics, mass matrix, bias
// forces, gravity compensation, affine actuators with force clamps, the
// soft-constraint solve (finger coupling equality + joint limits) and the
// implicitfast integrator.  How it computes it is chosen for CDNA4:
//
//  * welded bodies are folded into links on the host (model.cpp), so the loops run
//    over NL = 9 links instead of 14 bodies;
//  * all spatial quantities are Pluecker vectors about the WORLD ORIGIN.  With one
//    common reference point the composite-inertia and force recursions are plain
//    sums -- no frame transforms on the backward pass;
//  * the archetype (chain length, gripper or not) is a template parameter, so every
//    loop unrolls and every array index is a compile-time constant: the arrays
//    live in VGPR/AGPR, model constants arrive through scalar loads;
//  * the constraint Hessian differs from M only on the diagonal (limit rows are
//    +-e_i) and in the 2x2 finger block (coupling row), so the Newton solve is a
//    packed 9x9 LDL^T with a handful of diagonal updates.
//
// The same templates are instantiated on the host by model.cpp for one purpose only:
// dof_invweight0 = diag(M(qpos0)^-1) at model-finalise time (MuJoCo computes the same
// constant when it compiles a model).  No stepping entry point runs on the CPU.
#pragma once
#include <cmath>
#include <cstdint>

#include "model.h"

#if defined(__HIP__)
#define RCSH_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define RCSH_HD inline
#endif

namespace rcsh {

template <int NARM_, bool GRIP_>
struct Topo {
  static constexpr int NARM = NARM_;
  static constexpr bool GRIP = GRIP_;
  static constexpr int NL = NARM_ + (GRIP_ ? 2 : 0);
  static constexpr int NU = NARM_ + (GRIP_ ? 1 : 0);
  static constexpr int NTRI = NL * (NL + 1) / 2;
  RCSH_HD static constexpr int parent(int i) { return i < NARM_ ? i - 1 : NARM_ - 1; }
};

RCSH_HD constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }  // i >= j

constexpr double kMinVal = 1e-15;
constexpr double kMinImp = 0.0001;
constexpr double kMaxImp = 0.9999;

RCSH_HD double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
// Compiler-only fence (no instruction): values parked in the LDS staging area must really be
// re-read after it, otherwise store-to-load forwarding keeps them alive in registers and the
// staging buys nothing.
RCSH_HD void stage_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" ::: "memory");
#endif
}
// Scheduling fence: everything written above it is issued before anything below it.  Used right after a batch of
// LDS reads so the reads go out back to back and ONE s_waitcnt covers them; without it the compiler sinks each
// read next to its first use and every read exposes the full LDS round trip (measured: 1.2 reads per wait).
RCSH_HD void sched_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_sched_barrier(0);
#endif
}
RCSH_HD double fast_rcp(double x);
RCSH_HD void fast_sincos(double x, double* sn, double* cs);

RCSH_HD void cross3(const double* a, const double* b, double* r) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
RCSH_HD double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
RCSH_HD double dot6(const double* a, const double* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}
// r = A(3x3 row-major) * v
RCSH_HD void mulmv(const double* A, const double* v, double* r) {
  double x = A[0] * v[0] + A[1] * v[1] + A[2] * v[2];
  double y = A[3] * v[0] + A[4] * v[1] + A[5] * v[2];
  double z = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
// C = A * B (3x3 row-major)
RCSH_HD void mulmm(const double* A, const double* B, double* C) {
  double t[9];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) t[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
#pragma unroll
  for (int k = 0; k < 9; ++k) C[k] = t[k];
}
// spatial inertia about the world origin (Ixx Iyy Izz Ixy Ixz Iyz, m*c, m) times motion vector [w; v]
RCSH_HD void inert_mul(const double* I, const double* v, double* r) {
  r[0] = I[0] * v[0] + I[3] * v[1] + I[4] * v[2] - I[8] * v[4] + I[7] * v[5];
  r[1] = I[3] * v[0] + I[1] * v[1] + I[5] * v[2] + I[8] * v[3] - I[6] * v[5];
  r[2] = I[4] * v[0] + I[5] * v[1] + I[2] * v[2] - I[7] * v[3] + I[6] * v[4];
  r[3] = I[8] * v[1] - I[7] * v[2] + I[9] * v[3];
  r[4] = I[6] * v[2] - I[8] * v[0] + I[9] * v[4];
  r[5] = I[7] * v[0] - I[6] * v[1] + I[9] * v[5];
}
RCSH_HD void cross_motion(const double* vel, const double* s, double* r) {
  double a[3], b[3];
  cross3(vel, s, r);
  cross3(vel, s + 3, a);
  cross3(vel + 3, s, b);
  r[3] = a[0] + b[0]; r[4] = a[1] + b[1]; r[5] = a[2] + b[2];
}
RCSH_HD void cross_force(const double* vel, const double* f, double* r) {
  double a[3], b[3];
  cross3(vel, f, a);
  cross3(vel + 3, f + 3, b);
  r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2];
  cross3(vel, f + 3, r + 3);
}

// World frame of link i from its parent link's frame (R, p are updated in place): the kinematic step of the
// forward sweep on its own, for code that needs frames only (contact detection, IK).
RCSH_HD void advance_link_frame(const DevModel& m, int i, double qi, double* R, double* p) {
  double o[3], R0[9];
  mulmv(R, m.pos0[i], o);
  o[0] += p[0]; o[1] += p[1]; o[2] += p[2];
  mulmm(R, m.rot0[i], R0);
  const double dq = qi - m.qpos0[i];
  if (m.jtype[i] == kSlide) {
    double ax[3];
    mulmv(R0, m.axis[i], ax);
    for (int k = 0; k < 9; ++k) R[k] = R0[k];
    p[0] = o[0] + ax[0] * dq; p[1] = o[1] + ax[1] * dq; p[2] = o[2] + ax[2] * dq;
    return;
  }
  double s, c;
  fast_sincos(dq, &s, &c);
  const double* a = m.axis[i];
  const double t = 1.0 - c;
  const double Q[9] = {c + t * a[0] * a[0],        t * a[0] * a[1] - s * a[2], t * a[0] * a[2] + s * a[1],
                       t * a[0] * a[1] + s * a[2], c + t * a[1] * a[1],        t * a[1] * a[2] - s * a[0],
                       t * a[0] * a[2] - s * a[1], t * a[1] * a[2] + s * a[0], c + t * a[2] * a[2]};
  double anchor[3], rj[3];
  mulmv(R0, m.jpos[i], anchor);
  anchor[0] += o[0]; anchor[1] += o[1]; anchor[2] += o[2];
  mulmm(R0, Q, R);
  mulmv(R, m.jpos[i], rj);
  p[0] = anchor[0] - rj[0]; p[1] = anchor[1] - rj[1]; p[2] = anchor[2] - rj[2];
}

// ---- staging area for what must survive from the forward to the backward sweep (per-link spatial
// inertia, bias wrench, gravity-compensation moment) plus the mass matrix, which two factorisations
// consume.  On the GPU it is one column of an LDS array laid out [slot][lane] (STRIDE = 64: a wave's
// accesses to one slot are 64 consecutive 8-byte words, conflict-free, and the slot index is an
// immediate offset); on the host (model finalisation) it is a plain array (STRIDE = 1).
template <class T, int STRIDE>
struct Stage {
  static constexpr int I0 = 0;
  static constexpr int F0 = I0 + 10 * T::NL;
  static constexpr int H0 = F0 + 6 * T::NL;
  static constexpr int M0 = H0 + 3 * T::NL;
  static constexpr int S0 = M0 + T::NTRI;
  static constexpr int Q0 = S0 + 6 * T::NL;   // qpos
  static constexpr int V0 = Q0 + T::NL;       // qvel
  static constexpr int C0 = V0 + T::NL;       // ctrl
  static constexpr int L0 = C0 + T::NU;       // limit rows: D, aref, sign per joint
  static constexpr int K0 = L0 + 3 * T::NL;   // frame of the site link at the last position stage: R(9) p(3)
  static constexpr int P0 = K0 + 12;          // qpos before the last integration (what the last mj_step1 saw)
  static constexpr int X0 = P0 + T::NL;       // caller's slots (sim_kernels.h parks rarely-touched state here)
  static constexpr int NX = 6 + 2 * T::NARM;
  static constexpr int COUNT = X0 + NX;
  double* base;
  RCSH_HD double& q(int i) const { return base[(Q0 + i) * STRIDE]; }
  RCSH_HD double& v(int i) const { return base[(V0 + i) * STRIDE]; }
  RCSH_HD double& c(int i) const { return base[(C0 + i) * STRIDE]; }
  RCSH_HD double& lim(int i, int k) const { return base[(L0 + 3 * i + k) * STRIDE]; }
  RCSH_HD double& link(int k) const { return base[(K0 + k) * STRIDE]; }
  RCSH_HD double& qpre(int i) const { return base[(P0 + i) * STRIDE]; }
  RCSH_HD double& S(int i, int k) const { return base[(S0 + 6 * i + k) * STRIDE]; }
  RCSH_HD double& X(int k) const { return base[(X0 + k) * STRIDE]; }
  RCSH_HD double& I(int i, int k) const { return base[(I0 + 10 * i + k) * STRIDE]; }
  RCSH_HD double& f(int i, int k) const { return base[(F0 + 6 * i + k) * STRIDE]; }
  RCSH_HD double& hg(int i, int k) const { return base[(H0 + 3 * i + k) * STRIDE]; }
  RCSH_HD double& M(int k) const { return base[(M0 + k) * STRIDE]; }  // packed lower triangle, incl. armature
};

// ---- results of the position + velocity stage that the rest of the substep consumes
// (the mass matrix goes to Stage::M)
template <class T>
struct Smooth {
  double bias[T::NL];   // Coriolis + centrifugal + gravity
  double gc[T::NL];     // gravity-compensation generalized force
  double linkR[9];      // world frame of the link carrying the attachment site
  double linkP[3];
};

// Position + velocity stage.  Forward sweep root->leaves builds, per link, the world frame, the
// motion axis S (kept in registers), the spatial inertia I and the bias wrench f (staged); the backward
// sweep leaves->root carries running subtree sums in registers and projects them on the axes.
template <class T, int STRIDE>
RCSH_HD void smooth_dynamics(const DevModel& m, const double* q, const double* qd, const Stage<T, STRIDE>& st,
                             Smooth<T>& out) {
  constexpr int NL = T::NL;
  // frames of the arm chain tip are reused by both fingers, so one running copy suffices
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {0, 0, 0};
  double vel[6] = {0, 0, 0, 0, 0, 0};
  double acc[6] = {0, 0, 0, -m.gravity[0], -m.gravity[1], -m.gravity[2]};
  double Rt[9], pt[3], velt[6], acct[6];  // arm tip, kept for the fingers
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    if (T::GRIP && i == T::NARM) {
#pragma unroll
      for (int k = 0; k < 9; ++k) Rt[k] = R[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) pt[k] = p[k];
#pragma unroll
      for (int k = 0; k < 6; ++k) { velt[k] = vel[k]; acct[k] = acc[k]; }
    }
    if (T::GRIP && i > T::NARM) {
#pragma unroll
      for (int k = 0; k < 9; ++k) R[k] = Rt[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) p[k] = pt[k];
#pragma unroll
      for (int k = 0; k < 6; ++k) { vel[k] = velt[k]; acc[k] = acct[k]; }
    }
    // frame at qpos0
    double o[3], R0[9];
    mulmv(R, m.pos0[i], o);
    o[0] += p[0]; o[1] += p[1]; o[2] += p[2];
    mulmm(R, m.rot0[i], R0);
    double ax[3];
    const double dq = q[i] - m.qpos0[i];
    double Si[6];
    if (m.axis_z[i]) {
      // hinge about the link's +z through the link origin (every FR3 / xArm7 joint): the rotation only mixes
      // the first two columns of R0, the world axis is its third column, the anchor is the origin
      double s, c;
      fast_sincos(dq, &s, &c);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        R[3 * r + 0] = c * R0[3 * r + 0] + s * R0[3 * r + 1];
        R[3 * r + 1] = c * R0[3 * r + 1] - s * R0[3 * r + 0];
        R[3 * r + 2] = R0[3 * r + 2];
        ax[r] = R0[3 * r + 2];
        p[r] = o[r];
      }
      Si[0] = ax[0]; Si[1] = ax[1]; Si[2] = ax[2];
      cross3(o, ax, Si + 3);
    } else if (m.jtype[i] == kSlide) {
      mulmv(R0, m.axis[i], ax);
#pragma unroll
      for (int k = 0; k < 9; ++k) R[k] = R0[k];
      p[0] = o[0] + ax[0] * dq; p[1] = o[1] + ax[1] * dq; p[2] = o[2] + ax[2] * dq;
      Si[0] = 0; Si[1] = 0; Si[2] = 0; Si[3] = ax[0]; Si[4] = ax[1]; Si[5] = ax[2];
    } else {
      // general hinge: Rodrigues rotation about the link-frame axis
      mulmv(R0, m.axis[i], ax);
      double s, c;
      fast_sincos(dq, &s, &c);
      const double* a = m.axis[i];
      const double t = 1.0 - c;
      double Q[9] = {c + t * a[0] * a[0],        t * a[0] * a[1] - s * a[2], t * a[0] * a[2] + s * a[1],
                     t * a[0] * a[1] + s * a[2], c + t * a[1] * a[1],        t * a[1] * a[2] - s * a[0],
                     t * a[0] * a[2] - s * a[1], t * a[1] * a[2] + s * a[0], c + t * a[2] * a[2]};
      double anchor[3], rj[3];
      mulmv(R0, m.jpos[i], anchor);
      anchor[0] += o[0]; anchor[1] += o[1]; anchor[2] += o[2];
      mulmm(R0, Q, R);
      mulmv(R, m.jpos[i], rj);
      p[0] = anchor[0] - rj[0]; p[1] = anchor[1] - rj[1]; p[2] = anchor[2] - rj[2];
      Si[0] = ax[0]; Si[1] = ax[1]; Si[2] = ax[2];
      cross3(anchor, ax, Si + 3);
    }
    if (i == m.site_link) {
#pragma unroll
      for (int k = 0; k < 9; ++k) out.linkR[k] = R[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) out.linkP[k] = p[k];
    }
    // spatial inertia about the world origin
    double c[3], cg[3];
    mulmv(R, m.com[i], c);
    c[0] += p[0]; c[1] += p[1]; c[2] += p[2];
    if (m.gc_same_com[i]) {  // uniform gravcomp over the link: its centre is the centre of mass
      cg[0] = c[0]; cg[1] = c[1]; cg[2] = c[2];
    } else {
      mulmv(R, m.gccom[i], cg);
      cg[0] += p[0]; cg[1] += p[1]; cg[2] += p[2];
    }
    st.hg(i, 0) = m.gcm[i] * cg[0];
    st.hg(i, 1) = m.gcm[i] * cg[1];
    st.hg(i, 2) = m.gcm[i] * cg[2];
    double Ii[10];
    {
      const double* J = m.inertia[i];
      const double Jm[9] = {J[0], J[3], J[4], J[3], J[1], J[5], J[4], J[5], J[2]};
      double Tm[9];
      mulmm(R, Jm, Tm);
      const double ms = m.mass[i];
      Ii[0] = Tm[0] * R[0] + Tm[1] * R[1] + Tm[2] * R[2] + ms * (c[1] * c[1] + c[2] * c[2]);
      Ii[1] = Tm[3] * R[3] + Tm[4] * R[4] + Tm[5] * R[5] + ms * (c[0] * c[0] + c[2] * c[2]);
      Ii[2] = Tm[6] * R[6] + Tm[7] * R[7] + Tm[8] * R[8] + ms * (c[0] * c[0] + c[1] * c[1]);
      Ii[3] = Tm[0] * R[3] + Tm[1] * R[4] + Tm[2] * R[5] - ms * c[0] * c[1];
      Ii[4] = Tm[0] * R[6] + Tm[1] * R[7] + Tm[2] * R[8] - ms * c[0] * c[2];
      Ii[5] = Tm[3] * R[6] + Tm[4] * R[7] + Tm[5] * R[8] - ms * c[1] * c[2];
      Ii[6] = ms * c[0]; Ii[7] = ms * c[1]; Ii[8] = ms * c[2];
      Ii[9] = ms;
#pragma unroll
      for (int k = 0; k < 10; ++k) st.I(i, k) = Ii[k];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) st.S(i, k) = Si[k];
    // velocity, bias acceleration (S x S = 0, so the parent's velocity is enough), bias wrench
    double sd[6];
    cross_motion(vel, Si, sd);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      vel[k] += Si[k] * qd[i];
      acc[k] += sd[k] * qd[i];
    }
    double Ia[6], Iv[6], vf[6];
    inert_mul(Ii, acc, Ia);
    inert_mul(Ii, vel, Iv);
    cross_force(vel, Iv, vf);
#pragma unroll
    for (int k = 0; k < 6; ++k) st.f(i, k) = Ia[k] + vf[k];
  }
  stage_fence();
  // backward sweep: Ic / fs / hs are the sums over the links visited so far.  Links are visited
  // leaves first (fingers, then the arm from tip to base), so for an arm link the running sums are
  // exactly its subtree; a finger is a leaf and projects its own values only.
  const double* g = m.gravity;
  const double ng[3] = {-g[0], -g[1], -g[2]};
  double Ic[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, fs[6] = {0, 0, 0, 0, 0, 0}, hs[3] = {0, 0, 0};
#pragma unroll
  for (int i = NL - 1; i >= 0; --i) {
    double Il[10], fl[6], hl[3];
#pragma unroll
    for (int k = 0; k < 10; ++k) { Il[k] = st.I(i, k); Ic[k] += Il[k]; }
#pragma unroll
    for (int k = 0; k < 6; ++k) { fl[k] = st.f(i, k); fs[k] += fl[k]; }
#pragma unroll
    for (int k = 0; k < 3; ++k) { hl[k] = st.hg(i, k); hs[k] += hl[k]; }
    const bool leaf = T::GRIP && i >= T::NARM;
    const double* Iu = leaf ? Il : Ic;
    const double* fu = leaf ? fl : fs;
    const double* hu = leaf ? hl : hs;
    double F[6], Sl[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) Sl[k] = st.S(i, k);
    inert_mul(Iu, Sl, F);
    st.M(tri(i, i)) = dot6(Sl, F) + m.armature[i];
    // ancestors: for the arm that is every j < i; a finger's ancestors are all arm links
#pragma unroll
    for (int j = (i >= T::NARM ? T::NARM - 1 : i - 1); j >= 0; --j) {
      double Sj[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) Sj[k] = st.S(j, k);
      st.M(tri(i, j)) = dot6(Sj, F);
    }
    if (T::GRIP && i == T::NARM + 1) st.M(tri(i, i - 1)) = 0.0;  // the fingers are siblings
    out.bias[i] = dot6(Sl, fu);
    // gravity compensation wrench of the subtree: force -g * sum(gcm), moment sum(gcm * c) x (-g)
    double w[6];
    cross3(hu, ng, w);
    w[3] = m.gcm_sub[i] * ng[0]; w[4] = m.gcm_sub[i] * ng[1]; w[5] = m.gcm_sub[i] * ng[2];
    out.gc[i] = dot6(Sl, w);
  }
}

// packed LDL^T of an SPD matrix, in place: strict lower part <- L, diagonal <- 1/D
template <int N>
RCSH_HD void ldl_factor(double* A) {
  double D[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    double w[N];
    double d = A[tri(j, j)];
#pragma unroll
    for (int k = 0; k < j; ++k) {
      w[k] = A[tri(j, k)] * D[k];
      d -= A[tri(j, k)] * w[k];
    }
    D[j] = d;
    const double inv = fast_rcp(d);
#pragma unroll
    for (int i = j + 1; i < N; ++i) {
      double t = A[tri(i, j)];
#pragma unroll
      for (int k = 0; k < j; ++k) t -= A[tri(i, k)] * w[k];
      A[tri(i, j)] = t * inv;
    }
    A[tri(j, j)] = inv;
  }
}
template <int N>
RCSH_HD void ldl_solve(const double* A, double* x) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int k = 0; k < i; ++k) x[i] -= A[tri(i, k)] * x[k];
  }
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] *= A[tri(i, i)];
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
#pragma unroll
    for (int k = i + 1; k < N; ++k) x[i] -= A[tri(k, i)] * x[k];
  }
}

// 1/x to full double precision without the IEEE division sequence (hardware seed + two Newton steps)
RCSH_HD double fast_rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
#else
  return 1.0 / x;
#endif
}

// sin and cos of a joint angle.  Cody-Waite reduction by pi/2 (33-bit pieces, exact for |x| < 2^20)
// followed by the classic minimax kernels on [-pi/4, pi/4]; < 1 ulp, no table, no slow path.
RCSH_HD void fast_sincos(double x, double* sn, double* cs) {
  const double fn = rint(x * 6.36619772367581382433e-01);
  const double r = fma(-fn, 1.57079632673412561417e+00, x);
  const double w = fn * 6.07710050650619224932e-11;
  const double y0 = r - w;
  const double y1 = (r - y0) - w;
  const double z = y0 * y0;
  // sin kernel
  const double v = z * y0;
  const double rs = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 +
                    z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
  const double ks = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * -1.66666666666666324348e-01);
  // cos kernel
  const double w2 = z * z;
  const double rc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * 2.48015872894767294178e-05)) +
                    (w2 * w2) * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11));
  const double hz = 0.5 * z;
  const double w1 = 1.0 - hz;
  const double kc = w1 + (((1.0 - w1) - hz) + (z * rc - y0 * y1));
  const int n = ((int)fn) & 3;
  const double s_ = (n & 1) ? kc : ks;
  const double c_ = (n & 1) ? ks : kc;
  *sn = (n & 2) ? -s_ : s_;
  *cs = ((n + 1) & 2) ? -c_ : c_;
}

// general-power branch of the impedance sigmoid, kept out of line: the RCS scenes use power 2
#if defined(__HIP__)
__host__ __device__ __attribute__((noinline))
#endif
inline double impedance_general(const Imp& p, double x) {
  if (x <= p.mid) return pow(x, p.power) / pow(p.mid, p.power - 1);
  return 1 - pow(1 - x, p.power) / pow(1 - p.mid, p.power - 1);
}

// solimp -> impedance at distance |pos - margin| (the sigmoid MuJoCo documents for solimp)
RCSH_HD double impedance(const Imp& p, double pos, double margin) {
  if (p.mode == 0) return 0.5 * (p.d0 + p.d1);
  const double x = fabs((pos - margin) * p.inv_width);
  if (x >= 1) return p.d1;
  if (x <= 0) return p.d0;
  double y;
  if (p.mode == 1) y = x;
  else if (p.mode == 2) y = x <= p.mid ? x * x * p.inv_mid : 1 - (1 - x) * (1 - x) * p.inv_1mmid;
  else y = impedance_general(p, x);
  return p.d0 + y * (p.d1 - p.d0);
}
// regulariser R = (1 - imp) / imp * diagApprox, floored; returns D = 1 / R
RCSH_HD double row_D(double imp, double invweight) {
  const double num = (1 - imp) * invweight;
  return num < kMinVal * imp ? 1.0 / kMinVal : imp * fast_rcp(num);
}

// One physics substep on the environment parked in `st`: reads qpos / qvel / ctrl from the staging
// column, advances them by one timestep and writes them back, together with the world frame of the
// attachment-site link computed from the PRE-step qpos (the reference reads site_xpos / site_xmat of
// the last mj_step1).  Everything that is not needed between two phases is parked in
// the column, so the register allocator only ever sees one phase's working set.
template <class T, int STRIDE>
RCSH_HD void substep(const DevModel& m, const Stage<T, STRIDE>& st) {
  constexpr int NL = T::NL;
  constexpr int NA = T::NARM;
  const double h = m.timestep;
  double smooth[NL];        // qfrc_smooth
  uint32_t clampmask = 0;   // bit i: arm actuator i saturated its forcerange (no velocity derivative)
  double gblock = 0.0;      // -bias_vel of the gripper actuator (2x2 block coef_a * coef_b * gblock)
  double eqD = 0, eqAref = 0, eqJ1 = 0;  // coupling row = e_f1 + eqJ1 * e_f2
  uint32_t limrows = 0;     // bit i: joint i has a limit row this substep
  {
    double q[NL], qd[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) { q[i] = st.q(i); qd[i] = st.v(i); st.qpre(i) = q[i]; }
    Smooth<T> sm;
    smooth_dynamics<T, STRIDE>(m, q, qd, st, sm);
#pragma unroll
    for (int k = 0; k < 9; ++k) st.link(k) = sm.linkR[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) st.link(9 + k) = sm.linkP[k];

    // ---- actuation: affine actuators, force limits, actuator-side gravity compensation, joint clamp
    double tau[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) tau[i] = 0;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      if (!m.arm_has_act[i]) continue;
      double c = st.c(i);
      if (m.arm_ctrllimited[i]) c = clampd(c, m.arm_ctrlrange[i][0], m.arm_ctrlrange[i][1]);
      const double gear = m.arm_gear[i];
      double force = m.arm_gain[i] * c;
      if (m.arm_biasaffine[i]) force += m.arm_bias[i][0] + m.arm_bias[i][1] * (gear * q[i]) + m.arm_bias[i][2] * (gear * qd[i]);
      if (m.arm_forcelimited[i]) {
        if (force <= m.arm_forcerange[i][0] || force >= m.arm_forcerange[i][1]) clampmask |= 1u << i;
        force = clampd(force, m.arm_forcerange[i][0], m.arm_forcerange[i][1]);
      }
      tau[i] = gear * force;
    }
    if (T::GRIP && m.grp_has_act) {
      double c = st.c(NA);
      if (m.grp_ctrllimited) c = clampd(c, m.grp_ctrlrange[0], m.grp_ctrlrange[1]);
      const double len = m.grp_coef[0] * q[NA] + m.grp_coef[1] * q[NA + 1];
      const double vel = m.grp_coef[0] * qd[NA] + m.grp_coef[1] * qd[NA + 1];
      double force = m.grp_gain * c;
      if (m.grp_biasaffine) force += m.grp_bias[0] + m.grp_bias[1] * len + m.grp_bias[2] * vel;
      bool clamped = false;
      if (m.grp_forcelimited) {
        clamped = force <= m.grp_forcerange[0] || force >= m.grp_forcerange[1];
        force = clampd(force, m.grp_forcerange[0], m.grp_forcerange[1]);
      }
      tau[NA] += m.grp_coef[0] * force;
      tau[NA + 1] += m.grp_coef[1] * force;
      if (m.grp_biasaffine && !clamped) gblock = -m.grp_bias[2];
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      double passive = -m.damping[i] * qd[i];
      if (m.actgravcomp[i]) tau[i] += sm.gc[i]; else passive += sm.gc[i];
      if (m.actfrclimited[i]) tau[i] = clampd(tau[i], m.actfrcrange[i][0], m.actfrcrange[i][1]);
      smooth[i] = passive - sm.bias[i] + tau[i];
    }

    // ---- constraint rows: finger coupling (equality, always active) and joint limits (one-sided)
    if (T::GRIP && m.eq_active) {
      const double* pc = m.eq_polycoef;
      const double dif = q[NA + 1] - m.qpos0[NA + 1];
      const double poly = pc[0] + dif * (pc[1] + dif * (pc[2] + dif * (pc[3] + dif * pc[4])));
      const double deriv = pc[1] + dif * (2 * pc[2] + dif * (3 * pc[3] + dif * 4 * pc[4]));
      const double pos = q[NA] - m.qpos0[NA] - poly;
      eqJ1 = -deriv;
      const double imp = impedance(m.eq_imp, pos, 0.0);
      eqD = row_D(imp, m.invweight0[NA] + m.invweight0[NA + 1]);
      eqAref = -m.eq_K * imp * pos - m.eq_B * (qd[NA] + eqJ1 * qd[NA + 1]);
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      if (!m.limited[i]) continue;
      const double dlo = q[i] - m.range[i][0], dhi = m.range[i][1] - q[i];
      double dist = 0, sgn = 0;
      if (dlo < m.margin[i]) { dist = dlo; sgn = 1; }
      else if (dhi < m.margin[i]) { dist = dhi; sgn = -1; }
      if (sgn != 0) {
        const double imp = impedance(m.lim_imp[i], dist, m.margin[i]);
        st.lim(i, 0) = row_D(imp, m.invweight0[i]);
        st.lim(i, 1) = -m.lim_K[i] * imp * (dist - m.margin[i]) - m.lim_B[i] * (sgn * qd[i]);
        st.lim(i, 2) = sgn;
        limrows |= 1u << i;
      }
    }
  }
  stage_fence();

  // ---- qacc = argmin 1/2 |qacc - M^-1 smooth|_M^2 + sum s_i(J_i qacc - aref_i), s_i quadratic (coupling
  // row) or one-sided quadratic (limit rows).  Newton on the piecewise-quadratic cost: the minimiser under a
  // guessed active set is the answer if the set it lands in equals the guess; otherwise an exact line search
  // along the Newton direction is taken from the current iterate and the step repeated (finite convergence).
  double fc[NL];  // qfrc_constraint
#pragma unroll
  for (int i = 0; i < NL; ++i) fc[i] = 0;
  if ((T::GRIP && m.eq_active) || limrows) {
    const bool has_eq = T::GRIP && m.eq_active;
    double x[NL];
    uint32_t act = limrows;  // first guess: every limit row that exists is active
    bool have_x = false;
    for (int iter = 0; iter < 16; ++iter) {
      double xn[NL];
      {
        double H[T::NTRI];
#pragma unroll
        for (int k = 0; k < T::NTRI; ++k) H[k] = st.M(k);
#pragma unroll
        for (int i = 0; i < NL; ++i) {
          xn[i] = smooth[i];
          if (act & (1u << i)) {
            const double D = st.lim(i, 0);
            H[tri(i, i)] += D;
            xn[i] += st.lim(i, 2) * D * st.lim(i, 1);
          }
        }
        if (has_eq) {
          H[tri(NA, NA)] += eqD;
          H[tri(NA + 1, NA)] += eqD * eqJ1;
          H[tri(NA + 1, NA + 1)] += eqD * eqJ1 * eqJ1;
          xn[NA] += eqD * eqAref;
          xn[NA + 1] += eqD * eqAref * eqJ1;
        }
        ldl_factor<NL>(H);
        ldl_solve<NL>(H, xn);
      }
      uint32_t now = 0;
#pragma unroll
      for (int i = 0; i < NL; ++i)
        if ((limrows & (1u << i)) && st.lim(i, 2) * xn[i] - st.lim(i, 1) < 0) now |= 1u << i;
      if (now == act || !have_x) {
        // either the exact minimiser, or the starting point of the line-search iteration
#pragma unroll
        for (int i = 0; i < NL; ++i) x[i] = xn[i];
        have_x = true;
        if (now == act) break;
        act = now;
        continue;
      }
      // exact line search from x along d = xn - x:
      // phi'(a) = p0 + a p1 + sum_{rows active at a} D_i jd_i (jar_i + a jd_i)
      double d[NL], jar[NL], jd[NL];
      double p0 = 0, p1 = 0;
#pragma unroll
      for (int i = 0; i < NL; ++i) d[i] = xn[i] - x[i];
#pragma unroll
      for (int r = 0; r < NL; ++r) {
        double mx = -smooth[r], md = 0;
#pragma unroll
        for (int c = 0; c < NL; ++c) {
          const double mrc = st.M(r >= c ? tri(r, c) : tri(c, r));
          mx += mrc * x[c];
          md += mrc * d[c];
        }
        p0 += mx * d[r];
        p1 += md * d[r];
      }
      if (has_eq) {
        const double je = x[NA] + eqJ1 * x[NA + 1] - eqAref, jde = d[NA] + eqJ1 * d[NA + 1];
        p0 += eqD * je * jde;
        p1 += eqD * jde * jde;
      }
      uint32_t on = 0;
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        jar[i] = 0; jd[i] = 0;
        if (limrows & (1u << i)) {
          const double sgn = st.lim(i, 2);
          jar[i] = sgn * x[i] - st.lim(i, 1);
          jd[i] = sgn * d[i];
          if (jar[i] < 0 || (jar[i] == 0 && jd[i] < 0)) on |= 1u << i;
        }
      }
      double alpha = 0;
      for (int guard = 0; guard < NL + 2; ++guard) {
        double c0 = p0, c1 = p1, a_next = INFINITY;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
          if (!(limrows & (1u << i))) continue;
          const double D = st.lim(i, 0);
          if (on & (1u << i)) { c0 += D * jar[i] * jd[i]; c1 += D * jd[i] * jd[i]; }
          if (jd[i] != 0) {
            const double ab = -jar[i] / jd[i];
            if (ab > alpha && ab < a_next) a_next = ab;
          }
        }
        const double a_star = -c0 / c1;
        if (a_star <= a_next) { if (a_star > alpha) alpha = a_star; break; }
        alpha = a_next;
#pragma unroll
        for (int i = 0; i < NL; ++i)
          if ((limrows & (1u << i)) && jd[i] != 0 && -jar[i] / jd[i] == a_next) on ^= 1u << i;
      }
      act = 0;
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        x[i] += alpha * d[i];
        if ((limrows & (1u << i)) && st.lim(i, 2) * x[i] - st.lim(i, 1) < 0) act |= 1u << i;
      }
    }
#pragma unroll
    for (int i = 0; i < NL; ++i)
      if ((limrows & (1u << i))) {
        const double sgn = st.lim(i, 2);
        const double r = sgn * x[i] - st.lim(i, 1);
        if (r < 0) fc[i] = -sgn * st.lim(i, 0) * r;
      }
    if (has_eq) {
      const double fe = -eqD * (x[NA] + eqJ1 * x[NA + 1] - eqAref);
      fc[NA] += fe;
      fc[NA + 1] += fe * eqJ1;
    }
  }
  stage_fence();

  // ---- implicitfast: (M - h dF/dqd) qacc = smooth + constraint, then semi-implicit Euler
  {
    double A[T::NTRI];
#pragma unroll
    for (int k = 0; k < T::NTRI; ++k) A[k] = st.M(k);
    double rhs[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      double d = m.damping[i];
      if (i < NA && m.arm_has_act[i] && m.arm_biasaffine[i] && !(clampmask & (1u << i)))
        d -= m.arm_gear[i] * m.arm_gear[i] * m.arm_bias[i][2];
      A[tri(i, i)] += h * d;
      rhs[i] = smooth[i] + fc[i];
    }
    if (T::GRIP) {
      A[tri(NA, NA)] += h * gblock * m.grp_coef[0] * m.grp_coef[0];
      A[tri(NA + 1, NA)] += h * gblock * m.grp_coef[0] * m.grp_coef[1];
      A[tri(NA + 1, NA + 1)] += h * gblock * m.grp_coef[1] * m.grp_coef[1];
    }
    ldl_factor<NL>(A);
    ldl_solve<NL>(A, rhs);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const double v = st.v(i) + h * rhs[i];
      st.v(i) = v;
      st.q(i) += h * v;
    }
  }
  stage_fence();
}

}  // namespace rcsh
