// dyn.h -- shared math of the kernels (3-vectors, spatial inertia, packed LDL', fast reciprocal / sincos, the impedance
// sigmoid, the robot archetypes) and the SERIAL formulation of the smooth dynamics of one environment
// (smooth_dynamics: kinematics, composite-inertia mass matrix, bias forces, gravity compensation -- what mj_step1
// computes for the RCS scenes, reference call sites src/sim/sim.cpp:110,112), all spatial quantities as Pluecker vectors
// about the WORLD ORIGIN so that the composite-inertia and force recursions are plain sums.
//
// The serial formulation is instantiated on the HOST only (model.cpp): dof_invweight0 = diag(M(qpos0)^-1) and
// body_invweight0 at model-finalise time, the constants MuJoCo derives when it compiles a model; and by the per-lane CLIK
// utilities (ik.h).  Through round 1 it was also a second GPU kernel, one lane per environment; that kernel lost at every
// batch size, stepped neither dry friction nor free bodies, and was removed.  No stepping entry point runs on the CPU.
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

}  // namespace rcsh
