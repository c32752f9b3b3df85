// dyn4.h -- the physics substep split across the FOUR waves of a workgroup (one per SIMD of a CU).
//
// One environment is still one lane index, but lane l of each of the 4 waves works on environment l at the same
// time, each wave running a different ROLE of the substep; hand-off is through the environment's LDS column and
// workgroup barriers.  At the headline batch (4096 environments) only 256 of the 1024 SIMDs hold a wave when one
// wave does everything; the launch is bound by the length of that wave's dependent FP64 instruction stream, not by
// chip throughput.  Splitting the stream four ways shortens the critical path per substep from ~6.5k to ~2.7k
// VALU instructions:
//
//   phase A   W0,W1,W2: frame chain -> world-origin spatial inertia + gravcomp moment of "their" links
//             W3:       frame chain -> motion axes S of all links, velocity / bias-acceleration chain
//   barrier
//   phase B   W0..W3:   bias wrench f_i = I_i a_i + v_i x* I_i v_i of their links
//   barrier
//   phase C   W0..W3:   suffix sums of I / f / gravcomp from the column, then their rows of M, bias, gravcomp
//   barrier
//   phase D   W3: actuation -> qfrc_smooth      W0: LDL^T of the implicit matrix      W1: constraint rows, LDL^T of H
//   barrier   W1: constraint solve -> qfrc_constraint
//   barrier   W0: implicit solve, semi-implicit Euler, write back
//   barrier
//
// The arithmetic of every quantity is the same expression as in dyn.h (same helpers), so both paths agree with
// the oracle to round-off; which path runs is a launch-time choice (rcs_hip.hip).
#pragma once
#include "dyn.h"

namespace rcsh {

// Scheduling fence: everything written above it is issued before anything below it.  Used right after a batch of
// LDS reads so the reads go out back to back and ONE s_waitcnt covers them; without it the compiler sinks each
// read next to its first use and every read exposes the full LDS round trip (measured: 1.2 reads per wait).
RCSH_HD void sched_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_sched_barrier(0);
#endif
}

// per-link kinematic constants of the frame chain, copied from the (LDS-resident) model tables in one batch
template <int N>
struct ChainConsts {
  double pos0[N][3], rot0[N][9], dq[N];
  int32_t axis_z[N], jtype[N];
};
template <int N, class ST>
RCSH_HD void load_chain_consts(const DevModel& m, const ST& st, int count, ChainConsts<N>& c) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (i >= count) break;
#pragma unroll
    for (int k = 0; k < 3; ++k) c.pos0[i][k] = m.pos0[i][k];
#pragma unroll
    for (int k = 0; k < 9; ++k) c.rot0[i][k] = m.rot0[i][k];
    c.dq[i] = st.q(i) - m.qpos0[i];
    c.axis_z[i] = m.axis_z[i];
    c.jtype[i] = m.jtype[i];
  }
}

// LDS column of one environment for the 4-wave path
template <class T, int STRIDE>
struct Stage4 {
  static constexpr int NL = T::NL;
  static constexpr int Q0 = 0;                  // qpos
  static constexpr int V0 = Q0 + NL;            // qvel
  static constexpr int C0 = V0 + NL;            // ctrl
  static constexpr int P0 = C0 + T::NU;         // qpos seen by the last position stage
  static constexpr int I0 = P0 + NL;            // spatial inertia per link (10)
  static constexpr int H0 = I0 + 10 * NL;       // gravcomp first moment per link (3)
  static constexpr int S0 = H0 + 3 * NL;        // motion axis per link (6)
  static constexpr int VA0 = S0 + 6 * NL;       // velocity (6) + bias acceleration (6) per link
  static constexpr int F0 = VA0 + 12 * NL;      // bias wrench per link (6)
  static constexpr int M0 = F0 + 6 * NL;        // mass matrix, packed lower triangle
  static constexpr int B0 = M0 + T::NTRI;       // qfrc_bias
  static constexpr int G0 = B0 + NL;            // qfrc_gravcomp
  static constexpr int SM0 = G0 + NL;           // qfrc_smooth
  static constexpr int FC0 = SM0 + NL;          // qfrc_constraint
  static constexpr int K0 = FC0 + NL;           // frame of the site link: R(9) p(3)
  static constexpr int A0 = K0 + 12;            // 1.0 while the environment still steps in this launch
  static constexpr int X0 = A0 + 1;             // caller's slots
  static constexpr int NX = 6 + 2 * T::NARM;
  static constexpr int COUNT = X0 + NX;
  double* base;
  RCSH_HD double& at(int k) const { return base[k * STRIDE]; }
  RCSH_HD double& q(int i) const { return at(Q0 + i); }
  RCSH_HD double& v(int i) const { return at(V0 + i); }
  RCSH_HD double& c(int i) const { return at(C0 + i); }
  RCSH_HD double& qpre(int i) const { return at(P0 + i); }
  RCSH_HD double& I(int i, int k) const { return at(I0 + 10 * i + k); }
  RCSH_HD double& hg(int i, int k) const { return at(H0 + 3 * i + k); }
  RCSH_HD double& S(int i, int k) const { return at(S0 + 6 * i + k); }
  RCSH_HD double& vel(int i, int k) const { return at(VA0 + 12 * i + k); }
  RCSH_HD double& acc(int i, int k) const { return at(VA0 + 12 * i + 6 + k); }
  RCSH_HD double& f(int i, int k) const { return at(F0 + 6 * i + k); }
  RCSH_HD double& M(int k) const { return at(M0 + k); }
  RCSH_HD double& bias(int i) const { return at(B0 + i); }
  RCSH_HD double& gc(int i) const { return at(G0 + i); }
  RCSH_HD double& smooth(int i) const { return at(SM0 + i); }
  RCSH_HD double& fc(int i) const { return at(FC0 + i); }
  RCSH_HD double& link(int k) const { return at(K0 + k); }
  RCSH_HD double& active() const { return at(A0); }
  RCSH_HD double& X(int k) const { return at(X0 + k); }
};

// which wave computes the inertia / bias wrench of link i, and which wave owns row i of M
template <class T>
RCSH_HD constexpr int inertia_owner(int i) { return i * 3 / T::NL; }  // contiguous thirds: W0, W1, W2
template <class T>
RCSH_HD constexpr int wrench_owner(int i) { return i == 0 ? 3 : inertia_owner<T>(i); }  // W3 takes link 0's
template <class T>
RCSH_HD constexpr int row_owner(int i) { return (i < T::NL - 1 - i ? i : T::NL - 1 - i) % 4; }

// one link of the frame chain: like the kinematic part of smooth_dynamics, also returning axis and anchor
template <int N>
RCSH_HD void chain_step(const DevModel& m, const ChainConsts<N>& cc, int i, double* R, double* p, double* ax, double* anchor) {
  double o[3], R0[9];
  mulmv(R, cc.pos0[i], o);
  o[0] += p[0]; o[1] += p[1]; o[2] += p[2];
  mulmm(R, cc.rot0[i], R0);
  const double dq = cc.dq[i];
  if (cc.axis_z[i]) {
    double s, c;
    fast_sincos(dq, &s, &c);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      R[3 * r + 0] = c * R0[3 * r + 0] + s * R0[3 * r + 1];
      R[3 * r + 1] = c * R0[3 * r + 1] - s * R0[3 * r + 0];
      R[3 * r + 2] = R0[3 * r + 2];
      ax[r] = R0[3 * r + 2];
      p[r] = o[r];
      anchor[r] = o[r];
    }
  } else if (cc.jtype[i] == kSlide) {
    mulmv(R0, m.axis[i], ax);
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = R0[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) { p[k] = o[k] + ax[k] * dq; anchor[k] = o[k]; }
  } else {
    mulmv(R0, m.axis[i], ax);
    double s, c;
    fast_sincos(dq, &s, &c);
    const double* a = m.axis[i];
    const double t = 1.0 - c;
    const double Q[9] = {c + t * a[0] * a[0],        t * a[0] * a[1] - s * a[2], t * a[0] * a[2] + s * a[1],
                         t * a[0] * a[1] + s * a[2], c + t * a[1] * a[1],        t * a[1] * a[2] - s * a[0],
                         t * a[0] * a[2] - s * a[1], t * a[1] * a[2] + s * a[0], c + t * a[2] * a[2]};
    double rj[3];
    mulmv(R0, m.jpos[i], anchor);
    anchor[0] += o[0]; anchor[1] += o[1]; anchor[2] += o[2];
    mulmm(R0, Q, R);
    mulmv(R, m.jpos[i], rj);
    p[0] = anchor[0] - rj[0]; p[1] = anchor[1] - rj[1]; p[2] = anchor[2] - rj[2];
  }
}

// ---- phase A, waves 0-2: frames up to the last link of the wave's third; inertia + gravcomp moment of its links
template <class T, int STRIDE, int W>
RCSH_HD void phaseA_inertia(const DevModel& m, const Stage4<T, STRIDE>& st, bool stepping) {
  constexpr int NL = T::NL;
  // last link this wave needs a frame for
  constexpr int kLast = W == 0 ? (NL + 2) / 3 - 1 : (W == 1 ? (2 * NL + 2) / 3 - 1 : NL - 1);
  ChainConsts<NL> cc;
  load_chain_consts<NL>(m, st, kLast + 1, cc);
  // inertial constants of the wave's own links
  double kcom[NL][3], kin[NL][6], kmass[NL], kgcm[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    if (inertia_owner<T>(i) != W) continue;
#pragma unroll
    for (int k = 0; k < 3; ++k) kcom[i][k] = m.com[i][k];
#pragma unroll
    for (int k = 0; k < 6; ++k) kin[i][k] = m.inertia[i][k];
    kmass[i] = m.mass[i];
    kgcm[i] = m.gcm[i];
  }
  sched_fence();
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {0, 0, 0};
  double Rt[9], pt[3];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    // links after the wave's last one are not needed
    bool later_mine = false;
#pragma unroll
    for (int j = i; j < NL; ++j) later_mine = later_mine || inertia_owner<T>(j) == W;
    if (!later_mine) break;
    if (T::GRIP && i == T::NARM) {
#pragma unroll
      for (int k = 0; k < 9; ++k) Rt[k] = R[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) pt[k] = p[k];
    }
    if (T::GRIP && i > T::NARM) {
#pragma unroll
      for (int k = 0; k < 9; ++k) R[k] = Rt[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) p[k] = pt[k];
    }
    double ax[3], anchor[3];
    chain_step<NL>(m, cc, i, R, p, ax, anchor);
    if (inertia_owner<T>(i) != W) continue;
    if (i == m.site_link && stepping) {
#pragma unroll
      for (int k = 0; k < 9; ++k) st.link(k) = R[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) st.link(9 + k) = p[k];
    }
    double c[3], cg[3];
    mulmv(R, kcom[i], c);
    c[0] += p[0]; c[1] += p[1]; c[2] += p[2];
    if (m.gc_same_com[i]) {
      cg[0] = c[0]; cg[1] = c[1]; cg[2] = c[2];
    } else {
      mulmv(R, m.gccom[i], cg);
      cg[0] += p[0]; cg[1] += p[1]; cg[2] += p[2];
    }
    st.hg(i, 0) = kgcm[i] * cg[0];
    st.hg(i, 1) = kgcm[i] * cg[1];
    st.hg(i, 2) = kgcm[i] * cg[2];
    const double* J = kin[i];
    const double Jm[9] = {J[0], J[3], J[4], J[3], J[1], J[5], J[4], J[5], J[2]};
    double Tm[9];
    mulmm(R, Jm, Tm);
    const double ms = kmass[i];
    st.I(i, 0) = Tm[0] * R[0] + Tm[1] * R[1] + Tm[2] * R[2] + ms * (c[1] * c[1] + c[2] * c[2]);
    st.I(i, 1) = Tm[3] * R[3] + Tm[4] * R[4] + Tm[5] * R[5] + ms * (c[0] * c[0] + c[2] * c[2]);
    st.I(i, 2) = Tm[6] * R[6] + Tm[7] * R[7] + Tm[8] * R[8] + ms * (c[0] * c[0] + c[1] * c[1]);
    st.I(i, 3) = Tm[0] * R[3] + Tm[1] * R[4] + Tm[2] * R[5] - ms * c[0] * c[1];
    st.I(i, 4) = Tm[0] * R[6] + Tm[1] * R[7] + Tm[2] * R[8] - ms * c[0] * c[2];
    st.I(i, 5) = Tm[3] * R[6] + Tm[4] * R[7] + Tm[5] * R[8] - ms * c[1] * c[2];
    st.I(i, 6) = ms * c[0]; st.I(i, 7) = ms * c[1]; st.I(i, 8) = ms * c[2];
    st.I(i, 9) = ms;
  }
}

// ---- phase A, wave 3: frames of all links -> motion axes, velocity / bias-acceleration chain
template <class T, int STRIDE>
RCSH_HD void phaseA_motion(const DevModel& m, const Stage4<T, STRIDE>& st, bool stepping) {
  constexpr int NL = T::NL;
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {0, 0, 0};
  double vel[6] = {0, 0, 0, 0, 0, 0};
  double acc[6] = {0, 0, 0, -m.gravity[0], -m.gravity[1], -m.gravity[2]};
  double Rt[9], pt[3], velt[6], acct[6];
  ChainConsts<NL> cc;
  load_chain_consts<NL>(m, st, NL, cc);
  double qv[NL], qq[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) { qv[i] = st.v(i); qq[i] = st.q(i); }
  sched_fence();
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
    if (stepping) st.qpre(i) = qq[i];
    double ax[3], anchor[3], Si[6];
    chain_step<NL>(m, cc, i, R, p, ax, anchor);
    if (cc.jtype[i] == kSlide) {
      Si[0] = 0; Si[1] = 0; Si[2] = 0; Si[3] = ax[0]; Si[4] = ax[1]; Si[5] = ax[2];
    } else {
      Si[0] = ax[0]; Si[1] = ax[1]; Si[2] = ax[2];
      cross3(anchor, ax, Si + 3);
    }
    double sd[6];
    cross_motion(vel, Si, sd);
    const double qdi = qv[i];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      st.S(i, k) = Si[k];
      vel[k] += Si[k] * qdi;
      acc[k] += sd[k] * qdi;
      st.vel(i, k) = vel[k];
      st.acc(i, k) = acc[k];
    }
  }
}

// ---- phase B: bias wrench of the wave's links
template <class T, int STRIDE, int W>
RCSH_HD void phaseB_wrench(const Stage4<T, STRIDE>& st) {
#pragma unroll
  for (int i = 0; i < T::NL; ++i) {
    if (wrench_owner<T>(i) != W) continue;
    double Ii[10], vel[6], acc[6], Ia[6], Iv[6], vf[6];
#pragma unroll
    for (int k = 0; k < 10; ++k) Ii[k] = st.I(i, k);
#pragma unroll
    for (int k = 0; k < 6; ++k) { vel[k] = st.vel(i, k); acc[k] = st.acc(i, k); }
    inert_mul(Ii, acc, Ia);
    inert_mul(Ii, vel, Iv);
    cross_force(vel, Iv, vf);
#pragma unroll
    for (int k = 0; k < 6; ++k) st.f(i, k) = Ia[k] + vf[k];
  }
}

// ---- phase C: the wave's rows of M, bias and gravity compensation.  Subtree sums are rebuilt from the column
// (links are visited leaves first; a finger is a leaf and projects its own values only).
template <class T, int STRIDE, int W>
RCSH_HD void phaseC_rows(const DevModel& m, const Stage4<T, STRIDE>& st) {
  constexpr int NL = T::NL;
  const double ng[3] = {-m.gravity[0], -m.gravity[1], -m.gravity[2]};
  double Ic[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, fs[6] = {0, 0, 0, 0, 0, 0}, hs[3] = {0, 0, 0};
#pragma unroll
  for (int i = NL - 1; i >= 0; --i) {
    // rows above the wave's last (smallest-index) row are not needed
    bool earlier_mine = false;
#pragma unroll
    for (int j = 0; j <= i; ++j) earlier_mine = earlier_mine || row_owner<T>(j) == W;
    if (!earlier_mine) break;
    double Il[10], fl[6], hl[3];
#pragma unroll
    for (int k = 0; k < 10; ++k) { Il[k] = st.I(i, k); Ic[k] += Il[k]; }
#pragma unroll
    for (int k = 0; k < 6; ++k) { fl[k] = st.f(i, k); fs[k] += fl[k]; }
#pragma unroll
    for (int k = 0; k < 3; ++k) { hl[k] = st.hg(i, k); hs[k] += hl[k]; }
    if (row_owner<T>(i) != W) continue;
    const bool leaf = T::GRIP && i >= T::NARM;
    const double* Iu = leaf ? Il : Ic;
    const double* fu = leaf ? fl : fs;
    const double* hu = leaf ? hl : hs;
    double F[6], Sl[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) Sl[k] = st.S(i, k);
    inert_mul(Iu, Sl, F);
    st.M(tri(i, i)) = dot6(Sl, F) + m.armature[i];
#pragma unroll
    for (int j = (i >= T::NARM ? T::NARM - 1 : i - 1); j >= 0; --j) {
      double Sj[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) Sj[k] = st.S(j, k);
      st.M(tri(i, j)) = dot6(Sj, F);
    }
    if (T::GRIP && i == T::NARM + 1) st.M(tri(i, i - 1)) = 0.0;
    st.bias(i) = dot6(Sl, fu);
    double w[6];
    cross3(hu, ng, w);
    w[3] = m.gcm_sub[i] * ng[0]; w[4] = m.gcm_sub[i] * ng[1]; w[5] = m.gcm_sub[i] * ng[2];
    st.gc(i) = dot6(Sl, w);
  }
}

// gripper actuator force and whether forcerange saturates it (needed by actuation and by the implicit matrix)
template <class T, int STRIDE>
RCSH_HD double gripper_force(const DevModel& m, const Stage4<T, STRIDE>& st, bool* clamped) {
  constexpr int NA = T::NARM;
  double c = st.c(NA);
  if (m.grp_ctrllimited) c = clampd(c, m.grp_ctrlrange[0], m.grp_ctrlrange[1]);
  const double len = m.grp_coef[0] * st.q(NA) + m.grp_coef[1] * st.q(NA + 1);
  const double vel = m.grp_coef[0] * st.v(NA) + m.grp_coef[1] * st.v(NA + 1);
  double force = m.grp_gain * c;
  if (m.grp_biasaffine) force += m.grp_bias[0] + m.grp_bias[1] * len + m.grp_bias[2] * vel;
  *clamped = false;
  if (m.grp_forcelimited) {
    *clamped = force <= m.grp_forcerange[0] || force >= m.grp_forcerange[1];
    force = clampd(force, m.grp_forcerange[0], m.grp_forcerange[1]);
  }
  return force;
}
template <class T, int STRIDE>
RCSH_HD double arm_force(const DevModel& m, const Stage4<T, STRIDE>& st, int i, bool* clamped) {
  double c = st.c(i);
  if (m.arm_ctrllimited[i]) c = clampd(c, m.arm_ctrlrange[i][0], m.arm_ctrlrange[i][1]);
  const double gear = m.arm_gear[i];
  double force = m.arm_gain[i] * c;
  if (m.arm_biasaffine[i]) force += m.arm_bias[i][0] + m.arm_bias[i][1] * (gear * st.q(i)) + m.arm_bias[i][2] * (gear * st.v(i));
  *clamped = false;
  if (m.arm_forcelimited[i]) {
    *clamped = force <= m.arm_forcerange[i][0] || force >= m.arm_forcerange[i][1];
    force = clampd(force, m.arm_forcerange[i][0], m.arm_forcerange[i][1]);
  }
  return force;
}

// ---- phase D, wave 3: actuation -> qfrc_smooth
template <class T, int STRIDE>
RCSH_HD void phaseD_actuation(const DevModel& m, const Stage4<T, STRIDE>& st) {
  constexpr int NL = T::NL;
  constexpr int NA = T::NARM;
  double tau[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) tau[i] = 0;
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    if (!m.arm_has_act[i]) continue;
    bool cl;
    tau[i] = m.arm_gear[i] * arm_force<T, STRIDE>(m, st, i, &cl);
  }
  if (T::GRIP && m.grp_has_act) {
    bool cl;
    const double force = gripper_force<T, STRIDE>(m, st, &cl);
    tau[NA] += m.grp_coef[0] * force;
    tau[NA + 1] += m.grp_coef[1] * force;
  }
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    double passive = -m.damping[i] * st.v(i);
    const double gc = st.gc(i);
    if (m.actgravcomp[i]) tau[i] += gc; else passive += gc;
    if (m.actfrclimited[i]) tau[i] = clampd(tau[i], m.actfrcrange[i][0], m.actfrcrange[i][1]);
    st.smooth(i) = passive - st.bias(i) + tau[i];
  }
}

// ---- phase D, wave 0: LDL^T of the implicit matrix M - h dF/dqd (kept in registers across two barriers)
template <class T, int STRIDE>
RCSH_HD void phaseD_implicit_factor(const DevModel& m, const Stage4<T, STRIDE>& st, double* A) {
  constexpr int NL = T::NL;
  constexpr int NA = T::NARM;
  const double h = m.timestep;
#pragma unroll
  for (int k = 0; k < T::NTRI; ++k) A[k] = st.M(k);
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    double d = m.damping[i];
    if (i < NA && m.arm_has_act[i] && m.arm_biasaffine[i]) {
      bool cl;
      (void)arm_force<T, STRIDE>(m, st, i, &cl);
      if (!cl) d -= m.arm_gear[i] * m.arm_gear[i] * m.arm_bias[i][2];
    }
    A[tri(i, i)] += h * d;
  }
  if (T::GRIP && m.grp_has_act && m.grp_biasaffine) {
    bool cl;
    (void)gripper_force<T, STRIDE>(m, st, &cl);
    const double gblock = cl ? 0.0 : -m.grp_bias[2];
    A[tri(NA, NA)] += h * gblock * m.grp_coef[0] * m.grp_coef[0];
    A[tri(NA + 1, NA)] += h * gblock * m.grp_coef[0] * m.grp_coef[1];
    A[tri(NA + 1, NA + 1)] += h * gblock * m.grp_coef[1] * m.grp_coef[1];
  }
  ldl_factor<NL>(A);
}

// ---- phase D, wave 1: constraint rows + Newton solve (same algorithm as dyn.h) -> qfrc_constraint
template <class T, int STRIDE>
struct Rows {
  double eqD, eqAref, eqJ1;
  double D[T::NL], aref[T::NL], sgn[T::NL];
  uint32_t limrows;
  bool has_eq;
};
template <class T, int STRIDE>
RCSH_HD void phaseD_rows(const DevModel& m, const Stage4<T, STRIDE>& st, Rows<T, STRIDE>& r) {
  constexpr int NL = T::NL;
  constexpr int NA = T::NARM;
  r.eqD = 0; r.eqAref = 0; r.eqJ1 = 0; r.limrows = 0;
  r.has_eq = T::GRIP && m.eq_active;
  if (r.has_eq) {
    const double* pc = m.eq_polycoef;
    const double dif = st.q(NA + 1) - m.qpos0[NA + 1];
    const double poly = pc[0] + dif * (pc[1] + dif * (pc[2] + dif * (pc[3] + dif * pc[4])));
    const double deriv = pc[1] + dif * (2 * pc[2] + dif * (3 * pc[3] + dif * 4 * pc[4]));
    const double pos = st.q(NA) - m.qpos0[NA] - poly;
    r.eqJ1 = -deriv;
    const double imp = impedance(m.eq_imp, pos, 0.0);
    r.eqD = row_D(imp, m.invweight0[NA] + m.invweight0[NA + 1]);
    r.eqAref = -m.eq_K * imp * pos - m.eq_B * (st.v(NA) + r.eqJ1 * st.v(NA + 1));
  }
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    r.D[i] = 0; r.aref[i] = 0; r.sgn[i] = 0;
    if (!m.limited[i]) continue;
    const double qi = st.q(i);
    const double dlo = qi - m.range[i][0], dhi = m.range[i][1] - qi;
    double dist = 0, sgn = 0;
    if (dlo < m.margin[i]) { dist = dlo; sgn = 1; }
    else if (dhi < m.margin[i]) { dist = dhi; sgn = -1; }
    if (sgn != 0) {
      const double imp = impedance(m.lim_imp[i], dist, m.margin[i]);
      r.D[i] = row_D(imp, m.invweight0[i]);
      r.aref[i] = -m.lim_K[i] * imp * (dist - m.margin[i]) - m.lim_B[i] * (sgn * st.v(i));
      r.sgn[i] = sgn;
      r.limrows |= 1u << i;
    }
  }
}

// H = M + rows under active set `act`, factorised in place
template <class T, int STRIDE>
RCSH_HD void build_factor_H(const Stage4<T, STRIDE>& st, const Rows<T, STRIDE>& r, uint32_t act, double* H) {
  constexpr int NA = T::NARM;
#pragma unroll
  for (int k = 0; k < T::NTRI; ++k) H[k] = st.M(k);
#pragma unroll
  for (int i = 0; i < T::NL; ++i)
    if (act & (1u << i)) H[tri(i, i)] += r.D[i];
  if (r.has_eq) {
    H[tri(NA, NA)] += r.eqD;
    H[tri(NA + 1, NA)] += r.eqD * r.eqJ1;
    H[tri(NA + 1, NA + 1)] += r.eqD * r.eqJ1 * r.eqJ1;
  }
  ldl_factor<T::NL>(H);
}
template <class T, int STRIDE>
RCSH_HD void solve_H(const Stage4<T, STRIDE>& st, const Rows<T, STRIDE>& r, uint32_t act, const double* H, double* x) {
  constexpr int NA = T::NARM;
#pragma unroll
  for (int i = 0; i < T::NL; ++i) {
    x[i] = st.smooth(i);
    if (act & (1u << i)) x[i] += r.sgn[i] * r.D[i] * r.aref[i];
  }
  if (r.has_eq) {
    x[NA] += r.eqD * r.eqAref;
    x[NA + 1] += r.eqD * r.eqAref * r.eqJ1;
  }
  ldl_solve<T::NL>(H, x);
}

// after qfrc_smooth is available: finish the constraint solve (H for the first guess is already factorised)
template <class T, int STRIDE>
RCSH_HD void phaseD_constraint_solve(const Stage4<T, STRIDE>& st, const Rows<T, STRIDE>& r, double* H) {
  constexpr int NL = T::NL;
  constexpr int NA = T::NARM;
  if (!r.has_eq && !r.limrows) {
#pragma unroll
    for (int i = 0; i < NL; ++i) st.fc(i) = 0;
    return;
  }
  double x[NL];
  uint32_t act = r.limrows;
  bool have_x = false;
  for (int iter = 0; iter < 16; ++iter) {
    double xn[NL];
    if (iter > 0) build_factor_H<T, STRIDE>(st, r, act, H);
    solve_H<T, STRIDE>(st, r, act, H, xn);
    uint32_t now = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i)
      if ((r.limrows & (1u << i)) && r.sgn[i] * xn[i] - r.aref[i] < 0) now |= 1u << i;
    if (now == act || !have_x) {
#pragma unroll
      for (int i = 0; i < NL; ++i) x[i] = xn[i];
      have_x = true;
      if (now == act) break;
      act = now;
      continue;
    }
    // exact line search from x along d = xn - x (see dyn.h)
    double d[NL], jar[NL], jd[NL];
    double p0 = 0, p1 = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) d[i] = xn[i] - x[i];
#pragma unroll
    for (int rr = 0; rr < NL; ++rr) {
      double mx = -st.smooth(rr), md = 0;
#pragma unroll
      for (int c = 0; c < NL; ++c) {
        const double mrc = st.M(rr >= c ? tri(rr, c) : tri(c, rr));
        mx += mrc * x[c];
        md += mrc * d[c];
      }
      p0 += mx * d[rr];
      p1 += md * d[rr];
    }
    if (r.has_eq) {
      const double je = x[NA] + r.eqJ1 * x[NA + 1] - r.eqAref, jde = d[NA] + r.eqJ1 * d[NA + 1];
      p0 += r.eqD * je * jde;
      p1 += r.eqD * jde * jde;
    }
    uint32_t on = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      jar[i] = 0; jd[i] = 0;
      if (r.limrows & (1u << i)) {
        jar[i] = r.sgn[i] * x[i] - r.aref[i];
        jd[i] = r.sgn[i] * d[i];
        if (jar[i] < 0 || (jar[i] == 0 && jd[i] < 0)) on |= 1u << i;
      }
    }
    double alpha = 0;
    for (int guard = 0; guard < NL + 2; ++guard) {
      double c0 = p0, c1 = p1, a_next = INFINITY;
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        if (!(r.limrows & (1u << i))) continue;
        if (on & (1u << i)) { c0 += r.D[i] * jar[i] * jd[i]; c1 += r.D[i] * jd[i] * jd[i]; }
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
        if ((r.limrows & (1u << i)) && jd[i] != 0 && -jar[i] / jd[i] == a_next) on ^= 1u << i;
    }
    act = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      x[i] += alpha * d[i];
      if ((r.limrows & (1u << i)) && r.sgn[i] * x[i] - r.aref[i] < 0) act |= 1u << i;
    }
  }
  double fc[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    fc[i] = 0;
    if (r.limrows & (1u << i)) {
      const double rr = r.sgn[i] * x[i] - r.aref[i];
      if (rr < 0) fc[i] = -r.sgn[i] * r.D[i] * rr;
    }
  }
  if (r.has_eq) {
    const double fe = -r.eqD * (x[NA] + r.eqJ1 * x[NA + 1] - r.eqAref);
    fc[NA] += fe;
    fc[NA + 1] += fe * r.eqJ1;
  }
#pragma unroll
  for (int i = 0; i < NL; ++i) st.fc(i) = fc[i];
}

// ---- last step, wave 0: implicit solve + semi-implicit Euler (only while the environment is still stepping)
template <class T, int STRIDE>
RCSH_HD void phaseD_integrate(const DevModel& m, const Stage4<T, STRIDE>& st, const double* A) {
  constexpr int NL = T::NL;
  const double h = m.timestep;
  double rhs[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) rhs[i] = st.smooth(i) + st.fc(i);
  ldl_solve<NL>(A, rhs);
  if (st.active() != 0.0) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const double v = st.v(i) + h * rhs[i];
      st.v(i) = v;
      st.q(i) += h * v;
    }
  }
}

}  // namespace rcsh
