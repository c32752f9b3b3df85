// contact_wide.h -- the coupled solve of a scene WITHOUT a free box, for ANY number of contacts the arena holds: a lane per contact,
// its rows in the lane's registers.
//
// The contact-resolving kernel of per-environment escalation (sim_kernels.h: RunOp::esc_role 2) puts all 64 lanes of its wavefront on
// ONE environment, and without a free box the problem has only the robot's NL dofs.  Three rows of NL entries are 27 doubles: lane c
// keeps J_c -- the rows of contact c -- in registers, and so its products with a vector over the dofs (the rows' residuals, the
// search direction's effect) are 27 multiply-adds without a single LDS access; the transposed products (the gradient J' f: NL sums
// over the contacts; the Hessian's J' H J: NL (NL + 1) / 2 of them) are per-lane partial sums reduced across the wavefront -- a DPP
// butterfly each, independent of the number of contacts.  No pass over the kinematic tree, no stiffness accumulators, no bound on
// how many links or pairs of links are in contact (contact_team.h's tree formulation keeps 5 links / 4 pairs and costs ~90k cycles a
// Newton iteration and ~4.5k a noslip update whatever the count; contact_dense.h's rows-in-LDS stop at 21 contacts).
//   * Newton on the primal cost: the same iteration, line search and stopping rules as contact_newton / contact_newton_dense (oracle
//     orc_solve_coupled, rcs_contact.c) -- results agree to round-off;
//   * noslip: Gauss-Seidel in contact order, as mj_solNoSlip.  Every lane carries the CHANGE of qacc the sweep has made so far (NL
//     doubles, the same in all lanes); the owner of contact c forms its rows' residuals from it (27 multiply-adds), solves its
//     2 x 2 problem, and hands the change's increment M^-1 J_c' df (its Y_c rows, also in registers) to the wavefront through 2 NL
//     v_readlane -- no LDS, no barrier, nothing that grows with the number of contacts but the sweep itself.
#pragma once

namespace rcsh {

#if defined(__HIP__)

template <class T, bool FRIC, class AR>
RCSH_CONTACT_FN void contact_newton_wide(const BoxCfg& b_, const StageTeam<T>& st_, double* bs_, AR& ar_, const double* gravity_, const LinkRec* links_) {
  const LinkRec* links = in_lds(links_);
  (void)links;
  const BoxCfg& b = *in_lds(&b_);
  const StageTeam<T> st{in_lds(st_.base)};
  double* bs = in_lds(bs_);
  AR& ar = *in_lds(&ar_);
  const double* gravity = in_lds(gravity_);
  constexpr int NL = T::NL, NA = T::NARM, NV = NL + 6, NTRI = NL * (NL + 1) / 2;
  constexpr int kWorld = NL + 1;
  static_assert(AR::kCap <= 64, "a lane per contact");
  static_assert(NTRI <= 64, "a lane per Hessian entry");
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
  // ---- rows of the lane's contact: regulariser, reference accelerations -> record (as contact_newton)
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
  // ---- qacc_smooth: the robot's by its own factorisation (every lane); the phantom box keeps its own (closed form, as contact_newton)
  const double Mb[6] = {b.mass, b.mass, b.mass, b.inertia[0], b.inertia[1], b.inertia[2]};
  {
    double LM[T::NTRI], a0[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) LM[tri(i, j)] = st.M(i, j);
    ldl_factor<NL>(LM);
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
  // ---- the lane's contact and its three rows
  ConLane c;
  con_load(ar, b, lane, ncon, kWorld, c);
  const bool on = c.on;
  double J[3][NL];
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const int s = ((c.B < NL && is_anc<T>(j, c.B)) ? 1 : 0) - ((c.A < NL && is_anc<T>(j, c.A)) ? 1 : 0);
    double sj[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) sj[k] = st.S(j, k);
#pragma unroll
    for (int k = 0; k < 3; ++k) J[k][j] = on && s != 0 ? s * dot6(c.G[k], sj) : 0.0;
  }
  const bool has_eq = T::GRIP && st.eq(0) != 0.0;
  const double eqD = T::GRIP ? st.eq(0) : 0.0, eqAref = T::GRIP ? st.eq(1) : 0.0, eqJ1 = T::GRIP ? st.eq(2) : 0.0;

  // the lane's contact at x (an LDS vector over the dofs): jar, force, cone Hessian; returns the contact's cost
  double jar[3] = {0, 0, 0}, Hc[6] = {0, 0, 0, 0, 0, 0}, f[3] = {0, 0, 0};
  auto eval_rows = [&](const double* x, double* jar_out, double* f_out, double* Hc_out) -> double {
    double xr[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) xr[j] = x[j];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      double jr = -c.aref[k];
#pragma unroll
      for (int j = 0; j < NL; ++j) jr += J[k][j] * xr[j];
      jar_out[k] = jr;
    }
    return on ? cone_eval(c.D, c.mu, c.fr, jar_out, f_out, Hc_out) : 0.0;
  };
  // robot rows (limit rows, the finger coupling, dry friction) and the Gauss term at x: as contact_newton_dense
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
    }
    *grad_out = g;
    return cost;
  };

  TEAM_MARK(27)
  // ---- start: the cheapest of qacc_smooth, the warm start and the previous coupled solve's minimiser (contact_newton's three)
  {
    if (lane < NL) ar.P[lane] = st.xs(lane);
    else if (lane < NV) ar.P[lane] = bs[kBoxW + lane - NL];
    if (lane < NV) { ar.X[lane] = ar.A0[lane]; ar.Gd[lane] = bs[kBoxX + lane]; }
    __syncthreads();
    double g, ja[3], fa[3], Ha[6];
    const double c_smooth = wave_sum(eval_rows(ar.X, ja, fa, Ha) + robot_terms(ar.X, &g));
    const double c_warm = wave_sum(eval_rows(ar.P, ja, fa, Ha) + robot_terms(ar.P, &g));
    const double c_prev = wave_sum(eval_rows(ar.Gd, ja, fa, Ha) + robot_terms(ar.Gd, &g));
    __syncthreads();
    if (lane < NL) {
      if (c_warm < c_smooth) ar.X[lane] = ar.P[lane];
      if (c_prev < fmin(c_warm, c_smooth)) ar.X[lane] = ar.Gd[lane];
    }
    __syncthreads();
  }
  // the Hessian entry of this lane (lower triangle, lane < NTRI)
  int ha = 0, hb = 0;
  {
    while ((ha + 1) * (ha + 2) / 2 <= lane) ++ha;
    hb = lane - ha * (ha + 1) / 2;
    if (lane >= NTRI) { ha = 0; hb = 0; }
  }
  bool at_x = false;
  int newton_done = 100;
  for (int newton_it = 0; newton_it < 100; ++newton_it) {
    TEAM_MARK(55)
    TEAM_COUNT(29)
    eval_rows(ar.X, jar, f, Hc);
    at_x = true;
    const bool curved = on && (Hc[0] != 0.0 || Hc[2] != 0.0 || Hc[5] != 0.0);  // (the cone's top zone has no force and no curvature)
    const uint64_t hmask = __ballot(curved);
    // ---- gradient: the robot's rows on their lanes, less J' f -- every lane's share of it, summed across the wavefront dof by dof
    double gl;
    robot_terms(ar.X, &gl);
    double qf = 0.0;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const double tot = wave_sum(J[0][j] * f[0] + J[1][j] * f[1] + J[2][j] * f[2]);
      qf = lane == j ? tot : qf;
    }
    gl -= qf;
    if (lane < NL) ar.Gd[lane] = gl;
    const double g2 = wave_sum(lane < NL ? gl * gl : 0.0);
    const double q2 = wave_sum(lane < NL ? qf * qf : 0.0);
    if (b.scale * sqrt(g2) < 1e-12 || g2 <= kNewtonRel * kNewtonRel * q2) { newton_done = newton_it; break; }
    TEAM_MARK(48)
    // ---- Hessian H = M + the robot rows' curvature + sum over the curved contacts of J_c' Hc J_c
    {
      double v = 0.0;
      if (lane < NTRI) {
        v = st.M(ha, hb);
        if (ha == hb) {
          const double sgn = st.limS(ha);
          if (sgn != 0.0 && sgn * ar.X[ha] - st.limA(ha) < 0) v += st.limD(ha);
          if constexpr (FRIC) {
            const double fF = links[ha].fl_floss, fR = links[ha].fl_R, jf = ar.X[ha] - st.fa(ha);
            if (fF > 0 && jf > -fR && jf < fR) v += links[ha].fl_D;
          }
        }
        if (T::GRIP && has_eq) {
          if (ha == NA && hb == NA) v += eqD;
          if (ha == NA + 1 && hb == NA) v += eqD * eqJ1;
          if (ha == NA + 1 && hb == NA + 1) v += eqD * eqJ1 * eqJ1;
        }
      }
      if (hmask) {
        // T = Hc J_c (3 x NL) on the lane; entry (a, b) of its J_c' T, summed over the lanes, to the entry's lane
        double Tm[3][NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) {
          Tm[0][j] = curved ? Hc[0] * J[0][j] + Hc[1] * J[1][j] + Hc[3] * J[2][j] : 0.0;
          Tm[1][j] = curved ? Hc[1] * J[0][j] + Hc[2] * J[1][j] + Hc[4] * J[2][j] : 0.0;
          Tm[2][j] = curved ? Hc[3] * J[0][j] + Hc[4] * J[1][j] + Hc[5] * J[2][j] : 0.0;
        }
#pragma unroll
        for (int a = 0; a < NL; ++a)
#pragma unroll
          for (int bb = 0; bb <= a; ++bb) {
            const double tot = wave_sum(J[0][a] * Tm[0][bb] + J[1][a] * Tm[1][bb] + J[2][a] * Tm[2][bb]);
            v += lane == tri(a, bb) ? tot : 0.0;
          }
      }
      if (lane < NTRI) ar.H[lane] = v;
    }
    __syncthreads();
    TEAM_MARK(50)
    // ---- Newton direction p = -H^-1 grad: the cooperative LDL' of contact_newton_dense (lane i < NL holds row i)
    double dphi0 = 0;
    {
      const int row = lane < NL ? lane : NL - 1;
      double hr[NL];
#pragma unroll
      for (int k = 0; k < NL; ++k) hr[k] = ar.H[row >= k ? tri(row, k) : tri(k, row)];
      TEAM_MARK(51)
#pragma unroll
      for (int j = 0; j < NL - 1; ++j) {
        const double dj = wave_read(hr[j], j);
        const double lij = lane > j ? hr[j] / dj : 0.0;
#pragma unroll
        for (int k = j + 1; k < NL; ++k) hr[k] -= lij * wave_read(hr[j], k);
      }
      double dg = 0;
#pragma unroll
      for (int k = 0; k < NL; ++k) dg = row == k ? hr[k] : dg;
      const double dinv = 1.0 / dg, gl_ = lane < NL ? ar.Gd[lane] : 0.0;
      double acc = -gl_;
#pragma unroll
      for (int k = 0; k < NL - 1; ++k) {
        const double yk = wave_read(acc, k) * wave_read(dinv, k);
        if (lane > k) acc -= hr[k] * yk;
      }
      acc *= dinv;
#pragma unroll
      for (int k = NL - 1; k >= 1; --k) {
        const double xk = wave_read(acc, k);
        if (lane < k) acc -= hr[k] * dinv * xk;
      }
      dphi0 = wave_sum(lane < NL ? gl_ * acc : 0.0);
      if (lane < NL) ar.P[lane] = acc;
    }
    TEAM_MARK(52)
    if (!(dphi0 < 0)) { newton_done = 1000 + newton_it; break; }
    __syncthreads();
    // ---- line search: root of phi'(a) by safeguarded 1-D Newton (contact_newton's)
    double jd[3];
    {
      double pr[NL];
#pragma unroll
      for (int j = 0; j < NL; ++j) pr[j] = ar.P[j];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        double jr = 0.0;
#pragma unroll
        for (int j = 0; j < NL; ++j) jr += J[k][j] * pr[j];
        jd[k] = jr;
      }
    }
    double gM0l = 0, pMpl = 0;
    if (lane < NL) {
      double mx = 0, mp = 0;
      for (int j = 0; j < NL; ++j) {
        const double mij = lane >= j ? st.M(lane, j) : st.M(j, lane);
        mx += mij * (ar.X[j] - ar.A0[j]);
        mp += mij * ar.P[j];
      }
      gM0l = mx * ar.P[lane]; pMpl = mp * ar.P[lane];
    }
    const double gM0 = wave_sum(gM0l), pMp = wave_sum(pMpl);
    TEAM_MARK(53)
    double lo = 0, hi = -1, a = 1, best = 1, dx = 1e300, dxold = 1e300;
    for (int ls = 0; ls < 30; ++ls) {
      TEAM_COUNT(35)
      double dl = 0, ddl = 0;
      if (on) {
        double ja[3], fa[3], Ha[6];
#pragma unroll
        for (int k = 0; k < 3; ++k) ja[k] = jar[k] + a * jd[k];
        cone_eval(c.D, c.mu, c.fr, ja, fa, Ha);
        dl = -(jd[0] * fa[0] + jd[1] * fa[1] + jd[2] * fa[2]);
        ddl = jd[0] * (Ha[0] * jd[0] + Ha[1] * jd[1] + Ha[3] * jd[2]) + jd[1] * (Ha[1] * jd[0] + Ha[2] * jd[1] + Ha[4] * jd[2]) +
              jd[2] * (Ha[3] * jd[0] + Ha[4] * jd[1] + Ha[5] * jd[2]);
      }
      // (the robot's rows sit on lanes < NL, which own contacts too: their terms are ADDED to the lane's contact terms)
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
    if (lane < NL) {
      const double xo = ar.X[lane], xn = xo + best * ar.P[lane];
      moved = xn != xo;
      ar.X[lane] = xn;
    }
    at_x = false;
    __syncthreads();
    if (!__ballot(moved)) { newton_done = newton_it + 1; break; }
  }
#ifdef RCSH_PHASE_TIMING
  if (lane == 0) {
    atomicMax(&g_team_cycles[56], (unsigned long long)(newton_done % 1000));
    if (newton_done % 1000 > 20) atomicAdd(&g_team_cycles[57], 1ull);
    if (newton_done == 100) atomicAdd(&g_team_cycles[58], 1ull);
    if (newton_done >= 1000) atomicAdd(&g_team_cycles[59], 1ull);
    atomicAdd(&g_team_cycles[60], 1ull);
    atomicAdd(&g_team_cycles[67], (unsigned long long)(newton_done % 1000));
  }
#endif
  (void)newton_done;
  if (!at_x) eval_rows(ar.X, jar, f, Hc);
  if (lane < NV) bs[kBoxX + lane] = ar.X[lane];
  if (on) {
    double* r = ar.rec[lane];
    r[11] = f[0]; r[12] = f[1]; r[13] = f[2];
  }
  __syncthreads();
  TEAM_MARK(28)
}

// noslip + results.  In: ar.X (the Newton minimiser), M's factor in ar.V, the records (forces included).
template <class T, class AR>
RCSH_CONTACT_FN void contact_noslip_wide(const BoxCfg& b_, const StageTeam<T>& st_, double* bs_, AR& ar_) {
  const BoxCfg& b = *in_lds(&b_);
  const StageTeam<T> st{in_lds(st_.base)};
  double* bs = in_lds(bs_);
  AR& ar = *in_lds(&ar_);
  constexpr int NL = T::NL, NA = T::NARM, NV = NL + 6;
  constexpr int kWorld = NL + 1;
  const int lane = wave_lane();
  const int ncon = ar.ncon;
  ConLane c;
  con_load(ar, b, lane, ncon, kWorld, c);
  const bool on = c.on;
  const bool has_eq = T::GRIP && st.eq(0) != 0.0;
  const double eqD = T::GRIP ? st.eq(0) : 0.0, eqAref = T::GRIP ? st.eq(1) : 0.0, eqJ1 = T::GRIP ? st.eq(2) : 0.0;
  double J[3][NL];
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const int s = ((c.B < NL && is_anc<T>(j, c.B)) ? 1 : 0) - ((c.A < NL && is_anc<T>(j, c.A)) ? 1 : 0);
    double sj[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) sj[k] = st.S(j, k);
#pragma unroll
    for (int k = 0; k < 3; ++k) J[k][j] = on && s != 0 ? s * dot6(c.G[k], sj) : 0.0;
  }
  if (b.noslip_iterations > 0) {
    // Y_k = M^-1 J_k' (the factor contact_newton_wide left in ar.V), the contact's own 3 x 3 block of A = J M^-1 J' (no regulariser)
    double Y[3][NL], Ac[3][3];
    {
      double LM[T::NTRI];
#pragma unroll
      for (int e = 0; e < T::NTRI; ++e) LM[e] = (&ar.V[0][0])[e];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        double col[NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) col[j] = J[k][j];
        ldl_solve<NL>(LM, col);
#pragma unroll
        for (int j = 0; j < NL; ++j) Y[k][j] = col[j];
      }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < NL; ++j) s += J[r][j] * Y[k][j];
        Ac[r][k] = s;
      }
    const double qS11 = Ac[1][1] * c.fr * c.fr, qS22 = Ac[2][2] * c.fr * c.fr, qS12 = Ac[1][2] * c.fr * c.fr;
    const double qdet = qS11 * qS22 - qS12 * qS12, qdi = 1 / qdet;
    // the rows' residuals at the Newton solution
    double res0[3];
    {
      double xr[NL];
#pragma unroll
      for (int j = 0; j < NL; ++j) xr[j] = ar.X[j];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        double jr = -c.aref[k];
#pragma unroll
        for (int j = 0; j < NL; ++j) jr += J[k][j] * xr[j];
        res0[k] = jr;
      }
    }
    double du[NL];  // what the sweeps have changed qacc by so far: the same in every lane
#pragma unroll
    for (int j = 0; j < NL; ++j) du[j] = 0.0;
    TEAM_MARK(30)
    int iter = 0;
    while (iter < b.noslip_iterations) {
      TEAM_COUNT(36)
      double improvement = 0;
      if (iter == 0) {
        double s = on ? 0.5 * (c.f[0] * c.f[0] * c.Rr[0] + c.f[1] * c.f[1] * c.Rr[1] + c.f[2] * c.f[2] * c.Rr[2]) : 0.0;
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
        double change = 0, ddu[NL];
        int moved = 0;
#pragma unroll
        for (int j = 0; j < NL; ++j) ddu[j] = 0.0;
        if (lane == cc) {
          double rs[3];
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            double r = res0[k];
#pragma unroll
            for (int j = 0; j < NL; ++j) r += J[k][j] * du[j];
            rs[k] = r;
          }
          const double old[3] = {c.f[0], c.f[1], c.f[2]};
          double nf[3] = {old[0], old[1], old[2]};
          if (old[0] < kMinVal) {
            // (a contact the Newton solution left without normal force: all three rows go to zero -- the general update)
            nf[0] = nf[1] = nf[2] = 0;
            const double dl[3] = {-old[0], -old[1], -old[2]};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
#pragma unroll
              for (int l = 0; l < 3; ++l) change += 0.5 * dl[k] * Ac[k][l] * dl[l];
              change += dl[k] * rs[k];
            }
            if (change > 1e-10) { nf[0] = old[0]; nf[1] = old[1]; nf[2] = old[2]; change = 0; }
          } else {
            const double res1 = rs[1], res2 = rs[2];
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
          }
          moved = nf[0] != old[0] || nf[1] != old[1] || nf[2] != old[2];
          const double df[3] = {nf[0] - old[0], nf[1] - old[1], nf[2] - old[2]};
#pragma unroll
          for (int j = 0; j < NL; ++j) ddu[j] = Y[0][j] * df[0] + Y[1][j] * df[1] + Y[2][j] * df[2];
          c.f[0] = nf[0]; c.f[1] = nf[1]; c.f[2] = nf[2];
        }
        TEAM_MARK(41)
        if (wave_read(moved, cc)) {
#pragma unroll
          for (int j = 0; j < NL; ++j) du[j] += wave_read(ddu[j], cc);
        }
        TEAM_MARK(44)
        TEAM_COUNT(45)
        improvement -= wave_read(change, cc);
      }
      improvement *= b.scale;
      ++iter;
      if (improvement < b.noslip_tolerance) break;
    }
  }
  TEAM_MARK(31)
  // ---- results: qfrc_constraint of the robot (the phantom box's acceleration stays its own)
  {
    double qf = 0.0;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const double tot = wave_sum(J[0][j] * c.f[0] + J[1][j] * c.f[1] + J[2][j] * c.f[2]);
      qf = lane == j ? tot : qf;
    }
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
    }
    // (the phantom box's acceleration is nobody's to read: its block in LDS ends with its state -- sim_kernels.h: kBoxStride)
  }
  __syncthreads();
  TEAM_MARK(32)
}

#endif  // __HIP__

}  // namespace rcsh
