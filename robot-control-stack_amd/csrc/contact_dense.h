// contact_dense.h -- the coupled solve for FEW contacts (at most kDenseCon): the rows written out.
//
// contact_team.h never forms the contact Jacobian: a grasp can bring 48 contacts, 144 rows of 15 entries do not fit the arena, so
// J x, J' f and J' D J travel through the bodies' spatial quantities -- a pass over the kinematic tree per product, a 6 x 6 stiffness
// per body pair, ~36k cycles per Newton iteration and ~2.7k per noslip update whatever the number of contacts.  But the contacts
// the headline workload runs into are one or two (a finger on link 1, a link on the floor), and a pinch holds a dozen.  With at
// most kDenseCon contacts the rows DO fit -- 42 rows of NVD entries in the areas the tree formulation uses for its wrenches and
// stiffness accumulators -- and every product becomes a short loop over LDS:
//   * lane r = 3 c + k owns ROW k of contact c: J_r in its registers (and in LDS for the transposed products), jar_r = J_r x - aref_r
//     is NVD multiply-adds; the three lanes of a contact exchange their jar over the LDS crossbar and evaluate the cone redundantly;
//   * the gradient entry of dof j is a loop over the rows' column j; the Hessian entry (a, b) a loop over the contacts whose cone
//     Hessian is not zero, one or two entries per lane;
//   * noslip: Y_r = M^-1 J_r' once per row (the factor of M is there), then every row lane keeps the residual of ITS row current --
//     an update of contact c changes it by (J_r Y_3c+k) df_k, three dot products that do not depend on the update's result -- so a
//     Gauss-Seidel update is the owner's 2 x 2 problem plus three multiply-adds, with no barrier and no tree pass.
// Same problem, same algorithm (Newton to the minimiser of the primal cost with the same line search; mj_solNoSlip sweep by sweep
// in contact order), the same results as the tree formulation to round-off: oracle orc_solve_coupled (rcs_contact.c).
// BOXD: the scene has a free box (its six dofs are columns NL..NL+5); without one the rows have the robot's NL columns only.
#pragma once

namespace rcsh {

#if defined(__HIP__)

// (with a free box the rows have 15 entries: J's 54 rows of 18 contacts fill the stage and the stiffness accumulators' area, Y the records'
// tail and -- during the noslip pass, when nobody needs it -- the Hessian's; without one 9 entries, and 21 contacts are 63 rows: a lane each)
constexpr int kDenseConBox = 18, kDenseConNoBox = 21;

template <class T, bool BOXD, class AR_>
struct DenseLds {
  static constexpr int NL = T::NL, NVD = BOXD ? T::NL + 6 : T::NL;
  static constexpr int kDenseCon = BOXD ? kDenseConBox : kDenseConNoBox;
  static constexpr int kRows = 3 * kDenseCon;
  static constexpr int kRowsInStage = (64 * 8) / NVD;
  using AR = AR_;
  // row r of J: the stage area first, the stiffness accumulators' area for the rest
  RCSH_D static double* jrow(AR& ar, int r) { return r < kRowsInStage ? &ar.stage[0][0] + r * NVD : &ar.KA[0][0] + (r - kRowsInStage) * NVD; }
  // Y_r = (M^-1 J_r')' of the FRICTION rows (k = 1, 2) of contact c: the records' area behind the kDenseCon records in use.  (The noslip
  // pass changes normal forces only where the Newton solution left none -- by less than kMinVal = 1e-15 N: that change's effect on the
  // other rows' residuals is dropped, and the normal rows' Y, needed for the contact's own 3 x 3 block only, stay in registers.)
  static constexpr int kYRowsBehindRecords = ((AR::kCap - kDenseCon) * 14) / NVD, kYRowsInHessian = (AR::NV * (AR::NV + 1) / 2) / NVD;
  RCSH_D static double* yrow(AR& ar, int c, int k) {
    const int i = 2 * c + k - 1;
    return i < kYRowsBehindRecords ? &ar.rec[kDenseCon][0] + i * NVD : &ar.H[0] + (i - kYRowsBehindRecords) * NVD;
  }
  // the rows' forces, the contacts' cone Hessians (00 10 11 20 21 22): the bodies' accelerations' area (U, Up, W)
  RCSH_D static double* frow(AR& ar) { return &ar.U[0][0]; }
  RCSH_D static double* hcone(AR& ar, int c) { return &ar.U[0][0] + kRows + 6 * c; }
  static_assert(kRows <= kRowsInStage || (kRows - kRowsInStage) * NVD <= AR::kAcc * 21, "J fits stage + KA");
  static_assert(2 * kDenseCon <= kYRowsBehindRecords + kYRowsInHessian, "Y fits behind the records and into the Hessian's area (free during the noslip pass)");
  static_assert(kRows + 6 * kDenseCon <= 3 * AR::NB * 6, "forces and cone Hessians fit U, Up, W");
  static_assert(kRows <= 64, "a lane per row");
};

// Rows + Newton.  In: the contact records of contact_collide (at most kDenseCon).  Out, as contact_newton: the minimiser in ar.X
// (and bs[kBoxX..]), qacc_smooth in ar.A0, M's factor in ar.V, reference accelerations / regularisers / forces in the records --
// and J in LDS (DenseLds::jrow) for the noslip pass.
template <class T, bool FRIC, bool BOXD, class AR>
RCSH_CONTACT_FN void contact_newton_dense(const BoxCfg& b_, const StageTeam<T>& st_, double* bs_, AR& ar_, const double* gravity_,
                                          const LinkRec* links_) {
  const LinkRec* links = in_lds(links_);
  (void)links;
  const BoxCfg& b = *in_lds(&b_);
  const StageTeam<T> st{in_lds(st_.base)};
  double* bs = in_lds(bs_);
  AR& ar = *in_lds(&ar_);
  const double* gravity = in_lds(gravity_);
  using DL = DenseLds<T, BOXD, AR>;
  constexpr int NL = T::NL, NA = T::NARM, NV = NL + 6, NVD = DL::NVD, NTRID = NVD * (NVD + 1) / 2;
  constexpr int kBox = NL, kWorld = NL + 1;
  const int lane = wave_lane();
  TEAM_COUNT(34)
  const int ncon = ar.ncon, nrow = 3 * ncon;
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
  // ---- qacc_smooth: the robot's by its own factorisation (every lane), the box's in closed form (as contact_newton)
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
  // ---- the lane's row: contact crow, row krow of its frame
  const int crow = lane / 3, krow = lane - 3 * crow;
  const bool row_on = lane < nrow, own = row_on && krow == 0;  // own: the lane that speaks for its contact in sums over contacts
  const int base = (3 * crow) & 63;                            // first lane of the contact's three
  ConLane c;
  con_load(ar, b, crow, ncon, kWorld, c);
  double Jr[NVD];
  {
    double g[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) g[k] = krow == 0 ? c.G[0][k] : (krow == 1 ? c.G[1][k] : c.G[2][k]);
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int s = ((c.B < NL && is_anc<T>(j, c.B)) ? 1 : 0) - ((c.A < NL && is_anc<T>(j, c.A)) ? 1 : 0);
      double v = 0.0;
      if (s != 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) v += g[k] * st.S(j, k);
      }
      Jr[j] = row_on ? s * v : 0.0;
    }
    if constexpr (BOXD) {
      const int sb = (c.B == kBox ? 1 : 0) - (c.A == kBox ? 1 : 0);
#pragma unroll
      for (int k2 = 0; k2 < 6; ++k2) {
        double col[6];
        box_column(bR, bp, k2, col);
        Jr[NL + k2] = row_on && sb != 0 ? sb * dot6(g, col) : 0.0;
      }
    }
    if (row_on) {
      double* dst = DL::jrow(ar, lane);
#pragma unroll
      for (int j = 0; j < NVD; ++j) dst[j] = Jr[j];
    }
  }
  const double aref_r = krow == 0 ? c.aref[0] : (krow == 1 ? c.aref[1] : c.aref[2]);
  const bool has_eq = T::GRIP && st.eq(0) != 0.0;
  const double eqD = T::GRIP ? st.eq(0) : 0.0, eqAref = T::GRIP ? st.eq(1) : 0.0, eqJ1 = T::GRIP ? st.eq(2) : 0.0;

  // the lane's contact at x (an LDS vector): jar of its three rows, force, cone Hessian; returns the contact's cost on its `own` lane
  double jar[3] = {0, 0, 0}, Hc[6] = {0, 0, 0, 0, 0, 0}, f[3] = {0, 0, 0};
  auto eval_rows = [&](const double* x, double* jar_out, double* f_out, double* Hc_out) -> double {
    double jr = -aref_r;
#pragma unroll
    for (int j = 0; j < NVD; ++j) jr += Jr[j] * x[j];
    jar_out[0] = lane_get(jr, base); jar_out[1] = lane_get(jr, base + 1); jar_out[2] = lane_get(jr, base + 2);
    double cost = 0;
    if (row_on) cost = cone_eval(c.D, c.mu, c.fr, jar_out, f_out, Hc_out);
    return own ? cost : 0.0;
  };
  // robot rows (limit rows, the finger coupling, dry friction) and the Gauss term at x: as contact_newton
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
    } else if (lane < NVD) {
      const int k = lane - NL;
      const double dx = x[lane] - ar.A0[lane];
      g = Mb[k] * dx;
      cost = 0.5 * Mb[k] * dx * dx;
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
    if (lane < NVD) {
      if (c_warm < c_smooth) ar.X[lane] = ar.P[lane];
      if (c_prev < fmin(c_warm, c_smooth)) ar.X[lane] = ar.Gd[lane];
    }
    __syncthreads();
  }
  // the Hessian entries of this lane: lower triangle, entries lane and lane + 64
  int ha[2], hb[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int idx = lane + 64 * u;
    int a = 0;
    while ((a + 1) * (a + 2) / 2 <= idx) ++a;
    ha[u] = a; hb[u] = idx - a * (a + 1) / 2;
  }
  double* const fl = DL::frow(ar);
  bool at_x = false;
  int newton_done = 100;
  for (int newton_it = 0; newton_it < 100; ++newton_it) {
    TEAM_MARK(55)
    TEAM_COUNT(29)
    eval_rows(ar.X, jar, f, Hc);
    at_x = true;
    if (row_on) fl[lane] = krow == 0 ? f[0] : (krow == 1 ? f[1] : f[2]);
    const bool curved = own && (Hc[0] != 0.0 || Hc[2] != 0.0 || Hc[5] != 0.0);  // (the cone's top zone has no force and no curvature)
    if (curved) {
      double* h = DL::hcone(ar, crow);
#pragma unroll
      for (int k = 0; k < 6; ++k) h[k] = Hc[k];
    }
    const uint64_t hmask = __ballot(curved);
    __syncthreads();
    double gl;
    robot_terms(ar.X, &gl);
    double qf = 0.0;
    if (lane < NVD) {
      for (int r = 0; r < nrow; ++r) qf += DL::jrow(ar, r)[lane] * fl[r];
      gl -= qf;
      ar.Gd[lane] = gl;
    }
    const double g2 = wave_sum(lane < NVD ? gl * gl : 0.0);
    const double q2 = wave_sum(lane < NVD ? qf * qf : 0.0);
    if (b.scale * sqrt(g2) < 1e-12 || g2 <= kNewtonRel * kNewtonRel * q2) { newton_done = newton_it; break; }
    TEAM_MARK(48)
    // ---- Hessian H = M + the robot rows' curvature + sum over the curved contacts of J_c' Hc J_c
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = lane + 64 * u;
      if (idx >= NTRID) continue;
      const int a = ha[u], bb = hb[u];
      double v = 0.0;
      if (a < NL) {
        v = st.M(a, bb);
        if (a == bb) {
          const double sgn = st.limS(a);
          if (sgn != 0.0 && sgn * ar.X[a] - st.limA(a) < 0) v += st.limD(a);
          if constexpr (FRIC) {
            const double fF = links[a].fl_floss, fR = links[a].fl_R, jf = ar.X[a] - st.fa(a);
            if (fF > 0 && jf > -fR && jf < fR) v += links[a].fl_D;
          }
        }
        if (T::GRIP && has_eq) {
          if (a == NA && bb == NA) v += eqD;
          if (a == NA + 1 && bb == NA) v += eqD * eqJ1;
          if (a == NA + 1 && bb == NA + 1) v += eqD * eqJ1 * eqJ1;
        }
      } else if (a == bb) {
        v = Mb[a - NL];
      }
      for (uint64_t m = hmask; m; m &= m - 1) {
        const int cc = (__ffsll((long long)m) - 1) / 3;
        const double* h = DL::hcone(ar, cc);
        const double* j0 = DL::jrow(ar, 3 * cc);
        const double* j1 = DL::jrow(ar, 3 * cc + 1);
        const double* j2 = DL::jrow(ar, 3 * cc + 2);
        const double a0 = j0[a], a1 = j1[a], a2 = j2[a], b0 = j0[bb], b1 = j1[bb], b2 = j2[bb];
        v += a0 * (h[0] * b0 + h[1] * b1 + h[3] * b2) + a1 * (h[1] * b0 + h[2] * b1 + h[4] * b2) + a2 * (h[3] * b0 + h[4] * b1 + h[5] * b2);
      }
      ar.H[idx] = v;
    }
    __syncthreads();
    TEAM_MARK(50)
    // ---- Newton direction p = -H^-1 grad: contact_newton's cooperative LDL' (lane i < NVD holds row i)
    double dphi0 = 0;
    {
      const int row = lane < NVD ? lane : NVD - 1;
      double hr[NVD];
#pragma unroll
      for (int k = 0; k < NVD; ++k) hr[k] = ar.H[row >= k ? tri(row, k) : tri(k, row)];
      TEAM_MARK(51)
#pragma unroll
      for (int j = 0; j < NVD - 1; ++j) {
        const double dj = wave_read(hr[j], j);
        const double lij = lane > j ? hr[j] / dj : 0.0;
#pragma unroll
        for (int k = j + 1; k < NVD; ++k) hr[k] -= lij * wave_read(hr[j], k);
      }
      double dg = 0;
#pragma unroll
      for (int k = 0; k < NVD; ++k) dg = row == k ? hr[k] : dg;
      const double dinv = 1.0 / dg, gl_ = lane < NVD ? ar.Gd[lane] : 0.0;
      double acc = -gl_;
#pragma unroll
      for (int k = 0; k < NVD - 1; ++k) {
        const double yk = wave_read(acc, k) * wave_read(dinv, k);
        if (lane > k) acc -= hr[k] * yk;
      }
      acc *= dinv;
#pragma unroll
      for (int k = NVD - 1; k >= 1; --k) {
        const double xk = wave_read(acc, k);
        if (lane < k) acc -= hr[k] * dinv * xk;
      }
      dphi0 = wave_sum(lane < NVD ? gl_ * acc : 0.0);
      if (lane < NVD) ar.P[lane] = acc;
    }
    TEAM_MARK(52)
    if (!(dphi0 < 0)) { newton_done = 1000 + newton_it; break; }
    __syncthreads();
    // ---- line search: root of phi'(a) by safeguarded 1-D Newton (contact_newton's, the rows' part from the explicit rows)
    double jd[3];
    {
      double jr = 0.0;
#pragma unroll
      for (int j = 0; j < NVD; ++j) jr += Jr[j] * ar.P[j];
      jd[0] = lane_get(jr, base); jd[1] = lane_get(jr, base + 1); jd[2] = lane_get(jr, base + 2);
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
    } else if (lane < NVD) {
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
      if (own) {
        double ja[3], fa[3], Ha[6];
#pragma unroll
        for (int k = 0; k < 3; ++k) ja[k] = jar[k] + a * jd[k];
        cone_eval(c.D, c.mu, c.fr, ja, fa, Ha);
        dl = -(jd[0] * fa[0] + jd[1] * fa[1] + jd[2] * fa[2]);
        ddl = jd[0] * (Ha[0] * jd[0] + Ha[1] * jd[1] + Ha[3] * jd[2]) + jd[1] * (Ha[1] * jd[0] + Ha[2] * jd[1] + Ha[4] * jd[2]) +
              jd[2] * (Ha[3] * jd[0] + Ha[4] * jd[1] + Ha[5] * jd[2]);
      }
      // (the robot's rows sit on lanes < NL, which are row lanes too: their terms are ADDED to the lane's contact terms)
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
    if (lane < NVD) {
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
  if (own) {
    double* r = ar.rec[crow];
    r[11] = f[0]; r[12] = f[1]; r[13] = f[2];
  }
  __syncthreads();
  TEAM_MARK(28)
}

// noslip + results for the explicit rows.  In: ar.X (the Newton minimiser), M's factor in ar.V, J in LDS, the records.
template <class T, bool BOXD, class AR>
RCSH_CONTACT_FN void contact_noslip_dense(const BoxCfg& b_, const StageTeam<T>& st_, double* bs_, AR& ar_) {
  const BoxCfg& b = *in_lds(&b_);
  const StageTeam<T> st{in_lds(st_.base)};
  double* bs = in_lds(bs_);
  AR& ar = *in_lds(&ar_);
  using DL = DenseLds<T, BOXD, AR>;
  constexpr int NL = T::NL, NA = T::NARM, NV = NL + 6, NVD = DL::NVD;
  constexpr int kWorld = NL + 1;
  const int lane = wave_lane();
  const int ncon = ar.ncon, nrow = 3 * ncon;
  const double Mbi[6] = {b.inv_mass, b.inv_mass, b.inv_mass, b.inv_inertia[0], b.inv_inertia[1], b.inv_inertia[2]};
  const int crow = lane / 3, krow = lane - 3 * crow;
  const bool row_on = lane < nrow, own = row_on && krow == 0;
  const int base = (3 * crow) & 63;
  ConLane c;
  con_load(ar, b, crow, ncon, kWorld, c);
  const bool has_eq = T::GRIP && st.eq(0) != 0.0;
  const double eqD = T::GRIP ? st.eq(0) : 0.0, eqAref = T::GRIP ? st.eq(1) : 0.0, eqJ1 = T::GRIP ? st.eq(2) : 0.0;
  double* const fl = DL::frow(ar);
  if (b.noslip_iterations > 0) {
    double Jr[NVD];
    {
      const double* src = DL::jrow(ar, row_on ? lane : 0);
#pragma unroll
      for (int j = 0; j < NVD; ++j) Jr[j] = row_on ? src[j] : 0.0;
    }
    // Y_r = M^-1 J_r': the robot's block by the factor contact_newton left in ar.V, the box's by its inverse inertia
    double own_diag = 0.0;
    {
      double LM[T::NTRI], col[NL];
#pragma unroll
      for (int e = 0; e < T::NTRI; ++e) LM[e] = (&ar.V[0][0])[e];
#pragma unroll
      for (int j = 0; j < NL; ++j) col[j] = Jr[j];
      ldl_solve<NL>(LM, col);
      double Yr[NVD];
#pragma unroll
      for (int j = 0; j < NL; ++j) Yr[j] = col[j];
      if constexpr (BOXD) {
#pragma unroll
        for (int k = 0; k < 6; ++k) Yr[NL + k] = Mbi[k] * Jr[NL + k];
      }
      if (row_on && krow > 0) {
        double* dst = DL::yrow(ar, crow, krow);
#pragma unroll
        for (int j = 0; j < NVD; ++j) dst[j] = Yr[j];
      }
      double a00 = 0.0;  // J_r Y_r: the normal row's diagonal entry of the contact's block
#pragma unroll
      for (int j = 0; j < NVD; ++j) a00 += Jr[j] * Yr[j];
      own_diag = a00;
    }
    __syncthreads();
    // the 3 x 3 block of A = J M^-1 J' (no regulariser) of the lane's contact: this lane's row against the two friction rows' Y, the
    // block's first column by symmetry from the normal row's lane
    double Ac[3][3];
    {
      double arow[3];
      arow[0] = own_diag;
#pragma unroll
      for (int l = 1; l < 3; ++l) {
        const double* y = DL::yrow(ar, row_on ? crow : 0, l);
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < NVD; ++j) s += Jr[j] * y[j];
        arow[l] = s;
      }
      // lane (c, 0): arow = (A00, A01, A02); lanes (c, 1), (c, 2): arow[1..2] = (A11, A12) / (A21, A22)
      const double a00 = lane_get(arow[0], base), a01 = lane_get(arow[1], base), a02 = lane_get(arow[2], base);
      const double a11 = lane_get(arow[1], base + 1), a12 = lane_get(arow[2], base + 1);
      const double a21 = lane_get(arow[1], base + 2), a22 = lane_get(arow[2], base + 2);
      Ac[0][0] = a00; Ac[0][1] = a01; Ac[0][2] = a02;
      Ac[1][0] = a01; Ac[1][1] = a11; Ac[1][2] = a12;
      Ac[2][0] = a02; Ac[2][1] = a21; Ac[2][2] = a22;
    }
    const double qS11 = Ac[1][1] * c.fr * c.fr, qS22 = Ac[2][2] * c.fr * c.fr, qS12 = Ac[1][2] * c.fr * c.fr;
    const double qdet = qS11 * qS22 - qS12 * qS12, qdi = 1 / qdet;
    // the residual of the lane's row at the Newton solution; kept current through the sweeps
    double res = 0.0;
    {
      double jr = -(krow == 0 ? c.aref[0] : (krow == 1 ? c.aref[1] : c.aref[2]));
#pragma unroll
      for (int j = 0; j < NVD; ++j) jr += Jr[j] * ar.X[j];
      res = row_on ? jr : 0.0;
    }
    TEAM_MARK(30)
    int iter = 0;
    while (iter < b.noslip_iterations) {
      TEAM_COUNT(36)
      double improvement = 0;
      if (iter == 0) {
        double s = own ? 0.5 * (c.f[0] * c.f[0] * c.Rr[0] + c.f[1] * c.f[1] * c.Rr[1] + c.f[2] * c.f[2] * c.Rr[2]) : 0.0;
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
        // what an update of contact cc does to this lane's row, per unit of force change: independent of the update's result
        double w3[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int l = 1; l < 3; ++l) {
          const double* y = DL::yrow(ar, cc, l);
          double s = 0.0;
#pragma unroll
          for (int j = 0; j < NVD; ++j) s += Jr[j] * y[j];
          w3[l] = s;
        }
        // the contact's three residuals, on its three lanes
        const double r0 = lane_get(res, base), r1 = lane_get(res, base + 1), r2 = lane_get(res, base + 2);
        double change = 0, df[3] = {0, 0, 0};
        int moved = 0;
        if (row_on && crow == cc) {
          const double old[3] = {c.f[0], c.f[1], c.f[2]};
          double nf[3] = {old[0], old[1], old[2]};
          if (old[0] < kMinVal) {
            // (a contact the Newton solution left without normal force: all three rows go to zero -- the general update)
            const double rs[3] = {r0, r1, r2};
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
            const double res1 = r1, res2 = r2;
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
          df[0] = nf[0] - old[0]; df[1] = nf[1] - old[1]; df[2] = nf[2] - old[2];
          c.f[0] = nf[0]; c.f[1] = nf[1]; c.f[2] = nf[2];
        }
        TEAM_MARK(41)
        if (wave_read(moved, 3 * cc)) {
          const double d1 = wave_read(df[1], 3 * cc), d2 = wave_read(df[2], 3 * cc);
          res += w3[1] * d1 + w3[2] * d2;  // (the normal force's change, below kMinVal where there is one: DenseLds::yrow)
        }
        TEAM_MARK(44)
        TEAM_COUNT(45)
        improvement -= wave_read(change, 3 * cc);
      }
      improvement *= b.scale;
      ++iter;
      if (improvement < b.noslip_tolerance) break;
    }
  }
  TEAM_MARK(31)
  // ---- results: qfrc_constraint of the robot, qacc of the box
  __syncthreads();
  if (row_on) fl[lane] = krow == 0 ? c.f[0] : (krow == 1 ? c.f[1] : c.f[2]);
  __syncthreads();
  {
    double qf = 0.0;
    if (lane < NVD)
      for (int r = 0; r < nrow; ++r) qf += DL::jrow(ar, r)[lane] * fl[r];
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
    } else if (BOXD && lane < NV) {  // (without a free box the phantom's block in LDS ends with its state: sim_kernels.h, kBoxStride)
      const int k = lane - NL;
      bs[kBoxA + k] = ar.A0[lane] + Mbi[k] * qf;
    }
  }
  __syncthreads();
  TEAM_MARK(32)
}

// The contact phase of one environment, executed by the whole wavefront.  `st` / `bs`: the environment's LDS blocks
// (robot: pre-step q, qd, motion axes S, mass matrix, qfrc_smooth, limit / equality rows of this substep; box: state).
// Returns bit 0: coupled (a robot geom is in contact: st.fcon holds the robot's constraint force, bs[kBoxA..] the box's
// acceleration); bit 4: more than kMaxCon contacts or more than kMaxActive links in contact (results then differ from MuJoCo's);
// bits 8-9: contact classes (bit 8 arm collision geoms, bit 9 gripper collision geoms) of this position stage.
#ifdef RCSH_PHASE_TIMING
#define PHASE_CLOCK(var) const unsigned long long var = __builtin_readcyclecounter();
#else
#define PHASE_CLOCK(var)
#endif
template <class T, bool FRIC = false, bool BOXD = true, class AR>
RCSH_D uint32_t contact_phase(const ContactTable& tab, const CheckTable& ck, const BoxCfg& b, const LinkRec* links, const StageTeam<T>& st, double* bs,
                              AR& ar, const double* gravity, int env, bool frames_ready = false) {
  PHASE_CLOCK(pc0)
  const uint32_t r = contact_collide<T, AR>(tab, ck, b, links, st, bs, ar, env, frames_ready);
  PHASE_CLOCK(pc1)
#ifdef RCSH_PHASE_TIMING
  if ((threadIdx.x & 63) == 0) {  // (all workgroups) contact phases / coupled / contacts / most contacts / solves on the tree formulation
    atomicAdd(&g_team_cycles[71], 1ull);
    if (r & 1u) {
      const unsigned long long nc_ = (unsigned long long)in_lds(&ar)->ncon;
      atomicAdd(&g_team_cycles[72], 1ull); atomicAdd(&g_team_cycles[70], nc_); atomicMax(&g_team_cycles[68], nc_);
      if (nc_ > (unsigned long long)DenseLds<T, BOXD, AR>::kDenseCon) atomicAdd(&g_team_cycles[69], 1ull);
    } else {
      atomicAdd(&g_team_cycles[82], pc1 - pc0);  // collision passes that found nothing
      s_wg_acc[7] += pc1 - pc0;
    }
    s_wg_acc[3] += 1;
  }
#endif
  if (!(r & 1u) || !b.resolve) return r & ~1u;
  if constexpr (!BOXD) {
    if (in_lds(&ar)->ncon > DenseLds<T, BOXD, AR>::kDenseCon || (ck.pad & 32)) {
    // no free box, many contacts: a lane per contact, its rows in registers, whatever their number (contact_wide.h).  (With few -- the
    // one or two of a link on the floor -- the rows-in-LDS form below is the cheaper one: its sums run over the rows there are, the
    // wide form's reductions over all 64 lanes whatever the count.)
    contact_newton_wide<T, FRIC, AR>(b, st, bs, ar, gravity, links);
    PHASE_CLOCK(pc2)
    contact_noslip_wide<T, AR>(b, st, bs, ar);
    PHASE_CLOCK(pc3)
#ifdef RCSH_PHASE_TIMING
    if ((threadIdx.x & 63) == 0) {
      const bool few_ = in_lds(&ar)->ncon <= 21;
      atomicAdd(&g_team_cycles[few_ ? 78 : 73], pc1 - pc0); atomicAdd(&g_team_cycles[few_ ? 79 : 74], pc2 - pc1); atomicAdd(&g_team_cycles[few_ ? 80 : 75], pc3 - pc2); atomicAdd(&g_team_cycles[few_ ? 81 : 76], 1ull);
      s_wg_acc[0] += pc1 - pc0; s_wg_acc[1] += pc2 - pc1; s_wg_acc[2] += pc3 - pc2; s_wg_acc[4] += 1; s_wg_acc[5] += (unsigned long long)in_lds(&ar)->ncon;
    }
#endif
    return r | (in_lds(&ar)->pad[0] ? 16u : 0u);
    }
  }
  const bool few = in_lds(&ar)->ncon <= DenseLds<T, BOXD, AR>::kDenseCon;
#ifndef RCSH_NO_DENSE
  if (few) {
    // few contacts (the headline's one or two, a pinch's dozen): the rows written out
    contact_newton_dense<T, FRIC, BOXD, AR>(b, st, bs, ar, gravity, links);
    PHASE_CLOCK(pc2)
    contact_noslip_dense<T, BOXD, AR>(b, st, bs, ar);
    PHASE_CLOCK(pc3)
#ifdef RCSH_PHASE_TIMING
    if ((threadIdx.x & 63) == 0) {
      atomicAdd(&g_team_cycles[78], pc1 - pc0); atomicAdd(&g_team_cycles[79], pc2 - pc1); atomicAdd(&g_team_cycles[80], pc3 - pc2); atomicAdd(&g_team_cycles[81], 1ull);
      s_wg_acc[0] += pc1 - pc0; s_wg_acc[1] += pc2 - pc1; s_wg_acc[2] += pc3 - pc2; s_wg_acc[4] += 1; s_wg_acc[5] += (unsigned long long)in_lds(&ar)->ncon;
    }
#endif
    return r | (in_lds(&ar)->pad[0] ? 16u : 0u);
  }
#endif
  if constexpr (BOXD) {  // (without a free box many contacts went to the wide form above)
  contact_newton<T, FRIC, AR>(b, st, bs, ar, gravity, links);
  PHASE_CLOCK(pc2)
  contact_noslip<T, AR>(b, st, bs, ar);
  PHASE_CLOCK(pc3)
#ifdef RCSH_PHASE_TIMING
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&g_team_cycles[73], pc1 - pc0); atomicAdd(&g_team_cycles[74], pc2 - pc1); atomicAdd(&g_team_cycles[75], pc3 - pc2); atomicAdd(&g_team_cycles[76], 1ull);
    s_wg_acc[0] += pc1 - pc0; s_wg_acc[1] += pc2 - pc1; s_wg_acc[2] += pc3 - pc2; s_wg_acc[4] += 1; s_wg_acc[5] += (unsigned long long)in_lds(&ar)->ncon; s_wg_acc[6] += 1;
  }
#endif
  (void)few;
  }
  return r | (in_lds(&ar)->pad[0] ? 16u : 0u);  // bit 4: a capacity of the contact phase overflowed in this substep
}

#endif  // __HIP__

}  // namespace rcsh
