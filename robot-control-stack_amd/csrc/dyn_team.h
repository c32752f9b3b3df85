// dyn_team.h -- one physics substep computed by a TEAM of 16 lanes per environment (team.h), lane t owning
// link / joint t.  Same physics as dyn.h's substep (what mj_step1 / mj_step2 compute for the RCS scenes,
// reference call sites src/sim/sim.cpp:110,112); a different decomposition:
//
//  * every per-link quantity (local frame, spatial inertia, bias wrench, actuator force, limit row) is computed
//    by the link's own lane, all links at once;
//  * root->leaf recursions (world frames, velocities, bias accelerations) are Hillis-Steele scans over the team
//    with DPP row shifts: 3 rounds for the 8-link chain + one round that hands the second finger its parent;
//  * leaf->root recursions (composite inertia, subtree wrench, gravity-compensation moment) are the mirrored
//    scans -- in world-origin Pluecker coordinates they are plain sums;
//  * the 9x9 factorisations do not parallelise over 9 lanes (a distributed LDL^T costs more exchange than it
//    saves), so the lanes factor DIFFERENT matrices instead: lane s < 2^k solves the constraint problem under the
//    s-th guess of the active set of the k joint-limit rows, lanes 11..15 factor the implicit-integrator matrix,
//    each with another right-hand side (qfrc_smooth and the Jacobians of the rows).  The cost is convex piecewise
//    quadratic, so the guess whose solution reproduces its own active set IS the minimiser: the Newton /
//    line-search iteration of dyn.h collapses into one factorisation slot, and the implicit solve for
//    qfrc_smooth + qfrc_constraint is a superposition of the helpers' solutions.  (k > 3 or no self-consistent
//    guess -- ties -- falls back to the iteration, run redundantly by the team; models with dry joint friction,
//    three zones per row, always iterate.)
//
// Exchange between lanes that is not a shift (motion axes for the mass-matrix rows, the matrix itself, the
// solver result) goes through the environment's LDS block (StageTeam), [slot] contiguous per environment.
#pragma once
#include "team.h"

namespace rcsh {

#if defined(__HIP__)

constexpr int even_up(int x) { return (x + 1) & ~1; }

// LDS block of one environment.  The first slots (q .. X) are the interface sim_kernels.h' environment code
// uses (same accessor names as dyn.h's Stage); the rest is the team's exchange area.
template <class T>
struct StageTeam {
  static constexpr int NL = T::NL;
  static constexpr int NLP = even_up(NL);
  static constexpr int Q0 = 0;                       // qpos
  static constexpr int V0 = Q0 + NLP;                // qvel
  static constexpr int C0 = V0 + NLP;                // ctrl
  static constexpr int P0 = C0 + even_up(T::NU);     // qpos seen by the last position stage
  static constexpr int K0 = P0 + NLP;                // frame of the site link: R(9) p(3)
  static constexpr int X0 = K0 + 12;                 // caller's slots
  static constexpr int NX = 6 + 2 * T::NARM;
  static constexpr int S0 = X0 + even_up(NX);        // motion axes [NL][6]
  static constexpr int MROW = NLP;                   // row stride of the mass matrix (rows 16-byte aligned)
  static constexpr int M0 = S0 + 6 * NL + (NL & 1) * 0;  // mass matrix rows (lower triangle valid)
  static constexpr int SM0 = even_up(M0 + MROW * NL);    // qfrc_smooth
  static constexpr int LD0 = SM0 + NLP;              // limit rows: D (0 = no row)
  static constexpr int LA0 = LD0 + NLP;              // limit rows: aref
  static constexpr int LS0 = LA0 + NLP;              // limit rows: sign of the Jacobian entry
  static constexpr int DG0 = LS0 + NLP;              // diagonal of -h dF/dqd (implicitfast)
  static constexpr int E0 = DG0 + NLP;               // eqD, eqAref, eqJ1, gblock
  static constexpr int XS0 = E0 + 4;                 // solver result (constrained qacc of the soft problem)
  static constexpr int QA0 = XS0 + NLP;              // qacc of the implicit solve
  static constexpr int FA0 = QA0 + NLP;              // dry-friction rows: aref (-B * qvel)
  static constexpr int Y0 = FA0 + NLP;               // helper lanes' solutions with the implicit matrix, [5][NLP]
  static constexpr int FC0 = Y0 + 5 * NLP;           // qfrc_constraint of a step whose contacts couple robot and box (contact_team.h)
  static constexpr int COUNT = FC0 + NLP;
  double* base;
  RCSH_D double& at(int k) const { return base[k]; }
  RCSH_D double& q(int i) const { return base[Q0 + i]; }
  RCSH_D double& v(int i) const { return base[V0 + i]; }
  RCSH_D double& c(int i) const { return base[C0 + i]; }
  RCSH_D double& qpre(int i) const { return base[P0 + i]; }
  RCSH_D double& link(int k) const { return base[K0 + k]; }
  RCSH_D double& X(int k) const { return base[X0 + k]; }
  RCSH_D double& S(int i, int k) const { return base[S0 + 6 * i + k]; }
  RCSH_D double& M(int i, int j) const { return base[M0 + MROW * i + j]; }
  RCSH_D double& smooth(int i) const { return base[SM0 + i]; }
  RCSH_D double& limD(int i) const { return base[LD0 + i]; }
  RCSH_D double& limA(int i) const { return base[LA0 + i]; }
  RCSH_D double& limS(int i) const { return base[LS0 + i]; }
  RCSH_D double& dg(int i) const { return base[DG0 + i]; }
  RCSH_D double& eq(int k) const { return base[E0 + k]; }
  RCSH_D double& xs(int i) const { return base[XS0 + i]; }
  RCSH_D double& qacc(int i) const { return base[QA0 + i]; }
  RCSH_D double& fa(int i) const { return base[FA0 + i]; }
  RCSH_D double& Y(int k, int i) const { return base[Y0 + NLP * k + i]; }
  RCSH_D double& fcon(int i) const { return base[FC0 + i]; }
};

#ifdef RCSH_PHASE_TIMING
// Development instrumentation (tools/team_timing.py): cycle counter deltas between marks, accumulated in LDS by lane
// 0 of workgroup 0 (an LDS round trip per mark, ~100 cycles) and flushed to global memory once per launch.
__device__ double g_slack_dbg[16];  // the contact phase's slack test: an example (contact_team.h)
__device__ unsigned long long g_team_cycles[96];  // 64..95: statistics over ALL workgroups (the contact-resolving launch of an escalated step)
__shared__ unsigned long long s_team_cycles[64];
__shared__ unsigned long long s_team_mark;
__shared__ unsigned long long s_wg_acc[8];  // this workgroup's contact phases: collide / Newton / noslip cycles, phases, coupled, contacts, tree-formulation solves
#define TEAM_MARK(idx)                                            \
  if (blockIdx.x == 0 && threadIdx.x == 0) {                      \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    s_team_cycles[idx] += now_ - s_team_mark;                     \
    s_team_mark = now_;                                           \
  }
#define TEAM_COUNT(idx) \
  if (blockIdx.x == 0 && threadIdx.x == 0) s_team_cycles[idx] += 1;
#define TEAM_CLOCK_START()                                        \
  if (blockIdx.x == 0 && threadIdx.x == 0) {                      \
    for (int k_ = 0; k_ < 64; ++k_) s_team_cycles[k_] = 0;        \
    for (int k_ = 0; k_ < 8; ++k_) s_wg_acc[k_] = 0;              \
    s_team_mark = __builtin_readcyclecounter();                   \
  }
#define TEAM_CLOCK_FLUSH()                                        \
  if (blockIdx.x == 0 && threadIdx.x == 0)                        \
    for (int k_ = 0; k_ < 64; ++k_) g_team_cycles[k_] += s_team_cycles[k_];
#else
#define TEAM_MARK(idx)
#define TEAM_COUNT(idx)
#define TEAM_CLOCK_START()
#define TEAM_CLOCK_FLUSH()
#endif

// LDS traffic of one wave is ordered; the barrier is there for the compiler (and costs nothing with one wave)
RCSH_D void team_sync() { __syncthreads(); }

// ---- scans over the link tree.  "Both fingers hang off the last arm link": lanes 0..NARM (the arm and the first finger)
// scan as a chain, lane NARM + 1 (second finger) is kept out of the chain rounds and receives its parent (two lanes
// up) in a round of its own.

// inclusive sum over a link's ancestors and itself.  `chain` is 1.0 on lanes 0..NARM and `second` 1.0 on lane NARM + 1
// (else 0.0): the shifts run unmasked with zero fill (no destination to initialise) and the masks ride in the FMA.
template <class T>
RCSH_D double scan_from_root(double x, double chain, double second) {
  if (T::GRIP) {
    x = fma(row_up<1>(x), chain, x);
    x = fma(row_up<2>(x), chain, x);
    x = fma(row_up<4>(x), chain, x);
    x = fma(row_up<2>(x), second, x);
  } else {
    x += row_up<1>(x);
    x += row_up<2>(x);
    x += row_up<4>(x);
  }
  return x;
}

// sum over a link's subtree (itself and all descendants).  Lanes >= NL must hold zero.
template <class T>
RCSH_D double scan_from_leaves(double x, bool first_finger) {
  double y = x;
  y += row_down<1>(y);
  y += row_down<2>(y);
  y += row_down<4>(y);
  if (T::NL > 8) y += row_down<8>(y);
  // the first finger is a leaf but sits below the second in lane order
  if (T::GRIP) y = first_finger ? x : y;
  return y;
}

// world frame of every link from the local frames: X_t = X_parent(t) * A_t
template <int N, int BANKS>
RCSH_D void compose_round(double* R, double* p) {
  double Rq[9], pq[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) Rq[k] = row_up_or_banks<N, BANKS>((k == 0 || k == 4 || k == 8) ? 1.0 : 0.0, R[k]);
#pragma unroll
  for (int k = 0; k < 3; ++k) pq[k] = row_up_or_banks<N, BANKS>(0.0, p[k]);
  double pn[3];
  mulmv(Rq, p, pn);
  p[0] = pn[0] + pq[0]; p[1] = pn[1] + pq[1]; p[2] = pn[2] + pq[2];
  mulmm(Rq, R, R);
}
// the same with the receiving lanes chosen by a predicate instead of by DPP bank (2 x 12 selects more per round)
template <int N>
RCSH_D void compose_round_if(double* R, double* p, bool take) {
  double Rq[9], pq[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const double id = (k == 0 || k == 4 || k == 8) ? 1.0 : 0.0;
    const double up = row_up_or<N>(id, R[k]);
    Rq[k] = take ? up : id;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double up = row_up<N>(p[k]);
    pq[k] = take ? up : 0.0;
  }
  double pn[3];
  mulmv(Rq, p, pn);
  p[0] = pn[0] + pq[0]; p[1] = pn[1] + pq[1]; p[2] = pn[2] + pq[2];
  mulmm(Rq, R, R);
}
template <class T>
RCSH_D void scan_frames(double* R, double* p) {
  if (T::GRIP && T::NARM == 7) {
    // fingers on lanes 7 and 8: the chain is banks 0-1, the second finger alone in bank 2 -- the DPP bank mask selects
    compose_round<1, 0x3>(R, p);
    compose_round<2, 0x3>(R, p);
    compose_round<4, 0x3>(R, p);
    compose_round<2, 0x4>(R, p);
  } else if (T::GRIP) {
    const int t = threadIdx.x & (kTeamLanes - 1);
    const bool chain = t <= T::NARM, second = t == T::NARM + 1;
    compose_round_if<1>(R, p, chain);
    compose_round_if<2>(R, p, chain);
    if (T::NARM + 1 > 4) compose_round_if<4>(R, p, chain);
    compose_round_if<2>(R, p, second);
  } else {
    compose_round<1, 0xf>(R, p);
    compose_round<2, 0xf>(R, p);
    compose_round<4, 0xf>(R, p);
  }
}

// Newton iteration with exact line search over the soft rows, run redundantly by every lane of a team from the LDS
// copy of the problem: the coupling equality (quadratic), the existing joint-limit rows (one-sided quadratic) and the
// dry-friction rows (Huber: quadratic inside |x_i - aref_i| < R * frictionloss, linear outside).  An iteration
// freezes every row in its current zone, solves the resulting linear system, and accepts the solution if it lands
// in the same zones (then it is the exact minimiser of the convex cost); otherwise it line-searches exactly along
// the step -- phi' is piecewise linear -- and repeats.  Reached when more than 3 limit rows exist, when no
// active-set guess was self-consistent, and always for models with dry friction (3 zones per row: too many
// guesses to enumerate).  Same algorithm as the oracle's solve_constraints (oracle/rcs_physics.c).
// FRIC = false compiles the friction rows out (frows is the constant 0), leaving dyn.h's loop.
template <class T, bool FRIC>
RCSH_D void newton_rows(const LinkRec* links, const StageTeam<T>& st, uint32_t limrows, bool has_eq, double eqD, double eqAref,
                        double eqJ1, double* x, bool seeded = false) {
  constexpr int NL = T::NL, NA = T::NARM;
  // dry-friction rows: D, frictionloss, aref, half-width of the quadratic zone
  double fD[NL], fF[NL], fA[NL], fR[NL];
  uint32_t frows = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    fF[i] = 0; fD[i] = 0; fA[i] = 0; fR[i] = 0;
    if constexpr (FRIC) {
      fF[i] = links[i].fl_floss; fD[i] = links[i].fl_D; fA[i] = st.fa(i); fR[i] = links[i].fl_R;
      if (fF[i] > 0) frows |= 1u << i;
    }
  }
  uint32_t act = limrows;      // limit rows in their quadratic zone (first guess: all)
  // friction rows in the linear zones.  First guess: the zones of the previous substep's solution, which the
  // environment's block still holds (zero at the first substep of a launch) -- MuJoCo warm-starts its solver the same
  // way (qacc_warmstart); the minimiser found does not depend on the guess, the iteration count does.
  uint32_t fneg = 0, fpos = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i)
    if (frows & (1u << i)) {
      const double jf = st.xs(i) - fA[i];
      if (jf <= -fR[i]) fneg |= 1u << i;
      else if (jf >= fR[i]) fpos |= 1u << i;
    }
  // seeded: x already holds an iterate (the factorisation slot's last solution): start from it and from the zones it lies in
  bool have_x = seeded;
  if (seeded) {
    act = 0; fneg = 0; fpos = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      if ((limrows & (1u << i)) && st.limS(i) * x[i] - st.limA(i) < 0) act |= 1u << i;
      if (frows & (1u << i)) {
        const double jf = x[i] - fA[i];
        if (jf <= -fR[i]) fneg |= 1u << i;
        else if (jf >= fR[i]) fpos |= 1u << i;
      }
    }
  }
  int iters_done = 0;
  for (int iter = 0; iter < 32; ++iter) {
    iters_done = iter + 1;
    double xn[NL];
    {
      double H[T::NTRI];
#pragma unroll
      for (int i = 0; i < NL; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) H[tri(i, j)] = st.M(i, j);
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        xn[i] = st.smooth(i);
        if (act & (1u << i)) {
          const double D = st.limD(i);
          H[tri(i, i)] += D;
          xn[i] += st.limS(i) * D * st.limA(i);
        }
        if (frows & (1u << i)) {
          if (fneg & (1u << i)) xn[i] += fF[i];
          else if (fpos & (1u << i)) xn[i] -= fF[i];
          else { H[tri(i, i)] += fD[i]; xn[i] += fD[i] * fA[i]; }
        }
      }
      if constexpr (T::GRIP) if (has_eq) {
        H[tri(NA, NA)] += eqD;
        H[tri(NA + 1, NA)] += eqD * eqJ1;
        H[tri(NA + 1, NA + 1)] += eqD * eqJ1 * eqJ1;
        xn[NA] += eqD * eqAref;
        xn[NA + 1] += eqD * eqAref * eqJ1;
      }
      ldl_factor<NL>(H);
      ldl_solve<NL>(H, xn);
    }
    uint32_t now = 0, nneg = 0, npos = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      if ((limrows & (1u << i)) && st.limS(i) * xn[i] - st.limA(i) < 0) now |= 1u << i;
      if (frows & (1u << i)) {
        const double jf = xn[i] - fA[i];
        if (jf <= -fR[i]) nneg |= 1u << i;
        else if (jf >= fR[i]) npos |= 1u << i;
      }
    }
    const bool same = now == act && nneg == fneg && npos == fpos;
    if (same || !have_x) {
#pragma unroll
      for (int i = 0; i < NL; ++i) x[i] = xn[i];
      have_x = true;
      if (same) break;
      act = now; fneg = nneg; fpos = npos;
      continue;
    }
    // exact line search from x along d = xn - x
    double d[NL], jar[NL], jd[NL];
    double p0 = 0, p1 = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) d[i] = xn[i] - x[i];
    for (int r = 0; r < NL; ++r) {
      double mx = -st.smooth(r), md = 0, dr = 0;
#pragma unroll
      for (int c = 0; c < NL; ++c) {
        const double mrc = r >= c ? st.M(r, c) : st.M(c, r);
        mx += mrc * x[c];
        md += mrc * d[c];
        dr = c == r ? d[c] : dr;
      }
      p0 += mx * dr;
      p1 += md * dr;
    }
    if constexpr (T::GRIP) if (has_eq) {
      const double je = x[NA] + eqJ1 * x[NA + 1] - eqAref, jde = d[NA] + eqJ1 * d[NA + 1];
      p0 += eqD * je * jde;
      p1 += eqD * jde * jde;
    }
    uint32_t on = 0, wneg = 0, wpos = 0;  // zones just past alpha while walking the line
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      jar[i] = 0; jd[i] = 0;
      if (limrows & (1u << i)) {
        const double sgn = st.limS(i);
        jar[i] = sgn * x[i] - st.limA(i);
        jd[i] = sgn * d[i];
        if (jar[i] < 0 || (jar[i] == 0 && jd[i] < 0)) on |= 1u << i;
      }
      if (frows & (1u << i)) {
        const double jf = x[i] - fA[i];
        if (jf < -fR[i] || (jf == -fR[i] && d[i] <= 0)) wneg |= 1u << i;
        else if (jf > fR[i] || (jf == fR[i] && d[i] >= 0)) wpos |= 1u << i;
      }
    }
    // The minimum of phi along the step.  phi' is piecewise linear and nondecreasing; its pieces end where a row crosses a
    // zone boundary: the limit row of joint i at aL, its friction row through -R at aN and through +R at aP.  All lanes of
    // the team hold the same x and d, so lane t takes crossing t (+ 16 per pass), evaluates phi' there with the slopes of the
    // pieces on either side, and the team picks the first crossing at which phi' has turned non-negative: the root lies on
    // the piece that ends there.  (A serial walk from piece to piece costs an order of magnitude more.)
    double alpha = 0;
    if constexpr (FRIC) {
      const int tlane = (int)(threadIdx.x & 15u);
      double jfv[NL], rdv[NL];
#pragma unroll
      for (int i = 0; i < NL; ++i) { jfv[i] = x[i] - fA[i]; rdv[i] = d[i] != 0 ? 1.0 / d[i] : 0.0; }
      // the first piece (zones just past alpha = 0): the root if nothing is crossed before it
      double c0 = p0, c1 = p1;
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        if ((limrows & (1u << i)) && (on & (1u << i))) { const double D = st.limD(i); c0 += D * jar[i] * jd[i]; c1 += D * jd[i] * jd[i]; }
        if (frows & (1u << i)) {
          if (wneg & (1u << i)) c0 -= fF[i] * d[i];
          else if (wpos & (1u << i)) c0 += fF[i] * d[i];
          else { c0 += fD[i] * jfv[i] * d[i]; c1 += fD[i] * d[i] * d[i]; }
        }
      }
      double a_first = INFINITY, root_first = INFINITY;  // the first crossing with phi' >= 0, the root on the piece before it
      double a_last = -INFINITY, root_last = INFINITY;   // the last crossing at all, the root on the piece after it
      const int ncross = limrows ? 3 * NL : 2 * NL;
      for (int k0 = 0; k0 < ncross; k0 += kTeamLanes) {
        const int k = k0 + tlane;
        // this lane's crossing: row, which boundary, where
        int row = -1, kind = 0;  // kind 0: friction through -R, 1: friction through +R, 2: limit
        double ak = INFINITY;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
          const bool fr = (frows >> i) & 1u, lr = (limrows >> i) & 1u;
          if (k == i && fr && d[i] != 0) { row = i; kind = 0; ak = (-fR[i] - jfv[i]) * rdv[i]; }
          if (k == NL + i && fr && d[i] != 0) { row = i; kind = 1; ak = (fR[i] - jfv[i]) * rdv[i]; }
          if (k == 2 * NL + i && lr && jd[i] != 0) { row = i; kind = 2; ak = -jar[i] / jd[i]; }
        }
        if (!(ak > 0)) ak = INFINITY;  // (behind the starting point: not on the way)
        const bool have = ak < INFINITY;
        const double at = have ? ak : 0.0;
        double val = p0 + at * p1, s_before = p1, s_after = p1;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
          if ((limrows >> i) & 1u) {
            const double D = st.limD(i), ra = jar[i] + at * jd[i];
            const bool own = kind == 2 && row == i;
            bool on_b = ra < 0, on_a = ra < 0;
            if (own) { on_b = jd[i] > 0; on_a = !on_b; }  // (rising through 0: active before, off after)
            val += ra < 0 && !own ? D * ra * jd[i] : 0.0;   // (at its own crossing the row's residual is 0)
            s_before += on_b ? D * jd[i] * jd[i] : 0.0;
            s_after += on_a ? D * jd[i] * jd[i] : 0.0;
          }
          if ((frows >> i) & 1u) {
            const double ja = jfv[i] + at * d[i];
            int zb = ja <= -fR[i] ? 0 : (ja >= fR[i] ? 2 : 1), za = zb;  // 0: below -R, 1: quadratic, 2: above +R
            if (kind != 2 && row == i) {
              if (kind == 0) { zb = d[i] > 0 ? 0 : 1; za = d[i] > 0 ? 1 : 0; }
              else { zb = d[i] > 0 ? 1 : 2; za = d[i] > 0 ? 2 : 1; }
            }
            // (phi' is continuous: at the row's own crossing both neighbouring pieces give the same value; the piece before is used)
            val += zb == 0 ? -fF[i] * d[i] : (zb == 2 ? fF[i] * d[i] : fD[i] * ja * d[i]);
            s_before += zb == 1 ? fD[i] * d[i] * d[i] : 0.0;
            s_after += za == 1 ? fD[i] * d[i] * d[i] : 0.0;
          }
        }
        if (have && val >= 0 && ak < a_first) { a_first = ak; root_first = ak - val / s_before; }
        if (have && ak > a_last) { a_last = ak; root_last = ak - val / s_after; }
      }
      const double af = team_min(a_first);
      if (af < INFINITY) {
        alpha = team_min(a_first == af ? root_first : INFINITY);
      } else {
        const double al = -team_min(-a_last);
        alpha = al > -INFINITY ? team_min(a_last == al ? root_last : INFINITY) : -c0 / c1;
      }
      if (!(alpha > 0)) alpha = -c0 / c1 > 0 ? -c0 / c1 : 0.0;  // (round-off at a crossing: fall back to the first piece)
    } else {
      // (models without dry friction reach this routine rarely -- more than three limit rows, or no self-consistent guess --
      // and keep the serial walk from piece to piece: the headline kernel's code stays what it was)
      for (int guard = 0; guard < 3 * NL + 2; ++guard) {
        double c0 = p0, c1 = p1, a_next = INFINITY;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
          if (limrows & (1u << i)) {
            const double D = st.limD(i);
            if (on & (1u << i)) { c0 += D * jar[i] * jd[i]; c1 += D * jd[i] * jd[i]; }
            if (jd[i] != 0) {
              const double ab = -jar[i] / jd[i];
              if (ab > alpha && ab < a_next) a_next = ab;
            }
          }
          if (frows & (1u << i)) {
            const double jf = x[i] - fA[i];
            const bool ln = wneg & (1u << i), lp = wpos & (1u << i);
            if (ln) c0 -= fF[i] * d[i];
            else if (lp) c0 += fF[i] * d[i];
            else { c0 += fD[i] * jf * d[i]; c1 += fD[i] * d[i] * d[i]; }
            if (d[i] != 0 && !(d[i] > 0 ? lp : ln)) {
              const double bound = d[i] > 0 ? (ln ? -fR[i] : fR[i]) : (lp ? fR[i] : -fR[i]);
              const double ab = (bound - jf) / d[i];
              if (ab > alpha && ab < a_next) a_next = ab;
            }
          }
        }
        const double a_star = -c0 / c1;
        if (a_star <= a_next) { if (a_star > alpha) alpha = a_star; break; }
        alpha = a_next;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
          if ((limrows & (1u << i)) && jd[i] != 0 && -jar[i] / jd[i] == a_next) on ^= 1u << i;
          if ((frows & (1u << i)) && d[i] != 0) {
            const double jf = x[i] - fA[i];
            const bool ln = wneg & (1u << i), lp = wpos & (1u << i);
            if (d[i] > 0) {
              if (ln && (-fR[i] - jf) / d[i] == a_next) wneg &= ~(1u << i);
              else if (!ln && !lp && (fR[i] - jf) / d[i] == a_next) wpos |= 1u << i;
            } else {
              if (lp && (fR[i] - jf) / d[i] == a_next) wpos &= ~(1u << i);
              else if (!ln && !lp && (-fR[i] - jf) / d[i] == a_next) wneg |= 1u << i;
            }
          }
        }
      }
    }
    act = 0; fneg = 0; fpos = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      x[i] += alpha * d[i];
      if ((limrows & (1u << i)) && st.limS(i) * x[i] - st.limA(i) < 0) act |= 1u << i;
      if (frows & (1u << i)) {
        const double jf = x[i] - fA[i];
        if (jf <= -fR[i]) fneg |= 1u << i;
        else if (jf >= fR[i]) fpos |= 1u << i;
      }
    }
  }
#ifdef RCSH_PHASE_TIMING
  if ((threadIdx.x & 15) == 0) {  // (every team) solves, their iterations, the worst, solves over 4 / at the cap
    atomicAdd(&g_team_cycles[44], 1ull);
    atomicAdd(&g_team_cycles[45], (unsigned long long)iters_done);
    atomicMax(&g_team_cycles[46], (unsigned long long)iters_done);
    if (iters_done > 4) atomicAdd(&g_team_cycles[47], 1ull);
    if (iters_done >= 32) atomicAdd(&g_team_cycles[43], 1ull);
  }
#endif
  (void)iters_done;
}

// ---- per-lane model constants, fetched from the LDS model tables in one batch per phase.  A batch is issued
// BEFORE the compute of the previous phase (sched_fence keeps it there), so its LDS latency hides behind that
// compute: the wave is alone on its SIMD and nothing else would.
struct KinK {
  double qpos0, rot0[9], pos0[3], axis[3], jpos[3];
  int32_t axis_z, jtype;
  RCSH_D void load(const LinkRec& r) {
    qpos0 = r.qpos0;
#pragma unroll
    for (int k = 0; k < 9; ++k) rot0[k] = r.rot0[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) { pos0[k] = r.pos0[k]; axis[k] = r.axis[k]; jpos[k] = r.jpos[k]; }
    axis_z = r.axis_z;
    jtype = r.jtype;
  }
};
// frame of a link in its parent link's frame at joint position q (kinematic part of mj_kinematics for one joint)
RCSH_D void link_local_frame(const KinK& kk, double q, double* R, double* p) {
  const double dq = q - kk.qpos0;
  const double* r0 = kk.rot0;
  const double* p0 = kk.pos0;
  if (kk.axis_z) {
    double s, c;
    fast_sincos(dq, &s, &c);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      R[3 * r + 0] = c * r0[3 * r + 0] + s * r0[3 * r + 1];
      R[3 * r + 1] = c * r0[3 * r + 1] - s * r0[3 * r + 0];
      R[3 * r + 2] = r0[3 * r + 2];
      p[r] = p0[r];
    }
  } else if (kk.jtype == kSlide) {
    double ax[3];
    mulmv(r0, kk.axis, ax);
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = r0[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = p0[k] + ax[k] * dq;
  } else {
    double s, c;
    fast_sincos(dq, &s, &c);
    const double* a = kk.axis;
    const double u = 1.0 - c;
    const double Q[9] = {c + u * a[0] * a[0],        u * a[0] * a[1] - s * a[2], u * a[0] * a[2] + s * a[1],
                         u * a[0] * a[1] + s * a[2], c + u * a[1] * a[1],        u * a[1] * a[2] - s * a[0],
                         u * a[0] * a[2] - s * a[1], u * a[1] * a[2] + s * a[0], c + u * a[2] * a[2]};
    double anchor[3], rj[3];
    mulmv(r0, kk.jpos, anchor);
    mulmm(r0, Q, R);
    mulmv(R, kk.jpos, rj);
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = p0[k] + anchor[k] - rj[k];
  }
}

struct InertK {
  double mass, gcm, com[3], J[6];
  int32_t gc_same_com;
  RCSH_D void load(const LinkRec& r) {
    mass = r.mass;
    gcm = r.gcm;
#pragma unroll
    for (int k = 0; k < 3; ++k) com[k] = r.com[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) J[k] = r.inertia[k];
    gc_same_com = r.gc_same_com;
  }
};
struct ActK {
  double ctrlrange[2], gear, gain, bias[3], forcerange[2];
  double damping, actfrcrange[2], armature, gcm_sub, range[2], margin, actgravcomp_w;
  RCSH_D void load(const LinkRec& r) {
    ctrlrange[0] = r.arm_ctrlrange[0]; ctrlrange[1] = r.arm_ctrlrange[1];
    gear = r.arm_gear; gain = r.arm_gain;
    bias[0] = r.arm_bias[0]; bias[1] = r.arm_bias[1]; bias[2] = r.arm_bias[2];
    forcerange[0] = r.arm_forcerange[0]; forcerange[1] = r.arm_forcerange[1];
    damping = r.damping;
    actfrcrange[0] = r.actfrcrange[0]; actfrcrange[1] = r.actfrcrange[1];
    armature = r.armature; gcm_sub = r.gcm_sub;
    range[0] = r.range[0]; range[1] = r.range[1]; margin = r.margin;
    actgravcomp_w = r.actgravcomp_w;
  }
};

// true when every link's gravity-compensated mass and centre are its mass and centre of mass (gravcomp = 1 on all
// bodies): then two of the leaf->root scans of team_substep are redundant.  Same answer on every lane of the wave.
template <class T>
RCSH_D bool team_gc_is_mass(const LinkRec* links, int t) {
  const int tl = t < T::NL ? t : T::NL - 1;
  return __ballot(!(links[tl].gcm == links[tl].mass && links[tl].gc_same_com)) == 0;
}

// wave-uniform model scalars team_substep reads every substep, fetched from the LDS copy once per launch
struct SubstepK {
  double h, gravity[3], grp_c0, grp_c1;
  int32_t site_link, eq_active;
  RCSH_D void load(const DevModelHead& m) {
    h = m.timestep;
    gravity[0] = m.gravity[0]; gravity[1] = m.gravity[1]; gravity[2] = m.gravity[2];
    grp_c0 = m.grp_coef[0]; grp_c1 = m.grp_coef[1];
    site_link = m.site_link; eq_active = m.eq_active;
  }
};

// One substep of the environment whose LDS block is `st`, executed by its 16 lanes together (t = lane in team).
// Reads qpos / qvel / ctrl from the block and, if `stepping`, writes the advanced qpos / qvel, the pre-step qpos
// and the pre-step world frame of the attachment-site link back (same contract as dyn.h's substep).
// `on_frame(R, p)` is called on every lane with the world frame of the lane's link at the pre-step qpos (what the
// contact detection of the last mj_step1 sees).
// Contains team_sync()s: every lane of the wave must call it.
// `gc_is_mass` (wave-uniform, see team_gc_is_mass): shortcut for fully gravity-compensated models.
// FRIC: the model has dry joint friction rows (dof_frictionloss); a separate instantiation so that models without
// them carry none of that code.
// `pre_solve()` is called by every lane once the substep's rows and qfrc_smooth are in the block; it returns true (for the
// whole team) when the contact phase solved the step's constraints itself (contact_team.h: st.fcon then holds
// qfrc_constraint), which leaves only the implicit solve and the integration to this function.
template <class T, bool FRIC, class FrameFn, class PreSolveFn>
RCSH_D void team_substep(const DevModelHead& m, const SubstepK& sk, const LinkRec* links, const StageTeam<T>& st, int t, bool stepping,
                         bool gc_is_mass, FrameFn&& on_frame, PreSolveFn&& pre_solve) {
  static_assert(!T::GRIP || T::NARM + 1 < 8 || T::NARM == 7, "the chain rounds of the scans reach 8 lanes; NARM = 7 uses the bank masks");
  static_assert(T::NL <= kTeamLanes - 1, "lane 15 is the implicit-integrator lane");
  constexpr int NL = T::NL, NA = T::NARM;
  const bool valid = t < NL;
  const int tl = valid ? t : NL - 1;
  const int ta = t < NA ? t : 0;
  const double h = sk.h;
  const LinkRec& lk = links[tl];
  KinK kk;
  kk.load(lk);
  const double q = st.q(tl), qd = st.v(tl);
  const double ctrl = st.c(ta);
  sched_fence();
  const bool is_slide = kk.jtype == kSlide;

  // ---- local frame of the link in its parent link's frame
  double R[9], p[3];
  link_local_frame(kk, q, R, p);
  TEAM_MARK(0)
  InertK ik;
  ik.load(lk);
  sched_fence();
  scan_frames<T>(R, p);  // now the world frame
  TEAM_MARK(1)
  on_frame(R, p);
  if (stepping && valid) st.qpre(tl) = q;
  if (stepping && t == sk.site_link) {
#pragma unroll
    for (int k = 0; k < 9; ++k) st.link(k) = R[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) st.link(9 + k) = p[k];
  }

  // ---- motion axis about the world origin, velocity and bias acceleration of the link
  double S[6];
  {
    double ax[3], anchor[3];
    mulmv(R, kk.axis, ax);
    mulmv(R, kk.jpos, anchor);
    anchor[0] += p[0]; anchor[1] += p[1]; anchor[2] += p[2];
    double mom[3];
    cross3(anchor, ax, mom);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      S[k] = is_slide ? 0.0 : ax[k];
      S[3 + k] = is_slide ? ax[k] : mom[k];
    }
  }
  if (valid) {
#pragma unroll
    for (int k = 0; k < 6; ++k) st.S(tl, k) = S[k];
  }
  double vel[6], acc[6];
  const double chain = t <= NA ? 1.0 : 0.0, second = t == NA + 1 ? 1.0 : 0.0;  // (only read by archetypes with a gripper)
#pragma unroll
  for (int k = 0; k < 6; ++k) vel[k] = scan_from_root<T>(S[k] * qd, chain, second);
  {
    double sd[6];
    cross_motion(vel, S, sd);  // S x S = 0: the link's own joint velocity does not contribute
#pragma unroll
    for (int k = 0; k < 6; ++k) acc[k] = scan_from_root<T>(sd[k] * qd, chain, second);
    acc[3] -= sk.gravity[0]; acc[4] -= sk.gravity[1]; acc[5] -= sk.gravity[2];
  }
  TEAM_MARK(2)
  ActK ak;
  ak.load(lk);
  sched_fence();

  // ---- spatial inertia about the world origin, bias wrench, gravity-compensation first moment
  double Ic[10], F[6], hs[3];
  {
    const double ms = valid ? ik.mass : 0.0;
    const double gcm = valid ? ik.gcm : 0.0;
    double c[3], cg[3];
    mulmv(R, ik.com, c);
    c[0] += p[0]; c[1] += p[1]; c[2] += p[2];
    if (ik.gc_same_com) {
      cg[0] = c[0]; cg[1] = c[1]; cg[2] = c[2];
    } else {
      mulmv(R, lk.gccom, cg);
      cg[0] += p[0]; cg[1] += p[1]; cg[2] += p[2];
    }
    const double* J = ik.J;
    const double vz = valid ? 1.0 : 0.0;
    const double Jm[9] = {vz * J[0], vz * J[3], vz * J[4], vz * J[3], vz * J[1], vz * J[5], vz * J[4], vz * J[5], vz * J[2]};
    double Tm[9];
    mulmm(R, Jm, Tm);
    double Ii[10];
    Ii[0] = Tm[0] * R[0] + Tm[1] * R[1] + Tm[2] * R[2] + ms * (c[1] * c[1] + c[2] * c[2]);
    Ii[1] = Tm[3] * R[3] + Tm[4] * R[4] + Tm[5] * R[5] + ms * (c[0] * c[0] + c[2] * c[2]);
    Ii[2] = Tm[6] * R[6] + Tm[7] * R[7] + Tm[8] * R[8] + ms * (c[0] * c[0] + c[1] * c[1]);
    Ii[3] = Tm[0] * R[3] + Tm[1] * R[4] + Tm[2] * R[5] - ms * c[0] * c[1];
    Ii[4] = Tm[0] * R[6] + Tm[1] * R[7] + Tm[2] * R[8] - ms * c[0] * c[2];
    Ii[5] = Tm[3] * R[6] + Tm[4] * R[7] + Tm[5] * R[8] - ms * c[1] * c[2];
    Ii[6] = ms * c[0]; Ii[7] = ms * c[1]; Ii[8] = ms * c[2];
    Ii[9] = ms;
    double Ia[6], Iv[6], vf[6];
    inert_mul(Ii, acc, Ia);
    inert_mul(Ii, vel, Iv);
    cross_force(vel, Iv, vf);
    const bool ff = T::GRIP && t == NA;
#pragma unroll
    for (int k = 0; k < 9; ++k) Ic[k] = scan_from_leaves<T>(Ii[k], ff);
#pragma unroll
    for (int k = 0; k < 6; ++k) F[k] = scan_from_leaves<T>(Ia[k] + vf[k], ff);
    if (gc_is_mass) {
      // every body fully gravity-compensated (gravcomp = 1, the RCS scenes): the compensated first moment IS the
      // first moment of the composite inertia, and the subtree mass is the model constant gcm_sub
      Ic[9] = valid ? ak.gcm_sub : 0.0;
      hs[0] = Ic[6]; hs[1] = Ic[7]; hs[2] = Ic[8];
    } else {
      Ic[9] = scan_from_leaves<T>(Ii[9], ff);
#pragma unroll
      for (int k = 0; k < 3; ++k) hs[k] = scan_from_leaves<T>(gcm * cg[k], ff);
    }
  }
  TEAM_MARK(3)

  // ---- mass-matrix row of the link: M[t][j] = S_j . (Ic_t S_t) for the ancestors j (and itself)
  double G[6];
  inert_mul(Ic, S, G);
  team_sync();  // every lane's S is in the block
  {
    double row[StageTeam<T>::MROW];
#pragma unroll
    for (int j0 = 0; j0 < NL; j0 += 3) {
      double Sj[3][6];
#pragma unroll
      for (int j = j0; j < j0 + 3 && j < NL; ++j)
#pragma unroll
        for (int k = 0; k < 6; ++k) Sj[j - j0][k] = st.S(j, k);
      sched_fence();
#pragma unroll
      for (int j = j0; j < j0 + 3 && j < NL; ++j) row[j] = dot6(Sj[j - j0], G);
    }
    if (T::GRIP) row[NA] = t == NA + 1 ? 0.0 : row[NA];  // the fingers are siblings
#pragma unroll
    for (int j = 0; j < NL; ++j) row[j] += j == tl ? ak.armature : 0.0;
    if (valid) {
#pragma unroll
      for (int j = 0; j < NL; ++j) st.M(tl, j) = row[j];  // entries right of the diagonal are never read
    }
  }
  const double bias = dot6(S, F);
  double gc;
  {
    const double ng[3] = {-sk.gravity[0], -sk.gravity[1], -sk.gravity[2]};
    double w[6];
    cross3(hs, ng, w);
    w[3] = ak.gcm_sub * ng[0]; w[4] = ak.gcm_sub * ng[1]; w[5] = ak.gcm_sub * ng[2];
    gc = dot6(S, w);
  }
  TEAM_MARK(4)

  // ---- actuation (lane t: actuator of joint t; the gripper actuator pulls on both finger lanes)
  // (one formula for every lane: LinkRec holds zero coefficients / infinite ranges where the model has no actuator, no
  // bias, no limit)
  double tau;
  bool unclamped;  // the actuator's velocity derivative enters the implicit matrix unless forcerange saturates it
  {
    const double c = clampd(ctrl, ak.ctrlrange[0], ak.ctrlrange[1]);
    const double force = ak.gain * c + (ak.bias[0] + ak.bias[1] * (ak.gear * q) + ak.bias[2] * (ak.gear * qd));
    unclamped = !(force <= ak.forcerange[0] || force >= ak.forcerange[1]);
    tau = ak.gear * clampd(force, ak.forcerange[0], ak.forcerange[1]);
  }
  double gblock = 0.0, eqD = 0.0, eqAref = 0.0, eqJ1 = 0.0;
  if (T::GRIP) {
    // both finger lanes see both fingers' state
    const bool f1 = t == NA;
    const double q_up = row_up<1>(q), q_dn = row_down<1>(q), v_up = row_up<1>(qd), v_dn = row_down<1>(qd);
    const double q1 = f1 ? q : q_up, q2 = f1 ? q_dn : q, v1 = f1 ? qd : v_up, v2 = f1 ? v_dn : qd;
    if (t == NA || t == NA + 1) {
      // gripper constants: one batch
      const int32_t g_has = m.grp_has_act, g_cl = m.grp_ctrllimited, g_ba = m.grp_biasaffine, g_fl = m.grp_forcelimited;
      const int32_t e_on = sk.eq_active;
      const double g_c0 = sk.grp_c0, g_c1 = sk.grp_c1, g_gain = m.grp_gain;
      const double g_b0 = m.grp_bias[0], g_b1 = m.grp_bias[1], g_b2 = m.grp_bias[2];
      const double g_cr0 = m.grp_ctrlrange[0], g_cr1 = m.grp_ctrlrange[1], g_fr0 = m.grp_forcerange[0], g_fr1 = m.grp_forcerange[1];
      const double gctrl = st.c(NA);
      const double pc0 = m.eq_polycoef[0], pc1 = m.eq_polycoef[1], pc2 = m.eq_polycoef[2], pc3 = m.eq_polycoef[3], pc4 = m.eq_polycoef[4];
      const double q0a = links[NA].qpos0, q0b = links[NA + 1].qpos0, iwa = links[NA].invweight0, iwb = links[NA + 1].invweight0;
      const double eK = m.eq_K, eB = m.eq_B;
      const Imp eimp = m.eq_imp;
      sched_fence();
      if (g_has) {
        double c = gctrl;
        if (g_cl) c = clampd(c, g_cr0, g_cr1);
        const double len = g_c0 * q1 + g_c1 * q2;
        const double lv = g_c0 * v1 + g_c1 * v2;
        double force = g_gain * c;
        if (g_ba) force += g_b0 + g_b1 * len + g_b2 * lv;
        bool clamped = false;
        if (g_fl) {
          clamped = force <= g_fr0 || force >= g_fr1;
          force = clampd(force, g_fr0, g_fr1);
        }
        tau += (f1 ? g_c0 : g_c1) * force;
        if (g_ba && !clamped) gblock = -g_b2;
      }
      if (e_on) {
        const double dif = q2 - q0b;
        const double poly = pc0 + dif * (pc1 + dif * (pc2 + dif * (pc3 + dif * pc4)));
        const double deriv = pc1 + dif * (2 * pc2 + dif * (3 * pc3 + dif * 4 * pc4));
        const double pos = q1 - q0a - poly;
        eqJ1 = -deriv;
        const double imp = impedance(eimp, pos, 0.0);
        eqD = row_D(imp, iwa + iwb);
        eqAref = -eK * imp * pos - eB * (v1 + eqJ1 * v2);
      }
      if (f1) { st.eq(0) = eqD; st.eq(1) = eqAref; st.eq(2) = eqJ1; st.eq(3) = gblock; }
    }
  }
  TEAM_MARK(15)
  double smooth;
  {
    const double passive = -ak.damping * qd + (1.0 - ak.actgravcomp_w) * gc;
    tau = clampd(tau + ak.actgravcomp_w * gc, ak.actfrcrange[0], ak.actfrcrange[1]);
    smooth = passive - bias + tau;
  }
  // ---- joint-limit row of the lane's joint
  double lD = 0.0, lA = 0.0, lS = 0.0;
  {
    const double dlo = q - ak.range[0], dhi = ak.range[1] - q;
    const double mg = ak.margin;
    double dist = 0, sgn = 0;
    if (dlo < mg) { dist = dlo; sgn = 1; }
    else if (dhi < mg) { dist = dhi; sgn = -1; }
    if (sgn != 0) {
      const Imp limp = lk.lim_imp;
      const double lK = lk.lim_K, lB = lk.lim_B, iw = lk.invweight0;
      sched_fence();
      const double imp = impedance(limp, dist, mg);
      lD = row_D(imp, iw);
      lA = -lK * imp * (dist - mg) - lB * (sgn * qd);
      lS = sgn;
    }
  }
  const uint32_t limrows = team_ballot(valid && lS != 0.0);
  if (valid) {
    const double d = ak.damping - (unclamped ? ak.gear * ak.gear * ak.bias[2] : 0.0);
    st.smooth(tl) = smooth;
    st.limD(tl) = lD;
    st.limA(tl) = lA;
    st.limS(tl) = lS;
    st.dg(tl) = h * d;
    if constexpr (FRIC) st.fa(tl) = -lk.fl_B * qd;  // dry-friction row of the joint: aref (zero stiffness)
  }
  team_sync();
  TEAM_MARK(5)
  const bool coupled = pre_solve();

  // ---- the factorisation slot.  Roles by lane (all run the same instructions on different matrices / right-hand sides):
  //   solver lanes s < 2^k   H = M + rows of the s-th active-set guess,        rhs = qfrc_smooth + row terms
  //   helper lanes 11..15    H = A = M - h dF/dqd (the implicitfast matrix),   rhs = qfrc_smooth (15), the coupling
  //                          row's Jacobian (14), the unit vectors of the up-to-3 limit-row joints (13, 12, 11)
  // The constraint force is a combination of those Jacobians, so once the winning guess is known
  //   qacc = A^-1 (qfrc_smooth + qfrc_constraint) = y15 + fe y14 + c0 y13 + c1 y12 + c2 y11
  // is five multiply-adds per lane -- no second, serial solve.  (Models with dry friction put a force on every joint:
  // they keep one implicit lane, 15, that solves again after the constraint solve; so does the rare fallback.)
  const bool has_eq = T::GRIP && sk.eq_active;
  const int nrows = __popc(limrows);
  const bool helper_lane = FRIC ? t == kTeamLanes - 1 : t >= kTeamLanes - 5;
  // idx0..2: the joints of the first three limit rows (-1: no such row); lane s guesses that the rows whose bit is set
  // in s are the active ones (only read when there are at most three rows)
  const uint32_t rows1 = limrows & (limrows - 1), rows2 = rows1 & (rows1 - 1);
  const int idx0 = __ffs(limrows) - 1, idx1 = __ffs(rows1) - 1, idx2 = __ffs(rows2) - 1;
  const uint32_t act = ((t & 1) && idx0 >= 0 ? 1u << idx0 : 0u) | ((t & 2) && idx1 >= 0 ? 1u << idx1 : 0u) |
                       ((t & 4) && idx2 >= 0 ? 1u << idx2 : 0u);
  const bool fast = nrows <= 3 && !FRIC;
  const bool solver_lane = fast && t < (1 << nrows);
  double H[T::NTRI], x[NL];
  // (dry friction: the lane's candidate -- friction rows at +frictionloss / at -frictionloss, limit rows active -- and the rows' data)
  uint32_t frows = 0, c_neg = 0, c_pos = 0, c_act = 0;
  bool cand_lane = false;
  double fFv[FRIC ? NL : 1], fDv[FRIC ? NL : 1], fRv[FRIC ? NL : 1], fAv[FRIC ? NL : 1];
  // (dry friction: the slot runs up to kFricRounds times -- a round without a self-consistent candidate hands the zones lane 0's
  // solution landed in to the next one as its base, which is what the serial iteration's next step would solve)
  constexpr int kFricRounds = 3;
  uint32_t b_neg = 0, b_pos = 0, b_act = 0, b_side = 0, winners = 0;
  int rounds_used = 0;
  uint32_t dbg_n0 = 0, dbg_p0 = 0;
  (void)dbg_n0; (void)dbg_p0;
  int b_m = 0;
  double xprev[FRIC ? NL : 1];
  for (int round = 0; round < (FRIC ? kFricRounds : 1); ++round) {
#pragma unroll
  for (int i = 0; i < NL; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) H[tri(i, j)] = st.M(i, j);
  {
    double sm[NL], lDv[NL], lAv[NL], lSv[NL], dgv[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) { sm[i] = st.smooth(i); lDv[i] = st.limD(i); lAv[i] = st.limA(i); lSv[i] = st.limS(i); dgv[i] = st.dg(i); }
    if (T::GRIP) { eqD = st.eq(0); eqAref = st.eq(1); eqJ1 = st.eq(2); gblock = st.eq(3); }
    sched_fence();
    // Role by arithmetic instead of selects: wh = 1 on helper lanes (their diagonal term), ws = 1 where the right-hand
    // side starts from qfrc_smooth (solver lanes and the smooth helper); the guess `act` is empty on helper lanes, so
    // the row terms vanish there by themselves.  Unit-vector helpers: which joint; coupling helper: e_NA + eqJ1 e_NA+1.
    const int unit = FRIC ? -1 : (t == kTeamLanes - 3 ? idx0 : (t == kTeamLanes - 4 ? idx1 : (t == kTeamLanes - 5 ? idx2 : -1)));
    const bool eq_helper = !FRIC && T::GRIP && t == kTeamLanes - 2;
    const double wh = helper_lane ? 1.0 : 0.0, ws = (!helper_lane || t == kTeamLanes - 1) ? 1.0 : 0.0;
    if constexpr (FRIC) {
      // Dry friction: three zones per row are too many to enumerate, but the zones rarely move by more than one row from one
      // substep to the next.  Lane 0 takes the zones of the previous substep's solution (friction rows and limit rows alike),
      // lanes 1..14 the same with ONE friction row moved to one of its two other zones (joint (t - 1) / 2, alternative
      // (t - 1) % 2); a lane whose solution lands in the zones it assumed has the minimiser of the convex cost.  Nobody:
      // another round from the zones lane 0's solution landed in (single moves to the neighbouring zone and pairs with the
      // row closest to its boundary), then a third; only after that newton_rows below, seeded with lane 0's last solution.
      frows = 0; c_neg = 0; c_pos = 0; c_act = 0; cand_lane = false;
      // (selects, not branches: the rows' data is the same in every lane, but the compiler cannot know, and a scalar branch
      // per row and test costs this lonely wavefront more than the arithmetic it would skip)
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        fFv[i] = links[i].fl_floss; fDv[i] = links[i].fl_D; fRv[i] = links[i].fl_R; fAv[i] = st.fa(i);
        const double xw = st.xs(i), jf = xw - fAv[i];
        const bool fr = fFv[i] > 0, below = jf <= -fRv[i], above = jf >= fRv[i];
        frows |= fr ? 1u << i : 0u;
        c_neg |= fr && below ? 1u << i : 0u;
        c_pos |= fr && !below && above ? 1u << i : 0u;
        c_act |= ((limrows >> i) & 1u) && lSv[i] * xw - lAv[i] < 0 ? 1u << i : 0u;
      }
      // which way each row would leave its zone (a quadratic row through the nearer boundary) and the row closest to doing so
      uint32_t c_side = 0;
      int c_m = 0;
      {
        double best = INFINITY;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
          const double jf = st.xs(i) - fAv[i], aj = fabs(jf);
          const bool fr = (frows >> i) & 1u;
          c_side |= jf > 0 ? 1u << i : 0u;
          const double margin = fr ? fabs(aj - fRv[i]) : INFINITY;
          c_m = margin < best ? i : c_m;
          best = fmin(best, margin);
        }
      }
      if (round > 0) { c_neg = b_neg; c_pos = b_pos; c_act = b_act; c_side = b_side; c_m = b_m; }
      // lane 0: the base.  Later rounds: lanes 1..7: row t - 1 moved to the zone next to its own; lanes 8..14: row t - 8 AND the row
      // closest to its boundary both moved (that row alone, to its far side, on the lane where the two coincide)
      auto move_row = [&](int j, bool far) {
        const uint32_t bit = 1u << j;
        const int zone0 = (c_neg & bit) ? 0 : ((c_pos & bit) ? 2 : 1);  // 0: f = +frictionloss, 1: quadratic, 2: f = -frictionloss
        const bool up = (c_side >> j) & 1u;
        const int near1 = zone0 == 1 ? (up ? 2 : 0) : 1, far1 = zone0 == 1 ? (up ? 0 : 2) : 2 - zone0;
        const int zone1 = far ? far1 : near1;
        c_neg = (c_neg & ~bit) | (zone1 == 0 ? bit : 0u);
        c_pos = (c_pos & ~bit) | (zone1 == 2 ? bit : 0u);
      };
      if (t == 0) cand_lane = true;
      else if (t < kTeamLanes - 1) {
        if (round == 0) {
          // (first round: every row to either of its two other zones -- one row moving is by far the commonest change)
          const int j = (t - 1) >> 1;
          if (j < NL && ((frows >> j) & 1u)) { move_row(j, (t - 1) & 1); cand_lane = true; }
        } else {
          const int j = t < 8 ? t - 1 : t - 8;
          if (j < NL && ((frows >> j) & 1u)) {
            if (t < 8) move_row(j, false);
            else if (j == c_m) move_row(j, true);
            else { move_row(j, false); move_row(c_m, false); }
            cand_lane = true;
          }
        }
      }
    }
    const uint32_t guess = helper_lane ? 0u : (FRIC ? c_act : act);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const bool on = (guess >> i) & 1u;
      const double dsolve = on ? lDv[i] : 0.0;
      H[tri(i, i)] += fma(wh, dgv[i], dsolve);
      double xi = fma(ws, sm[i], dsolve * (lSv[i] * lAv[i]));
      if (i == unit) xi = 1.0;
      if (T::GRIP && eq_helper && (i == NA || i == NA + 1)) xi = i == NA ? 1.0 : eqJ1;
      if constexpr (FRIC) {
        const bool fr = !helper_lane && ((frows >> i) & 1u), ng = (c_neg >> i) & 1u, ps = (c_pos >> i) & 1u;
        const double push = fr && ng ? fFv[i] : (fr && ps ? -fFv[i] : 0.0);  // the linear zones' constant force
        const double dq = fr && !ng && !ps ? fDv[i] : 0.0;                    // the quadratic zone's stiffness
        H[tri(i, i)] += dq;
        xi += push;
        xi += dq * fAv[i];
      }
      x[i] = xi;
    }
  }
  if constexpr (T::GRIP) {
    const double c0 = sk.grp_c0, c1 = sk.grp_c1, hg = h * gblock;
    const double e = has_eq && !helper_lane ? eqD : 0.0, g = helper_lane ? hg : 0.0;
    H[tri(NA, NA)] += fma(g * c0, c0, e);
    H[tri(NA + 1, NA)] += fma(g * c0, c1, e * eqJ1);
    H[tri(NA + 1, NA + 1)] += fma(g * c1, c1, e * eqJ1 * eqJ1);
    x[NA] += e * eqAref;
    x[NA + 1] += e * eqAref * eqJ1;
  }
  ldl_factor<NL>(H);
  ldl_solve<NL>(H, x);
  uint32_t now = 0;
  {
    double lAv[NL], lSv[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) { lAv[i] = st.limA(i); lSv[i] = st.limS(i); }
    sched_fence();
    // (every row's test, then the mask of the rows that exist: a test per row behind a scalar branch on its bit costs more)
#pragma unroll
    for (int i = 0; i < NL; ++i) now |= lSv[i] * x[i] - lAv[i] < 0 ? 1u << i : 0u;
    now &= limrows;
  }
  bool hit_guess = solver_lane && now == act;
  if constexpr (FRIC) {
    uint32_t nneg = 0, npos = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const double jf = x[i] - fAv[i];
      const bool fr = (frows >> i) & 1u, below = jf <= -fRv[i], above = jf >= fRv[i];
      nneg |= fr && below ? 1u << i : 0u;
      npos |= fr && !below && above ? 1u << i : 0u;
    }
    hit_guess = cand_lane && now == c_act && nneg == c_neg && npos == c_pos;
    // the next round's base: where lane 0's solution landed -- from the third round on, where the point half way between its
    // last two solutions lies (two zone sets that send the solve to each other enclose the minimiser between their solutions)
    uint32_t zn = 0, zp = 0, za = 0, zs = 0;
    int zm = 0;
    {
      double best = INFINITY;
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const double xm = round > 0 ? 0.5 * (x[i] + xprev[i]) : x[i];
        const double jf = xm - fAv[i], aj = fabs(jf);
        const bool fr = (frows >> i) & 1u, below = jf <= -fRv[i], above = jf >= fRv[i];
        zn |= fr && below ? 1u << i : 0u;
        zp |= fr && !below && above ? 1u << i : 0u;
        zs |= jf > 0 ? 1u << i : 0u;
        const double margin = fr ? fabs(aj - fRv[i]) : INFINITY;
        zm = margin < best ? i : zm;
        best = fmin(best, margin);
        za |= ((limrows >> i) & 1u) && st.limS(i) * xm - st.limA(i) < 0 ? 1u << i : 0u;
      }
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) xprev[i] = x[i];
    const int l0 = (int)(threadIdx.x & 48u) << 2;
    b_neg = (uint32_t)__builtin_amdgcn_ds_bpermute(l0, (int)zn);
    b_pos = (uint32_t)__builtin_amdgcn_ds_bpermute(l0, (int)zp);
    b_act = (uint32_t)__builtin_amdgcn_ds_bpermute(l0, (int)za);
    b_side = (uint32_t)__builtin_amdgcn_ds_bpermute(l0, (int)zs);
    b_m = __builtin_amdgcn_ds_bpermute(l0, zm);
  }
  winners = coupled ? 0u : team_ballot(hit_guess);
  rounds_used = round + 1;
#ifdef RCSH_PHASE_TIMING
  if constexpr (FRIC) {
    if (round == 0 && winners && t == 0) { const int wl_ = __ffs(winners) - 1; atomicAdd(&g_team_cycles[wl_ == 0 ? 24 : (((wl_ - 1) & 1) ? 26 : 25)], 1ull); }  // first-round winners: base / near move / far move (slots 24-26: the contact phase's marks in the kernels that have one)
    if (round == 0) { dbg_n0 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(threadIdx.x & 48u) << 2, (int)c_neg); dbg_p0 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(threadIdx.x & 48u) << 2, (int)c_pos); }
    if (winners && round > 0) {
      const int wl = ((int)(threadIdx.x & 48u) + __ffs(winners) - 1) << 2;
      const uint32_t wn = (uint32_t)__builtin_amdgcn_ds_bpermute(wl, (int)c_neg), wp = (uint32_t)__builtin_amdgcn_ds_bpermute(wl, (int)c_pos);
      const int nd = __popc((wn ^ dbg_n0) | (wp ^ dbg_p0));
      if (t == 0) atomicAdd(&g_team_cycles[55 + (nd > 3 ? 3 : nd)], 1ull);  // slots 55..58: rows whose zone differs between the first base and the winner: 0, 1, 2, 3+
    }
  }
#endif
  if (winners || coupled) break;
  }  // (rounds)
#ifdef RCSH_PHASE_TIMING
  if constexpr (FRIC) {
    const uint64_t r2 = __ballot(rounds_used >= 2), r3 = __ballot(rounds_used >= 3);
    if (t == 0) atomicAdd(&g_team_cycles[48 + (rounds_used > 3 ? 3 : rounds_used)], 1ull);        // team-substeps by rounds used: slots 49..51
    if ((threadIdx.x & 63) == 0) atomicAdd(&g_team_cycles[52 + (r3 ? 2 : (r2 ? 1 : 0))], 1ull);  // wavefront-substeps by passes: slots 52..54
  }
#endif
  (void)rounds_used;
#ifdef RCSH_PHASE_TIMING
  if (t == 0) {  // (every team of every workgroup) team-substeps / with a self-consistent candidate / wavefront-substeps that ran newton_rows
    atomicAdd(&g_team_cycles[61], 1ull);
    if (winners) atomicAdd(&g_team_cycles[62], 1ull);
    if ((threadIdx.x & 63) == 0 && __ballot(!coupled && winners == 0)) atomicAdd(&g_team_cycles[63], 1ull);
  }
#endif
  const bool superpose = !FRIC && winners != 0;  // uniform within the team
  if (coupled) {
    // the contact phase solved the coupled problem: nothing to do here
  } else if (winners) {
    const bool winner = t == __ffs(winners) - 1;
    if (winner || (!FRIC && helper_lane)) {
      // the winner publishes x, the helpers their solutions y (row t - 11 of Y)
      double* dst = winner ? &st.xs(0) : &st.Y(t - (kTeamLanes - 5), 0);
#pragma unroll
      for (int i = 0; i < NL; ++i) dst[i] = x[i];
    }
  } else {
    double xs[NL];
    TEAM_MARK(41)
    TEAM_COUNT(42)
    if constexpr (FRIC) {
      // (the rounds' last solution of lane 0 is as good an iterate as any: the serial routine picks up there)
      const int l0 = (int)(threadIdx.x & 48u);
#pragma unroll
      for (int i = 0; i < NL; ++i) xs[i] = lane_get(x[i], l0);
    }
    newton_rows<T, FRIC>(links, st, limrows, has_eq, eqD, eqAref, eqJ1, xs, FRIC);
    TEAM_MARK(40)
    if (t == 0) {
#pragma unroll
      for (int i = 0; i < NL; ++i) st.xs(i) = xs[i];
    }
  }
  team_sync();
  TEAM_MARK(6)

  // ---- implicitfast: (M - h dF/dqd) qacc = qfrc_smooth + qfrc_constraint
  double qacc = 0.0;
  if (superpose) {
    // qacc_t = y15[t] + fe y14[t] + sum_k c_k y(13-k)[t], the coefficients from the winner's x (every lane recomputes
    // the same four numbers; the joints of the limit rows are run-time indices into the LDS block)
    qacc = st.Y(4, tl);
    if constexpr (T::GRIP) if (has_eq) {
      const double fe = -eqD * (st.xs(NA) + eqJ1 * st.xs(NA + 1) - eqAref);
      qacc += fe * st.Y(3, tl);
    }
    const int idx[3] = {idx0, idx1, idx2};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (idx[k] < 0) continue;
      const int i = idx[k];
      const double sgn = st.limS(i), r = sgn * st.xs(i) - st.limA(i);
      if (r < 0) qacc += -sgn * st.limD(i) * r * st.Y(2 - k, tl);
    }
  } else {
    // one lane solves after the constraint solve (models with dry friction; fallback of the others, which must
    // factor A again because the slot's factor is gone)
    if (t == kTeamLanes - 1) {
      double xs[NL], rhs[NL], lDv[NL], lAv[NL], lSv[NL];
#pragma unroll
      for (int i = 0; i < NL; ++i) { xs[i] = st.xs(i); rhs[i] = st.smooth(i); lDv[i] = st.limD(i); lAv[i] = st.limA(i); lSv[i] = st.limS(i); }
      sched_fence();
      if (coupled) {
#pragma unroll
        for (int i = 0; i < NL; ++i) rhs[i] += st.fcon(i);
      }
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const double r = lSv[i] * xs[i] - lAv[i];
        rhs[i] -= !coupled && ((limrows >> i) & 1u) && r < 0 ? lSv[i] * lDv[i] * r : 0.0;
      }
      if constexpr (FRIC) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
          const double fF = links[i].fl_floss, fD = links[i].fl_D, fR = links[i].fl_R;
          const double jf = xs[i] - st.fa(i);
          const double f = jf <= -fR ? fF : (jf >= fR ? -fF : -fD * jf);
          rhs[i] += fF > 0 ? f : 0.0;
        }
      }
      if constexpr (T::GRIP) if (has_eq && !coupled) {
        const double fe = -eqD * (xs[NA] + eqJ1 * xs[NA + 1] - eqAref);
        rhs[NA] += fe;
        rhs[NA + 1] += fe * eqJ1;
      }
      if constexpr (!FRIC) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
#pragma unroll
          for (int j = 0; j <= i; ++j) H[tri(i, j)] = st.M(i, j);
          H[tri(i, i)] += st.dg(i);
        }
        if constexpr (T::GRIP) {
          const double c0 = sk.grp_c0, c1 = sk.grp_c1, hg = h * gblock;
          H[tri(NA, NA)] += hg * c0 * c0;
          H[tri(NA + 1, NA)] += hg * c0 * c1;
          H[tri(NA + 1, NA + 1)] += hg * c1 * c1;
        }
        ldl_factor<NL>(H);
      }
      ldl_solve<NL>(H, rhs);
#pragma unroll
      for (int i = 0; i < NL; ++i) st.qacc(i) = rhs[i];
    }
    team_sync();
    qacc = st.qacc(tl);
  }
  if (stepping && valid) {
    const double vn = qd + h * qacc;
    st.v(tl) = vn;
    st.q(tl) = q + h * vn;
  }
  TEAM_MARK(7)
}

#endif  // __HIP__

}  // namespace rcsh
