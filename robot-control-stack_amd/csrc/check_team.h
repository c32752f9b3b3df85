// check_team.h -- "is this environment in a contact nobody resolves?", asked once per stepping launch.
//
// The reference runs MuJoCo's collision pass in every substep (mj_step1, reference src/sim/sim.cpp:110) and whatever it finds
// is resolved by mj_step2 (sim.cpp:112) -- in EVERY mode, also in Sim::step(k), where no collision callback looks at the
// contact list (sim.cpp:108-115; the flags come from SimRobot::collision_callback, src/sim/SimRobot.cpp:172-182, which only
// step_until_convergence invokes).  The kernels resolve the contacts they are compiled for: none in the lean instantiations
// (the headline), robot <-> floor / cube in the CON ones, never robot <-> robot.  An environment whose geoms touch outside
// that set steps on as if nothing had happened -- the arm passes through the floor or itself -- and from that substep on its
// trajectory is not MuJoCo's.  This check makes that a reported fact instead of a premise: after the last substep of every
// stepping launch the position the NEXT mj_step1 would collide is tested exactly (same predicates as the flag-only detection
// of the convergence launches: sample points against the plane, MPR on every geom pair that survives two bounding tests), and
// a hit sets the environment's sticky kContactUnresolved flag (info byte 7, cleared by Sim::reset).
//
// Cost control, for a launch whose 17 substeps take ~240k cycles:
//  * it runs once per launch, after the state has been written back: the team's LDS block is free then and serves as its
//    workspace (no LDS of its own, nothing live in registers);
//  * geom pairs are grouped by BODY pair; one bounding-sphere test per body pair (36 for the FR3: three per lane) rules out
//    all geom pairs of two links that are far apart -- nearly all of them, for an arm in its workspace;
//  * a pair that is always within millimetres (links 5 and 7 wrap around the same wrist) is settled by ONE support query along
//    the direction that separated it last time, kept per environment in the state (Lay::SEP), as a collision library keeps
//    per-pair caches; any direction with negative support of A - B proves the hulls apart, so the cache decides cost, never
//    the result.
#pragma once
#include "contact_team.h"

namespace rcsh {

#if defined(__HIP__)

#ifdef RCSH_CHECK_DEBUG
__device__ int g_chk_dbg[64];  // development: [0] plane hits, [1] pair hits, [2..] pair indices of the first hits, [32] body pairs surviving, [33] geom pairs surviving
#endif
// "In contact" for this check: penetrating by more than a nanometre.  MuJoCo lists a contact as soon as dist < 0, but a pair that
// touches EXACTLY has no reproducible sign -- and the model has one at every reset: mj_resetData leaves the fingers at qpos 0,
// where the left and right fingertip pads meet face to face with a gap of 0.0; whether a collider then reports a penetration of
// 1e-17 m or none is round-off (it exerts no force either way).  The oracle-side restatement of this check uses the same bar.
constexpr double kCheckTouch = 1e-9;
constexpr int kCheckSep = 8;  // per environment: two remembered separating directions (pair index + 1, direction in geom 0's link frame)

// doubles of LDS workspace the check needs for an archetype with NL links
constexpr int check_work_doubles(int nl) { return 4 * (nl + 1) * 4 + 4 * kCheckSep + kSelfStage; }

// `frames`: LDS room for [4][NL][12] doubles (the link records' memory: the check is their last reader); `work`: LDS room for
// check_work_doubles(NL).  q: the lane's joint position (lane t < NL).  sep: the environment's SEP fields in the state ([8][n], at e).
// Every lane of the wavefront calls this; returns, on every lane of a team, whether the team's environment is in contact.
template <class T, class CollT>
RCSH_D bool unresolved_contact_check(const CheckTable& ck, const ContactTable& tab, const CollT& lc, const LinkRec* links, double* frames,
                                     double* work, double q, bool live, bool check_plane, double* sep, int n_env) {
  constexpr int NL = T::NL, NB = NL + 1;
  const int lane = threadIdx.x & 63, t = lane & (kTeamLanes - 1), team = lane / kTeamLanes;
  const bool valid = t < NL;
  const int tl = valid ? t : NL - 1;
  // ---- world frames of the links at the final qpos (what the next launch's first position stage will see)
  double R[9], p[3];
  {
    KinK kk;
    kk.load(links[tl]);
    link_local_frame(kk, q, R, p);
    scan_frames<T>(R, p);
  }
  __syncthreads();  // (the link records have been read: their memory becomes the frames')
  double* F = frames + 12 * NL * team;
  double* wsph = work + 4 * NB * team;
  double* slots = work + 4 * NB * 4 + kCheckSep * team;
  double* stage = work + 4 * NB * 4 + 4 * kCheckSep;
  if (valid) {
#pragma unroll
    for (int k = 0; k < 9; ++k) F[12 * t + k] = R[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) F[12 * t + 9 + k] = p[k];
    const double* bs = ck.bsphere[t + 1];
    double c[3];
    const double b3[3] = {bs[0], bs[1], bs[2]};
    mulmv(R, b3, c);
    wsph[4 * (t + 1) + 0] = c[0] + p[0]; wsph[4 * (t + 1) + 1] = c[1] + p[1]; wsph[4 * (t + 1) + 2] = c[2] + p[2];
    wsph[4 * (t + 1) + 3] = bs[3];
  } else if (t == NL) {
#pragma unroll
    for (int k = 0; k < 4; ++k) wsph[k] = ck.bsphere[0][k];
  }
  if (t < kCheckSep) slots[t] = live ? sep[(size_t)t * n_env] : 0.0;
  __syncthreads();
  bool mine = false;
  // ---- the floor against the sample points of the link's collision geoms (as the DET launches test it)
  if (check_plane && ck.plane_points && valid && live) {
    const double* nrm = lc.plane_n;
    const double a[3] = {R[0] * nrm[0] + R[3] * nrm[1] + R[6] * nrm[2], R[1] * nrm[0] + R[4] * nrm[1] + R[7] * nrm[2],
                         R[2] * nrm[0] + R[5] * nrm[1] + R[8] * nrm[2]};
    const double b = dot3(nrm, p) - lc.plane_d;
    const double* sph = lc.link_sphere[t];
    if (b + a[0] * sph[0] + a[1] * sph[1] + a[2] * sph[2] - sph[3] < -kCheckTouch) {
      for (int k = lc.link_adr[t]; k < lc.link_adr[t + 1]; ++k) {
        const double* v = lc.xyzr + 4 * (size_t)k;
        if (b + a[0] * v[0] + a[1] * v[1] + a[2] * v[2] - v[3] < -kCheckTouch) {
          mine = true;
#ifdef RCSH_CHECK_DEBUG
          atomicAdd(&g_chk_dbg[0], 1);
#endif
        }
      }
    }
  }
  // ---- body pairs: bounding spheres
  uint32_t bmask = 0;
  if (live) {
    for (int j = 0, i = t; i < ck.nbpair; ++j, i += kTeamLanes) {
      const CheckBodyPair bp = ck.bpairs[i];
      const double* sa = wsph + 4 * bp.ba;
      const double* sb = wsph + 4 * bp.bb;
      const double d[3] = {sa[0] - sb[0], sa[1] - sb[1], sa[2] - sb[2]}, rs = sa[3] + sb[3];
      if (dot3(d, d) <= rs * rs) {
        bmask |= 1u << j;
#ifdef RCSH_CHECK_DEBUG
        atomicAdd(&g_chk_dbg[32], 1);
#endif
      }
    }
  }
  for (uint64_t pend = __ballot(bmask != 0); pend; pend = __ballot(bmask != 0)) {
    const int src = __ffsll((long long)pend) - 1;  // wave-uniform
    const uint32_t sm = (uint32_t)__builtin_amdgcn_readlane((int)bmask, src);
    const int j = __ffs((int)sm) - 1, t0 = src & (kTeamLanes - 1);
    // the teams in which this body pair survived (their lane t0 holds bit j) take it together
    const bool holder = t == t0 && ((bmask >> j) & 1u);
    const bool take = team_ballot(holder) != 0;
    if (holder) bmask &= ~(1u << j);
    const CheckBodyPair bp = ck.bpairs[t0 + kTeamLanes * j];
    // its geom pairs, one per lane and round: bounding spheres, then the oriented boxes (all from the pair record)
    uint32_t cmask = 0;
    for (int u = 0, i = bp.adr + t; i < bp.adr + bp.num; ++u, i += kTeamLanes) {
      if (!take) continue;
      const SelfPair& pr = ck.pairs[i];
      double Ra[9], Rb[9], ca[3], cb[3];
      self_box_world(F, pr.l0, pr.c0, pr.rot0, ca, Ra);
      self_box_world(F, pr.l1, pr.c1, pr.rot1, cb, Rb);
      const double d[3] = {ca[0] - cb[0], ca[1] - cb[1], ca[2] - cb[2]}, rs = pr.r0 + pr.r1;
      if (dot3(d, d) > rs * rs) continue;
      if (!obb_disjoint(Ra, ca, pr.h0, Rb, cb, pr.h1)) {
        cmask |= 1u << u;
#ifdef RCSH_CHECK_DEBUG
        atomicAdd(&g_chk_dbg[33], 1);
#endif
      }
    }
    // narrow phase: a surviving geom pair has its hulls staged by the whole wavefront, once for all the teams it survived in;
    // each of those teams runs the refinement on its 16 lanes, which share the vertex scans of the support queries
    for (uint64_t pc = __ballot(cmask != 0); pc; pc = __ballot(cmask != 0)) {
      const int src2 = __ffsll((long long)pc) - 1;
      const uint32_t sm2 = (uint32_t)__builtin_amdgcn_readlane((int)cmask, src2);
      const int u = __ffs((int)sm2) - 1, t1 = src2 & (kTeamLanes - 1);
      const bool holder2 = t == t1 && ((cmask >> u) & 1u);
      const bool take2 = team_ballot(holder2) != 0;
      if (holder2) cmask &= ~(1u << u);
      const int pidx = bp.adr + t1 + kTeamLanes * u;
      const SelfPair& pr = ck.pairs[pidx];
      const ContactGeom& a = tab.geoms[pr.g0];
      const ContactGeom& b = tab.geoms[pr.g1];
      const int na = 3 * a.vert_num, nb = 3 * b.vert_num;
      {
        const double* va = tab.verts + 3 * (size_t)a.vert_adr;
        const double* vb = tab.verts + 3 * (size_t)b.vert_adr;
        for (int k = lane; k < na; k += 64) stage[k] = va[k];
        for (int k = lane; k < nb; k += 64) stage[na + k] = vb[k];
        stage_fence();  // (LDS traffic of one wavefront is ordered)
      }
      if (take2) {
        double Ra[9], pa[3], Rb[9], pb[3];
        self_geom_world(a, F, Ra, pa);
        self_geom_world(b, F, Rb, pb);
        Shape A = make_shape(a.type == 7 ? 0 : a.type == 6 ? 1 : 2, pa, Ra, a.size, stage, a.vert_num);
        Shape B = make_shape(b.type == 7 ? 0 : b.type == 6 ? 1 : 2, pb, Rb, b.size, stage + na, b.vert_num);
        if (a.type == 7) { mulmv(Ra, a.center, A.center); A.center[0] += pa[0]; A.center[1] += pa[1]; A.center[2] += pa[2]; }
        else { A.center[0] = pa[0]; A.center[1] = pa[1]; A.center[2] = pa[2]; }
        if (b.type == 7) { mulmv(Rb, b.center, B.center); B.center[0] += pb[0]; B.center[1] += pb[1]; B.center[2] += pb[2]; }
        else { B.center[0] = pb[0]; B.center[1] = pb[1]; B.center[2] = pb[2]; }
        double* slot = slots + 4 * (pidx & 1);
        bool apart = false;
        double LR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        if (pr.l0 >= 0) {
#pragma unroll
          for (int k = 0; k < 9; ++k) LR[k] = F[12 * pr.l0 + k];
        }
        if (slot[0] == (double)(pidx + 1)) {
          const double dl[3] = {slot[1], slot[2], slot[3]};
          double dw[3];
          mulmv(LR, dl, dw);
          MprPt s;
          mpr_support<true>(A, B, dw, s);
          apart = dot3(s.v, dw) < 0;  // the support of A - B along the remembered direction is still negative: apart
        }
        if (!apart) {
          double dir[3], depth = 0.0;
          if (mpr_penetration<true, kMprDepth>(A, B, &depth, dir, nullptr) && depth > kCheckTouch) {
            mine = true;
#ifdef RCSH_CHECK_DEBUG
            if (t == 0) { const int k = atomicAdd(&g_chk_dbg[1], 1); if (k < 28) g_chk_dbg[2 + k] = pidx; }
#endif
          }
          else if (dot3(dir, dir) > 0.5) {  // (a unit separating direction came back)
            double dl[3];
            mulTv(LR, dir, dl);
            stage_fence();
            slot[0] = (double)(pidx + 1); slot[1] = dl[0]; slot[2] = dl[1]; slot[3] = dl[2];
          }
        }
      }
      stage_fence();
    }
  }
  __syncthreads();
  if (t < kCheckSep && live) sep[(size_t)t * n_env] = slots[t];
  return team_ballot(mine) != 0;
}

#endif  // __HIP__

}  // namespace rcsh
