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

#ifdef RCSH_WAVE_TIMES
__device__ unsigned long long g_wave_times[4][4096];  // sim_kernels.h: the lean launch's wavefronts, 100 MHz clock
#endif
#ifdef RCSH_CHECK_TAIL
// development: how long the check takes per wavefront -- [0] sum of cycles, [1] wavefronts, [2] the longest, [3] wavefronts that left at the slack
// test, then what the longest one did: [4] narrow-phase rounds, [5] Gilbert runs, [6] support queries at the start frames, [7] full refinements, [8] box rounds
__device__ unsigned long long g_chk_tail[16];
__device__ unsigned long long g_chk_hist[64];  // [0..15] wavefronts by total cycles (8k bins), [16..31] by narrow-phase cycles (4k bins), [32..47] by cycles before the narrow phase (4k bins), [48..63] ENVIRONMENTS of the lean launch by the narrow-phase rounds their team took part in (0..15+)
#define TAIL_COUNT(i) { if ((int)(threadIdx.x & 63) == __ffsll((long long)__ballot(true)) - 1) atomicAdd(&tail_sh_[i], 1u); }
#define TAIL_END(early) { if ((threadIdx.x & 63) == 0) { const unsigned long long dt_ = __builtin_readcyclecounter() - tail_t0_; atomicAdd(&g_chk_tail[0], dt_); atomicAdd(&g_chk_tail[1], 1ull); \
    { unsigned long long b_ = dt_ / 8192; atomicAdd(&g_chk_hist[b_ > 15 ? 15 : b_], 1ull); b_ = tail_nar_ / 4096; atomicAdd(&g_chk_hist[16 + (b_ > 15 ? 15 : b_)], 1ull); b_ = tail_pre_ / 4096; atomicAdd(&g_chk_hist[32 + (b_ > 15 ? 15 : b_)], 1ull); } \
    if (early) atomicAdd(&g_chk_tail[3], 1ull); if (atomicMax(&g_chk_tail[2], dt_) < dt_) { for (int k_ = 0; k_ < 5; ++k_) g_chk_tail[4 + k_] = tail_sh_[k_]; g_chk_tail[9] = tail_stage_; g_chk_tail[10] = tail_miss_; g_chk_tail[11] = tail_nar_; g_chk_tail[12] = tail_pre_; g_chk_tail[13] = tail_sh_[5]; g_chk_tail[14] = tail_sh_[6]; g_chk_tail[15] = tail_sh_[7]; } } }
#else
#define TAIL_COUNT(i)
#define TAIL_END(early)
#endif
#ifdef RCSH_CHECK_DEBUG
__device__ int g_chk_dbg[64];
__device__ double g_chk_dbgf[128];  // certificate failures of the narrow phase: (pair, margin, gap at the end, gap at the start) x 32
__device__ unsigned long long g_chk_cyc[16];
#define CHK_MARK(i) if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); g_chk_cyc[i] += now_ - chk_t0_; chk_t0_ = now_; }
#else
#define CHK_MARK(i)  // development: [0] plane hits, [1] pair hits, [2..] pair indices of the first hits, [32] body pairs surviving, [33] geom pairs surviving
#endif
// "In contact" for this check: penetrating by more than a nanometre.  MuJoCo lists a contact as soon as dist < 0, but a pair that
// touches EXACTLY has no reproducible sign -- and the model has one at every reset: mj_resetData leaves the fingers at qpos 0,
// where the left and right fingertip pads meet face to face with a gap of 0.0; whether a collider then reports a penetration of
// 1e-17 m or none is round-off (it exerts no force either way).  The oracle-side restatement of this check uses the same bar.
constexpr double kCheckTouch = 1e-9;
constexpr int kCheckRoundBudget = 12;  // narrow-phase rounds a certificate may take (certifying mode; unresolved_contact_check: THE BUDGET)
constexpr int kCheckSep = 16;  // per environment: FOUR remembered separating directions (pair index + 1, direction in geom 0's link frame) -- two
                                // through round 5's last session: a folded arm keeps three or four pairs near, the xArm7's gripper linkage more, and a pair
                                // without its direction costs Gilbert steps or a full refinement in every launch
constexpr int kCheckSlots = kCheckSep / 4;
static_assert(kCheckSep <= kTeamLanes && kCheckSep % 4 == 0, "a lane of the team per word");

// LDS workspace of the check (doubles): per team the world boxes of the geoms ([ngeom][12]: centre, axes) -- later overlaid by the
// stage of the one pair of hulls the narrow phase works on --, the remembered directions, the two geom records of that pair
constexpr int kCheckBox = 12 * kMaxCGeom;
constexpr int kCheckGeomWords = (int)(sizeof(ContactGeom) / 8);
static_assert(sizeof(ContactGeom) % 8 == 0 && 2 * kCheckGeomWords <= 64, "a lane per word of the narrow phase's two geom records");
constexpr int check_work_doubles(int) { return 4 * kCheckBox + 64 + 4 * 12; }  // (+ the teams' joint travel, certifying mode)
// certifying mode: the teams' tables of how far that travel can have moved a geom on link l relative to the frame of an ancestor link
// c, moved[l][c + 1] -- room of its own (`mv`), which the caller finds where its kernel has some to spare
// -- plus the links' world frames at the launch's START position ([4][nl][12]) and the geoms' boxes in their links' frames ([kMaxCGeom][12])
// (two such tables -- one per kind of margin, see unresolved_contact_check -- in single precision, rounded up) and the teams' second travel vector
// ... and the fingers' slide axes (world) with the travel of the gripper's opening / common shift: [4][12]
constexpr int check_mv_doubles(int nl) { return 4 * nl * (nl + 1) + 4 * 12 * nl + 12 * kMaxCGeom + 4 * 12 + 4 * 12; }
static_assert(4 * kCheckBox >= kSelfStage, "the hull stage overlays the world boxes");
constexpr int kCheckPer = kMaxCheckPairs / kTeamLanes;
constexpr int kCheckTrips = (3 * 152 + 63) / 64;  // vertex words per lane and hull (a hull has at most 152 vertices)

// obb_disjoint for this check: true when the two oriented boxes are apart OR overlap by at most `touch` along one of their six
// face normals (then whatever they contain overlaps by at most that much).  For two BOXES this is the narrow phase already: the
// fingertip pads of a closed gripper -- two dozen box pairs that touch exactly -- never reach the portal refinement.
// `margin` > 0 (certifying mode): true only when one of the six face normals separates the boxes by MORE than the margin.
RCSH_D bool obb_apart_or_touching(const double* Ra, const double* ca, const double* ha, const double* Rb, const double* cb, const double* hb, double touch,
                                  double margin = 0.0) {
  double C[9], A[9], tv[3];
  const double d[3] = {cb[0] - ca[0], cb[1] - ca[1], cb[2] - ca[2]};
  mulTv(Ra, d, tv);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      C[3 * i + j] = Ra[i] * Rb[j] + Ra[3 + i] * Rb[3 + j] + Ra[6 + i] * Rb[6 + j];
      A[3 * i + j] = fabs(C[3 * i + j]) + 1e-9;
    }
  double sep = -INFINITY;
#pragma unroll
  for (int i = 0; i < 3; ++i) sep = fmax(sep, fabs(tv[i]) - (ha[i] + hb[0] * A[3 * i] + hb[1] * A[3 * i + 1] + hb[2] * A[3 * i + 2]));
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const double tw = tv[0] * C[j] + tv[1] * C[3 + j] + tv[2] * C[6 + j];
    sep = fmax(sep, fabs(tw) - (hb[j] + ha[0] * A[j] + ha[1] * A[3 + j] + ha[2] * A[6 + j]));
  }
  if (margin > 0.0) return sep > margin - touch;
  if (sep > -touch) return true;
  bool apart = false;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      const double ra = ha[i1] * A[3 * i2 + j] + ha[i2] * A[3 * i1 + j];
      const double rb = hb[j1] * A[3 * i + j2] + hb[j2] * A[3 * i + j1];
      apart = apart || fabs(tv[i2] * C[3 * i1 + j] - tv[i1] * C[3 * i2 + j]) - (ra + rb) > 0;  // (a cross axis is not a unit vector: strict separation only)
    }
  return apart;
}

// The largest separation of two oriented boxes along one of their six face normals (negative: they overlap along all six): a lower
// bound of the distance between the boxes, and of whatever they contain.
RCSH_D double obb_face_sep(const double* Ra, const double* ca, const double* ha, const double* Rb, const double* cb, const double* hb) {
  double C[9], A[9], tv[3];
  const double d[3] = {cb[0] - ca[0], cb[1] - ca[1], cb[2] - ca[2]};
  mulTv(Ra, d, tv);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      C[3 * i + j] = Ra[i] * Rb[j] + Ra[3 + i] * Rb[3 + j] + Ra[6 + i] * Rb[6 + j];
      A[3 * i + j] = fabs(C[3 * i + j]) + 1e-9;
    }
  double sep = -INFINITY;
#pragma unroll
  for (int i = 0; i < 3; ++i) sep = fmax(sep, fabs(tv[i]) - (ha[i] + hb[0] * A[3 * i] + hb[1] * A[3 * i + 1] + hb[2] * A[3 * i + 2]));
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const double tw = tv[0] * C[j] + tv[1] * C[3 + j] + tv[2] * C[6 + j];
    sep = fmax(sep, fabs(tw) - (hb[j] + ha[0] * A[j] + ha[1] * A[3 + j] + ha[2] * A[6 + j]));
  }
  return sep;
}

// Two boxes on the two FINGERS of a gripper: only the fingers' slides move them relative to each other -- a pure translation, along
// known axes, and the separation along a face normal n changes by exactly n . (that translation).  The largest, over the six face
// normals, of (separation along n) - (the most the slides' travel can have taken from it): positive certifies the launch.  The travel
// that counts is the OPENING's (q0 + q1 for mirrored axes): after a reset, and whenever the arm accelerates, two shut fingers shift
// TOGETHER by 100-400 um a step against their soft coupling -- the pads stay 30 um apart, each finger's own travel is ten times that, and
// charged with it half the batch failed the certificate for a dozen steps after every reset.
// the six face-normal separations of two oriented boxes, and |n . e| for two vectors e (world) per normal
RCSH_D void obb_face_seps6(const double* Ra, const double* ca, const double* ha, const double* Rb, const double* cb, const double* hb, double* sep,
                           const double* ep, const double* em, double* np6, double* nm6, double* cs6 = nullptr) {
  double C[9], A[9], tv[3];
  const double d[3] = {cb[0] - ca[0], cb[1] - ca[1], cb[2] - ca[2]};
  mulTv(Ra, d, tv);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      C[3 * i + j] = Ra[i] * Rb[j] + Ra[3 + i] * Rb[3 + j] + Ra[6 + i] * Rb[6 + j];
      A[3 * i + j] = fabs(C[3 * i + j]) + 1e-9;
    }
#pragma unroll
  for (int i = 0; i < 3; ++i) sep[i] = fabs(tv[i]) - (ha[i] + hb[0] * A[3 * i] + hb[1] * A[3 * i + 1] + hb[2] * A[3 * i + 2]);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const double tw = tv[0] * C[j] + tv[1] * C[3 + j] + tv[2] * C[6 + j];
    sep[3 + j] = fabs(tw) - (hb[j] + ha[0] * A[j] + ha[1] * A[3 + j] + ha[2] * A[6 + j]);
  }
  if (ep) {
    double u0[3], u1[3], w0[3], w1[3];
    mulTv(Ra, ep, u0); mulTv(Ra, em, u1); mulTv(Rb, ep, w0); mulTv(Rb, em, w1);
#pragma unroll
    for (int i = 0; i < 3; ++i) { np6[i] = fabs(u0[i]); nm6[i] = fabs(u1[i]); np6[3 + i] = fabs(w0[i]); nm6[3 + i] = fabs(w1[i]); }
    if (cs6) {
      // what a unit of translation of B against A along ep adds to each separation: the normal's component of ep, the normal pointing
      // from A to B (|x| >= sgn(x0) x: the straight line through the separation now is a lower bound of it after any translation)
#pragma unroll
      for (int i = 0; i < 3; ++i) cs6[i] = tv[i] >= 0.0 ? u0[i] : -u0[i];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const double tw = tv[0] * C[j] + tv[1] * C[3 + j] + tv[2] * C[6 + j];
        cs6[3 + j] = tw >= 0.0 ? w0[j] : -w0[j];
      }
    }
  }
}
// sl: the two fingers' slide axes (world, at the launch's end) a0 = sl[0..2], a1 = sl[4..6]; how far the opening o+ = q0 + q1 / the common
// shift o- = q0 - q1 have been from their values at the launch's end: sl[3], sl[7]; their effective paths: sl[8], sl[9].  The geoms'
// relative translation is o+ (a1 - a0) / 2 - o- (a1 + a0) / 2.  Certified: along one of the six face normals either the separation at
// the launch's end exceeds what the slides can have taken from it, or the separations at its two ends together exceed their effective
// path along it (touching at an end, to within `touch`, is no contact).  Double precision throughout and the slides' own axes, no lever:
// fingers that open from pads touching EXACTLY -- every reset leaves them so -- end the first step with a gap that EQUALS the opening's
// travel; a margin rounded up by one part in a million failed half the batch there.
// ... or -- the ENVELOPE -- no single normal does, but at every opening the launch went through some normal separates the boxes:
// the geoms' relative motion is a translation along ep by the change of the opening (and along em by that of the common shift, charged
// as a margin), so each normal's separation is a straight line in the opening -- its value at the launch's end, slope cs6 --, the
// largest of the six a convex function of it, and its minimum over the interval of openings the launch covered (sl[10], sl[11]: how
// far below / above its final value the opening has been) sits at an end of the interval or where two lines cross.  (Adjacent pads of
// the two fingertips, tilted a little against the slides: their facing edges touch -- a separation of 0.0 along the finger at rest, and
// what the slides' jitter takes from it -- while tens of micrometres separate them ACROSS the gap whenever it does; in
// step_until_convergence, whose launches are hundreds of substeps long, no single normal certified them and 1900 of 4096 environments
// stayed on the contact-resolving launch without ever meeting a contact.)  sAB: +1 when A hangs on the first finger's slide (a0) and B on
// the second's, -1 the other way round.
RCSH_D bool finger_boxes_certified(const double* Ra, const double* ca, const double* ha, const double* Rb, const double* cb, const double* hb,
                                   const double* Ra0, const double* ca0, const double* Rb0, const double* cb0, const double* sl, double touch, double sAB) {
  const double ep[3] = {0.5 * (sl[4] - sl[0]), 0.5 * (sl[5] - sl[1]), 0.5 * (sl[6] - sl[2])}, em[3] = {0.5 * (sl[4] + sl[0]), 0.5 * (sl[5] + sl[1]), 0.5 * (sl[6] + sl[2])};
  double s1[6], s0[6], np6[6], nm6[6], cs6[6];
  obb_face_seps6(Ra, ca, ha, Rb, cb, hb, s1, ep, em, np6, nm6, cs6);
  obb_face_seps6(Ra0, ca0, ha, Rb0, cb0, hb, s0, nullptr, nullptr, nullptr, nullptr);
  bool ok = false;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const double mD = np6[k] * sl[3] + nm6[k] * sl[7], mP = np6[k] * sl[8] + nm6[k] * sl[9];
    ok = ok || s1[k] - mD > -touch || (s0[k] > -touch && s1[k] > -touch && s0[k] + s1[k] > mP - 2.0 * touch);
  }
  if (ok) return true;
  const double lo = sl[10], hi = sl[11];
  if (!(lo > -1e300 && hi < 1e300 && sl[7] < 1e300)) return false;
  double b[6], c[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) { b[k] = s1[k] - nm6[k] * sl[7]; c[k] = sAB * cs6[k]; }
  auto env = [&](double x) {
    double f = b[0] + c[0] * x;
#pragma unroll
    for (int k = 1; k < 6; ++k) f = fmax(f, b[k] + c[k] * x);
    return f;
  };
  double fmin_ = fmin(env(lo), env(hi));
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = i + 1; j < 6; ++j) {
      const double dc = c[i] - c[j];
      if (fabs(dc) > 1e-12) {
        const double x = (b[j] - b[i]) / dc;
        if (x > lo && x < hi) fmin_ = fmin(fmin_, env(x));
      }
    }
  return fmin_ > -touch;
}

// The support VALUE of a shape along a direction -- the largest x . dir over its points -- computed by the 16 lanes of a team on
// vertices staged in LDS.  The check's certificates read nothing else of a support query: the support of A - B along d is
// h_A(d) + h_B(-d), its negative the gap d proves.  (shape_support also finds WHICH vertex, lowest index among ties, in a second pass
// over the hull, and fetches and transforms it: what the refinement that builds contact points needs, twice the cost.)
RCSH_D double shape_support_value(const Shape& s, const double* dir) {
  double l[3];
  mulTv(s.R, dir, l);
  double h;
  if (s.type == 0) {
    const double* verts = in_lds(s.verts);
    const int t = threadIdx.x & (kTeamLanes - 1);
    double best = -INFINITY;
    for (int i0 = t; i0 < s.nvert; i0 += 4 * kTeamLanes) {
      double x[4][3];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * kTeamLanes < s.nvert ? i0 + u * kTeamLanes : i0;
        x[u][0] = verts[3 * i]; x[u][1] = verts[3 * i + 1]; x[u][2] = verts[3 * i + 2];
      }
      sched_fence();
#pragma unroll
      for (int u = 0; u < 4; ++u) best = fmax(best, x[u][0] * l[0] + x[u][1] * l[1] + x[u][2] * l[2]);
    }
    best = fmax(best, row_rotate<8>(best));
    best = fmax(best, row_rotate<4>(best));
    best = fmax(best, row_rotate<2>(best));
    best = fmax(best, row_rotate<1>(best));
    h = best;
  } else if (s.type == 1) {
    h = fabs(l[0]) * s.size[0] + fabs(l[1]) * s.size[1] + fabs(l[2]) * s.size[2];
  } else {
    h = fabs(l[2]) * s.size[1] + s.size[0] * sqrt(dot3(l, l));
  }
  return h + dot3(s.p, dir);
}
// the gap a unit direction d proves between A and B (negative: none): minus the support of A - B along d
RCSH_D double support_gap(const Shape& A, const Shape& B, const double* d) {
  const double nd[3] = {-d[0], -d[1], -d[2]};
  return -(shape_support_value(A, d) + shape_support_value(B, nd));
}

// A remembered direction's slot: [0] = (pair index + 1) + 1024 (g0 + 32 g1) -- zero: empty --, [1..3] the direction in the frame of
// geom 0's link.  The geoms ride along so that the NEXT launch can ask for their records and vertices before it knows anything else.
RCSH_D double check_slot_key(int pidx, int g0, int g1) { return (double)((pidx + 1) + 1024 * (g0 + 32 * g1)); }

// Everything the check reads from memory, asked for BEFORE the launch's epilogue (the leader lane's observation arithmetic: ~10k
// cycles during which the other 60 lanes and the memory pipeline idle): the lane's pair entries, the boxes of "its" two geoms, and --
// a guess -- the records and hull vertices of the pair whose remembered direction sits in the first live team's first used slot
// (links 5 and 7, nearly always).  After the epilogue the check computes from registers and LDS alone.
struct CheckPrefetch {
  CheckEntry ent[kCheckPer];
  double grec[2][12];
  int guess_key, guess_g0, guess_g1;  // wave-uniform; key 0: no guess
  double gword;                       // this lane's word of the guessed pair's two ContactGeom records
  double va[kCheckTrips], vb[kCheckTrips];
  float lev[12];                      // CheckTable::lev[j][t]: what a radian / metre of joint j does to a point of a geom on the lane's link
  float rem[kCheckPer], remL;         // what is left of the lane's pairs' gaps / of its link's height above the floor (CheckTable::slack)
};
// slack_env: the environment's record of CheckTable::slack (null: none kept -- every pair is looked at)
RCSH_D void check_prefetch(const CheckTable& ck, const ContactTable& tab, double sep_in, bool live, CheckPrefetch& pf, const float* slack_env) {
  const int lane = threadIdx.x & 63, t = lane & (kTeamLanes - 1);
  const int npair = ck.npair, ngeom = ck.ngeom;
#pragma unroll
  for (int j = 0; j < kCheckPer; ++j) {
    const int i = t + kTeamLanes * j;
    // (a scene with a floor but no admitted geom pair -- a single collision geom -- has no entry table at all: advisor, round 4)
    pf.ent[j] = npair > 0 ? ck.ent[i < npair ? i : 0] : CheckEntry{0u, 0.0f};
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int g = t + kTeamLanes * u;
    const double* src = reinterpret_cast<const double*>(&ck.geoms[g < ngeom ? g : 0]);
#pragma unroll
    for (int k = 0; k < 12; ++k) pf.grec[u][k] = ngeom > 0 && ck.geoms ? src[k] : 0.0;
  }
#pragma unroll
  for (int j = 0; j < 12; ++j) pf.lev[j] = ck.lev && t < 12 ? ck.lev[12 * j + t] : 0.0f;
#pragma unroll
  for (int j = 0; j < kCheckPer; ++j) pf.rem[j] = slack_env && t + kTeamLanes * j < npair ? slack_env[t + kTeamLanes * j] : 0.0f;
  pf.remL = slack_env ? slack_env[kSlackLink + t] : 0.0f;
  // the guess: slot 0's key of the first live team, else its slot 1's (lanes 0 and 4 of the team hold them)
  const uint64_t lv = __ballot(live && t == 0);
  pf.guess_key = 0; pf.guess_g0 = 0; pf.guess_g1 = 0;
  pf.gword = 0.0;
#pragma unroll
  for (int u = 0; u < kCheckTrips; ++u) { pf.va[u] = 0.0; pf.vb[u] = 0.0; }
  if (lv && npair > 0) {
    const int l0 = __ffsll((long long)lv) - 1;
    const double k0 = wave_read(sep_in, l0), k1 = wave_read(sep_in, l0 + 4);
    const int key = (int)(k0 != 0.0 ? k0 : k1);
    const int pid = key & 1023, gg = key >> 10, g0 = gg & 31, g1 = gg >> 5;
    if (pid >= 1 && pid <= npair && g0 < ngeom && g1 < ngeom) {
      pf.guess_key = key; pf.guess_g0 = g0; pf.guess_g1 = g1;
      const int w = lane & 31;
      const double* rec = reinterpret_cast<const double*>(&tab.geoms[lane < 32 ? g0 : g1]);
      pf.gword = rec[w < kCheckGeomWords ? w : 0];
      const int na = 3 * ck.gvert[g0][1], nb = 3 * ck.gvert[g1][1];
      const double* va = tab.verts + 3 * (size_t)ck.gvert[g0][0];
      const double* vb = tab.verts + 3 * (size_t)ck.gvert[g1][0];
#pragma unroll
      for (int u = 0; u < kCheckTrips; ++u) {
        const int k = lane + 64 * u;
        pf.va[u] = va[k < na ? k : 0];
        pf.vb[u] = vb[k < nb ? k : 0];
      }
    }
  }
}

// `frames`: LDS room for [4][NL][12] doubles (the link records' memory: the check is their last reader); `work`: LDS room for
// check_work_doubles(NL); `mv`: for check_mv_doubles(NL); dend / psum: how far the lane's joint has been from where the launch ended, at
// most / its effective path over the launch (see below; 0: it did not move -- or the check of the final position alone is asked for)
// and q0 its position when the launch began.  slack_env: the environment's record of CheckTable::slack (null: none), use_slack: the
// gaps it holds are valid lower bounds for a position of this launch (see "the slack" below: the one it began on -- the lean launch --
// or the one its last substep began on -- the contact-resolving launch, whose collision passes keep the record), keep_slack: the
// pairs' gaps this check ends with are written back (1: the lean launch; 2: the contact-resolving launch -- both unless the
// environment is found in contact; 2 also stops looking once every environment of the wavefront is: the answer is all it is asked for).
// known_hit (team-uniform): the caller knows the answer already -- a substep of this launch resolved a contact.  dlo / dhi: how far
// below / above its value at the launch's end the lane's joint has been (<= 0, >= 0; dend is the larger of the two sizes).  q: the lane's joint position (lane t < NL).  sep: the environment's SEP fields in the state ([8][n], at e).
// Every lane of the wavefront calls this; returns, on every lane of a team, whether the team's environment is in contact.
template <class T, class CollT>
RCSH_D bool unresolved_contact_check(const CheckTable& ck, const ContactTable& tab, const CollT& lc, const LinkRec* links, double* frames,
                                     double* work, double q, bool live, bool check_plane, double sep_in, double* sep, int n_env, const CheckPrefetch& pf,
                                     double dend, double psum, double* mv, double q0, float* slack_env, bool use_slack, int keep_slack, bool known_hit = false,
                                     double dlo = 0.0, double dhi = 0.0) {
  constexpr int NL = T::NL;
  const int lane = threadIdx.x & 63, t = lane & (kTeamLanes - 1), team = lane / kTeamLanes;
  const bool valid = t < NL;
  const int tl = valid ? t : NL - 1;
  const int npair = ck.npair, ngeom = ck.ngeom;
#ifdef RCSH_CHECK_DEBUG
  unsigned long long chk_t0_ = __builtin_readcyclecounter();
#endif
#ifdef RCSH_CHECK_TAIL
  const unsigned long long tail_t0_ = __builtin_readcyclecounter();
  __shared__ unsigned tail_sh_[8];
  if ((threadIdx.x & 63) < 8) tail_sh_[threadIdx.x & 63] = 0;
  __syncthreads();
  unsigned long long tail_nar_ = 0, tail_pre_ = 0, tail_stage_ = 0, tail_miss_ = 0, tail_rounds_ = 0;
#endif
  double* wbox = work + kCheckBox * team;
  // (the remembered directions stay in the lanes that loaded them -- lane t of a team holds word t of its four slots; reads and updates go
  // across the team's lanes: no LDS, which the lean detection kernel has none of to spare)
  double sepw = sep_in;
  const int tbase = lane & ~(kTeamLanes - 1);
  double* gstage = work + 4 * kCheckBox;
  double* travel = gstage + 64 + 12 * team;  // travel of the team's joints over the launch (certifying mode; zeros otherwise)
  double* stage = work;  // (overlays the world boxes once the broad phase is through)
  double* travelP = mv + 4 * NL * (NL + 1) + 4 * 12 * NL + 12 * kMaxCGeom + 12 * team;
  // (the two fingers' lanes carry the travel of the gripper's opening q0 + q1 / of the fingers' common shift q0 - q1 -- sim_kernels.h --:
  // either finger's own travel is at most half their sum)
  double dend_j = dend, psum_j = psum;
  if (T::GRIP) {
    const double da = lane_get(dend, tbase + T::NARM), db = lane_get(dend, tbase + T::NARM + 1);
    const double pa_ = lane_get(psum, tbase + T::NARM), pb_ = lane_get(psum, tbase + T::NARM + 1);
    if (t == T::NARM || t == T::NARM + 1) { dend_j = 0.5 * (da + db); psum_j = 0.5 * (pa_ + pb_); }
  }
  if (t < 12) { travel[t] = valid ? dend_j : 0.0; travelP[t] = valid ? psum_j : 0.0; }
  double* slides = mv + 4 * NL * (NL + 1) + 4 * 12 * NL + 12 * kMaxCGeom + 4 * 12 + 12 * team;  // finger_boxes_certified's `sl`
  __syncthreads();
  // Certifying mode (RunOp::check 2; per-environment escalation): a pair counts as apart only if it is PROVEN apart by more than
  // its margin -- the most the joints between its two links can have moved the two geoms relative to each other, the sum over those
  // joints of (a travel of the joint) x (CheckTable::lev[joint][link]: how far a radian / metre of it moves a point of a geom ON that
  // link, in any configuration of the joints in between).  Two travels, two margins, two tests (q: a position the joint took):
  //  * `dend` >= |q - q_end|: the widest the joint has been from where the launch ended.  A contact at some substep needs the gap
  //    to have been zero then, and the final gap can exceed the gap at that substep by at most the margin of dend:
  //    gap at the end > that margin certifies the WHOLE launch, the contacts that begin and end inside it included (the blind spot of
  //    a check of the final position alone: 6 of 512 rollouts of 1000 steps, round 5);
  //  * `psum` >= |q - q_start| + |q - q_end|: the joint's effective path (twice the width of the interval it has been in less its net
  //    displacement).  A pair that is d0 apart when the launch begins and d1 apart when it ends cannot have touched in between unless
  //    some position on the way is d0 from the first AND d1 from the last: d0 + d1 > the margin of psum certifies too (both lower
  //    bounds of the gaps, both positive).  Closing fingers, an arm sinking towards the floor: the gap shrinks by exactly what the
  //    joints travel, the first test fails for the last step or two before every meeting, this one does not.
  // Whatever is not proven apart sends the environment to the contact-resolving kernel, which looks in every substep.  (Round 5
  // charged every joint with the reach of the whole arm below it
  // moved[l][c + 1]: the geoms ON link l relative to the frame of link c (an ancestor-or-self of l; c = -1: the world).  A pair is
  // charged moved[la][c] + moved[lb][c], c the deepest common ancestor of its links (CheckEntry); the floor moved[l][-1].
  float* mvD = reinterpret_cast<float*>(mv) + NL * (NL + 1) * team;            // margins of dend
  float* mvP = reinterpret_cast<float*>(mv) + NL * (NL + 1) * (4 + team);      // margins of psum
  double* F0 = mv + 4 * NL * (NL + 1) + 12 * NL * team;   // the links' world frames where the launch began
  double* gbox = mv + 4 * NL * (NL + 1) + 4 * 12 * NL;    // the geoms' boxes in their links' frames (one copy: the teams share the model)
  if (valid) {
    double x[NL], y[NL];
    const uint32_t am = anc_mask<T>(t);
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      x[j] = (am >> j) & 1u ? (double)pf.lev[j] * travel[j] : 0.0;
      y[j] = (am >> j) & 1u ? (double)pf.lev[j] * travelP[j] : 0.0;
    }
    // (the arm is a chain: the joints of l's chain that are not on c's are those above c -- running sums from the chain's end; only
    // the fingers' own entries need the masks)
    double m = 0.0, mp_ = 0.0;
#pragma unroll
    for (int c = NL - 1; c >= -1; --c) {
      if (c >= T::NARM) {
        const uint32_t jm = am & ~anc_mask<T>(c);
        double m2 = 0.0, mp2 = 0.0;
#pragma unroll
        for (int j = T::NARM; j < NL; ++j) { m2 += (jm >> j) & 1u ? x[j] : 0.0; mp2 += (jm >> j) & 1u ? y[j] : 0.0; }
        mvD[t * (NL + 1) + c + 1] = m2 > 0.0 ? (float)m2 * 1.000001f + 1e-12f : 0.0f;
        mvP[t * (NL + 1) + c + 1] = mp2 > 0.0 ? (float)mp2 * 1.000001f + 1e-12f : 0.0f;
        if (c == T::NARM) {  // (from here down every finger joint of the lane's chain counts)
#pragma unroll
          for (int j = T::NARM; j < NL; ++j) { m += x[j]; mp_ += y[j]; }
        }
        continue;
      }
      // (single precision, rounded up: a margin errs on the large side)
      mvD[t * (NL + 1) + c + 1] = m > 0.0 ? (float)m * 1.000001f + 1e-12f : 0.0f;
      mvP[t * (NL + 1) + c + 1] = mp_ > 0.0 ? (float)mp_ * 1.000001f + 1e-12f : 0.0f;
      if (c >= 0) { m += x[c]; mp_ += y[c]; }
    }
  }
  __syncthreads();
  if (keep_slack == 2 && __ballot(live && !known_hit) == 0) {
    // the contact-resolving launch, every environment of the wavefront in contact during the launch: it stays whatever a certificate
    // says.  The pairs' record is the collision passes' (valid where the last substep began, which is where their next pass charges
    // from); the links' heights above the floor, which only this check keeps, are unknown from here on
    if (slack_env && live && t < NL && check_plane) slack_env[kSlackLink + t] = 0.0f;
    TAIL_END(true)
    return known_hit;
  }
  double mj[kCheckPer], mjP[kCheckPer];  // the margins of the lane's pairs (of dend, of psum)
#pragma unroll
  for (int j = 0; j < kCheckPer; ++j) {
    const int g0 = pf.ent[j].geoms & 0xff, g1 = (pf.ent[j].geoms >> 8) & 0xff, cc = (pf.ent[j].geoms >> 16) & 0xff;
    const int la = ck.glink[g0], lb = ck.glink[g1];
    mj[j] = (double)(la >= 0 ? mvD[la * (NL + 1) + cc] : 0.0f) + (double)(lb >= 0 ? mvD[lb * (NL + 1) + cc] : 0.0f);
    mjP[j] = (double)(la >= 0 ? mvP[la * (NL + 1) + cc] : 0.0f) + (double)(lb >= 0 ? mvP[lb * (NL + 1) + cc] : 0.0f);
  }
  const double mfl = mvD[tl * (NL + 1)], mflP = mvP[tl * (NL + 1)];
  // ---- THE SLACK.  A pair proven g apart at the position the launch BEGAN on (or at any other position of its path: any two are at
  // most the effective path apart) cannot have touched during it unless some position on
  // the way is g from that first one -- |q - q_start| is at most the joints' effective path: g > the margin of psum certifies the pair
  // for this launch with no geometry at all, and g less that margin is a lower bound of its gap where the launch ended, the next
  // launch's g (CheckTable::slack, per environment; the contact-resolving launch keeps the same record substep by substep,
  // contact_team.h: contact_collide).  Only the pairs whose slack is used up are LOOKED at -- link frames, boxes, support queries -- and
  // come back with a fresh gap; in a rollout that touches nothing that is a pair every few launches, and most launches end right here.
  // (Round 5 looked at every pair in every launch: 11.5 us of 121; the certifying tests on top of that: 25 of 137.)
  float rem[kCheckPer], remL = use_slack ? pf.remL : 0.0f;
  uint32_t due = 0;
#pragma unroll
  for (int j = 0; j < kCheckPer; ++j) {
    rem[j] = use_slack ? pf.rem[j] : 0.0f;
    const bool d = live && t + kTeamLanes * j < npair && !((double)rem[j] > mjP[j]);
    due |= d ? 1u << j : 0u;
    if (!d && mjP[j] > 0.0) rem[j] -= (float)mjP[j] * 1.00001f + 2.5e-7f;  // (rounded up, and by more than the subtraction's own rounding)
  }
  const bool floor_on = check_plane && ck.plane_points && valid && live;
  const bool dueL = floor_on && !((double)remL > mflP);
  if (floor_on && !dueL && mflP > 0.0) remL -= (float)mflP * 1.00001f + 2.5e-7f;
#ifdef RCSH_CHECK_DEBUG
  atomicAdd(&g_chk_dbg[44], __popc(due) + (dueL ? 1 : 0));
  if (lane == 0) { atomicAdd(&g_chk_dbg[43], 1); if (__ballot(due != 0 || dueL) == 0) atomicAdd(&g_chk_dbg[42], 1); }
  if (dueL) atomicAdd(&g_chk_dbg[45], 1);
#endif
  if (__ballot(due != 0 || dueL) == 0) {  // nobody of the wavefront's four environments has anything to look at
    if (keep_slack && slack_env && live && !(ck.pad & 16)) {
#pragma unroll
      for (int j = 0; j < kCheckPer; ++j)
        if (t + kTeamLanes * j < npair) slack_env[t + kTeamLanes * j] = rem[j];
      if (valid) slack_env[kSlackLink + t] = remL;
    }
    TAIL_END(true)
    return false;
  }
  // ---- world frames of the links at the final qpos (what the next launch's first position stage will see)
  double R[9], p[3], R0[9], p0[3], axw[3];
  bool is_slide = false;
  {
    KinK kk;
    kk.load(links[tl]);
    link_local_frame(kk, q, R, p);
    scan_frames<T>(R, p);
    // ... and where the launch began (certifying mode: a gap at BOTH ends of the launch's path certifies more than one at its end)
    link_local_frame(kk, q0, R0, p0);
    scan_frames<T>(R0, p0);
    mulmv(R, kk.axis, axw);  // (a finger's lane: its slide's axis, world frame, at the launch's end)
    is_slide = kk.jtype == kSlide;
  }
  CHK_MARK(0)
  __syncthreads();  // (the link records have been read: their memory becomes the frames')
  double* F = frames + 12 * NL * team;
  if (valid) {
#pragma unroll
    for (int k = 0; k < 9; ++k) { F[12 * t + k] = R[k]; F0[12 * t + k] = R0[k]; }
#pragma unroll
    for (int k = 0; k < 3; ++k) { F[12 * t + 9 + k] = p[k]; F0[12 * t + 9 + k] = p0[k]; }
  }
  if (T::GRIP && (t == T::NARM || t == T::NARM + 1)) {
    double* sl = slides + 4 * (t - T::NARM);
    sl[0] = axw[0]; sl[1] = axw[1]; sl[2] = axw[2];
    sl[3] = is_slide ? dend : INFINITY;  // (a hinged finger: no such certificate)
    slides[8 + t - T::NARM] = is_slide ? psum : INFINITY;
    if (t == T::NARM) { slides[10] = is_slide ? dlo : -INFINITY; slides[11] = is_slide ? dhi : INFINITY; }  // (the opening's lane)
  }
  if (team == 0) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int g = t + kTeamLanes * u;
      if (g < ngeom) {
#pragma unroll
        for (int k = 0; k < 12; ++k) gbox[12 * g + k] = pf.grec[u][k];
      }
    }
  }
  __syncthreads();
  // ---- world boxes of the geoms: lane t takes geoms t, t + 16 (their link-frame boxes came with the prefetch)
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int g = t + kTeamLanes * u;
    if (g < ngeom) {
      const int link = ck.glink[g];
      double cw[3] = {pf.grec[u][0], pf.grec[u][1], pf.grec[u][2]}, Rw[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) Rw[k] = pf.grec[u][3 + k];
      if (link >= 0) {
        double LR[9], LP[3], c2[3], R2[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) LR[k] = F[12 * link + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) LP[k] = F[12 * link + 9 + k];
        mulmv(LR, cw, c2);
        mulmm(LR, Rw, R2);
#pragma unroll
        for (int k = 0; k < 3; ++k) cw[k] = c2[k] + LP[k];
#pragma unroll
        for (int k = 0; k < 9; ++k) Rw[k] = R2[k];
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) wbox[12 * g + k] = cw[k];
#pragma unroll
      for (int k = 0; k < 9; ++k) wbox[12 * g + 3 + k] = Rw[k];
    }
  }
  __syncthreads();
  CHK_MARK(1)
  bool mine = known_hit && live;
  // ---- the floor against the sample points of the link's collision geoms (as the DET launches test it)
  if (dueL) {
    const double* nrm = lc.plane_n;
    const double a[3] = {R[0] * nrm[0] + R[3] * nrm[1] + R[6] * nrm[2], R[1] * nrm[0] + R[4] * nrm[1] + R[7] * nrm[2],
                         R[2] * nrm[0] + R[5] * nrm[1] + R[8] * nrm[2]};
    const double b = dot3(nrm, p) - lc.plane_d;
    const double* sph = lc.link_sphere[t];
    const double thr = mfl - kCheckTouch;  // (a contact is a penetration by more than kCheckTouch; nothing moved: the exact test)
    const double a0[3] = {R0[0] * nrm[0] + R0[3] * nrm[1] + R0[6] * nrm[2], R0[1] * nrm[0] + R0[4] * nrm[1] + R0[7] * nrm[2],
                          R0[2] * nrm[0] + R0[5] * nrm[1] + R0[8] * nrm[2]};
    const double b0 = dot3(nrm, p0) - lc.plane_d;
    double low = b + a[0] * sph[0] + a[1] * sph[1] + a[2] * sph[2] - sph[3];  // (the lowest any of the link's points can be)
    if (low < thr) {
      low = INFINITY;
      // (four points a trip, their sixteen words asked for together: the points live in global memory, and a loop that asked for one,
      // waited, tested and asked for the next spent a memory round trip per point -- a link has up to 150 -- on the one lane whose
      // link hangs low, with its wavefront and, late in a long rollout when many arms hang low, the whole launch waiting)
      const int k1 = lc.link_adr[t + 1];
      for (int k0 = lc.link_adr[t]; k0 < k1; k0 += 4) {
        double v[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const double* src = lc.xyzr + 4 * (size_t)(k0 + u < k1 ? k0 + u : k0);
          v[u][0] = src[0]; v[u][1] = src[1]; v[u][2] = src[2]; v[u][3] = src[3];
        }
        sched_fence();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (k0 + u >= k1) continue;
          const double h1 = b + a[0] * v[u][0] + a[1] * v[u][1] + a[2] * v[u][2] - v[u][3];
          low = fmin(low, h1);
          if (h1 < thr) {
            // (the point's height where the launch began: the path-length form)
            const double h0 = b0 + a0[0] * v[u][0] + a0[1] * v[u][1] + a0[2] * v[u][2] - v[u][3];
            if (mfl > 0.0 && h1 > -kCheckTouch && h0 > -kCheckTouch && h0 + h1 > mflP - 2.0 * kCheckTouch) continue;
            mine = true;
#ifdef RCSH_CHECK_DEBUG
            atomicAdd(&g_chk_dbg[0], 1);
#endif
          }
        }
      }
    }
    remL = low < 1e30 ? fmaxf((float)low * 0.999999f - 1e-6f, 0.0f) : 1e30f;  // (a link without sample points: nothing to touch the floor with)
  }
  CHK_MARK(2)
  // ---- geom pairs: bounding spheres (bit j of smask: pair t + 16 j survived), then -- one pair per lane and round -- the boxes
  uint32_t smask = 0;
  if (live && !(ck.pad & 4)) {
    // (all the centres first, then the arithmetic: read pair by pair the wavefront would wait for LDS a dozen times)
    double ca[kCheckPer][3], cb[kCheckPer][3];
#pragma unroll
    for (int j = 0; j < kCheckPer; ++j) {
      const int g0 = pf.ent[j].geoms & 0xff, g1 = (pf.ent[j].geoms >> 8) & 0xff;
#pragma unroll
      for (int k = 0; k < 3; ++k) { ca[j][k] = wbox[12 * g0 + k]; cb[j][k] = wbox[12 * g1 + k]; }
    }
    sched_fence();
#pragma unroll
    for (int j = 0; j < kCheckPer; ++j) {
      const int i = t + kTeamLanes * j;
      const double d[3] = {ca[j][0] - cb[j][0], ca[j][1] - cb[j][1], ca[j][2] - cb[j][2]};
      const double rs = pf.ent[j].rsum + mj[j], d2 = dot3(d, d);
      const bool dj = (due >> j) & 1u;
      smask |= dj && i < npair && d2 <= rs * rs ? 1u << j : 0u;
      if (dj && d2 > rs * rs) rem[j] = fmaxf((float)(sqrt(d2) - (double)pf.ent[j].rsum) * 0.999999f - 1e-6f, 0.0f);  // (settled by the spheres: their gap)
    }
  }
#ifdef RCSH_CHECK_DEBUG
  atomicAdd(&g_chk_dbg[32], __popc(smask));
#endif
  CHK_MARK(3)
  uint32_t cmask = 0;
  if (ck.pad & 2) smask = 0;
  // (the contact-resolving launch is asked one thing -- may the environment go back -- and has its answer with the first pair that
  // fails: every live team has one -> nothing more to look at; what was not looked at is not written back, see the end)
  auto all_hit = [&]() { return keep_slack == 2 && __ballot(live && team_ballot(mine) == 0) == 0; };
  while (__ballot(smask != 0)) {
    if (all_hit()) break;
    TAIL_COUNT(4)
    if (smask) {
      const int j = __ffs((int)smask) - 1;
      smask &= smask - 1;
      // (the entry of round j: a select chain over the lane's registers -- a run-time index would put them into scratch)
      uint32_t gg = 0;
      double mm = 0.0, mmP = 0.0;
#pragma unroll
      for (int k = 0; k < kCheckPer; ++k) { gg = k == j ? pf.ent[k].geoms : gg; mm = k == j ? mj[k] : mm; mmP = k == j ? mjP[k] : mmP; }
      const int g0 = gg & 0xff, g1 = (gg >> 8) & 0xff;
      double Ra[9], Rb[9], ca[3], cb[3], ha[3], hb[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) { ca[k] = wbox[12 * g0 + k]; cb[k] = wbox[12 * g1 + k]; ha[k] = ck.gh[g0][k]; hb[k] = ck.gh[g1][k]; }
#pragma unroll
      for (int k = 0; k < 9; ++k) { Ra[k] = wbox[12 * g0 + 3 + k]; Rb[k] = wbox[12 * g1 + 3 + k]; }
      bool settled;
      double sep_end = 0.0;
      const bool finger_pair = T::GRIP && ck.glink[g0] >= T::NARM && ck.glink[g1] >= T::NARM;  // (only the two slides between them)
      if (mm > 0.0) {
        const double sep1 = obb_face_sep(Ra, ca, ha, Rb, cb, hb);
        sep_end = sep1;
        settled = sep1 > mm - kCheckTouch;
        if (!settled && (sep1 > -kCheckTouch || finger_pair)) {
          // the same boxes where the launch began (their links' frames then, the boxes in their links' frames)
          const int la = ck.glink[g0], lb = ck.glink[g1];
          double Ra0[9], Rb0[9], ca0[3], cb0[3];
          self_box_world(F0, la, gbox + 12 * g0, gbox + 12 * g0 + 3, ca0, Ra0);
          self_box_world(F0, lb, gbox + 12 * g1, gbox + 12 * g1 + 3, cb0, Rb0);
          if (finger_pair) {
            settled = finger_boxes_certified(Ra, ca, ha, Rb, cb, hb, Ra0, ca0, Rb0, cb0, slides, kCheckTouch, la == T::NARM ? 1.0 : -1.0);
          } else {
            const double sep0 = obb_face_sep(Ra0, ca0, ha, Rb0, cb0, hb);
            settled = sep0 > -kCheckTouch && sep0 + sep1 > mmP - 2.0 * kCheckTouch;
          }
        }
      } else {
        settled = obb_apart_or_touching(Ra, ca, ha, Rb, cb, hb, kCheckTouch, 0.0);
      }
      {
        // (settled: what the boxes prove of the pair's gap at the launch's end is its new slack)
        const float nr = settled ? fmaxf((float)sep_end * 0.999999f - 1e-6f, 0.0f) : 0.0f;
#pragma unroll
        for (int k = 0; k < kCheckPer; ++k) rem[k] = k == j ? nr : rem[k];
      }
      if (!settled) {
        cmask |= 1u << j;
#ifdef RCSH_CHECK_DEBUG
        atomicAdd(&g_chk_dbg[33], 1);
#endif
      }
    }
  }
  CHK_MARK(4)
  // ---- narrow phase: a surviving geom pair has its hulls staged by the whole wavefront, once for all the teams it survived in;
  // each of those teams runs the refinement on its 16 lanes, which share the vertex scans of the support queries
  __syncthreads();  // (the stage takes the world boxes' place)
#ifdef RCSH_CHECK_TAIL
  const unsigned long long tail_t1_ = __builtin_readcyclecounter();
  tail_pre_ = tail_t1_ - tail_t0_;
#endif
  if (ck.pad & 1) cmask = 0;
  int nround = 0;  // (team-uniform) narrow-phase rounds this team's environment has taken part in
  for (uint64_t pc = __ballot(cmask != 0); pc; pc = __ballot(cmask != 0)) {
    if (all_hit()) break;
    const int src = __ffsll((long long)pc) - 1;  // wave-uniform
    TAIL_COUNT(0)

    const uint32_t sm = (uint32_t)__builtin_amdgcn_readlane((int)cmask, src);
    const int u = __ffs((int)sm) - 1, t1 = src & (kTeamLanes - 1);
    const bool holder = t == t1 && ((cmask >> u) & 1u);
    bool take = team_ballot(holder) != 0;
#ifdef RCSH_CHECK_TAIL
    tail_rounds_ += take ? 1 : 0;
#endif
    if (holder) cmask &= ~(1u << u);
    if (keep_slack != 0 && take) {
      // THE BUDGET.  A launch waits for its slowest wavefront, and late in a long rollout that is one with an environment NEAR contact
      // whose certificate takes ten to twenty of these rounds (13k cycles each: at launch 1000 of the headline rollout the lean launch
      // took 300 us instead of 137, `profiles/r6_escalated3`).  Such an environment is where the contact-resolving launch looks in every
      // substep anyway: past the budget it is not certified -- conservative, like every other way of failing -- and stays there until
      // its certificate fits the budget again (the same rule on both launches: no going back and forth).
      nround += 1;
      if (nround > kCheckRoundBudget) { mine = true; cmask = 0; take = false; }
    }
    const int pidx = t1 + kTeamLanes * u;
    uint32_t gg = 0;
    double mu_ = 0.0, muP_ = 0.0;
#pragma unroll
    for (int k = 0; k < kCheckPer; ++k) { gg = k == u ? pf.ent[k].geoms : gg; mu_ = k == u ? mj[k] : mu_; muP_ = k == u ? mjP[k] : muP_; }
    const double mteam = lane_get(mu_, (lane & 48) + t1);  // the margins of this pair in the lane's team (its lane t1 holds them)
    const double mteamP = lane_get(muP_, (lane & 48) + t1);
    gg = (uint32_t)__builtin_amdgcn_readlane((int)gg, src);
    const int g0 = gg & 0xff, g1 = (gg >> 8) & 0xff;
    const int na = 3 * ck.gvert[g0][1], nb = 3 * ck.gvert[g1][1];
    const double key = check_slot_key(pidx, g0, g1);
    // the two geom records and the hulls' vertices: from the prefetch where the guess was right, else from memory now
#ifdef RCSH_CHECK_TAIL
    const unsigned long long tail_s0_ = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0 && pf.guess_key != (int)key) tail_miss_ += 1;
#endif
    if (pf.guess_key == (int)key) {
      gstage[lane] = pf.gword;
#pragma unroll
      for (int v = 0; v < kCheckTrips; ++v) { const int k = lane + 64 * v; if (k < na) stage[k] = pf.va[v]; if (k < nb) stage[na + k] = pf.vb[v]; }
    } else {
      const int w = lane & 31;
      const double* rec = reinterpret_cast<const double*>(&tab.geoms[lane < 32 ? g0 : g1]);
      const double word = rec[w < kCheckGeomWords ? w : 0];
      const double* va = tab.verts + 3 * (size_t)ck.gvert[g0][0];
      const double* vb = tab.verts + 3 * (size_t)ck.gvert[g1][0];
      double xa[kCheckTrips], xb[kCheckTrips];
#pragma unroll
      for (int v = 0; v < kCheckTrips; ++v) { const int k = lane + 64 * v; xa[v] = va[k < na ? k : 0]; xb[v] = vb[k < nb ? k : 0]; }
      gstage[lane] = word;
#pragma unroll
      for (int v = 0; v < kCheckTrips; ++v) { const int k = lane + 64 * v; if (k < na) stage[k] = xa[v]; if (k < nb) stage[na + k] = xb[v]; }
#ifdef RCSH_CHECK_DEBUG
      if (lane == 0) atomicAdd(&g_chk_dbg[39], 1);  // pairs whose data was not prefetched
#endif
    }
    stage_fence();  // (LDS traffic of one wavefront is ordered)
#ifdef RCSH_CHECK_TAIL
    if ((threadIdx.x & 63) == 0) tail_stage_ += __builtin_readcyclecounter() - tail_s0_;
#endif
    bool apart_out = false;
    double gcert_out = 0.0;
#ifdef RCSH_CHECK_TAIL
    const unsigned long long tail_c0_ = __builtin_readcyclecounter();
#endif
    if (take) {
      const ContactGeom& a = *reinterpret_cast<const ContactGeom*>(gstage);
      const ContactGeom& b = *reinterpret_cast<const ContactGeom*>(gstage + 32);
      double Ra[9], pa[3], Rb[9], pb[3];
      self_geom_world(a, F, Ra, pa);
      self_geom_world(b, F, Rb, pb);
      Shape A = make_shape(a.type == 7 ? 0 : a.type == 6 ? 1 : 2, pa, Ra, a.size, stage, a.vert_num);
      Shape B = make_shape(b.type == 7 ? 0 : b.type == 6 ? 1 : 2, pb, Rb, b.size, stage + na, b.vert_num);
      if (a.type == 7) { mulmv(Ra, a.center, A.center); A.center[0] += pa[0]; A.center[1] += pa[1]; A.center[2] += pa[2]; }
      if (b.type == 7) { mulmv(Rb, b.center, B.center); B.center[0] += pb[0]; B.center[1] += pb[1]; B.center[2] += pb[2]; }
      // the slot of this pair: the one that holds it, else the first empty one, else slot (pair index mod the number of slots)
      int s_hold = -1, s_free = -1;
#pragma unroll
      for (int k = kCheckSlots - 1; k >= 0; --k) {
        const double kk = lane_get(sepw, tbase + 4 * k);
        s_hold = kk == key ? k : s_hold;
        s_free = kk == 0.0 ? k : s_free;  // (the first empty one)
      }
      const int s_use = s_hold >= 0 ? s_hold : (s_free >= 0 ? s_free : (pidx & (kCheckSlots - 1)));
      auto slot_store = [&](const double* dl_) {  // the pair's direction into slot s_use: its four lanes take their words
        const int w_ = t - 4 * s_use;
        if (w_ >= 0 && w_ < 4) sepw = w_ == 0 ? key : (w_ == 1 ? dl_[0] : (w_ == 2 ? dl_[1] : dl_[2]));
      };
      bool apart = false;
      double gcert = 0.0;  // the gap the pair is found apart by (its new slack)
      double LR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      if (a.link >= 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) LR[k] = F[12 * a.link + k];
      }
      // certifying mode: is a gap g1 (> 0, proven along the link-frame direction dl at the launch's end) enough -- by itself, or together
      // with the gap the same direction proves where the launch began (the path-length form; one more support query, at the start frames)
      double dbg_g1 = -1.0, dbg_g0 = -1.0;
      (void)dbg_g1; (void)dbg_g0;
      const bool finger_pair = T::GRIP && a.link >= T::NARM && b.link >= T::NARM;  // (only the two slides between the geoms: a translation)
      auto certified = [&](double g1, const double* dl) -> bool {
        dbg_g1 = g1;
        double mD = mteam, mP = mteamP;
        if (finger_pair) {
          // the gap along a direction changes by exactly that direction's component of the slides' translation (finger_boxes_certified)
          double dwn[3];
          mulmv(LR, dl, dwn);
          const double np_ = fabs(0.5 * (dot3(dwn, slides + 4) - dot3(dwn, slides))), nm_ = fabs(0.5 * (dot3(dwn, slides + 4) + dot3(dwn, slides)));
          const double mDd = np_ * slides[3] + nm_ * slides[7], mPd = np_ * slides[8] + nm_ * slides[9];
          if (mDd < mD) { mD = mDd; mP = mPd; }  // (a hinged finger: infinite -- the levers' margins stay)
        }
        else if (mteam > 0.0 && !(g1 > mD - kCheckTouch)) {
          // second chance: the levers of these two GEOMS instead of their links' (a link's lever is its farthest geom's; kLevGeom).  The
          // joints between the two links, each charged its travel: not for chains through the fingers' slides, whose travel entries are
          // the gripper's opening and shift
          const int cc_ = (int)((gg >> 16) & 0xff) - 1;
          const uint32_t ja = a.link >= 0 ? anc_mask<T>(a.link) & ~anc_mask<T>(cc_) : 0u, jb = b.link >= 0 ? anc_mask<T>(b.link) & ~anc_mask<T>(cc_) : 0u;
          if (!T::GRIP || ((ja | jb) >> T::NARM) == 0u) {
            const float* lg = ck.lev + kLevGeom;
            const int ga_ = (int)(gg & 0xff), gb_ = (int)((gg >> 8) & 0xff);
            double sD = 0.0, sP = 0.0;
#pragma unroll
            for (int j = 0; j < NL; ++j) {
              const double w = ((ja >> j) & 1u ? (double)lg[32 * j + ga_] : 0.0) + ((jb >> j) & 1u ? (double)lg[32 * j + gb_] : 0.0);
              sD += w * travel[j]; sP += w * travelP[j];
            }
            sD = sD * 1.000001 + 1e-12; sP = sP * 1.000001 + 1e-12;
            mD = fmin(mD, sD); mP = fmin(mP, sP);
          }
        }
        if (g1 > mD - kCheckTouch) return true;
        if (!(g1 > -kCheckTouch)) return false;
        double Ra0[9], pa0[3], Rb0[9], pb0[3], LR0[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, dw0[3];
        self_geom_world(a, F0, Ra0, pa0);
        self_geom_world(b, F0, Rb0, pb0);
        const Shape A0 = make_shape(A.type, pa0, Ra0, a.size, A.verts, A.nvert), B0 = make_shape(B.type, pb0, Rb0, b.size, B.verts, B.nvert);
        if (a.link >= 0) {
#pragma unroll
          for (int k = 0; k < 9; ++k) LR0[k] = F0[12 * a.link + k];
        }
        mulmv(LR0, dl, dw0);
        TAIL_COUNT(2)
        const double g0 = support_gap(A0, B0, dw0);
        dbg_g0 = g0;
        return g0 > -kCheckTouch && g0 + g1 > mP - 2.0 * kCheckTouch;
      };
      double x0[3] = {A.center[0] - B.center[0], A.center[1] - B.center[1], A.center[2] - B.center[2]};  // (a point of A - B)
#ifdef RCSH_CHECK_TAIL
      const unsigned long long tail_c1_ = __builtin_readcyclecounter();
      if ((int)(threadIdx.x & 63) == __ffsll((long long)__ballot(true)) - 1) atomicAdd(&tail_sh_[6], (unsigned)(tail_c1_ - tail_c0_));
#endif
      if (s_hold >= 0) {
        const double dl[3] = {lane_get(sepw, tbase + 4 * s_use + 1), lane_get(sepw, tbase + 4 * s_use + 2), lane_get(sepw, tbase + 4 * s_use + 3)};
        double dw[3];
        mulmv(LR, dl, dw);
        // the support of A - B along the remembered direction is still negative: apart (certifying mode: by enough)
        const double g1 = support_gap(A, B, dw);
        apart = mteam > 0.0 ? certified(g1, dl) : g1 > 0.0;
        gcert = g1;
      }
#ifdef RCSH_CHECK_TAIL
      if ((int)(threadIdx.x & 63) == __ffsll((long long)__ballot(true)) - 1) atomicAdd(&tail_sh_[5], (unsigned)(__builtin_readcyclecounter() - tail_c1_));
#endif
      if (!apart && s_hold < 0 && finger_pair && slides[3] < 1e300) {
        // two geoms on the two fingers without a remembered direction (a shut or shutting gripper brings eleven hull pairs near at
        // once -- each pad against the other finger's hull, the two hulls -- and an environment remembers four): what separates them is
        // the slides' own axis, nearly always -- one support query along it before any iteration.  (Without it every such pair ran
        // Gilbert's iteration in every launch in which the fingers moved: the one wavefront in a hundred the whole launch then waits for,
        // tools/check_tail.py.)
        double e[3] = {slides[4] - slides[0], slides[5] - slides[1], slides[6] - slides[2]};
        const double en = sqrt(dot3(e, e));
        if (en > 1e-9) {
          const double sg = (e[0] * (B.center[0] - A.center[0]) + e[1] * (B.center[1] - A.center[1]) + e[2] * (B.center[2] - A.center[2]) >= 0.0 ? 1.0 : -1.0) / en;
          const double dw[3] = {sg * e[0], sg * e[1], sg * e[2]};
          double dl[3];
          mulTv(LR, dw, dl);
          const double g1 = support_gap(A, B, dw);
          apart = mteam > 0.0 ? certified(g1, dl) : g1 > 0.0;
          gcert = g1;
          if (apart) { stage_fence(); slot_store(dl); }
        }
      }
#ifndef RCSH_NO_GILBERT
      if (!apart) {
        // a few support queries towards a separating direction before the full refinement (contact_team.h: gilbert_apart); the margin
        // keeps its verdicts far from the nanometre the check calls contact.  Certifying mode: a few more, after the first proof, for a
        // direction with a larger gap -- it is the gap that certifies, and the direction is remembered
        double dg[3], gap = 0.0;
        TAIL_COUNT(1)
        if (gilbert_apart<true>(A, B, x0, mteam > 0.0 ? 6 : 5, 1e-5, dg, &gap, mteam > 0.0 ? 1 : 0)) {
          double dl[3];
          mulTv(LR, dg, dl);
          apart = mteam > 0.0 ? certified(gap, dl) : true;
          gcert = gap;
          stage_fence();
          slot_store(dl);
        }
      }
#endif
      if (!apart && mteam > 0.0) {
        mine = true;  // (certifying mode: not proven apart by enough)
#ifdef RCSH_CHECK_DEBUG
        if (t == 0) {
          const int k = atomicAdd(&g_chk_dbg[1], 1);
          if (k < 28) g_chk_dbg[2 + k] = pidx;
          if (k < 32) { g_chk_dbgf[4 * k] = pidx; g_chk_dbgf[4 * k + 1] = mteam; g_chk_dbgf[4 * k + 2] = dbg_g1; g_chk_dbgf[4 * k + 3] = dbg_g0; }
        }
#endif
      }
      else if (!apart && !(ck.pad & 8)) {
        TAIL_COUNT(3)
        double dir[3], depth = 0.0;
        if (mpr_penetration<true, kMprDepth>(A, B, &depth, dir, nullptr)) {
          if (depth > kCheckTouch) {
            mine = true;
#ifdef RCSH_CHECK_DEBUG
            if (t == 0) { const int k = atomicAdd(&g_chk_dbg[1], 1); if (k < 28) g_chk_dbg[2 + k] = pidx; }
#endif
          }
        } else if (dot3(dir, dir) > 0.5) {  // (a unit separating direction came back)
          double dl[3];
          mulTv(LR, dir, dl);
          stage_fence();
          slot_store(dl);
        }
#ifdef RCSH_CHECK_DEBUG
        if (t == 0) { atomicAdd(&g_chk_dbg[38], 1); atomicAdd(&g_chk_dbg[s_hold >= 0 ? 41 : 40], 1); }  // full refinements: [40] no slot held the pair, [41] its direction failed
#endif
      }
      apart_out = apart;
      gcert_out = gcert;
    }
#ifdef RCSH_CHECK_TAIL
    if ((threadIdx.x & 63) == 0) atomicAdd(&tail_sh_[7], (unsigned)(__builtin_readcyclecounter() - tail_c0_));
#endif
    {
      // the pair's new slack, on the lane that holds the pair (the team's lanes agree on it)
      const float nr = apart_out ? fmaxf((float)gcert_out * 0.999999f - 1e-6f, 0.0f) : 0.0f;
      if (holder) {
#pragma unroll
        for (int k = 0; k < kCheckPer; ++k) rem[k] = k == u ? nr : rem[k];
      }
    }
    stage_fence();
  }
  CHK_MARK(5)
#ifdef RCSH_CHECK_TAIL
  tail_nar_ = __builtin_readcyclecounter() - tail_t1_;
  if (keep_slack == 1 && live && t == 0) atomicAdd(&g_chk_hist[48 + (tail_rounds_ > 15 ? 15 : tail_rounds_)], 1ull);
#endif
  __syncthreads();
  if (t < kCheckSep && live) sep[(size_t)t * n_env] = sepw;
  const bool hit = team_ballot(mine) != 0;
  // the slack goes back -- unless the environment is found in contact.  The lean launch's is then redone from the position the launch
  // BEGAN on, which is what the record as it stands describes; the contact-resolving launch's stays there, and the record its collision
  // passes keep (valid where the last SUBSTEP began, and charged from there by their next pass) is the one to go on with -- this check may
  // not even have looked at every due pair (all_hit).  An environment that goes BACK has had every due pair looked at where the launch
  // ended and every other one charged the launch's whole path: that describes the launch's last position, which is what the lean launch
  // will read it as.  The links' heights above the floor only this check keeps: they go back either way.
  if (slack_env && live && (!hit || keep_slack == 2) && !(ck.pad & 16)) {  // (pad bit 4: a timing experiment -- RCSH_CHECK_SKIP, rcs_hip.hip)
    if (keep_slack && !hit) {
#pragma unroll
      for (int j = 0; j < kCheckPer; ++j)
        if (t + kTeamLanes * j < npair) slack_env[t + kTeamLanes * j] = rem[j];
    }
    if (valid && check_plane) slack_env[kSlackLink + t] = remL;
  }
  CHK_MARK(6)
  TAIL_END(false)
  return hit;
}

#endif  // __HIP__

}  // namespace rcsh
