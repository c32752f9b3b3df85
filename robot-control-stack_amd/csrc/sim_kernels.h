// sim_kernels.h -- the batched Sim / SimRobot / SimGripper / Gymnasium-loop kernels.
//
// One launch is one call of the reference's API for every environment: a team of 16 lanes (k_run_team) owns an
// environment for the whole launch, loads its state from the SoA arrays ([field][env]),
// keeps it in LDS / registers across all physics substeps of the call, and writes it back once.  The reference's
// per-substep callback scheduler (reference src/sim/sim.cpp:14-61) is evaluated by the environment's leader lane with
// the same double-precision timestamps and strict '>' compares, so callback cadence -- and with it every flag and
// substep count -- follows the reference.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "box_team.h"
#include "contact_team.h"
#include "check_team.h"
#include "dyn.h"
#include "dyn_team.h"
#include "pose.h"

namespace rcsh {

// ---- per-environment flag word
enum : uint32_t {
  kAnyRet0 = 1u << 0,   // last_return_value of any_callbacks[0] (robot collision)
  kAnyRet1 = 1u << 1,   // any_callbacks[1] (gripper collision)
  kAllRet0 = 1u << 2,   // all_callbacks[0] (robot convergence)
  kAllRet1 = 1u << 3,   // all_callbacks[1] (gripper convergence)
  kConverged = 1u << 4,
  kIkSuccess = 1u << 5,
  kRobotCollision = 1u << 6,
  kIsMoving = 1u << 7,
  kIsArrived = 1u << 8,
  kGripMoving = 1u << 9,
  kGripCollision = 1u << 10,
  kHasPrevAction = 1u << 11,
  kHasGripCmd = 1u << 12,
  kGripCmd = 1u << 13,
  kHasLastAction = 1u << 14,
  kContactOverflow = 1u << 15,  // sticky until Sim::reset: a contact phase of this environment ran out of contact / link slots
  kEscQuiet = 1u << 17,         // (unused since round 6: an environment goes back to the lean launch by the certificate, not by a count of quiet launches)
  kContactResolved = 1u << 18,  // sticky until Sim::reset: a contact of this environment's robot geoms was resolved (per-environment escalation)
  kContactUnresolved = 1u << 16,  // sticky until Sim::reset: the environment's geoms were found in a contact this configuration does
                                  // not resolve (check_team.h): from then on its trajectory is not what MuJoCo's would be
};

struct SimCfg {
  int32_t async_control, realtime, frequency, max_convergence_steps;
};
struct RobotCfg {
  int32_t present, conv_registered;
  double tolerance, period;
  double tcp[7];  // xyz + xyzw
  double q_home[kMaxArm];
};
struct GripperCfg {
  int32_t present, finger;  // finger: 0/1 = which finger dof is the gripper joint
  double eps_inner, eps_outer, period;
  double max_act, min_act, max_joint, min_joint;
};
struct EnvCfg {
  int32_t mode, relative_to, binary_gripper, pad;
  double max_mov[2];
  double low[kMaxArm], high[kMaxArm];
};

// SoA layout: field f of environment e lives at base[(f) * n + e]
template <class T>
struct Lay {
  static constexpr int QPOS = 0;
  static constexpr int QVEL = QPOS + T::NL;
  static constexpr int CTRL = QVEL + T::NL;
  static constexpr int TIME = CTRL + T::NU;
  static constexpr int CB = TIME + 1;            // plain0 plain1 any0 any1 all0 all1 last_call_timestamp
  static constexpr int PREVQ = CB + 6;           // SimRobotState.previous_angles
  static constexpr int TARGET = PREVQ + T::NARM; // SimRobotState.target_angles
  static constexpr int GRIP = TARGET + T::NARM;  // last_commanded_width, last_width
  static constexpr int SITE = GRIP + 2;          // frame of the site link at the last mj_step1: R(9) p(3)
  static constexpr int PREVA = SITE + 12;        // RobotEnv.prev_action (7 wide: joints or tquat)
  static constexpr int ORIGIN = PREVA + 7;       // RelativeActionSpace._origin
  static constexpr int LASTA = ORIGIN + 7;       // RelativeActionSpace._last_action
  static constexpr int BOX = LASTA + 7;           // free box (box_team.h): qpos 7, qvel 6, qacc_warmstart 6, pose seen by the last position stage 7, minimiser of the last coupled solve 15
  static constexpr int QPRE = BOX + kBoxState;    // qpos seen by the last mj_step1 (what mjData.xpos / geom_xpos / cam_xpos derive from: the renderer's frames)
  static constexpr int XS = QPRE + T::NL;         // models with dry friction: the constraint solve's last solution (MuJoCo: qacc_warmstart) -- the
                                                  // zones it sits in are the first guess of the next solve, also across launches
  static constexpr int SEP = XS + T::NL;          // the unresolved-contact check's remembered separating directions (check_team.h)
  static constexpr int COUNT = SEP + kCheckSep;
};

// PickCubeSuccessWrapper (reference python/rcs/envs/sim.py:386-431)
struct TaskCfg {
  int32_t pick_cube, pad;
  double ee_home[3];
  double success_z;  // the cube counts as lifted above this world height
};

struct RunOp {
  int32_t do_reset;       // env.reset(): gripper reset, sim reset, robot reset (then nsteps = 1)
  int32_t apply_action;   // env.step(): wrappers' action() + RobotEnv.step
  int32_t nsteps;         // >= 0: Sim.step(nsteps); < 0: Sim.step_until_convergence()
  int32_t write_obs;
  int32_t observe_only;   // an observation-only pass (nsteps = 0) that must leave the rendering records of the last stepping launch alone
  int32_t check;          // end the launch with the check for contacts nobody resolves (check_team.h; the host decides the cadence);
                          // 2: certifying -- whatever cannot be PROVEN apart by more than it travelled during the launch counts as a hit
  const uint8_t* mask;    // optional, device
  const double* action;   // [n][action_width], device
  const float* gripper;   // [n], device
  double* obs;            // [n][obs_width]
  uint8_t* info;          // [n][8]
  double* gripper_width;  // [n]
  int32_t* substeps;      // [n]
  const double* box_qpos; // [n][7] env.reset() of the task env: RandomCubePos places the box (null: it stays at qpos0)
  double* task;           // [n][9] box qpos 7, reward, success (PickCubeSuccessWrapper.step)
  // Per-environment escalation (round 5; the certificate and the way back: round 6; host: launch_run).  MuJoCo resolves every contact
  // of every substep (reference src/sim/sim.cpp:108-115); the lean kernels resolve none.  So a step of a scene whose robot contacts are
  // to be resolved is TWO launches over disjoint sets of environments: role 1, the lean kernel over the environments NOT escalated,
  // which keeps a copy of every state field it read (snap) and whose end-of-launch check (check_team.h) marks an environment it cannot
  // CERTIFY -- no substep of the launch can have been in contact -- in esc[1] ("new"); then role 2, the contact-resolving kernel with
  // the contact phase in every substep, over the escalated environments esc[0] | esc[1]: a NEW one is stepped again from the copy --
  // the launch it was flagged in is redone with its contacts resolved from their first substep --, the others continue from their
  // state.  An environment none of whose substeps met a contact and whose launch passes the same certificate is marked in esc[2]
  // ("leave").  The last workgroup of the role-2 launch merges: esc[0] = (esc[0] & ~esc[2]) | esc[1].
  int32_t esc_role;
  int32_t force_contact;  // the contact phase runs in every substep of every environment (a whole batch on the contact-resolving kernel with
                          // self contacts resolved: exact, and slow -- the fast path's broad phase knows the floor and the free body only)
  int32_t conv_chunk;     // step_until_convergence in PIECES (host: launch_run): at most this many substeps in this launch (0: all of them).  The
                          // certificate's margins are the joints' travel over ONE launch: a 500-substep launch that moves a joint five degrees
                          // cannot clear pairs that are 15 mm apart in every pose, a quarter of it can -- and an environment that fails is
                          // redone for that piece only
  // Role 2 in two parts (esc_part): the environments that WERE escalated when the step began do not depend on the step's lean launch --
  // part 1 steps them on a stream of its own, next to the lean launch, and leaves the masks alone; part 2, behind both, redoes the newly
  // flagged ones and merges.  0: one launch does both (behind the lean one).  The host splits when it knows of escalated environments.
  int32_t esc_part;
  int32_t esc_seq;        // role 1: the step's number, for esc_host
  int32_t conv_resume;    // this launch CONTINUES a step_until_convergence begun by an earlier one: substep count, verdicts and the cap carry over
  uint32_t* esc_host;     // [2] host memory: the last step whose lean launch has (all but) ended, the escalated count it started with
  uint64_t* esc;          // [3][(n + 63) / 64]
  uint32_t* esc_ctr;      // [0] workgroups of the role-2 launch that are done, [1] environments escalated after the last merge,
                          // [2] environments the role-1 launch of this step has flagged (a role-2 launch that finds [1] = [2] = 0 ends at once)
  double* snap;           // [Lay::COUNT][n]
  uint32_t* snap_flags;   // [n]
  int32_t* snap_conv;     // [n]
};

// inclusive prefix sum over the 64 lanes of the wavefront
__device__ __forceinline__ int wave_incl_scan(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);
  const int t0 = __builtin_amdgcn_readlane(x, 15), t1 = __builtin_amdgcn_readlane(x, 31), t2 = __builtin_amdgcn_readlane(x, 47);
  const int row = (int)(threadIdx.x & 63u) >> 4;
  return x + (row >= 1 ? t0 : 0) + (row >= 2 ? t1 : 0) + (row >= 3 ? t2 : 0);
}
__device__ __forceinline__ uint64_t wave_read_u64(uint64_t x, int src) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, src), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), src);
  return (uint64_t)hi << 32 | lo;
}
// The environment workgroup slot r0 + team stands for under RunOp::esc_role (1: the r-th environment that is not escalated; 2: the
// r-th that is); -1: there is none.  `redo`: role 2, the environment is newly escalated (its step is redone from the copy).
// Every lane of the wavefront calls this; no LDS.
// Role 2 SPREADS its environments over the launch's workgroups -- one per workgroup while there are no more of them than workgroups,
// then two, three, four: workgroup slot r (r0 / 4) takes environments r per, ..., r per + per - 1 on its first teams, the other teams idle -- the contact phase puts all 64 lanes of a wavefront on ONE environment at a
// time (contact_team.h), so four escalated environments in one wavefront take turns while three quarters of the chip's SIMDs, which the
// lean launch has just left, stand empty.  (`spread`: r0 is then the workgroup's slot, not four times it.)
__device__ __forceinline__ int esc_select(const RunOp& op, int n, int r0, int team, bool& redo, bool spread = false) {
  const int lane = (int)(threadIdx.x & 63u);
  const int nw = (n + 63) >> 6;
  const uint64_t* A = op.esc;
  const uint64_t* B = op.esc + nw;
  int result = -1, base = 0, per = 4;
  redo = false;
  if (spread) {
    // how many are there?  (one pass over the masks; n <= 64 k environments: at most 16 words a lane)
    int cnt = 0;
    for (int wi = lane; wi < nw; wi += 64) {
      const uint64_t valid = (wi == nw - 1 && (n & 63)) ? ((1ull << (n & 63)) - 1) : ~0ull;
      cnt += __popcll(((op.esc_part == 2 ? 0ull : A[wi]) | (op.esc_part == 1 ? 0ull : B[wi])) & valid);
    }
    const int total = __builtin_amdgcn_readlane(wave_incl_scan(cnt), 63), G = (int)gridDim.x;
    per = total <= G ? 1 : (total <= 2 * G ? 2 : (total <= 3 * G ? 3 : 4));  // environments per workgroup: as few as the launch's workgroups allow
    r0 = (r0 / 4) * per;
  }
  for (int c0 = 0; c0 < nw; c0 += 64) {
    const int wi = c0 + lane;
    uint64_t a = 0, b = 0, valid = 0;
    if (wi < nw) {
      a = A[wi];
      b = op.esc_role == 2 && op.esc_part != 1 ? B[wi] : 0;
      valid = (wi == nw - 1 && (n & 63)) ? ((1ull << (n & 63)) - 1) : ~0ull;
    }
    const uint64_t w = (op.esc_role == 1 ? ~a : ((op.esc_part == 2 ? 0ull : a) | b)) & valid;
    const int pc = __popcll(w);
    const int incl = wave_incl_scan(pc);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    for (int k = 0; k < per; ++k) {
      const int rk = r0 + k - base;
      if (rk < 0 || rk >= total) continue;  // (wave-uniform)
      const int L = __ffsll((long long)__ballot(incl > rk)) - 1;
      const int excl = __builtin_amdgcn_readlane(incl, L) - __builtin_amdgcn_readlane(pc, L);
      uint64_t wl = wave_read_u64(w, L);
      for (int q = rk - excl; q > 0; --q) wl &= wl - 1;
      const int bit = __ffsll((long long)wl) - 1;
      if (team == k) {
        result = (c0 + L) * 64 + bit;
        redo = op.esc_role == 2 && !((wave_read_u64(a, L) >> bit) & 1ull);
      }
    }
    base += total;
    if (base > r0 + per - 1) break;
  }
  return result;
}
// role 2, at the very end of a workgroup: the last one to finish merges the masks (nobody reads them any more: the next launch of the
// stream starts after this one has ended)
__device__ __forceinline__ void esc_finish(const RunOp& op, int n) {
  __threadfence();
  int last = 0;
  if ((threadIdx.x & 63u) == 0) last = atomicAdd(op.esc_ctr, 1u) == gridDim.x - 1 ? 1 : 0;
  last = __builtin_amdgcn_readfirstlane(last);
  if (!last) return;
  __threadfence();
  const int nw = (n + 63) >> 6;
  int count = 0;
  for (int wi = (int)(threadIdx.x & 63u); wi < nw; wi += 64) {
    const uint64_t a = (__hip_atomic_load(op.esc + wi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & ~__hip_atomic_load(op.esc + 2 * nw + wi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) |
                       __hip_atomic_load(op.esc + nw + wi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    op.esc[wi] = a;
    op.esc[nw + wi] = 0;
    op.esc[2 * nw + wi] = 0;
    count += __popcll(a);
  }
  count = wave_incl_scan(count);
  if ((threadIdx.x & 63u) == 63) { op.esc_ctr[1] = (uint32_t)count; op.esc_ctr[0] = 0; op.esc_ctr[2] = 0; }
}

// Contact detection against the scene's static plane (flags only): sample points of the collision geoms, link frame
struct CollTable {
  const double* xyzr;      // [npts][4]
  const uint8_t* cls;      // [npts] bit 0: geom is one of SimRobot's arm collision geoms; bit 1: SimGripper's
  int32_t link_adr[kMaxLinks + 1];
  double link_sphere[kMaxLinks][4];  // broad phase: bounding sphere of the link's points (link frame)
  double link_aabb[kMaxLinks][6];    // broad phase of the contact phase: bounding box of the link's points (centre, half extents)
  int32_t has_plane, has_static;     // has_static: link_aabb[NL] bounds collision geoms welded to the world (world frame)
  double plane_n[3], plane_d;
};

// Rendering callbacks (reference Sim::invoke_rendering_callbacks, src/sim/sim.cpp:63-81,108-115): cameras with a frame rate are
// due after a substep when more than 1 / frame_rate of simulated time has passed since their last frame.  A kernel cannot
// call the renderer, so the substep loop RECORDS what the renderer would have seen -- the qpos of that substep's position
// stage, the free box's pre-step pose, the new time, which cameras are due -- and the host renders the records after the
// launch (rcsh_camera_render_snapshot): same frames, same timestamps, no change to the stepping.
constexpr int kMaxRateCams = 4;
struct RendCfg {
  int32_t ncam, capacity;          // capacity: records per environment and launch
  double period[kMaxRateCams];     // seconds_between_calls
  double* last;                    // [kMaxRateCams][n] last_call_timestamp
  double* snap;                    // [capacity][nfields][n]; fields: qpre (NL), box pre-step pose (7), time, camera mask
  int32_t* count;                  // [n] records written by the last launch (more than capacity: the rest were dropped)
};

struct Params {
  const DevModel* model;  // HBM copy; each workgroup stages it into LDS once per launch
  CollTable coll;
  double* S;
  uint32_t* flags;
  int32_t* conv_steps;
  int32_t n;
  int32_t keep_qpre;  // a render scene exists: keep the qpos the last position stage saw (Lay::QPRE) for its frames
  SimCfg sim;
  RobotCfg robot;
  GripperCfg grip;
  EnvCfg env;
  const struct BoxTaskCfg* boxtask;  // scenes with a free box: its constants and the task layer's (HBM; staged by k_run_team<.., BOX>)
  ContactTable ctab;                 // the robot's collision geoms for the contact phase (contact_team.h; scenes with a free box)
  RendCfg rend;                      // rate-driven cameras (ncam = 0: none)
  CheckTable chk;                    // the once-per-launch check for contacts nobody resolves (check_team.h)
};
struct BoxTaskCfg {
  BoxCfg box;
  TaskCfg task;
};

// ---- everything one environment keeps in registers during a launch
template <class T, class ST>
struct EnvRegs {
  double time;
  double last_cmd_width, last_width;
  // callback timestamps, previous_angles and target_angles are touched once per 25-50 substeps:
  // they live in the lane's LDS column (Stage::X), not in registers
  ST st;
  __device__ __forceinline__ double& cb(int i) const { return st.X(i); }
  __device__ __forceinline__ double& prevq(int i) const { return st.X(6 + i); }
  __device__ __forceinline__ double& target(int i) const { return st.X(6 + T::NARM + i); }
  uint32_t flags;
  int32_t conv_steps;
};

template <class T, class ST>
__device__ __forceinline__ void load_env(const Params& P, int e, EnvRegs<T, ST>& r) {
  using L = Lay<T>;
  const int n = P.n;
  const double* S = P.S;
#pragma unroll
  for (int i = 0; i < T::NL; ++i) { r.st.q(i) = S[(L::QPOS + i) * n + e]; r.st.v(i) = S[(L::QVEL + i) * n + e]; r.st.qpre(i) = P.keep_qpre ? S[(L::QPRE + i) * n + e] : 0.0; }
#pragma unroll
  for (int i = 0; i < T::NU; ++i) r.st.c(i) = S[(L::CTRL + i) * n + e];
  r.time = S[L::TIME * n + e];
#pragma unroll
  for (int i = 0; i < 6; ++i) r.cb(i) = S[(L::CB + i) * n + e];
#pragma unroll
  for (int i = 0; i < T::NARM; ++i) { r.prevq(i) = S[(L::PREVQ + i) * n + e]; r.target(i) = S[(L::TARGET + i) * n + e]; }
  r.last_cmd_width = S[(L::GRIP + 0) * n + e];
  r.last_width = S[(L::GRIP + 1) * n + e];
  r.flags = P.flags[e];
  r.conv_steps = P.conv_steps[e];
}

// kStaged false: only the fields EnvRegs holds in registers (the team kernel moves the LDS-staged ones with all
// 16 lanes of the team, team_staged_fields)
template <class T, class ST, bool kStaged = true>
__device__ __forceinline__ void store_env(const Params& P, int e, const EnvRegs<T, ST>& r) {
  using L = Lay<T>;
  const int n = P.n;
  double* S = P.S;
  if constexpr (kStaged) {
#pragma unroll
    for (int i = 0; i < T::NL; ++i) { S[(L::QPOS + i) * n + e] = r.st.q(i); S[(L::QVEL + i) * n + e] = r.st.v(i); }
    if (P.keep_qpre) {
#pragma unroll
      for (int i = 0; i < T::NL; ++i) S[(L::QPRE + i) * n + e] = r.st.qpre(i);
    }
#pragma unroll
    for (int i = 0; i < T::NU; ++i) S[(L::CTRL + i) * n + e] = r.st.c(i);
#pragma unroll
    for (int i = 0; i < 6; ++i) S[(L::CB + i) * n + e] = r.cb(i);
#pragma unroll
    for (int i = 0; i < T::NARM; ++i) { S[(L::PREVQ + i) * n + e] = r.prevq(i); S[(L::TARGET + i) * n + e] = r.target(i); }
  }
  S[L::TIME * n + e] = r.time;
  S[(L::GRIP + 0) * n + e] = r.last_cmd_width;
  S[(L::GRIP + 1) * n + e] = r.last_width;
  P.flags[e] = r.flags;
  P.conv_steps[e] = r.conv_steps;
}

// The state fields that live in the team's LDS block during a launch (qpos, qvel, ctrl, callback timestamps,
// previous / target angles), as a list: entry k is HBM field `field` <-> LDS slot `slot`.  The team kernel moves them
// with all 16 lanes (three rounds) instead of one lane issuing them one after the other.
template <class T, class ST>
struct TeamStagedFields {
  using L = Lay<T>;
  static constexpr int kCount = 3 * T::NL + T::NU + 6 + 2 * T::NARM;
  static constexpr int kRounds = (kCount + kTeamLanes - 1) / kTeamLanes;
  __device__ __forceinline__ static void locate(int k, int& field, int& slot) {
    if (k < T::NL) { field = L::QPOS + k; slot = ST::Q0 + k; return; }
    k -= T::NL;
    if (k < T::NL) { field = L::QVEL + k; slot = ST::V0 + k; return; }
    k -= T::NL;
    if (k < T::NU) { field = L::CTRL + k; slot = ST::C0 + k; return; }
    k -= T::NU;
    if (k < 6) { field = L::CB + k; slot = ST::X0 + k; return; }
    k -= 6;
    if (k < T::NARM) { field = L::PREVQ + k; slot = ST::X0 + 6 + k; return; }
    k -= T::NARM;
    if (k < T::NARM) { field = L::TARGET + k; slot = ST::X0 + 6 + T::NARM + k; return; }
    k -= T::NARM;
    field = L::QPRE + k; slot = ST::P0 + k;
  }
};

__device__ __forceinline__ void set_flag(uint32_t& f, uint32_t bit, bool v) { f = v ? (f | bit) : (f & ~bit); }

// SimGripper::get_normalized_width, reference src/sim/SimGripper.cpp:93-106
template <class T, class ST>
__device__ __forceinline__ double gripper_width(const Params& P, const EnvRegs<T, ST>& r) {
  const double qf = P.grip.finger ? r.st.q(T::NL - 1) : r.st.q(T::NL - 2);
  double w = (qf - P.grip.min_joint) / (P.grip.max_joint - P.grip.min_joint);
  return w < 0 ? 0 : (w > 1 ? 1 : w);
}

// Sim::invoke_callbacks, reference src/sim/sim.cpp:38-47, with SimRobot::is_arrived_callback /
// is_moving_callback (src/sim/SimRobot.cpp:156-170) as the two registered callbacks
template <class T, class ST>
__device__ __forceinline__ void plain_callbacks(const Params& P, EnvRegs<T, ST>& r) {
  if (!(P.robot.present && P.robot.conv_registered)) return;
  if (r.time - r.cb(0) > P.robot.period) {
    double mx = 0;
#pragma unroll
    for (int i = 0; i < T::NARM; ++i) mx = fmax(mx, fabs(r.st.q(i) - r.target(i)));
    set_flag(r.flags, kIsArrived, mx < P.robot.tolerance);
    r.cb(0) = r.time;
  }
  if (r.time - r.cb(1) > P.robot.period) {
    double mx = 0;
#pragma unroll
    for (int i = 0; i < T::NARM; ++i) { mx = fmax(mx, fabs(r.st.q(i) - r.prevq(i))); r.prevq(i) = r.st.q(i); }
    set_flag(r.flags, kIsMoving, mx > 0.0001);
    r.cb(1) = r.time;
  }
}

// Sim::invoke_condition_callbacks, reference src/sim/sim.cpp:14-23,49-61.  Callback bodies:
// SimRobot::collision_callback / convergence_callback (SimRobot.cpp:172-191),
// SimGripper::collision_callback / convergence_callback (SimGripper.cpp:108-130,143-151).
// Contacts: plane (floor) against the collision geoms only; geom-geom pairs are not detected in this revision.
// `contacts()` returns the plane-contact classes of the last position stage (bit 0 arm geoms, bit 1 gripper geoms).
template <class T, class ST, class HitFn>
__device__ __forceinline__ bool condition_callbacks(const DevModelHead& m, const Params& P, EnvRegs<T, ST>& r, HitFn&& contacts) {
  const bool has_g = T::GRIP && P.grip.present;
  const bool fire_r = P.robot.present && r.time - r.cb(2) > P.robot.period;
  const bool fire_g = has_g && r.time - r.cb(3) > P.grip.period;
  if (fire_r || fire_g) {
    const uint32_t hit = contacts();
    if (fire_r) {
      set_flag(r.flags, kRobotCollision, hit & 1u);
      set_flag(r.flags, kAnyRet0, hit & 1u);
      r.cb(2) = r.time;
    }
    if (fire_g) {
      set_flag(r.flags, kGripCollision, hit & 2u);
      set_flag(r.flags, kAnyRet1, hit & 2u);
      r.cb(3) = r.time;
    }
  }
  if (P.robot.present && P.robot.conv_registered && r.time - r.cb(4) > P.robot.period) {
    const bool conv = !(r.flags & kIkSuccess) || ((r.flags & kIsArrived) && !(r.flags & kIsMoving));
    set_flag(r.flags, kAllRet0, conv);
    r.cb(4) = r.time;
  }
  if (has_g && r.time - r.cb(5) > P.grip.period) {
    const double w = gripper_width<T, ST>(P, r);
    const bool moving = fabs(r.last_width - w) > 0.001 * (P.grip.max_act - P.grip.min_act);
    set_flag(r.flags, kGripMoving, moving);
    r.last_width = w;
    set_flag(r.flags, kAllRet1, !moving);
    r.cb(5) = r.time;
  }
  bool any = false, all = true;
  if (P.robot.present) any = any || (r.flags & kAnyRet0);
  if (has_g) any = any || (r.flags & kAnyRet1);
  if (P.robot.present && P.robot.conv_registered) all = all && (r.flags & kAllRet0);
  if (has_g) all = all && (r.flags & kAllRet1);
  return any || all;
}

// The smallest last-call timestamp among the condition callbacks that exist.  A callback fires when time - last > period, so
// nothing can fire while time - (this minimum) <= the smallest period: the leader lane keeps the minimum in a register and only
// walks the callbacks (four timestamps in LDS, a dozen compares) in the substeps where that test passes.
template <class T, class ST>
__device__ __forceinline__ double condition_callbacks_oldest(const Params& P, const EnvRegs<T, ST>& r) {
  const bool has_g = T::GRIP && P.grip.present;
  double lo = INFINITY;
  if (P.robot.present) lo = fmin(lo, r.cb(2));
  if (has_g) lo = fmin(lo, r.cb(3));
  if (P.robot.present && P.robot.conv_registered) lo = fmin(lo, r.cb(4));
  if (has_g) lo = fmin(lo, r.cb(5));
  return lo;
}

// SimRobot::set_joint_position, reference src/sim/SimRobot.cpp:123-131
template <class T, class ST>
__device__ __forceinline__ void robot_set_joint_position(EnvRegs<T, ST>& r, const double* a) {
#pragma unroll
  for (int i = 0; i < T::NARM; ++i) {
    r.target(i) = a[i];
    r.prevq(i) = r.st.q(i);
    r.st.c(i) = a[i];
  }
  r.flags = (r.flags | kIsMoving) & ~kIsArrived;
}

// SimGripper::set_normalized_width, reference src/sim/SimGripper.cpp:79-92
template <class T, class ST>
__device__ __forceinline__ void gripper_set_width(const Params& P, EnvRegs<T, ST>& r, double w) {
  r.last_cmd_width = w;
  r.st.c(T::NU - 1) = w * (P.grip.max_act - P.grip.min_act) + P.grip.min_act;
}

// SimRobot::get_cartesian_position, reference src/sim/SimRobot.cpp:114-121 + src/rcs/Robot.cpp:5-9
__device__ __forceinline__ void cartesian_position(const DevModelHead& m, const RobotCfg& rc, const double* linkR,
                                                   const double* linkP, Pose& out) {
  double sp[3], sR[9];
  mulmv(linkR, m.site_pos, sp);
  sp[0] += linkP[0]; sp[1] += linkP[1]; sp[2] += linkP[2];
  mulmm(linkR, m.site_rot, sR);
  Pose site, base, base_inv, in_robot, tcp;
  pose_from_mat(sR, sp, site);
  const double bq[4] = {m.base_quat[1], m.base_quat[2], m.base_quat[3], m.base_quat[0]};
  pose_from_quat(bq, m.base_pos, base);
  pose_inverse(base, base_inv);
  pose_mul(base_inv, site, in_robot);
  tcp.t[0] = rc.tcp[0]; tcp.t[1] = rc.tcp[1]; tcp.t[2] = rc.tcp[2];
  tcp.q[0] = rc.tcp[3]; tcp.q[1] = rc.tcp[4]; tcp.q[2] = rc.tcp[5]; tcp.q[3] = rc.tcp[6];
  pose_mul(in_robot, tcp, out);
}

// What env.step() reads per environment before the simulator is stepped: the action, the gripper command and the
// wrappers' remembered vectors.  StepInGlobal names where each value lives in HBM; the team kernel has its 16 lanes fetch all of
// it ahead of time into LDS (StepInStaged; entry k of the list is `fetch(k)`).
template <class T>
struct StepInGlobal {
  using L = Lay<T>;
  const Params& P;
  const RunOp& op;
  int e;
  static constexpr int kCount = 4 * T::NARM + 1;
  __device__ __forceinline__ double action(int i) const { return op.action[e * T::NARM + i]; }
  __device__ __forceinline__ double origin(int i) const { return P.S[(L::ORIGIN + i) * P.n + e]; }
  __device__ __forceinline__ double lasta(int i) const { return P.S[(L::LASTA + i) * P.n + e]; }
  __device__ __forceinline__ double preva(int i) const { return P.S[(L::PREVA + i) * P.n + e]; }
  __device__ __forceinline__ float gripper() const { return op.gripper[e]; }
  __device__ __forceinline__ double fetch(int k) const {
    if (k < T::NARM) return action(k);
    // origin / last action are only read back with RelativeTo.CONFIGURED_ORIGIN (LAST_STEP re-derives both every step)
    if (k < 3 * T::NARM && P.env.relative_to != 2) return 0.0;
    if (k < 2 * T::NARM) return origin(k - T::NARM);
    if (k < 3 * T::NARM) return lasta(k - 2 * T::NARM);
    if (k < 4 * T::NARM) return preva(k - 3 * T::NARM);
    return op.gripper ? (double)op.gripper[e] : 0.0;
  }
};
template <class T>
struct StepInStaged {
  const double* s;  // LDS, kCount entries in fetch() order
  __device__ __forceinline__ double action(int i) const { return s[i]; }
  __device__ __forceinline__ double origin(int i) const { return s[T::NARM + i]; }
  __device__ __forceinline__ double lasta(int i) const { return s[2 * T::NARM + i]; }
  __device__ __forceinline__ double preva(int i) const { return s[3 * T::NARM + i]; }
  __device__ __forceinline__ float gripper() const { return (float)s[4 * T::NARM]; }
};

// Wrappers' reset() / action() side effects on one environment: everything env.reset() / env.step() do before
// the simulator is stepped (reference python/rcs/envs/base.py, envs/sim.py; see the inline citations).
template <class T, class ST, class In>
__device__ __forceinline__ void env_prologue(const Params& P, const RunOp& op, const DevModelHead& m, int e, EnvRegs<T, ST>& r, const In& in) {
  using L = Lay<T>;
  const int n = P.n;
  if (op.do_reset) {
    // GripperWrapper.reset -> SimGripper::m_reset (python/rcs/envs/base.py:703-708, SimGripper.cpp:158-165)
    if (T::GRIP && P.grip.present) {
      r.last_cmd_width = 0; r.last_width = 0;
      r.flags &= ~(kGripMoving | kGripCollision | kHasGripCmd | kGripCmd);
    }
    // RobotSimWrapper.reset -> Sim::reset = mj_resetData + reset_callbacks (envs/sim.py:68-76, sim.cpp:117-138);
    // it overwrites what the gripper reset just wrote to qpos / ctrl
#pragma unroll
    for (int i = 0; i < T::NL; ++i) { r.st.q(i) = m.qpos0[i]; r.st.v(i) = 0; }
#pragma unroll
    for (int i = 0; i < T::NU; ++i) r.st.c(i) = 0;
    r.time = 0;
    r.flags &= ~(kContactOverflow | kContactUnresolved | kContactResolved | kEscQuiet);  // (sticky until Sim::reset: this is it)
#pragma unroll
    for (int i = 0; i < 6; ++i) r.cb(i) = 0;
    // RobotEnv.reset -> SimRobot::m_reset -> set_joints_hard(q_home) (base.py:290-304, SimRobot.cpp:193-205)
#pragma unroll
    for (int i = 0; i < T::NARM; ++i) { r.st.q(i) = P.robot.q_home[i]; r.st.c(i) = P.robot.q_home[i]; }
  }

  if (op.apply_action) {
    // ---- RelativeActionSpace.action (python/rcs/envs/base.py:468-488), JOINTS mode
    double a[T::NARM];
#pragma unroll
    for (int i = 0; i < T::NARM; ++i) a[i] = in.action(i);
    if (P.env.relative_to != 0) {
      const bool last_step = P.env.relative_to == 1;
      const bool fresh = last_step || !(r.flags & kHasLastAction);
#pragma unroll
      for (int i = 0; i < T::NARM; ++i) {
        double origin = last_step ? r.st.q(i) : in.origin(i);
        double lim;
        if (fresh) {
          lim = clampd(a[i], -P.env.max_mov[0], P.env.max_mov[0]);
        } else {
          const double la = in.lasta(i);
          lim = clampd(a[i] - la, -P.env.max_mov[0], P.env.max_mov[0]) + la;
        }
        if (last_step) P.S[(L::ORIGIN + i) * n + e] = origin;
        P.S[(L::LASTA + i) * n + e] = lim;
        a[i] = clampd(origin + lim, P.env.low[i], P.env.high[i]);
      }
      r.flags |= kHasLastAction;
    }
    // ---- GripperWrapper.action (base.py:721-735)
    if (T::GRIP && P.grip.present && op.gripper) {
      float g = in.gripper();
      if (P.env.binary_gripper) g = rintf(g);  // np.round: half to even
      g = fminf(fmaxf(g, 0.0f), 1.0f);
      if (P.env.binary_gripper) {
        gripper_set_width<T, ST>(P, r, g == 0.0f ? 0.0 : 1.0);  // grasp() = shut() : open()
        set_flag(r.flags, kGripCmd, g != 0.0f);
      } else {
        gripper_set_width<T, ST>(P, r, (double)g);
        set_flag(r.flags, kGripCmd, g >= 0.5f);
      }
      r.flags |= kHasGripCmd;
    }
    // ---- RobotEnv.step (base.py:255-288): command only when the action moved by more than atol = 1e-3
    bool changed = !(r.flags & kHasPrevAction);
#pragma unroll
    for (int i = 0; i < T::NARM; ++i) {
      const double pa = in.preva(i);
      changed = changed || !(fabs(a[i] - pa) <= 1e-3);
      P.S[(L::PREVA + i) * n + e] = a[i];
    }
    if (changed) robot_set_joint_position<T, ST>(r, a);
    r.flags |= kHasPrevAction;
  }

}

// The same on the team's lanes: lane i does joint i of everything that is per joint (the relative action's clamps and
// remembered vectors, the "did the action move" test, SimRobot::set_joint_position), the leader lane what is per environment
// (gripper, flags, clocks).  Per-joint inputs arrive in the lane's registers (in_*), the leader's flag word is spread to the team
// first.  One lane walking the seven joints through LDS took 5.1k cycles of every launch.  Every lane of a live team calls this.
template <class T, class ST>
__device__ __forceinline__ void env_prologue_team(const Params& P, const RunOp& op, const DevModelHead& m, int e, int t, EnvRegs<T, ST>& r,
                                                  double in_action, double in_origin, double in_lasta, double in_preva, float in_grip) {
  using L = Lay<T>;
  const int n = P.n;
  const bool leader = t == 0, joint = t < T::NARM;
  const int ti = joint ? t : 0, tl = t < T::NL ? t : 0, tu = t < T::NU ? t : 0;
  uint32_t flags = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(threadIdx.x & 48u) << 2, (int)r.flags);  // the leader's word, on every lane
  if (op.do_reset) {
    // GripperWrapper.reset -> SimGripper::m_reset (python/rcs/envs/base.py:703-708, SimGripper.cpp:158-165)
    if (T::GRIP && P.grip.present) {
      r.last_cmd_width = 0; r.last_width = 0;
      flags &= ~(kGripMoving | kGripCollision | kHasGripCmd | kGripCmd);
    }
    // RobotSimWrapper.reset -> Sim::reset = mj_resetData + reset_callbacks (envs/sim.py:68-76, sim.cpp:117-138);
    // it overwrites what the gripper reset just wrote to qpos / ctrl
    if (t < T::NL) { r.st.q(tl) = m.qpos0[tl]; r.st.v(tl) = 0; }
    if (t < T::NU) r.st.c(tu) = 0;
    r.time = 0;
    flags &= ~(kContactOverflow | kContactUnresolved | kContactResolved | kEscQuiet);  // (sticky until Sim::reset: this is it)
    if (t < 6) r.cb(t) = 0;
    // RobotEnv.reset -> SimRobot::m_reset -> set_joints_hard(q_home) (base.py:290-304, SimRobot.cpp:193-205)
    if (joint) { r.st.q(ti) = P.robot.q_home[ti]; r.st.c(ti) = P.robot.q_home[ti]; }
  }
  if (op.apply_action) {
    // ---- RelativeActionSpace.action (python/rcs/envs/base.py:468-488), JOINTS mode
    double a = in_action;
    if (P.env.relative_to != 0) {
      const bool last_step = P.env.relative_to == 1;
      const bool fresh = last_step || !(flags & kHasLastAction);
      if (joint) {
        const double origin = last_step ? r.st.q(ti) : in_origin;
        const double lim = fresh ? clampd(a, -P.env.max_mov[0], P.env.max_mov[0]) : clampd(a - in_lasta, -P.env.max_mov[0], P.env.max_mov[0]) + in_lasta;
        if (last_step) P.S[(size_t)(L::ORIGIN + ti) * n + e] = origin;
        P.S[(size_t)(L::LASTA + ti) * n + e] = lim;
        a = clampd(origin + lim, P.env.low[ti], P.env.high[ti]);
      }
      flags |= kHasLastAction;
    }
    // ---- GripperWrapper.action (base.py:721-735)
    if (T::GRIP && P.grip.present && op.gripper) {
      float g = in_grip;
      if (P.env.binary_gripper) g = rintf(g);  // np.round: half to even
      g = fminf(fmaxf(g, 0.0f), 1.0f);
      const double w = P.env.binary_gripper ? (g == 0.0f ? 0.0 : 1.0) : (double)g;  // binary: grasp() = shut() : open()
      if (leader) gripper_set_width<T, ST>(P, r, w);
      set_flag(flags, kGripCmd, P.env.binary_gripper ? g != 0.0f : g >= 0.5f);
      flags |= kHasGripCmd;
    }
    // ---- RobotEnv.step (base.py:255-288): command only when the action moved by more than atol = 1e-3
    const bool moved = joint && !(fabs(a - in_preva) <= 1e-3);
    const bool changed = !(flags & kHasPrevAction) || team_ballot(moved) != 0;
    if (joint) P.S[(size_t)(L::PREVA + ti) * n + e] = a;
    if (changed) {
      // SimRobot::set_joint_position, reference src/sim/SimRobot.cpp:123-131
      if (joint) { r.target(ti) = a; r.prevq(ti) = r.st.q(ti); r.st.c(ti) = a; }
      flags = (flags | kIsMoving) & ~kIsArrived;
    }
    flags |= kHasPrevAction;
  }
  r.flags = flags;  // (every lane carries the word; the leader's copy is the one that lives on)
}

// After stepping: park the site frame, refresh the relative-action origin on reset, write the state back and
// produce observation + info (RobotEnv.get_obs, GripperWrapper.observation, RobotSimWrapper.step, GripperWrapperSim).
template <class T, class ST, bool kStoreStaged = true>
__device__ __forceinline__ void env_epilogue(const Params& P, const RunOp& op, const DevModelHead& m, int e, EnvRegs<T, ST>& r,
                                             const ST& st, bool have_frames, int nsteps) {
  using L = Lay<T>;
  const int n = P.n;
  // (with frames: the team's lanes have parked the site link's frame already, one component per lane -- run_team_body)
  if (!have_frames && op.write_obs) {
#pragma unroll
    for (int k = 0; k < 12; ++k) st.link(k) = P.S[(L::SITE + k) * n + e];
  }
  if (op.do_reset && P.env.relative_to != 0) {
    // RelativeActionSpace.reset (base.py:461-466): origin := current, _last_action := None
    if (P.env.mode == 0) {
#pragma unroll
      for (int i = 0; i < T::NARM; ++i) P.S[(L::ORIGIN + i) * n + e] = r.st.q(i);
    } else {
      double oR[9], oP[3];
#pragma unroll
      for (int k = 0; k < 9; ++k) oR[k] = st.link(k);
#pragma unroll
      for (int k = 0; k < 3; ++k) oP[k] = st.link(9 + k);
      Pose origin;
      cartesian_position(m, P.robot, oR, oP, origin);
#pragma unroll
      for (int k = 0; k < 3; ++k) P.S[(L::ORIGIN + k) * n + e] = origin.t[k];
#pragma unroll
      for (int k = 0; k < 4; ++k) P.S[(L::ORIGIN + 3 + k) * n + e] = origin.q[k];
    }
    r.flags &= ~kHasLastAction;
  }
  store_env<T, ST, kStoreStaged>(P, e, r);

  if (op.write_obs) {
    // RobotEnv.get_obs (base.py:246-253) + GripperWrapper.observation (base.py:710-719) +
    // RobotSimWrapper.step info (envs/sim.py:60-66) + GripperWrapperSim.observation (envs/sim.py:125-131)
    constexpr int OW = 14 + T::NARM;
    Pose tcp;
    double linkR[9], linkP[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) linkR[k] = st.link(k);
#pragma unroll
    for (int k = 0; k < 3; ++k) linkP[k] = st.link(9 + k);
    cartesian_position(m, P.robot, linkR, linkP, tcp);
    double* o = op.obs + (size_t)e * OW;
    o[0] = tcp.t[0]; o[1] = tcp.t[1]; o[2] = tcp.t[2];
    o[3] = tcp.q[0]; o[4] = tcp.q[1]; o[5] = tcp.q[2]; o[6] = tcp.q[3];
    // (o[7 .. 7 + NARM): the joints, written by the joints' lanes -- run_team_body)
    double rpy[3];
    pose_rpy(tcp, rpy);
    o[7 + T::NARM + 0] = tcp.t[0]; o[7 + T::NARM + 1] = tcp.t[1]; o[7 + T::NARM + 2] = tcp.t[2];
    o[7 + T::NARM + 3] = rpy[0]; o[7 + T::NARM + 4] = rpy[1]; o[7 + T::NARM + 5] = rpy[2];
    double gobs = 1.0, w = 0.0;
    const bool has_g = T::GRIP && P.grip.present;
    if (has_g) {
      w = gripper_width<T, ST>(P, r);
      if (P.env.binary_gripper) gobs = (r.flags & kHasGripCmd) ? ((r.flags & kGripCmd) ? 1.0 : 0.0) : 1.0;
      else gobs = w;
    }
    o[13 + T::NARM] = gobs;
    if (op.info) {
      const bool rc = r.flags & kRobotCollision, ik = r.flags & kIkSuccess, gc = has_g && (r.flags & kGripCollision);
      // the row's eight bytes in one store: collision, ik_success, is_sim_converged, is_grasped, truncated, gripper collision,
      // contact_overflow (a contact phase ran out of contact / link slots since the last Sim.reset), contact_unresolved (found in a
      // contact this configuration does not resolve, since the last Sim.reset)
      const uint64_t row = (uint64_t)(rc || gc) | (uint64_t)ik << 8 | (uint64_t)((r.flags & kConverged) != 0) << 16 | (uint64_t)(has_g && (w > 0.01 && w < 0.99)) << 24 |
                           (uint64_t)(rc || !ik) << 32 | (uint64_t)gc << 40 | (uint64_t)((r.flags & kContactOverflow) != 0) << 48 |
                           (uint64_t)((r.flags & kContactUnresolved) != 0) << 56;
      reinterpret_cast<uint64_t*>(op.info)[e] = row;
    }
    if (op.gripper_width) op.gripper_width[e] = w;
    if (op.substeps) op.substeps[e] = nsteps >= 0 ? nsteps : r.conv_steps;
  }
}

// Global -> LDS copy of NW 8-byte words by the 64 lanes of a workgroup, in two phases: `load` asks for all of a lane's words at once,
// `store` writes them.  A loop of load-then-store waits for memory once per trip (~700 cycles each at one wavefront per SIMD: the
// launch's tables took 8.5k cycles that way); several copies issue their loads together and wait once.
template <int NW>
struct LdsCopy {
  static constexpr int kTrips = (NW + 63) / 64;
  double w[kTrips];
  __device__ __forceinline__ void load(const void* src_) {
    const double* src = reinterpret_cast<const double*>(src_);
#pragma unroll
    for (int i = 0; i < kTrips; ++i) {
      const int k = (int)threadIdx.x + 64 * i;
      w[i] = src[k < NW ? k : 0];
    }
  }
  __device__ __forceinline__ void store(void* dst_) const {
    double* dst = reinterpret_cast<double*>(dst_);
#pragma unroll
    for (int i = 0; i < kTrips; ++i) {
      const int k = (int)threadIdx.x + 64 * i;
      if (k < NW) dst[k] = w[i];
    }
  }
};

// Stages the model into a workgroup's LDS for the team kernels (64 threads): the DevModelHead (what is not per link) and the
// per-link LinkRec records stored right behind the DevModel.  Every workgroup of the launch fetches these same lines from
// L2 at the same moment.
template <int NREC>
__device__ __forceinline__ void stage_team_model(const DevModel* gm, DevModelHead& lm, LinkRec* llinks) {
  constexpr int kWords = sizeof(DevModel) / 8, kHeadWords = sizeof(DevModelHead) / 8;
  const double* src = reinterpret_cast<const double*>(gm);
  double* dst = reinterpret_cast<double*>(&lm);
  for (int k = threadIdx.x; k < kHeadWords; k += 64) dst[k] = src[k];
  constexpr int kRecWords = sizeof(LinkRec) * NREC / 8;
  double* rdst = reinterpret_cast<double*>(llinks);
#pragma unroll
  for (int it = 0; it < (kRecWords + 63) / 64; ++it) {
    const int k = it * 64 + threadIdx.x;
    if (k < kRecWords) rdst[k] = src[kWords + k];
  }
}

// The N-environment form of Sim.step / Sim.step_until_convergence / env.reset / env.step: one TEAM of 16 lanes per
// environment (dyn_team.h), four environments per wavefront, one wavefront per workgroup.  Lane 0 of a team (the leader)
// owns the RCS bookkeeping -- wrappers, callback scheduler, observation; all 16 lanes run the physics.
// BOX: the scene has a free box (box_team.h).  CON: contacts of the robot's collision geoms are resolved (contact_team.h);
// without a box the contact phase sees a phantom one parked far above the scene (zero size: it touches nothing, its six
// dofs stay decoupled), so that one formulation serves the pick-up scene and the robot-on-the-floor case alike.
// DET: this launch runs the collision callbacks (step_until_convergence) of a model with collision geoms: in the substeps
// after which one of them is due the position stage also detects contacts -- the floor against the sample points of the
// collision geoms, and the robot's geoms against each other.  Launches that never look at the flags (Sim::step(k)) and
// models without collision geoms run the instantiation without that code.
// The body is a device function so that two kernels can share it: `k_run_team` (whatever registers it takes: one wavefront
// per SIMD, the headline batch of 4096 environments IS one wavefront per SIMD) and `k_run_team_occ2`, compiled for two resident
// wavefronts per SIMD (at most 256 VGPRs + AGPRs), for batches that bring more than one wavefront per SIMD.
template <class T, bool FRIC, bool BOX, bool CON, bool DET>
__device__ __attribute__((always_inline)) inline void run_team_body(const Params& Pk, const RunOp& opk) {
  using ST = StageTeam<T>;
  constexpr int kTeams = 64 / kTeamLanes;
  __shared__ DevModelHead lm;
  // The kernel arguments move to LDS too.  As arguments they sit in ~100 SGPRs that the leader-only code (wrappers,
  // callbacks, observation) keeps alive across the whole substep loop, and the loop then spills and reloads them
  // around its own scalar needs every iteration; from LDS they are read where that rare code runs.  (The
  // collision table inside is also indexed per lane, which a kernel argument cannot be.)
  __shared__ Params lp;
  __shared__ RunOp lop;
  // (sized for the end-of-launch contact check too, which takes the block over once the state has been written back)
  // (CON: the check's workspace is the contact arena, idle by then -- the instantiation has no LDS to spare)
  // (... and, kernels that carry neither the contact arena nor the detection's frames, for the certifying check's travel tables behind it)
  constexpr int kLdsMain = (CON || ST::COUNT * kTeams > check_work_doubles(T::NL)) ? ST::COUNT * kTeams : check_work_doubles(T::NL);
  constexpr int kLdsDoubles = kLdsMain + ((CON || DET) ? 0 : check_mv_doubles(T::NL));
  __shared__ __attribute__((aligned(16))) double lds[kLdsDoubles];
  // everything the launch reads from memory at its start is asked for at once -- arguments, model tables and, further down, the
  // environment's state -- and stored to LDS after one wait
  static_assert(sizeof(Params) % 8 == 0 && sizeof(RunOp) % 8 == 0 && sizeof(DevModelHead) % 8 == 0 && sizeof(LinkRec) % 8 == 0, "copied in 8-byte words");
  // (the contact-resolving launch of a step in which no environment is escalated and none has just been flagged -- every step of a
  // rollout that touches nothing -- ends here, before a single load has gone out: two scalar loads; a wavefront that has asked for
  // its tables waits for them before it may end, 4.5 us for the launch instead of 1.5)
  if constexpr (CON) {
    if (opk.esc_role == 2 && opk.esc_ctr[1] == 0 && (opk.esc_part == 1 || opk.esc_ctr[2] == 0)) return;
  }
  LdsCopy<sizeof(Params) / 8> cp_params;
  LdsCopy<sizeof(RunOp) / 8> cp_op;
  LdsCopy<sizeof(DevModelHead) / 8> cp_head;
  LdsCopy<sizeof(LinkRec) * T::NL / 8> cp_links;
  cp_params.load(&Pk);
  cp_op.load(&opk);
  cp_head.load(Pk.model);
  cp_links.load(reinterpret_cast<const char*>(Pk.model) + sizeof(DevModel));
  const Params& P = lp;
  const RunOp& op = lop;
  const CollTable& lc = lp.coll;
  TEAM_CLOCK_START()
#ifdef RCSH_WAVE_TIMES
  // development (tools/wave_times.py): when this wavefront started, left its substep loop, entered the end-of-launch check and ended --
  // the constant 100 MHz clock all XCDs share
  const unsigned long long wt0_ = __builtin_amdgcn_s_memrealtime();
  unsigned long long wt1_ = 0, wt2_ = 0;
#endif
#ifdef RCSH_PHASE_TIMING
  const unsigned long long wg_clock0 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) for (int k_ = 0; k_ < 8; ++k_) s_wg_acc[k_] = 0;
#endif
  const int team = threadIdx.x / kTeamLanes, t = threadIdx.x % kTeamLanes;
  // Workgroups are dealt round-robin to the 8 XCDs (workgroup b runs on XCD b % 8), each with its own L2.  Give
  // every XCD one contiguous range of environments, so a 128-byte line of a state field ([field][env], 16
  // environments) is fetched into one L2 instead of four.  The grid is rounded up to a multiple of 8.
  const int per_xcd = gridDim.x / 8;
  const int e_slot0 = ((blockIdx.x % 8) * per_xcd + blockIdx.x / 8) * kTeams;
  int e = e_slot0 + team;
  bool esc_redo = false;  // role 2: this environment was flagged by the lean launch of this step; its step is redone from the copy
  const int esc_role = opk.esc_role;  // (wave-uniform)
  if (esc_role != 0) {
    if (esc_role == 2) {
      // (role 2 may spread its environments one per workgroup: every workgroup asks)
      const int pick = esc_select(opk, Pk.n, e_slot0, team, esc_redo, true);
      if (__ballot(pick >= 0) == 0) {  // no environment for any of this workgroup's four slots
        if (opk.esc_part != 1) esc_finish(opk, Pk.n);
        return;
      }
      e = pick >= 0 ? pick : Pk.n;
    }
  }
  // Role 1 (the lean launch) keeps the plain slot -> environment map and sits an escalated environment's team out: the word of the
  // escalated mask is asked for together with the state, so the launch does not begin with a memory round trip of its own (ranking the
  // environments that are not escalated, as role 2 ranks those that are, cost the headline 10 us a step: one dependent load before the
  // state could be asked for, and the copy's stores -- which need the state -- ahead of the model staging).  The team's loads go out
  // whether or not it is escalated; `live` is settled when they come back.
  const bool in_range = e < Pk.n && !(opk.mask && !opk.mask[e < Pk.n ? e : 0]);
  uint64_t esc_word = 0;
  if (esc_role == 1 && in_range) esc_word = opk.esc[e >> 6];
  // where this launch reads the environment's state from: its state, or (a step that is redone) the copy the lean launch kept
  const double* const Sin = esc_redo ? opk.snap : Pk.S;
  // The environment's state goes out first -- every lane asks for up to three of the fields that are staged in LDS,
  // the leader for the five it keeps in registers -- so that the model staging below hides the round trip to HBM.
  using SF = TeamStagedFields<T, ST>;
  const int nstaged = Pk.keep_qpre ? SF::kCount : SF::kCount - T::NL;  // (the QPRE entries are the list's last)
  double staged[SF::kRounds];
  // env.step()'s inputs, per joint on the joint's lane: the action, the relative action space's remembered vectors, the previous action
  double in_action = 0.0, in_origin = 0.0, in_lasta = 0.0, in_preva = 0.0;
  float in_grip = 0.0f;
  double pre_time = 0, pre_cmd = 0, pre_width = 0;
  uint32_t pre_flags = 0;
  int32_t pre_conv = 0;
  double xs_in = 0.0;
  // the contact check's remembered directions (lanes 0..7 of a team; check_team.h): asked for with the state, used after the last substep
  const double sep_in = in_range && t < kCheckSep && !opk.do_reset ? Pk.S[(size_t)(Lay<T>::SEP + t) * Pk.n + e] : 0.0;
  if (in_range) {
#pragma unroll
    for (int rd = 0; rd < SF::kRounds; ++rd) {
      const int k = t + rd * kTeamLanes;
      int field = 0, slot = 0;
      SF::locate(k < nstaged ? k : 0, field, slot);
      staged[rd] = Sin[(size_t)field * Pk.n + e];
    }
    if constexpr (FRIC) {
      if (t < T::NL && !opk.do_reset) xs_in = Sin[(size_t)(Lay<T>::XS + t) * Pk.n + e];  // (Sim::reset: mj_resetData zeroes the warm start)
    }
    if (opk.apply_action) {
      using L = Lay<T>;
      if (t < T::NARM) {
        in_action = opk.action[(size_t)e * T::NARM + t];
        // (origin / last action are only read back with RelativeTo.CONFIGURED_ORIGIN: LAST_STEP re-derives both every step)
        if (Pk.env.relative_to == 2) { in_origin = Sin[(size_t)(L::ORIGIN + t) * Pk.n + e]; in_lasta = Sin[(size_t)(L::LASTA + t) * Pk.n + e]; }
        in_preva = Sin[(size_t)(L::PREVA + t) * Pk.n + e];
      }
      if (t == 0 && opk.gripper) in_grip = opk.gripper[e];
    }
    if (t == 0) {
      using L = Lay<T>;
      pre_time = Sin[(size_t)L::TIME * Pk.n + e];
      pre_cmd = Sin[(size_t)(L::GRIP + 0) * Pk.n + e];
      pre_width = Sin[(size_t)(L::GRIP + 1) * Pk.n + e];
      pre_flags = esc_redo ? opk.snap_flags[e] : Pk.flags[e];
      pre_conv = esc_redo ? opk.snap_conv[e] : Pk.conv_steps[e];
    }
  }
  __shared__ LinkRec llinks[T::NL];  // per-link records, stored behind the DevModel (model.h)
  __shared__ std::conditional_t<(BOX || CON), BoxTaskCfg, char> lbt[1];
  {
    if constexpr (BOX || CON) {
      static_assert(sizeof(BoxTaskCfg) % 8 == 0, "copied in 8-byte words");
      LdsCopy<sizeof(BoxTaskCfg) / 8> cp_bt;
      cp_bt.load(Pk.boxtask);
      cp_bt.store(&lbt[0]);
    }
    cp_params.store(&lp);
    cp_op.store(&lop);
    cp_head.store(&lm);
    cp_links.store(&llinks[0]);
    for (int k = threadIdx.x; k < kLdsDoubles; k += 64) lds[k] = 0.0;
    __syncthreads();
  }
  TEAM_MARK(12)
  const bool live = in_range && !((esc_word >> (e & 63)) & 1ull);
  const bool leader = t == 0 && live;
  if (esc_role == 1 && __ballot(live) == 0) {  // (every environment of this workgroup is on the contact-resolving kernel, or out of range)
    if (blockIdx.x == 0 && threadIdx.x == 0 && opk.esc_host) {  // (the host's hint: see the launch's end)
      __hip_atomic_store(opk.esc_host + 1, opk.esc_ctr[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(opk.esc_host, (uint32_t)opk.esc_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  if (esc_role == 1 && live && opk.snap) {
    // per-environment escalation, the lean launch: everything just read that this launch will overwrite, kept for the case that
    // the step has to be redone with its contacts resolved (RunOp::esc_role)
    using L = Lay<T>;
#pragma unroll
    for (int rd = 0; rd < SF::kRounds; ++rd) {
      const int k = t + rd * kTeamLanes;
      int field = 0, slot = 0;
      SF::locate(k < nstaged ? k : 0, field, slot);
      if (k < nstaged) opk.snap[(size_t)field * Pk.n + e] = staged[rd];
    }
    if constexpr (FRIC) {
      if (t < T::NL && !opk.do_reset) opk.snap[(size_t)(L::XS + t) * Pk.n + e] = xs_in;
    }
    if (opk.apply_action && t < T::NARM) {
      if (Pk.env.relative_to == 2) { opk.snap[(size_t)(L::ORIGIN + t) * Pk.n + e] = in_origin; opk.snap[(size_t)(L::LASTA + t) * Pk.n + e] = in_lasta; }
      opk.snap[(size_t)(L::PREVA + t) * Pk.n + e] = in_preva;
    }
    if (t == 0) {
      opk.snap[(size_t)L::TIME * Pk.n + e] = pre_time;
      opk.snap[(size_t)(L::GRIP + 0) * Pk.n + e] = pre_cmd;
      opk.snap[(size_t)(L::GRIP + 1) * Pk.n + e] = pre_width;
      opk.snap_flags[e] = pre_flags;
      opk.snap_conv[e] = pre_conv;
    }
  }
  const DevModelHead& m = lm;
  const ST st{lds + team * ST::COUNT};
  EnvRegs<T, ST> r;  // meaningful on the leader only
  r.st = st;
  r.time = 0; r.last_cmd_width = 0; r.last_width = 0; r.flags = 0; r.conv_steps = 0;
  bool have_frames = false;
  // env.reset() steps once (RobotSimWrapper.reset, envs/sim.py:72-79); with RandomCubePos under it twice, the box
  // being placed in between (sim.py:365-383): super().reset(), sim.step(1), qpos of box_joint := ..., then the
  // RobotSimWrapper's own sim.step(1)
  const bool place_box = BOX && op.do_reset && op.box_qpos != nullptr;
  int nsteps = op.do_reset ? (place_box ? 2 : 1) : op.nsteps;
  const bool until_conv = nsteps < 0;
  int budget = 0;
  bool converged = false;
  if (live) {
#pragma unroll
    for (int rd = 0; rd < SF::kRounds; ++rd) {
      const int k = t + rd * kTeamLanes;
      int field = 0, slot = 0;
      SF::locate(k < nstaged ? k : 0, field, slot);
      if (k < nstaged) st.at(slot) = staged[rd];
    }
    if constexpr (FRIC) {
      if (t < T::NL) st.xs(t) = xs_in;
    }
  }
  __syncthreads();
  if (leader) { r.time = pre_time; r.last_cmd_width = pre_cmd; r.last_width = pre_width; r.flags = pre_flags; r.conv_steps = pre_conv; }
  TEAM_MARK(13)
  if (live) {
    in_grip = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((int)(threadIdx.x & 48u) << 2, __builtin_bit_cast(int, in_grip)));
    env_prologue_team<T, ST>(Pk, opk, m, e, t, r, in_action, in_origin, in_lasta, in_preva, in_grip);  // (Pk, opk: still in registers here)
  }
  TEAM_MARK(14)
  if (leader) {
    budget = nsteps;
    if (until_conv) {
      const int cap = P.sim.max_convergence_steps;
      if (!opk.conv_resume) {
        r.conv_steps = 0;
        r.flags &= ~(kConverged | kAnyRet0 | kAnyRet1 | kAllRet0 | kAllRet1);
        budget = cap == -1 ? 0x7fffffff : cap;
      } else {
        // (a later piece: whoever converged stays put, the others go on with what the cap leaves them)
        converged = (r.flags & kConverged) != 0;
        budget = converged ? 0 : (cap == -1 ? 0x7fffffff : cap - r.conv_steps);
      }
      if (opk.conv_chunk > 0 && budget > opk.conv_chunk) budget = opk.conv_chunk;
    }
  }
  // Which teams still step is decided by their leaders and spread with a ballot (no LDS flag, no barrier); the
  // two plain callbacks fire every `period` of simulated time, so the leader only looks at them when the earlier
  // of their two timestamps is due.
  // the free box of the scene: its state sits in the team's LDS block between substeps (box_team.h)
  // (a free box's block: its state, the rows of its floor contacts, its acceleration; the PHANTOM box of a scene without one -- zero size,
  // parked a kilometre up, there so that one formulation serves both -- keeps its state only: 2.4 KB that let four workgroups of the
  // contact-resolving kernel share a CU again with 64 contact slots)
  constexpr int kBoxStride = BOX ? kBoxLds : kBoxState + 1;
  __shared__ double lbox[(BOX || CON) ? kBoxStride * kTeams : 1];
  using Arena = ContactArena<T, BOX ? kMaxCon : kMaxConNoBox>;
  __shared__ std::conditional_t<CON, Arena, char> larena[1];  // the contact phase's workspace: one per wavefront
  // self collision (DET): the world frames of every team's links for the pair tests.  The contact phase's workspace is idle
  // while the position stage runs (it is rebuilt from scratch whenever the contact phase starts), so with CON the frames
  // borrow it: that instantiation has no LDS to spare (38.7 of the 40 KB that let four workgroups share a CU).
  constexpr int kSelfF = 12 * T::NL;
  __shared__ double lself[(DET && !CON) ? kSelfF * kTeams + kSelfStage + kSelfCache + kSelfTag : 1];
  static_assert(!CON || sizeof(Arena) >= sizeof(double) * (kSelfF * kTeams + kSelfStage + kSelfCache + kSelfTag), "link frames and hull stage fit into the contact arena");
  double* const selfAll = (DET && CON) ? reinterpret_cast<double*>(&larena[0]) : lself;
  double* const selfF = selfAll + (DET ? team * kSelfF : 0);
  const int npair = DET ? lp.ctab.npair : 0;
  // (lean kernels without a free box only: the others have no LDS to spare and read the pair records from memory)
  constexpr bool kSelfLds = DET && !CON && !BOX;
  static_assert(!kSelfLds || T::NL < kSlackJ, "the slack cache keeps NL joints and NL + 1 prefix sums per team");
  __shared__ float lsph[kSelfLds ? kMaxSelfPairs * kSelfSphereWords : 1];
  const float* const sph = (kSelfLds && npair <= kMaxSelfPairs) ? lsph : nullptr;
  __shared__ std::conditional_t<kSelfLds, SelfSlack, char> lslack[1];
  SelfSlack* slack = nullptr;
  if constexpr (kSelfLds) {
    if (sph) {
      self_sphere_table_fill(lp.ctab.pairs, npair, lsph, T::NARM);
      self_slack_clear(lslack[0]);
      slack = &lslack[0];
    }
  }
  if constexpr (DET && !CON) {
    if (threadIdx.x == 0) lself[kSelfF * kTeams + kSelfStage + kSelfCache] = 0.0;  // nothing staged yet
  }
  double* const bs = lbox + ((BOX || CON) ? team * kBoxStride : 0);
  if constexpr (CON && !BOX) {
    // the phantom box: at rest where the host parked it; what it carries from launch to launch is the minimiser of the last coupled
    // solve (kBoxX: the warm start of the next one -- mjData.qacc_warmstart outlives a Sim.step call, only Sim::reset zeroes it)
    for (int k = t; k < kBoxState; k += kTeamLanes) {
      double v = k < 7 ? lbt[0].box.qpos0[k] : (k >= kBoxPre && k < kBoxPre + 7 ? lbt[0].box.qpos0[k - kBoxPre] : 0.0);
      if (k >= kBoxX && live && !op.do_reset) v = P.S[(size_t)(Lay<T>::BOX + k) * P.n + e];
      bs[k] = v;
    }
  }
  if constexpr (BOX) {
    if (live) {
      using L = Lay<T>;
      for (int k = t; k < kBoxState; k += kTeamLanes)
        bs[k] = op.do_reset ? (k < 7 ? lbt[0].box.qpos0[k] : (k >= kBoxPre && k < kBoxPre + 7 ? lbt[0].box.qpos0[k - kBoxPre] : 0.0))
                            : P.S[(size_t)(L::BOX + k) * P.n + e];  // Sim::reset: mj_resetData
    }
  }
  // rate-driven cameras: the teams' camera clocks and record counters (RendCfg)
  __shared__ double lrend[kTeams][kMaxRateCams + 2];
  const int rend_ncam = lp.rend.ncam;
  const bool rend_on = rend_ncam > 0 && !op.observe_only;  // (wave-uniform)
  if (rend_on && leader) {
    // Sim::reset -> reset_callbacks (sim.cpp:131-137): "negative so that we will directly render the cameras in the first step"
    for (int c = 0; c < rend_ncam; ++c) {
      // (a step that is redone starts from the camera clocks the lean launch started from: kept behind the state's copy)
      const double last = esc_redo ? lop.snap[(size_t)(Lay<T>::COUNT + c) * P.n + e] : lp.rend.last[(size_t)c * P.n + e];
      if (esc_role == 1 && lop.snap) lop.snap[(size_t)(Lay<T>::COUNT + c) * P.n + e] = last;
      lrend[team][c] = op.do_reset ? -lp.rend.period[c] : last;
    }
    lrend[team][kMaxRateCams + 1] = 0.0;
  }
  if (rend_on && t == 0 && e < P.n && !live) lp.rend.count[e] = 0;  // (masked out of this launch: nothing recorded)
  bool esc_contact = false;  // (leader) a substep of this launch resolved a contact of the robot's geoms
  bool box_placed = false;  // env.reset() with RandomCubePos: the box got its pose after the first of the two substeps
  bool more = leader && budget > 0;
  double cb_due = leader ? fmin(r.cb(0), r.cb(1)) : 0.0;
  // (step_until_convergence: the condition callbacks' oldest timestamp and shortest period, see condition_callbacks_oldest)
  double cond_lo = leader && until_conv ? condition_callbacks_oldest<T, ST>(P, r) : 0.0;
  bool cond_first = true;
  const bool has_cb = P.robot.present && P.robot.conv_registered;
  // the few scalars the loop control reads every substep, out of LDS once
  const double robot_period = P.robot.period, grip_period = P.grip.period, timestep = m.timestep;
  const bool robot_present = P.robot.present, grip_present = T::GRIP && P.grip.present, has_plane = lc.has_plane;
  const double cond_pmin = robot_present ? (grip_present ? fmin(robot_period, grip_period) : robot_period) : (grip_present ? grip_period : INFINITY);
  uint64_t going = __ballot(more);
  const bool gc_is_mass = team_gc_is_mass<T>(llinks, t);
  SubstepK sk;
  sk.load(m);
  // broad phase of the contact phase, per lane: the link's bounding box, the floor plane, the box's bounding sphere
  double bbc[3] = {0, 0, 0}, bbh[3] = {0, 0, 0}, pln[3] = {0, 0, 1}, pld = 0, box_r2 = 0;
  bool con_lane = false;
  if constexpr (CON) {
    // (lane NL: the geoms welded to the world -- link 0's hull -- against the free body; their frame is the world's)
    const bool static_lane = BOX && t == T::NL && lc.has_static;
    con_lane = (t < T::NL || static_lane) && lp.ctab.ngeom > 0;
    const int tl = t <= T::NL ? t : 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) { bbc[k] = lc.link_aabb[tl][k]; bbh[k] = lc.link_aabb[tl][3 + k]; pln[k] = lc.plane_n[k]; }
    pld = lc.plane_d;
    const BoxCfg& bc = lbt[0].box;
    box_r2 = bc.size[0] * bc.size[0] + bc.size[1] * bc.size[1] + bc.size[2] * bc.size[2];
  }
  TEAM_MARK(11)
  // where every joint has been over this launch's substeps -- first position, lowest, highest (the certifying check, check_team.h: a
  // contact that begins AND ends inside the launch needs the geoms to cover their gaps on the way)
  double chk_qmin = 0.0, chk_qmax = 0.0, chk_q0 = 0.0;
  bool chk_first = true;
  while (going) {
    const bool stepping = (going >> (threadIdx.x & 48)) & 1u;
    if (op.check == 2 && t < T::NL) {
      const double qn = st.q(t);
      chk_q0 = chk_first ? qn : chk_q0;
      // (the two fingers' lanes follow the gripper's opening q0 + q1 and the fingers' common shift q0 - q1 instead of their own joints:
      // shut fingers shift together by far more than the gap between their pads changes -- check_team.h: obb_face_sep_slides)
      double wn = qn;
      if (T::GRIP && t >= T::NARM) wn = t == T::NARM ? st.q(T::NARM) + st.q(T::NARM + 1) : st.q(T::NARM) - st.q(T::NARM + 1);
      chk_qmin = chk_first ? wn : fmin(chk_qmin, wn);
      chk_qmax = chk_first ? wn : fmax(chk_qmax, wn);
      chk_first = false;
    }
    if (leader && stepping && has_cb && r.time - cb_due > robot_period) {
      plain_callbacks<T, ST>(P, r);
      cb_due = fmin(r.cb(0), r.cb(1));
    }
    __syncthreads();
    TEAM_MARK(8)
    // plane contacts of the position stage: every lane tests its own link with the frame the substep just built,
    // in the substeps after which a collision callback is due (condition_callbacks, same comparisons)
    bool due = false;
    if constexpr (DET) {
      if (leader && stepping && until_conv && (has_plane || npair > 0)) {
        const double t_next = r.time + timestep;
        if (t_next - cond_lo > cond_pmin)  // (necessary for any callback to fire; the exact tests read the timestamps from LDS)
          due = (robot_present && t_next - r.cb(2) > robot_period) || (grip_present && t_next - r.cb(3) > grip_period);
      }
    }
    const bool team_due = DET && team_ballot(due) != 0;
    const bool want_contacts = DET && __ballot(due) != 0;  // (the wavefront detects together: self_collision_pairs)
    uint32_t hit = 0, overflow = 0, det_mine = 0;
    bool near = false;  // broad phase of the contact phase: the lane's link may touch the floor or the box
    bool team_coupled = false;  // the contact phase solved this substep's constraints for robot and box together
    team_substep<T, FRIC>(m, sk, llinks, st, t, stepping, gc_is_mass, [&](const double* R, const double* p) {
      if constexpr (CON && !DET) {
        // (the contact-resolving launch of per-environment escalation: team 0's environment is the first -- in a spread launch the only --
        // one the contact phase takes, and the link frames its collision pass begins with are these: parked where it reads them, the
        // pass skips its own forward kinematics.  Nothing uses the arena between here and that pass.)
        if (esc_role == 2 && team == 0 && t < T::NL && stepping) {
          double (*F)[12] = larena[0].frames();
#pragma unroll
          for (int k = 0; k < 9; ++k) F[t][k] = R[k];
#pragma unroll
          for (int k = 0; k < 3; ++k) F[t][9 + k] = p[k];
        }
      }
      if constexpr (CON) {
        if (stepping && con_lane && t < T::NL) {
          // the link's bounding box (link frame) against the floor -- its support along the plane normal -- and against
          // the box's bounding sphere: tight enough that an arm in its workspace does not wake the contact phase
          double c[3];
          mulmv(R, bbc, c);
          c[0] += p[0]; c[1] += p[1]; c[2] += p[2];
          if constexpr (BOX) {
            const double d[3] = {bs[kBoxQ] - c[0], bs[kBoxQ + 1] - c[1], bs[kBoxQ + 2] - c[2]};
            double v[3];
            mulTv(R, d, v);
            const double ex = fmax(fabs(v[0]) - bbh[0], 0.0), ey = fmax(fabs(v[1]) - bbh[1], 0.0), ez = fmax(fabs(v[2]) - bbh[2], 0.0);
            near = ex * ex + ey * ey + ez * ez <= box_r2;
          }
          if (has_plane) {
            double nl[3];
            mulTv(R, pln, nl);
            near = near || dot3(pln, c) - pld - (fabs(nl[0]) * bbh[0] + fabs(nl[1]) * bbh[1] + fabs(nl[2]) * bbh[2]) <= 0;
          }
          // second level: the same tests on the boxes of the link's geoms
          if (near)
            near = geom_level_near(lp.ctab.geoms, lp.ctab.link_geom_adr[t], lp.ctab.link_geom_adr[t + 1], R, p, pln, pld, has_plane, bs, box_r2, BOX, lbt[0].box.size);
        }
        if constexpr (BOX) {
          if (stepping && con_lane && t == T::NL) {
            // the lane of the world-welded geoms (link 0's hull): their box is in the world frame; no floor test (MuJoCo
            // filters that pair)
            const double ex = fmax(fabs(bs[kBoxQ] - bbc[0]) - bbh[0], 0.0), ey = fmax(fabs(bs[kBoxQ + 1] - bbc[1]) - bbh[1], 0.0),
                         ez = fmax(fabs(bs[kBoxQ + 2] - bbc[2]) - bbh[2], 0.0);
            if (ex * ex + ey * ey + ez * ez <= box_r2) {
              const double I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, z3[3] = {0, 0, 0};
              near = geom_level_near(lp.ctab.geoms, 0, lp.ctab.link_geom_adr[0], I9, z3, pln, pld, false, bs, box_r2, true, lbt[0].box.size);
            }
          }
        }
      }
      if constexpr (DET && !CON) {
        // (flag-only launches: the lane publishes its link's frame and that is all that happens HERE -- the floor test and the pair
        // test read the frames back after the substep: code in the middle of the substep, taken or not, shapes the register
        // allocation of the whole loop)
        if (!want_contacts) return;
        if (t < T::NL) {
#pragma unroll
          for (int k = 0; k < 9; ++k) selfF[12 * t + k] = R[k];
#pragma unroll
          for (int k = 0; k < 3; ++k) selfF[12 * t + 9 + k] = p[k];
        }
      }
      if constexpr (DET && CON) {
        if (!want_contacts) return;
        uint32_t mine = 0;
        if (has_plane) {
          const double* nrm = lc.plane_n;
          const double a[3] = {R[0] * nrm[0] + R[3] * nrm[1] + R[6] * nrm[2], R[1] * nrm[0] + R[4] * nrm[1] + R[7] * nrm[2],
                               R[2] * nrm[0] + R[5] * nrm[1] + R[8] * nrm[2]};
          const double b = dot3(nrm, p) - lc.plane_d;
          const int tl = t < T::NL ? t : T::NL - 1;
          const double* sph = lc.link_sphere[tl];
          if (t < T::NL && b + a[0] * sph[0] + a[1] * sph[1] + a[2] * sph[2] - sph[3] < 0) {
            for (int k = lc.link_adr[tl]; k < lc.link_adr[tl + 1]; ++k) {
              const double* v = lc.xyzr + 4 * (size_t)k;
              if (b + a[0] * v[0] + a[1] * v[1] + a[2] * v[2] - v[3] < 0) mine |= lc.cls[k];
            }
          }
        }
        if (npair > 0) {
          // the robot's geoms against each other: every lane publishes its link's frame, the due teams' lanes take their
          // share of the pairs, the wavefront settles the survivors together
          if (t < T::NL) {
#pragma unroll
            for (int k = 0; k < 9; ++k) selfF[12 * t + k] = R[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) selfF[12 * t + 9 + k] = p[k];
          }
          stage_fence();  // (LDS traffic of one wavefront is ordered; the fence is for the compiler)
          // (frames and stage sit in the contact arena, which the contact phase may enter next: the pairs are tested here)
          mine |= self_collision_pairs(lp.ctab.geoms, lp.ctab.verts, lp.ctab.pairs, npair, selfAll, selfAll + kSelfF * kTeams, T::NL, team_due, !CON, sph, slack,
                                       lp.ctab.self_lever, st.q(t < T::NL ? t : T::NL - 1));
          stage_fence();
        }
        hit = (team_ballot(mine & 1u) ? 1u : 0u) | (team_ballot(mine & 2u) ? 2u : 0u);
      }
    }, [&]() -> bool {
      bool coupled = false;
      if constexpr (CON) {
        // teams whose broad phase fired take turns: the whole wavefront works on one environment's contacts (contact_team.h)
        // (an escalated environment -- RunOp::esc_role 2 -- runs the contact phase in every substep: all of MuJoCo's collision pass,
        // the robot's geoms against each other included, none of the fast path's broad phase)
        const uint64_t nearw = __ballot(near || ((esc_role == 2 || lop.force_contact) && stepping && t == 0));
#ifdef RCSH_PHASE_TIMING
        if (threadIdx.x == 0) {  // (all workgroups) substeps of wavefronts / with a woken contact phase / teams woken
          atomicAdd(&g_team_cycles[41], 1ull);
          if (nearw) { atomicAdd(&g_team_cycles[40], 1ull); atomicAdd(&g_team_cycles[42], (unsigned long long)__popcll(nearw & 0x0001000100010001ull | (nearw >> 1 & 0) )); }
        }
#endif
        if (nearw) {
          for (int k = 0; k < kTeams; ++k) {
            if (!((nearw >> (k * kTeamLanes)) & 0xffffu)) continue;
            const uint32_t r = contact_phase<T, FRIC, BOX>(lp.ctab, lp.chk, lbt[0].box, llinks, ST{lds + k * ST::COUNT}, lbox + k * kBoxStride, larena[0], lm.gravity,
                                                      __builtin_amdgcn_readlane(e, k * kTeamLanes), !DET && esc_role == 2 && k == 0);
            if (team == k) {
              coupled = r & 1u;
              hit |= (r >> 8) & 3u;
              overflow |= (r >> 4) & 1u;
            }
          }
        }
      }
      team_coupled = coupled;
      return coupled;
    });
    if constexpr (DET && !CON) {
      if (want_contacts) {
        stage_fence();  // (LDS traffic of one wavefront is ordered; the fence is for the compiler)
        if (has_plane && t < T::NL) {
          // the floor against the sample points of the lane's link, on the frame the position stage published
          double R[9], p[3];
#pragma unroll
          for (int k = 0; k < 9; ++k) R[k] = selfF[12 * t + k];
#pragma unroll
          for (int k = 0; k < 3; ++k) p[k] = selfF[12 * t + 9 + k];
          const double* nrm = lc.plane_n;
          const double a[3] = {R[0] * nrm[0] + R[3] * nrm[1] + R[6] * nrm[2], R[1] * nrm[0] + R[4] * nrm[1] + R[7] * nrm[2],
                               R[2] * nrm[0] + R[5] * nrm[1] + R[8] * nrm[2]};
          const double b = dot3(nrm, p) - lc.plane_d;
          const double* sph = lc.link_sphere[t];
          if (b + a[0] * sph[0] + a[1] * sph[1] + a[2] * sph[2] - sph[3] < 0) {
            for (int k = lc.link_adr[t]; k < lc.link_adr[t + 1]; ++k) {
              const double* v = lc.xyzr + 4 * (size_t)k;
              if (b + a[0] * v[0] + a[1] * v[1] + a[2] * v[2] - v[3] < 0) det_mine |= lc.cls[k];
            }
          }
        }
        if (npair > 0) {
          // the robot's geoms against each other, on the frames the position stage published (and at the joint positions it saw:
          // the slack cache integrates joint motion between calls)
          const int tq = t < T::NL ? t : T::NL - 1;
          const double q_seen = stepping ? st.qpre(tq) : st.q(tq);
          det_mine |= self_collision_pairs(lp.ctab.geoms, lp.ctab.verts, lp.ctab.pairs, npair, selfAll, selfAll + kSelfF * kTeams, T::NL, team_due, true, sph, slack,
                                           lp.ctab.self_lever, q_seen);
          stage_fence();
        }
        hit = (team_ballot(det_mine & 1u) ? 1u : 0u) | (team_ballot(det_mine & 2u) ? 2u : 0u);
      }
    }
    __syncthreads();
    if constexpr (BOX) {
      if (stepping) {
        // 0.5 f^2 R of the robot's active joint-limit rows at the constrained solution (f = -D r, R = 1 / D)
        const int tl = t < T::NL ? t : T::NL - 1;
        const double rr = st.limS(tl) * st.xs(tl) - st.limA(tl);
        // (models with dry friction rows run without the noslip pass, the only reader of this sum: rcsh_sim_add_free_box)
        double imp0 = t < T::NL && st.limS(tl) != 0.0 && rr < 0 ? 0.5 * st.limD(tl) * rr * rr : 0.0;
        imp0 = quad_sum(imp0);  // sum over the team's 16 lanes: quads, then rotations by 4 and 8 within the row
        imp0 += row_rotate<4>(imp0);
        imp0 += row_rotate<8>(imp0);
        imp0 = lane_get(imp0, threadIdx.x & 48);  // one lane's bits for the whole team
        box_substep(lbt[0].box, bs, m.gravity, timestep, imp0, t, team_coupled);
      }
      __syncthreads();
      if (place_box && live && !box_placed) {
        if (t < 7) bs[kBoxQ + t] = op.box_qpos[(size_t)e * 7 + t];  // position only: velocity and warm start stay
        box_placed = true;
      }
    }
    if (leader && stepping) {
      if (CON && overflow) r.flags |= kContactOverflow;
      if (CON && team_coupled) esc_contact = true;
      r.time += timestep;
      have_frames = true;
      --budget;
      if (until_conv) {
        r.conv_steps++;
        // (the callbacks' verdict only changes when one of them fires -- and is formed once at the launch's first substep, where
        // "no callback registered" already means converged: the empty all_of, sim.cpp:49-61)
        if (cond_first || r.time - cond_lo > cond_pmin) {
          converged = condition_callbacks<T, ST>(m, P, r, [&] { return hit; });
          cond_lo = condition_callbacks_oldest<T, ST>(P, r);
          cond_first = false;
        }
      }
      more = budget > 0 && !converged;
    }
    if (rend_on) {
      // rendering callbacks after mj_step2 (sim.cpp:108-115): the leader looks at the cameras' clocks, the team records
      uint32_t due_cams = 0;
      if (leader && stepping) {
        for (int c = 0; c < rend_ncam; ++c)
          if (r.time - lrend[team][c] > lp.rend.period[c]) { due_cams |= 1u << c; lrend[team][c] = r.time; }
        if (due_cams) lrend[team][kMaxRateCams] = r.time;
      }
      due_cams = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(threadIdx.x & 48) << 2, (int)due_cams);
      if (due_cams && live) {
        stage_fence();
        const int slot = (int)lrend[team][kMaxRateCams + 1];
        if (slot < lp.rend.capacity) {
          constexpr int NF = T::NL + 9;
          double* dst = lp.rend.snap + (size_t)slot * NF * P.n + e;
          for (int k = t; k < NF; k += kTeamLanes) {
            double v;
            if (k < T::NL) v = st.qpre(k);
            else if (k < T::NL + 7) v = (BOX || CON) ? bs[kBoxPre + (k - T::NL)] : (k == T::NL + 3 ? 1.0 : 0.0);
            else v = k == T::NL + 7 ? lrend[team][kMaxRateCams] : (double)due_cams;
            dst[(size_t)k * P.n] = v;
          }
        }
        stage_fence();
        if (t == 0) lrend[team][kMaxRateCams + 1] += 1.0;
      }
    }
    going = __ballot(more);
    TEAM_MARK(9)
  }
  if (rend_on && leader) {
    for (int c = 0; c < rend_ncam; ++c) lp.rend.last[(size_t)c * P.n + e] = lrend[team][c];
    lp.rend.count[e] = (int32_t)lrend[team][kMaxRateCams + 1];
  }
#ifdef RCSH_WAVE_TIMES
  wt1_ = __builtin_amdgcn_s_memrealtime();
#endif
  // what the end-of-launch contact check reads from memory is asked for now: it arrives while the leader lanes run the epilogue
  // (the contact-resolving launch of per-environment escalation asks for its environments' way BACK: the floor counts there too)
  const bool chk_plane = !CON || esc_role == 2;
  const bool do_check = op.check && (lp.chk.npair > 0 || (chk_plane && lp.chk.plane_points));  // (wave-uniform)
  CheckPrefetch chk_pf;
  // (the environment's record of remaining gaps -- check_team.h: "the slack" --: valid for the position this launch began on unless a
  // reset moved the robot; a step that is redone started from the copy's position, which is what the record still describes)
  float* const chk_slack = lp.chk.slack && live ? lp.chk.slack + (size_t)e * kSlackStride : nullptr;
  // (the contact-resolving launch: its collision passes keep the same record, valid where the last substep began -- a position of this
  // launch's path, which is all the check's slack test asks for; round 6 first looked at every pair there, 300-500k cycles of every
  // escalated environment's launch, a quarter of a quiet one's)
  const bool chk_use_slack = esc_role != 0 && !op.do_reset && op.check == 2;
  if (do_check) check_prefetch(lp.chk, lp.ctab, sep_in, live, chk_pf, chk_use_slack ? chk_slack : nullptr);
  {
    // per-component stores of the epilogue, one per lane instead of a dozen from the leader: the site link's frame of the last
    // position stage, the joints of the observation row
    const bool team_frames = team_ballot(have_frames) != 0;
    if (live && team_frames && t < 12) P.S[(size_t)(Lay<T>::SITE + t) * P.n + e] = st.link(t);
    if (live && op.write_obs && t < T::NARM) op.obs[(size_t)e * (14 + T::NARM) + 7 + t] = st.q(t);
  }
  bool esc_leave = false;
  if (leader) {
    if (until_conv) set_flag(r.flags, kConverged, converged);
    // (per-environment escalation, the contact-resolving launch: whether the environment goes BACK to the lean launch is settled after
    // the end-of-launch check below)
    if (CON && esc_role == 2 && have_frames && esc_contact) r.flags |= kContactResolved;
    env_epilogue<T, ST, false>(P, op, m, e, r, st, have_frames, nsteps);
  }
  __syncthreads();
  if (live) {
    // the LDS-staged state fields go back with all lanes
#pragma unroll
    for (int rd = 0; rd < SF::kRounds; ++rd) {
      const int k = t + rd * kTeamLanes;
      int field = 0, slot = 0;
      SF::locate(k < nstaged ? k : 0, field, slot);
      if (k < nstaged) P.S[(size_t)field * P.n + e] = st.at(slot);
    }
    if constexpr (FRIC) {
      if (t < T::NL) P.S[(size_t)(Lay<T>::XS + t) * P.n + e] = st.xs(t);
    }
  }
  if constexpr (CON && !BOX) {
    if (live)
      for (int k = kBoxX + t; k < kBoxState; k += kTeamLanes) P.S[(size_t)(Lay<T>::BOX + k) * P.n + e] = bs[k];
  }
  if constexpr (BOX) {
    if (live) {
      using L = Lay<T>;
      for (int k = t; k < kBoxState; k += kTeamLanes) P.S[(size_t)(L::BOX + k) * P.n + e] = bs[k];
    }
    if (leader && op.task && lbt[0].task.pick_cube) {
      // PickCubeSuccessWrapper.step (sim.py:396-431): success, else the ManiSkill-style shaped reward; both / 5
      Pose tcp;
      double linkR[9], linkP[3];
#pragma unroll
      for (int k = 0; k < 9; ++k) linkR[k] = st.link(k);
#pragma unroll
      for (int k = 0; k < 3; ++k) linkP[k] = st.link(9 + k);
      cartesian_position(m, P.robot, linkR, linkP, tcp);
      const bool has_g = T::GRIP && P.grip.present;
      const double w = has_g ? gripper_width<T, ST>(P, r) : 0.0;
      const bool grasped = has_g && (w > 0.01 && w < 0.99);
      const bool closed = has_g && (r.flags & kHasGripCmd) && !(r.flags & kGripCmd);  // obs["gripper"] == BINARY_GRIPPER_CLOSED
      const double bx = bs[kBoxQ], by = bs[kBoxQ + 1], bz = bs[kBoxQ + 2];
      const bool success = bz > lbt[0].task.success_z && closed;
      double reward = 5.0;
      if (!success) {
        const double d1 = sqrt((bx - tcp.t[0]) * (bx - tcp.t[0]) + (by - tcp.t[1]) * (by - tcp.t[1]) + (bz - tcp.t[2]) * (bz - tcp.t[2]));
        const double* eh = lbt[0].task.ee_home;
        const double d2 = sqrt((bx - eh[0]) * (bx - eh[0]) + (by - eh[1]) * (by - eh[1]) + (bz - eh[2]) * (bz - eh[2]));
        const double g = grasped ? 1.0 : 0.0;
        reward = (1 - tanh(5 * d1)) + g + (1 - tanh(5 * d2)) * g;
      }
      double* o = op.task + (size_t)e * 9;
#pragma unroll
      for (int k = 0; k < 7; ++k) o[k] = bs[kBoxQ + k];
      o[7] = reward / 5;
      o[8] = success ? 1.0 : 0.0;
    }
  }
  // ---- contacts nobody resolves (check_team.h): once per stepping launch, on the position the next position stage will see.
  // Everything of the launch is in HBM by now; the LDS block and the link records' memory are the check's workspace.
  if (do_check) {  // (wave-uniform)
    static_assert(sizeof(LinkRec) * T::NL >= sizeof(double) * 12 * T::NL * kTeams, "the links' world frames fit where their records were");
    const double q_final = live && t < T::NL ? st.q(t) : 0.0;
    // (certifying check, check_team.h: how far the lane's joint has been from where the launch ends, at most -- and its "effective
    // path" over the launch, twice the width of the interval it has been in less its net displacement: for every position q it took,
    // |q - q_start| + |q - q_end| is at most that; a joint that moved one way: its displacement; one that chattered around a value, as the
    // fingers' servo does for a dozen steps after a reset: twice the band, where the summed substep-to-substep travel is ten times that.
    // 0: the check of the final position alone.  The contact-resolving launch asks the SAME questions of its own launch: an environment
    // goes back to the lean launch when the lean launch's certificate would have passed this one -- no wider margin: an environment that
    // goes back one step too early costs its redone step what staying would have cost, and a gripper opening from shut pads (gap at the
    // start exactly 0, gap at the end exactly its travel) passes a test with the travel itself and never one with twice the travel)
    double chk_dend = 0.0, chk_psum = 0.0, chk_dlo = 0.0, chk_dhi = 0.0;
    {
      // (the fingers' lanes: the same for the opening / the common shift -- their values at the launch's two ends from both fingers' lanes)
      double w_final = q_final, w0 = chk_q0;
      if (T::GRIP) {
        const int fb = (int)(threadIdx.x & 48u) + T::NARM;
        const double qa = lane_get(q_final, fb), qb = lane_get(q_final, fb + 1), sa = lane_get(chk_q0, fb), sb = lane_get(chk_q0, fb + 1);
        if (t == T::NARM) { w_final = qa + qb; w0 = sa + sb; }
        if (t == T::NARM + 1) { w_final = qa - qb; w0 = sa - sb; }
      }
      if (op.check == 2 && live && t < T::NL && !chk_first) {
        const double lo = fmin(chk_qmin, w_final), hi = fmax(chk_qmax, w_final);
        chk_dend = fmax(hi - w_final, w_final - lo);
        chk_dlo = lo - w_final; chk_dhi = hi - w_final;
        chk_psum = 2.0 * (hi - lo) - fabs(w_final - w0);
      }
    }
    __syncthreads();
    double* const sep = P.S + (size_t)Lay<T>::SEP * P.n + (live ? e : 0);
    // (an environment that carries the flag already has nothing to find out: its team sits the check out -- a pair that stays in
    // contact has no separating direction to remember and would go through the full refinement in every launch, and the launch
    // waits for its slowest wavefront)
    const bool flagged = ((uint32_t)__builtin_amdgcn_ds_bpermute((int)(threadIdx.x & 48u) << 2, (int)r.flags) & kContactUnresolved) != 0;
    // (with per-environment escalation the check is what sends an environment to the contact-resolving kernel: nobody sits it out)
    const bool checked = live && (!flagged || esc_role != 0);
    double* check_work = lds;
    double* check_mv = lds + kLdsMain;
    if constexpr (CON) {
      static_assert(sizeof(Arena) >= sizeof(double) * check_work_doubles(T::NL), "the check's workspace fits the contact arena");
      static_assert(kLdsMain >= check_mv_doubles(T::NL), "... and its travel tables the teams' state block (written back by now)");
      check_work = reinterpret_cast<double*>(&larena[0]);
      check_mv = lds;
    } else if constexpr (DET) {
      static_assert(kSelfF * kTeams + kSelfStage + kSelfCache + kSelfTag >= check_mv_doubles(T::NL), "the travel tables fit where the detection kept its frames");
      check_mv = lself;  // (the detection of the substep loop is over)
    }
#ifdef RCSH_WAVE_TIMES
    wt2_ = __builtin_amdgcn_s_memrealtime();
#endif
    const bool hit = unresolved_contact_check<T>(lp.chk, lp.ctab, lp.coll, llinks, reinterpret_cast<double*>(&llinks[0]), check_work, q_final, checked, chk_plane, sep_in, sep, P.n, chk_pf,
                                                 chk_dend, chk_psum, check_mv, chk_first ? q_final : chk_q0, esc_role != 0 ? chk_slack : nullptr, chk_use_slack, op.check == 2 ? esc_role : 0,
                                                 CON && esc_role == 2 && team_ballot(esc_contact) != 0, chk_dlo, chk_dhi);
#ifdef RCSH_CHECK_DEBUG
    if (leader) { atomicAdd(&g_chk_dbg[34], hit ? 1 : 0); atomicAdd(&g_chk_dbg[37], 1); }
#endif
    if (esc_role == 1) {
      // per-environment escalation: the step is redone by the contact-resolving launch that follows (RunOp::esc_role)
      if (leader && hit) {
        atomicOr(reinterpret_cast<unsigned long long*>(lop.esc + ((P.n + 63) >> 6) + (e >> 6)), 1ull << (e & 63));
        atomicAdd(lop.esc_ctr + 2, 1u);
      }
    } else if (esc_role == 2) {
      // Back to the lean launch: an environment none of whose substeps met a contact AND whose launch passes the lean launch's own
      // certificate.  Whoever is in contact or near one stays: here every substep looks.  (Round 5 kept an
      // environment here until its reset: 1146 of 4096 by step 1000 of the headline rollout, 87 % of their collision passes quiet.)
      if (leader && have_frames && !esc_contact && !hit) esc_leave = true;
    } else if (leader && hit && !(r.flags & kContactUnresolved)) {
      P.flags[e] = r.flags | kContactUnresolved;
      if (op.write_obs && op.info) op.info[(size_t)e * 8 + 7] = 1;
    }
  } else if (esc_role == 2 && leader && have_frames && !esc_contact) {
    esc_leave = true;  // (no check in this launch: quiet is enough)
  }
  if (esc_role == 2) {
    if (esc_leave) atomicOr(reinterpret_cast<unsigned long long*>(lop.esc + 2 * ((P.n + 63) >> 6) + (e >> 6)), 1ull << (e & 63));
    if (lop.esc_part != 1) esc_finish(lop, P.n);
  }
  if (esc_role == 1 && blockIdx.x == 0 && threadIdx.x == 0 && lop.esc_host) {
    // for the host, which runs ahead of the device: which step's lean launch has (all but) ended, and how many environments were
    // escalated when it began (host memory; read without synchronisation: a hint for how the next steps are enqueued, nothing more)
    __hip_atomic_store(lop.esc_host + 1, lop.esc_ctr[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(lop.esc_host, (uint32_t)lop.esc_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  TEAM_MARK(10)
  TEAM_CLOCK_FLUSH()
#ifdef RCSH_WAVE_TIMES
  if (esc_role == 1 && threadIdx.x == 0 && blockIdx.x < 4096) {
    g_wave_times[0][blockIdx.x] = wt0_; g_wave_times[1][blockIdx.x] = wt1_; g_wave_times[2][blockIdx.x] = wt2_;
    g_wave_times[3][blockIdx.x] = __builtin_amdgcn_s_memrealtime();
  }
#endif
#ifdef RCSH_PHASE_TIMING
  if (CON && esc_role == 2 && threadIdx.x == 0) {  // this workgroup's share of the contact-resolving launch: sum, count, worst
    const unsigned long long dt = __builtin_readcyclecounter() - wg_clock0;
    atomicAdd(&g_team_cycles[64], dt); atomicAdd(&g_team_cycles[65], 1ull);
    if (atomicMax(&g_team_cycles[66], dt) < dt) {  // the worst workgroup so far: what it did (racy among near-ties: a development figure)
      for (int k_ = 0; k_ < 8; ++k_) g_team_cycles[83 + k_] = s_wg_acc[k_];
      g_team_cycles[91] = (unsigned long long)e;
    }
  }
#endif
}
template <class T, bool FRIC, bool BOX = false, bool CON = false, bool DET = false>
__global__ void __launch_bounds__(64) k_run_team(Params Pk, RunOp opk) {
  run_team_body<T, FRIC, BOX, CON, DET>(Pk, opk);
}
template <class T, bool FRIC, bool BOX = false, bool CON = false, bool DET = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) k_run_team_occ2(Params Pk, RunOp opk) {
  run_team_body<T, FRIC, BOX, CON, DET>(Pk, opk);
}

// ---- small elementwise kernels behind the 1:1 SimRobot / SimGripper / mjData accessors

// dst[e][i] = S[field0 + i][e]
__global__ void k_gather(const double* S, int n, int field0, int width, double* dst) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  for (int i = 0; i < width; ++i) dst[(size_t)e * width + i] = S[(size_t)(field0 + i) * n + e];
}
// S[field0 + i][e] = src[e][i] where mask
__global__ void k_scatter(double* S, int n, int field0, int width, const double* src, const uint8_t* mask) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n || (mask && !mask[e])) return;
  for (int i = 0; i < width; ++i) S[(size_t)(field0 + i) * n + e] = src[(size_t)e * width + i];
}
// flags[e] = (flags[e] | set) & ~clear where mask
__global__ void k_flags_update(uint32_t* flags, int n, uint32_t set, uint32_t clear, const uint8_t* mask) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n && (!mask || mask[e])) flags[e] = (flags[e] | set) & ~clear;
}
__global__ void k_flags_to_bytes(const uint32_t* flags, int n, uint32_t bit, uint8_t* dst) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) dst[e] = (flags[e] & bit) != 0;
}

}  // namespace rcsh

// ---------------------------------------------------------------------------------------------------------
// Cartesian control path (reference python/rcs/envs/base.py:273-286,490-565 and src/sim/SimRobot.cpp:145-155).
// These kernels work directly on the SoA state: one thread per environment, no staging, IK in registers.
#include "ik.h"
#include "ik_team.h"

namespace rcsh {

struct CartOp {
  int32_t env_layer;        // 1: wrappers' action() + RobotEnv.step; 0: bare SimRobot::set_cartesian_position
  const uint8_t* mask;
  const double* action;     // env_layer: [n][6|7] xyzrpy / tquat action; else [n][7] target pose (xyz + xyzw)
  const float* gripper;     // env_layer only, may be null
};

template <class T>
__device__ __forceinline__ void load_pose7(const double* S, int n, int e, int field0, Pose& p) {
  p.t[0] = S[(field0 + 0) * n + e]; p.t[1] = S[(field0 + 1) * n + e]; p.t[2] = S[(field0 + 2) * n + e];
  p.q[0] = S[(field0 + 3) * n + e]; p.q[1] = S[(field0 + 4) * n + e]; p.q[2] = S[(field0 + 5) * n + e];
  p.q[3] = S[(field0 + 6) * n + e];
}
template <class T>
__device__ __forceinline__ void store_pose7(double* S, int n, int e, int field0, const Pose& p) {
  S[(field0 + 0) * n + e] = p.t[0]; S[(field0 + 1) * n + e] = p.t[1]; S[(field0 + 2) * n + e] = p.t[2];
  S[(field0 + 3) * n + e] = p.q[0]; S[(field0 + 4) * n + e] = p.q[1]; S[(field0 + 5) * n + e] = p.q[2];
  S[(field0 + 6) * n + e] = p.q[3];
}

// Everything of a Cartesian env-step that precedes the IK (wrappers' action(), RobotEnv.step bookkeeping): returns
// whether a new target is commanded and, if so, the TCP target in robot coordinates.  env_layer 0: the bare
// SimRobot::set_cartesian_position target.
template <class T>
__device__ __forceinline__ bool cart_prepare(const Params& P, const CartOp& op, const DevModelHead& m, int e, uint32_t& flags, Pose& target) {
  using L = Lay<T>;
  const int n = P.n;
  double* S = P.S;
  bool command = true;
  if (!op.env_layer) {
    const double* a = op.action + (size_t)e * 7;
    pose_from_quat(a + 3, a, target);
  } else {
    const bool trpy = P.env.mode == 1;
    const int aw = trpy ? 6 : 7;
    double a[7];
    for (int i = 0; i < aw; ++i) a[i] = op.action[(size_t)e * aw + i];
    if (P.env.relative_to != 0) {
      // RelativeActionSpace.action, Cartesian branches (base.py:490-565)
      const bool last_step = P.env.relative_to == 1;
      Pose origin;
      if (last_step) {
        double linkR[9], linkP[3];
        for (int k = 0; k < 9; ++k) linkR[k] = S[(L::SITE + k) * n + e];
        for (int k = 0; k < 3; ++k) linkP[k] = S[(L::SITE + 9 + k) * n + e];
        cartesian_position(m, P.robot, linkR, linkP, origin);
        store_pose7<T>(S, n, e, L::ORIGIN, origin);
      } else {
        load_pose7<T>(S, n, e, L::ORIGIN, origin);
      }
      Pose given, offset, tmp;
      if (trpy) pose_from_rpy(a + 3, a, given); else pose_from_quat(a + 3, a, given);
      if (last_step || !(flags & kHasLastAction)) {
        pose_limit_translation_length(given, P.env.max_mov[0], tmp);
        pose_limit_rotation_angle(tmp, P.env.max_mov[1], offset);
      } else {
        Pose last, last_inv, diff, lim;
        load_pose7<T>(S, n, e, L::LASTA, last);
        pose_inverse(last, last_inv);
        pose_mul(given, last_inv, diff);
        pose_limit_translation_length(diff, P.env.max_mov[0], tmp);
        pose_limit_rotation_angle(tmp, P.env.max_mov[1], lim);
        pose_mul(lim, last, offset);
      }
      store_pose7<T>(S, n, e, L::LASTA, offset);
      flags |= kHasLastAction;
      Pose rot;
      pose_mul(offset, origin, rot);
      const double t[3] = {origin.t[0] + offset.t[0], origin.t[1] + offset.t[1], origin.t[2] + offset.t[2]};
      const double lo[3] = {-0.855, -0.855, 0.0}, hi[3] = {0.855, 0.855, 1.188};  // base.py:31-38
      Pose unclipped;
      if (trpy) {
        double rpy[3];
        pose_rpy(rot, rpy);
        pose_from_rpy(rpy, t, unclipped);
        pose_rpy(unclipped, a + 3);
      } else {
        pose_from_quat(rot.q, t, unclipped);
        a[3] = unclipped.q[0]; a[4] = unclipped.q[1]; a[5] = unclipped.q[2]; a[6] = unclipped.q[3];
      }
      for (int k = 0; k < 3; ++k) a[k] = clampd(unclipped.t[k], lo[k], hi[k]);
    }
    // GripperWrapper.action (base.py:721-735)
    if (T::GRIP && P.grip.present && op.gripper) {
      float g = op.gripper[e];
      if (P.env.binary_gripper) g = rintf(g);
      g = fminf(fmaxf(g, 0.0f), 1.0f);
      const double w = P.env.binary_gripper ? (g == 0.0f ? 0.0 : 1.0) : (double)g;
      S[(L::GRIP + 0) * n + e] = w;
      S[(L::CTRL + T::NU - 1) * n + e] = w * (P.grip.max_act - P.grip.min_act) + P.grip.min_act;
      set_flag(flags, kGripCmd, P.env.binary_gripper ? g != 0.0f : g >= 0.5f);
      flags |= kHasGripCmd;
    }
    // RobotEnv.step (base.py:255-288)
    command = !(flags & kHasPrevAction);
    for (int i = 0; i < aw; ++i) {
      const double pa = S[(L::PREVA + i) * n + e];
      command = command || !(fabs(a[i] - pa) <= 1e-3);
      S[(L::PREVA + i) * n + e] = a[i];
    }
    flags |= kHasPrevAction;
    if (trpy) pose_from_rpy(a + 3, a, target); else pose_from_quat(a + 3, a, target);
  }
  return command;
}

// The Cartesian launch, a team of 16 lanes per environment (ik_team.h): the leader lane runs the wrapper logic, the team
// the CLIK, lane t commits joint t.
template <class T>
__global__ void __launch_bounds__(64) k_cartesian_team(Params P, CartOp op) {
  using L = Lay<T>;
  constexpr int kTeams = 64 / kTeamLanes;
  __shared__ DevModelHead lm;
  __shared__ LinkRec llinks[T::NARM];
  __shared__ IkTeamBlock<T> blocks[kTeams];
  __shared__ double desired[kTeams][12];
  {
    stage_team_model<T::NARM>(P.model, lm, llinks);
    __syncthreads();
  }
  const DevModelHead& m = lm;
  const int team = threadIdx.x / kTeamLanes, t = threadIdx.x % kTeamLanes;
  const int per_xcd = gridDim.x / 8;  // XCD-contiguous environment ranges, as in k_run_team
  const int e = ((blockIdx.x % 8) * per_xcd + blockIdx.x / 8) * kTeams + team;
  const bool live = e < P.n && !(op.mask && !op.mask[e < P.n ? e : 0]);
  const bool leader = t == 0 && live;
  const int n = P.n;
  double* S = P.S;
  uint32_t flags = 0;
  bool command = false;
  if (leader) {
    flags = P.flags[e];
    Pose tcp, target;
    tcp.t[0] = P.robot.tcp[0]; tcp.t[1] = P.robot.tcp[1]; tcp.t[2] = P.robot.tcp[2];
    tcp.q[0] = P.robot.tcp[3]; tcp.q[1] = P.robot.tcp[4]; tcp.q[2] = P.robot.tcp[5]; tcp.q[3] = P.robot.tcp[6];
    command = cart_prepare<T>(P, op, m, e, flags, target);
    if (command) clik_desired(m, target, tcp, desired[team], desired[team] + 9);
  }
  const bool run = team_ballot(command) != 0;
  __syncthreads();
  double Rd[9], td[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) Rd[k] = desired[team][k];
#pragma unroll
  for (int k = 0; k < 3; ++k) td[k] = desired[team][9 + k];
  const bool joint = run && t < T::NARM;
  const double q_now = joint ? S[(L::QPOS + t) * n + e] : 0.0;
  double q = q_now;
  int iters = 0;
  const bool ok = clik_team<T>(m, llinks, blocks[team], t, run, Rd, td, q, &iters);
  if (joint && ok) {  // SimRobot::set_joint_position(joint_vals), SimRobot.cpp:123-131,145-155
    S[(L::TARGET + t) * n + e] = q;
    S[(L::PREVQ + t) * n + e] = q_now;
    S[(L::CTRL + t) * n + e] = q;
  }
  if (leader) {
    if (command) {
      if (ok) flags = (flags | kIkSuccess | kIsMoving) & ~kIsArrived;
      else flags &= ~kIkSuccess;
    }
    P.flags[e] = flags;
  }
}

// Kinematics.inverse on caller-supplied targets and start configurations (reference src/rcs/Kinematics.cpp:28-82), a team of 16
// lanes per target like k_cartesian_team: the leader lane composes the desired site placement, the team runs the CLIK, lane t
// writes joint t.  (Through round 2 a lane per target did this: 512 VGPRs and scratch for a 1300-instruction serial iteration.)
template <class T>
__global__ void __launch_bounds__(64) k_ik_team(Params P, const double* pose, const double* q0, const double* tcp7, double* q_out,
                                                uint8_t* success, int32_t* iterations) {
  constexpr int kTeams = 64 / kTeamLanes;
  __shared__ DevModelHead lm;
  __shared__ LinkRec llinks[T::NARM];
  __shared__ IkTeamBlock<T> blocks[kTeams];
  __shared__ double desired[kTeams][12];
  stage_team_model<T::NARM>(P.model, lm, llinks);
  __syncthreads();
  const DevModelHead& m = lm;
  const int team = threadIdx.x / kTeamLanes, t = threadIdx.x % kTeamLanes;
  const int e = blockIdx.x * kTeams + team;
  const bool live = e < P.n;
  if (t == 0 && live) {
    Pose tcp, target;
    const double* tv = tcp7 ? tcp7 : P.robot.tcp;
    tcp.t[0] = tv[0]; tcp.t[1] = tv[1]; tcp.t[2] = tv[2];
    tcp.q[0] = tv[3]; tcp.q[1] = tv[4]; tcp.q[2] = tv[5]; tcp.q[3] = tv[6];
    quat_normalize(tcp.q);
    pose_from_quat(pose + (size_t)e * 7 + 3, pose + (size_t)e * 7, target);
    clik_desired(m, target, tcp, desired[team], desired[team] + 9);
  }
  __syncthreads();
  double Rd[9], td[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) Rd[k] = desired[team][k];
#pragma unroll
  for (int k = 0; k < 3; ++k) td[k] = desired[team][9 + k];
  const bool joint = live && t < T::NARM;
  double q = joint ? q0[(size_t)e * T::NARM + t] : 0.0;
  int iters = 0;
  const bool ok = clik_team<T>(m, llinks, blocks[team], t, live, Rd, td, q, &iters);
  if (live && t < T::NL) q_out[(size_t)e * T::NL + t] = t < T::NARM ? q : 0.0;  // model.nq entries, fingers zero (quirk Q7)
  if (live && t == 0) {
    if (success) success[e] = ok;
    if (iterations) iterations[e] = iters;
  }
}

// Kinematics.inverse / forward on caller-supplied configurations (reference src/rcs/Kinematics.cpp:28-82)
template <class T>
__global__ void __launch_bounds__(64) k_ik(Params P, const double* pose, const double* q0, const double* tcp7, double* q_out,
                                           uint8_t* success, int32_t* iterations, int forward) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= P.n) return;
  const DevModel& m = *P.model;
  Pose tcp;
  const double* tv = tcp7 ? tcp7 : P.robot.tcp;
  tcp.t[0] = tv[0]; tcp.t[1] = tv[1]; tcp.t[2] = tv[2];
  tcp.q[0] = tv[3]; tcp.q[1] = tv[4]; tcp.q[2] = tv[5]; tcp.q[3] = tv[6];
  quat_normalize(tcp.q);
  double q[T::NARM];
#pragma unroll
  for (int i = 0; i < T::NARM; ++i) q[i] = q0[(size_t)e * T::NARM + i];
  if (forward) {
    // Pin::forward returns frame * tcp_offset.inverse() (reference quirk Q7), in robot coordinates
    double Rs[9], ps[3];
    site_fk<T>(m, q, Rs, ps, nullptr, nullptr);
    Pose site, base, base_inv, in_robot, tinv, out;
    pose_from_mat(Rs, ps, site);
    const double bq[4] = {m.base_quat[1], m.base_quat[2], m.base_quat[3], m.base_quat[0]};
    pose_from_quat(bq, m.base_pos, base);
    pose_inverse(base, base_inv);
    pose_mul(base_inv, site, in_robot);
    pose_inverse(tcp, tinv);
    pose_mul(in_robot, tinv, out);
    double* o = q_out + (size_t)e * 7;
    o[0] = out.t[0]; o[1] = out.t[1]; o[2] = out.t[2]; o[3] = out.q[0]; o[4] = out.q[1]; o[5] = out.q[2]; o[6] = out.q[3];
  }
  // (the inverse: k_ik_team)
}

}  // namespace rcsh
