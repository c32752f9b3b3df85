// model.h -- device-side model tables for the batched simulation backend.
//
// The scene arrives across the C-ABI as flat mjModel-like tables
// (include/rcs_hip.h: rcsh_model_desc).  finalize_model() folds welded bodies
// into "links" (one 1-dof joint each), checks that the scene matches one of the
// compiled robot archetypes (serial arm of NARM hinges, optionally a two-finger
// parallel gripper on the last link) and produces the constant tables the
// kernels read with wave-uniform (scalar) loads.
#pragma once
#include <cstdint>
#include <cmath>
#include <cstring>

namespace rcsh {

constexpr int kMaxLinks = 12;
constexpr int kMaxArm = 8;

enum JointKind : int32_t { kSlide = 2, kHinge = 3 };

// solimp, preprocessed on the host so the kernel evaluates the impedance sigmoid without divisions
struct Imp {
  double d0, d1;        // impedance at zero / full penetration (clamped to [0.0001, 0.9999])
  double inv_width;     // 1 / width
  double mid, inv_mid, inv_1mmid;
  double power;
  int32_t mode;         // 0 constant (0.5 (d0 + d1)), 1 linear, 2 quadratic, 3 general power
  int32_t pad;
};

// All members are wave-uniform in the kernels.  Fixed maximum sizes keep the struct a POD that is
// copied to HBM once per GPU (a few KB); kernels template on the real sizes.
//
// DevModelHead: everything that is NOT per link (options, qpos0, the gripper's tendon actuator and coupling equality, the
// frames SimRobot reads).  The team kernels keep only this part in LDS -- their per-link constants are the LinkRec records
// below -- which leaves the rest of a workgroup's 40 KB share of the CU's LDS to the contact phase (contact_team.h).
struct DevModelHead {
  int32_t nl;    // links == dofs
  int32_t narm;  // arm dofs; a gripper (if any) occupies dofs narm, narm+1
  int32_t has_gripper;
  int32_t pad0;
  double timestep;
  double gravity[3];
  double qpos0[kMaxLinks];
  // ---- gripper: tendon actuator (ctrl slot narm) over the two finger dofs + coupling equality
  int32_t grp_has_act;
  int32_t grp_biasaffine;
  int32_t grp_ctrllimited;
  int32_t grp_forcelimited;
  double grp_coef[2];  // moment of the actuator on finger dofs (gear * tendon coefficients)
  double grp_gain;
  double grp_bias[3];
  double grp_ctrlrange[2];
  double grp_forcerange[2];
  int32_t eq_active;
  int32_t pad1;
  double eq_polycoef[5];
  Imp eq_imp;
  double eq_K, eq_B;
  // ---- frames read by SimRobot
  int32_t site_link;  // link carrying the attachment site (-1: static)
  int32_t pad2;
  double site_pos[3];
  double site_rot[9];
  double base_pos[3];   // world pose of the robot base body (static)
  double base_quat[4];  // wxyz
  int32_t has_friction;  // any fl_floss > 0
  int32_t pad3;
};
static_assert(sizeof(DevModelHead) % 8 == 0, "staged into LDS in 8-byte words");

struct DevModel : DevModelHead {
  // ---- link tree (link i's parent is i-1 for the arm; both fingers hang off link narm-1)
  double pos0[kMaxLinks][3];   // origin of link frame in parent link frame at q = qpos0
  double rot0[kMaxLinks][9];   // parent-link <- link rotation at q = qpos0 (row-major)
  double axis[kMaxLinks][3];   // joint axis, link frame
  double jpos[kMaxLinks][3];   // joint anchor, link frame
  int32_t jtype[kMaxLinks];
  // composite inertial of the link with everything welded to it, link frame
  double mass[kMaxLinks];
  double com[kMaxLinks][3];
  double inertia[kMaxLinks][6];  // about com: xx yy zz xy xz yz
  double gcm[kMaxLinks];         // sum of gravcomp * mass over the welded bodies
  double gccom[kMaxLinks][3];    // gcm-weighted centre (link frame)
  double gcm_sub[kMaxLinks];     // gcm summed over the link and its descendants
  double armature[kMaxLinks];
  double damping[kMaxLinks];
  // joint limits (soft constraint rows)
  int32_t limited[kMaxLinks];
  double range[kMaxLinks][2];
  double margin[kMaxLinks];
  Imp lim_imp[kMaxLinks];
  double lim_K[kMaxLinks], lim_B[kMaxLinks];  // reference-acceleration stiffness / damping (from solref)
  double invweight0[kMaxLinks];
  // joint-level actuator force clamp and gravity compensation routing
  int32_t actfrclimited[kMaxLinks];
  double actfrcrange[kMaxLinks][2];
  int32_t actgravcomp[kMaxLinks];
  // ---- arm actuators: one affine actuator per arm dof (ctrl slot i), joint transmission
  int32_t arm_has_act[kMaxArm];
  double arm_gear[kMaxArm];
  double arm_gain[kMaxArm];
  double arm_bias[kMaxArm][3];
  int32_t arm_biasaffine[kMaxArm];
  int32_t arm_ctrllimited[kMaxArm];
  double arm_ctrlrange[kMaxArm][2];
  int32_t arm_forcelimited[kMaxArm];
  double arm_forcerange[kMaxArm][2];
  int32_t axis_z[kMaxLinks];    // joint axis is +z and the anchor is the link origin (every FR3 / xArm7 hinge)
  int32_t gc_same_com[kMaxLinks];  // gccom == com (uniform gravcomp over the welded bodies)
  // ---- dry joint friction (dof_frictionloss): one soft row per joint with a Huber cost.  The row's position is 0,
  // so its regulariser is a model constant: D = 1/R, R = (1 - imp(0)) / imp(0) * invweight0; aref = -B * qvel
  double fl_floss[kMaxLinks];
  double fl_D[kMaxLinks];
  double fl_B[kMaxLinks];
  double fl_R[kMaxLinks];  // half-width of the quadratic zone: frictionloss / D
  double inertia_diag_sum;  // trace of M(qpos0) (host side: mjModel.stat.meaninertia of scenes with free bodies)
};
static_assert(sizeof(DevModel) % 8 == 0, "copied in 8-byte words");

// Per-link constants of the team kernels (dyn_team.h, ik_team.h) as an array of structures: lane t reads link t's
// record through ONE base address with immediate offsets, and neighbouring fields merge into 16-byte LDS reads -- with
// DevModel's per-field tables every table costs its own per-lane address register.  560 bytes: the records of links
// 0..8 start in distinct LDS banks.  Filled from a DevModel by fill_link_records(); stored right behind the DevModel in
// device memory.
struct LinkRec {
  double qpos0, rot0[9], pos0[3], axis[3], jpos[3];
  double mass, gcm, com[3], inertia[6], gccom[3];
  double damping, armature, gcm_sub, actfrcrange[2], range[2], margin;
  // The on/off flags of the model are folded into the numbers, so the kernels run one branch-free formula on every
  // lane: an absent limit is an infinite range, an absent bias / actuator is zero coefficients, actuator-side gravity
  // compensation is a 0/1 weight.  (Clamping to +-inf and adding 0 * x are exact.)
  double arm_ctrlrange[2], arm_gear, arm_gain, arm_bias[3], arm_forcerange[2];  // arm dofs only, else zero / infinite
  double lim_K, lim_B, invweight0;
  double fl_floss, fl_D, fl_B, fl_R;
  Imp lim_imp;
  double actgravcomp_w;  // 1: gravity compensation goes through the actuator (before the joint-level force clamp)
  int32_t axis_z, jtype, gc_same_com, pad0;
  double pad1[2];
};
static_assert(sizeof(LinkRec) == 560, "LDS bank spread of the records relies on this size");

inline void fill_link_records(const DevModel& m, LinkRec* out) {
  for (int i = 0; i < kMaxLinks; ++i) {
    LinkRec& k = out[i];
    std::memset(&k, 0, sizeof(k));
    k.qpos0 = m.qpos0[i];
    for (int j = 0; j < 9; ++j) k.rot0[j] = m.rot0[i][j];
    for (int j = 0; j < 3; ++j) { k.pos0[j] = m.pos0[i][j]; k.axis[j] = m.axis[i][j]; k.jpos[j] = m.jpos[i][j]; k.com[j] = m.com[i][j]; k.gccom[j] = m.gccom[i][j]; }
    for (int j = 0; j < 6; ++j) k.inertia[j] = m.inertia[i][j];
    k.mass = m.mass[i]; k.gcm = m.gcm[i]; k.gcm_sub = m.gcm_sub[i];
    k.damping = m.damping[i]; k.armature = m.armature[i];
    k.actfrcrange[0] = m.actfrcrange[i][0]; k.actfrcrange[1] = m.actfrcrange[i][1];
    k.range[0] = m.range[i][0]; k.range[1] = m.range[i][1]; k.margin = m.margin[i];
    k.lim_K = m.lim_K[i]; k.lim_B = m.lim_B[i]; k.invweight0 = m.invweight0[i];
    k.fl_floss = m.fl_floss[i]; k.fl_D = m.fl_D[i]; k.fl_B = m.fl_B[i]; k.fl_R = m.fl_R[i];
    k.lim_imp = m.lim_imp[i];
    k.axis_z = m.axis_z[i]; k.jtype = m.jtype[i]; k.gc_same_com = m.gc_same_com[i];
    const double inf = HUGE_VAL;
    k.actgravcomp_w = m.actgravcomp[i] ? 1.0 : 0.0;
    if (!m.actfrclimited[i]) { k.actfrcrange[0] = -inf; k.actfrcrange[1] = inf; }
    if (!m.limited[i]) { k.range[0] = -inf; k.range[1] = inf; }
    k.arm_ctrlrange[0] = -inf; k.arm_ctrlrange[1] = inf;
    k.arm_forcerange[0] = -inf; k.arm_forcerange[1] = inf;
    if (i < m.narm && i < kMaxArm && m.arm_has_act[i]) {
      if (m.arm_ctrllimited[i]) { k.arm_ctrlrange[0] = m.arm_ctrlrange[i][0]; k.arm_ctrlrange[1] = m.arm_ctrlrange[i][1]; }
      k.arm_gear = m.arm_gear[i]; k.arm_gain = m.arm_gain[i];
      if (m.arm_biasaffine[i])
        for (int j = 0; j < 3; ++j) k.arm_bias[j] = m.arm_bias[i][j];
      if (m.arm_forcelimited[i]) { k.arm_forcerange[0] = m.arm_forcerange[i][0]; k.arm_forcerange[1] = m.arm_forcerange[i][1]; }
    }
  }
}

}  // namespace rcsh
