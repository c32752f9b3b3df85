/*
 * rcs_sim.c -- TEST INFRASTRUCTURE (see rcs_oracle.h).
 *
 * Restatement of the reference's simulation adapters for ONE environment:
 *   Sim        reference src/sim/sim.cpp, src/sim/sim.h
 *   SimRobot   reference src/sim/SimRobot.cpp, src/sim/SimRobot.h
 *   SimGripper reference src/sim/SimGripper.cpp, src/sim/SimGripper.h
 * The std::function callback lists are unrolled into the fixed registration
 * order SimEnvCreator produces (reference python/rcs/envs/creators.py:88,105).
 */
#include <math.h>
#include <string.h>

#include "rcs_oracle.h"

void orc_sim_init(orc_sim* s, orc_model* m) {
  memset(s, 0, sizeof(*s));
  s->m = m;
  /* SimConfig defaults, sim.h:29-34 */
  s->async_control = 0; s->realtime = 0; s->frequency = 30; s->max_convergence_steps = 500;
  s->converged = 1; /* sim.h:70 */
  orc_reset_data(m, &s->d);
}

/* SimRobot::SimRobot, SimRobot.cpp:27-43 */
void orc_sim_add_robot(orc_sim* s, int n, const int* jnt_ids, const int* act_ids, int site, int base_body,
                       const double* q_home, const orc_pose* tcp_offset, int register_convergence_callback) {
  s->has_robot = 1;
  s->arm_n = n;
  for (int i = 0; i < n; i++) { s->arm_jnt[i] = jnt_ids[i]; s->arm_act[i] = act_ids[i]; s->q_home[i] = q_home[i]; }
  s->attachment_site = site;
  s->base_body = base_body;
  s->tcp_offset = *tcp_offset;
  s->joint_rotational_tolerance = .05 * (M_PI / 180.0); /* SimRobot.h:15-16 */
  s->robot_period = 0.1;                                /* SimRobot.h:17 */
  s->robot_conv_registered = register_convergence_callback;
  /* SimRobotState defaults, SimRobot.h:49-57 */
  s->ik_success = 1; s->robot_collision = 0; s->is_moving = 0; s->is_arrived = 0;
  s->ik.m = s->m; s->ik.site = site;
  orc_robot_reset(s); /* m_reset */
}

/* SimGripper::SimGripper, SimGripper.cpp:13-39 */
void orc_sim_add_gripper(orc_sim* s, int jnt, int act) {
  s->has_gripper = 1;
  s->grp_jnt = jnt; s->grp_act = act;
  /* SimGripperConfig defaults, SimGripper.h:15-23 */
  s->epsilon_inner = 0.005; s->epsilon_outer = 0.005;
  s->grp_period = 0.05;
  s->max_actuator_width = 255; s->min_actuator_width = 0;
  s->max_joint_width = 0.04; s->min_joint_width = 0.0;
  orc_gripper_reset(s);
}

/* ---- SimRobot callbacks */
static void robot_is_arrived_cb(orc_sim* s) { /* SimRobot.cpp:165-170 */
  double mx = 0;
  for (int i = 0; i < s->arm_n; i++) {
    double a = fabs(s->d.qpos[s->arm_jnt[i]] - s->target_angles[i]);
    if (a > mx) mx = a;
  }
  s->is_arrived = mx < s->joint_rotational_tolerance;
}
static void robot_is_moving_cb(orc_sim* s) { /* SimRobot.cpp:156-163 */
  double mx = 0;
  for (int i = 0; i < s->arm_n; i++) {
    double cur = s->d.qpos[s->arm_jnt[i]];
    double a = fabs(cur - s->previous_angles[i]);
    if (a > mx) mx = a;
    s->previous_angles[i] = cur;
  }
  s->is_moving = mx > 0.0001;
}
static int in_set(const int* set, int n, int g) {
  for (int i = 0; i < n; i++) if (set[i] == g) return 1;
  return 0;
}
/* geoms of contact i of the position stage: contact[] first, then the robot's geom-geom contacts (orc_data.self_geom) */
static const int* contact_pair(const orc_sim* s, int i) { return i < s->d.ncon ? s->d.contact_geom[i] : s->d.self_geom[i - s->d.ncon]; }
static int robot_collision_cb(orc_sim* s) { /* SimRobot.cpp:172-182: scan of d->contact[0..ncon) */
  s->robot_collision = 0;
  for (int i = 0; i < s->d.ncon + s->d.nself; i++) {
    const int* pr = contact_pair(s, i);
    if (in_set(s->arm_cgeom, s->arm_ncgeom, pr[0]) || in_set(s->arm_cgeom, s->arm_ncgeom, pr[1])) {
      s->robot_collision = 1;
      break;
    }
  }
  return s->robot_collision;
}
static int robot_convergence_cb(orc_sim* s) { /* SimRobot.cpp:184-191 */
  if (!s->ik_success) return 1;
  return s->is_arrived && !s->is_moving;
}
/* ---- SimGripper callbacks */
static int gripper_convergence_cb(orc_sim* s) { /* SimGripper.cpp:143-151 */
  double w = orc_gripper_get_normalized_width(s);
  s->grp_is_moving = fabs(s->last_width - w) > 0.001 * (s->max_actuator_width - s->min_actuator_width);
  s->last_width = w;
  return !s->grp_is_moving;
}
static int gripper_collision_cb(orc_sim* s) { /* SimGripper.cpp:108-130 */
  s->grp_collision = 0;
  for (int i = 0; i < s->d.ncon + s->d.nself; i++) {
    int g0 = contact_pair(s, i)[0], g1 = contact_pair(s, i)[1];
    if (in_set(s->grp_cfgeom, s->grp_ncfgeom, g0) && in_set(s->grp_cfgeom, s->grp_ncfgeom, g1)) continue; /* finger-finger */
    if ((in_set(s->grp_cgeom, s->grp_ncgeom, g0) || in_set(s->grp_cgeom, s->grp_ncgeom, g1)) &&
        !(in_set(s->grp_ignored, s->grp_nignored, g1) || in_set(s->grp_ignored, s->grp_nignored, g1))) { /* geom[1] twice: Q6 */
      s->grp_collision = 1;
      break;
    }
  }
  return s->grp_collision;
}

/* Sim::invoke_callbacks, sim.cpp:38-47 */
static void invoke_callbacks(orc_sim* s) {
  if (!(s->has_robot && s->robot_conv_registered)) return;
  for (int i = 0; i < 2; i++) {
    double dt = s->d.time - s->cb_last[i];
    if (dt > s->robot_period) {
      if (i == 0) robot_is_arrived_cb(s); else robot_is_moving_cb(s);
      s->cb_last[i] = s->d.time;
    }
  }
}

/* Sim::invoke_condition_callbacks + process_condition_callbacks, sim.cpp:14-23,49-61 */
static int invoke_condition_callbacks(orc_sim* s) {
  double time = s->d.time;
  int n_any = 0, n_all = 0;
  int any_id[2], all_id[2];
  double any_p[2], all_p[2];
  if (s->has_robot) { any_id[n_any] = 0; any_p[n_any++] = s->robot_period; }
  if (s->has_gripper) { any_id[n_any] = 1; any_p[n_any++] = s->grp_period; }
  if (s->has_robot && s->robot_conv_registered) { all_id[n_all] = 0; all_p[n_all++] = s->robot_period; }
  if (s->has_gripper) { all_id[n_all] = 1; all_p[n_all++] = s->grp_period; }
  for (int i = 0; i < n_any; i++) {
    int k = any_id[i];
    if (time - s->any_last[k] > any_p[i]) {
      s->any_ret[k] = k == 0 ? robot_collision_cb(s) : gripper_collision_cb(s);
      s->any_last[k] = time;
    }
  }
  for (int i = 0; i < n_all; i++) {
    int k = all_id[i];
    if (time - s->all_last[k] > all_p[i]) {
      s->all_ret[k] = k == 0 ? robot_convergence_cb(s) : gripper_convergence_cb(s);
      s->all_last[k] = time;
    }
  }
  for (int i = 0; i < n_any; i++) if (s->any_ret[any_id[i]]) return 1;
  int all = 1; /* std::all_of over an empty list is true */
  for (int i = 0; i < n_all; i++) if (!s->all_ret[all_id[i]]) all = 0;
  return all;
}

/* Sim::step, sim.cpp:108-115 (no rendering callbacks in this revision) */
void orc_sim_step(orc_sim* s, long k) {
  for (long i = 0; i < k; i++) {
    orc_step1(s->m, &s->d);
    invoke_callbacks(s);
    orc_step2(s->m, &s->d);
  }
}

/* Sim::step_until_convergence, sim.cpp:84-106 */
void orc_sim_step_until_convergence(orc_sim* s) {
  s->convergence_steps = 0;
  s->converged = 0;
  s->any_ret[0] = s->any_ret[1] = 0;
  s->all_ret[0] = s->all_ret[1] = 0;
  while (!s->converged && (s->max_convergence_steps == -1 || s->convergence_steps < s->max_convergence_steps)) {
    orc_sim_step(s, 1);
    s->convergence_steps++;
    s->converged = invoke_condition_callbacks(s);
  }
}

/* Sim::reset + reset_callbacks, sim.cpp:117-138 */
void orc_sim_reset(orc_sim* s) {
  orc_reset_data(s->m, &s->d);
  s->cb_last[0] = s->cb_last[1] = 0;
  s->any_last[0] = s->any_last[1] = 0;
  s->all_last[0] = s->all_last[1] = 0;
}

/* ---- SimRobot methods */
void orc_robot_get_joint_position(const orc_sim* s, double* q) { /* SimRobot.cpp:133-139 */
  for (int i = 0; i < s->arm_n; i++) q[i] = s->d.qpos[s->arm_jnt[i]];
}
void orc_robot_set_joint_position(orc_sim* s, const double* q) { /* SimRobot.cpp:123-131 */
  for (int i = 0; i < s->arm_n; i++) s->target_angles[i] = q[i];
  orc_robot_get_joint_position(s, s->previous_angles);
  s->is_moving = 1;
  s->is_arrived = 0;
  for (int i = 0; i < s->arm_n; i++) s->d.ctrl[s->arm_act[i]] = q[i];
}
void orc_robot_get_base_pose(const orc_sim* s, orc_pose* out) { /* SimRobot.cpp:207-213 */
  const double* xq = s->d.xquat[s->base_body]; /* wxyz */
  double q[4] = {xq[1], xq[2], xq[3], xq[0]};
  orc_pose_from_quat_t(q, s->d.xpos[s->base_body], out);
}
void orc_robot_get_cartesian_position(const orc_sim* s, orc_pose* out) { /* SimRobot.cpp:114-121, Robot.cpp:5-9 */
  orc_pose site, base, base_inv, in_robot;
  orc_pose_from_rotm_t(s->d.site_xmat[s->attachment_site], s->d.site_xpos[s->attachment_site], &site);
  orc_robot_get_base_pose(s, &base);
  orc_pose_inverse(&base, &base_inv);
  orc_pose_mul(&base_inv, &site, &in_robot);
  orc_pose_mul(&in_robot, &s->tcp_offset, out);
}
void orc_robot_set_cartesian_position(orc_sim* s, const orc_pose* pose) { /* SimRobot.cpp:145-155 */
  double q0[ORC_MAXARM], q[ORC_MAXV];
  orc_robot_get_joint_position(s, q0);
  int ok = orc_ik_inverse(&s->ik, pose, q0, s->arm_n, &s->tcp_offset, q, &s->last_ik_iterations);
  if (ok) {
    s->ik_success = 1;
    /* joint_vals has model.nq entries (Q7); only the arm entries reach ctrl */
    double qa[ORC_MAXARM];
    for (int i = 0; i < s->arm_n; i++) qa[i] = q[i];
    orc_robot_set_joint_position(s, qa);
  } else {
    s->ik_success = 0;
  }
}
void orc_robot_set_joints_hard(orc_sim* s, const double* q) { /* SimRobot.cpp:198-205 */
  for (int i = 0; i < s->arm_n; i++) {
    s->d.qpos[s->arm_jnt[i]] = q[i];
    s->d.ctrl[s->arm_act[i]] = q[i];
  }
}
void orc_robot_reset(orc_sim* s) { orc_robot_set_joints_hard(s, s->q_home); } /* SimRobot.cpp:193-196,215 */
void orc_robot_move_home(orc_sim* s) { orc_robot_set_joint_position(s, s->q_home); } /* SimRobot.cpp:47-50 */

/* ---- SimGripper methods */
int orc_gripper_set_normalized_width(orc_sim* s, double width, double force) { /* SimGripper.cpp:79-92 */
  if (width < 0 || width > 1 || force < 0) return 1; /* std::invalid_argument */
  s->last_commanded_width = width;
  s->d.ctrl[s->grp_act] = width * (s->max_actuator_width - s->min_actuator_width) + s->min_actuator_width;
  return 0;
}
double orc_gripper_get_normalized_width(const orc_sim* s) { /* SimGripper.cpp:93-106 */
  double w = (s->d.qpos[s->grp_jnt] - s->min_joint_width) / (s->max_joint_width - s->min_joint_width);
  if (w < 0) w = 0; else if (w > 1) w = 1;
  return w;
}
int orc_gripper_is_grasped(const orc_sim* s) { /* SimGripper.cpp:132-141 */
  double w = orc_gripper_get_normalized_width(s);
  return s->last_commanded_width - s->epsilon_inner < w && w < s->last_commanded_width + s->epsilon_outer;
}
void orc_gripper_reset(orc_sim* s) { /* SimGripper.cpp:158-165 */
  s->last_commanded_width = 0; s->grp_is_moving = 0; s->last_width = 0; s->grp_collision = 0;
  s->d.qpos[s->grp_jnt] = s->max_joint_width;
  s->d.ctrl[s->grp_act] = s->max_actuator_width;
}

void orc_sim_set_robot_cgeoms(orc_sim* s, int n, const int* ids) { /* SimRobot::init_ids, SimRobot.cpp:55-62 */
  s->arm_ncgeom = n;
  for (int i = 0; i < n; i++) s->arm_cgeom[i] = ids[i];
}
void orc_sim_set_gripper_cgeoms(orc_sim* s, int n, const int* cgeom, int nf, const int* cfgeom, int ni, const int* ignored) {
  s->grp_ncgeom = n; s->grp_ncfgeom = nf; s->grp_nignored = ni; /* SimGripper.cpp:31-33,57-63 */
  for (int i = 0; i < n; i++) s->grp_cgeom[i] = cgeom[i];
  for (int i = 0; i < nf; i++) s->grp_cfgeom[i] = cfgeom[i];
  for (int i = 0; i < ni; i++) s->grp_ignored[i] = ignored[i];
}

unsigned long orc_sizeof_model(void) { return sizeof(orc_model); }
unsigned long orc_sizeof_sim(void) { return sizeof(orc_sim); }
