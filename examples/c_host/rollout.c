/* A host of the batched backend in plain C: nothing but include/rcs_hip.h and librcs_hip.so -- no Python, no torch, no pybind.
 * What the reference does as
 *     env = SimEnvCreator()(ControlMode.JOINTS, default_sim_robot_cfg(...), gripper_cfg=default_sim_gripper_cfg(),
 *                           max_relative_movement=np.deg2rad(5), relative_to=RelativeTo.LAST_STEP)
 *     obs, info = env.reset();  obs, ... = env.step({"joints": a, "gripper": g})      (examples/fr3/fr3_env_joint_control.py:34-60)
 * for N environments at once.  The scene's tables come from model.inc (tools/export_model_c.py; with MuJoCo present they
 * are the fields of an mjModel).  Prints the observations of the first and the last environment after every step as hex
 * doubles; tests/test_gpu_parity.py::test_c_host_equals_python_host compares them with the Python host's, bit for bit.
 *
 *   python tools/export_model_c.py > examples/c_host/model.inc
 *   gcc -O2 -std=c11 -Iinclude examples/c_host/rollout.c -Lrobot-control-stack_amd/rcs_amd -lrcs_hip -Wl,-rpath,$PWD/robot-control-stack_amd/rcs_amd -lm -o rollout
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rcs_hip.h"
#include "model.inc"

#define CHECK(call)                                                          \
  do {                                                                       \
    int rc_ = (call);                                                        \
    if (rc_ != RCSH_OK) {                                                    \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, rcsh_last_error());      \
      return 1;                                                              \
    }                                                                        \
  } while (0)

/* the action stream both hosts replay: a 64-bit LCG mapped to [-1, 1) */
static uint64_t lcg_state = 0x9e3779b97f4a7c15ull;
static double lcg_unit(void) {
  lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull;
  return (double)(lcg_state >> 11) / 9007199254740992.0 * 2.0 - 1.0;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 64, steps = argc > 2 ? atoi(argv[2]) : 5;
  rcsh_model_desc model;
  memset(&model, 0, sizeof(model));
  fill_model(&model);
  rcsh_sim* sim = NULL;
  CHECK(rcsh_sim_create(&model, n, 0, &sim));
  CHECK(rcsh_sim_set_config(sim, /*async_control*/ 1, /*realtime*/ 0, /*frequency*/ 30, /*max_convergence_steps*/ 500));

  rcsh_robot_desc robot;
  memset(&robot, 0, sizeof(robot));
  robot.dof = 7;
  robot.joint_ids = m_robot_joints;
  robot.actuator_ids = m_robot_actuators;
  robot.attachment_site = M_SITE;
  robot.base_body = M_BASE;
  robot.q_home = m_q_home;
  robot.tcp_offset[6] = 1.0;                          /* identity pose (x y z, qx qy qz qw) */
  robot.joint_rotational_tolerance = 0.05 * 3.14159265358979323846 / 180.0; /* SimRobot.h:15 */
  robot.seconds_between_callbacks = 0.1;              /* SimRobot.h:17 */
  robot.register_convergence_callback = 1;
  robot.n_collision_geoms = 8;
  robot.collision_geom_ids = m_arm_geoms;
  CHECK(rcsh_sim_add_robot(sim, &robot));

  rcsh_gripper_desc grip;
  memset(&grip, 0, sizeof(grip));
  grip.joint_id = M_GRIPPER_JOINT;
  grip.actuator_id = M_GRIPPER_ACTUATOR;
  grip.epsilon_inner = grip.epsilon_outer = 0.005;    /* SimGripper.h:15-23 */
  grip.seconds_between_callbacks = 0.05;
  grip.max_actuator_width = 255; grip.min_actuator_width = 0;
  grip.max_joint_width = 0.04; grip.min_joint_width = 0;
  grip.n_collision_geoms = 4; grip.collision_geom_ids = m_gripper_geoms;
  grip.n_finger_geoms = 2; grip.finger_geom_ids = m_finger_geoms;
  grip.n_ignored_geoms = 0; grip.ignored_geom_ids = m_finger_geoms;
  CHECK(rcsh_sim_add_gripper(sim, &grip));

  rcsh_env_desc env;
  memset(&env, 0, sizeof(env));
  env.control_mode = RCSH_MODE_JOINTS;
  env.relative_to = RCSH_REL_LAST_STEP;
  env.max_mov[0] = 5.0 * 3.14159265358979323846 / 180.0;
  env.binary_gripper = 1;
  env.joint_low = m_joint_low;
  env.joint_high = m_joint_high;
  CHECK(rcsh_env_configure(sim, &env));

  const int ow = rcsh_env_obs_width(sim), aw = rcsh_env_action_width(sim);
  double* obs = calloc((size_t)n * ow, sizeof(double));
  double* action = calloc((size_t)n * aw, sizeof(double));
  double* width = calloc((size_t)n, sizeof(double));
  float* gripper = calloc((size_t)n, sizeof(float));
  uint8_t* info = calloc((size_t)n * 8, 1);
  int32_t* substeps = calloc((size_t)n, sizeof(int32_t));
  CHECK(rcsh_env_reset(sim, NULL, obs, info, width));
  for (int t = 0; t <= steps; ++t) {
    if (t > 0) {
      for (int e = 0; e < n; ++e) {
        for (int k = 0; k < aw; ++k) action[(size_t)e * aw + k] = env.max_mov[0] * lcg_unit();
        gripper[e] = lcg_unit() > 0 ? 1.0f : 0.0f;
      }
      CHECK(rcsh_env_step(sim, action, gripper, obs, info, width, substeps));
    }
    for (int pick = 0; pick < 2; ++pick) {
      const int e = pick ? n - 1 : 0;
      printf("step %d env %d substeps %d width %a obs", t, e, t ? substeps[e] : 0, width[e]);
      for (int k = 0; k < ow; ++k) printf(" %a", obs[(size_t)e * ow + k]);
      printf("\n");
    }
  }
  rcsh_sim_destroy(sim);
  free(obs); free(action); free(width); free(gripper); free(info); free(substeps);
  return 0;
}
