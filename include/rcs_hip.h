/*
 * rcs_hip.h -- C-ABI of the MI355X batched simulation backend (librcs_hip.so).
 *
 * Drop-in boundary for the RCS simulation hot path.  Each entry point is the
 * N-environment form of one method the reference exposes through its pybind11
 * module `rcs._core.sim` / `rcs._core.common` (reference src/pybind/rcs.cpp);
 * the reference method it replaces is cited on every declaration.  A reference
 * maintainer binds these exactly like the existing C++ classes (INTEGRATION.md
 * shows the pybind11 and ctypes stubs).
 *
 * Conventions
 *  - plain C: opaque handle, caller-owned buffers, no exceptions.  Every call
 *    returns RCSH_OK or an error code; rcsh_last_error() gives the message of the
 *    last failure on the calling thread.  The Python shim maps
 *    RCSH_ERR_NAME -> RuntimeError and RCSH_ERR_ARG -> ValueError, the exception
 *    types the reference raises (SimRobot.cpp:57-93, SimGripper.cpp:80-83).
 *  - arrays are row-major, environment-major: q[N][dof], pose[N][7] = x y z qx qy qz qw
 *    (Eigen coeff order, reference Pose.cpp:115), flags uint8 (0/1).
 *  - `mask` (uint8[N], may be NULL = all) selects the environments a call acts on.
 *  - entry points without a suffix take HOST pointers and synchronise; `_dev`
 *    entry points take DEVICE pointers, enqueue on the handle's HIP stream and
 *    return immediately (rcsh_sim_synchronize() or a stream wait orders them).
 *  - the library never falls back to the CPU: without a usable gfx950 device
 *    rcsh_sim_create fails with RCSH_ERR_DEVICE.
 */
#ifndef RCS_HIP_H
#define RCS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RCSH_ABI_VERSION 2

enum {
  RCSH_OK = 0,
  RCSH_ERR_ARG = 1,       /* std::invalid_argument in the reference */
  RCSH_ERR_NAME = 2,      /* std::runtime_error("No joint named ...") in the reference */
  RCSH_ERR_MODEL = 3,     /* scene outside the compiled archetypes / MJCF subset */
  RCSH_ERR_DEVICE = 4,    /* no gfx950 device, HIP failure */
  RCSH_ERR_STATE = 5      /* call order violated (e.g. robot not added) */
};

/* Flat scene tables; field names follow mjModel.  Stands where the reference hands
 * `mjModel*` to `Sim(mjmdl, mjdata)` (reference src/pybind/rcs.cpp:493-497). */
typedef struct rcsh_model_desc {
  int32_t nbody, njnt, nu, ntendon, nwrap, neq, nsite;
  double timestep;
  double gravity[3];
  const int32_t* body_parentid;   /* [nbody] */
  const int32_t* body_jntadr;     /* [nbody] joint id or -1 */
  const int32_t* body_jntnum;     /* [nbody] */
  const double* body_pos;         /* [nbody][3] */
  const double* body_quat;        /* [nbody][4] wxyz */
  const double* body_ipos;        /* [nbody][3] */
  const double* body_iquat;       /* [nbody][4] */
  const double* body_mass;        /* [nbody] */
  const double* body_inertia;     /* [nbody][3] principal */
  const double* body_gravcomp;    /* [nbody] */
  const int32_t* jnt_type;        /* [njnt] 2 slide, 3 hinge */
  const int32_t* jnt_bodyid;      /* [njnt] */
  const double* jnt_pos;          /* [njnt][3] */
  const double* jnt_axis;         /* [njnt][3] */
  const int32_t* jnt_limited;     /* [njnt] */
  const double* jnt_range;        /* [njnt][2] */
  const double* jnt_margin;       /* [njnt] */
  const double* jnt_solref;       /* [njnt][2] */
  const double* jnt_solimp;       /* [njnt][5] */
  const int32_t* jnt_actfrclimited;
  const double* jnt_actfrcrange;  /* [njnt][2] */
  const int32_t* jnt_actgravcomp;
  const double* dof_armature;     /* [njnt] */
  const double* dof_damping;      /* [njnt] */
  const double* dof_frictionloss; /* [njnt] dry joint friction (soft row per dof, team kernel only) */
  const double* qpos0;            /* [njnt] */
  const int32_t* tendon_adr;      /* [ntendon] */
  const int32_t* tendon_num;      /* [ntendon] */
  const int32_t* wrap_objid;      /* [nwrap] joint ids */
  const double* wrap_prm;         /* [nwrap] coefficients */
  const int32_t* eq_obj1id;       /* [neq] joint ids */
  const int32_t* eq_obj2id;
  const int32_t* eq_active0;
  const double* eq_data;          /* [neq][5] polycoef */
  const double* eq_solref;        /* [neq][2] */
  const double* eq_solimp;        /* [neq][5] */
  const int32_t* actuator_trntype;   /* [nu] 0 joint, 3 tendon */
  const int32_t* actuator_trnid;
  const double* actuator_gear;       /* [nu] */
  const double* actuator_gainprm;    /* [nu][3] */
  const double* actuator_biasprm;    /* [nu][3] */
  const int32_t* actuator_biastype;  /* [nu] 0 none, 1 affine */
  const int32_t* actuator_ctrllimited;
  const double* actuator_ctrlrange;  /* [nu][2] */
  const int32_t* actuator_forcelimited;
  const double* actuator_forcerange; /* [nu][2] */
  const int32_t* site_bodyid;        /* [nsite] */
  const double* site_pos;            /* [nsite][3] */
  const double* site_quat;           /* [nsite][4] */
  /* collision geoms: detected against the static plane geom everywhere (collision flags); in scenes with a free box
   * (rcsh_sim_add_free_box) their contacts with the floor and the box are resolved */
  int32_t ngeom, nmeshvert;
  const int32_t* geom_type;          /* [ngeom] mjtGeom: 0 plane, 2 sphere, 3 capsule, 6 box, 7 mesh */
  const int32_t* geom_bodyid;        /* [ngeom] */
  const int32_t* geom_contype;       /* [ngeom] */
  const int32_t* geom_conaffinity;   /* [ngeom] */
  const double* geom_pos;            /* [ngeom][3] */
  const double* geom_quat;           /* [ngeom][4] */
  const double* geom_size;           /* [ngeom][3] */
  const int32_t* geom_vertadr;       /* [ngeom] first row of mesh_vert */
  const int32_t* geom_vertnum;       /* [ngeom] 0: no vertex set (geom never reports contacts) */
  const double* mesh_vert;           /* [nmeshvert][3] convex-hull vertices, geom frame */
  const double* dof_solref;          /* [njnt][2] solreffriction */
  const double* dof_solimp;          /* [njnt][5] solimpfriction */
  const double* geom_friction;       /* [ngeom][3] sliding, torsional, rolling (contacts are condim 3: only [0] is used) */
} rcsh_model_desc;

/* SimRobotConfig after name -> id lookup (reference src/sim/SimRobot.h:14-47, SimRobot.cpp:52-94). */
typedef struct rcsh_robot_desc {
  int32_t dof;
  const int32_t* joint_ids;     /* [dof] */
  const int32_t* actuator_ids;  /* [dof] */
  int32_t attachment_site;
  int32_t base_body;
  const double* q_home;         /* [dof] robots_meta_config.q_home, reference include/rcs/Robot.h:24-95 */
  double tcp_offset[7];         /* pose, xyzw quaternion */
  double joint_rotational_tolerance; /* SimRobot.h:15 */
  double seconds_between_callbacks;  /* SimRobot.h:17 */
  int32_t register_convergence_callback;
  int32_t n_collision_geoms;         /* SimRobotConfig.arm_collision_geoms after name lookup (SimRobot.cpp:55-62) */
  const int32_t* collision_geom_ids;
} rcsh_robot_desc;

/* SimGripperConfig after name -> id lookup (reference src/sim/SimGripper.h:15-45). */
typedef struct rcsh_gripper_desc {
  int32_t joint_id;
  int32_t actuator_id;
  double epsilon_inner, epsilon_outer;
  double seconds_between_callbacks;
  double max_actuator_width, min_actuator_width;
  double max_joint_width, min_joint_width;
  /* SimGripperConfig.collision_geoms / collision_geoms_fingers / ignored_collision_geoms (SimGripper.cpp:31-33,57-63) */
  int32_t n_collision_geoms, n_finger_geoms, n_ignored_geoms;
  const int32_t* collision_geom_ids;
  const int32_t* finger_geom_ids;
  const int32_t* ignored_geom_ids;
} rcsh_gripper_desc;

/* Gymnasium-loop configuration of the fused env-step (reference python/rcs/envs/creators.py:43-128). */
enum { RCSH_MODE_JOINTS = 0, RCSH_MODE_CARTESIAN_TRPY = 1, RCSH_MODE_CARTESIAN_TQUAT = 2 };
enum { RCSH_REL_NONE = 0, RCSH_REL_LAST_STEP = 1, RCSH_REL_CONFIGURED_ORIGIN = 2 };
typedef struct rcsh_env_desc {
  int32_t control_mode;
  int32_t relative_to;
  double max_mov[2];            /* joints: [0]; cartesian: translation, rotation */
  int32_t binary_gripper;       /* GripperWrapper(binary=True) */
  const double* joint_low;      /* [dof] robots_meta_config.joint_limits */
  const double* joint_high;
} rcsh_env_desc;

typedef struct rcsh_sim rcsh_sim;

const char* rcsh_last_error(void);
int rcsh_abi_version(void);
int rcsh_device_count(void);

/* rcs._core.sim.Sim(mjmdl, mjdata)  -- rcs.cpp:493-497; one handle = n_envs independent mjData */
int rcsh_sim_create(const rcsh_model_desc* model, int32_t n_envs, int32_t device, rcsh_sim** out);
void rcsh_sim_destroy(rcsh_sim* sim);
int rcsh_sim_num_envs(const rcsh_sim* sim);
int rcsh_sim_synchronize(rcsh_sim* sim);
/* the HIP stream (hipStream_t) all work of this handle is enqueued on; set_stream adopts a caller-owned
 * stream (e.g. the host framework's current stream, so its collectives order after the env-step kernel) */
void* rcsh_sim_stream(rcsh_sim* sim);
int rcsh_sim_set_stream(rcsh_sim* sim, void* hip_stream);
/* Stream ordering between handles (no reference counterpart): everything enqueued on `sim`'s stream after this call runs after
 * what `producer`'s stream holds now (an event on the producer's stream, a stream-side wait on the other: no host blocking).
 * A host that runs several handles side by side -- sub-batches of different robot types on one GPU -- orders the places where
 * they meet (a shared observation block, an exchange slot) with it, without a HIP binding of its own. */
int rcsh_sim_wait_for(rcsh_sim* sim, rcsh_sim* producer);
/* Kernel selection (no reference counterpart).  Every scene runs on the team kernel (16 lanes per environment);
 * RCSH_KERNEL_AUTO and RCSH_KERNEL_TEAM both name it.  RCSH_KERNEL_LANE, the one-lane-per-environment kernel of ABI 1, was
 * removed (it lost at every batch size and stepped neither dry friction nor free bodies): selecting it is RCSH_ERR_ARG.
 * RCSH_KERNEL_TEAM_OCC2 is the same team kernel compiled for two resident wavefronts per SIMD (<= 256 registers); it exists
 * for the launches without free box / resolved contacts / collision callbacks and falls back to TEAM elsewhere.  AUTO picks
 * by batch size; the results of the two are bit-identical. */
enum { RCSH_KERNEL_AUTO = 0, RCSH_KERNEL_TEAM = 1, RCSH_KERNEL_LANE = 2, RCSH_KERNEL_TEAM_OCC2 = 3 };
int rcsh_sim_set_kernel(rcsh_sim* sim, int32_t variant);

/* Sim.set_config / get_config -- rcs.cpp:501-502, sim.cpp:27-32; SimConfig sim.h:29-34 */
int rcsh_sim_set_config(rcsh_sim* sim, int32_t async_control, int32_t realtime, int32_t frequency,
                        int32_t max_convergence_steps);
int rcsh_sim_get_config(const rcsh_sim* sim, int32_t* async_control, int32_t* realtime, int32_t* frequency,
                        int32_t* max_convergence_steps);
/* Sim.step(k) -- rcs.cpp:503, sim.cpp:108-115 */
int rcsh_sim_step(rcsh_sim* sim, int64_t k);
/* Sim.step_until_convergence() -- rcs.cpp:498-499, sim.cpp:84-106 */
int rcsh_sim_step_until_convergence(rcsh_sim* sim);
/* Sim.is_converged() -- rcs.cpp:500, sim.cpp:83; also the substeps the last call took per env */
int rcsh_sim_is_converged(rcsh_sim* sim, uint8_t* converged, int32_t* convergence_steps);
/* Sim.reset() -- rcs.cpp:504, sim.cpp:117-138 */
int rcsh_sim_reset(rcsh_sim* sim, const uint8_t* mask);

/* SimRobot(sim, ik, cfg, register_convergence_callback) -- rcs.cpp:516-527, SimRobot.cpp:27-43 */
int rcsh_sim_add_robot(rcsh_sim* sim, const rcsh_robot_desc* robot);
/* Robot.set_joint_position -- rcs.cpp:358-359, SimRobot.cpp:123-131 */
int rcsh_robot_set_joint_position(rcsh_sim* sim, const double* q, const uint8_t* mask);
/* Robot.get_joint_position -- rcs.cpp:360, SimRobot.cpp:133-139 */
int rcsh_robot_get_joint_position(rcsh_sim* sim, double* q);
/* Robot.get_cartesian_position -- rcs.cpp:356-357, SimRobot.cpp:114-121 */
int rcsh_robot_get_cartesian_position(rcsh_sim* sim, double* pose);
/* Robot.get_base_pose_in_world_coordinates -- rcs.cpp:369-370, SimRobot.cpp:207-213 */
int rcsh_robot_get_base_pose(rcsh_sim* sim, double* pose);
/* Robot.set_cartesian_position -- rcs.cpp:365-367, SimRobot.cpp:145-155 (Pin CLIK, Kinematics.cpp:28-68) */
int rcsh_robot_set_cartesian_position(rcsh_sim* sim, const double* pose, const uint8_t* mask);
/* SimRobot.set_joints_hard -- rcs.cpp:525, SimRobot.cpp:198-205 */
int rcsh_robot_set_joints_hard(rcsh_sim* sim, const double* q, const uint8_t* mask);
/* Robot.reset / move_home -- rcs.cpp:361-363, SimRobot.cpp:47-50,193-196 */
int rcsh_robot_reset(rcsh_sim* sim, const uint8_t* mask);
int rcsh_robot_move_home(rcsh_sim* sim, const uint8_t* mask);
/* SimRobot.get_state -- rcs.cpp:527, SimRobotState SimRobot.h:49-57; any pointer may be NULL */
int rcsh_robot_get_state(rcsh_sim* sim, uint8_t* ik_success, uint8_t* collision, uint8_t* is_moving,
                         uint8_t* is_arrived, double* previous_angles, double* target_angles);
/* Kinematics.inverse / forward on the robot's own chain -- rcs.cpp:289-300, Kinematics.cpp:28-82.
 * q0 [N][dof] -> q [N][nq] (nq = model dofs, quirk Q7), success [N], iterations [N] (may be NULL) */
int rcsh_ik_inverse(rcsh_sim* sim, const double* pose, const double* q0, const double* tcp_offset7, double* q,
                    uint8_t* success, int32_t* iterations);
int rcsh_ik_forward(rcsh_sim* sim, const double* q0, const double* tcp_offset7, double* pose);

/* SimGripper(sim, cfg) -- rcs.cpp:508-515, SimGripper.cpp:13-39 */
int rcsh_sim_add_gripper(rcsh_sim* sim, const rcsh_gripper_desc* gripper);
/* Gripper.set_normalized_width -- rcs.cpp:383-384, SimGripper.cpp:79-92 (RCSH_ERR_ARG outside [0,1] / force<0) */
int rcsh_gripper_set_normalized_width(rcsh_sim* sim, const double* width, double force, const uint8_t* mask);
/* Gripper.get_normalized_width -- rcs.cpp:385, SimGripper.cpp:93-106 */
int rcsh_gripper_get_normalized_width(rcsh_sim* sim, double* width);
/* Gripper.is_grasped -- rcs.cpp:388, SimGripper.cpp:132-141 */
int rcsh_gripper_is_grasped(rcsh_sim* sim, uint8_t* grasped);
/* Gripper.reset -- rcs.cpp:395, SimGripper.cpp:158-165 */
int rcsh_gripper_reset(rcsh_sim* sim, const uint8_t* mask);
/* SimGripper.get_state -- rcs.cpp:514, SimGripperState SimGripper.h:47-52 */
int rcsh_gripper_get_state(rcsh_sim* sim, double* last_commanded_width, uint8_t* is_moving, double* last_width,
                           uint8_t* collision);

/* mjData views the reference reads directly (python/rcs/envs/sim.py:343-411): qpos, qvel, ctrl, time */
int rcsh_sim_get_qpos(rcsh_sim* sim, double* qpos);  /* [N][nq] */
int rcsh_sim_get_qvel(rcsh_sim* sim, double* qvel);
int rcsh_sim_get_ctrl(rcsh_sim* sim, double* ctrl);  /* [N][nu] */
int rcsh_sim_get_time(rcsh_sim* sim, double* time);  /* [N] */
int rcsh_sim_set_qpos(rcsh_sim* sim, const double* qpos, const uint8_t* mask);
int rcsh_sim_set_qvel(rcsh_sim* sim, const double* qvel, const uint8_t* mask);
int rcsh_sim_nq(const rcsh_sim* sim);
int rcsh_sim_nu(const rcsh_sim* sim);

/* One free-floating box on the floor plane -- the `box_geom` body of the reference's pick-up scene
 * (assets/scenes/fr3_simple_pick_up/scene.xml:30-33), whose free joint the task wrappers write and read through
 * mjData: RandomCubePos `sim.data.joint("box_joint").qpos = [...]` (python/rcs/envs/sim.py:379-383),
 * PickCubeSuccessWrapper `sim.data.joint("box_joint").qpos[2]` / `[:3]` (sim.py:399-412).  The description carries
 * the mjModel constants of that body and of its contact pair with the floor (mj_contactParam already applied:
 * friction = element-wise max, solref / solimp mixed) plus the solver options the contact solve reads
 * (assets/fr3/mjcf/fr3_common.xml:3).  qpos is [x y z qw qx qy qz], qvel [linear (world), angular (body frame)].
 * With `resolve_robot_contacts` the robot's collision geoms (finger pads, finger / hand / link hulls, the camera capsule)
 * collide with the box and with the floor, and one constraint problem couples the robot's joints with the box's 6 dofs
 * (FR3 + hand archetype); without it the box touches the floor only. */
typedef struct rcsh_free_box_desc {
  double qpos0[7];
  double mass, inertia[3];   /* centre of mass at the body origin, principal axes = body axes */
  double size[3];            /* half extents of the box geom */
  double friction[3];        /* of the floor / box pair */
  double solref[2], solimp[5];
  double plane_z;            /* floor: z = plane_z, normal +z */
  double impratio;           /* mjOption.impratio */
  double noslip_tolerance;   /* mjOption.noslip_tolerance */
  int32_t noslip_iterations; /* mjOption.noslip_iterations */
  int32_t cone_elliptic;     /* mjOption.cone == mjCONE_ELLIPTIC (the only cone type built) */
  double geom_friction[3];   /* the box geom's own coefficients (mixed with a robot geom's per contact: element-wise max) */
  double floor_friction[3];  /* the floor geom's */
  int32_t resolve_robot_contacts;
  int32_t reserved;
} rcsh_free_box_desc;
int rcsh_sim_add_free_box(rcsh_sim* sim, const rcsh_free_box_desc* box);
/* Scenes WITHOUT a free body: the solver options mjModel.opt carries (reference assets/fr3/mjcf/fr3_common.xml:3) and the
 * default contact parameters, so that contacts of the robot's collision geoms with the floor are resolved (the arm stops
 * on the floor instead of only raising SimRobot's collision flag).  FR3 + hand archetype; elsewhere and with
 * resolve_robot_contacts = 0 robot contacts are detected only. */
typedef struct rcsh_contact_options {
  double impratio, noslip_tolerance;
  int32_t noslip_iterations, cone_elliptic;
  double solref[2], solimp[5];
  /* bit 0: contacts of the robot's geoms with the floor enter the constraint solve (0: detected only).  Bit 1 (round 5): so do
   * contacts between two geoms of the robot -- mj_step2 resolves every entry of mjData.contact (reference src/sim/sim.cpp:108-115).
   * Bit 2 (round 5): environment by environment -- a step is the lean launch over the environments that touch nothing plus the
   * contact-resolving launch over the others; an environment whose geoms are found in contact at the end of a lean launch has
   * that launch redone with its contacts resolved from their first substep (csrc/sim_kernels.h: RunOp::esc_role).  Without bit 2
   * the whole batch runs on the contact-resolving kernel. */
  int32_t resolve_robot_contacts, reserved;
} rcsh_contact_options;
int rcsh_sim_set_contact_options(rcsh_sim* sim, const rcsh_contact_options* options);
/* Per-environment escalation (rcsh_contact_options.resolve_robot_contacts bit 2), [N] flags each, either may be null:
 * `now`: the environment is stepped by the contact-resolving kernel at present; `ever`: a contact of its robot geoms has been
 * resolved since its last rcsh_sim_reset.  All zero where escalation is off. */
int rcsh_sim_contact_escalated(rcsh_sim* sim, uint8_t* now, uint8_t* ever);
/* [N] flags: the environment's geoms were found in a contact this configuration does not resolve -- robot <-> floor in scenes
 * that only detect contacts (the default without a free body), robot <-> robot everywhere -- at the end of a stepping launch,
 * since its last rcsh_sim_reset.  MuJoCo resolves every contact of d->contact in every mj_step2 (reference src/sim/sim.cpp:
 * 108-115), so from that launch on the environment's trajectory is not the reference's; the same bit is byte 7 of an
 * env-step's info row.  Checked once per launch on the position the next position stage will see (csrc/check_team.h). */
int rcsh_sim_contact_unresolved(rcsh_sim* sim, uint8_t* unresolved);
/* How often the check runs: at the end of every `every`-th stepping launch (default 1: every launch, the flag comes on in the
 * env-step the contact begins in; 0: never).  The check costs a launch about as much as four physics substeps (~25 us at 4096 environments); a resident
 * rollout that prefers throughput sets a larger cadence -- a contact is then found with a lag of up to `every` - 1 launches, one
 * that comes and goes between two checks is missed -- and rcsh_sim_contact_unresolved() always checks the present state first. */
int rcsh_sim_set_contact_check(rcsh_sim* sim, int32_t every);
/* Admitted geom pairs beyond the 192 the end-of-launch check (and the self-contact stage of the contact-resolving kernels) keeps
 * entries for: they are neither checked nor resolved; the handle is created all the same and the Python host warns.  0 for every
 * shipped scene (fr3_empty_world: 137 pairs).  No reference counterpart: MuJoCo has no such capacity. */
int rcsh_sim_contact_check_unchecked_pairs(rcsh_sim* sim, int32_t* count);
/* Collision geoms of the scene that exceed the contact table's capacity (28 geoms -- 32 until round 4, when the end-of-launch
 * check took LDS for the geoms' world boxes --, 10 boxes, 152 hull vertices) are left out of
 * GEOM-GEOM detection (the floor test still sees them): their mjModel ids (up to `capacity`), how many there are, and why.
 * rcsh_sim_add_robot / rcsh_sim_add_gripper refuse a collision geom that is on this list; the Python host warns about the
 * rest (a world-welded obstacle no callback list names would still count for SimRobot::collision_callback,
 * reference src/sim/SimRobot.cpp:172-182).  No reference counterpart: MuJoCo has no such capacity. */
int rcsh_sim_contact_table_dropped(rcsh_sim* sim, int32_t* geom_ids, int32_t capacity, int32_t* count, char* reason, size_t reason_capacity);
int rcsh_sim_reset_free_box(rcsh_sim* sim);
int rcsh_sim_get_free_qpos(rcsh_sim* sim, double* qpos);  /* [N][7] */
int rcsh_sim_get_free_qvel(rcsh_sim* sim, double* qvel);  /* [N][6] */
int rcsh_sim_set_free_qpos(rcsh_sim* sim, const double* qpos, const uint8_t* mask);
int rcsh_sim_set_free_qvel(rcsh_sim* sim, const double* qvel, const uint8_t* mask);

/* Snapshot / restore of EVERYTHING that evolves (the mjData fields above, the callback scheduler's timestamps and
 * return values, SimRobot / SimGripper state, the wrappers' prev_action / origin / last_action, flags): the reference's
 * closest facility is the GUI bridge's mjSTATE_FULLPHYSICS copy (src/sim/gui_server.cpp:49-60).  The blob is opaque,
 * `rcsh_sim_state_bytes` long, and valid for handles created from the same scene with the same n_envs (it begins with a
 * 16-byte header -- layout version, n_envs, number of state fields -- and rcsh_sim_set_state refuses a blob whose header says
 * otherwise: RCSH_ERR_ARG); restoring and re-running the same calls reproduces the continuation bit for bit. */
size_t rcsh_sim_state_bytes(const rcsh_sim* sim);
int rcsh_sim_get_state(rcsh_sim* sim, void* blob);
int rcsh_sim_set_state(rcsh_sim* sim, const void* blob);

/* ---- fused Gymnasium loop: SimEnvCreator()(...).reset() / .step(action) for N environments in one launch.
 * Wrapper stack restated in the kernel (reference python/rcs/envs/base.py:246-304,469-565,680-735;
 * envs/sim.py:49-76,119-131).  Observation row: tquat[7] joints[dof] xyzrpy[6] gripper[1]
 * (obs_width = 14 + dof); info row (uint8[8]): collision, ik_success, is_sim_converged, is_grasped,
 * truncated, gripper_collision, contact_overflow, contact_unresolved (the last two: no reference counterpart -- a contact
 * phase ran out of slots / the environment was found in a contact this configuration does not resolve, sticky until reset,
 * see rcsh_sim_contact_unresolved) -- `info` rows are written as one 8-byte word each: the buffer must be 8-byte aligned --;
 * gripper_width[N] (double). */
int rcsh_env_configure(rcsh_sim* sim, const rcsh_env_desc* env);
int rcsh_env_obs_width(const rcsh_sim* sim);
int rcsh_env_action_width(const rcsh_sim* sim);
int rcsh_env_reset(rcsh_sim* sim, const uint8_t* mask, double* obs, uint8_t* info, double* gripper_width);
int rcsh_env_step(rcsh_sim* sim, const double* action, const float* gripper, double* obs, uint8_t* info,
                  double* gripper_width, int32_t* substeps);
/* device-pointer forms used by the rollout loop (nothing crosses PCIe) */
int rcsh_env_reset_dev(rcsh_sim* sim, const uint8_t* mask_dev, double* obs_dev, uint8_t* info_dev,
                       double* gripper_width_dev);
int rcsh_env_step_dev(rcsh_sim* sim, const double* action_dev, const float* gripper_dev, double* obs_dev,
                      uint8_t* info_dev, double* gripper_width_dev, int32_t* substeps_dev);

/* Task layer of the pick-up scene: SimTaskEnvCreator()(...) = SimEnvCreator with RandomCubePos under the
 * RobotSimWrapper and PickCubeSuccessWrapper on top (python/rcs/envs/creators.py:131-187; the registered gym id
 * rcs/FR3SimplePickUpSim-v0, python/rcs/__init__.py:67-70).
 *  - reset: RandomCubePos.reset (python/rcs/envs/sim.py:365-383) runs the inner reset, steps once, writes the box
 *    pose and the RobotSimWrapper steps once more; `box_qpos` [N][7] is the pose each environment writes (the
 *    reference draws x, y and the quaternion's w from numpy's global generator; the caller draws them here).
 *  - step: PickCubeSuccessWrapper.step (sim.py:396-431); `task` [N][9] receives the box pose (7), the reward and
 *    success (0/1 -- the wrapper returns it as `terminated` and info["success"]).
 * ee_home / success_height are the wrapper's constants EE_HOME and 0.15 + 0.852. */
typedef struct rcsh_pick_task_desc {
  double ee_home[3];
  double success_height;
} rcsh_pick_task_desc;
int rcsh_env_configure_pick_task(rcsh_sim* sim, const rcsh_pick_task_desc* task);
int rcsh_env_reset_task(rcsh_sim* sim, const uint8_t* mask, const double* box_qpos, double* obs, uint8_t* info, double* gripper_width);
int rcsh_env_step_task(rcsh_sim* sim, const double* action, const float* gripper, double* obs, uint8_t* info, double* gripper_width,
                       int32_t* substeps, double* task);
int rcsh_env_reset_task_dev(rcsh_sim* sim, const uint8_t* mask_dev, const double* box_qpos_dev, double* obs_dev, uint8_t* info_dev,
                            double* gripper_width_dev);
int rcsh_env_step_task_dev(rcsh_sim* sim, const double* action_dev, const float* gripper_dev, double* obs_dev, uint8_t* info_dev,
                           double* gripper_width_dev, int32_t* substeps_dev, double* task_dev);


/* Depth images: the pixel path of SimCameraSet (reference src/sim/camera.cpp:86-140 render_single -> mjv_updateScene,
 * mjr_render, mjr_readPixels; python/rcs/camera/sim.py:45-115 for the row flip, the conversion to metres and the
 * uint16 millimetre image, the intrinsics and the extrinsics).  The reference rasterises MuJoCo's visual geoms with
 * OpenGL; this backend casts one ray per pixel against the shapes given here (floor plane, boxes, convex hulls of the
 * collision meshes), expressed in the frame of the link they ride on (-1: world, -2: the free box).  Colour: rcsh_sim_set_render_colours.
 * Frames are those of the last position stage, as mjData.geom_xpos / cam_xpos are. */
typedef struct rcsh_render_scene_desc {
  int32_t nshape, nplanes;
  const int32_t* shape;      /* [nshape] 0 plane (z = 0 of the shape frame, seen from +z), 1 box, 2 convex hull, 3 capsule (axis z) */
  const int32_t* link;       /* [nshape] */
  const double* pos;         /* [nshape][3] shape frame in the link frame */
  const double* rot;         /* [nshape][9] row-major */
  const double* size;        /* [nshape][3] box half extents; hulls: half extents of the bounding box centred at sphere[0..2]; capsules: (r, r, r + half length) */
  const int32_t* plane_adr;  /* [nshape] hulls: first row of `planes` */
  const int32_t* plane_num;  /* [nshape] */
  const double* sphere;      /* [nshape][4] bounding sphere, shape frame: centre, radius (< 0: unbounded) */
  const double* planes;      /* [nplanes][4] n . x <= d */
  double znear, zfar;        /* mjModel.vis.map.znear / zfar times mjModel.stat.extent */
} rcsh_render_scene_desc;
typedef struct rcsh_camera_desc {
  int32_t link, width, height;  /* mjModel.cam_bodyid folded to its link; resolution of SimCameraConfig */
  double pos[3], rot[9];        /* camera frame in the link frame (mjModel.cam_pos / cam_quat) */
  double fovy_deg;              /* mjModel.cam_fovy */
} rcsh_camera_desc;
int rcsh_sim_set_render_scene(rcsh_sim* sim, const rcsh_render_scene_desc* scene);
int rcsh_sim_add_camera(rcsh_sim* sim, const rcsh_camera_desc* cam, int32_t* cam_id);
/* Host only (no device needed): the edges of the convex polytope { x : n_i . x <= d_i } that `planes` ([nplanes][4]) describe --
 * what rcsh_sim_set_render_scene works out for every hull so that the ray caster can find the hull's outline as the camera
 * sees it (a ray hits the hull exactly when it passes inside the outline; its entry depth comes from the planes facing the
 * camera alone).  edge_planes: [capacity][2] the two planes meeting in the edge, edge_verts: [capacity][6] its end points (both may
 * be NULL to ask for the count), centre: [3] a point inside.  *nedges = 0: the planes bound no polytope the routine vouches
 * for (Euler's formula failed on what it found); such a hull is drawn by walking all its planes. */
int rcsh_hull_edges(const double* planes, int32_t nplanes, int32_t capacity, int32_t* edge_planes, double* edge_verts, int32_t* nedges, double* centre);
/* One image per environment.  depth_gl: [N][H][W] f32 in [0, 1], rows bottom-up, what mjr_readPixels returns;
 * depth_mm: [N][H][W] u16, rows top-down, millimetres: DataFrame.data of SimCameraSet(physical_units=True);
 * cam_pose: [N][12] mjData.cam_xmat (9) + cam_xpos (3).  Each may be NULL.  _dev: device pointers, enqueued on the
 * handle's stream. */
int rcsh_camera_render(rcsh_sim* sim, int32_t cam_id, float* depth_gl, uint16_t* depth_mm, double* cam_pose);
int rcsh_camera_render_dev(rcsh_sim* sim, int32_t cam_id, float* depth_gl_dev, uint16_t* depth_mm_dev, double* cam_pose_dev);
/* Colour frames (the ColorFrame of FrameSet, src/sim/camera.cpp:103-118: mjr_readPixels' rgb buffer).  The image is
 * ray-cast like the depth: the colour of the shape a ray enters first (mjModel.geom_rgba / mat_rgba; a material with the
 * builtin checker texture alternates its two colours in squares), flat-shaded on the entry face's normal by the headlight
 * (mjModel.vis.headlight ambient / diffuse) and one directional light of the scene; rays that leave the scene see the
 * skybox gradient.  No textures beyond the checker, no shadows, no specular term, no transparency: not OpenGL's pixels.
 * colour: [nshape][8] = rgb (3), second checker colour (3), edge of a checker square [m], checker flag (0 / 1). */
typedef struct rcsh_render_colours {
  const double* colour;
  double headlight_ambient[3], headlight_diffuse[3];
  double light_dir[3], light_diffuse[3]; /* directional light (world frame, pointing away from the light); diffuse 0: none */
  double sky_rgb1[3], sky_rgb2[3];       /* zenith / nadir colour of the background */
} rcsh_render_colours;
int rcsh_sim_set_render_colours(rcsh_sim* sim, const rcsh_render_colours* colours);
/* rgb: [N][H][W][3] u8, rows bottom-up (as mjr_readPixels returns them); the depth outputs as above; each may be NULL. */
int rcsh_camera_render_rgb(rcsh_sim* sim, int32_t cam_id, uint8_t* rgb, float* depth_gl, uint16_t* depth_mm, double* cam_pose);
/* The ray caster's arithmetic type.  Default (0): float32 -- the reference's depth image is a float32 OpenGL z-buffer read back
 * and quantised to uint16 millimetres (python/rcs/camera/sim.py:57-86), src/sim/camera.cpp:86-140 computes nothing in double.
 * 1: every ray in double -- the instantiation whose pixels equal the tests' numpy restatement bit for bit (about 1.4x the time).
 * (RCSH_RENDER_F64=1 in the environment selects it at rcsh_sim_create.) */
int rcsh_sim_set_render_f64(rcsh_sim* sim, int32_t on);
int rcsh_camera_render_rgb_dev(rcsh_sim* sim, int32_t cam_id, uint8_t* rgb_dev, float* depth_gl_dev, uint16_t* depth_mm_dev, double* cam_pose_dev);
/* Rendering callbacks: SimCameraSet(render_on_demand = false) (src/sim/camera.cpp:23-47,97-102; Sim::register_rendering_callback,
 * invoke_rendering_callbacks, reset_callbacks: src/sim/sim.cpp:63-81,108-115,131-137,160-173).  A camera with a frame rate is due
 * after a substep when more than seconds_between_calls (= 1 / frame_rate) of simulated time passed since its last frame; the
 * clocks start at -seconds_between_calls, and again after Sim::reset, so the first substep renders.  A kernel cannot call the
 * renderer: the substep loop RECORDS per environment what the renderer would have seen at that moment (the kinematic state
 * of the substep's position stage, the new time, which cameras are due), at most `capacity` records per environment and
 * launch, and the host renders them afterwards:
 *   launch (any stepping / env entry point) -> rcsh_render_pending(count[N]) -> for slot < max(count):
 *   rcsh_camera_render_snapshot(cam, slot, ..., timestamp[N], due[N]) for each scheduled camera.
 * Images of environments whose `due` is 0 for that slot and camera are meaningless.  ncam = 0 removes the schedule. */
int rcsh_sim_set_render_schedule(rcsh_sim* sim, const int32_t* cam_ids, const double* seconds_between_calls, int32_t ncam, int32_t capacity);
int rcsh_render_pending(rcsh_sim* sim, int32_t* count);
/* Calling rcsh_sim_set_render_schedule again with the SAME cameras and periods and a larger capacity grows the schedule in
 * place: clocks and pending records stay.  Records beyond the capacity are not written; rcsh_render_pending clamps its counts
 * to the capacity and adds what was lost to a counter that rcsh_render_dropped reads (reset with the schedule). */
int rcsh_render_dropped(rcsh_sim* sim, int64_t* dropped);
int rcsh_camera_render_snapshot(rcsh_sim* sim, int32_t cam_id, int32_t slot, uint8_t* rgb, float* depth_gl, uint16_t* depth_mm, double* cam_pose,
                                double* timestamp, uint8_t* due);

/* ---- multi-GPU: one process (one handle) per GPU, contiguous ranges of environments per rank, no data-path collective
 * except ONE exchange step: the all-gather of the observation tensor [n][obs_width] f64 of every rank into
 * [world * n][obs_width] on every GPU (the reference holds one model / data pair per Sim, src/sim/sim.h:75-77,
 * and has nothing to exchange).  RCCL (librccl.so, loaded on first use) over xGMI; no host framework involved:
 *   rank 0: rcsh_comm_get_unique_id -> ship the 128 bytes to the other ranks by any side channel -> every rank:
 *   rcsh_comm_init(sim, id, rank, world).
 * rcsh_env_allgather_obs_dev is ordered after the work already enqueued on the handle's stream (an event) but runs on
 * the communicator's OWN stream, so the next env-step overlaps it; the caller alternates two send / receive buffer pairs
 * (`slot` 0 / 1) and calls rcsh_comm_wait(slot) before it reads that slot's gathered tensor or overwrites its send buffer. */
#define RCSH_COMM_ID_BYTES 128
int rcsh_comm_get_unique_id(uint8_t id[RCSH_COMM_ID_BYTES]);
int rcsh_comm_init(rcsh_sim* sim, const uint8_t id[RCSH_COMM_ID_BYTES], int32_t rank, int32_t world);
int rcsh_comm_rank(const rcsh_sim* sim, int32_t* rank, int32_t* world);
int rcsh_env_allgather_obs_dev(rcsh_sim* sim, int32_t slot, const double* local_obs_dev, double* all_obs_dev);
int rcsh_comm_allgather_dev(rcsh_sim* sim, int32_t slot, const void* send_dev, void* recv_dev, size_t bytes_per_rank);
int rcsh_comm_wait(rcsh_sim* sim, int32_t slot, int32_t block_host); /* 0: the handle's stream waits for the slot's gather; 1: the host does */
int rcsh_comm_destroy(rcsh_sim* sim);
/* The same exchange over COPY ENGINES instead of RCCL's collective kernel (which needs 261-280 registers a lane and finds no SIMD free
 * beside a full batch's stepping wavefronts: its gather then starts when the env-step ends).  Every rank writes its block into the
 * receive buffer of every peer with asynchronous copies (SDMA over xGMI, one stream per peer), followed by an 8-byte sequence number
 * into a flag word of the peer; one 64-lane wavefront per gather waits for the flag words.  Set-up:
 *   every rank: rcsh_comm_copy_create(sim, rank, world, bytes_per_rank, blob)  -- allocates the two receive buffers and the flags,
 *   exports them; all-gather the blobs over any side channel; every rank: rcsh_comm_copy_connect(sim, blobs[world]).
 * Then rcsh_comm_allgather_dev / rcsh_env_allgather_obs_dev / rcsh_comm_wait / rcsh_comm_destroy as with RCCL, with ONE difference:
 * the receive buffer of a slot is the carrier's (rcsh_comm_copy_recv_buffer), since peers have to have it mapped.  World <= 16,
 * one rank per process. */
#define RCSH_COMM_COPY_BLOB_BYTES 256
int rcsh_comm_copy_create(rcsh_sim* sim, int32_t rank, int32_t world, size_t bytes_per_rank, uint8_t blob[RCSH_COMM_COPY_BLOB_BYTES]);
int rcsh_comm_copy_connect(rcsh_sim* sim, const uint8_t* blobs /* [world][RCSH_COMM_COPY_BLOB_BYTES], by rank */);
int rcsh_comm_copy_recv_buffer(rcsh_sim* sim, int32_t slot, void** recv_dev);

/* device allocation helpers so a host language without a HIP binding can keep rollouts resident */
int rcsh_dev_alloc(rcsh_sim* sim, size_t bytes, void** ptr);
int rcsh_dev_free(rcsh_sim* sim, void* ptr);
int rcsh_dev_upload(rcsh_sim* sim, void* dst_dev, const void* src_host, size_t bytes);
int rcsh_dev_download(rcsh_sim* sim, void* dst_host, const void* src_dev, size_t bytes);

/* development hook: copies the finalised device model tables (csrc/model.h DevModel) to `buf`; used by
 * tools/kbench to replay the exact FR3 tables in kernel micro-benchmarks */
int rcsh_debug_dump_model(rcsh_sim* sim, void* buf, size_t cap, size_t* size);

/* kernel timing hooks for bench.py: HIP events on the handle's stream around every `enable`-th fused env-step launch
 * (1: every launch; 0: off).  A pair of event records around EVERY launch costs ~8 us of dispatch gap per step.
 * enable < 0: REGION mode -- one event before the first stepping launch after this call, one when rcsh_prof_read is called;
 * read returns the stream time between them and the number of stepping launches it covers (the launches' durations plus the
 * dispatch gaps between them: an upper bound of the kernel's average duration, no event traffic inside the region). */
int rcsh_prof_enable(rcsh_sim* sim, int32_t enable);
int rcsh_prof_read(rcsh_sim* sim, double* total_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* RCS_HIP_H */
