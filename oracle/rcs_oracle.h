/*
 * rcs_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar double-precision CPU restatement of the reference hot path
 * (rcs.sim.Sim.step / step_until_convergence + SimRobot + SimGripper + Pose +
 * Pin CLIK).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; nothing under robot-control-stack_amd/ links,
 * imports or calls it.
 *
 * PARITY UNPINNED AGAINST MUJOCO for the physics: the arithmetic of
 * mj_step1/mj_step2 lives in third-party MuJoCo 3.2.6 (reference
 * pyproject.toml:23), absent from /root/reference and from this image.
 * orc_step1/orc_step2 restate MuJoCo's published forward-dynamics pipeline
 * (kinematics, comPos, CRBA, RNE, passive, actuation, soft constraints,
 * implicitfast) for the RCS scenes.  What pins them instead: the reference's own
 * tests (tests/test_oracle_pins.py lists them); since round 3 an independent
 * first-principles derivation of the robots' dynamics from the reference's MJCF
 * constants (tests/golden/{fr3,fr3_arm,xarm7}_dynamics.json,
 * tools/derive_fr3_dynamics.py: mass matrix, bias, gravity compensation,
 * actuation, one implicitfast step, held at 1e-10) and closed forms of the
 * soft-constraint model (tests/test_closed_forms.py).
 * The RCS-side semantics (callback scheduler, SimRobot, SimGripper, Pose) are
 * restated line by line from sources that ARE under /root/reference; each
 * function cites the file:line it follows.
 */
#ifndef RCS_ORACLE_H
#define RCS_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAXBODY 32
#define ORC_MAXV 16
#define ORC_MAXU 16
#define ORC_MAXEQ 4
#define ORC_MAXTENDON 4
#define ORC_MAXWRAP 8
#define ORC_MAXSITE 32
#define ORC_MAXCON 256 /* storage only: MuJoCo's contact list has no fixed bound; orc_set_contact_cap() truncates for tests that mirror a backend's capacity */
#define ORC_MAXEFC (ORC_MAXEQ + 3 * ORC_MAXV + 3 * ORC_MAXCON)
#define ORC_NVT (ORC_MAXV + 6) /* dofs of the coupled system: the robot's joints, then the free box's 6 */
#define ORC_MAXARM 8
#define ORC_MAXGEOM 32
#define ORC_MAXCGEOM 16

enum { ORC_JNT_SLIDE = 2, ORC_JNT_HINGE = 3 };
enum { ORC_TRN_JOINT = 0, ORC_TRN_TENDON = 3 };
/* ORC_EFC_CONTACT: normal row of an elliptic-cone contact (condim 3); its two friction rows follow as ORC_EFC_CONTACT_T */
enum { ORC_EFC_EQUALITY = 0, ORC_EFC_LIMIT = 1, ORC_EFC_FRICTION = 2, ORC_EFC_CONTACT = 3, ORC_EFC_CONTACT_T = 4 };
/* body / geom id of the free box in contact records (the box is kept outside the robot's body and geom tables) */
#define ORC_BODY_BOX (-2)
#define ORC_MAXSELF 256
#define ORC_MAXSEP 8

/* ---- one free rigid box on the floor plane (rcs_object.c) */
typedef struct orc_box {
  int present;
  double qpos0[7];         /* x y z, qw qx qy qz (free joint) */
  double mass, inertia[3]; /* centre of mass at the body origin, principal axes = body axes */
  double size[3];          /* half extents of the box geom */
  double friction[3];      /* mixed with the plane geom: element-wise maximum (equal priority) */
  double geom_friction[3]; /* the box geom's own coefficients (mixed with a robot geom's per contact) */
  double solref[2], solimp[5];
  double plane_z;          /* floor plane z = plane_z, normal +z */
  double impratio, noslip_tolerance;
  int noslip_iterations, nv_total; /* nv of the whole scene (robot + 6) */
  double meaninertia;              /* mjModel.stat.meaninertia of the whole scene; derived by orc_set0 */
} orc_box;

typedef struct orc_box_data {
  double qpos[7], qvel[6], qacc[6], qacc_warmstart[6];
  double qfrc_smooth[6], qacc_smooth[6];
  double xpos[3], xquat[4]; /* pose the last position stage saw (mjData.xpos / xquat of the body) */
  int ncon, zone[4]; /* zone: 0 top (separating), 1 middle (sliding), 2 bottom (sticking) */
  double con_dist[4], con_pos[4][3], con_mu[4];
  double J[12][6], aref[12], D[12], R[12], force[12];
  int newton_iter, noslip_iter;
} orc_box_data;

void orc_box_reset(const orc_box* b, orc_box_data* d);
void orc_box_step1(const orc_box* b, orc_box_data* d, double timestep);
void orc_box_step2(const orc_box* b, orc_box_data* d, const double* gravity, double h, double improvement0);
void orc_box_smooth(const orc_box* b, orc_box_data* d, const double* gravity);
void orc_box_integrate(const orc_box* b, orc_box_data* d, double h);

/* ---- model constants (mjModel subset; filled by the Python side from the compiled scene) */
typedef struct orc_model {
  int nbody, njnt, nu, ntendon, nwrap, neq, nsite;
  double timestep;
  double gravity[3];
  /* bodies */
  int body_parentid[ORC_MAXBODY];
  int body_rootid[ORC_MAXBODY];
  int body_jntadr[ORC_MAXBODY]; /* -1: welded to parent */
  double body_pos[ORC_MAXBODY][3];
  double body_quat[ORC_MAXBODY][4]; /* wxyz */
  double body_ipos[ORC_MAXBODY][3];
  double body_iquat[ORC_MAXBODY][4];
  double body_mass[ORC_MAXBODY];
  double body_inertia[ORC_MAXBODY][3];
  double body_gravcomp[ORC_MAXBODY];
  /* joints == dofs (hinge / slide only) */
  int jnt_type[ORC_MAXV];
  int jnt_bodyid[ORC_MAXV];
  double jnt_pos[ORC_MAXV][3];
  double jnt_axis[ORC_MAXV][3];
  int jnt_limited[ORC_MAXV];
  double jnt_range[ORC_MAXV][2];
  double jnt_margin[ORC_MAXV];
  double jnt_solref[ORC_MAXV][2];
  double jnt_solimp[ORC_MAXV][5];
  int jnt_actfrclimited[ORC_MAXV];
  double jnt_actfrcrange[ORC_MAXV][2];
  int jnt_actgravcomp[ORC_MAXV];
  double dof_armature[ORC_MAXV];
  double dof_damping[ORC_MAXV];
  double qpos0[ORC_MAXV];
  /* fixed tendons */
  int tendon_adr[ORC_MAXTENDON];
  int tendon_num[ORC_MAXTENDON];
  int wrap_objid[ORC_MAXWRAP];
  double wrap_prm[ORC_MAXWRAP];
  /* joint equalities */
  int eq_obj1id[ORC_MAXEQ];
  int eq_obj2id[ORC_MAXEQ];
  int eq_active0[ORC_MAXEQ];
  double eq_data[ORC_MAXEQ][5];
  double eq_solref[ORC_MAXEQ][2];
  double eq_solimp[ORC_MAXEQ][5];
  /* actuators */
  int actuator_trntype[ORC_MAXU];
  int actuator_trnid[ORC_MAXU];
  double actuator_gear[ORC_MAXU];
  double actuator_gainprm[ORC_MAXU][3];
  double actuator_biasprm[ORC_MAXU][3];
  int actuator_biastype[ORC_MAXU];
  int actuator_ctrllimited[ORC_MAXU];
  double actuator_ctrlrange[ORC_MAXU][2];
  int actuator_forcelimited[ORC_MAXU];
  double actuator_forcerange[ORC_MAXU][2];
  /* sites */
  int site_bodyid[ORC_MAXSITE];
  double site_pos[ORC_MAXSITE][3];
  double site_quat[ORC_MAXSITE][4];
  /* collision geoms (plane vs convex contact detection only) */
  int ngeom;
  int geom_type[ORC_MAXGEOM];       /* mjtGeom: 0 plane, 2 sphere, 3 capsule, 6 box, 7 mesh */
  int geom_bodyid[ORC_MAXGEOM];
  int geom_contype[ORC_MAXGEOM], geom_conaffinity[ORC_MAXGEOM];
  int geom_vertadr[ORC_MAXGEOM], geom_vertnum[ORC_MAXGEOM];
  double geom_pos[ORC_MAXGEOM][3], geom_quat[ORC_MAXGEOM][4], geom_size[ORC_MAXGEOM][3];
  double geom_friction[ORC_MAXGEOM][3];
  const double* mesh_vert;          /* [nvert][3] hull vertices, geom frame (owned by the caller) */
  int body_weldid[ORC_MAXBODY];
  int resolve_contacts;             /* 1: contacts of robot geoms (with the floor, with the free box) enter the constraint solve */
  /* derived by orc_set0 */
  double dof_invweight0[ORC_MAXV];
  double body_invweight0[ORC_MAXBODY]; /* translational component (mjModel.body_invweight0[.][0]) */
  double geom_aabb[ORC_MAXGEOM][6];    /* bounding box in the geom frame: centre, half extents (mjModel.geom_aabb) */
  double geom_rbound[ORC_MAXGEOM];     /* radius of the bounding sphere about the geom's origin (mjModel.geom_rbound) */
  double geom_center[ORC_MAXGEOM][3];  /* mesh geoms: mean of the hull's vertices, geom frame (the interior point MPR starts from) */
  /* dry joint friction (mjModel dof_frictionloss, dof_solref, dof_solimp) */
  double dof_frictionloss[ORC_MAXV];
  double dof_solref[ORC_MAXV][2];
  double dof_solimp[ORC_MAXV][5];
  orc_box box;
} orc_model;

/* one contact of the last position stage (mjContact subset); frame rows: normal (geom[0] -> geom[1]), two tangents */
typedef struct orc_contact {
  int geom[2], body[2]; /* robot geom / body ids; the free box: geom id = model ngeom, body ORC_BODY_BOX */
  double pos[3], frame[9], dist, mu;
  int efc_address, zone; /* first of its three rows in the coupled problem (-1: decoupled step); cone zone at the solution */
} orc_contact;

/* ---- per-environment state + scratch (mjData subset) */
typedef struct orc_data {
  double time;
  double qpos[ORC_MAXV], qvel[ORC_MAXV], ctrl[ORC_MAXU];
  double qacc[ORC_MAXV], qacc_warmstart[ORC_MAXV];
  /* position stage */
  double xpos[ORC_MAXBODY][3], xquat[ORC_MAXBODY][4], xmat[ORC_MAXBODY][9];
  double xipos[ORC_MAXBODY][3], ximat[ORC_MAXBODY][9];
  double xanchor[ORC_MAXV][3], xaxis[ORC_MAXV][3];
  double site_xpos[ORC_MAXSITE][3], site_xmat[ORC_MAXSITE][9];
  double subtree_com[ORC_MAXBODY][3];
  double cinert[ORC_MAXBODY][10];
  double cdof[ORC_MAXV][6];
  double qM[ORC_MAXV][ORC_MAXV];
  double ten_length[ORC_MAXTENDON];
  double actuator_length[ORC_MAXU];
  /* velocity stage */
  double cvel[ORC_MAXBODY][6], cdof_dot[ORC_MAXV][6];
  double actuator_velocity[ORC_MAXU];
  double qfrc_bias[ORC_MAXV], qfrc_passive[ORC_MAXV], qfrc_gravcomp[ORC_MAXV];
  /* actuation / acceleration */
  double actuator_force[ORC_MAXU];
  double qfrc_actuator[ORC_MAXV], qfrc_smooth[ORC_MAXV], qacc_smooth[ORC_MAXV];
  /* constraints */
  int nefc, ncon;
  int efc_type[ORC_MAXEFC];
  double efc_J[ORC_MAXEFC][ORC_NVT];
  double efc_pos[ORC_MAXEFC], efc_margin[ORC_MAXEFC], efc_vel[ORC_MAXEFC];
  double efc_D[ORC_MAXEFC], efc_aref[ORC_MAXEFC], efc_force[ORC_MAXEFC];
  double efc_K[ORC_MAXEFC], efc_B[ORC_MAXEFC], efc_I[ORC_MAXEFC];
  double efc_frictionloss[ORC_MAXEFC];
  double efc_R[ORC_MAXEFC], efc_mu[ORC_MAXEFC]; /* contact rows: regulariser; efc_mu on the normal row: the cone's regularised mu */
  double qfrc_constraint[ORC_NVT];
  int solver_niter, noslip_niter;
  int contact_geom[ORC_MAXCON][2];  /* d->contact[i].geom, i < ncon */
  orc_contact contact[ORC_MAXCON];
  int coupled; /* the last step solved robot and box in one problem (a contact involved a robot geom) */
  /* contacts between two geoms of the robot (self collision): detected for the collision callbacks, which scan them after
     contact[]; they carry no constraint rows in this revision */
  int nself;
  int self_geom[ORC_MAXSELF][2];
  /* warm start of the self-collision narrow phase (as a collision library keeps per-pair caches): the direction that
     separated a pair last, in the frame of its first geom; one support query along it settles the pair while it still does */
  int sep_n;
  int sep_pair[ORC_MAXSEP][2];
  double sep_dir[ORC_MAXSEP][3];
  double self_depth[ORC_MAXSELF]; /* penetration depth of self contact i (MPR) */
  double pen_seen; /* test support: the deepest penetration of a robot geom (floor / box / self) any collision pass has seen since the caller zeroed it */
  orc_box_data box;
} orc_data;

void orc_set0(orc_model* m);
/* rcs_contact.c: contacts of the robot's geoms and the coupled constraint problem */
void orc_collide(const orc_model* m, orc_data* d);
/* test support: keep only the first `cap` contacts of MuJoCo's order where contacts are resolved (0: all, the default) */
void orc_set_contact_cap(int cap);
void orc_make_coupled_rows(const orc_model* m, orc_data* d);
void orc_solve_coupled(const orc_model* m, orc_data* d);
double orc_impedance(const double* solimp, double pos, double margin);
/* narrow-phase pieces exposed for known-answer tests: each returns the number of contacts written */
int orc_box_box(const double* p1, const double* R1, const double* s1, const double* p2, const double* R2, const double* s2,
                double* pos /* [8][3] */, double* normal /* [8][3], box 1 -> box 2 */, double* dist /* [8] */);
int orc_mpr_hull_box(const double* verts, int nvert, const double* ph, const double* Rh, const double* pb, const double* Rb,
                     const double* sb, double* pos, double* normal /* hull -> box */, double* dist);
void orc_reset_data(const orc_model* m, orc_data* d);
void orc_step1(const orc_model* m, orc_data* d);
void orc_step2(const orc_model* m, orc_data* d);
/* position-stage pieces exposed for known-answer tests */
void orc_kinematics(const orc_model* m, orc_data* d);
void orc_mass_matrix(const orc_model* m, orc_data* d);

/* ---- Pose (reference: include/rcs/Pose.h, src/rcs/Pose.cpp); quaternion order xyzw */
typedef struct orc_pose {
  double t[3];
  double q[4];
} orc_pose;

void orc_pose_identity(orc_pose* p);
void orc_pose_from_matrix4(const double* m16_rowmajor, orc_pose* out);
void orc_pose_from_rotm_t(const double* r9_rowmajor, const double* t3, orc_pose* out);
void orc_pose_from_quat_t(const double* q4_xyzw, const double* t3, orc_pose* out);
void orc_pose_from_rpy_t(const double* rpy3, const double* t3, orc_pose* out);
void orc_pose_rotation_m(const orc_pose* p, double* r9_rowmajor);
void orc_pose_matrix(const orc_pose* p, double* m16_rowmajor);
void orc_pose_rpy(const orc_pose* p, double* rpy3);
void orc_pose_xyzrpy(const orc_pose* p, double* out6);
void orc_pose_mul(const orc_pose* a, const orc_pose* b, orc_pose* out);
void orc_pose_inverse(const orc_pose* a, orc_pose* out);
double orc_pose_total_angle(const orc_pose* a);
void orc_pose_limit_rotation_angle(const orc_pose* a, double max_angle, orc_pose* out);
void orc_pose_limit_translation_length(const orc_pose* a, double max_length, orc_pose* out);
void orc_pose_interpolate(const orc_pose* a, const orc_pose* dest, double progress, orc_pose* out);
int orc_pose_is_close(const orc_pose* a, const orc_pose* b, double eps_r, double eps_t);
void orc_franka_hand_tcp_offset(orc_pose* out);

/* ---- Pin CLIK (reference: src/rcs/Kinematics.cpp:28-82) on the compiled chain */
typedef struct orc_ik {
  const orc_model* m;
  int site; /* frame = this site */
} orc_ik;
int orc_ik_inverse(const orc_ik* ik, const orc_pose* pose, const double* q0, int nq0, const orc_pose* tcp_offset,
                   double* q_out /* m->njnt */, int* iterations);
void orc_ik_forward(const orc_ik* ik, const double* q0, int nq0, const orc_pose* tcp_offset, orc_pose* out);

/* ---- Sim + SimRobot + SimGripper (reference: src/sim/sim.cpp, SimRobot.cpp, SimGripper.cpp) */
typedef struct orc_sim {
  orc_model* m;
  orc_data d;
  /* SimConfig (sim.h:29-34) */
  int async_control, realtime, frequency, max_convergence_steps;
  long convergence_steps;
  int converged;
  /* callback bookkeeping, in registration order of SimEnvCreator (creators.py:88,105):
     plain: [robot.is_arrived, robot.is_moving]
     any:   [robot.collision, gripper.collision]
     all:   [robot.convergence, gripper.convergence] */
  int has_robot, robot_conv_registered, has_gripper;
  double cb_last[2];
  double any_last[2], all_last[2];
  int any_ret[2], all_ret[2];
  /* SimRobot (SimRobot.h) */
  int arm_n;
  int arm_jnt[ORC_MAXARM], arm_act[ORC_MAXARM];
  int attachment_site, base_body;
  double joint_rotational_tolerance, robot_period;
  orc_pose tcp_offset;
  double q_home[ORC_MAXARM];
  double previous_angles[ORC_MAXARM], target_angles[ORC_MAXARM];
  int ik_success, robot_collision, is_moving, is_arrived;
  orc_ik ik;
  int last_ik_iterations;
  int arm_ncgeom, arm_cgeom[ORC_MAXCGEOM];
  /* SimGripper (SimGripper.h) */
  int grp_jnt, grp_act;
  double grp_period, max_actuator_width, min_actuator_width, max_joint_width, min_joint_width;
  double epsilon_inner, epsilon_outer;
  double last_commanded_width, last_width;
  int grp_is_moving, grp_collision;
  int grp_ncgeom, grp_cgeom[ORC_MAXCGEOM], grp_ncfgeom, grp_cfgeom[ORC_MAXCGEOM], grp_nignored, grp_ignored[ORC_MAXCGEOM];
} orc_sim;

void orc_sim_init(orc_sim* s, orc_model* m);
void orc_sim_add_robot(orc_sim* s, int n, const int* jnt_ids, const int* act_ids, int site, int base_body,
                       const double* q_home, const orc_pose* tcp_offset, int register_convergence_callback);
void orc_sim_add_gripper(orc_sim* s, int jnt, int act);
void orc_sim_set_robot_cgeoms(orc_sim* s, int n, const int* ids);
void orc_sim_set_gripper_cgeoms(orc_sim* s, int n, const int* cgeom, int nf, const int* cfgeom, int ni, const int* ignored);
void orc_sim_step(orc_sim* s, long k);
void orc_sim_step_until_convergence(orc_sim* s);
void orc_sim_reset(orc_sim* s);
/* SimRobot */
void orc_robot_set_joint_position(orc_sim* s, const double* q);
void orc_robot_get_joint_position(const orc_sim* s, double* q);
void orc_robot_get_cartesian_position(const orc_sim* s, orc_pose* out);
void orc_robot_get_base_pose(const orc_sim* s, orc_pose* out);
void orc_robot_set_cartesian_position(orc_sim* s, const orc_pose* pose);
void orc_robot_set_joints_hard(orc_sim* s, const double* q);
void orc_robot_reset(orc_sim* s);
void orc_robot_move_home(orc_sim* s);
/* SimGripper */
int orc_gripper_set_normalized_width(orc_sim* s, double width, double force);
double orc_gripper_get_normalized_width(const orc_sim* s);
int orc_gripper_is_grasped(const orc_sim* s);
void orc_gripper_reset(orc_sim* s);

unsigned long orc_sizeof_model(void);
unsigned long orc_sizeof_sim(void);

#ifdef __cplusplus
}
#endif
#endif
