// rcs_hip._core -- pybind11 binding of the MI355X batched simulation backend (librcs_hip.so, include/rcs_hip.h).
//
// The reference attaches hardware back-ends as separate pybind modules next to rcs._core
// (reference extensions/rcs_fr3/src/pybind/rcs.cpp:45-60); this is that module for the batched simulator.  Class and
// method names, argument names and the GIL policy follow the reference's `rcs._core.sim` (reference
// src/pybind/rcs.cpp:421-527, 351-396; signatures frozen in python/rcs/_core/sim.pyi) -- with ONE difference that runs
// through everything: a handle owns N environments, so every array carries a leading environment axis
// (q[N][dof], pose[N][7] = x y z qx qy qz qw, flags bool[N]) and setters take an optional `mask` (bool[N]).
//
//   reference                                   here
//   Sim(mjmdl: int, mjdata: int)                Sim(model: dict, n_envs: int, device: int = 0, free_box: dict | None = None)
//       raw mjModel* / mjData* cannot cross without MuJoCo: `model` holds the mjModel-named tables
//       (rcs_amd.mjcf.Model.arrays + scalars + name lists) and the state lives in HBM
//   SimRobot(sim, ik, cfg, register_convergence_callback=True)     the same; `ik` is accepted and ignored (the CLIK runs on the device)
//   SimGripper(sim, cfg)                                           the same
//
// The module converts numpy arrays to pointers and error codes to the reference's exception types; nothing else.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <algorithm>
#include <cmath>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "rcs_hip.h"

namespace py = pybind11;
using darr = py::array_t<double, py::array::c_style | py::array::forcecast>;
using farr = py::array_t<float, py::array::c_style | py::array::forcecast>;
using iarr = py::array_t<int32_t, py::array::c_style | py::array::forcecast>;
using barr = py::array_t<uint8_t, py::array::c_style | py::array::forcecast>;
using mask_t = std::optional<py::array_t<bool, py::array::c_style | py::array::forcecast>>;

namespace {

// error codes -> the exception types the reference raises (SimRobot.cpp:57-93: runtime_error; SimGripper.cpp:80-83: invalid_argument)
void check(int rc) {
  if (rc == RCSH_OK) return;
  const std::string msg = rcsh_last_error();
  if (rc == RCSH_ERR_ARG) throw std::invalid_argument(msg);
  throw std::runtime_error(msg);
}

struct Mask {
  std::vector<uint8_t> v;
  const uint8_t* p = nullptr;
  Mask(const mask_t& m, int n) {
    if (!m) return;
    if (m->size() != n) throw std::invalid_argument("mask must have one entry per environment");
    v.assign(m->data(), m->data() + n);
    p = v.data();
  }
};

struct SimConfig {  // reference src/sim/sim.h:29-34
  bool async_control = false, realtime = false;
  int frequency = 30, max_convergence_steps = 500;
};

struct SimCameraSet;
struct CamPose { int link; double pos[3], rot[9], fovy; };
struct Sim {
  rcsh_sim* h = nullptr;
  int n = 0;
  // cameras (set_render_scene): poses of the MJCF cameras by name, plus "" = the default free camera and "<free>" = an
  // untouched mjvCamera; the camera sets that render from inside the stepping (render_on_demand = false)
  std::map<std::string, CamPose> cam_poses;
  double znear = 0, zfar = 0, timestep = 0.002;
  std::vector<SimCameraSet*> rate_sets;
  void collect_frames();
  std::map<std::string, std::vector<std::string>> names;  // mj_name2id tables: "joint", "actuator", "body", "site", "geom"
  std::vector<py::array> keep;

  Sim(const py::dict& model, int n_envs, int device, const py::object& free_box) : n(n_envs) {
    rcsh_model_desc d{};
    auto geti = [&](const char* k) { return model.contains(k) ? model[k].cast<int>() : 0; };
    d.nbody = geti("nbody"); d.njnt = geti("njnt"); d.nu = geti("nu"); d.ntendon = geti("ntendon"); d.nwrap = geti("nwrap");
    d.neq = geti("neq"); d.nsite = geti("nsite"); d.ngeom = geti("ngeom"); d.nmeshvert = geti("nmeshvert");
    d.timestep = model["timestep"].cast<double>();
    timestep = d.timestep;
    {
      darr g = model["gravity"].cast<darr>();
      for (int k = 0; k < 3; ++k) d.gravity[k] = g.data()[k];
    }
    auto f64 = [&](const char* k) -> const double* {
      if (!model.contains(k)) throw std::invalid_argument(std::string("model table missing: ") + k);
      darr a = model[k].cast<darr>();
      keep.push_back(a);
      return a.size() ? a.data() : nullptr;
    };
    auto i32 = [&](const char* k) -> const int32_t* {
      if (!model.contains(k)) throw std::invalid_argument(std::string("model table missing: ") + k);
      iarr a = model[k].cast<iarr>();
      keep.push_back(a);
      return a.size() ? a.data() : nullptr;
    };
#define F(name) d.name = f64(#name)
#define I(name) d.name = i32(#name)
    I(body_parentid); I(body_jntadr); I(body_jntnum); F(body_pos); F(body_quat); F(body_ipos); F(body_iquat); F(body_mass);
    F(body_inertia); F(body_gravcomp); I(jnt_type); I(jnt_bodyid); F(jnt_pos); F(jnt_axis); I(jnt_limited); F(jnt_range);
    F(jnt_margin); F(jnt_solref); F(jnt_solimp); I(jnt_actfrclimited); F(jnt_actfrcrange); I(jnt_actgravcomp); F(dof_armature);
    F(dof_damping); F(dof_frictionloss); F(qpos0); I(tendon_adr); I(tendon_num); I(wrap_objid); F(wrap_prm); I(eq_obj1id);
    I(eq_obj2id); I(eq_active0); F(eq_data); F(eq_solref); F(eq_solimp); I(actuator_trntype); I(actuator_trnid); F(actuator_gear);
    F(actuator_gainprm); F(actuator_biasprm); I(actuator_biastype); I(actuator_ctrllimited); F(actuator_ctrlrange);
    I(actuator_forcelimited); F(actuator_forcerange); I(site_bodyid); F(site_pos); F(site_quat); I(geom_type); I(geom_bodyid);
    I(geom_contype); I(geom_conaffinity); F(geom_pos); F(geom_quat); F(geom_size); I(geom_vertadr); I(geom_vertnum); F(mesh_vert);
    F(dof_solref); F(dof_solimp); F(geom_friction);
#undef F
#undef I
    if (model.contains("names")) names = model["names"].cast<std::map<std::string, std::vector<std::string>>>();
    check(rcsh_sim_create(&d, n_envs, device, &h));
    if (!free_box.is_none()) {
      const py::dict fb = free_box.cast<py::dict>();
      rcsh_free_box_desc b{};
      auto vec = [&](const char* k, double* dst, int cnt) {
        darr a = fb[k].cast<darr>();
        if (a.size() != cnt) throw std::invalid_argument(std::string("free_box.") + k + ": wrong length");
        for (int i = 0; i < cnt; ++i) dst[i] = a.data()[i];
      };
      vec("qpos0", b.qpos0, 7); vec("inertia", b.inertia, 3); vec("size", b.size, 3); vec("friction", b.friction, 3);
      vec("solref", b.solref, 2); vec("solimp", b.solimp, 5); vec("geom_friction", b.geom_friction, 3); vec("floor_friction", b.floor_friction, 3);
      b.mass = fb["mass"].cast<double>(); b.plane_z = fb["plane_z"].cast<double>(); b.impratio = fb["impratio"].cast<double>();
      b.noslip_tolerance = fb["noslip_tolerance"].cast<double>(); b.noslip_iterations = fb["noslip_iterations"].cast<int>();
      b.cone_elliptic = fb["cone_elliptic"].cast<int>();
      b.resolve_robot_contacts = fb.contains("resolve_robot_contacts") ? fb["resolve_robot_contacts"].cast<int>() : 1;
      check(rcsh_sim_add_free_box(h, &b));
    }
  }
  ~Sim() { rcsh_sim_destroy(h); }
  Sim(const Sim&) = delete;

  // mj_name2id; the reference's "No joint named ..." runtime_error (SimRobot.cpp:57-93, SimGripper.cpp:16-28)
  int id(const std::string& kind, const std::string& name) const {
    auto it = names.find(kind);
    if (it != names.end())
      for (size_t i = 0; i < it->second.size(); ++i)
        if (it->second[i] == name) return (int)i;
    throw std::runtime_error("No " + kind + " named " + name);
  }
};

struct SimRobotConfig {  // reference src/sim/SimRobot.h:14-47 (+ common::RobotConfig: robot_type, tcp_offset)
  std::string robot_type = "FR3";
  darr tcp_offset;  // pose [7], xyzw quaternion; identity by default
  double joint_rotational_tolerance = .05 * (3.14159265358979323846 / 180.0), seconds_between_callbacks = 0.1;
  bool trajectory_trace = false;
  std::vector<std::string> arm_collision_geoms{"fr3_link0_collision", "fr3_link1_collision", "fr3_link2_collision", "fr3_link3_collision",
                                               "fr3_link4_collision", "fr3_link5_collision", "fr3_link6_collision", "fr3_link7_collision"};
  std::vector<std::string> joints{"fr3_joint1", "fr3_joint2", "fr3_joint3", "fr3_joint4", "fr3_joint5", "fr3_joint6", "fr3_joint7"};
  std::vector<std::string> actuators = joints;
  std::string attachment_site = "attachment_site", base = "base", mjcf_scene_path, kinematic_model_path;
  darr q_home;  // robots_meta_config(robot_type).q_home (include/rcs/Robot.h:24-95)
  void add_id(const std::string& id) {  // SimRobot.h:36-46
    for (auto* v : {&arm_collision_geoms, &joints, &actuators})
      for (auto& s : *v) s = s + "_" + id;
    attachment_site += "_" + id;
    base += "_" + id;
  }
};
struct SimRobotState {  // SimRobot.h:49-57, one entry per environment
  darr previous_angles, target_angles, inverse_tcp_offset;
  py::array_t<bool> ik_success, collision, is_moving, is_arrived;
};

// Pose algebra on [x y z qx qy qz qw] rows (reference src/rcs/Pose.cpp: operator*, inverse)
void quat_rot(const double* q, const double* v, double* o) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * (y * v[2] - z * v[1]), ty = 2 * (z * v[0] - x * v[2]), tz = 2 * (x * v[1] - y * v[0]);
  o[0] = v[0] + w * tx + (y * tz - z * ty);
  o[1] = v[1] + w * ty + (z * tx - x * tz);
  o[2] = v[2] + w * tz + (x * ty - y * tx);
}
void pose_mul(const double* a, const double* b, double* o) {
  double r[3];
  quat_rot(a + 3, b, r);
  const double ax = a[3], ay = a[4], az = a[5], aw = a[6], bx = b[3], by = b[4], bz = b[5], bw = b[6];
  o[0] = a[0] + r[0]; o[1] = a[1] + r[1]; o[2] = a[2] + r[2];
  o[3] = aw * bx + ax * bw + ay * bz - az * by;
  o[4] = aw * by - ax * bz + ay * bw + az * bx;
  o[5] = aw * bz + ax * by - ay * bx + az * bw;
  o[6] = aw * bw - ax * bx - ay * by - az * bz;
}
void pose_inv(const double* a, double* o) {
  const double qi[4] = {-a[3], -a[4], -a[5], a[6]};
  double r[3];
  quat_rot(qi, a, r);
  o[0] = -r[0]; o[1] = -r[1]; o[2] = -r[2];
  o[3] = qi[0]; o[4] = qi[1]; o[5] = qi[2]; o[6] = qi[3];
}

py::array_t<bool> to_bool(const barr& a) {
  py::array_t<bool> o(a.size());
  for (py::ssize_t i = 0; i < a.size(); ++i) o.mutable_data()[i] = a.data()[i] != 0;
  return o;
}

struct SimRobot {
  std::shared_ptr<Sim> sim;
  SimRobotConfig cfg;
  int dof = 0;
  SimRobot(std::shared_ptr<Sim> s, const py::object& /*ik*/, const SimRobotConfig& c, bool register_convergence_callback) : sim(std::move(s)), cfg(c) {
    std::vector<int32_t> cg, jn, ac;
    for (auto& g : cfg.arm_collision_geoms) cg.push_back(sim->id("geom", g));
    const int site = sim->id("site", cfg.attachment_site), base = sim->id("body", cfg.base);
    for (auto& j : cfg.joints) jn.push_back(sim->id("joint", j));
    for (auto& a : cfg.actuators) ac.push_back(sim->id("actuator", a));
    dof = (int)jn.size();
    if (cfg.q_home.size() < dof) throw std::invalid_argument("SimRobotConfig.q_home: one entry per joint (robots_meta_config(robot_type).q_home)");
    rcsh_robot_desc d{};
    d.dof = dof; d.joint_ids = jn.data(); d.actuator_ids = ac.data(); d.attachment_site = site; d.base_body = base;
    d.q_home = cfg.q_home.data();
    const double ident[7] = {0, 0, 0, 0, 0, 0, 1};
    for (int k = 0; k < 7; ++k) d.tcp_offset[k] = cfg.tcp_offset.size() == 7 ? cfg.tcp_offset.data()[k] : ident[k];
    d.joint_rotational_tolerance = cfg.joint_rotational_tolerance;
    d.seconds_between_callbacks = cfg.seconds_between_callbacks;
    d.register_convergence_callback = register_convergence_callback;
    d.n_collision_geoms = (int)cg.size();
    d.collision_geom_ids = cg.data();
    check(rcsh_sim_add_robot(sim->h, &d));
  }
  darr q_in(const darr& q) const {
    if (q.ndim() == 1 && q.shape(0) >= dof) {  // one configuration for every environment
      darr o({sim->n, dof});
      for (int e = 0; e < sim->n; ++e)
        for (int i = 0; i < dof; ++i) o.mutable_at(e, i) = q.at(i);
      return o;
    }
    if (q.ndim() != 2 || q.shape(0) != sim->n || q.shape(1) < dof) throw std::invalid_argument("q must be [n_envs, dof]");
    if (q.shape(1) == dof) return q;
    darr o({sim->n, dof});
    for (int e = 0; e < sim->n; ++e)
      for (int i = 0; i < dof; ++i) o.mutable_at(e, i) = q.at(e, i);
    return o;
  }
};

struct SimGripperConfig {  // reference src/sim/SimGripper.h:15-45
  double epsilon_inner = 0.005, epsilon_outer = 0.005, seconds_between_callbacks = 0.05;
  double max_actuator_width = 255, min_actuator_width = 0, max_joint_width = 0.04, min_joint_width = 0.0;
  std::vector<std::string> ignored_collision_geoms{};
  std::vector<std::string> collision_geoms{"hand_c", "d435i_collision", "finger_0_left", "finger_0_right"};
  std::vector<std::string> collision_geoms_fingers{"finger_0_left", "finger_0_right"};
  std::string joint = "finger_joint1", actuator = "actuator8";
  void add_id(const std::string& id) {
    for (auto* v : {&ignored_collision_geoms, &collision_geoms, &collision_geoms_fingers})
      for (auto& s : *v) s = s + "_" + id;
    joint += "_" + id;
    actuator += "_" + id;
  }
};
struct SimGripperState {  // SimGripper.h:47-52
  darr last_commanded_width, last_width;
  py::array_t<bool> is_moving, collision;
};
struct SimGripper {
  std::shared_ptr<Sim> sim;
  SimGripperConfig cfg;
  SimGripper(std::shared_ptr<Sim> s, const SimGripperConfig& c) : sim(std::move(s)), cfg(c) {
    std::vector<int32_t> cg, cf, ig;
    for (auto& g : cfg.collision_geoms) cg.push_back(sim->id("geom", g));
    for (auto& g : cfg.collision_geoms_fingers) cf.push_back(sim->id("geom", g));
    for (auto& g : cfg.ignored_collision_geoms) ig.push_back(sim->id("geom", g));
    rcsh_gripper_desc d{};
    d.actuator_id = sim->id("actuator", cfg.actuator);
    d.joint_id = sim->id("joint", cfg.joint);
    d.epsilon_inner = cfg.epsilon_inner; d.epsilon_outer = cfg.epsilon_outer; d.seconds_between_callbacks = cfg.seconds_between_callbacks;
    d.max_actuator_width = cfg.max_actuator_width; d.min_actuator_width = cfg.min_actuator_width;
    d.max_joint_width = cfg.max_joint_width; d.min_joint_width = cfg.min_joint_width;
    d.n_collision_geoms = (int)cg.size(); d.n_finger_geoms = (int)cf.size(); d.n_ignored_geoms = (int)ig.size();
    int32_t zero = 0;
    d.collision_geom_ids = cg.empty() ? &zero : cg.data();
    d.finger_geom_ids = cf.empty() ? &zero : cf.data();
    d.ignored_geom_ids = ig.empty() ? &zero : ig.data();
    check(rcsh_sim_add_gripper(sim->h, &d));
  }
  void set_width(const darr& width, double force, const mask_t& mask) {
    darr w(sim->n);
    for (int e = 0; e < sim->n; ++e) w.mutable_data()[e] = width.ndim() == 0 || width.size() == 1 ? width.data()[0] : width.at(e);
    if (width.ndim() > 0 && width.size() != 1 && width.size() != sim->n) throw std::invalid_argument("width must be a scalar or [n_envs]");
    Mask m(mask, sim->n);
    check(rcsh_gripper_set_normalized_width(sim->h, w.data(), force, m.p));
  }
};

// ---- cameras: reference src/sim/camera.h / camera.cpp, bound at src/pybind/rcs.cpp:560-597
enum class CameraType : int { free = 0, tracking = 1, fixed = 2, default_free = 3 };
struct SimCameraConfig {  // camera.h:26-34 (common::BaseCameraConfig + type)
  std::string identifier;
  int frame_rate = 0, resolution_width = 256, resolution_height = 256;
  CameraType type = CameraType::fixed;
};
// FrameSet (camera.h:36-40) of the batch: per camera [N, 3 W H] uint8 / [N, W H] float32 as mjr_readPixels returns them (rows
// bottom-up), one timestamp per environment
struct FrameSet {
  std::map<std::string, py::array_t<uint8_t>> color_frames;
  std::map<std::string, py::array_t<float>> depth_frames;
  darr timestamp;
  std::map<std::string, barr> rendered;  // rate-driven sets: which environments' frames of that camera belong to this set
};
struct SimCameraSet {
  std::shared_ptr<Sim> sim;
  std::map<std::string, SimCameraConfig> cfg;
  std::map<std::string, int32_t> ids;
  bool on_demand;
  std::vector<FrameSet> buffer;
  std::vector<double> last_ts;
  bool have_last = false;
  size_t max_framesets = 64;
  SimCameraSet(std::shared_ptr<Sim> s, std::map<std::string, SimCameraConfig> cams, bool render_on_demand)
      : sim(std::move(s)), cfg(std::move(cams)), on_demand(render_on_demand) {
    if (sim->cam_poses.empty()) throw std::runtime_error("no render scene: call Sim.set_render_scene(rcs_hip.render_tables(model)) first");
    std::vector<int32_t> rated;
    std::vector<double> periods;
    for (auto& [name, c] : cfg) {
      std::string key = c.identifier;
      if (c.type == CameraType::default_free) key = "";
      else if (c.type == CameraType::free) key = "<free>";
      else if (c.type == CameraType::tracking) throw std::runtime_error("track body id is outside valid range");  // camera.cpp:44-46 never sets one
      auto it = sim->cam_poses.find(key);
      if (it == sim->cam_poses.end()) throw std::runtime_error("No camera named " + c.identifier);
      rcsh_camera_desc d{};
      d.link = it->second.link; d.width = c.resolution_width; d.height = c.resolution_height; d.fovy_deg = it->second.fovy;
      for (int k = 0; k < 3; ++k) d.pos[k] = it->second.pos[k];
      for (int k = 0; k < 9; ++k) d.rot[k] = it->second.rot[k];
      int32_t id = -1;
      check(rcsh_sim_add_camera(sim->h, &d, &id));
      ids[name] = id;
      if (c.frame_rate != 0) { rated.push_back(id); periods.push_back(1.0 / c.frame_rate); }
    }
    if (!on_demand && !rated.empty()) {
      SimConfig sc; int32_t a, r, f, k;
      check(rcsh_sim_get_config(sim->h, &a, &r, &f, &k));
      double cap = 2;
      for (double pd : periods) cap += std::ceil((k > 0 ? k : 2000) * sim->timestep / pd);
      check(rcsh_sim_set_render_schedule(sim->h, rated.data(), periods.data(), (int)rated.size(), (int)std::min(256.0, cap)));
      sim->rate_sets.push_back(this);
    }
  }
  ~SimCameraSet() {
    auto& v = sim->rate_sets;
    v.erase(std::remove(v.begin(), v.end(), this), v.end());
  }
  int buffer_size() const { return (int)buffer.size(); }
  void clear_buffer() { buffer.clear(); have_last = false; }
  void push(FrameSet&& fs) {
    buffer.push_back(std::move(fs));
    if (buffer.size() > max_framesets) buffer.erase(buffer.begin());
  }
  void render_all() {  // camera.cpp:86-140
    darr ts(sim->n);
    check(rcsh_sim_get_time(sim->h, ts.mutable_data()));
    bool same = have_last;
    for (int e = 0; e < sim->n && same; ++e) same = ts.data()[e] == last_ts[e];
    if (!same) {
      FrameSet fs;
      fs.timestamp = ts;
      push(std::move(fs));
      last_ts.assign(ts.data(), ts.data() + sim->n);
      have_last = true;
    }
    FrameSet& fs = buffer.back();
    for (auto& [name, c] : cfg) {
      const py::ssize_t px = (py::ssize_t)c.resolution_width * c.resolution_height;
      py::array_t<uint8_t> rgb({(py::ssize_t)sim->n, 3 * px});
      py::array_t<float> depth({(py::ssize_t)sim->n, px});
      check(rcsh_camera_render_rgb(sim->h, ids[name], rgb.mutable_data(), depth.mutable_data(), nullptr, nullptr));
      fs.color_frames[name] = rgb;
      fs.depth_frames[name] = depth;
    }
  }
  void collect() {  // the frames that became due inside the launch that just ran
    std::vector<int32_t> count(sim->n);
    check(rcsh_render_pending(sim->h, count.data()));
    int slots = 0;
    for (int c : count) slots = std::max(slots, c);
    for (int slot = 0; slot < slots; ++slot) {
      FrameSet fs;
      fs.timestamp = darr(sim->n);
      for (int e = 0; e < sim->n; ++e) fs.timestamp.mutable_data()[e] = std::nan("");
      for (auto& [name, c] : cfg) {
        if (c.frame_rate == 0) continue;
        const py::ssize_t px = (py::ssize_t)c.resolution_width * c.resolution_height;
        py::array_t<uint8_t> rgb({(py::ssize_t)sim->n, 3 * px});
        py::array_t<float> depth({(py::ssize_t)sim->n, px});
        darr ts(sim->n);
        barr due(sim->n);
        check(rcsh_camera_render_snapshot(sim->h, ids[name], slot, rgb.mutable_data(), depth.mutable_data(), nullptr, nullptr, ts.mutable_data(),
                                          due.mutable_data()));
        bool any = false;
        for (int e = 0; e < sim->n; ++e)
          if (due.data()[e]) { fs.timestamp.mutable_data()[e] = ts.data()[e]; any = true; }
        if (!any) continue;
        fs.color_frames[name] = rgb;
        fs.depth_frames[name] = depth;
        fs.rendered[name] = due;
      }
      if (!fs.color_frames.empty()) push(std::move(fs));
    }
  }
  std::optional<FrameSet> get_latest_frameset() {  // camera.cpp:64-73
    if (on_demand) render_all();
    if (buffer.empty()) return std::nullopt;
    return buffer.back();
  }
  std::optional<FrameSet> get_timestamp_frameset(const darr& ts) {  // camera.cpp:74-83, one timestamp per environment
    for (auto it = buffer.rbegin(); it != buffer.rend(); ++it) {
      bool eq = ts.size() == it->timestamp.size();
      for (py::ssize_t e = 0; e < ts.size() && eq; ++e) eq = ts.data()[e] == it->timestamp.data()[e];
      if (eq) return *it;
    }
    return std::nullopt;
  }
};
void Sim::collect_frames() {
  for (auto* cs : rate_sets) cs->collect();
}

}  // namespace

void bind_common(py::module_& m);  // common.cpp: rcs_hip._core.common (Pose, RPY, Kinematics, Pin, robots_meta_config, ...)

PYBIND11_MODULE(_core, m) {
  m.doc() = "MI355X batched simulation backend for RCS: the N-environment form of rcs._core.sim (librcs_hip.so)";
  m.attr("__version__") = "0.2";
  m.def("abi_version", &rcsh_abi_version);
  m.def("device_count", &rcsh_device_count);
  bind_common(m);
  auto sim = m.def_submodule("sim", "sim module");

  py::class_<SimConfig>(sim, "SimConfig")
      .def(py::init<>())
      .def_readwrite("async_control", &SimConfig::async_control)
      .def_readwrite("realtime", &SimConfig::realtime)
      .def_readwrite("frequency", &SimConfig::frequency)
      .def_readwrite("max_convergence_steps", &SimConfig::max_convergence_steps);

  py::class_<Sim, std::shared_ptr<Sim>>(sim, "Sim")
      .def(py::init<const py::dict&, int, int, const py::object&>(), py::arg("model"), py::arg("n_envs") = 1, py::arg("device") = 0,
           py::arg("free_box") = py::none())
      // GIL policy of the reference: released for step_until_convergence (rcs.cpp:498-499), held for step (rcs.cpp:503)
      .def("step_until_convergence", [](Sim& s) {
        { py::gil_scoped_release nogil; check(rcsh_sim_step_until_convergence(s.h)); }
        s.collect_frames(); })
      .def("is_converged", [](Sim& s) { barr c(s.n); check(rcsh_sim_is_converged(s.h, c.mutable_data(), nullptr)); return to_bool(c); })
      .def("convergence_steps", [](Sim& s) { barr c(s.n); iarr k(s.n); check(rcsh_sim_is_converged(s.h, c.mutable_data(), k.mutable_data())); return k; })
      .def("set_config", [](Sim& s, const SimConfig& c) { check(rcsh_sim_set_config(s.h, c.async_control, c.realtime, c.frequency, c.max_convergence_steps)); return true; },
           py::arg("cfg"))
      .def("get_config", [](Sim& s) { SimConfig c; int32_t a, r, f, k; check(rcsh_sim_get_config(s.h, &a, &r, &f, &k)); c.async_control = a; c.realtime = r; c.frequency = f; c.max_convergence_steps = k; return c; })
      .def("step", [](Sim& s, size_t k) { check(rcsh_sim_step(s.h, (int64_t)k)); s.collect_frames(); }, py::arg("k"))
      // cameras: the drawn shapes, their colours and the camera poses (rcs_hip.render_tables(model, scene_dir)); before SimCameraSet
      .def("set_render_scene", [](Sim& s, const py::dict& r) {
        rcsh_render_scene_desc d{};
        std::vector<py::array> hold;
        auto f64 = [&](const char* k) { darr a = r[k].cast<darr>(); hold.push_back(a); return a.data(); };
        auto i32 = [&](const char* k) { iarr a = r[k].cast<iarr>(); hold.push_back(a); return a.data(); };
        d.nshape = r["nshape"].cast<int>(); d.nplanes = r["nplanes"].cast<int>();
        d.shape = i32("shape"); d.link = i32("link"); d.pos = f64("pos"); d.rot = f64("rot"); d.size = f64("size");
        d.plane_adr = i32("plane_adr"); d.plane_num = i32("plane_num"); d.sphere = f64("sphere"); d.planes = f64("planes");
        d.znear = r["znear"].cast<double>(); d.zfar = r["zfar"].cast<double>();
        check(rcsh_sim_set_render_scene(s.h, &d));
        rcsh_render_colours c{};
        c.colour = f64("colour");
        auto v3 = [&](const char* k, double* dst) { darr a = r[k].cast<darr>(); for (int i = 0; i < 3; ++i) dst[i] = a.data()[i]; };
        v3("headlight_ambient", c.headlight_ambient); v3("headlight_diffuse", c.headlight_diffuse); v3("light_dir", c.light_dir);
        v3("light_diffuse", c.light_diffuse); v3("sky_rgb1", c.sky_rgb1); v3("sky_rgb2", c.sky_rgb2);
        check(rcsh_sim_set_render_colours(s.h, &c));
        s.znear = d.znear; s.zfar = d.zfar;
        s.cam_poses.clear();
        for (auto item : r["cameras"].cast<py::dict>()) {
          const py::tuple t = item.second.cast<py::tuple>();
          CamPose cp{};
          cp.link = t[0].cast<int>();
          darr pos = t[1].cast<darr>(), rot = t[2].cast<darr>();
          for (int i = 0; i < 3; ++i) cp.pos[i] = pos.data()[i];
          for (int i = 0; i < 9; ++i) cp.rot[i] = rot.data()[i];
          cp.fovy = t[3].cast<double>();
          s.cam_poses[item.first.cast<std::string>()] = cp;
        } }, py::arg("render"))
      .def_property_readonly("time", [](Sim& s) { darr t(s.n); check(rcsh_sim_get_time(s.h, t.mutable_data())); return t; })
      .def("reset", [](Sim& s, const mask_t& mask) { Mask mk(mask, s.n); check(rcsh_sim_reset(s.h, mk.p)); }, py::arg("mask") = py::none())
      .def("_start_gui_server", [](Sim&, const std::string&) { throw std::runtime_error("the batched backend has no GUI server"); }, py::arg("id"))
      .def("_stop_gui_server", [](Sim&) {})
      .def_readonly("n_envs", &Sim::n)
      .def_property_readonly("qpos", [](Sim& s) { darr q({s.n, rcsh_sim_nq(s.h)}); check(rcsh_sim_get_qpos(s.h, q.mutable_data())); return q; })
      .def_property_readonly("qvel", [](Sim& s) { darr q({s.n, rcsh_sim_nq(s.h)}); check(rcsh_sim_get_qvel(s.h, q.mutable_data())); return q; })
      .def("free_joint_qpos", [](Sim& s) { darr q({s.n, 7}); check(rcsh_sim_get_free_qpos(s.h, q.mutable_data())); return q; })
      .def("set_free_joint_qpos", [](Sim& s, const darr& q, const mask_t& mask) {
        if (q.size() != (py::ssize_t)s.n * 7) throw std::invalid_argument("qpos must be [n_envs, 7]");
        Mask mk(mask, s.n); check(rcsh_sim_set_free_qpos(s.h, q.data(), mk.p)); }, py::arg("qpos"), py::arg("mask") = py::none());

  py::class_<SimRobotConfig>(sim, "SimRobotConfig")
      .def(py::init<>())
      .def_readwrite("robot_type", &SimRobotConfig::robot_type)
      .def_readwrite("tcp_offset", &SimRobotConfig::tcp_offset)
      .def_readwrite("q_home", &SimRobotConfig::q_home)
      .def_readwrite("joint_rotational_tolerance", &SimRobotConfig::joint_rotational_tolerance)
      .def_readwrite("seconds_between_callbacks", &SimRobotConfig::seconds_between_callbacks)
      .def_readwrite("trajectory_trace", &SimRobotConfig::trajectory_trace)
      .def_readwrite("arm_collision_geoms", &SimRobotConfig::arm_collision_geoms)
      .def_readwrite("joints", &SimRobotConfig::joints)
      .def_readwrite("actuators", &SimRobotConfig::actuators)
      .def_readwrite("attachment_site", &SimRobotConfig::attachment_site)
      .def_readwrite("base", &SimRobotConfig::base)
      .def_readwrite("mjcf_scene_path", &SimRobotConfig::mjcf_scene_path)
      .def_readwrite("kinematic_model_path", &SimRobotConfig::kinematic_model_path)
      .def("add_id", &SimRobotConfig::add_id, py::arg("id"));
  py::class_<SimRobotState>(sim, "SimRobotState")
      .def(py::init<>())
      .def_readonly("previous_angles", &SimRobotState::previous_angles)
      .def_readonly("target_angles", &SimRobotState::target_angles)
      .def_readonly("inverse_tcp_offset", &SimRobotState::inverse_tcp_offset)
      .def_readonly("ik_success", &SimRobotState::ik_success)
      .def_readonly("collision", &SimRobotState::collision)
      .def_readonly("is_moving", &SimRobotState::is_moving)
      .def_readonly("is_arrived", &SimRobotState::is_arrived);

  py::class_<SimRobot, std::shared_ptr<SimRobot>>(sim, "SimRobot")
      .def(py::init<std::shared_ptr<Sim>, const py::object&, const SimRobotConfig&, bool>(), py::arg("sim"), py::arg("ik"), py::arg("cfg"),
           py::arg("register_convergence_callback") = true)
      .def("get_config", [](SimRobot& r) { return r.cfg; })
      .def("set_config", [](SimRobot& r, const SimRobotConfig& c) { r.cfg = c; return true; }, py::arg("cfg"))
      .def("get_state", [](SimRobot& r) {
        const int n = r.sim->n;
        barr ik(n), col(n), mov(n), arv(n);
        SimRobotState st;
        st.previous_angles = darr({n, r.dof}); st.target_angles = darr({n, r.dof});
        check(rcsh_robot_get_state(r.sim->h, ik.mutable_data(), col.mutable_data(), mov.mutable_data(), arv.mutable_data(),
                                   st.previous_angles.mutable_data(), st.target_angles.mutable_data()));
        st.ik_success = to_bool(ik); st.collision = to_bool(col); st.is_moving = to_bool(mov); st.is_arrived = to_bool(arv);
        const double ident[7] = {0, 0, 0, 0, 0, 0, 1};
        st.inverse_tcp_offset = darr(7);
        pose_inv(r.cfg.tcp_offset.size() == 7 ? r.cfg.tcp_offset.data() : ident, st.inverse_tcp_offset.mutable_data());
        return st; })
      .def("get_cartesian_position", [](SimRobot& r) { darr p({r.sim->n, 7}); check(rcsh_robot_get_cartesian_position(r.sim->h, p.mutable_data())); return p; })
      // GIL released where the reference releases it (rcs.cpp:358-367)
      .def("set_joint_position", [](SimRobot& r, const darr& q, const mask_t& mask) {
        const darr qq = r.q_in(q); Mask mk(mask, r.sim->n);
        py::gil_scoped_release nogil; check(rcsh_robot_set_joint_position(r.sim->h, qq.data(), mk.p)); }, py::arg("q"), py::arg("mask") = py::none())
      .def("get_joint_position", [](SimRobot& r) { darr q({r.sim->n, r.dof}); check(rcsh_robot_get_joint_position(r.sim->h, q.mutable_data())); return q; })
      .def("move_home", [](SimRobot& r, const mask_t& mask) { Mask mk(mask, r.sim->n); py::gil_scoped_release nogil; check(rcsh_robot_move_home(r.sim->h, mk.p)); },
           py::arg("mask") = py::none())
      .def("reset", [](SimRobot& r, const mask_t& mask) { Mask mk(mask, r.sim->n); check(rcsh_robot_reset(r.sim->h, mk.p)); }, py::arg("mask") = py::none())
      .def("close", [](SimRobot&) {})
      .def("set_cartesian_position", [](SimRobot& r, const darr& pose, const mask_t& mask) {
        darr p({r.sim->n, 7});
        if (pose.ndim() == 1 && pose.size() == 7) { for (int e = 0; e < r.sim->n; ++e) for (int k = 0; k < 7; ++k) p.mutable_at(e, k) = pose.at(k); }
        else if (pose.ndim() == 2 && pose.shape(0) == r.sim->n && pose.shape(1) == 7) p = pose;
        else throw std::invalid_argument("pose must be [7] or [n_envs, 7] (x y z qx qy qz qw)");
        Mask mk(mask, r.sim->n);
        py::gil_scoped_release nogil; check(rcsh_robot_set_cartesian_position(r.sim->h, p.data(), mk.p)); }, py::arg("pose"), py::arg("mask") = py::none())
      .def("get_ik", [](SimRobot&) { return py::none(); })
      .def("get_base_pose_in_world_coordinates", [](SimRobot& r) { darr p({r.sim->n, 7}); check(rcsh_robot_get_base_pose(r.sim->h, p.mutable_data())); return p; })
      // src/rcs/Robot.cpp:5-14: base^-1 * pose / base * pose, on [7] or [n, 7] rows
      .def("to_pose_in_robot_coordinates", [](SimRobot& r, const darr& pose_in_world_coordinates) {
        darr base({r.sim->n, 7}); check(rcsh_robot_get_base_pose(r.sim->h, base.mutable_data()));
        double inv[7]; pose_inv(base.data(), inv);
        const py::ssize_t rows = pose_in_world_coordinates.size() / 7;
        if (rows * 7 != pose_in_world_coordinates.size()) throw std::invalid_argument("pose rows of 7 numbers (x y z qx qy qz qw)");
        darr o(pose_in_world_coordinates.request().shape);
        for (py::ssize_t k = 0; k < rows; ++k) pose_mul(inv, pose_in_world_coordinates.data() + 7 * k, o.mutable_data() + 7 * k);
        return o; }, py::arg("pose_in_world_coordinates"))
      .def("to_pose_in_world_coordinates", [](SimRobot& r, const darr& pose_in_robot_coordinates) {
        darr base({r.sim->n, 7}); check(rcsh_robot_get_base_pose(r.sim->h, base.mutable_data()));
        const py::ssize_t rows = pose_in_robot_coordinates.size() / 7;
        if (rows * 7 != pose_in_robot_coordinates.size()) throw std::invalid_argument("pose rows of 7 numbers (x y z qx qy qz qw)");
        darr o(pose_in_robot_coordinates.request().shape);
        for (py::ssize_t k = 0; k < rows; ++k) pose_mul(base.data(), pose_in_robot_coordinates.data() + 7 * k, o.mutable_data() + 7 * k);
        return o; }, py::arg("pose_in_robot_coordinates"))
      .def("set_joints_hard", [](SimRobot& r, const darr& q, const mask_t& mask) { const darr qq = r.q_in(q); Mask mk(mask, r.sim->n); check(rcsh_robot_set_joints_hard(r.sim->h, qq.data(), mk.p)); },
           py::arg("q"), py::arg("mask") = py::none());

  py::class_<SimGripperConfig>(sim, "SimGripperConfig")
      .def(py::init<>())
      .def_readwrite("epsilon_inner", &SimGripperConfig::epsilon_inner)
      .def_readwrite("epsilon_outer", &SimGripperConfig::epsilon_outer)
      .def_readwrite("seconds_between_callbacks", &SimGripperConfig::seconds_between_callbacks)
      .def_readwrite("ignored_collision_geoms", &SimGripperConfig::ignored_collision_geoms)
      .def_readwrite("collision_geoms", &SimGripperConfig::collision_geoms)
      .def_readwrite("collision_geoms_fingers", &SimGripperConfig::collision_geoms_fingers)
      .def_readwrite("joint", &SimGripperConfig::joint)
      .def_readwrite("max_actuator_width", &SimGripperConfig::max_actuator_width)
      .def_readwrite("min_actuator_width", &SimGripperConfig::min_actuator_width)
      .def_readwrite("max_joint_width", &SimGripperConfig::max_joint_width)
      .def_readwrite("min_joint_width", &SimGripperConfig::min_joint_width)
      .def_readwrite("actuator", &SimGripperConfig::actuator)
      .def("add_id", &SimGripperConfig::add_id, py::arg("id"));
  py::class_<SimGripperState>(sim, "SimGripperState")
      .def(py::init<>())
      .def_readonly("last_commanded_width", &SimGripperState::last_commanded_width)
      .def_readonly("is_moving", &SimGripperState::is_moving)
      .def_readonly("last_width", &SimGripperState::last_width)
      .def_readonly("collision", &SimGripperState::collision);

  py::class_<SimGripper, std::shared_ptr<SimGripper>>(sim, "SimGripper")
      .def(py::init<std::shared_ptr<Sim>, const SimGripperConfig&>(), py::arg("sim"), py::arg("cfg"))
      .def("get_config", [](SimGripper& g) { return g.cfg; })
      .def("set_config", [](SimGripper& g, const SimGripperConfig& c) { g.cfg = c; return true; }, py::arg("cfg"))
      .def("get_state", [](SimGripper& g) {
        const int n = g.sim->n;
        SimGripperState st;
        barr mv(n), col(n);
        st.last_commanded_width = darr(n); st.last_width = darr(n);
        check(rcsh_gripper_get_state(g.sim->h, st.last_commanded_width.mutable_data(), mv.mutable_data(), st.last_width.mutable_data(), col.mutable_data()));
        st.is_moving = to_bool(mv); st.collision = to_bool(col);
        return st; })
      .def("set_normalized_width", &SimGripper::set_width, py::arg("width"), py::arg("force") = 0, py::arg("mask") = py::none())
      .def("get_normalized_width", [](SimGripper& g) { darr w(g.sim->n); check(rcsh_gripper_get_normalized_width(g.sim->h, w.mutable_data())); return w; })
      .def("is_grasped", [](SimGripper& g) { barr b(g.sim->n); check(rcsh_gripper_is_grasped(g.sim->h, b.mutable_data())); return to_bool(b); })
      // grasp / open / shut / close / reset release the GIL in the reference (rcs.cpp:386-396)
      .def("grasp", [](SimGripper& g, const mask_t& mask) { darr z(1); z.mutable_data()[0] = 0.0; g.set_width(z, 0, mask); }, py::arg("mask") = py::none())
      .def("open", [](SimGripper& g, const mask_t& mask) { darr o(1); o.mutable_data()[0] = 1.0; g.set_width(o, 0, mask); }, py::arg("mask") = py::none())
      .def("shut", [](SimGripper& g, const mask_t& mask) { darr z(1); z.mutable_data()[0] = 0.0; g.set_width(z, 0, mask); }, py::arg("mask") = py::none())
      .def("close", [](SimGripper&) {})
      .def("reset", [](SimGripper& g, const mask_t& mask) { Mask mk(mask, g.sim->n); py::gil_scoped_release nogil; check(rcsh_gripper_reset(g.sim->h, mk.p)); },
           py::arg("mask") = py::none());

  // ---- the fused Gymnasium loop (SimEnvCreator()(...).reset() / .step(action): reference python/rcs/envs/creators.py:43-128)
  sim.def("env_configure", [](Sim& s, int control_mode, int relative_to, double max_mov_0, double max_mov_1, bool binary_gripper, const darr& low, const darr& high) {
    rcsh_env_desc d{};
    d.control_mode = control_mode; d.relative_to = relative_to; d.max_mov[0] = max_mov_0; d.max_mov[1] = max_mov_1;
    d.binary_gripper = binary_gripper; d.joint_low = low.data(); d.joint_high = high.data();
    check(rcsh_env_configure(s.h, &d)); },
    py::arg("sim"), py::arg("control_mode"), py::arg("relative_to"), py::arg("max_mov_0"), py::arg("max_mov_1"), py::arg("binary_gripper"), py::arg("joint_low"), py::arg("joint_high"));
  sim.def("env_reset", [](Sim& s, const mask_t& mask) {
    const int ow = rcsh_env_obs_width(s.h);
    darr obs({s.n, ow}); barr info({s.n, 8}); darr gw(s.n);
    Mask mk(mask, s.n);
    { py::gil_scoped_release nogil; check(rcsh_env_reset(s.h, mk.p, obs.mutable_data(), info.mutable_data(), gw.mutable_data())); }
    return py::make_tuple(obs, info, gw); }, py::arg("sim"), py::arg("mask") = py::none());
  sim.def("env_step", [](Sim& s, const darr& action, const std::optional<farr>& gripper) {
    const int ow = rcsh_env_obs_width(s.h), aw = rcsh_env_action_width(s.h);
    if (action.size() != (py::ssize_t)s.n * aw) throw std::invalid_argument("action must be [n_envs, action_width]");
    if (gripper && gripper->size() != s.n) throw std::invalid_argument("gripper must be [n_envs]");
    darr obs({s.n, ow}); barr info({s.n, 8}); darr gw(s.n); iarr sub(s.n);
    { py::gil_scoped_release nogil;
      check(rcsh_env_step(s.h, action.data(), gripper ? gripper->data() : nullptr, obs.mutable_data(), info.mutable_data(), gw.mutable_data(), sub.mutable_data())); }
    return py::make_tuple(obs, info, gw, sub); }, py::arg("sim"), py::arg("action"), py::arg("gripper") = py::none());
  py::enum_<CameraType>(sim, "CameraType")  // rcs.cpp:560-566
      .value("free", CameraType::free).value("tracking", CameraType::tracking).value("fixed", CameraType::fixed)
      .value("default_free", CameraType::default_free).export_values();
  py::class_<SimCameraConfig>(sim, "SimCameraConfig")  // rcs.cpp:567-574
      .def(py::init([](const std::string& identifier, int frame_rate, int w, int h, CameraType type) {
             SimCameraConfig c; c.identifier = identifier; c.frame_rate = frame_rate; c.resolution_width = w; c.resolution_height = h; c.type = type; return c; }),
           py::arg("identifier"), py::arg("frame_rate"), py::arg("resolution_width"), py::arg("resolution_height"), py::arg("type") = CameraType::fixed)
      .def_readwrite("identifier", &SimCameraConfig::identifier)
      .def_readwrite("frame_rate", &SimCameraConfig::frame_rate)
      .def_readwrite("resolution_width", &SimCameraConfig::resolution_width)
      .def_readwrite("resolution_height", &SimCameraConfig::resolution_height)
      .def_readwrite("type", &SimCameraConfig::type);
  py::class_<FrameSet>(sim, "FrameSet")  // rcs.cpp:575-580
      .def(py::init<>())
      .def_readonly("color_frames", &FrameSet::color_frames)
      .def_readonly("depth_frames", &FrameSet::depth_frames)
      .def_readonly("timestamp", &FrameSet::timestamp)
      .def_readonly("rendered", &FrameSet::rendered);
  py::class_<SimCameraSet, std::shared_ptr<SimCameraSet>>(sim, "SimCameraSet")  // rcs.cpp:586-597
      .def(py::init<std::shared_ptr<Sim>, std::map<std::string, SimCameraConfig>, bool>(), py::arg("sim"), py::arg("cameras"),
           py::arg("render_on_demand") = true)
      .def("buffer_size", &SimCameraSet::buffer_size)
      .def("clear_buffer", &SimCameraSet::clear_buffer)
      .def("get_latest_frameset", &SimCameraSet::get_latest_frameset)
      .def("get_timestamp_frameset", &SimCameraSet::get_timestamp_frameset, py::arg("ts"))
      .def_property_readonly("_sim", [](SimCameraSet& c) { return c.sim; });
}
