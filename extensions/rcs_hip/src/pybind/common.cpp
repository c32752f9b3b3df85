// rcs_hip._core.common -- the compiled counterpart of the reference's `rcs._core.common` for the classes the simulation path
// touches (reference src/pybind/rcs.cpp:224-336; names frozen in python/rcs/_core/common.pyi): Pose, RPY, Kinematics, Pin,
// RobotType, RobotPlatform, RobotMetaConfig, robots_meta_config, RobotConfig, BaseCameraConfig, GraspType and the identity /
// FrankaHandTCPOffset helpers.
//
// Pose / RPY are single host values, as in the reference; their arithmetic is csrc/pose.h -- the very functions the kernels
// run on the device (RCSH_HD compiles them for the host here), which follow src/rcs/Pose.cpp operation for operation (xyzw
// quaternions, re-normalisation in every constructor, the yaw-in-[0, pi] Euler extraction).  Pin is `Kinematics` over the
// C-ABI's rcsh_ik_* entry points (the CLIK of src/rcs/Kinematics.cpp:28-82 as a HIP kernel) on a one-environment handle of
// its own; the model file is MJCF (the reference's simulation path passes urdf=False, creators.py:81-85).
#include <pybind11/numpy.h>
#include <pybind11/operators.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cmath>
#include <cstring>
#include <memory>
#include <optional>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "pose.h"
#include "rcs_hip.h"

namespace py = pybind11;
using darr = py::array_t<double, py::array::c_style | py::array::forcecast>;

// Fixed-shape arguments of the Pose / RPY constructors.  The reference's overloads are told apart by Eigen's fixed-size types
// (Vector3d, Vector4d, Matrix3d, Matrix4d: src/pybind/rcs.cpp:248-262), whose casters REFUSE an array of another shape, so that
// pybind moves on to the next overload; a plain array_t accepts anything and a size check inside the constructor body throws
// instead of falling through (positional `Pose(np.zeros(3))` then died in the pose_matrix overload).  Shaped<R, C> restores the
// Eigen behaviour: C == 1 takes a 1-D array of R numbers (or R x 1 / 1 x R, as Eigen's vector caster does), otherwise exactly R x C.
template <int R, int C>
struct Shaped {
  darr a;
  const double* data() const { return a.data(); }
};
namespace pybind11 { namespace detail {
template <int R, int C>
struct type_caster<Shaped<R, C>> {
  using ShapedT = Shaped<R, C>;
  PYBIND11_TYPE_CASTER(ShapedT, const_name("numpy.ndarray[numpy.float64[") + const_name<R>() + const_name(", ") + const_name<C>() + const_name("]]"));
  bool load(handle src, bool convert) {
    if (!convert && !darr::check_(src)) return false;
    darr arr = darr::ensure(src);
    if (!arr) { PyErr_Clear(); return false; }
    const auto nd = arr.ndim();
    bool ok;
    if (C == 1) ok = (nd == 1 && arr.shape(0) == R) || (nd == 2 && ((arr.shape(0) == R && arr.shape(1) == 1) || (arr.shape(0) == 1 && arr.shape(1) == R)));
    else ok = nd == 2 && arr.shape(0) == R && arr.shape(1) == C;
    if (!ok) return false;
    value.a = std::move(arr);
    return true;
  }
  static handle cast(const Shaped<R, C>& s, return_value_policy, handle) { return darr(s.a).release(); }
};
}}  // namespace pybind11::detail
using Vec3 = Shaped<3, 1>;
using Vec4 = Shaped<4, 1>;
using Mat3 = Shaped<3, 3>;
using Mat4 = Shaped<4, 4>;

namespace {

constexpr double kPi = 3.141592653589793238462643383279502884;

void check(int rc) {
  if (rc == RCSH_OK) return;
  const std::string msg = rcsh_last_error();
  if (rc == RCSH_ERR_ARG) throw std::invalid_argument(msg);
  throw std::runtime_error(msg);
}

const double* need(const darr& a, py::ssize_t n, const char* what) {
  if (a.size() != n) throw std::invalid_argument(std::string(what) + ": expected " + std::to_string(n) + " numbers");
  return a.data();
}
darr vec(const double* p, py::ssize_t n) {
  darr o(n);
  std::memcpy(o.mutable_data(), p, sizeof(double) * n);
  return o;
}
darr mat(const double* p, py::ssize_t r, py::ssize_t c) {
  darr o({r, c});
  std::memcpy(o.mutable_data(), p, sizeof(double) * r * c);
  return o;
}

// ---- RPY (reference include/rcs/Pose.h:23-65)
struct RPY {
  double roll = 0, pitch = 0, yaw = 0;
  void quaternion(double* q) const {
    const double rpy[3] = {roll, pitch, yaw};
    // qz(yaw) * qy(pitch) * qx(roll), not yet normalised (as_quaternion_vector returns the product as is)
    double sz, cz, sy, cy, sx, cx;
    sincos(0.5 * rpy[2], &sz, &cz);
    sincos(0.5 * rpy[1], &sy, &cy);
    sincos(0.5 * rpy[0], &sx, &cx);
    const double qz[4] = {0, 0, sz, cz}, qy[4] = {0, sy, 0, cy}, qx[4] = {sx, 0, 0, cx};
    double tmp[4];
    rcsh::quat_mul(qz, qy, tmp);
    rcsh::quat_mul(tmp, qx, q);
  }
};

RPY operator+(const RPY& a, const RPY& b) { return RPY{a.roll + b.roll, a.pitch + b.pitch, a.yaw + b.yaw}; }

// orthogonal polar factor of a 3 x 3 matrix (an affine transform's rotation(), Pose.cpp:33-38): Newton iteration
// X <- (X + X^-T) / 2, quadratically convergent; an exactly orthogonal input comes back unchanged
void polar_rotation(const double* m, double* r) {
  double x[9];
  std::memcpy(x, m, sizeof(x));
  for (int it = 0; it < 60; ++it) {
    const double c00 = x[4] * x[8] - x[5] * x[7], c01 = x[5] * x[6] - x[3] * x[8], c02 = x[3] * x[7] - x[4] * x[6];
    const double det = x[0] * c00 + x[1] * c01 + x[2] * c02;
    if (!(std::fabs(det) > 1e-300)) break;
    // inverse transpose = cofactor matrix / det
    const double cof[9] = {c00, c01, c02,
                           x[2] * x[7] - x[1] * x[8], x[0] * x[8] - x[2] * x[6], x[1] * x[6] - x[0] * x[7],
                           x[1] * x[5] - x[2] * x[4], x[2] * x[3] - x[0] * x[5], x[0] * x[4] - x[1] * x[3]};
    double delta = 0, y[9];
    for (int k = 0; k < 9; ++k) {
      y[k] = 0.5 * (x[k] + cof[k] / det);
      delta = std::fmax(delta, std::fabs(y[k] - x[k]));
    }
    std::memcpy(x, y, sizeof(x));
    if (delta < 1e-16) break;
  }
  // An improper input (det < 0) converges to a reflection Q.  The reference's rotation() then flips the direction of the
  // SMALLEST singular value (U diag(1, 1, -1) V^T): R = Q (I - 2 v v^T) with v the eigenvector of S = Q^T M for its smallest
  // eigenvalue, found by cyclic Jacobi rotations of the symmetric 3 x 3 S.
  const double detq = x[0] * (x[4] * x[8] - x[5] * x[7]) - x[1] * (x[3] * x[8] - x[5] * x[6]) + x[2] * (x[3] * x[7] - x[4] * x[6]);
  if (detq < 0) {
    double S[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) S[i][j] = x[0 + i] * m[0 + j] + x[3 + i] * m[3 + j] + x[6 + i] * m[6 + j];
    for (int i = 0; i < 3; ++i)
      for (int j = i + 1; j < 3; ++j) S[i][j] = S[j][i] = 0.5 * (S[i][j] + S[j][i]);
    for (int sweep = 0; sweep < 50; ++sweep) {
      const double off = std::fabs(S[0][1]) + std::fabs(S[0][2]) + std::fabs(S[1][2]);
      if (off < 1e-300) break;
      for (int pq = 0; pq < 3; ++pq) {
        const int a = pq == 2 ? 1 : 0, b = pq == 0 ? 1 : 2;
        if (std::fabs(S[a][b]) < 1e-300) continue;
        const double theta = (S[b][b] - S[a][a]) / (2 * S[a][b]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1)), c = 1 / std::sqrt(t * t + 1), sn = t * c;
        for (int k = 0; k < 3; ++k) { const double ska = S[k][a], skb = S[k][b]; S[k][a] = c * ska - sn * skb; S[k][b] = sn * ska + c * skb; }
        for (int k = 0; k < 3; ++k) { const double sak = S[a][k], sbk = S[b][k]; S[a][k] = c * sak - sn * sbk; S[b][k] = sn * sak + c * sbk; }
        for (int k = 0; k < 3; ++k) { const double vka = V[k][a], vkb = V[k][b]; V[k][a] = c * vka - sn * vkb; V[k][b] = sn * vka + c * vkb; }
      }
    }
    int lo = 0;
    for (int k = 1; k < 3; ++k) if (S[k][k] < S[lo][lo]) lo = k;
    const double v[3] = {V[0][lo], V[1][lo], V[2][lo]};
    double y[9];
    for (int i = 0; i < 3; ++i) {
      const double qv = x[3 * i] * v[0] + x[3 * i + 1] * v[1] + x[3 * i + 2] * v[2];
      for (int j = 0; j < 3; ++j) y[3 * i + j] = x[3 * i + j] - 2 * qv * v[j];
    }
    std::memcpy(x, y, sizeof(x));
  }
  std::memcpy(r, x, sizeof(x));
}

// ---- Pose (reference include/rcs/Pose.h:67-170, src/rcs/Pose.cpp)
struct Pose {
  rcsh::Pose p{{0, 0, 0}, {0, 0, 0, 1}};
  Pose() = default;
  explicit Pose(const rcsh::Pose& r) : p(r) {}
  static Pose from_matrix(const darr& m16) {  // Pose.cpp:33-38
    const double* m = need(m16, 16, "pose_matrix");
    const double R[9] = {m[0], m[1], m[2], m[4], m[5], m[6], m[8], m[9], m[10]}, t[3] = {m[3], m[7], m[11]};
    double Rp[9];
    polar_rotation(R, Rp);
    Pose o;
    rcsh::pose_from_mat(Rp, t, o.p);
    return o;
  }
  static Pose from_rotation(const darr& rot, const darr* tr) {  // Pose.cpp:40-45, 101-104
    Pose o;
    const double zero[3] = {0, 0, 0};
    rcsh::mat_to_quat(need(rot, 9, "rotation"), o.p.q);
    if (tr) {
      rcsh::quat_normalize(o.p.q);
      std::memcpy(o.p.t, need(*tr, 3, "translation"), sizeof(o.p.t));
    } else std::memcpy(o.p.t, zero, sizeof(zero));
    return o;
  }
  static Pose from_quaternion(const darr& q, const darr* tr) {  // Pose.cpp:47-52, 96-99
    Pose o;
    const double zero[3] = {0, 0, 0};
    rcsh::pose_from_quat(need(q, 4, "quaternion"), tr ? need(*tr, 3, "translation") : zero, o.p);
    return o;
  }
  static Pose from_rpy(const RPY& r, const darr* tr) {  // Pose.cpp:61-73
    Pose o;
    const double zero[3] = {0, 0, 0}, v[3] = {r.roll, r.pitch, r.yaw};
    rcsh::pose_from_rpy(v, tr ? need(*tr, 3, "translation") : zero, o.p);
    return o;
  }
  static Pose from_translation(const darr& tr) {
    Pose o;
    std::memcpy(o.p.t, need(tr, 3, "translation"), sizeof(o.p.t));
    return o;
  }
  darr rotation_m() const { double m[9]; rcsh::quat_to_mat(p.q, m); return mat(m, 3, 3); }
  darr pose_matrix() const {
    double m[9], o[16] = {0};
    rcsh::quat_to_mat(p.q, m);
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) o[4 * i + j] = m[3 * i + j]; o[4 * i + 3] = p.t[i]; }
    o[15] = 1;
    return mat(o, 4, 4);
  }
  RPY rotation_rpy() const { double r[3]; rcsh::pose_rpy(p, r); return RPY{r[0], r[1], r[2]}; }
  darr xyzrpy() const { double o[6] = {p.t[0], p.t[1], p.t[2]}; rcsh::pose_rpy(p, o + 3); return vec(o, 6); }
  Pose mul(const Pose& b) const { Pose o; rcsh::pose_mul(p, b.p, o.p); return o; }
  Pose inverse() const { Pose o; rcsh::pose_inverse(p, o.p); return o; }
  double total_angle() const { const double id[4] = {0, 0, 0, 1}; return rcsh::quat_angular_distance(p.q, id); }
  Pose limit_rotation_angle(double a) const { Pose o; rcsh::pose_limit_rotation_angle(p, a, o.p); return o; }
  Pose limit_translation_length(double l) const { Pose o; rcsh::pose_limit_translation_length(p, l, o.p); return o; }
  Pose interpolate(const Pose& dest, double progress) const {  // Pose.cpp:180-191
    progress = progress > 1.0 ? 1.0 : progress;
    Pose o;
    rcsh::quat_slerp(p.q, progress, dest.p.q, o.p.q);
    rcsh::quat_normalize(o.p.q);
    for (int k = 0; k < 3; ++k) o.p.t[k] = p.t[k] + (dest.p.t[k] - p.t[k]) * progress;
    return o;
  }
  bool is_close(const Pose& b, double eps_r, double eps_t) const {  // Pose.cpp:193-198: L1 of the translation, angular distance
    const double dt = std::fabs(p.t[0] - b.p.t[0]) + std::fabs(p.t[1] - b.p.t[1]) + std::fabs(p.t[2] - b.p.t[2]);
    return dt < eps_t && rcsh::quat_angular_distance(p.q, b.p.q) < eps_r;
  }
  std::string str() const {
    std::ostringstream s;
    double m[9];
    rcsh::quat_to_mat(p.q, m);
    for (int i = 0; i < 3; ++i) s << m[3 * i] << " " << m[3 * i + 1] << " " << m[3 * i + 2] << " " << p.t[i] << "\n";
    s << "0 0 0 1\n";
    const RPY r = rotation_rpy();
    s << "roll: " << r.roll << "\tpitch: " << r.pitch << "\tyaw: " << r.yaw;
    return s.str();
  }
  void vec7(double* o) const { std::memcpy(o, p.t, 3 * sizeof(double)); std::memcpy(o + 3, p.q, 4 * sizeof(double)); }
};

// ---- robots_meta_config (reference include/rcs/Robot.h:16-95)
enum class RobotType : int { FR3 = 0, UR5e = 1, SO101 = 2, XArm7 = 3 };
enum class RobotPlatform : int { SIMULATION = 0, HARDWARE = 1 };
enum class GraspType : int { POWER_GRASP = 0, PRECISION_GRASP = 1, LATERAL_GRASP = 2, TRIPOD_GRASP = 3 };
struct RobotMetaConfig {
  std::vector<double> q_home, low, high;
  int dof = 0;
};
RobotMetaConfig robots_meta_config(RobotType t) {
  const double p2 = 2 * kPi;
  switch (t) {
    case RobotType::FR3:
      return {{0.0, -kPi / 4, 0.0, -3.0 * kPi / 4, 0.0, kPi / 2, kPi / 4},
              {-2.3093, -1.5133, -2.4937, -2.7478, -2.4800, 0.8521, -2.6895}, {2.3093, 1.5133, 2.4937, -0.4461, 2.4800, 4.2094, 2.6895}, 7};
    case RobotType::UR5e:
      return {{-0.4488354, -2.02711196, 1.64630026, -1.18999615, -1.57079762, -2.01963249},
              {-p2, -p2, -kPi, -p2, -p2, -p2}, {p2, p2, kPi, p2, p2, p2}, 6};
    case RobotType::XArm7:
      return {{0, -45.0 / 180.0 * kPi, 0, 15.0 / 180.0 * kPi, 0, -25.0 / 180.0 * kPi, 0},
              {-p2, -2.094395, -p2, -3.92699, -p2, -kPi, -p2}, {p2, 2.059488, p2, 0.191986, p2, 1.692969, p2}, 7};
    case RobotType::SO101:
      return {{-9.40612320177057, -99.66130397967824, 99.9124726477024, 69.96996996996998, -9.095744680851055},
              {-100, -100, -100, -100, -100}, {100, 100, 100, 100, 100}, 5};
  }
  throw std::invalid_argument("unknown robot type");
}
struct RobotConfig {  // Robot.h:97-104
  RobotType robot_type = RobotType::FR3;
  RobotPlatform robot_platform = RobotPlatform::SIMULATION;
  Pose tcp_offset;
  std::string attachment_site = "attachment_site";
  std::string kinematic_model_path = "assets/scenes/fr3_empty_world/robot.xml";
};
struct BaseCameraConfig {  // include/rcs/Camera.h
  std::string identifier;
  int frame_rate = 0, resolution_width = 0, resolution_height = 0;
};

// ---- Kinematics / Pin (reference include/rcs/Kinematics.h, src/rcs/Kinematics.cpp)
struct Kinematics {
  virtual ~Kinematics() = default;
  virtual std::optional<darr> inverse(const Pose& pose, const darr& q0, const Pose& tcp_offset) = 0;
  virtual Pose forward(const darr& q0, const Pose& tcp_offset) = 0;
};
struct Pin : Kinematics {
  rcsh_sim* h = nullptr;
  int dof = 0, nq = 0;
  std::vector<py::array> keep;
  Pin(const std::string& path, const std::string& frame_id, bool urdf) {
    // the scene compiler is host Python (rcs_amd.mjcf; a URDF is rewritten as MJCF first, rcs_amd.urdf: its links are the frames);
    // rcs_hip.pin_tables returns mjModel-named tables + the chain's ids
    const py::dict t = py::module_::import("rcs_hip").attr("pin_tables")(path, frame_id, 0, urdf).cast<py::dict>();
    const py::dict model = t["model"].cast<py::dict>();
    rcsh_model_desc d{};
    auto geti = [&](const char* k) { return model.contains(k) ? model[k].cast<int>() : 0; };
    d.nbody = geti("nbody"); d.njnt = geti("njnt"); d.nu = geti("nu"); d.ntendon = geti("ntendon"); d.nwrap = geti("nwrap");
    d.neq = geti("neq"); d.nsite = geti("nsite"); d.ngeom = geti("ngeom"); d.nmeshvert = geti("nmeshvert");
    d.timestep = model["timestep"].cast<double>();
    {
      darr g = model["gravity"].cast<darr>();
      for (int k = 0; k < 3; ++k) d.gravity[k] = g.data()[k];
    }
    using iarr = py::array_t<int32_t, py::array::c_style | py::array::forcecast>;
    auto f64 = [&](const char* k) -> const double* { darr a = model[k].cast<darr>(); keep.push_back(a); return a.size() ? a.data() : nullptr; };
    auto i32 = [&](const char* k) -> const int32_t* { iarr a = model[k].cast<iarr>(); keep.push_back(a); return a.size() ? a.data() : nullptr; };
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
    check(rcsh_sim_create(&d, 1, t.contains("device") ? t["device"].cast<int>() : 0, &h));
    const iarr joints = t["joints"].cast<iarr>(), acts = t["actuators"].cast<iarr>();
    dof = (int)joints.size();
    std::vector<double> q_home(dof, 0.0);
    rcsh_robot_desc r{};
    r.dof = dof; r.joint_ids = joints.data(); r.actuator_ids = acts.data();
    r.attachment_site = t["site"].cast<int>(); r.base_body = t["base"].cast<int>();
    r.q_home = q_home.data();
    r.tcp_offset[6] = 1.0;
    r.joint_rotational_tolerance = .05 * (kPi / 180.0); r.seconds_between_callbacks = 0.1;
    r.register_convergence_callback = 1;
    int32_t zero = 0;
    r.n_collision_geoms = 0; r.collision_geom_ids = &zero;
    const int rc = rcsh_sim_add_robot(h, &r);
    if (rc != RCSH_OK) { rcsh_sim_destroy(h); h = nullptr; check(rc); }
    nq = rcsh_sim_nq(h);
  }
  ~Pin() override { if (h) rcsh_sim_destroy(h); }
  Pin(const Pin&) = delete;
  darr q_in(const darr& q0) const {
    if (q0.size() < dof) throw std::invalid_argument("q0: one entry per joint of the chain");
    darr q(dof);
    for (int i = 0; i < dof; ++i) q.mutable_data()[i] = q0.data()[i];
    return q;
  }
  // Pin::inverse (Kinematics.cpp:28-68): model.nq entries (quirk Q7) or None when the CLIK hits its iteration cap
  std::optional<darr> inverse(const Pose& pose, const darr& q0, const Pose& tcp_offset) override {
    double p7[7], o7[7];
    pose.vec7(p7); tcp_offset.vec7(o7);
    const darr q = q_in(q0);
    darr out(nq);
    uint8_t ok = 0;
    int32_t it = 0;
    { py::gil_scoped_release nogil; check(rcsh_ik_inverse(h, p7, q.data(), o7, out.mutable_data(), &ok, &it)); }
    if (!ok) return std::nullopt;
    return out;
  }
  Pose forward(const darr& q0, const Pose& tcp_offset) override {  // Kinematics.cpp:70-82
    double o7[7], p7[7];
    tcp_offset.vec7(o7);
    const darr q = q_in(q0);
    check(rcsh_ik_forward(h, q.data(), o7, p7));
    rcsh::Pose r;
    std::memcpy(r.t, p7, sizeof(r.t)); std::memcpy(r.q, p7 + 3, sizeof(r.q));
    return Pose(r);
  }
};

// rcs_robotics_library._core.rl.RoboticsLibraryIK (reference extensions/rcs_robotics_library/src/pybind/RL.h:18-70): `Kinematics` on a
// URDF, operational frame 0 = the chain's last link.  The reference hands the solve to Robotics Library's JacobianInverseKinematics
// (eps 1e-3, no random restarts, a wall-clock budget of max_duration_ms); RL is a third-party dependency that is not vendored in
// the reference tree, so its iteration is not restated here: the solve is the backend's CLIK -- Pin::inverse's damped least squares
// (src/rcs/Kinematics.cpp:28-68) as a HIP kernel, 1000 iterations at most, which at ~6k cycles an iteration is far inside any
// budget a caller passes -- and `forward` is the same chain's forward map.  PARITY WITH RL IS UNPINNED (interface only):
// both return a joint vector that reaches the pose, or None; the iterates differ.
struct RoboticsLibraryIK : Pin {
  size_t max_duration_ms;
  RoboticsLibraryIK(const std::string& urdf_path, size_t max_duration_ms_) : Pin(urdf_path, "", true), max_duration_ms(max_duration_ms_) {}
};

}  // namespace

void bind_common(py::module_& m) {
  auto common = m.def_submodule("common", "common module");
  common.def("IdentityTranslation", [] { const double z[3] = {0, 0, 0}; return vec(z, 3); });
  common.def("IdentityRotMatrix", [] { const double e[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; return mat(e, 3, 3); });
  common.def("IdentityRotQuatVec", [] { const double q[4] = {0, 0, 0, 1}; return vec(q, 4); });
  common.def("FrankaHandTCPOffset", [] {  // src/rcs/Pose.cpp:11-15
    const double t[16] = {0.707, 0.707, 0, 0, -0.707, 0.707, 0, 0, 0, 0, 1, 0.1034, 0, 0, 0, 1};
    return mat(t, 4, 4); });

  py::class_<RPY>(common, "RPY")
      .def(py::init([](double roll, double pitch, double yaw) { return RPY{roll, pitch, yaw}; }), py::arg("roll") = 0.0, py::arg("pitch") = 0.0, py::arg("yaw") = 0.0)
      .def(py::init([](const Vec3& rpy) { const double* v = rpy.data(); return RPY{v[0], v[1], v[2]}; }), py::arg("rpy"))
      .def_readwrite("roll", &RPY::roll)
      .def_readwrite("pitch", &RPY::pitch)
      .def_readwrite("yaw", &RPY::yaw)
      .def("rotation_matrix", [](const RPY& r) { double q[4], mm[9]; r.quaternion(q); rcsh::quat_to_mat(q, mm); return mat(mm, 3, 3); })
      .def("as_vector", [](const RPY& r) { const double v[3] = {r.roll, r.pitch, r.yaw}; return vec(v, 3); })
      .def("as_quaternion_vector", [](const RPY& r) { double q[4]; r.quaternion(q); return vec(q, 4); })
      .def("is_close", [](const RPY& a, const RPY& b, double eps) {
        return std::fabs(a.roll - b.roll) + std::fabs(a.pitch - b.pitch) + std::fabs(a.yaw - b.yaw) < eps; }, py::arg("other"), py::arg("eps") = 1e-8)
      .def("__str__", [](const RPY& r) { std::ostringstream s; s << "RPY(" << r.roll << ", " << r.pitch << ", " << r.yaw << ")"; return s.str(); })
      .def(py::self + py::self)
      .def(py::pickle([](const RPY& r) { return py::make_tuple(r.roll, r.pitch, r.yaw); },
                      [](const py::tuple& t) { if (t.size() != 3) throw std::runtime_error("Invalid state!"); return RPY{t[0].cast<double>(), t[1].cast<double>(), t[2].cast<double>()}; }));

  py::class_<Pose>(common, "Pose")
      .def(py::init<>())
      .def(py::init([](const Mat4& m) { return Pose::from_matrix(m.a); }), py::arg("pose_matrix"))
      .def(py::init([](const Mat3& r, const Vec3& t) { return Pose::from_rotation(r.a, &t.a); }), py::arg("rotation"), py::arg("translation"))
      .def(py::init([](const Vec4& q, const Vec3& t) { return Pose::from_quaternion(q.a, &t.a); }), py::arg("quaternion"), py::arg("translation"))
      .def(py::init([](const RPY& r, const Vec3& t) { return Pose::from_rpy(r, &t.a); }), py::arg("rpy"), py::arg("translation"))
      .def(py::init([](const Vec3& v, const Vec3& t) { const double* x = v.data(); return Pose::from_rpy(RPY{x[0], x[1], x[2]}, &t.a); }),
           py::arg("rpy_vector"), py::arg("translation"))
      .def(py::init([](const Vec3& t) { return Pose::from_translation(t.a); }), py::arg("translation"))
      .def(py::init([](const Vec4& q) { return Pose::from_quaternion(q.a, nullptr); }), py::arg("quaternion"))
      .def(py::init([](const RPY& r) { return Pose::from_rpy(r, nullptr); }), py::arg("rpy"))
      .def(py::init([](const Mat3& r) { return Pose::from_rotation(r.a, nullptr); }), py::arg("rotation"))
      .def(py::init([](const Pose& p) { return Pose(p); }), py::arg("pose"))
      .def("translation", [](const Pose& p) { return vec(p.p.t, 3); })
      .def("rotation_m", &Pose::rotation_m)
      .def("rotation_q", [](const Pose& p) { return vec(p.p.q, 4); })
      .def("pose_matrix", &Pose::pose_matrix)
      .def("rotation_rpy", &Pose::rotation_rpy)
      .def("xyzrpy", &Pose::xyzrpy)
      .def("interpolate", &Pose::interpolate, py::arg("dest_pose"), py::arg("progress"))
      .def("inverse", &Pose::inverse)
      .def("limit_rotation_angle", &Pose::limit_rotation_angle, py::arg("max_angle"))
      .def("limit_translation_length", &Pose::limit_translation_length, py::arg("max_length"))
      .def("is_close", &Pose::is_close, py::arg("other"), py::arg("eps_r") = 1e-8, py::arg("eps_t") = 1e-8)
      .def("total_angle", &Pose::total_angle)
      .def("__str__", &Pose::str)
      .def("__mul__", &Pose::mul, py::is_operator())
      .def(py::pickle([](const Pose& p) { const darr m = p.pose_matrix(); return std::vector<double>(m.data(), m.data() + 16); },
                      [](const std::vector<double>& v) {
                        if (v.size() != 16) throw std::runtime_error("Invalid state!");
                        darr m({4, 4});
                        std::memcpy(m.mutable_data(), v.data(), sizeof(double) * 16);
                        return Pose::from_matrix(m); }));

  py::enum_<RobotType>(common, "RobotType")
      .value("FR3", RobotType::FR3).value("UR5e", RobotType::UR5e).value("SO101", RobotType::SO101).value("XArm7", RobotType::XArm7).export_values();
  py::enum_<RobotPlatform>(common, "RobotPlatform")
      .value("HARDWARE", RobotPlatform::HARDWARE).value("SIMULATION", RobotPlatform::SIMULATION).export_values();
  py::enum_<GraspType>(common, "GraspType")
      .value("POWER_GRASP", GraspType::POWER_GRASP).value("PRECISION_GRASP", GraspType::PRECISION_GRASP)
      .value("LATERAL_GRASP", GraspType::LATERAL_GRASP).value("TRIPOD_GRASP", GraspType::TRIPOD_GRASP).export_values();

  py::class_<RobotMetaConfig>(common, "RobotMetaConfig")
      .def_property_readonly("q_home", [](const RobotMetaConfig& c) { return vec(c.q_home.data(), (py::ssize_t)c.q_home.size()); })
      .def_readonly("dof", &RobotMetaConfig::dof)
      .def_property_readonly("joint_limits", [](const RobotMetaConfig& c) {
        darr o({(py::ssize_t)2, (py::ssize_t)c.dof});
        for (int i = 0; i < c.dof; ++i) { o.mutable_at(0, i) = c.low[i]; o.mutable_at(1, i) = c.high[i]; }
        return o; });
  common.def("robots_meta_config", &robots_meta_config, py::arg("robot_type"));

  py::class_<RobotConfig>(common, "RobotConfig")
      .def(py::init<>())
      .def_readwrite("robot_type", &RobotConfig::robot_type)
      .def_readwrite("kinematic_model_path", &RobotConfig::kinematic_model_path)
      .def_readwrite("attachment_site", &RobotConfig::attachment_site)
      .def_readwrite("robot_platform", &RobotConfig::robot_platform)
      .def_readwrite("tcp_offset", &RobotConfig::tcp_offset);
  py::class_<BaseCameraConfig>(common, "BaseCameraConfig")
      .def(py::init([](const std::string& identifier, int frame_rate, int w, int h) { return BaseCameraConfig{identifier, frame_rate, w, h}; }),
           py::arg("identifier"), py::arg("frame_rate"), py::arg("resolution_width"), py::arg("resolution_height"))
      .def_readwrite("identifier", &BaseCameraConfig::identifier)
      .def_readwrite("frame_rate", &BaseCameraConfig::frame_rate)
      .def_readwrite("resolution_width", &BaseCameraConfig::resolution_width)
      .def_readwrite("resolution_height", &BaseCameraConfig::resolution_height);

  py::class_<Kinematics, std::shared_ptr<Kinematics>>(common, "Kinematics")
      .def("forward", &Kinematics::forward, py::arg("q0"), py::arg("tcp_offset") = Pose())
      .def("inverse", &Kinematics::inverse, py::arg("pose"), py::arg("q0"), py::arg("tcp_offset") = Pose());
  py::class_<Pin, Kinematics, std::shared_ptr<Pin>>(common, "Pin")
      .def(py::init<const std::string&, const std::string&, bool>(), py::arg("path"), py::arg("frame_id") = "fr3_link8", py::arg("urdf") = true);
  // the reference binds this class in a module of its own, rcs_robotics_library._core.rl (extensions/rcs_robotics_library/src/pybind/rcs.cpp:37-47)
  auto rl = m.def_submodule("rl", "rcs robotics library module");
  py::class_<RoboticsLibraryIK, Kinematics, std::shared_ptr<RoboticsLibraryIK>>(rl, "RoboticsLibraryIK")
      .def(py::init<const std::string&, size_t>(), py::arg("urdf_path"), py::arg("max_duration_ms") = 300);
}
