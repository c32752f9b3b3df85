// model.cpp -- host-side model finalisation: mjModel-like tables -> link tables (model.h).
//
// Replaces, for the supported scene family, the part of MuJoCo's model compiler that
// runs after parsing: body-tree bookkeeping and the qpos0 constants (dof_invweight0).
// It additionally folds every body without a joint into the nearest moving ancestor,
// which MuJoCo does not do at run time; the dynamics are identical (a rigid union of
// rigid bodies) and the device loops shrink from 14 bodies to 9 links.
#include "model_host.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <sstream>

#include "dyn.h"

namespace rcsh {

namespace {

struct Xf {  // rigid transform child -> parent: x_parent = R x_child + p
  double R[9];
  double p[3];
};

Xf xf_identity() { return Xf{{1, 0, 0, 0, 1, 0, 0, 0, 1}, {0, 0, 0}}; }

void quat_to_mat(const double* q, double* m) {  // wxyz, normalised by the scene compiler
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}

Xf xf_from(const double* pos, const double* quat) {
  Xf t;
  quat_to_mat(quat, t.R);
  std::memcpy(t.p, pos, sizeof(t.p));
  return t;
}

Xf xf_mul(const Xf& a, const Xf& b) {  // a o b
  Xf r;
  mulmm(a.R, b.R, r.R);
  mulmv(a.R, b.p, r.p);
  for (int k = 0; k < 3; ++k) r.p[k] += a.p[k];
  return r;
}

void mat_to_quat_wxyz(const double* m, double* q) {
  const double t = m[0] + m[4] + m[8];
  if (t > 0) {
    const double s = std::sqrt(t + 1.0) * 2;
    q[0] = 0.25 * s; q[1] = (m[7] - m[5]) / s; q[2] = (m[2] - m[6]) / s; q[3] = (m[3] - m[1]) / s;
  } else if (m[0] > m[4] && m[0] > m[8]) {
    const double s = std::sqrt(1.0 + m[0] - m[4] - m[8]) * 2;
    q[0] = (m[7] - m[5]) / s; q[1] = 0.25 * s; q[2] = (m[1] + m[3]) / s; q[3] = (m[2] + m[6]) / s;
  } else if (m[4] > m[8]) {
    const double s = std::sqrt(1.0 + m[4] - m[0] - m[8]) * 2;
    q[0] = (m[2] - m[6]) / s; q[1] = (m[1] + m[3]) / s; q[2] = 0.25 * s; q[3] = (m[5] + m[7]) / s;
  } else {
    const double s = std::sqrt(1.0 + m[8] - m[0] - m[4]) * 2;
    q[0] = (m[3] - m[1]) / s; q[1] = (m[2] + m[6]) / s; q[2] = (m[5] + m[7]) / s; q[3] = 0.25 * s;
  }
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int k = 0; k < 4; ++k) q[k] /= n;
}

}  // namespace

// solimp -> kernel-side form (clamps follow MuJoCo's: d0, d1, midpoint in [0.0001, 0.9999], width >= 0, power >= 1)
Imp make_imp(const double* solimp) {
  Imp p;
  p.d0 = clampd(solimp[0], kMinImp, kMaxImp);
  p.d1 = clampd(solimp[1], kMinImp, kMaxImp);
  const double width = solimp[2] < 0 ? 0 : solimp[2];
  p.mid = clampd(solimp[3], kMinImp, kMaxImp);
  p.power = solimp[4] < 1 ? 1 : solimp[4];
  p.inv_width = width > kMinVal ? 1.0 / width : 0.0;
  p.inv_mid = 1.0 / p.mid;
  p.inv_1mmid = 1.0 / (1.0 - p.mid);
  p.pad = 0;
  if (p.d0 == p.d1 || width <= kMinVal) p.mode = 0;
  else if (p.power == 1) p.mode = 1;
  else if (p.power == 2) p.mode = 2;
  else p.mode = 3;
  return p;
}

// solref -> stiffness K and damping B of the reference acceleration (timeconst floored at 2 timesteps)
void make_kb(const double* solref, const double* solimp, double timestep, double& K, double& B) {
  const double dmax = clampd(solimp[1], kMinImp, kMaxImp);
  double tc = solref[0];
  const double dr = solref[1];
  if (tc > 0) {
    if (tc < 2 * timestep) tc = 2 * timestep;
    const double kd = dmax * dmax * tc * tc * dr * dr, bd = dmax * tc;
    K = 1.0 / (kd > kMinVal ? kd : kMinVal);
    B = 2.0 / (bd > kMinVal ? bd : kMinVal);
  } else {
    K = -tc / (dmax * dmax);
    B = -dr / dmax;
  }
}

namespace {

template <class T>
void compute_invweight0(DevModel& m) {
  double q[T::NL], qd[T::NL];
  for (int i = 0; i < T::NL; ++i) { q[i] = m.qpos0[i]; qd[i] = 0; }
  Smooth<T> sm;
  double buf[Stage<T, 1>::COUNT];
  Stage<T, 1> st{buf};
  smooth_dynamics<T, 1>(m, q, qd, st, sm);
  double* M = &st.M(0);
  m.inertia_diag_sum = 0;
  for (int j = 0; j < T::NL; ++j) m.inertia_diag_sum += st.M(tri(j, j));
  ldl_factor<T::NL>(M);
  for (int j = 0; j < T::NL; ++j) {
    double e[T::NL];
    for (int i = 0; i < T::NL; ++i) e[i] = i == j ? 1.0 : 0.0;
    ldl_solve<T::NL>(M, e);
    m.invweight0[j] = e[j];
  }
}

}  // namespace

void HostModel::copy_from(const rcsh_model_desc& d) {
  nbody = d.nbody; njnt = d.njnt; nu = d.nu; ntendon = d.ntendon; nwrap = d.nwrap; neq = d.neq; nsite = d.nsite;
  timestep = d.timestep;
  std::memcpy(gravity, d.gravity, sizeof(gravity));
  auto cpi = [](std::vector<int32_t>& v, const int32_t* p, size_t n) { v.assign(p, p + n); };
  auto cpd = [](std::vector<double>& v, const double* p, size_t n) { v.assign(p, p + n); };
  cpi(body_parentid, d.body_parentid, nbody); cpi(body_jntadr, d.body_jntadr, nbody); cpi(body_jntnum, d.body_jntnum, nbody);
  cpd(body_pos, d.body_pos, 3 * nbody); cpd(body_quat, d.body_quat, 4 * nbody);
  cpd(body_ipos, d.body_ipos, 3 * nbody); cpd(body_iquat, d.body_iquat, 4 * nbody);
  cpd(body_mass, d.body_mass, nbody); cpd(body_inertia, d.body_inertia, 3 * nbody); cpd(body_gravcomp, d.body_gravcomp, nbody);
  cpi(jnt_type, d.jnt_type, njnt); cpi(jnt_bodyid, d.jnt_bodyid, njnt);
  cpd(jnt_pos, d.jnt_pos, 3 * njnt); cpd(jnt_axis, d.jnt_axis, 3 * njnt);
  cpi(jnt_limited, d.jnt_limited, njnt); cpd(jnt_range, d.jnt_range, 2 * njnt); cpd(jnt_margin, d.jnt_margin, njnt);
  cpd(jnt_solref, d.jnt_solref, 2 * njnt); cpd(jnt_solimp, d.jnt_solimp, 5 * njnt);
  cpi(jnt_actfrclimited, d.jnt_actfrclimited, njnt); cpd(jnt_actfrcrange, d.jnt_actfrcrange, 2 * njnt);
  cpi(jnt_actgravcomp, d.jnt_actgravcomp, njnt);
  cpd(dof_armature, d.dof_armature, njnt); cpd(dof_damping, d.dof_damping, njnt); cpd(dof_frictionloss, d.dof_frictionloss, njnt);
  cpd(dof_solref, d.dof_solref, 2 * njnt); cpd(dof_solimp, d.dof_solimp, 5 * njnt);
  cpd(qpos0, d.qpos0, njnt);
  cpi(tendon_adr, d.tendon_adr, ntendon); cpi(tendon_num, d.tendon_num, ntendon);
  cpi(wrap_objid, d.wrap_objid, nwrap); cpd(wrap_prm, d.wrap_prm, nwrap);
  cpi(eq_obj1id, d.eq_obj1id, neq); cpi(eq_obj2id, d.eq_obj2id, neq); cpi(eq_active0, d.eq_active0, neq);
  cpd(eq_data, d.eq_data, 5 * neq); cpd(eq_solref, d.eq_solref, 2 * neq); cpd(eq_solimp, d.eq_solimp, 5 * neq);
  cpi(actuator_trntype, d.actuator_trntype, nu); cpi(actuator_trnid, d.actuator_trnid, nu);
  cpd(actuator_gear, d.actuator_gear, nu); cpd(actuator_gainprm, d.actuator_gainprm, 3 * nu);
  cpd(actuator_biasprm, d.actuator_biasprm, 3 * nu); cpi(actuator_biastype, d.actuator_biastype, nu);
  cpi(actuator_ctrllimited, d.actuator_ctrllimited, nu); cpd(actuator_ctrlrange, d.actuator_ctrlrange, 2 * nu);
  cpi(actuator_forcelimited, d.actuator_forcelimited, nu); cpd(actuator_forcerange, d.actuator_forcerange, 2 * nu);
  cpi(site_bodyid, d.site_bodyid, nsite); cpd(site_pos, d.site_pos, 3 * nsite); cpd(site_quat, d.site_quat, 4 * nsite);
  ngeom = d.ngeom; nmeshvert = d.nmeshvert;
  cpi(geom_type, d.geom_type, ngeom); cpi(geom_bodyid, d.geom_bodyid, ngeom);
  cpi(geom_contype, d.geom_contype, ngeom); cpi(geom_conaffinity, d.geom_conaffinity, ngeom);
  cpd(geom_pos, d.geom_pos, 3 * ngeom); cpd(geom_quat, d.geom_quat, 4 * ngeom); cpd(geom_size, d.geom_size, 3 * ngeom);
  if (d.geom_friction) cpd(geom_friction, d.geom_friction, 3 * ngeom);
  else geom_friction.assign(3 * (size_t)ngeom, 1.0);
  cpi(geom_vertadr, d.geom_vertadr, ngeom); cpi(geom_vertnum, d.geom_vertnum, ngeom);
  cpd(mesh_vert, d.mesh_vert, 3 * (size_t)nmeshvert);
}

// Transform of every body frame relative to the link (moving ancestor-or-self) that owns it, or
// relative to the world for static bodies.
static void body_owner_frames(const HostModel& h, std::vector<int>& owner, std::vector<Xf>& rel) {
  owner.assign(h.nbody, -1);
  rel.assign(h.nbody, xf_identity());
  for (int b = 1; b < h.nbody; ++b) {
    if (h.body_jntnum[b] > 0) {
      owner[b] = h.body_jntadr[b];  // link index == dof index == joint index
      rel[b] = xf_identity();
    } else {
      const int p = h.body_parentid[b];
      owner[b] = owner[p];
      rel[b] = xf_mul(rel[p], xf_from(&h.body_pos[3 * b], &h.body_quat[4 * b]));
    }
  }
}

std::string finalize_model(const HostModel& h, DevModel& m, std::vector<int>& act_slot) {
  std::memset(&m, 0, sizeof(m));
  std::ostringstream err;
  if (h.njnt < 1 || h.njnt > kMaxLinks) return "scene has an unsupported number of joints";
  for (int b = 0; b < h.nbody; ++b)
    if (h.body_jntnum[b] > 1) return "bodies with more than one joint are outside the supported archetypes";
  for (int j = 0; j < h.njnt; ++j) {
    if (h.jnt_type[j] != kSlide && h.jnt_type[j] != kHinge) return "only hinge and slide joints are supported";
    if (h.dof_frictionloss[j] < 0) return "negative joint frictionloss";
    if (h.body_jntadr[h.jnt_bodyid[j]] != j) return "joint/body addressing is inconsistent";
  }
  std::vector<int> owner;
  std::vector<Xf> rel;
  body_owner_frames(h, owner, rel);
  const int nl = h.njnt;
  // link parents
  std::vector<int> lparent(nl);
  for (int i = 0; i < nl; ++i) lparent[i] = owner[h.body_parentid[h.jnt_bodyid[i]]];
  // archetype: serial arm [0, narm), optionally two slide fingers hanging off the last arm link
  int narm = 0;
  while (narm < nl && lparent[narm] == narm - 1 && h.jnt_type[narm] == kHinge) ++narm;
  bool grip = false;
  if (narm == nl) {
    grip = false;
  } else if (nl == narm + 2 && lparent[narm] == narm - 1 && lparent[narm + 1] == narm - 1 &&
             h.jnt_type[narm] == kSlide && h.jnt_type[narm + 1] == kSlide) {
    grip = true;
  } else {
    return "scene does not match a compiled archetype (serial hinge arm, optional two-finger slide gripper)";
  }
  if (narm < 1 || narm > kMaxArm) return "arm length outside the compiled range";
  m.nl = nl; m.narm = narm; m.has_gripper = grip ? 1 : 0;
  m.timestep = h.timestep;
  std::memcpy(m.gravity, h.gravity, sizeof(m.gravity));
  m.site_link = -1;

  for (int i = 0; i < nl; ++i) {
    const int b = h.jnt_bodyid[i];
    const int pb = h.body_parentid[b];
    const Xf t = xf_mul(rel[pb], xf_from(&h.body_pos[3 * b], &h.body_quat[4 * b]));
    std::memcpy(m.pos0[i], t.p, sizeof(t.p));
    std::memcpy(m.rot0[i], t.R, sizeof(t.R));
    std::memcpy(m.axis[i], &h.jnt_axis[3 * i], 3 * sizeof(double));
    std::memcpy(m.jpos[i], &h.jnt_pos[3 * i], 3 * sizeof(double));
    m.jtype[i] = h.jnt_type[i];
    m.qpos0[i] = h.qpos0[i];
    m.armature[i] = h.dof_armature[i];
    m.damping[i] = h.dof_damping[i];
    m.limited[i] = h.jnt_limited[i];
    m.range[i][0] = h.jnt_range[2 * i]; m.range[i][1] = h.jnt_range[2 * i + 1];
    m.margin[i] = h.jnt_margin[i];
    m.lim_imp[i] = make_imp(&h.jnt_solimp[5 * i]);
    make_kb(&h.jnt_solref[2 * i], &h.jnt_solimp[5 * i], h.timestep, m.lim_K[i], m.lim_B[i]);
    m.axis_z[i] = h.jnt_type[i] == kHinge && m.axis[i][0] == 0 && m.axis[i][1] == 0 && m.axis[i][2] == 1 &&
                  m.jpos[i][0] == 0 && m.jpos[i][1] == 0 && m.jpos[i][2] == 0;
    m.actfrclimited[i] = h.jnt_actfrclimited[i];
    m.actfrcrange[i][0] = h.jnt_actfrcrange[2 * i]; m.actfrcrange[i][1] = h.jnt_actfrcrange[2 * i + 1];
    m.actgravcomp[i] = h.jnt_actgravcomp[i];
  }
  // composite inertials
  for (int i = 0; i < nl; ++i) {
    double mass = 0, mc[3] = {0, 0, 0}, gcm = 0, gmc[3] = {0, 0, 0};
    for (int b = 1; b < h.nbody; ++b) {
      if (owner[b] != i) continue;
      double c[3];
      mulmv(rel[b].R, &h.body_ipos[3 * b], c);
      for (int k = 0; k < 3; ++k) c[k] += rel[b].p[k];
      const double mb = h.body_mass[b], gb = h.body_gravcomp[b] * mb;
      mass += mb; gcm += gb;
      for (int k = 0; k < 3; ++k) { mc[k] += mb * c[k]; gmc[k] += gb * c[k]; }
    }
    double com[3] = {0, 0, 0};
    if (mass > 0) for (int k = 0; k < 3; ++k) com[k] = mc[k] / mass;
    double J[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 1; b < h.nbody; ++b) {
      if (owner[b] != i) continue;
      double Rq[9], Rb[9], tmp[9], Jb[9];
      quat_to_mat(&h.body_iquat[4 * b], Rq);
      mulmm(rel[b].R, Rq, Rb);
      const double* I3 = &h.body_inertia[3 * b];
      for (int r = 0; r < 3; ++r)
        for (int cc = 0; cc < 3; ++cc) tmp[3 * r + cc] = Rb[3 * r + cc] * I3[cc];
      for (int r = 0; r < 3; ++r)
        for (int cc = 0; cc < 3; ++cc)
          Jb[3 * r + cc] = tmp[3 * r] * Rb[3 * cc] + tmp[3 * r + 1] * Rb[3 * cc + 1] + tmp[3 * r + 2] * Rb[3 * cc + 2];
      double c[3], dd[3];
      mulmv(rel[b].R, &h.body_ipos[3 * b], c);
      for (int k = 0; k < 3; ++k) dd[k] = c[k] + rel[b].p[k] - com[k];
      const double mb = h.body_mass[b], d2 = dot3(dd, dd);
      for (int r = 0; r < 3; ++r)
        for (int cc = 0; cc < 3; ++cc) J[3 * r + cc] += Jb[3 * r + cc] + mb * ((r == cc ? d2 : 0.0) - dd[r] * dd[cc]);
    }
    m.mass[i] = mass;
    std::memcpy(m.com[i], com, sizeof(com));
    m.inertia[i][0] = J[0]; m.inertia[i][1] = J[4]; m.inertia[i][2] = J[8];
    m.inertia[i][3] = J[1]; m.inertia[i][4] = J[2]; m.inertia[i][5] = J[5];
    m.gcm[i] = gcm;
    for (int k = 0; k < 3; ++k) m.gccom[i][k] = gcm != 0 ? gmc[k] / gcm : 0.0;
    m.gc_same_com[i] = 1;
    for (int k = 0; k < 3; ++k)
      if (m.gccom[i][k] != m.com[i][k]) m.gc_same_com[i] = 0;
  }
  // subtree gravcomp mass (constant): the link and all of its descendants
  for (int i = 0; i < nl; ++i) {
    double s = 0;
    for (int k = i; k < nl; ++k) {
      int a = k;
      while (a >= 0 && a != i) a = lparent[a];
      if (a == i) s += m.gcm[k];
    }
    m.gcm_sub[i] = s;
  }
  // actuators
  act_slot.assign(h.nu, -1);
  for (int u = 0; u < h.nu; ++u) {
    const double* gp = &h.actuator_gainprm[3 * u];
    const double* bp = &h.actuator_biasprm[3 * u];
    if (h.actuator_trntype[u] == 0) {
      const int j = h.actuator_trnid[u];
      if (j < 0 || j >= narm) return "joint-transmission actuators are supported on arm joints only";
      if (m.arm_has_act[j]) return "more than one actuator on an arm joint";
      m.arm_has_act[j] = 1;
      m.arm_gear[j] = h.actuator_gear[u];
      m.arm_gain[j] = gp[0];
      std::memcpy(m.arm_bias[j], bp, 3 * sizeof(double));
      m.arm_biasaffine[j] = h.actuator_biastype[u];
      m.arm_ctrllimited[j] = h.actuator_ctrllimited[u];
      m.arm_ctrlrange[j][0] = h.actuator_ctrlrange[2 * u]; m.arm_ctrlrange[j][1] = h.actuator_ctrlrange[2 * u + 1];
      m.arm_forcelimited[j] = h.actuator_forcelimited[u];
      m.arm_forcerange[j][0] = h.actuator_forcerange[2 * u]; m.arm_forcerange[j][1] = h.actuator_forcerange[2 * u + 1];
      act_slot[u] = j;
    } else if (h.actuator_trntype[u] == 3) {
      if (!grip || m.grp_has_act) return "tendon actuator without a matching two-finger gripper";
      const int t = h.actuator_trnid[u];
      double coef[2] = {0, 0};
      for (int w = 0; w < h.tendon_num[t]; ++w) {
        const int k = h.tendon_adr[t] + w;
        const int j = h.wrap_objid[k];
        if (j != narm && j != narm + 1) return "gripper tendon must run over the two finger joints";
        coef[j - narm] += h.wrap_prm[k];
      }
      m.grp_has_act = 1;
      m.grp_coef[0] = h.actuator_gear[u] * coef[0];
      m.grp_coef[1] = h.actuator_gear[u] * coef[1];
      m.grp_gain = gp[0];
      std::memcpy(m.grp_bias, bp, 3 * sizeof(double));
      m.grp_biasaffine = h.actuator_biastype[u];
      m.grp_ctrllimited = h.actuator_ctrllimited[u];
      m.grp_ctrlrange[0] = h.actuator_ctrlrange[2 * u]; m.grp_ctrlrange[1] = h.actuator_ctrlrange[2 * u + 1];
      m.grp_forcelimited = h.actuator_forcelimited[u];
      m.grp_forcerange[0] = h.actuator_forcerange[2 * u]; m.grp_forcerange[1] = h.actuator_forcerange[2 * u + 1];
      act_slot[u] = narm;
    } else {
      return "unsupported actuator transmission";
    }
  }
  // equality: finger coupling only
  for (int e = 0; e < h.neq; ++e) {
    if (!grip || h.eq_obj1id[e] != narm || h.eq_obj2id[e] != narm + 1 || m.eq_active)
      return "only one joint equality coupling finger 1 to finger 2 is supported";
    m.eq_active = h.eq_active0[e];
    std::memcpy(m.eq_polycoef, &h.eq_data[5 * e], 5 * sizeof(double));
    m.eq_imp = make_imp(&h.eq_solimp[5 * e]);
    make_kb(&h.eq_solref[2 * e], &h.eq_solimp[5 * e], h.timestep, m.eq_K, m.eq_B);
  }
  // qpos0 constants
  bool ok = dispatch_topology(narm, grip, [&](auto topo) {
    using T = decltype(topo);
    compute_invweight0<T>(m);
  });
  if (!ok) return "arm length / gripper combination is not instantiated";
  // dry joint friction rows (need invweight0): mj_makeImpedance gives friction rows zero stiffness and the impedance
  // at distance 0; R = (1 - imp) / imp * dof_invweight0, floored like every regulariser
  m.has_friction = 0;
  m.pad3 = 0;
  for (int i = 0; i < kMaxLinks; ++i) { m.fl_floss[i] = 0; m.fl_D[i] = 0; m.fl_B[i] = 0; m.fl_R[i] = 0; }
  for (int i = 0; i < nl; ++i) {
    m.fl_floss[i] = h.dof_frictionloss[i];
    if (h.dof_frictionloss[i] <= 0) continue;
    m.has_friction = 1;
    const double imp = impedance(make_imp(&h.dof_solimp[5 * i]), 0.0, 0.0);
    m.fl_D[i] = row_D(imp, m.invweight0[i]);
    m.fl_R[i] = m.fl_floss[i] / m.fl_D[i];
    double K;
    make_kb(&h.dof_solref[2 * i], &h.dof_solimp[5 * i], h.timestep, K, m.fl_B[i]);
  }
  return "";
}

// Plane contacts only: the first static plane geom against every collision geom on a moving body that passes
// MuJoCo's pair filters (contype / conaffinity masks; bodies welded together never collide -- both are trivially
// satisfied or excluded here because the plane is on the world body).
std::string build_collision_points(const HostModel& h, CollisionPoints& out) {
  std::vector<int> owner;
  std::vector<Xf> rel;
  body_owner_frames(h, owner, rel);
  out = CollisionPoints();
  const int nl = h.njnt;
  for (int g = 0; g < h.ngeom; ++g) {
    if (h.geom_type[g] != 0 || owner[h.geom_bodyid[g]] >= 0) continue;
    if (h.geom_contype[g] == 0 && h.geom_conaffinity[g] == 0) continue;
    const Xf t = xf_mul(rel[h.geom_bodyid[g]], xf_from(&h.geom_pos[3 * g], &h.geom_quat[4 * g]));
    out.has_plane = true;
    out.plane_geom = g;
    for (int k = 0; k < 3; ++k) out.plane_n[k] = t.R[3 * k + 2];
    out.plane_d = dot3(out.plane_n, t.p);
    break;
  }
  std::vector<std::vector<double>> pts(nl);
  std::vector<std::vector<int32_t>> ids(nl);
  if (out.has_plane) {
    const int pg = out.plane_geom;
    for (int g = 0; g < h.ngeom; ++g) {
      const int link = owner[h.geom_bodyid[g]];
      if (link < 0) continue;
      const bool mask_ok = (h.geom_contype[g] & h.geom_conaffinity[pg]) || (h.geom_contype[pg] & h.geom_conaffinity[g]);
      if (!mask_ok) continue;
      const Xf t = xf_mul(rel[h.geom_bodyid[g]], xf_from(&h.geom_pos[3 * g], &h.geom_quat[4 * g]));
      auto add = [&](const double* local, double r) {
        double w[3];
        mulmv(t.R, local, w);
        for (int k = 0; k < 3; ++k) pts[link].push_back(w[k] + t.p[k]);
        pts[link].push_back(r);
        ids[link].push_back(g);
      };
      const double* sz = &h.geom_size[3 * g];
      switch (h.geom_type[g]) {
        case 7:  // mesh: hull vertices
          for (int v = 0; v < h.geom_vertnum[g]; ++v) add(&h.mesh_vert[3 * (size_t)(h.geom_vertadr[g] + v)], 0.0);
          break;
        case 6:  // box: corners
          for (int c = 0; c < 8; ++c) {
            const double p[3] = {(c & 1 ? sz[0] : -sz[0]), (c & 2 ? sz[1] : -sz[1]), (c & 4 ? sz[2] : -sz[2])};
            add(p, 0.0);
          }
          break;
        case 3: {  // capsule: segment end points, radius
          const double a[3] = {0, 0, sz[1]}, b[3] = {0, 0, -sz[1]};
          add(a, sz[0]); add(b, sz[0]);
          break;
        }
        case 2: {  // sphere
          const double c[3] = {0, 0, 0};
          add(c, sz[0]);
          break;
        }
        default:
          break;  // other primitives: not in the RCS scenes
      }
    }
  }
  out.link_adr.assign(nl + 1, 0);
  out.link_sphere.assign(4 * (size_t)nl, 0.0);
  out.link_aabb.assign(6 * (size_t)nl, 0.0);
  for (int i = 0; i < nl; ++i) {
    // broad phase: sphere around the centroid of the link's sample points
    const size_t np = ids[i].size();
    if (np) {
      double c[3] = {0, 0, 0}, rad = 0;
      for (size_t k = 0; k < np; ++k) for (int a = 0; a < 3; ++a) c[a] += pts[i][4 * k + a] / (double)np;
      for (size_t k = 0; k < np; ++k) {
        const double dx = pts[i][4 * k] - c[0], dy = pts[i][4 * k + 1] - c[1], dz = pts[i][4 * k + 2] - c[2];
        rad = std::fmax(rad, std::sqrt(dx * dx + dy * dy + dz * dz) + pts[i][4 * k + 3]);
      }
      for (int a = 0; a < 3; ++a) out.link_sphere[4 * i + a] = c[a];
      out.link_sphere[4 * i + 3] = rad;
      for (int a = 0; a < 3; ++a) {
        double lo = HUGE_VAL, hi = -HUGE_VAL;
        for (size_t k = 0; k < np; ++k) { lo = std::fmin(lo, pts[i][4 * k + a] - pts[i][4 * k + 3]); hi = std::fmax(hi, pts[i][4 * k + a] + pts[i][4 * k + 3]); }
        out.link_aabb[6 * i + a] = 0.5 * (lo + hi);
        out.link_aabb[6 * i + 3 + a] = 0.5 * (hi - lo);
      }
    }
    out.link_adr[i + 1] = out.link_adr[i] + (int)ids[i].size();
    out.xyzr.insert(out.xyzr.end(), pts[i].begin(), pts[i].end());
    out.geom.insert(out.geom.end(), ids[i].begin(), ids[i].end());
  }
  return "";
}

namespace {

// mjModel.body_invweight0[.][0] of every body: mean diagonal of J M^-1 J' for the translational Jacobian of the body's
// centre of mass at qpos0 (mj_setConst); 0 for static bodies
template <class T>
void body_invweight0(const HostModel& h, const DevModel& m, const std::vector<int>& owner, const std::vector<Xf>& rel, std::vector<double>& out) {
  constexpr int NL = T::NL;
  double q[NL], qd[NL];
  for (int i = 0; i < NL; ++i) { q[i] = m.qpos0[i]; qd[i] = 0; }
  Smooth<T> sm;
  double buf[Stage<T, 1>::COUNT];
  Stage<T, 1> st{buf};
  smooth_dynamics<T, 1>(m, q, qd, st, sm);
  double* M = &st.M(0);
  ldl_factor<NL>(M);
  // link frames, joint axes and anchors in the world at qpos0
  Xf F[NL];
  double axis[NL][3], anchor[NL][3];
  for (int i = 0; i < NL; ++i) {
    const int par = T::parent(i);
    Xf f = par >= 0 ? F[par] : xf_identity();
    advance_link_frame(m, i, q[i], f.R, f.p);
    F[i] = f;
    mulmv(f.R, m.axis[i], axis[i]);
    double a[3];
    mulmv(f.R, m.jpos[i], a);
    for (int k = 0; k < 3; ++k) anchor[i][k] = a[k] + f.p[k];
  }
  auto is_anc = [](int j, int i) {
    if (T::GRIP && i == T::NARM + 1 && j == T::NARM) return false;
    return j <= i;
  };
  out.assign(h.nbody, 0.0);
  for (int b = 1; b < h.nbody; ++b) {
    const int link = owner[b];
    if (link < 0) continue;
    const Xf bw = xf_mul(F[link], rel[b]);
    double c[3];
    mulmv(bw.R, &h.body_ipos[3 * b], c);
    for (int k = 0; k < 3; ++k) c[k] += bw.p[k];
    double tr = 0;
    for (int k = 0; k < 3; ++k) {
      double J[NL], y[NL];
      for (int j = 0; j < NL; ++j) {
        J[j] = 0;
        if (!is_anc(j, link)) continue;
        if (m.jtype[j] == kSlide) J[j] = axis[j][k];
        else {
          const double r[3] = {c[0] - anchor[j][0], c[1] - anchor[j][1], c[2] - anchor[j][2]};
          double col[3];
          cross3(axis[j], r, col);
          J[j] = col[k];
        }
        y[j] = 0;
      }
      for (int j = 0; j < NL; ++j) y[j] = J[j];
      ldl_solve<NL>(M, y);
      for (int j = 0; j < NL; ++j) tr += J[j] * y[j];
    }
    out[b] = tr / 3;
  }
}

}  // namespace

std::string build_contact_table(const HostModel& h, const DevModel& m, int plane_geom, std::vector<ContactGeom>& geoms, std::vector<double>& verts,
                                std::string& overflow, std::vector<int>* dropped) {
  overflow.clear();
  if (dropped) dropped->clear();
  auto drop = [&](int g, const std::string& why) {
    if (overflow.empty()) overflow = why;
    if (dropped) dropped->push_back(g);
  };
  std::vector<int> owner;
  std::vector<Xf> rel;
  body_owner_frames(h, owner, rel);
  std::vector<double> iw;
  const bool ok = dispatch_topology(m.narm, m.has_gripper != 0, [&](auto topo) { body_invweight0<decltype(topo)>(h, m, owner, rel, iw); });
  if (!ok) return "no archetype for the contact table";
  geoms.clear();
  verts = h.mesh_vert;
  for (int g = 0; g < h.ngeom; ++g) {
    const int type = h.geom_type[g];
    if (type == 0) continue;
    if (type != 3 && type != 6 && type != 7) continue;  // capsule, box, mesh: what the RCS scenes use
    if (h.geom_contype[g] == 0 && h.geom_conaffinity[g] == 0) continue;
    // Capacity limits: a geom beyond them stays OUT of the table (geom-geom detection does not see it; the floor test by
    // sample points still does) and the reason is kept -- fatal only once contacts are to be RESOLVED for this scene.
    if ((int)geoms.size() >= kMaxCGeom) { drop(g, "too many collision geoms for the contact phase (" + std::to_string(kMaxCGeom) + ")"); continue; }
    ContactGeom cg;
    std::memset(&cg, 0, sizeof(cg));
    const int b = h.geom_bodyid[g];
    cg.link = owner[b];
    cg.type = type;
    cg.geom_id = g;
    cg.body = b;
    if (type == 6) {
      int nbox = 0;
      for (const auto& o : geoms) nbox += o.type == 6;
      if (nbox >= 10) { drop(g, "too many box geoms for the contact phase (10)"); continue; }
      cg.box_slot = nbox;
    }
    cg.vert_adr = h.geom_vertadr[g];
    cg.vert_num = type == 7 ? h.geom_vertnum[g] : 0;
    if (cg.vert_num > kHullMaxVerts) { drop(g, "collision hull with more than 152 vertices (the self-collision test stages two hulls in LDS)"); continue; }
    const Xf t = xf_mul(rel[b], xf_from(&h.geom_pos[3 * g], &h.geom_quat[4 * g]));
    for (int k = 0; k < 3; ++k) { cg.pos[k] = t.p[k]; cg.size[k] = h.geom_size[3 * g + k]; }
    for (int k = 0; k < 9; ++k) cg.rot[k] = t.R[k];
    cg.mu = h.geom_friction[3 * g];
    cg.invweight = iw[b];
    if (plane_geom >= 0 && cg.link >= 0)
      cg.plane_ok = ((h.geom_contype[g] & h.geom_conaffinity[plane_geom]) || (h.geom_contype[plane_geom] & h.geom_conaffinity[g])) ? 1 : 0;
    const double* sz = cg.size;
    if (type == 7) {
      double lo[3] = {HUGE_VAL, HUGE_VAL, HUGE_VAL}, hi[3] = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL}, c[3] = {0, 0, 0}, rb = 0;
      for (int v = 0; v < cg.vert_num; ++v) {
        const double* w = &h.mesh_vert[3 * (size_t)(cg.vert_adr + v)];
        for (int k = 0; k < 3; ++k) { lo[k] = std::fmin(lo[k], w[k]); hi[k] = std::fmax(hi[k], w[k]); c[k] += w[k]; }
        rb = std::fmax(rb, w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
      }
      for (int k = 0; k < 3; ++k) {
        cg.center[k] = cg.vert_num ? c[k] / cg.vert_num : 0.0;
        cg.aabb_c[k] = cg.vert_num ? 0.5 * (lo[k] + hi[k]) : 0.0;
        cg.aabb_h[k] = cg.vert_num ? 0.5 * (hi[k] - lo[k]) : 0.0;
      }
      cg.rbound = std::sqrt(rb);
    } else if (type == 6) cg.rbound = std::sqrt(sz[0] * sz[0] + sz[1] * sz[1] + sz[2] * sz[2]);
    else cg.rbound = sz[0] + sz[1];
    geoms.push_back(cg);
  }
  return "";
}

bool build_hull_edges(const double* planes, int np, std::vector<HullEdge>& edges, double centre[3]) {
  edges.clear();
  centre[0] = centre[1] = centre[2] = 0;
  if (np < 4) return false;
  constexpr double kOn = 1e-8, kMerge = 1e-7;
  // vertices: intersection points of three planes that violate no other plane
  std::vector<double> V;
  for (int i = 0; i < np; ++i)
    for (int j = i + 1; j < np; ++j) {
      const double* a = planes + 4 * i;
      const double* b = planes + 4 * j;
      const double ab[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
      if (ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2] < 1e-12) continue;  // parallel
      for (int k = j + 1; k < np; ++k) {
        const double* c = planes + 4 * k;
        const double det = ab[0] * c[0] + ab[1] * c[1] + ab[2] * c[2];
        if (std::fabs(det) < 1e-9) continue;
        // x = (d_a (b x c) + d_b (c x a) + d_c (a x b)) / det
        const double bc[3] = {b[1] * c[2] - b[2] * c[1], b[2] * c[0] - b[0] * c[2], b[0] * c[1] - b[1] * c[0]};
        const double ca[3] = {c[1] * a[2] - c[2] * a[1], c[2] * a[0] - c[0] * a[2], c[0] * a[1] - c[1] * a[0]};
        double x[3];
        for (int t = 0; t < 3; ++t) x[t] = (a[3] * bc[t] + b[3] * ca[t] + c[3] * ab[t]) / det;
        bool inside = true;
        for (int m = 0; m < np && inside; ++m) {
          const double* q = planes + 4 * m;
          inside = q[0] * x[0] + q[1] * x[1] + q[2] * x[2] - q[3] <= kOn;
        }
        if (!inside) continue;
        bool seen = false;
        for (size_t v = 0; v < V.size() && !seen; v += 3) {
          const double d[3] = {V[v] - x[0], V[v + 1] - x[1], V[v + 2] - x[2]};
          seen = d[0] * d[0] + d[1] * d[1] + d[2] * d[2] < kMerge * kMerge;
        }
        if (!seen) V.insert(V.end(), x, x + 3);
      }
    }
  const int nv = (int)V.size() / 3;
  if (nv < 4) return false;
  // which vertices lie on which plane; a plane with fewer than three of them is no face
  std::vector<std::vector<int>> on(np);
  for (int i = 0; i < np; ++i) {
    const double* q = planes + 4 * i;
    for (int v = 0; v < nv; ++v)
      if (std::fabs(q[0] * V[3 * v] + q[1] * V[3 * v + 1] + q[2] * V[3 * v + 2] - q[3]) < kOn) on[i].push_back(v);
  }
  int nf = 0;
  for (int i = 0; i < np; ++i) nf += on[i].size() >= 3;
  for (int i = 0; i < np; ++i) {
    if (on[i].size() < 3) continue;
    for (int j = i + 1; j < np; ++j) {
      if (on[j].size() < 3) continue;
      // the vertices both faces share (sorted lists): two or more -- an edge, from the first to the last of them along it
      const double* a = planes + 4 * i;
      const double* b = planes + 4 * j;
      const double dir[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
      int lo = -1, hi = -1, cnt = 0;
      double slo = 0, shi = 0;
      size_t p = 0, q = 0;
      while (p < on[i].size() && q < on[j].size()) {
        if (on[i][p] < on[j][q]) ++p;
        else if (on[i][p] > on[j][q]) ++q;
        else {
          const int v = on[i][p];
          const double sv = dir[0] * V[3 * v] + dir[1] * V[3 * v + 1] + dir[2] * V[3 * v + 2];
          if (cnt == 0 || sv < slo) { slo = sv; lo = v; }
          if (cnt == 0 || sv > shi) { shi = sv; hi = v; }
          ++cnt; ++p; ++q;
        }
      }
      if (cnt < 2 || lo == hi) continue;
      HullEdge e;
      e.a = i; e.b = j;
      for (int t = 0; t < 3; ++t) { e.v1[t] = V[3 * lo + t]; e.v2[t] = V[3 * hi + t]; }
      edges.push_back(e);
    }
  }
  if (nv - (int)edges.size() + nf != 2) { edges.clear(); return false; }
  for (int v = 0; v < nv; ++v)
    for (int t = 0; t < 3; ++t) centre[t] += V[3 * v + t] / nv;
  return true;
}

std::string attach_robot_frames(const HostModel& h, DevModel& m, int site, int base_body) {
  std::vector<int> owner;
  std::vector<Xf> rel;
  body_owner_frames(h, owner, rel);
  if (site < 0 || site >= h.nsite) return "attachment site id out of range";
  if (base_body < 0 || base_body >= h.nbody) return "base body id out of range";
  const int sb = h.site_bodyid[site];
  if (owner[sb] < 0) return "attachment site must ride on a moving link";
  const Xf ts = xf_mul(rel[sb], xf_from(&h.site_pos[3 * site], &h.site_quat[4 * site]));
  m.site_link = owner[sb];
  std::memcpy(m.site_pos, ts.p, sizeof(ts.p));
  std::memcpy(m.site_rot, ts.R, sizeof(ts.R));
  if (owner[base_body] >= 0) return "robot base body must be static";
  std::memcpy(m.base_pos, rel[base_body].p, sizeof(m.base_pos));
  // MuJoCo keeps xquat as the product of body quaternions; for a static body that is the quaternion
  // of the accumulated rotation (sign chosen by the conversion below, w >= 0 for the RCS scenes)
  mat_to_quat_wxyz(rel[base_body].R, m.base_quat);
  return "";
}

}  // namespace rcsh
