// model_host.h -- host-side copy of the scene tables and the finalisation entry points.
#pragma once
#include <string>
#include <vector>

#include "../../include/rcs_hip.h"
#include "contact_types.h"
#include "dyn.h"
#include "model.h"

namespace rcsh {

// Deep copy of rcsh_model_desc (the caller's arrays need not outlive rcsh_sim_create).
struct HostModel {
  int nbody = 0, njnt = 0, nu = 0, ntendon = 0, nwrap = 0, neq = 0, nsite = 0;
  double timestep = 0.002;
  double gravity[3] = {0, 0, -9.81};
  std::vector<int32_t> body_parentid, body_jntadr, body_jntnum;
  std::vector<double> body_pos, body_quat, body_ipos, body_iquat, body_mass, body_inertia, body_gravcomp;
  std::vector<int32_t> jnt_type, jnt_bodyid, jnt_limited, jnt_actfrclimited, jnt_actgravcomp;
  std::vector<double> jnt_pos, jnt_axis, jnt_range, jnt_margin, jnt_solref, jnt_solimp, jnt_actfrcrange;
  std::vector<double> dof_armature, dof_damping, dof_frictionloss, dof_solref, dof_solimp, qpos0;
  std::vector<int32_t> tendon_adr, tendon_num, wrap_objid;
  std::vector<double> wrap_prm;
  std::vector<int32_t> eq_obj1id, eq_obj2id, eq_active0;
  std::vector<double> eq_data, eq_solref, eq_solimp;
  std::vector<int32_t> actuator_trntype, actuator_trnid, actuator_biastype, actuator_ctrllimited, actuator_forcelimited;
  std::vector<double> actuator_gear, actuator_gainprm, actuator_biasprm, actuator_ctrlrange, actuator_forcerange;
  std::vector<int32_t> site_bodyid;
  std::vector<double> site_pos, site_quat;
  int ngeom = 0, nmeshvert = 0;
  std::vector<int32_t> geom_type, geom_bodyid, geom_contype, geom_conaffinity, geom_vertadr, geom_vertnum;
  std::vector<double> geom_pos, geom_quat, geom_size, geom_friction, mesh_vert;
  void copy_from(const rcsh_model_desc& d);
};

// Contact detection tables: sample points (hull vertices, box corners, capsule / sphere centres with a radius) of
// every collision geom riding on a moving link, in link coordinates, tested against the scene's static plane.
struct CollisionPoints {
  std::vector<double> xyzr;       // [npts][4]
  std::vector<int32_t> geom;      // [npts] geom id the point belongs to
  std::vector<int32_t> link_adr;  // [nl + 1] points of link i are [link_adr[i], link_adr[i+1])
  std::vector<double> link_sphere;  // [nl][4] bounding sphere (centre, radius) of link i's points, link frame
  std::vector<double> link_aabb;    // [nl][6] bounding box of link i's points (radii included), link frame: centre, half extents
  bool has_plane = false;
  int plane_geom = -1;
  double plane_n[3] = {0, 0, 1};
  double plane_d = 0;             // plane: n . x = d
};
std::string build_collision_points(const HostModel& h, CollisionPoints& out);

// The robot's collision geoms for the contact phase (contact_team.h), in geom order; `verts` are the hull vertices they
// index (geom frame).  Class bits are filled in later from the SimRobot / SimGripper configurations.
std::string build_contact_table(const HostModel& h, const DevModel& m, int plane_geom, std::vector<ContactGeom>& geoms, std::vector<double>& verts,
                                std::string& overflow /* geoms left out of the table because of a capacity limit: why (empty: none) */,
                                std::vector<int>* dropped = nullptr /* their mjModel geom ids */);

// Edges of the convex polytope { x : n_i . x <= d_i } a drawn hull is given as (render.h: the ray caster finds a hull's outline
// as seen from the camera among them).  Each edge: the two planes that meet in it and its end points.  `centre`: a point
// inside (the mean of the polytope's vertices).  false: the planes do not bound a polytope this routine can vouch for
// (fewer than 4 faces, unbounded, or Euler's formula V - E + F = 2 fails on what it found) -- the caller draws the hull
// without the outline then.
struct HullEdge {
  int32_t a, b;
  double v1[3], v2[3];
};
bool build_hull_edges(const double* planes, int nplanes, std::vector<HullEdge>& edges, double centre[3]);

// Calls fn(Topo<NARM, GRIP>{}) for the compiled archetype matching (narm, grip); false if none does.
template <class F>
bool dispatch_topology(int narm, bool grip, F&& fn) {
#ifdef RCSH_DEV_ONLY_XARM7  // (development builds: the 7-dof arm without gripper alone)
  if (narm == 7 && !grip) { fn(Topo<7, false>{}); return true; }
  return false;
#endif
  if (narm == 7 && grip) { fn(Topo<7, true>{}); return true; }
#ifndef RCSH_DEV_ONLY_FR3  // (development builds define it: the FR3 + hand archetype alone, minutes -> seconds)
  if (narm == 7 && !grip) { fn(Topo<7, false>{}); return true; }
  if (narm == 6 && !grip) { fn(Topo<6, false>{}); return true; }
  if (narm == 5 && grip) { fn(Topo<5, true>{}); return true; }
#endif
  return false;
}

// solimp / solref in the kernels' form (mj_makeImpedance's clamps; time constant floored at 2 timesteps)
Imp make_imp(const double* solimp);
void make_kb(const double* solref, const double* solimp, double timestep, double& K, double& B);

// Returns "" on success, else the reason the scene is rejected.  act_slot[u] = ctrl slot of mj actuator u.
std::string finalize_model(const HostModel& h, DevModel& m, std::vector<int>& act_slot);
std::string attach_robot_frames(const HostModel& h, DevModel& m, int site, int base_body);

}  // namespace rcsh
