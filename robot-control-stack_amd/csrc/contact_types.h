// contact_types.h -- host-prepared tables of the contact phase (contact_team.h); plain data, shared with the host code.
#pragma once
#include <cstdint>

namespace rcsh {

constexpr int kHullMaxVerts = 152;  // a collision hull the contact table admits has at most this many vertices (model.cpp: build_contact_table)
constexpr int kMaxCon = 48;      // contacts per environment, scenes with a free box (its kernel's LDS must let four workgroups share a CU)
constexpr int kMaxConNoBox = 64; // ... without one (the contact-resolving kernel of per-environment escalation: a closed gripper pressed into the
                                 // arm brings 50-64 contacts, tools/oracle_ncon_probe.py; the contact phase gives a lane to each).  MuJoCo's list has
                                 // no bound: a phase that runs out sets kContactOverflow
constexpr int kMaxCGeom = 28;    // collision geoms of the robot (the end-of-launch contact check keeps the world boxes of all of them, for the
                                 // wavefront's four environments, in the team kernels' LDS block: check_team.h)
constexpr int kMaxActive = 5;    // links in contact at once that the noslip pass keeps M^-1 S' for
constexpr int kMaxPairs = 4;     // pairs of LINKS in contact with each other at once (self contact: a stiffness accumulator per pair)

// one collision geom of the robot, host-prepared (model.cpp: build_contact_table), in MuJoCo's geom order
struct ContactGeom {
  int32_t link;          // link carrying the geom (-1: welded to the world)
  int32_t type;          // mjtGeom: 3 capsule, 6 box, 7 mesh (convex hull)
  int32_t cls;           // bit 0: SimRobot arm collision geom, bit 1: SimGripper collision geom (not ignored), bit 2: finger geom,
                         // bit 3: in SimGripper's ignore list, bit 4: SimGripper collision geom (ignored or not)
  int32_t vert_adr, vert_num;
  int32_t geom_id;       // mjModel geom id
  int32_t plane_ok;      // the (floor, geom) pair passes MuJoCo's filters
  int32_t box_slot;      // box geoms: index among the box geoms (LDS slot of the collider's clipping polygons)
  int32_t body, pad0;    // mjModel body id of the geom (mjData.contact is ordered by body pair, then by geom)
  double pos[3], rot[9]; // geom frame in the link frame
  double size[3];
  double center[3];      // hull: an interior point (vertex mean), geom frame
  double aabb_c[3], aabb_h[3];  // hull: bounding box in the geom frame (centre, half extents)
  double rbound;
  double mu;             // geom_friction[0]
  double invweight;      // body_invweight0 (translational) of the geom's body
};

// A pair of the robot's collision geoms that MuJoCo's filters let collide (different weld bodies, no parent-child pair
// unless one is welded to the world) and whose contact one of the collision callbacks would react to.  g0 / g1 index
// ContactTable::geoms, in MuJoCo's order within a contact (by geom type, then by id).
struct SelfPair {
  int16_t g0, g1;
  int16_t l0, l1;  // their links (-1: welded to the world)
  int32_t cls;     // bit 0: SimRobot::collision_callback counts the contact, bit 1: SimGripper::collision_callback does
  int32_t joints;  // bit j: joint j lies on the tree path between the two links (only those joints move the geoms relative to each other)
  // broad phase, in the frames of the two links: centre of the geom's bounding box and its half diagonal (bounding sphere);
  // then the box itself -- axes (columns of rot: geom frame in the link frame) and half extents
  double c0[3], r0, c1[3], r1;
  double rot0[9], h0[3], rot1[9], h1[3];
};

// The once-per-launch check for contacts nobody resolves (check_team.h): EVERY geom pair MuJoCo's filters let collide -- whether a
// collision callback reacts to it or not.  Three levels: bounding spheres of the two geoms, their oriented bounding boxes (for two
// box geoms that is the exact test already), MPR.  A lane takes every 16th pair; all it needs per pair is an 8-byte entry.
constexpr int kMaxCheckPairs = 192;  // pairs whose entries a lane keeps in registers (12 each); a scene with more has the rest unchecked (refused at set-up)
struct CheckEntry {
  uint32_t geoms;  // g0 | g1 << 8 (indices into ContactTable::geoms, in MuJoCo's order within a contact: by type, then by id) | (1 + common ancestor link) << 16
  float rsum;      // sum of the two bounding-sphere radii, rounded up
};
// oriented bounding box of a collision geom in the frame of its link (world frame: welded to the world); its axes are the geom's
struct CheckGeom {
  double c[3], rot[9];
};
constexpr int kSlackFloor = kMaxCheckPairs + 24;  // ... then the geoms' remaining heights above the floor
constexpr int kSlackLink = kSlackFloor + 32;      // ... then, per LINK, what is left of its sample points' height above the floor (the lean launch's check)
constexpr int kSlackStride = kSlackLink + 16;     // floats per environment: the pairs' remaining gaps, then the joints seen last (12 doubles), ...
static_assert(kMaxCGeom <= 32, "a float per collision geom");
constexpr int kLevGeom = 144;  // CheckTable::lev: where the per-geom levers begin
struct CheckTable {
  const CheckEntry* ent;
  const CheckGeom* geoms;
  const float* lev; // [12][12] lev[j][l]: how far one radian (hinge) / metre (slide) of joint j moves a point of a geom ON link l (host: build_self_levers);
                    // behind it (kLevGeom) [12][32]: the same for the points of geom g alone
  float* slack;    // [n][kSlackStride] self-contact stage of the contact phase (contact_team.h: contact_collide); null: every pair, every substep
  int32_t npair, ngeom;
  int32_t plane_points;  // the scene has a floor plane and collision geoms with sample points to test against it
  int32_t pad;
  double gh[kMaxCGeom][3];     // half extents of the geoms' boxes
  int32_t gvert[kMaxCGeom][2]; // hull geoms: first vertex, number of vertices (ContactTable::verts)
  int8_t glink[kMaxCGeom];     // the geom's link (-1: welded to the world)
  int8_t pad2[4];
  int8_t gtype[kMaxCGeom];     // mjtGeom of the geom (ContactGeom::type)
  int8_t pad3[4];
};

struct ContactTable {
  const ContactGeom* geoms;
  const double* verts;   // [nvert][3] hull vertices, geom frame
  const SelfPair* pairs; // self collision (flags only): see self_collision_pairs, contact_team.h
  int32_t ngeom, has_plane;
  int32_t link_geom_adr[13];  // geoms of link i are [link_geom_adr[i], link_geom_adr[i + 1]) (kMaxLinks + 1 entries)
  int32_t npair;
  // self collision: lever[j] bounds how far one radian (hinge) / one metre (slide) of joint j can move any point of a collision
  // geom downstream of it (host: build_self_pairs) -- what turns joint motion into a bound on how much a pair's gap can close
  double self_lever[12];
  double plane_n[3], plane_d, plane_mu;
};

}  // namespace rcsh
