// rcs_hip.hip -- C-ABI (include/rcs_hip.h) over the batched kernels.  gfx950 only.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rcs_hip.h"
#include "model_host.h"
#include "sim_kernels.h"
#include "render.h"

using namespace rcsh;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess)                                                                          \
      return fail(RCSH_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));             \
  } while (0)

constexpr int kBlock = 64;     // accessor kernels
constexpr int kProfRing = 4096;

}  // namespace

struct CopyCarrier;
struct rcsh_sim {
  int device = 0;
  int kernel = RCSH_KERNEL_AUTO;
  int n_simd = 1024;  // SIMDs of the device (4 per compute unit)
  hipStream_t stream = nullptr;
  hipStream_t own_stream = nullptr;
  int n = 0;
  HostModel hm;
  DevModel dm;
  DevModel* d_model = nullptr;     // DevModel followed by LinkRec[kMaxLinks]
  std::vector<LinkRec> links;
  CollisionPoints cp;
  std::vector<uint8_t> cp_class;
  double* d_coll_xyzr = nullptr;
  uint8_t* d_coll_cls = nullptr;
  // contact phase (contact_team.h): the robot's collision geoms and their hull vertices
  std::vector<ContactGeom> cgeoms;
  std::string contact_overflow;  // why collision geoms were left out of `cgeoms` (capacity); empty: none were
  std::vector<int> cgeoms_dropped;  // mjModel ids of those geoms: invisible to geom-geom detection (the floor test by sample points still sees them)
  std::vector<double> cverts;
  ContactGeom* d_cgeoms = nullptr;
  SelfPair* d_pairs = nullptr;       // self-collision pairs a collision callback reacts to (rebuilt when the class bits change)
  std::vector<SelfPair> pairs;
  double self_lever[12] = {0};       // contact_types.h: ContactTable::self_lever
  double* d_cverts = nullptr;
  // the once-per-launch check for contacts nobody resolves (check_team.h): every geom pair MuJoCo's filters let collide, by body pair
  std::vector<SelfPair> chk_pairs;
  std::vector<CheckEntry> chk_ent;
  std::vector<CheckGeom> chk_geoms;
  CheckGeom* d_chk_geoms = nullptr;
  CheckEntry* d_chk_ent = nullptr;
  float link_lever[12 * 12 + 12 * 32] = {0};  // contact_types.h: CheckTable::lev ([joint][link]), then the same per GEOM ([joint][geom], kLevGeom)
  float* d_lev = nullptr;
  float* d_slack = nullptr;          // [n][kSlackStride]: the self-contact stage's remaining gaps per pair + the joints it saw last (CheckTable::slack)
  int chk_unchecked = 0;             // admitted geom pairs past kMaxCheckPairs: neither checked at the end of a launch nor resolved as self contacts
  // per-environment escalation (sim_kernels.h: RunOp::esc_role): a step is the lean launch over the environments not in contact plus
  // the contact-resolving launch over the others
  bool esc_mode = false;
  uint64_t* d_esc = nullptr;       // [3][(n + 63) / 64]: escalated, newly flagged, leaving
  uint32_t* d_esc_ctr = nullptr;   // [4] (sim_kernels.h: RunOp::esc_ctr)
  // the contact-resolving launch in two parts (RunOp::esc_part): the already escalated environments next to the lean launch, on a stream
  // of their own; chosen per step from what the device last reported (h_esc_hint: the lean launch's step number and the escalated count)
  hipStream_t esc_stream = nullptr;
  hipEvent_t esc_ev_prev = nullptr, esc_ev_old = nullptr, esc_ev_started = nullptr;
  int esc_split_max = 128;          // split while at most this many environments are escalated (RCSH_ESC_SPLIT_MAX)
  volatile uint32_t* h_esc_hint = nullptr;  // [2] host memory the lean launch writes (RunOp::esc_host)
  uint32_t esc_seq = 0;
  int conv_grow = 0;                // ... growing (RCSH_CONV_GROW=1)
  int conv_chunk = 48;              // step_until_convergence in pieces of this many substeps when contacts are resolved per environment (launch_run; RCSH_CONV_CHUNK)
  int esc_split = 0;                // 1: split (RCSH_ESC_SPLIT=1).  OFF by default -- measured (profiles/r5_v2/split_ab.txt): with the escalated
                                    // environments' launch dispatched first and the lean launch beside it a 300-step rollout takes 0.496 ms a step
                                    // against 0.433 one launch after the other, the 1000-step rollout 3.96 against 3.87: two kernels of different
                                    // scratch / LDS footprints from two queues do not share the chip the way the arithmetic (max instead of sum) promises
  double* d_snap = nullptr;        // [nfields][n]: what the lean launch of a step read (the step is redone from it on a hit)
  uint32_t* d_snap_flags = nullptr;
  int32_t* d_snap_conv = nullptr;
  bool contact_check = true;  // RCSH_CONTACT_CHECK=0 switches the check off (measurements of its cost)
  int check_every = 1;        // the check ends every check_every-th stepping launch (rcsh_sim_set_contact_check); 0: never
  int64_t check_seq = 0;
  double plane_mu = 1.0;
  std::vector<int> act_slot;
  int narm = 0, nl = 0, nu = 0;
  bool grip = false;
  int nfields = 0;
  double* S = nullptr;
  uint32_t* flags = nullptr;
  int32_t* conv = nullptr;
  SimCfg sim{0, 0, 30, 500};
  RobotCfg robot{};
  GripperCfg gripcfg{};
  EnvCfg env{};
  BoxCfg box{};
  TaskCfg task{};
  bool env_configured = false;
  BoxTaskCfg* d_boxtask = nullptr;
  // depth renderer (render.h)
  RenderScene rscene{};
  RenderShape* d_rshapes = nullptr;
  double* d_rplanes = nullptr;
  RenderColour* d_rcolours = nullptr;
  int32_t* d_redge_planes = nullptr;  // the outline method of the ray caster (render.h: k_hull_views)
  double* d_redge_verts = nullptr;
  double* d_rviews = nullptr;
  RendCfg rend{};                    // rate-driven cameras (rcsh_sim_set_render_schedule)
  int rend_cam_id[kMaxRateCams] = {0, 0, 0, 0};
  int64_t rend_dropped = 0;          // records that did not fit the schedule's capacity (rcsh_render_pending counts them)
  const double* frames_src = nullptr; // set while a record of the render schedule is being rendered
  const double* frames_src_base(int slot) const { return rend.snap + (size_t)slot * (size_t)(nl + 9) * (size_t)n; }
  double* d_frames = nullptr;
  double* d_wframes = nullptr;  // world frames of the shapes + camera per environment (k_shape_frames)
  bool render_f64 = false;      // the ray caster's arithmetic type (rcsh_sim_set_render_f64; RCSH_RENDER_F64=1 at creation)
  std::vector<RenderCam> cams;
  void* d_image = nullptr;  // staging for the host-pointer render call
  size_t image_cap = 0;  // device copy of {box, task} (scenes with a free box)
  double* pending_task = nullptr;  // task output of the env-step being enqueued (rcsh_env_step_task*)
  // staging for the host-pointer entry points
  double* d_stage = nullptr;   // n * kStageWidth doubles
  double* d_stage2 = nullptr;  // n * 32 doubles
  // host-buffer env.step / env.reset: page-locked memory between the caller's arrays and the device (a copy to or from pageable memory is
  // staged by the runtime and waited for, one array at a time: 0.19 of the call's 0.32 ms)
  char* h_pin = nullptr;
  size_t h_pin_bytes = 0;
  uint8_t* d_bytes = nullptr;  // n * 16 bytes
  uint8_t* d_mask = nullptr;   // n bytes
  int32_t* d_ints = nullptr;   // n ints
  float* d_floats = nullptr;   // n floats
  // multi-GPU exchange (RCCL, loaded on first use)
  void* comm = nullptr;            // ncclComm_t
  hipStream_t comm_stream = nullptr;
  hipEvent_t comm_ready = nullptr, comm_done[2] = {nullptr, nullptr};
  bool comm_pending[2] = {false, false};
  int comm_rank = 0, comm_world = 1;
  struct CopyCarrier* copy = nullptr;  // the all-gather's second carrier: copy engines + flags in peer memory (rcsh_comm_copy_*)
  // profiling
  hipEvent_t order_ev = nullptr;       // rcsh_sim_wait_for: marks this handle's stream for another handle's stream to wait on
  bool prof = false;
  bool prof_region = false;            // one event pair around the whole timed region instead of sampled launches
  int64_t prof_region_launches = 0;
  std::vector<hipEvent_t> ev_start, ev_stop;
  int prof_pending = 0;
  int prof_every = 1;       // HIP events around every prof_every-th stepping launch
  int64_t prof_seen = 0;
  double prof_ms = 0;
  int64_t prof_launches = 0;
};

namespace {

int grid_for(int n) { return (n + kBlock - 1) / kBlock; }

Params make_params(rcsh_sim* s) {
  Params P;
  P.model = s->d_model;
  P.coll.xyzr = s->d_coll_xyzr;
  P.coll.cls = s->d_coll_cls;
  for (int i = 0; i <= kMaxLinks; ++i) P.coll.link_adr[i] = i < (int)s->cp.link_adr.size() ? s->cp.link_adr[i] : (s->cp.link_adr.empty() ? 0 : s->cp.link_adr.back());
  for (int i = 0; i < kMaxLinks; ++i)
    for (int a = 0; a < 4; ++a) P.coll.link_sphere[i][a] = (size_t)(4 * i + a) < s->cp.link_sphere.size() ? s->cp.link_sphere[4 * i + a] : 0.0;
  for (int i = 0; i < kMaxLinks; ++i)
    for (int a = 0; a < 6; ++a) P.coll.link_aabb[i][a] = (size_t)(6 * i + a) < s->cp.link_aabb.size() ? s->cp.link_aabb[6 * i + a] : 0.0;
  {
    // entry nl: the box (world frame) around the robot's collision geoms that are welded to the world (link 0's hull): the
    // broad phase of the contact phase tests the free body against it on the first lane that carries no link
    double lo[3] = {HUGE_VAL, HUGE_VAL, HUGE_VAL}, hi[3] = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL};
    bool any = false;
    for (const auto& g : s->cgeoms) {
      if (g.link >= 0 || (g.type == 7 && g.vert_num == 0)) continue;
      double lc[3] = {0, 0, 0}, h[3] = {g.size[0], g.size[1], g.size[2]};
      if (g.type == 7) for (int k = 0; k < 3; ++k) { lc[k] = g.aabb_c[k]; h[k] = g.aabb_h[k]; }
      else if (g.type == 3) { h[0] = h[1] = g.size[0]; h[2] = g.size[0] + g.size[1]; }
      for (int a = 0; a < 3; ++a) {
        const double c = g.rot[3 * a] * lc[0] + g.rot[3 * a + 1] * lc[1] + g.rot[3 * a + 2] * lc[2] + g.pos[a];
        const double e = std::fabs(g.rot[3 * a]) * h[0] + std::fabs(g.rot[3 * a + 1]) * h[1] + std::fabs(g.rot[3 * a + 2]) * h[2];
        lo[a] = std::fmin(lo[a], c - e); hi[a] = std::fmax(hi[a], c + e);
      }
      any = true;
    }
    if (any && s->nl < kMaxLinks)
      for (int a = 0; a < 3; ++a) { P.coll.link_aabb[s->nl][a] = 0.5 * (lo[a] + hi[a]); P.coll.link_aabb[s->nl][3 + a] = 0.5 * (hi[a] - lo[a]); }
    P.coll.has_static = any ? 1 : 0;
  }
  P.coll.has_plane = s->cp.has_plane && !s->cp.geom.empty();
  for (int k = 0; k < 3; ++k) P.coll.plane_n[k] = s->cp.plane_n[k];
  P.coll.plane_d = s->cp.plane_d;
  P.S = s->S;
  P.flags = s->flags;
  P.conv_steps = s->conv;
  P.n = s->n;
  P.keep_qpre = s->d_frames != nullptr;
  P.sim = s->sim;
  P.robot = s->robot;
  P.grip = s->gripcfg;
  P.env = s->env;
  P.boxtask = s->d_boxtask;
  P.ctab.geoms = s->d_cgeoms;
  P.ctab.verts = s->d_cverts;
  P.rend = s->rend;
  P.ctab.pairs = s->d_pairs;
  for (int k = 0; k < 12; ++k) P.ctab.self_lever[k] = s->self_lever[k];
  P.ctab.npair = (int)s->pairs.size();
  P.ctab.ngeom = s->box.resolve ? (int)s->cgeoms.size() : 0;
  P.ctab.has_plane = s->cp.has_plane;
  {
    // geoms of a link are contiguous in the table (geom order = body order); geoms welded to the world sit in front
    static_assert(kMaxLinks + 1 == 13, "ContactTable::link_geom_adr");
    for (int i = 0; i <= kMaxLinks; ++i) P.ctab.link_geom_adr[i] = 0;
    int idx = 0;
    const int ng = (int)s->cgeoms.size();
    while (idx < ng && s->cgeoms[idx].link < 0) ++idx;
    for (int i = 0; i < kMaxLinks; ++i) {
      P.ctab.link_geom_adr[i] = idx;
      while (idx < ng && s->cgeoms[idx].link == i) ++idx;
    }
    P.ctab.link_geom_adr[kMaxLinks] = idx;
  }
  for (int k = 0; k < 3; ++k) P.ctab.plane_n[k] = s->cp.plane_n[k];
  P.ctab.plane_d = s->cp.plane_d;
  P.ctab.plane_mu = s->plane_mu;
  P.chk.ent = s->d_chk_ent;
  P.chk.lev = s->d_lev;
  P.chk.geoms = s->d_chk_geoms;
  P.chk.slack = s->d_slack;
  P.chk.npair = (int)s->chk_ent.size();  // (also the pair table of the contact phase's self-contact stage)
  P.chk.ngeom = (int)s->cgeoms.size();
  P.chk.plane_points = P.coll.has_plane ? 1 : 0;
  P.chk.pad = 0;
  if (const char* dm = std::getenv("RCSH_CHECK_SKIP")) P.chk.pad = std::atoi(dm);  // development: bit 0 no narrow phase, 1 no boxes, 2 no spheres
  std::memset(P.chk.gh, 0, sizeof(P.chk.gh));
  std::memset(P.chk.gvert, 0, sizeof(P.chk.gvert));
  std::memset(P.chk.glink, 0, sizeof(P.chk.glink));
  std::memset(P.chk.pad2, 0, sizeof(P.chk.pad2));
  std::memset(P.chk.gtype, 0, sizeof(P.chk.gtype));
  std::memset(P.chk.pad3, 0, sizeof(P.chk.pad3));
  for (size_t g = 0; g < s->cgeoms.size() && g < (size_t)kMaxCGeom; ++g) {
    const ContactGeom& cg = s->cgeoms[g];
    double* h = P.chk.gh[g];
    if (cg.type == 7) for (int k = 0; k < 3; ++k) h[k] = cg.aabb_h[k];
    else if (cg.type == 6) for (int k = 0; k < 3; ++k) h[k] = cg.size[k];
    else { h[0] = h[1] = cg.size[0]; h[2] = cg.size[0] + cg.size[1]; }
    P.chk.gvert[g][0] = cg.vert_adr; P.chk.gvert[g][1] = cg.type == 7 ? cg.vert_num : 0;
    P.chk.glink[g] = (int8_t)cg.link;
    P.chk.gtype[g] = (int8_t)cg.type;
  }
  return P;
}

int upload_boxtask(rcsh_sim* s) {
  BoxTaskCfg bt{s->box, s->task};
  if (!s->d_boxtask) HIP_TRY(hipMalloc(&s->d_boxtask, sizeof(BoxTaskCfg)));
  HIP_TRY(hipMemcpyAsync(s->d_boxtask, &bt, sizeof(bt), hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RCSH_OK;
}

int upload_model(rcsh_sim* s) {
  HIP_TRY(hipMemcpyAsync(s->d_model, &s->dm, sizeof(DevModel), hipMemcpyHostToDevice, s->stream));
  // the team kernels' per-link records live right behind the DevModel
  s->links.resize(kMaxLinks);
  fill_link_records(s->dm, s->links.data());
  HIP_TRY(hipMemcpyAsync(reinterpret_cast<char*>(s->d_model) + sizeof(DevModel), s->links.data(), sizeof(LinkRec) * kMaxLinks,
                         hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RCSH_OK;
}

// class bits of the contact sample points: bit 0 arm collision geoms, bit 1 gripper collision geoms
int upload_coll_classes(rcsh_sim* s) {
  if (s->cp.geom.empty()) return RCSH_OK;
  HIP_TRY(hipMemcpyAsync(s->d_coll_cls, s->cp_class.data(), s->cp_class.size(), hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RCSH_OK;
}

// MuJoCo's pair filters on two collision geoms of the robot (mj_collision: different weld bodies -- here: links; no
// parent-child pair unless one of the two is welded to the world), then what the callbacks make of a contact of the pair
// (SimRobot.cpp:172-182: either geom is an arm collision geom; SimGripper.cpp:108-130: not finger-finger, either geom is a
// gripper collision geom, geom[1] is not in the ignore list -- quirk Q6).  Pairs nobody reacts to are dropped.
constexpr int kSelfStageVertsHost = 304;  // contact_team.h: kSelfStageVerts
// all pairs MuJoCo's filters admit; `reacting_only`: drop those no collision callback reacts to (cls == 0)
// A geom welded to the world against a geom of the arm's first link (MuJoCo's parent-child filter lets the pair through: the parent is
// static): ONE hinge moves them relative to each other, and a rotation leaves every point's coordinate ALONG the hinge's axis alone.  If
// the two geoms' extents along that axis do not overlap, no joint angle brings them into contact -- the pair is dropped from the tables
// (the FR3's link 1 sits on link 0 with 0.1 mm between the hulls: the pair survived every bounding test in every pose, and the contact
// phase refined it in every substep of every escalated environment).
bool never_touch_across_first_hinge(const rcsh_sim* s, const ContactGeom& a, const ContactGeom& b) {
  const ContactGeom* w = a.link < 0 ? &a : (b.link < 0 ? &b : nullptr);
  const ContactGeom* l = a.link == 0 ? &a : (b.link == 0 ? &b : nullptr);
  if (!w || !l || s->dm.jtype[0] == kSlide) return false;
  const DevModel& m = s->dm;
  double ax[3] = {m.axis[0][0], m.axis[0][1], m.axis[0][2]}, aw[3];
  for (int k = 0; k < 3; ++k) aw[k] = m.rot0[0][3 * k] * ax[0] + m.rot0[0][3 * k + 1] * ax[1] + m.rot0[0][3 * k + 2] * ax[2];
  auto extent = [&](const ContactGeom& g, const double* u, double& lo, double& hi) {
    // of the geom along u, both in the frame of the geom's link
    double ug[3];  // u in the geom's frame
    for (int k = 0; k < 3; ++k) ug[k] = g.rot[k] * u[0] + g.rot[3 + k] * u[1] + g.rot[6 + k] * u[2];
    const double c = u[0] * g.pos[0] + u[1] * g.pos[1] + u[2] * g.pos[2];
    if (g.type == 7) {
      lo = 1e300; hi = -1e300;
      for (int v = 0; v < g.vert_num; ++v) {
        const double* x = &s->cverts[3 * (size_t)(g.vert_adr + v)];
        const double d = c + ug[0] * x[0] + ug[1] * x[1] + ug[2] * x[2];
        lo = std::min(lo, d); hi = std::max(hi, d);
      }
    } else {
      double e;
      if (g.type == 6) e = std::fabs(ug[0]) * g.size[0] + std::fabs(ug[1]) * g.size[1] + std::fabs(ug[2]) * g.size[2];
      else if (g.type == 3) e = std::fabs(ug[2]) * g.size[1] + g.size[0];
      else e = g.size[0];
      lo = c - e; hi = c + e;
    }
  };
  double wlo, whi, llo, lhi;
  extent(*w, aw, wlo, whi);
  extent(*l, ax, llo, lhi);
  const double off = aw[0] * m.pos0[0][0] + aw[1] * m.pos0[0][1] + aw[2] * m.pos0[0][2];
  llo += off; lhi += off;
  if (w->type == 7 && w->vert_num == 0) return false;
  if (l->type == 7 && l->vert_num == 0) return false;
  return llo - whi > 1e-7 || wlo - lhi > 1e-7;
}

std::vector<SelfPair> list_geom_pairs(const rcsh_sim* s, bool reacting_only) {
  std::vector<SelfPair> out;
  const int ng = (int)s->cgeoms.size();
  auto parent = [&](int link) { return link < s->narm ? link - 1 : s->narm - 1; };
  for (int i = 0; i < ng; ++i)
    for (int j = i + 1; j < ng; ++j) {
      const ContactGeom &a = s->cgeoms[i], &b = s->cgeoms[j];
      if (a.link == b.link) continue;
      if (a.link >= 0 && b.link >= 0 && (parent(a.link) == b.link || parent(b.link) == a.link)) continue;
      {
        // MuJoCo's mask filter: the pair collides if (contype0 & conaffinity1) || (contype1 & conaffinity0) (advisor, round 4: pairs the
        // masks exclude raised the sticky flags -- and, since round 5, would send an environment to the contact-resolving launch)
        const auto& ct = s->hm.geom_contype;
        const auto& ca = s->hm.geom_conaffinity;
        if (a.geom_id < (int)ct.size() && b.geom_id < (int)ct.size() && a.geom_id < (int)ca.size() && b.geom_id < (int)ca.size() &&
            !((ct[a.geom_id] & ca[b.geom_id]) || (ct[b.geom_id] & ca[a.geom_id])))
          continue;
      }
      if ((a.type == 7 && a.vert_num == 0) || (b.type == 7 && b.vert_num == 0)) continue;  // mesh blob missing from the checkout
      if (a.vert_num + b.vert_num > kSelfStageVertsHost) continue;  // (the contact table admits hulls of at most 152 vertices each: model.cpp build_contact_table)
      if (never_touch_across_first_hinge(s, a, b)) continue;
      const bool swap = a.type > b.type;  // geom[0] / geom[1] of the contact: by type, then by id (the table is in id order)
      const ContactGeom &g0 = swap ? b : a, &g1 = swap ? a : b;
      int cls = 0;
      if ((g0.cls | g1.cls) & 1) cls |= 1;
      if (!((g0.cls & 4) && (g1.cls & 4)) && ((g0.cls | g1.cls) & 16) && !(g1.cls & 8)) cls |= 2;
      if (!cls && reacting_only) continue;
      SelfPair pr{};
      pr.g0 = (int16_t)(swap ? j : i); pr.g1 = (int16_t)(swap ? i : j);
      pr.l0 = (int16_t)g0.link; pr.l1 = (int16_t)g1.link;
      pr.cls = cls;
      {
        // joints on the tree path between the two links: root paths' symmetric difference (arm link i: joints 0..i; a finger:
        // the whole arm and its own slide; welded to the world: none)
        auto root_path = [&](int link) -> int {
          if (link < 0) return 0;
          if (link < s->narm) return (1 << (link + 1)) - 1;
          return ((1 << s->narm) - 1) | (1 << link);
        };
        pr.joints = root_path(g0.link) ^ root_path(g1.link);
      }
      auto bounds = [](const ContactGeom& g, double* c, double& r, double* rot, double* h) {
        // bounding box of the geom (geom frame: centre lc, half extents h), carried into the link frame
        double lc[3] = {0, 0, 0};
        if (g.type == 7) { for (int k = 0; k < 3; ++k) { lc[k] = g.aabb_c[k]; h[k] = g.aabb_h[k]; } }
        else if (g.type == 6) { for (int k = 0; k < 3; ++k) h[k] = g.size[k]; }
        else { h[0] = h[1] = g.size[0]; h[2] = g.size[0] + g.size[1]; }
        for (int k = 0; k < 3; ++k) c[k] = g.rot[3 * k] * lc[0] + g.rot[3 * k + 1] * lc[1] + g.rot[3 * k + 2] * lc[2] + g.pos[k];
        for (int k = 0; k < 9; ++k) rot[k] = g.rot[k];
        r = std::sqrt(h[0] * h[0] + h[1] * h[1] + h[2] * h[2]);
      };
      bounds(g0, pr.c0, pr.r0, pr.rot0, pr.h0);
      bounds(g1, pr.c1, pr.r1, pr.rot1, pr.h1);
      out.push_back(pr);
    }
  return out;
}
void build_self_pairs(rcsh_sim* s) {
  s->pairs.clear();
  if (std::getenv("RCSH_DEBUG_NO_SELF_PAIRS")) return;  // development switch: what the pair tests cost
  s->pairs = list_geom_pairs(s, true);
}

// The tables of the end-of-launch check for contacts nobody resolves (check_team.h): ALL admitted geom pairs (full records for the
// box test and the narrow phase, packed entries for the sphere test).  Pairs of the same two bodies stay together, so that a
// cluster of pairs that come near at once (the two fingers' pads when the gripper closes) spreads over the lanes.
void build_check_table(rcsh_sim* s) {
  s->chk_pairs = list_geom_pairs(s, false);
  auto key = [](const SelfPair& p) { const int a = std::min(p.l0, p.l1) + 1, b = std::max(p.l0, p.l1) + 1; return a * 64 + b; };
  std::stable_sort(s->chk_pairs.begin(), s->chk_pairs.end(), [&](const SelfPair& x, const SelfPair& y) { return key(x) < key(y); });
  s->chk_ent.clear();
  for (const auto& p : s->chk_pairs) {
    CheckEntry e{};
    // (bits 16-23: 1 + the deepest link both geoms' links descend from or are -- 0: none, the world; the slack test charges each geom
    // with the joints below it)
    int ca = -1;
    {
      const int na = s->narm;
      auto parent = [&](int link) { return link < na ? link - 1 : na - 1; };
      auto is_anc = [&](int a, int l) { for (int k = l; k >= 0; k = parent(k)) if (k == a) return true; return false; };
      for (int k = p.l0; k >= 0 && ca < 0; k = parent(k)) if (p.l1 >= 0 && is_anc(k, p.l1)) ca = k;
    }
    e.geoms = (uint32_t)p.g0 | ((uint32_t)p.g1 << 8) | ((uint32_t)(ca + 1) << 16);
    e.rsum = (float)(p.r0 + p.r1) * 1.000001f + 2e-6f;  // (single-precision centres: the sum of the radii rounded up)
    s->chk_ent.push_back(e);
  }
  // the geoms' bounding boxes in their links' frames (what SelfPair carries per pair, once per geom)
  s->chk_geoms.clear();
  for (const auto& cg : s->cgeoms) {
    CheckGeom g{};
    double lc[3] = {0, 0, 0};
    if (cg.type == 7) for (int k = 0; k < 3; ++k) lc[k] = cg.aabb_c[k];
    for (int k = 0; k < 3; ++k) g.c[k] = cg.rot[3 * k] * lc[0] + cg.rot[3 * k + 1] * lc[1] + cg.rot[3 * k + 2] * lc[2] + cg.pos[k];
    for (int k = 0; k < 9; ++k) g.rot[k] = cg.rot[k];
    s->chk_geoms.push_back(g);
  }
}

// lever[j]: how far one radian of hinge j (one metre of a slide) can move a point of any collision geom downstream of it.
// Distances between consecutive joint anchors are constants of the links; a finger's anchor slides, so its stroke is added.
void build_self_levers(rcsh_sim* s) {
  const DevModel& m = s->dm;
  const int na = s->narm, nl = s->nl;
  auto parent = [&](int link) { return link < na ? link - 1 : na - 1; };
  // reach[L]: from link L's joint anchor to the farthest point of a collision geom ON link L (link frame)
  std::vector<double> reach(nl, 0.0), hop(nl, 0.0), stroke(nl, 0.0);
  for (const auto& g : s->cgeoms) {
    if (g.link < 0) continue;
    double c[3], h[3] = {g.size[0], g.size[1], g.size[2]}, lc[3] = {0, 0, 0};
    if (g.type == 7) for (int k = 0; k < 3; ++k) { lc[k] = g.aabb_c[k]; h[k] = g.aabb_h[k]; }
    else if (g.type == 3) { h[0] = h[1] = g.size[0]; h[2] = g.size[0] + g.size[1]; }
    for (int k = 0; k < 3; ++k) c[k] = g.rot[3 * k] * lc[0] + g.rot[3 * k + 1] * lc[1] + g.rot[3 * k + 2] * lc[2] + g.pos[k] - m.jpos[g.link][k];
    const double r = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]) + std::sqrt(h[0] * h[0] + h[1] * h[1] + h[2] * h[2]);
    reach[g.link] = std::max(reach[g.link], r);
  }
  for (int L = 0; L < nl; ++L) {
    // hop[L]: from the parent link's anchor to link L's anchor (parent link frame, at qpos0; a hinge's anchor does not move)
    const int p = parent(L);
    double a[3];
    for (int k = 0; k < 3; ++k) a[k] = m.pos0[L][k] + m.rot0[L][3 * k] * m.jpos[L][0] + m.rot0[L][3 * k + 1] * m.jpos[L][1] + m.rot0[L][3 * k + 2] * m.jpos[L][2] - (p >= 0 ? m.jpos[p][k] : 0.0);
    hop[L] = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    if (m.jtype[L] == kSlide) stroke[L] = std::max(std::fabs(m.range[L][0] - m.qpos0[L]), std::fabs(m.range[L][1] - m.qpos0[L]));
  }
  // far[L]: from link L's anchor to the farthest geom point on L or downstream of it
  std::vector<double> far(nl, 0.0);
  for (int L = nl - 1; L >= 0; --L) {
    far[L] = std::max(far[L], reach[L] + stroke[L]);
    const int p = parent(L);
    if (p >= 0) far[p] = std::max(far[p], hop[L] + stroke[L] + far[L]);
  }
  for (int j = 0; j < 12; ++j) s->self_lever[j] = 0.0;
  for (int j = 0; j < nl; ++j) s->self_lever[j] = m.jtype[j] == kSlide ? 1.0 : 1.01 * far[j] + 1e-3;
  // link_lever[j][l]: the same bound for the geoms ON link l alone (j an ancestor-or-self joint of l) -- what the contact phase's slack
  // test charges a geom pair / a geom above the floor with.  The isotropic lever above takes the whole arm's reach for joint 1; the pair
  // (link 0, link 2), whose hulls stay a centimetre apart in every pose, sits 0.2 m from that axis: charged with 1.2 m per radian it was
  // due in nearly every substep, and with it the whole collision pass.
  for (int k = 0; k < 144 + 12 * 32; ++k) s->link_lever[k] = 0.0f;
  for (int l = 0; l < nl; ++l) {
    double acc = reach[l];  // from link l's anchor to the farthest point of a geom on l
    for (int j = l; j >= 0; j = parent(j)) {
      // acc: from joint j's anchor to the farthest point of a geom on l, over every configuration of the joints in between
      s->link_lever[j * 12 + l] = (float)(m.jtype[j] == kSlide ? 1.0 : (1.01 * (acc + stroke[l]) + 1e-3) * 1.000001);
      acc += hop[j] + stroke[j];
    }
  }
  // ... and per GEOM (kLevGeom + j * 32 + g): the same bound for the points of geom g alone.  A link's lever is its farthest geom's: link 7
  // carries the flange's hull AND the hand's, 0.2 m from its joint, and the pair (link 5's hull, the flange's hull) -- 14-17 mm apart in
  // every pose -- was charged the hand's reach: in step_until_convergence, whose launches move a joint by up to five degrees, its
  // certificate failed in a third of the batch at every step (check_team.h: the narrow phase's second chance).
  static_assert(kMaxCGeom <= 32, "a column per collision geom");
  for (size_t gi = 0; gi < s->cgeoms.size() && gi < 32; ++gi) {
    const auto& g = s->cgeoms[gi];
    if (g.link < 0) continue;
    double c[3], h[3] = {g.size[0], g.size[1], g.size[2]}, lc[3] = {0, 0, 0};
    if (g.type == 7) for (int k = 0; k < 3; ++k) { lc[k] = g.aabb_c[k]; h[k] = g.aabb_h[k]; }
    else if (g.type == 3) { h[0] = h[1] = g.size[0]; h[2] = g.size[0] + g.size[1]; }
    for (int k = 0; k < 3; ++k) c[k] = g.rot[3 * k] * lc[0] + g.rot[3 * k + 1] * lc[1] + g.rot[3 * k + 2] * lc[2] + g.pos[k] - m.jpos[g.link][k];
    double acc = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]) + std::sqrt(h[0] * h[0] + h[1] * h[1] + h[2] * h[2]);
    const int l = g.link;
    // (the geom's OWN hinge: what a radian of it moves is a point's distance from the AXIS, not from the anchor -- the largest over the
    // eight corners of the geom's box; a flange that is a cylinder about its joint's axis: its radius instead of its length)
    double radial = 0.0;
    for (int corner = 0; corner < 8; ++corner) {
      const double sg[3] = {corner & 1 ? h[0] : -h[0], corner & 2 ? h[1] : -h[1], corner & 4 ? h[2] : -h[2]};
      double v[3];
      for (int k = 0; k < 3; ++k) v[k] = c[k] + g.rot[3 * k] * sg[0] + g.rot[3 * k + 1] * sg[1] + g.rot[3 * k + 2] * sg[2];
      const double* ax = m.axis[l];
      const double an = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
      const double al = an > 0 ? (v[0] * ax[0] + v[1] * ax[1] + v[2] * ax[2]) / an : 0.0;
      const double r2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] - al * al;
      radial = std::max(radial, std::sqrt(std::max(r2, 0.0)));
    }
    for (int j = l; j >= 0; j = parent(j)) {
      const double arm = j == l && m.jtype[j] != kSlide ? std::min(radial, acc) : acc + stroke[l];
      s->link_lever[144 + j * 32 + (int)gi] = (float)(m.jtype[j] == kSlide ? 1.0 : (1.01 * arm + 1e-3) * 1.000001);
      acc += hop[j] + stroke[j];
    }
  }
}

int upload_contact_table(rcsh_sim* s) {
  if (s->cgeoms.empty()) return RCSH_OK;
  build_self_pairs(s);
  build_self_levers(s);
  if (s->d_pairs) { HIP_TRY(hipStreamSynchronize(s->stream)); HIP_TRY(hipFree(s->d_pairs)); s->d_pairs = nullptr; }
  if (!s->pairs.empty()) {
    HIP_TRY(hipMalloc(&s->d_pairs, sizeof(SelfPair) * s->pairs.size()));
    HIP_TRY(hipMemcpyAsync(s->d_pairs, s->pairs.data(), sizeof(SelfPair) * s->pairs.size(), hipMemcpyHostToDevice, s->stream));
  }
  build_check_table(s);
  if (s->d_chk_ent) { HIP_TRY(hipStreamSynchronize(s->stream)); HIP_TRY(hipFree(s->d_chk_ent)); s->d_chk_ent = nullptr; }
  if (s->d_chk_geoms) { HIP_TRY(hipFree(s->d_chk_geoms)); s->d_chk_geoms = nullptr; }
  // (a scene with more admitted pairs than a lane keeps entries for is NOT refused -- plain Sim.step users never read the flag --: the
  // pairs past the capacity stay unchecked, counted and reported: rcsh_sim_contact_check_unchecked_pairs; advisor, round 4)
  s->chk_unchecked = 0;
  if ((int)s->chk_ent.size() > kMaxCheckPairs) {
    s->chk_unchecked = (int)s->chk_ent.size() - kMaxCheckPairs;
    s->chk_ent.resize(kMaxCheckPairs);
    s->chk_pairs.resize(kMaxCheckPairs);
  }
  if (!s->chk_ent.empty()) {
    HIP_TRY(hipMalloc(&s->d_chk_ent, sizeof(CheckEntry) * s->chk_ent.size()));
    HIP_TRY(hipMemcpyAsync(s->d_chk_ent, s->chk_ent.data(), sizeof(CheckEntry) * s->chk_ent.size(), hipMemcpyHostToDevice, s->stream));
  }
  if (!s->d_lev) HIP_TRY(hipMalloc(&s->d_lev, sizeof(s->link_lever)));
  HIP_TRY(hipMemcpyAsync(s->d_lev, s->link_lever, sizeof(s->link_lever), hipMemcpyHostToDevice, s->stream));
  if (!s->chk_geoms.empty()) {
    HIP_TRY(hipMalloc(&s->d_chk_geoms, sizeof(CheckGeom) * s->chk_geoms.size()));
    HIP_TRY(hipMemcpyAsync(s->d_chk_geoms, s->chk_geoms.data(), sizeof(CheckGeom) * s->chk_geoms.size(), hipMemcpyHostToDevice, s->stream));
  }
  if (!s->d_cgeoms) HIP_TRY(hipMalloc(&s->d_cgeoms, sizeof(ContactGeom) * s->cgeoms.size()));
  HIP_TRY(hipMemcpyAsync(s->d_cgeoms, s->cgeoms.data(), sizeof(ContactGeom) * s->cgeoms.size(), hipMemcpyHostToDevice, s->stream));
  if (!s->d_cverts && !s->cverts.empty()) {
    HIP_TRY(hipMalloc(&s->d_cverts, sizeof(double) * s->cverts.size()));
    HIP_TRY(hipMemcpyAsync(s->d_cverts, s->cverts.data(), sizeof(double) * s->cverts.size(), hipMemcpyHostToDevice, s->stream));
  }
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RCSH_OK;
}

int prof_flush(rcsh_sim* s) {
  for (int i = 0; i < s->prof_pending; ++i) {
    float ms = 0;
    HIP_TRY(hipEventSynchronize(s->ev_stop[i]));
    HIP_TRY(hipEventElapsedTime(&ms, s->ev_start[i], s->ev_stop[i]));
    s->prof_ms += ms;
  }
  s->prof_launches += s->prof_pending;
  s->prof_pending = 0;
  return RCSH_OK;
}

// Which compilation of the team kernel a launch without free box / contacts / detection takes.  Measured (profiles/r3_occ2):
// a second resident wavefront per SIMD is worth 1.4-1.6x once the batch brings more than one wavefront per SIMD -- for a build
// that fits 256 registers by itself (the 6-dof arms: 67.6 M env-steps/s at 8192 environments against 49.0 M at 4096) or spills
// a few values (SO101: 200 bytes of scratch, 1.21-1.31x); a build squeezed into 256 registers at the price of hundreds of
// scratch accesses per substep loses (FR3 + hand: 488 bytes, 0.70x; xArm7: 888 bytes, 0.9x).  Hence AUTO: the <= 256-register
// build when the batch has more wavefronts than the device has SIMDs AND that build's private segment is at most 256 bytes.
// RCSH_OCC2=0/1 in the environment overrides the rule (measurements).
template <class T, bool F>
bool occ2_build_pays() {
  static const bool pays = [] {
    hipFuncAttributes a{};
    if (hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_run_team_occ2<T, F, false, false, false>)) != hipSuccess) return false;
    return a.localSizeBytes <= 256;
  }();
  return pays;
}
template <class T, bool F>
bool use_occ2(const rcsh_sim* s) {
  if (s->kernel == RCSH_KERNEL_TEAM_OCC2) return true;
  if (s->kernel == RCSH_KERNEL_TEAM) return false;
  static const int forced = [] { const char* e = std::getenv("RCSH_OCC2"); return e ? std::atoi(e) : -1; }();
  if (forced >= 0) return forced != 0;
  return (s->n + 3) / 4 > s->n_simd && occ2_build_pays<T, F>();
}

int launch_run_once(rcsh_sim* s, const RunOp& op_in, bool timed);

// step_until_convergence with the robot's contacts resolved environment by environment runs in PIECES of conv_chunk substeps (RunOp::
// conv_chunk; RCSH_CONV_CHUNK, 0: one launch as before): a pair of launches per piece, each lean launch with its own copy of the state and
// its own certificate over the piece's travel alone.  Every piece is enqueued -- the host does not know who has converged --: a
// workgroup whose environments all have leaves after its prologue.
int launch_run(rcsh_sim* s, const RunOp& op_in, bool timed) {
  const int cap = s->sim.max_convergence_steps;
  const bool esc = s->esc_mode && s->box.resolve && !s->box.present && !op_in.observe_only;
  if (op_in.nsteps < 0 && !op_in.do_reset && esc && s->conv_chunk > 0 && cap > s->conv_chunk) {
    // (a servo's travel is front-loaded: the first pieces are the ones whose certificates fail, the last ones only cost their launches'
    // fixed parts -- with conv_grow the piece doubles after every second one, up to 128 substeps)
    int piece = s->conv_chunk, k = 0;
    for (int done = 0; done < cap; done += piece, ++k) {
      if (s->conv_grow && k > 0 && k % 2 == 0 && piece < 128) piece = std::min(2 * piece, 128);
      RunOp op = op_in;
      op.conv_chunk = piece;
      if (done > 0) { op.conv_resume = 1; op.apply_action = 0; }
      if (int rc = launch_run_once(s, op, timed)) return rc;
    }
    return RCSH_OK;
  }
  return launch_run_once(s, op_in, timed);
}

int launch_run_once(rcsh_sim* s, const RunOp& op_in, bool timed) {
  Params P = make_params(s);
  RunOp op = op_in;
  // the end-of-launch check for contacts nobody resolves: stepping launches at the handle's cadence; a caller may ask for it itself
  if (!op.check && (op.nsteps != 0 || op.do_reset) && s->contact_check && s->check_every > 0) op.check = (s->check_seq++ % s->check_every) == 0;
  if (!s->contact_check) op.check = 0;
  hipError_t err = hipSuccess;
  if (timed && s->prof_region) {
    // region mode: one event before the first timed launch, one after the last (rcsh_prof_read): no event traffic in between
    if (s->prof_region_launches == 0) HIP_TRY(hipEventRecord(s->ev_start[0], s->stream));
    s->prof_region_launches++;
  }
  const bool sample = timed && s->prof && !s->prof_region && (s->prof_seen++ % s->prof_every) == 0;
  if (sample) {
    if (s->prof_pending == kProfRing) {
      int rc = prof_flush(s);
      if (rc) return rc;
    }
    HIP_TRY(hipEventRecord(s->ev_start[s->prof_pending], s->stream));
  }
  // k_run_team: 16 lanes per environment, 4 environments per wavefront.  4096 environments are 1024 wavefronts = one per
  // SIMD of the chip.  (A one-lane-per-environment kernel existed through round 1; it lost at every batch size and could
  // step neither dry friction nor free bodies, and was removed: csrc/dyn.h keeps its formulas for the host-side model
  // finalisation and the shared math.)
  // DET: launches that run the collision callbacks (step_until_convergence) of a model with collision geoms carry the
  // contact detection of the position stage; Sim::step(k) never looks at the flags (sim.cpp:108-115)
  const bool det = op.nsteps < 0 && (P.coll.has_plane || !s->pairs.empty());
  // per-environment escalation: stepping launches of a box-less scene whose robot contacts are resolved environment by environment
  const bool esc = s->esc_mode && s->box.resolve && !s->box.present && (op.nsteps != 0 || op.do_reset) && !op.observe_only;
  if (esc) {
    op.esc = s->d_esc; op.esc_ctr = s->d_esc_ctr;
    op.snap = s->d_snap; op.snap_flags = s->d_snap_flags; op.snap_conv = s->d_snap_conv;
  }
  bool launched = false;
  hipError_t esc_err = hipSuccess;  // (stream / event calls of the split contact-resolving launch)
  bool ok = dispatch_topology(s->narm, s->grip, [&](auto topo) {
    using T = decltype(topo);
    const dim3 grid(((s->n + 31) / 32) * 8), block(64);
    auto go = [&](auto fric, auto box, auto con) {
      constexpr bool F = decltype(fric)::value, B = decltype(box)::value, C = decltype(con)::value;
      if (det) hipLaunchKernelGGL((k_run_team<T, F, B, C, true>), grid, block, 0, s->stream, P, op);
      else if constexpr (!B && !C) {
        if (use_occ2<T, F>(s)) hipLaunchKernelGGL((k_run_team_occ2<T, F, false, false, false>), grid, block, 0, s->stream, P, op);
        else hipLaunchKernelGGL((k_run_team<T, F, false, false, false>), grid, block, 0, s->stream, P, op);
      } else hipLaunchKernelGGL((k_run_team<T, F, B, C, false>), grid, block, 0, s->stream, P, op);
      launched = true;
    };
    using Y = std::true_type;
    using N = std::false_type;
#ifdef RCSH_DEV_NO_CONTACT_KERNELS  // development builds: the lean instantiations alone
    if (s->dm.has_friction) go(Y{}, N{}, N{});
    else go(N{}, N{}, N{});
#else
    if (s->box.present && s->box.resolve) {
      // free box + contacts of the robot's geoms (FR3 + hand; xArm7 + gripper: friction rows in the coupled solve):
      // rcsh_sim_add_free_box checked the archetype
      if constexpr (T::NARM == 7 && T::GRIP) {
        if (s->dm.has_friction) go(Y{}, Y{}, Y{});
        else go(N{}, Y{}, Y{});
      }
    } else if (s->box.present) {
      // scenes with a free box that only touches the floor: FR3 + hand, and the 7-dof arm with dry joint friction
      if constexpr (T::NARM == 7 && T::GRIP) go(N{}, Y{}, N{});
      else if constexpr (T::NARM == 7) go(Y{}, Y{}, N{});
    } else if (s->box.resolve) {
      // no free body, contacts of the robot with the floor and with itself resolved (rcsh_sim_set_contact_options; FR3 + hand):
      // by the whole batch on the contact-resolving kernel, or environment by environment (RunOp::esc_role)
      if constexpr (T::NARM == 7 && T::GRIP) {
        if (esc) {
          // (the certifying check -- check_team.h -- is the default; RCSH_CHECK_CERTIFY=0: the check of the final position alone, round 5's)
          static const int certify = [] { const char* e = std::getenv("RCSH_CHECK_CERTIFY"); return e ? std::atoi(e) : 1; }();
          // How this step's contact-resolving work is enqueued.  The environments that are escalated already do not depend on the step's
          // lean launch: with any of them around (as far as the host knows: the device's last report, at most two steps old -- the host
          // does not run further ahead than that) they go FIRST, on a stream of their own, and the lean launch fills the rest of the chip
          // beside them; the newly flagged ones follow behind both.  The step then takes max(escalated, ~2 x lean) instead of their sum
          // (the lean launch's last workgroups wait for a SIMD: 4096 environments are one wavefront per SIMD).
          auto esc_chk = [&](hipError_t e_) { if (e_ != hipSuccess && esc_err == hipSuccess) esc_err = e_; };
          bool split = false;
          if (s->h_esc_hint && s->esc_split) {
            ++s->esc_seq;
            if (s->esc_seq > 2) {  // (throttle: the lean launch of step seq - 2 has ended)
              const uint32_t want = s->esc_seq - 2;
              for (long spins = 0; (int32_t)(s->h_esc_hint[0] - want) < 0; ++spins) {
                if (spins > 2000000) { esc_chk(hipStreamSynchronize(s->stream)); break; }  // (a hint that never comes: wait the plain way)
                __builtin_ia32_pause();
              }
            }
            // (few escalated environments: their workgroups leave the lean launch nearly the whole chip.  Many: the two kernels' wavefronts
            // share CUs for milliseconds and both run slower -- measured on the 1000-step rollout, 3.87 -> 4.53 ms a step -- so not then.)
            split = s->h_esc_hint[1] > 0 && s->h_esc_hint[1] <= (uint32_t)s->esc_split_max;
            op.esc_seq = (int32_t)s->esc_seq;
            op.esc_host = const_cast<uint32_t*>(s->h_esc_hint);
          }
          if (split) {
            esc_chk(hipEventRecord(s->esc_ev_prev, s->stream));
            esc_chk(hipStreamWaitEvent(s->esc_stream, s->esc_ev_prev, 0));
            RunOp op_old = op;
            op_old.esc_role = 2; op_old.esc_part = 1; op_old.check = 0;
            // (the lean launch must not win the race for the SIMDs: it waits until the other stream has got as far as its kernel)
            esc_chk(hipEventRecord(s->esc_ev_started, s->esc_stream));
            esc_chk(hipStreamWaitEvent(s->stream, s->esc_ev_started, 0));
            hipLaunchKernelGGL((k_run_team<T, false, false, true, false>), grid, block, 0, s->esc_stream, P, op_old);
            esc_chk(hipGetLastError());
            esc_chk(hipEventRecord(s->esc_ev_old, s->esc_stream));
          }
          op.esc_role = 1; op.check = certify ? 2 : 1;
          // (measurement switches: what the copy of the state / the end-of-launch check cost a step -- results are wrong with either)
          static const int no_snap = [] { const char* e = std::getenv("RCSH_ESC_MEASURE_NO_SNAP"); return e ? std::atoi(e) : 0; }();
          static const int no_check = [] { const char* e = std::getenv("RCSH_ESC_MEASURE_NO_CHECK"); return e ? std::atoi(e) : 0; }();
          double* const snap_keep = op.snap;
          if (no_snap) op.snap = nullptr;
          if (no_check) op.check = 0;
          go(N{}, N{}, N{});
          op.snap = snap_keep;
          op.esc_role = 2; op.check = certify ? 2 : 0;
          op.esc_part = split ? 2 : 0;
          if (split) esc_chk(hipStreamWaitEvent(s->stream, s->esc_ev_old, 0));
          go(N{}, N{}, Y{});
        } else {
          op.force_contact = (s->box.resolve & 2) ? 1 : 0;
          go(N{}, N{}, Y{});
        }
      }
    } else if (s->dm.has_friction)
      go(Y{}, N{}, N{});
    else
      go(N{}, N{}, N{});
#endif
    err = hipGetLastError();
  });
  if (!ok || !launched) return fail(RCSH_ERR_MODEL, "no kernel instantiated for this archetype");
  if (err != hipSuccess) return fail(RCSH_ERR_DEVICE, std::string("k_run launch: ") + hipGetErrorString(err));
  if (esc_err != hipSuccess) return fail(RCSH_ERR_DEVICE, std::string("contact-resolving launch (streams / events): ") + hipGetErrorString(esc_err));
  if (sample) {
    HIP_TRY(hipEventRecord(s->ev_stop[s->prof_pending], s->stream));
    s->prof_pending++;
  }
  return RCSH_OK;
}

template <class F>
auto with_layout(rcsh_sim* s, F&& fn) {
  int out = -1;
  dispatch_topology(s->narm, s->grip, [&](auto topo) { out = fn(topo); });
  return out;
}

int field_of(rcsh_sim* s, const char* name) {
  return with_layout(s, [&](auto topo) {
    using L = Lay<decltype(topo)>;
    std::string f(name);
    if (f == "qpos") return (int)L::QPOS;
    if (f == "qvel") return (int)L::QVEL;
    if (f == "ctrl") return (int)L::CTRL;
    if (f == "time") return (int)L::TIME;
    if (f == "cb") return (int)L::CB;
    if (f == "prevq") return (int)L::PREVQ;
    if (f == "target") return (int)L::TARGET;
    if (f == "grip") return (int)L::GRIP;
    if (f == "site") return (int)L::SITE;
    if (f == "preva") return (int)L::PREVA;
    if (f == "origin") return (int)L::ORIGIN;
    if (f == "lasta") return (int)L::LASTA;
    if (f == "box") return (int)L::BOX;
    if (f == "qpre") return (int)L::QPRE;
    if (f == "xs") return (int)L::XS;
    if (f == "sep") return (int)L::SEP;
    return -1;
  });
}

int upload_mask(rcsh_sim* s, const uint8_t* mask, const uint8_t** dev) {
  *dev = nullptr;
  if (!mask) return RCSH_OK;
  HIP_TRY(hipMemcpyAsync(s->d_mask, mask, s->n, hipMemcpyHostToDevice, s->stream));
  *dev = s->d_mask;
  return RCSH_OK;
}

// host [n][width] -> state fields
constexpr int kStageWidth = 48;  // doubles per environment of the staging buffer (the widest field group: the free body's state)
static_assert(kBoxState <= kStageWidth, "the free body's state passes through the staging buffer in one piece");
int scatter_host(rcsh_sim* s, int field0, int width, const double* src, const uint8_t* mask) {
  if (width > kStageWidth) return fail(RCSH_ERR_ARG, "scatter_host: field group wider than the staging buffer");
  const uint8_t* dm = nullptr;
  int rc = upload_mask(s, mask, &dm);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(s->d_stage, src, sizeof(double) * s->n * width, hipMemcpyHostToDevice, s->stream));
  // (a write into the state from outside the stepping launches -- set_joints_hard, mjData.qpos = ... -- may move the robot: the gaps the
  // contact check and the contact phase remember for the old position are void; zero = "look at everything")
  if (s->d_slack) HIP_TRY(hipMemsetAsync(s->d_slack, 0, sizeof(float) * (size_t)kSlackStride * s->n, s->stream));
  hipLaunchKernelGGL(k_scatter, dim3(grid_for(s->n)), dim3(kBlock), 0, s->stream, s->S, s->n, field0, width, s->d_stage, dm);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RCSH_OK;
}

int gather_host(rcsh_sim* s, int field0, int width, double* dst) {
  if (width > kStageWidth) return fail(RCSH_ERR_ARG, "gather_host: field group wider than the staging buffer");
  hipLaunchKernelGGL(k_gather, dim3(grid_for(s->n)), dim3(kBlock), 0, s->stream, (const double*)s->S, s->n, field0, width, s->d_stage);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(dst, s->d_stage, sizeof(double) * s->n * width, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RCSH_OK;
}

int flag_host(rcsh_sim* s, uint32_t bit, uint8_t* dst) {
  if (!dst) return RCSH_OK;
  hipLaunchKernelGGL(k_flags_to_bytes, dim3(grid_for(s->n)), dim3(kBlock), 0, s->stream, (const uint32_t*)s->flags, s->n, bit, s->d_bytes);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(dst, s->d_bytes, s->n, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RCSH_OK;
}

// flags |= set, &= ~clear for masked environments: a small kernel on the handle's stream, so that it is ordered with
// everything the *_dev entry points enqueued there
int flags_update_host(rcsh_sim* s, uint32_t set, uint32_t clear, const uint8_t* mask) {
  const uint8_t* dm = nullptr;
  int rc = upload_mask(s, mask, &dm);
  if (rc) return rc;
  hipLaunchKernelGGL(k_flags_update, dim3(grid_for(s->n)), dim3(kBlock), 0, s->stream, s->flags, s->n, set, clear, dm);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RCSH_OK;
}

// Host-buffer env.reset(mask): the reset launch writes observation / info / gripper width of the masked environments only,
// the rows of the others in the shared staging buffers would be whatever an earlier call left there.  An observation-only
// pass (no stepping) over the complement fills them with the environments' current observation.
int observe_unmasked(rcsh_sim* s, const uint8_t* mask) {
  if (!mask) return RCSH_OK;
  std::vector<uint8_t> inv(s->n);
  bool any = false;
  for (int e = 0; e < s->n; ++e) { inv[e] = !mask[e]; any = any || inv[e]; }
  if (!any) return RCSH_OK;
  uint8_t* d_inv = s->d_bytes + (size_t)s->n * 8;  // (info occupies the first 8 n bytes)
  HIP_TRY(hipMemcpyAsync(d_inv, inv.data(), s->n, hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));  // `inv` goes out of scope
  RunOp op{};
  op.nsteps = 0;
  op.write_obs = 1;
  op.observe_only = 1;  // the frames the reset launch just recorded for the masked environments stay pending (rend.count / last)
  op.mask = d_inv;
  op.obs = s->d_stage2; op.info = s->d_bytes; op.gripper_width = s->d_stage;
  return launch_run(s, op, false);
}

std::vector<double> tile(const double* row, int width, int n) {
  std::vector<double> v((size_t)n * width);
  for (int e = 0; e < n; ++e) std::memcpy(&v[(size_t)e * width], row, sizeof(double) * width);
  return v;
}

// every entry point works on the handle's own GPU, whatever device the calling thread had current
#define REQUIRE_SIM(s)                                        \
  if (!(s)) return fail(RCSH_ERR_ARG, "null sim handle"); \
  HIP_TRY(hipSetDevice((s)->device))
#define REQUIRE_ROBOT(s) \
  if (!(s)->robot.present) return fail(RCSH_ERR_STATE, "no robot attached: call rcsh_sim_add_robot first")
#define REQUIRE_GRIPPER(s) \
  if (!(s)->gripcfg.present) return fail(RCSH_ERR_STATE, "no gripper attached: call rcsh_sim_add_gripper first")

}  // namespace

extern "C" {

const char* rcsh_last_error(void) { return g_err.c_str(); }
int rcsh_abi_version(void) { return RCSH_ABI_VERSION; }
int rcsh_device_count(void) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) return 0;
  return c;
}

int rcsh_sim_create(const rcsh_model_desc* model, int32_t n_envs, int32_t device, rcsh_sim** out) {
  if (!model || !out || n_envs < 1) return fail(RCSH_ERR_ARG, "rcsh_sim_create: bad arguments");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
    return fail(RCSH_ERR_DEVICE, "no HIP device available: the batched backend has no CPU path");
  if (device < 0 || device >= count) return fail(RCSH_ERR_DEVICE, "device index out of range");
  rcsh_sim* s = new rcsh_sim();
  s->hm.copy_from(*model);
  std::string why = finalize_model(s->hm, s->dm, s->act_slot);
  if (!why.empty()) {
    delete s;
    return fail(RCSH_ERR_MODEL, why);
  }
  s->device = device;
  s->n = n_envs;
  s->narm = s->dm.narm; s->nl = s->dm.nl; s->grip = s->dm.has_gripper != 0;
  s->nu = s->narm + (s->grip ? 1 : 0);
  s->nfields = with_layout(s, [&](auto topo) { return (int)Lay<decltype(topo)>::COUNT; });
  auto cleanup = [&](int code, const std::string& msg) {
    rcsh_sim_destroy(s);
    return fail(code, msg);
  };
#define HIP_NEW(expr)                                                                  \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) return cleanup(RCSH_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)
  HIP_NEW(hipSetDevice(device));
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) s->n_simd = 4 * cus;
  }
  HIP_NEW(hipStreamCreateWithFlags(&s->own_stream, hipStreamNonBlocking));
  s->stream = s->own_stream;
  const size_t n = (size_t)n_envs;
  HIP_NEW(hipMalloc(&s->d_model, sizeof(DevModel) + sizeof(LinkRec) * kMaxLinks));
  {
    std::string cwhy = build_collision_points(s->hm, s->cp);
    if (!cwhy.empty()) return cleanup(RCSH_ERR_MODEL, cwhy);
    s->cp_class.assign(s->cp.geom.size(), 0);
    cwhy = build_contact_table(s->hm, s->dm, s->cp.has_plane ? s->cp.plane_geom : -1, s->cgeoms, s->cverts, s->contact_overflow, &s->cgeoms_dropped);
    if (!cwhy.empty()) return cleanup(RCSH_ERR_MODEL, cwhy);
    if (s->cp.has_plane) s->plane_mu = s->hm.geom_friction[3 * (size_t)s->cp.plane_geom];
    if (!s->cp.geom.empty()) {
      HIP_NEW(hipMalloc(&s->d_coll_xyzr, sizeof(double) * s->cp.xyzr.size()));
      HIP_NEW(hipMalloc(&s->d_coll_cls, s->cp_class.size()));
      HIP_NEW(hipMemcpy(s->d_coll_xyzr, s->cp.xyzr.data(), sizeof(double) * s->cp.xyzr.size(), hipMemcpyHostToDevice));
      HIP_NEW(hipMemcpy(s->d_coll_cls, s->cp_class.data(), s->cp_class.size(), hipMemcpyHostToDevice));
    }
  }
  HIP_NEW(hipMalloc(&s->S, sizeof(double) * n * s->nfields));
  HIP_NEW(hipMalloc(&s->flags, sizeof(uint32_t) * n));
  HIP_NEW(hipMalloc(&s->conv, sizeof(int32_t) * n));
  HIP_NEW(hipMalloc(&s->d_stage, sizeof(double) * n * kStageWidth));
  HIP_NEW(hipMalloc(&s->d_stage2, sizeof(double) * n * 32));
  HIP_NEW(hipMalloc(&s->d_bytes, n * 16));
  HIP_NEW(hipMalloc(&s->d_mask, n));
  HIP_NEW(hipMalloc(&s->d_ints, sizeof(int32_t) * n));
  HIP_NEW(hipMalloc(&s->d_floats, sizeof(float) * n));
  HIP_NEW(hipMemsetAsync(s->S, 0, sizeof(double) * n * s->nfields, s->stream));
  HIP_NEW(hipMemsetAsync(s->conv, 0, sizeof(int32_t) * n, s->stream));
  // Sim::converged starts true (reference src/sim/sim.h:70); SimRobotState.ik_success starts true
  {
    std::vector<uint32_t> f(n, kConverged | kIkSuccess);
    HIP_NEW(hipMemcpyAsync(s->flags, f.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, s->stream));
    HIP_NEW(hipStreamSynchronize(s->stream));
  }
  // mjData starts at qpos0
  {
    std::vector<double> q0 = tile(s->dm.qpos0, s->nl, n_envs);
    int rc = scatter_host(s, field_of(s, "qpos"), s->nl, q0.data(), nullptr);
    if (!rc) rc = scatter_host(s, field_of(s, "qpre"), s->nl, q0.data(), nullptr);
    if (rc) return cleanup(rc, g_err);
  }
  if (upload_model(s)) return cleanup(RCSH_ERR_DEVICE, g_err);
  if (const char* cc = std::getenv("RCSH_CONTACT_CHECK")) s->contact_check = std::atoi(cc) != 0;
  if (const char* rf = std::getenv("RCSH_RENDER_F64")) s->render_f64 = std::atoi(rf) != 0;
  if (upload_contact_table(s)) return cleanup(RCSH_ERR_DEVICE, g_err);  // (the contact check's pair tables exist before any robot is attached)
#undef HIP_NEW
  *out = s;
  return RCSH_OK;
}

void rcsh_sim_destroy(rcsh_sim* s) {
  if (!s) return;
  hipSetDevice(s->device);
  if (s->stream) hipStreamSynchronize(s->stream);
  if (s->comm || s->copy) rcsh_comm_destroy(s);
  for (auto e : s->ev_start) hipEventDestroy(e);
  if (s->order_ev) hipEventDestroy(s->order_ev);
  for (auto e : s->ev_stop) hipEventDestroy(e);
  hipFree(s->d_model); hipFree(s->d_coll_xyzr); hipFree(s->d_coll_cls); hipFree(s->S); hipFree(s->flags); hipFree(s->conv);
  hipFree(s->d_cgeoms); hipFree(s->d_cverts); hipFree(s->d_pairs); hipFree(s->d_chk_geoms); hipFree(s->d_chk_ent); hipFree(s->d_lev);
  hipFree(s->d_slack);
  if (s->esc_stream) { hipStreamSynchronize(s->esc_stream); hipStreamDestroy(s->esc_stream); }
  if (s->esc_ev_prev) hipEventDestroy(s->esc_ev_prev);
  if (s->esc_ev_old) hipEventDestroy(s->esc_ev_old);
  if (s->esc_ev_started) hipEventDestroy(s->esc_ev_started);
  if (s->h_esc_hint) hipHostFree((void*)s->h_esc_hint);
  hipFree(s->d_esc); hipFree(s->d_esc_ctr); hipFree(s->d_snap); hipFree(s->d_snap_flags); hipFree(s->d_snap_conv);
  hipFree(s->rend.last); hipFree(s->rend.snap); hipFree(s->rend.count);
  hipFree(s->d_boxtask); hipFree(s->d_rshapes); hipFree(s->d_rplanes); hipFree(s->d_rcolours); hipFree(s->d_frames); hipFree(s->d_wframes); hipFree(s->d_image);
  hipFree(s->d_redge_planes); hipFree(s->d_redge_verts); hipFree(s->d_rviews);
  if (s->h_pin) hipHostFree(s->h_pin);
  hipFree(s->d_stage); hipFree(s->d_stage2); hipFree(s->d_bytes); hipFree(s->d_mask); hipFree(s->d_ints); hipFree(s->d_floats);
  if (s->own_stream) hipStreamDestroy(s->own_stream);
  delete s;
}

int rcsh_sim_num_envs(const rcsh_sim* s) { return s ? s->n : 0; }
int rcsh_sim_nq(const rcsh_sim* s) { return s ? s->nl : 0; }
int rcsh_sim_nu(const rcsh_sim* s) { return s ? (int)s->act_slot.size() : 0; }
void* rcsh_sim_stream(rcsh_sim* s) { return s ? (void*)s->stream : nullptr; }

int rcsh_sim_set_stream(rcsh_sim* s, void* hip_stream) {
  REQUIRE_SIM(s);
  HIP_TRY(hipStreamSynchronize(s->stream));
  s->stream = hip_stream ? (hipStream_t)hip_stream : s->own_stream;
  return RCSH_OK;
}

int rcsh_sim_wait_for(rcsh_sim* s, rcsh_sim* producer) {
  REQUIRE_SIM(s);
  if (!producer) return fail(RCSH_ERR_ARG, "null producer handle");
  if (producer == s || producer->stream == s->stream) return RCSH_OK;  // same stream: already ordered
  HIP_TRY(hipSetDevice(producer->device));
  if (!producer->order_ev) HIP_TRY(hipEventCreateWithFlags(&producer->order_ev, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(producer->order_ev, producer->stream));
  HIP_TRY(hipStreamWaitEvent(s->stream, producer->order_ev, 0));
  return RCSH_OK;
}

int rcsh_sim_set_kernel(rcsh_sim* s, int32_t variant) {
  REQUIRE_SIM(s);
  if (variant == RCSH_KERNEL_LANE) return fail(RCSH_ERR_ARG, "the one-lane-per-environment kernel was removed (ABI 2): every scene runs on the team kernel");
  if (variant != RCSH_KERNEL_AUTO && variant != RCSH_KERNEL_TEAM && variant != RCSH_KERNEL_TEAM_OCC2) return fail(RCSH_ERR_ARG, "unknown kernel variant");
  s->kernel = variant;
  return RCSH_OK;
}

int rcsh_sim_synchronize(rcsh_sim* s) {
  REQUIRE_SIM(s);
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RCSH_OK;
}

int rcsh_sim_set_config(rcsh_sim* s, int32_t async_control, int32_t realtime, int32_t frequency, int32_t max_convergence_steps) {
  REQUIRE_SIM(s);
  if (frequency <= 0) return fail(RCSH_ERR_ARG, "SimConfig.frequency must be positive");
  s->sim = SimCfg{async_control, realtime, frequency, max_convergence_steps};
  return RCSH_OK;
}
int rcsh_sim_get_config(const rcsh_sim* s, int32_t* a, int32_t* r, int32_t* f, int32_t* m) {
  REQUIRE_SIM(s);
  if (a) *a = s->sim.async_control;
  if (r) *r = s->sim.realtime;
  if (f) *f = s->sim.frequency;
  if (m) *m = s->sim.max_convergence_steps;
  return RCSH_OK;
}

int rcsh_sim_step(rcsh_sim* s, int64_t k) {
  REQUIRE_SIM(s);
  if (k < 0) return fail(RCSH_ERR_ARG, "step count must be non-negative");
  HIP_TRY(hipSetDevice(s->device));
  RunOp op{};
  op.nsteps = (int32_t)k;
  int rc = launch_run(s, op, false);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RCSH_OK;
}

int rcsh_sim_step_until_convergence(rcsh_sim* s) {
  REQUIRE_SIM(s);
  HIP_TRY(hipSetDevice(s->device));
  RunOp op{};
  op.nsteps = -1;
  int rc = launch_run(s, op, false);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RCSH_OK;
}

int rcsh_sim_is_converged(rcsh_sim* s, uint8_t* converged, int32_t* steps) {
  REQUIRE_SIM(s);
  int rc = flag_host(s, kConverged, converged);
  if (rc) return rc;
  if (steps) {
    HIP_TRY(hipMemcpyAsync(steps, s->conv, sizeof(int32_t) * s->n, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
  }
  return RCSH_OK;
}

int rcsh_sim_reset(rcsh_sim* s, const uint8_t* mask) {
  REQUIRE_SIM(s);
  // mj_resetData: qpos := qpos0, qvel := 0, ctrl := 0, time := 0; reset_callbacks: timestamps := 0
  std::vector<double> z((size_t)s->n * 32, 0.0);
  static_assert(kBoxState - kBoxX <= 32, "the zero block covers the coupled solve's warm start");
  std::vector<double> q0 = tile(s->dm.qpos0, s->nl, s->n);
  int rc = scatter_host(s, field_of(s, "qpos"), s->nl, q0.data(), mask);
  if (!rc) rc = scatter_host(s, field_of(s, "qpre"), s->nl, q0.data(), mask);
  if (!rc) rc = scatter_host(s, field_of(s, "qvel"), s->nl, z.data(), mask);
  if (!rc) rc = scatter_host(s, field_of(s, "ctrl"), s->nu, z.data(), mask);
  if (!rc) rc = scatter_host(s, field_of(s, "time"), 1, z.data(), mask);
  if (!rc) rc = scatter_host(s, field_of(s, "cb"), 6, z.data(), mask);
  if (!rc) rc = scatter_host(s, field_of(s, "xs"), s->nl, z.data(), mask);  // (mj_resetData: qacc_warmstart := 0)
  if (!rc) rc = flags_update_host(s, 0, kContactOverflow | kContactUnresolved | kContactResolved | kEscQuiet, mask);
  if (!rc) rc = scatter_host(s, field_of(s, "sep"), kCheckSep, z.data(), mask);
  if (!rc && s->d_esc) {
    // per-environment escalation: a reset environment starts over on the lean kernel
    const size_t nw = ((size_t)s->n + 63) / 64;
    std::vector<uint64_t> w(3 * nw, 0);
    if (mask) {
      HIP_TRY(hipMemcpyAsync(w.data(), s->d_esc, sizeof(uint64_t) * nw, hipMemcpyDeviceToHost, s->stream));
      HIP_TRY(hipStreamSynchronize(s->stream));
      for (int e = 0; e < s->n; ++e)
        if (mask[e]) w[e >> 6] &= ~(1ull << (e & 63));
    }
    HIP_TRY(hipMemcpyAsync(s->d_esc, w.data(), sizeof(uint64_t) * 3 * nw, hipMemcpyHostToDevice, s->stream));
    uint32_t ctr[4] = {0, 0, 0, 0};  // (RunOp::esc_ctr: [1] has to say how many are escalated -- a role-2 launch trusts a zero)
    for (size_t i = 0; i < nw; ++i) ctr[1] += (uint32_t)__builtin_popcountll(w[i]);
    HIP_TRY(hipMemcpyAsync(s->d_esc_ctr, ctr, sizeof(ctr), hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
  }
  if (!rc && s->box.present) {
    std::vector<double> b0((size_t)s->n * kBoxState, 0.0);
    for (int e = 0; e < s->n; ++e)
      for (int k = 0; k < 7; ++k) b0[(size_t)e * kBoxState + k] = b0[(size_t)e * kBoxState + kBoxPre + k] = s->box.qpos0[k];
    rc = scatter_host(s, field_of(s, "box"), kBoxState, b0.data(), mask);
  } else if (!rc && s->box.resolve) {
    // contacts resolved without a free body: the phantom box's slot carries the coupled solve's warm start from launch to launch
    // (mjData.qacc_warmstart); mj_resetData zeroes it, so that a reset sim reproduces a fresh one bit for bit (advisor, round 3)
    rc = scatter_host(s, field_of(s, "box") + kBoxX, kBoxState - kBoxX, z.data(), mask);
  }
  if (!rc && s->rend.ncam > 0) {
    // reset_callbacks (sim.cpp:131-137): the cameras' clocks go back to -seconds_between_calls
    const size_t n = (size_t)s->n;
    std::vector<double> last(kMaxRateCams * n);
    HIP_TRY(hipMemcpyAsync(last.data(), s->rend.last, sizeof(double) * last.size(), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    for (int c = 0; c < s->rend.ncam; ++c)
      for (size_t e = 0; e < n; ++e)
        if (!mask || mask[e]) last[c * n + e] = -s->rend.period[c];
    HIP_TRY(hipMemcpyAsync(s->rend.last, last.data(), sizeof(double) * last.size(), hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
  }
  return rc;
}

int rcsh_sim_add_robot(rcsh_sim* s, const rcsh_robot_desc* r) {
  REQUIRE_SIM(s);
  if (!r) return fail(RCSH_ERR_ARG, "null robot description");
  if (s->robot.present) return fail(RCSH_ERR_STATE, "a robot is already attached to this sim");
  if (r->dof != s->narm) return fail(RCSH_ERR_MODEL, "robot dof does not match the arm chain of the scene");
  for (int i = 0; i < r->dof; ++i) {
    if (r->joint_ids[i] != i) return fail(RCSH_ERR_MODEL, "robot joints must be the arm chain in order");
    const int u = r->actuator_ids[i];
    if (u < 0 || u >= (int)s->act_slot.size() || s->act_slot[u] != i)
      return fail(RCSH_ERR_MODEL, "robot actuators must drive the arm joints one to one");
  }
  std::string why = attach_robot_frames(s->hm, s->dm, r->attachment_site, r->base_body);
  if (!why.empty()) return fail(RCSH_ERR_MODEL, why);
  int rc = upload_model(s);
  if (rc) return rc;
  s->robot.present = 1;
  s->robot.conv_registered = r->register_convergence_callback;
  s->robot.tolerance = r->joint_rotational_tolerance;
  s->robot.period = r->seconds_between_callbacks;
  std::memcpy(s->robot.tcp, r->tcp_offset, sizeof(s->robot.tcp));
  for (int i = 0; i < r->dof; ++i) s->robot.q_home[i] = r->q_home[i];
  for (int c = 0; c < r->n_collision_geoms; ++c) {
    const int g = r->collision_geom_ids[c];
    if (g < 0 || g >= s->hm.ngeom) return fail(RCSH_ERR_NAME, "arm collision geom id out of range");
    // a collision callback must not be registered on a geom the geom-geom detection cannot see (advisor, round 3): its contacts
    // with other geoms would silently never raise the flag
    for (int d : s->cgeoms_dropped)
      if (d == g) return fail(RCSH_ERR_MODEL, "arm collision geom " + std::to_string(g) + " is not in the contact table (" + s->contact_overflow + "): its geom-geom collisions would go undetected");
    for (size_t k = 0; k < s->cp.geom.size(); ++k)
      if (s->cp.geom[k] == g) s->cp_class[k] |= 1u;
    for (auto& cg : s->cgeoms)
      if (cg.geom_id == g) cg.cls |= 1;
  }
  rc = upload_coll_classes(s);
  if (!rc) rc = upload_contact_table(s);
  if (rc) return rc;
  return rcsh_robot_reset(s, nullptr);  // SimRobot ctor ends with m_reset()
}

int rcsh_robot_set_joints_hard(rcsh_sim* s, const double* q, const uint8_t* mask) {
  REQUIRE_SIM(s); REQUIRE_ROBOT(s);
  int rc = scatter_host(s, field_of(s, "qpos"), s->narm, q, mask);
  if (!rc) rc = scatter_host(s, field_of(s, "ctrl"), s->narm, q, mask);
  return rc;
}

int rcsh_robot_reset(rcsh_sim* s, const uint8_t* mask) {
  REQUIRE_SIM(s); REQUIRE_ROBOT(s);
  std::vector<double> q = tile(s->robot.q_home, s->narm, s->n);
  return rcsh_robot_set_joints_hard(s, q.data(), mask);
}

int rcsh_robot_set_joint_position(rcsh_sim* s, const double* q, const uint8_t* mask) {
  REQUIRE_SIM(s); REQUIRE_ROBOT(s);
  // target := q, previous := current q, ctrl := q, is_moving := true, is_arrived := false
  std::vector<double> cur((size_t)s->n * s->narm);
  int rc = gather_host(s, field_of(s, "qpos"), s->narm, cur.data());
  if (!rc) rc = scatter_host(s, field_of(s, "target"), s->narm, q, mask);
  if (!rc) rc = scatter_host(s, field_of(s, "prevq"), s->narm, cur.data(), mask);
  if (!rc) rc = scatter_host(s, field_of(s, "ctrl"), s->narm, q, mask);
  if (!rc) rc = flags_update_host(s, kIsMoving, kIsArrived, mask);
  return rc;
}

int rcsh_robot_move_home(rcsh_sim* s, const uint8_t* mask) {
  REQUIRE_SIM(s); REQUIRE_ROBOT(s);
  std::vector<double> q = tile(s->robot.q_home, s->narm, s->n);
  return rcsh_robot_set_joint_position(s, q.data(), mask);
}

int rcsh_robot_get_joint_position(rcsh_sim* s, double* q) {
  REQUIRE_SIM(s); REQUIRE_ROBOT(s);
  return gather_host(s, field_of(s, "qpos"), s->narm, q);
}

int rcsh_robot_get_cartesian_position(rcsh_sim* s, double* pose) {
  REQUIRE_SIM(s); REQUIRE_ROBOT(s);
  HIP_TRY(hipSetDevice(s->device));
  RunOp op{};
  op.nsteps = 0;
  op.write_obs = 1;
  op.observe_only = 1;
  op.obs = s->d_stage2;
  int rc = launch_run(s, op, false);
  if (rc) return rc;
  const int ow = 14 + s->narm;
  std::vector<double> obs((size_t)s->n * ow);
  HIP_TRY(hipMemcpyAsync(obs.data(), s->d_stage2, sizeof(double) * obs.size(), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  for (int e = 0; e < s->n; ++e) std::memcpy(pose + 7 * e, &obs[(size_t)e * ow], 7 * sizeof(double));
  return RCSH_OK;
}

int rcsh_robot_get_base_pose(rcsh_sim* s, double* pose) {
  REQUIRE_SIM(s); REQUIRE_ROBOT(s);
  for (int e = 0; e < s->n; ++e) {
    double* p = pose + 7 * e;
    p[0] = s->dm.base_pos[0]; p[1] = s->dm.base_pos[1]; p[2] = s->dm.base_pos[2];
    p[3] = s->dm.base_quat[1]; p[4] = s->dm.base_quat[2]; p[5] = s->dm.base_quat[3]; p[6] = s->dm.base_quat[0];
  }
  return RCSH_OK;
}

namespace {
int launch_cartesian(rcsh_sim* s, const CartOp& op) {
  Params P = make_params(s);
  hipError_t err = hipSuccess;
  bool ok = dispatch_topology(s->narm, s->grip, [&](auto topo) {
    using T = decltype(topo);
    hipLaunchKernelGGL(k_cartesian_team<T>, dim3(((s->n + 31) / 32) * 8), dim3(64), 0, s->stream, P, op);
    err = hipGetLastError();
  });
  if (!ok) return fail(RCSH_ERR_MODEL, "no kernel instantiated for this archetype");
  if (err != hipSuccess) return fail(RCSH_ERR_DEVICE, std::string("k_cartesian launch: ") + hipGetErrorString(err));
  return RCSH_OK;
}
}  // namespace

int rcsh_robot_set_cartesian_position(rcsh_sim* s, const double* pose, const uint8_t* mask) {
  REQUIRE_SIM(s); REQUIRE_ROBOT(s);
  if (!pose) return fail(RCSH_ERR_ARG, "null pose");
  HIP_TRY(hipSetDevice(s->device));
  const uint8_t* dm = nullptr;
  int rc = upload_mask(s, mask, &dm);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(s->d_stage, pose, sizeof(double) * s->n * 7, hipMemcpyHostToDevice, s->stream));
  CartOp op{};
  op.env_layer = 0;
  op.mask = dm;
  op.action = s->d_stage;
  rc = launch_cartesian(s, op);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RCSH_OK;
}

namespace {
int run_ik(rcsh_sim* s, const double* pose, const double* q0, const double* tcp7, double* out, int out_width, uint8_t* success,
           int32_t* iterations, int forward) {
  HIP_TRY(hipSetDevice(s->device));
  const size_t n = s->n;
  double* d_pose = s->d_stage;            // n*7
  double* d_q0 = s->d_stage + n * 8;      // n*narm
  double* d_tcp = s->d_stage + n * 16;    // 7
  double* d_out = s->d_stage2;            // n*out_width
  if (pose) HIP_TRY(hipMemcpyAsync(d_pose, pose, sizeof(double) * n * 7, hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipMemcpyAsync(d_q0, q0, sizeof(double) * n * s->narm, hipMemcpyHostToDevice, s->stream));
  if (tcp7) HIP_TRY(hipMemcpyAsync(d_tcp, tcp7, sizeof(double) * 7, hipMemcpyHostToDevice, s->stream));
  Params P = make_params(s);
  hipError_t err = hipSuccess;
  dispatch_topology(s->narm, s->grip, [&](auto topo) {
    using T = decltype(topo);
    if (forward)
      hipLaunchKernelGGL(k_ik<T>, dim3((s->n + 63) / 64), dim3(64), 0, s->stream, P, (const double*)d_pose, (const double*)d_q0,
                         tcp7 ? (const double*)d_tcp : (const double*)nullptr, d_out, s->d_bytes, s->d_ints, forward);
    else
      hipLaunchKernelGGL(k_ik_team<T>, dim3((s->n + 3) / 4), dim3(64), 0, s->stream, P, (const double*)d_pose, (const double*)d_q0,
                         tcp7 ? (const double*)d_tcp : (const double*)nullptr, d_out, s->d_bytes, s->d_ints);
    err = hipGetLastError();
  });
  if (err != hipSuccess) return fail(RCSH_ERR_DEVICE, std::string("k_ik launch: ") + hipGetErrorString(err));
  HIP_TRY(hipMemcpyAsync(out, d_out, sizeof(double) * n * out_width, hipMemcpyDeviceToHost, s->stream));
  if (success) HIP_TRY(hipMemcpyAsync(success, s->d_bytes, n, hipMemcpyDeviceToHost, s->stream));
  if (iterations) HIP_TRY(hipMemcpyAsync(iterations, s->d_ints, sizeof(int32_t) * n, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RCSH_OK;
}
}  // namespace

int rcsh_ik_inverse(rcsh_sim* s, const double* pose, const double* q0, const double* tcp7, double* q, uint8_t* success,
                    int32_t* iterations) {
  REQUIRE_SIM(s); REQUIRE_ROBOT(s);
  if (!pose || !q0 || !q) return fail(RCSH_ERR_ARG, "null argument");
  return run_ik(s, pose, q0, tcp7, q, s->nl, success, iterations, 0);
}
int rcsh_ik_forward(rcsh_sim* s, const double* q0, const double* tcp7, double* pose) {
  REQUIRE_SIM(s); REQUIRE_ROBOT(s);
  if (!q0 || !pose) return fail(RCSH_ERR_ARG, "null argument");
  return run_ik(s, nullptr, q0, tcp7, pose, 7, nullptr, nullptr, 1);
}

int rcsh_robot_get_state(rcsh_sim* s, uint8_t* ik_success, uint8_t* collision, uint8_t* is_moving, uint8_t* is_arrived,
                         double* previous_angles, double* target_angles) {
  REQUIRE_SIM(s); REQUIRE_ROBOT(s);
  int rc = flag_host(s, kIkSuccess, ik_success);
  if (!rc) rc = flag_host(s, kRobotCollision, collision);
  if (!rc) rc = flag_host(s, kIsMoving, is_moving);
  if (!rc) rc = flag_host(s, kIsArrived, is_arrived);
  if (!rc && previous_angles) rc = gather_host(s, field_of(s, "prevq"), s->narm, previous_angles);
  if (!rc && target_angles) rc = gather_host(s, field_of(s, "target"), s->narm, target_angles);
  return rc;
}

int rcsh_sim_add_gripper(rcsh_sim* s, const rcsh_gripper_desc* g) {
  REQUIRE_SIM(s);
  if (!g) return fail(RCSH_ERR_ARG, "null gripper description");
  if (!s->grip) return fail(RCSH_ERR_MODEL, "scene has no two-finger gripper");
  if (s->gripcfg.present) return fail(RCSH_ERR_STATE, "a gripper is already attached to this sim");
  if (g->joint_id != s->narm && g->joint_id != s->narm + 1) return fail(RCSH_ERR_MODEL, "gripper joint must be a finger joint");
  if (g->actuator_id < 0 || g->actuator_id >= (int)s->act_slot.size() || s->act_slot[g->actuator_id] != s->narm)
    return fail(RCSH_ERR_MODEL, "gripper actuator must be the finger tendon actuator");
  s->gripcfg.present = 1;
  s->gripcfg.finger = g->joint_id - s->narm;
  s->gripcfg.eps_inner = g->epsilon_inner; s->gripcfg.eps_outer = g->epsilon_outer;
  s->gripcfg.period = g->seconds_between_callbacks;
  s->gripcfg.max_act = g->max_actuator_width; s->gripcfg.min_act = g->min_actuator_width;
  s->gripcfg.max_joint = g->max_joint_width; s->gripcfg.min_joint = g->min_joint_width;
  // SimGripper::collision_callback (SimGripper.cpp:108-130) on plane contacts {geom[0] = plane, geom[1] = robot geom}:
  // counted when geom[1] is a gripper collision geom and not in the ignore list (the finger-finger skip cannot
  // apply: the plane is no finger geom; the ignore test reads geom[1] twice, reference quirk Q6 -- same set here)
  for (int c = 0; c < g->n_collision_geoms; ++c) {
    const int gid = g->collision_geom_ids[c];
    if (gid < 0 || gid >= s->hm.ngeom) return fail(RCSH_ERR_NAME, "gripper collision geom id out of range");
    for (int d : s->cgeoms_dropped)
      if (d == gid) return fail(RCSH_ERR_MODEL, "gripper collision geom " + std::to_string(gid) + " is not in the contact table (" + s->contact_overflow + "): its geom-geom collisions would go undetected");
    for (auto& cg : s->cgeoms)
      if (cg.geom_id == gid) cg.cls |= 16;
    bool ignored = false;
    for (int q = 0; q < g->n_ignored_geoms; ++q) ignored = ignored || g->ignored_geom_ids[q] == gid;
    if (ignored) continue;
    for (size_t k = 0; k < s->cp.geom.size(); ++k)
      if (s->cp.geom[k] == gid) s->cp_class[k] |= 2u;
    for (auto& cg : s->cgeoms)
      if (cg.geom_id == gid) cg.cls |= 2;
  }
  for (int c = 0; c < g->n_finger_geoms; ++c)
    for (auto& cg : s->cgeoms)
      if (cg.geom_id == g->finger_geom_ids[c]) cg.cls |= 4;
  for (int q = 0; q < g->n_ignored_geoms; ++q)
    for (auto& cg : s->cgeoms)
      if (cg.geom_id == g->ignored_geom_ids[q]) cg.cls |= 8;
  {
    int rc = upload_coll_classes(s);
    if (!rc) rc = upload_contact_table(s);
    if (rc) return rc;
  }
  return rcsh_gripper_reset(s, nullptr);  // SimGripper ctor ends with m_reset()
}

int rcsh_gripper_reset(rcsh_sim* s, const uint8_t* mask) {
  REQUIRE_SIM(s); REQUIRE_GRIPPER(s);
  std::vector<double> z((size_t)s->n * 2, 0.0);
  std::vector<double> qj((size_t)s->n, s->gripcfg.max_joint), ca((size_t)s->n, s->gripcfg.max_act);
  int rc = scatter_host(s, field_of(s, "grip"), 2, z.data(), mask);
  if (!rc) rc = scatter_host(s, field_of(s, "qpos") + s->narm + s->gripcfg.finger, 1, qj.data(), mask);
  if (!rc) rc = scatter_host(s, field_of(s, "ctrl") + s->narm, 1, ca.data(), mask);
  if (!rc) rc = flags_update_host(s, 0, kGripMoving | kGripCollision, mask);
  return rc;
}

int rcsh_gripper_set_normalized_width(rcsh_sim* s, const double* width, double force, const uint8_t* mask) {
  REQUIRE_SIM(s); REQUIRE_GRIPPER(s);
  if (force < 0) return fail(RCSH_ERR_ARG, "width must be between 0 and 1, force must be positive");
  std::vector<double> c((size_t)s->n);
  for (int e = 0; e < s->n; ++e) {
    if (mask && !mask[e]) continue;
    if (width[e] < 0 || width[e] > 1) return fail(RCSH_ERR_ARG, "width must be between 0 and 1, force must be positive");
    c[e] = width[e] * (s->gripcfg.max_act - s->gripcfg.min_act) + s->gripcfg.min_act;
  }
  int rc = scatter_host(s, field_of(s, "grip"), 1, width, mask);
  if (!rc) rc = scatter_host(s, field_of(s, "ctrl") + s->narm, 1, c.data(), mask);
  return rc;
}

int rcsh_gripper_get_normalized_width(rcsh_sim* s, double* width) {
  REQUIRE_SIM(s); REQUIRE_GRIPPER(s);
  int rc = gather_host(s, field_of(s, "qpos") + s->narm + s->gripcfg.finger, 1, width);
  if (rc) return rc;
  for (int e = 0; e < s->n; ++e) {
    double w = (width[e] - s->gripcfg.min_joint) / (s->gripcfg.max_joint - s->gripcfg.min_joint);
    width[e] = w < 0 ? 0 : (w > 1 ? 1 : w);
  }
  return RCSH_OK;
}

int rcsh_gripper_is_grasped(rcsh_sim* s, uint8_t* grasped) {
  REQUIRE_SIM(s); REQUIRE_GRIPPER(s);
  std::vector<double> w(s->n), st((size_t)s->n * 2);
  int rc = rcsh_gripper_get_normalized_width(s, w.data());
  if (!rc) rc = gather_host(s, field_of(s, "grip"), 2, st.data());
  if (rc) return rc;
  for (int e = 0; e < s->n; ++e) {
    const double lc = st[2 * e];
    grasped[e] = (lc - s->gripcfg.eps_inner < w[e]) && (w[e] < lc + s->gripcfg.eps_outer);
  }
  return RCSH_OK;
}

int rcsh_gripper_get_state(rcsh_sim* s, double* last_commanded_width, uint8_t* is_moving, double* last_width, uint8_t* collision) {
  REQUIRE_SIM(s); REQUIRE_GRIPPER(s);
  std::vector<double> st((size_t)s->n * 2);
  int rc = gather_host(s, field_of(s, "grip"), 2, st.data());
  if (rc) return rc;
  for (int e = 0; e < s->n; ++e) {
    if (last_commanded_width) last_commanded_width[e] = st[2 * e];
    if (last_width) last_width[e] = st[2 * e + 1];
  }
  rc = flag_host(s, kGripMoving, is_moving);
  if (!rc) rc = flag_host(s, kGripCollision, collision);
  return rc;
}

int rcsh_sim_get_qpos(rcsh_sim* s, double* q) { REQUIRE_SIM(s); return gather_host(s, field_of(s, "qpos"), s->nl, q); }
int rcsh_sim_get_qvel(rcsh_sim* s, double* q) { REQUIRE_SIM(s); return gather_host(s, field_of(s, "qvel"), s->nl, q); }
int rcsh_sim_get_time(rcsh_sim* s, double* t) { REQUIRE_SIM(s); return gather_host(s, field_of(s, "time"), 1, t); }
int rcsh_sim_get_ctrl(rcsh_sim* s, double* c) {
  REQUIRE_SIM(s);
  std::vector<double> slots((size_t)s->n * s->nu);
  int rc = gather_host(s, field_of(s, "ctrl"), s->nu, slots.data());
  if (rc) return rc;
  const int nu = (int)s->act_slot.size();
  for (int e = 0; e < s->n; ++e)
    for (int u = 0; u < nu; ++u) c[(size_t)e * nu + u] = slots[(size_t)e * s->nu + s->act_slot[u]];
  return RCSH_OK;
}
int rcsh_sim_set_qpos(rcsh_sim* s, const double* q, const uint8_t* mask) { REQUIRE_SIM(s); return scatter_host(s, field_of(s, "qpos"), s->nl, q, mask); }
int rcsh_sim_set_qvel(rcsh_sim* s, const double* q, const uint8_t* mask) { REQUIRE_SIM(s); return scatter_host(s, field_of(s, "qvel"), s->nl, q, mask); }

// ---- free box of the scene (reference: mjData.joint("box_joint").qpos, python/rcs/envs/sim.py:379-383,399,412)
int rcsh_sim_add_free_box(rcsh_sim* s, const rcsh_free_box_desc* d) {
  REQUIRE_SIM(s);
  if (!d) return fail(RCSH_ERR_ARG, "null free-box description");
  if (s->box.present) return fail(RCSH_ERR_STATE, "a free box is already attached to this sim");
  if (!(s->narm == 7 && (s->grip || s->dm.has_friction != 0)))
    return fail(RCSH_ERR_MODEL, "free bodies are compiled for three archetypes: 7-dof arm + two-finger gripper without dry joint friction (FR3) "
                                "and with it (xArm7 + gripper), 7-dof arm without gripper with it (xArm7)");
  if (s->grip && s->dm.has_friction && !d->resolve_robot_contacts)
    return fail(RCSH_ERR_MODEL, "7-dof arm + gripper with dry joint friction next to a free body: compiled with robot contacts resolved only");
  if (s->dm.has_friction && d->noslip_iterations > 0)
    return fail(RCSH_ERR_MODEL, "the noslip pass over dry joint friction rows is not built: scenes with frictionloss need noslip_iterations = 0");
  if (!d->cone_elliptic) return fail(RCSH_ERR_MODEL, "contacts use elliptic friction cones (option cone=\"elliptic\")");
  if (!(d->mass > 0) || !(d->inertia[0] > 0) || !(d->inertia[1] > 0) || !(d->inertia[2] > 0) || !(d->impratio > 0))
    return fail(RCSH_ERR_ARG, "free box: mass, inertia and impratio must be positive");
  BoxCfg b{};
  b.present = 1;
  b.noslip_iterations = d->noslip_iterations;
  for (int k = 0; k < 7; ++k) b.qpos0[k] = d->qpos0[k];
  b.mass = d->mass; b.inv_mass = 1.0 / d->mass;
  for (int k = 0; k < 3; ++k) { b.inertia[k] = d->inertia[k]; b.inv_inertia[k] = 1.0 / d->inertia[k]; b.size[k] = d->size[k]; }
  b.fr = d->friction[0];
  b.geom_mu = d->geom_friction[0] > 0 ? d->geom_friction[0] : d->friction[0];
  // contacts of the robot's geoms with the floor and the box enter one constraint problem with the robot's own rows
  // (limit / equality rows; dry-friction rows in the xArm7 + gripper archetype)
  b.resolve = (d->resolve_robot_contacts && s->grip && !s->cgeoms.empty()) ? (1 | (d->resolve_robot_contacts & 2)) : 0;  // (bit 1: self contact too)
  if (b.resolve && !s->contact_overflow.empty()) return fail(RCSH_ERR_MODEL, "robot contacts cannot be resolved in this scene: " + s->contact_overflow);
  make_kb(d->solref, d->solimp, s->dm.timestep, b.K, b.B);
  b.imp = make_imp(d->solimp);
  b.inv_impratio = 1.0 / d->impratio;
  b.plane_z = d->plane_z;
  // mjModel.stat.meaninertia: mean diagonal of M(qpos0) over all dofs of the scene
  const int nv = s->nl + 6;
  const double meaninertia = (s->dm.inertia_diag_sum + 3 * d->mass + d->inertia[0] + d->inertia[1] + d->inertia[2]) / nv;
  b.scale = 1.0 / (meaninertia * nv);
  b.noslip_tolerance = d->noslip_tolerance;
  s->box = b;
  HIP_TRY(hipSetDevice(s->device));
  if (int rc = upload_boxtask(s)) return rc;
  if (int rc = upload_contact_table(s)) return rc;
  return rcsh_sim_reset_free_box(s);
}
int rcsh_sim_set_contact_options(rcsh_sim* s, const rcsh_contact_options* o) {
  REQUIRE_SIM(s);
  if (!o) return fail(RCSH_ERR_ARG, "null contact options");
  if (s->box.present) return fail(RCSH_ERR_STATE, "this scene has a free box: its description (rcsh_sim_add_free_box) carries the contact options");
  // (whenever the option is switched -- off, on, or between its forms -- the escalation masks, their counters and the phantom box's warm
  // start begin empty: stale bits of a rollout under another option would send environments to a launch that no longer knows them;
  // advisor, round 5)
  if (s->d_esc) {
    const size_t nw = ((size_t)s->n + 63) / 64;
    HIP_TRY(hipMemsetAsync(s->d_esc, 0, sizeof(uint64_t) * 3 * nw, s->stream));
    HIP_TRY(hipMemsetAsync(s->d_esc_ctr, 0, sizeof(uint32_t) * 4, s->stream));
    int f_box = -1;
    dispatch_topology(s->narm, s->grip, [&](auto topo) { f_box = (int)Lay<decltype(topo)>::BOX; });
    if (f_box >= 0) HIP_TRY(hipMemsetAsync(s->S + (size_t)(f_box + kBoxX) * s->n, 0, sizeof(double) * (size_t)(kBoxState - kBoxX) * s->n, s->stream));
    if (s->d_slack) HIP_TRY(hipMemsetAsync(s->d_slack, 0, sizeof(float) * (size_t)kSlackStride * s->n, s->stream));
  }
  if (!o->resolve_robot_contacts) { s->box = BoxCfg{}; s->esc_mode = false; return RCSH_OK; }
  if (!(s->narm == 7 && s->grip && !s->dm.has_friction)) return fail(RCSH_ERR_MODEL, "contacts of the robot's geoms are resolved for the FR3 + hand archetype (no dry joint friction)");
  if (!o->cone_elliptic) return fail(RCSH_ERR_MODEL, "contacts use elliptic friction cones (option cone=\"elliptic\")");
  if (!(o->impratio > 0)) return fail(RCSH_ERR_ARG, "impratio must be positive");
  if (!s->contact_overflow.empty()) return fail(RCSH_ERR_MODEL, "robot contacts cannot be resolved in this scene: " + s->contact_overflow);
  if (s->cgeoms.empty() || !s->cp.has_plane) return RCSH_OK;  // nothing the robot could touch
  // the phantom box of the contact phase: unit inertia, zero size, parked 1 km above the scene
  BoxCfg b{};
  b.present = 0;
  b.resolve = 1 | (o->resolve_robot_contacts & 2);  // (bit 1: contacts between two geoms of the robot too)
  s->esc_mode = (o->resolve_robot_contacts & 4) != 0;  // (bit 2: environment by environment instead of the whole batch)
  if (s->esc_mode && !s->d_esc) {
    const size_t nw = ((size_t)s->n + 63) / 64;
    HIP_TRY(hipMalloc(&s->d_esc, sizeof(uint64_t) * 3 * nw));
    HIP_TRY(hipMalloc(&s->d_esc_ctr, sizeof(uint32_t) * 4));
    HIP_TRY(hipMalloc(&s->d_snap, sizeof(double) * (size_t)(s->nfields + kMaxRateCams) * s->n));  // (+ the rate-driven cameras' clocks)
    HIP_TRY(hipMalloc(&s->d_snap_flags, sizeof(uint32_t) * s->n));
    HIP_TRY(hipMalloc(&s->d_snap_conv, sizeof(int32_t) * s->n));
    if (const char* e = std::getenv("RCSH_ESC_SPLIT")) s->esc_split = std::atoi(e);
    if (const char* e = std::getenv("RCSH_CONV_CHUNK")) s->conv_chunk = std::atoi(e);
    if (const char* e = std::getenv("RCSH_CONV_GROW")) s->conv_grow = std::atoi(e);
    // (the split form's stream exists only where it is asked for: one more stream in the process changes how the runtime maps streams
    // to hardware queues -- the four sub-batches of `bench.py --robot mixed`, a stream each, ran one after the other with it: 18 -> 8 M)
    if (s->esc_split) {
      int lo = 0, hi = 0;
      HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));  // (hi: the numerically lowest = the highest priority)
      HIP_TRY(hipStreamCreateWithPriority(&s->esc_stream, hipStreamNonBlocking, hi));
      HIP_TRY(hipEventCreateWithFlags(&s->esc_ev_prev, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&s->esc_ev_old, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&s->esc_ev_started, hipEventDisableTiming));
      if (const char* e = std::getenv("RCSH_ESC_SPLIT_MAX")) s->esc_split_max = std::atoi(e);
      void* hp = nullptr;
      HIP_TRY(hipHostMalloc(&hp, 2 * sizeof(uint32_t), hipHostMallocDefault));
      s->h_esc_hint = (volatile uint32_t*)hp;
      s->h_esc_hint[0] = 0; s->h_esc_hint[1] = 0;
      s->esc_seq = 0;
    }
    HIP_TRY(hipMemsetAsync(s->d_esc, 0, sizeof(uint64_t) * 3 * nw, s->stream));
    HIP_TRY(hipMemsetAsync(s->d_esc_ctr, 0, sizeof(uint32_t) * 4, s->stream));
    HIP_TRY(hipMemsetAsync(s->d_snap, 0, sizeof(double) * (size_t)(s->nfields + kMaxRateCams) * s->n, s->stream));
  }
  if ((b.resolve & 2) && !s->d_slack) {
    HIP_TRY(hipMalloc(&s->d_slack, sizeof(float) * (size_t)kSlackStride * s->n));
    HIP_TRY(hipMemsetAsync(s->d_slack, 0, sizeof(float) * (size_t)kSlackStride * s->n, s->stream));  // (no gap known: every pair is looked at)
  }
  b.noslip_iterations = o->noslip_iterations;
  b.qpos0[2] = 1000.0; b.qpos0[3] = 1.0;
  b.mass = b.inv_mass = 1.0;
  for (int k = 0; k < 3; ++k) { b.inertia[k] = b.inv_inertia[k] = 1.0; b.size[k] = 0.0; }
  b.fr = b.geom_mu = 1.0;
  make_kb(o->solref, o->solimp, s->dm.timestep, b.K, b.B);
  b.imp = make_imp(o->solimp);
  b.inv_impratio = 1.0 / o->impratio;
  b.plane_z = s->cp.plane_d;
  b.scale = 1.0 / (s->dm.inertia_diag_sum / s->nl * s->nl);  // 1 / (meaninertia * nv)
  b.noslip_tolerance = o->noslip_tolerance;
  s->box = b;
  if (int rc = upload_boxtask(s)) return rc;
  return upload_contact_table(s);
}
int rcsh_sim_contact_unresolved(rcsh_sim* s, uint8_t* unresolved) {
  REQUIRE_SIM(s);
  if (s->contact_check && s->check_every != 1) {
    // the handle checks less often than every launch: check the present state now, so that the answer is up to date
    RunOp op{};
    op.nsteps = 0;
    op.observe_only = 1;
    op.check = 1;
    if (int rc = launch_run(s, op, false)) return rc;
  }
  return flag_host(s, kContactUnresolved, unresolved);
}
int rcsh_sim_contact_escalated(rcsh_sim* s, uint8_t* now, uint8_t* ever) {
  REQUIRE_SIM(s);
  if (now) {
    std::memset(now, 0, s->n);
    if (s->esc_mode && s->d_esc) {
      const size_t nw = ((size_t)s->n + 63) / 64;
      std::vector<uint64_t> w(nw);
      HIP_TRY(hipMemcpyAsync(w.data(), s->d_esc, sizeof(uint64_t) * nw, hipMemcpyDeviceToHost, s->stream));
      HIP_TRY(hipStreamSynchronize(s->stream));
      for (int e = 0; e < s->n; ++e) now[e] = (uint8_t)((w[e >> 6] >> (e & 63)) & 1u);
    }
  }
  return flag_host(s, kContactResolved, ever);
}
int rcsh_sim_set_contact_check(rcsh_sim* s, int32_t every) {
  REQUIRE_SIM(s);
  if (every < 0) return fail(RCSH_ERR_ARG, "contact check cadence must be >= 0 (0: off, 1: every stepping launch)");
  s->check_every = every;
  s->check_seq = 0;
  return RCSH_OK;
}
int rcsh_sim_contact_check_unchecked_pairs(rcsh_sim* s, int32_t* count) {
  REQUIRE_SIM(s);
  if (!count) return fail(RCSH_ERR_ARG, "count is null");
  *count = s->chk_unchecked;
  return RCSH_OK;
}
int rcsh_sim_contact_table_dropped(rcsh_sim* s, int32_t* geom_ids, int32_t capacity, int32_t* count, char* reason, size_t reason_capacity) {
  REQUIRE_SIM(s);
  if (count) *count = (int32_t)s->cgeoms_dropped.size();
  for (int i = 0; geom_ids && i < capacity && i < (int)s->cgeoms_dropped.size(); ++i) geom_ids[i] = s->cgeoms_dropped[i];
  if (reason && reason_capacity > 0) {
    std::strncpy(reason, s->contact_overflow.c_str(), reason_capacity - 1);
    reason[reason_capacity - 1] = 0;
  }
  return RCSH_OK;
}

int rcsh_sim_reset_free_box(rcsh_sim* s) {
  REQUIRE_SIM(s);
  if (!s->box.present) return fail(RCSH_ERR_STATE, "no free box attached: call rcsh_sim_add_free_box first");
  std::vector<double> b0((size_t)s->n * kBoxState, 0.0);
  for (int e = 0; e < s->n; ++e)
    for (int k = 0; k < 7; ++k) b0[(size_t)e * kBoxState + k] = b0[(size_t)e * kBoxState + kBoxPre + k] = s->box.qpos0[k];
  return scatter_host(s, field_of(s, "box"), kBoxState, b0.data(), nullptr);
}
#define REQUIRE_BOX(s) \
  if (!(s)->box.present) return fail(RCSH_ERR_STATE, "no free box attached: call rcsh_sim_add_free_box first")
int rcsh_sim_get_free_qpos(rcsh_sim* s, double* q) { REQUIRE_SIM(s); REQUIRE_BOX(s); return gather_host(s, field_of(s, "box") + kBoxQ, 7, q); }
int rcsh_sim_get_free_qvel(rcsh_sim* s, double* v) { REQUIRE_SIM(s); REQUIRE_BOX(s); return gather_host(s, field_of(s, "box") + kBoxV, 6, v); }
int rcsh_sim_set_free_qpos(rcsh_sim* s, const double* q, const uint8_t* mask) { REQUIRE_SIM(s); REQUIRE_BOX(s); return scatter_host(s, field_of(s, "box") + kBoxQ, 7, q, mask); }
int rcsh_sim_set_free_qvel(rcsh_sim* s, const double* v, const uint8_t* mask) { REQUIRE_SIM(s); REQUIRE_BOX(s); return scatter_host(s, field_of(s, "box") + kBoxV, 6, v, mask); }

// (the blob begins with a header -- a magic word with the layout's version, n_envs, the number of state fields: a blob of another
// build or handle is refused by what it says, not only when its size happens to differ; advisor, round 5)
constexpr size_t kStateHeader = 16;
constexpr char kStateMagic[8] = {'R', 'C', 'S', 'H', 'S', 'T', '0', '2'};
size_t rcsh_sim_state_bytes(const rcsh_sim* s) {
  if (!s) return 0;
  // (the tail: which environments are on the contact-resolving kernel -- per-environment escalation; a replay from a snapshot takes
  // the same kernels environment by environment as the rollout it was taken from)
  return kStateHeader + (size_t)s->n * (sizeof(double) * s->nfields + sizeof(uint32_t) + sizeof(int32_t)) + sizeof(uint64_t) * (((size_t)s->n + 63) / 64);
}
int rcsh_sim_get_state(rcsh_sim* s, void* blob) {
  REQUIRE_SIM(s);
  if (!blob) return fail(RCSH_ERR_ARG, "null state blob");
  HIP_TRY(hipSetDevice(s->device));
  char* b = static_cast<char*>(blob);
  {
    const uint32_t hn = (uint32_t)s->n, hf = (uint32_t)s->nfields;
    std::memcpy(b, kStateMagic, 8); std::memcpy(b + 8, &hn, 4); std::memcpy(b + 12, &hf, 4);
    b += kStateHeader;
  }
  const size_t ns = sizeof(double) * (size_t)s->n * s->nfields, nf = sizeof(uint32_t) * (size_t)s->n, nc = sizeof(int32_t) * (size_t)s->n;
  HIP_TRY(hipMemcpyAsync(b, s->S, ns, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipMemcpyAsync(b + ns, s->flags, nf, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipMemcpyAsync(b + ns + nf, s->conv, nc, hipMemcpyDeviceToHost, s->stream));
  const size_t ne = sizeof(uint64_t) * (((size_t)s->n + 63) / 64);
  if (s->d_esc) HIP_TRY(hipMemcpyAsync(b + ns + nf + nc, s->d_esc, ne, hipMemcpyDeviceToHost, s->stream));
  else std::memset(b + ns + nf + nc, 0, ne);
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RCSH_OK;
}
int rcsh_sim_set_state(rcsh_sim* s, const void* blob) {
  REQUIRE_SIM(s);
  if (!blob) return fail(RCSH_ERR_ARG, "null state blob");
  HIP_TRY(hipSetDevice(s->device));
  const char* b = static_cast<const char*>(blob);
  {
    uint32_t hn = 0, hf = 0;
    std::memcpy(&hn, b + 8, 4); std::memcpy(&hf, b + 12, 4);
    if (std::memcmp(b, kStateMagic, 8) != 0 || hn != (uint32_t)s->n || hf != (uint32_t)s->nfields)
      return fail(RCSH_ERR_ARG, "state blob of another layout (taken by another build, or from a handle with another scene / n_envs)");
    b += kStateHeader;
  }
  const size_t ns = sizeof(double) * (size_t)s->n * s->nfields, nf = sizeof(uint32_t) * (size_t)s->n, nc = sizeof(int32_t) * (size_t)s->n;
  if (s->d_slack) HIP_TRY(hipMemsetAsync(s->d_slack, 0, sizeof(float) * (size_t)kSlackStride * s->n, s->stream));  // (see scatter_host)
  HIP_TRY(hipMemcpyAsync(s->S, b, ns, hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipMemcpyAsync(s->flags, b + ns, nf, hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipMemcpyAsync(s->conv, b + ns + nf, nc, hipMemcpyHostToDevice, s->stream));
  if (s->d_esc) {
    const size_t ne = sizeof(uint64_t) * (((size_t)s->n + 63) / 64);
    HIP_TRY(hipMemcpyAsync(s->d_esc, b + ns + nf + nc, ne, hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipMemsetAsync(s->d_esc + ne / sizeof(uint64_t), 0, 2 * ne, s->stream));
    uint32_t ctr[4] = {0, 0, 0, 0};  // (RunOp::esc_ctr[1]: how many are escalated)
    const uint64_t* w = reinterpret_cast<const uint64_t*>(b + ns + nf + nc);
    for (size_t i = 0; i < ne / sizeof(uint64_t); ++i) { uint64_t v; std::memcpy(&v, w + i, sizeof(v)); ctr[1] += (uint32_t)__builtin_popcountll(v); }
    HIP_TRY(hipMemcpyAsync(s->d_esc_ctr, ctr, sizeof(ctr), hipMemcpyHostToDevice, s->stream));
  }
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RCSH_OK;
}

// ---- fused Gymnasium loop

int rcsh_env_configure(rcsh_sim* s, const rcsh_env_desc* env) {
  REQUIRE_SIM(s); REQUIRE_ROBOT(s);
  if (!env) return fail(RCSH_ERR_ARG, "null env description");
  if (env->control_mode < RCSH_MODE_JOINTS || env->control_mode > RCSH_MODE_CARTESIAN_TQUAT)
    return fail(RCSH_ERR_ARG, "bad control_mode");
  if (env->relative_to < 0 || env->relative_to > 2) return fail(RCSH_ERR_ARG, "bad relative_to");
  s->env.mode = env->control_mode;
  s->env.relative_to = env->relative_to;
  s->env.binary_gripper = env->binary_gripper;
  s->env.max_mov[0] = env->max_mov[0]; s->env.max_mov[1] = env->max_mov[1];
  for (int i = 0; i < s->narm; ++i) {
    s->env.low[i] = env->joint_low ? env->joint_low[i] : -INFINITY;
    s->env.high[i] = env->joint_high ? env->joint_high[i] : INFINITY;
  }
  s->env_configured = true;
  return RCSH_OK;
}

int rcsh_env_obs_width(const rcsh_sim* s) { return s ? 14 + s->narm : 0; }
int rcsh_env_action_width(const rcsh_sim* s) {
  if (!s) return 0;
  return s->env.mode == RCSH_MODE_JOINTS ? s->narm : (s->env.mode == RCSH_MODE_CARTESIAN_TRPY ? 6 : 7);
}

int rcsh_env_reset_dev(rcsh_sim* s, const uint8_t* mask_dev, double* obs_dev, uint8_t* info_dev, double* gw_dev) {
  REQUIRE_SIM(s); REQUIRE_ROBOT(s);
  if (!s->env_configured) return fail(RCSH_ERR_STATE, "call rcsh_env_configure first");
  HIP_TRY(hipSetDevice(s->device));
  RunOp op{};
  op.do_reset = 1;
  op.nsteps = 1;
  op.write_obs = obs_dev != nullptr;
  op.mask = mask_dev;
  op.obs = obs_dev; op.info = info_dev; op.gripper_width = gw_dev;
  return launch_run(s, op, false);
}

int rcsh_env_step_dev(rcsh_sim* s, const double* action_dev, const float* gripper_dev, double* obs_dev, uint8_t* info_dev,
                      double* gw_dev, int32_t* substeps_dev) {
  REQUIRE_SIM(s); REQUIRE_ROBOT(s);
  if (!s->env_configured) return fail(RCSH_ERR_STATE, "call rcsh_env_configure first");
  if (!action_dev) return fail(RCSH_ERR_ARG, "null action");
  HIP_TRY(hipSetDevice(s->device));
  RunOp op{};
  op.apply_action = 1;
  if (s->env.mode != RCSH_MODE_JOINTS) {
    // Cartesian modes: wrappers' action() + IK run in their own launch, the stepping launch follows on the stream
    CartOp cop{};
    cop.env_layer = 1;
    cop.action = action_dev;
    cop.gripper = gripper_dev;
    int rc = launch_cartesian(s, cop);
    if (rc) return rc;
    op.apply_action = 0;
  }
  // RobotSimWrapper.step (reference python/rcs/envs/sim.py:49-59)
  // (Python's round(): half to even, as std::nearbyint under the default rounding mode)
  op.nsteps = s->sim.async_control ? (int32_t)std::nearbyint(1.0 / s->sim.frequency / s->dm.timestep) : -1;
  op.write_obs = obs_dev != nullptr;
  op.action = action_dev; op.gripper = gripper_dev;
  op.obs = obs_dev; op.info = info_dev; op.gripper_width = gw_dev; op.substeps = substeps_dev;
  op.task = s->pending_task;
  return launch_run(s, op, true);
}

// ---- task layer of the pick-up scene: SimTaskEnvCreator = SimEnvCreator + RandomCubePos under the RobotSimWrapper +
// PickCubeSuccessWrapper on top (reference python/rcs/envs/creators.py:131-187)
int rcsh_env_configure_pick_task(rcsh_sim* s, const rcsh_pick_task_desc* t) {
  REQUIRE_SIM(s); REQUIRE_ROBOT(s);
  if (!t) return fail(RCSH_ERR_ARG, "null task description");
  if (!s->box.present) return fail(RCSH_ERR_STATE, "the pick task needs the scene's free box: call rcsh_sim_add_free_box first");
  s->task.pick_cube = 1;
  for (int k = 0; k < 3; ++k) s->task.ee_home[k] = t->ee_home[k];
  s->task.success_z = t->success_height;
  HIP_TRY(hipSetDevice(s->device));
  return upload_boxtask(s);
}

int rcsh_env_reset_task_dev(rcsh_sim* s, const uint8_t* mask_dev, const double* box_qpos_dev, double* obs_dev, uint8_t* info_dev,
                            double* gw_dev) {
  REQUIRE_SIM(s); REQUIRE_ROBOT(s);
  if (!s->env_configured) return fail(RCSH_ERR_STATE, "call rcsh_env_configure first");
  if (!s->task.pick_cube) return fail(RCSH_ERR_STATE, "call rcsh_env_configure_pick_task first");
  if (!box_qpos_dev) return fail(RCSH_ERR_ARG, "null box pose");
  HIP_TRY(hipSetDevice(s->device));
  RunOp op{};
  op.do_reset = 1;
  op.nsteps = 1;
  op.write_obs = obs_dev != nullptr;
  op.mask = mask_dev;
  op.box_qpos = box_qpos_dev;
  op.obs = obs_dev; op.info = info_dev; op.gripper_width = gw_dev;
  return launch_run(s, op, false);
}

int rcsh_env_step_task_dev(rcsh_sim* s, const double* action_dev, const float* gripper_dev, double* obs_dev, uint8_t* info_dev,
                           double* gw_dev, int32_t* substeps_dev, double* task_dev) {
  REQUIRE_SIM(s);
  if (!s->task.pick_cube) return fail(RCSH_ERR_STATE, "call rcsh_env_configure_pick_task first");
  s->pending_task = task_dev;
  int rc = rcsh_env_step_dev(s, action_dev, gripper_dev, obs_dev, info_dev, gw_dev, substeps_dev);
  s->pending_task = nullptr;
  return rc;
}

int rcsh_env_reset_task(rcsh_sim* s, const uint8_t* mask, const double* box_qpos, double* obs, uint8_t* info, double* gw) {
  REQUIRE_SIM(s);
  if (!box_qpos) return fail(RCSH_ERR_ARG, "null box pose");
  const uint8_t* dm = nullptr;
  int rc = upload_mask(s, mask, &dm);
  if (rc) return rc;
  double* d_box = s->d_stage + (size_t)s->n * 16;
  HIP_TRY(hipMemcpyAsync(d_box, box_qpos, sizeof(double) * s->n * 7, hipMemcpyHostToDevice, s->stream));
  rc = rcsh_env_reset_task_dev(s, dm, d_box, s->d_stage2, s->d_bytes, s->d_stage);
  if (!rc) rc = observe_unmasked(s, mask);
  if (rc) return rc;
  const int ow = 14 + s->narm;
  if (obs) HIP_TRY(hipMemcpyAsync(obs, s->d_stage2, sizeof(double) * s->n * ow, hipMemcpyDeviceToHost, s->stream));
  if (info) HIP_TRY(hipMemcpyAsync(info, s->d_bytes, (size_t)s->n * 8, hipMemcpyDeviceToHost, s->stream));
  if (gw) HIP_TRY(hipMemcpyAsync(gw, s->d_stage, sizeof(double) * s->n, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RCSH_OK;
}

int rcsh_env_step_task(rcsh_sim* s, const double* action, const float* gripper, double* obs, uint8_t* info, double* gw,
                       int32_t* substeps, double* task) {
  REQUIRE_SIM(s);
  if (!s->task.pick_cube) return fail(RCSH_ERR_STATE, "call rcsh_env_configure_pick_task first");
  double* d_task = s->d_stage + (size_t)s->n * 16;
  s->pending_task = d_task;
  int rc = rcsh_env_step(s, action, gripper, obs, info, gw, substeps);
  s->pending_task = nullptr;
  if (rc) return rc;
  if (task) {
    HIP_TRY(hipMemcpyAsync(task, d_task, sizeof(double) * s->n * 9, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
  }
  return RCSH_OK;
}

namespace {
// page-locked staging for the host-buffer entry points: [action | gripper | obs | info | gripper width | substeps]
struct PinLayout { size_t action, gripper, obs, info, gw, sub, total; };
PinLayout pin_layout(const rcsh_sim* s, int aw, int ow) {
  PinLayout L{};
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 63) & ~size_t(63); return at; };
  L.action = take(sizeof(double) * s->n * aw);
  L.gripper = take(sizeof(float) * s->n);
  L.obs = take(sizeof(double) * s->n * ow);
  L.info = take((size_t)s->n * 8);
  L.gw = take(sizeof(double) * s->n);
  L.sub = take(sizeof(int32_t) * s->n);
  L.total = o;
  return L;
}
int pin_reserve(rcsh_sim* s, size_t bytes) {
  if (s->h_pin_bytes >= bytes) return RCSH_OK;
  if (s->h_pin) { HIP_TRY(hipStreamSynchronize(s->stream)); hipHostFree(s->h_pin); s->h_pin = nullptr; s->h_pin_bytes = 0; }
  void* p = nullptr;
  HIP_TRY(hipHostMalloc(&p, bytes, hipHostMallocDefault));
  s->h_pin = (char*)p; s->h_pin_bytes = bytes;
  return RCSH_OK;
}
}  // namespace

int rcsh_env_reset(rcsh_sim* s, const uint8_t* mask, double* obs, uint8_t* info, double* gw) {
  REQUIRE_SIM(s);
  const uint8_t* dm = nullptr;
  int rc = upload_mask(s, mask, &dm);
  if (rc) return rc;
  rc = rcsh_env_reset_dev(s, dm, s->d_stage2, s->d_bytes, s->d_stage);
  if (!rc) rc = observe_unmasked(s, mask);
  if (rc) return rc;
  const int ow = 14 + s->narm;
  const PinLayout L = pin_layout(s, rcsh_env_action_width(s), ow);
  rc = pin_reserve(s, L.total);
  if (rc) return rc;
  if (obs) HIP_TRY(hipMemcpyAsync(s->h_pin + L.obs, s->d_stage2, sizeof(double) * s->n * ow, hipMemcpyDeviceToHost, s->stream));
  if (info) HIP_TRY(hipMemcpyAsync(s->h_pin + L.info, s->d_bytes, (size_t)s->n * 8, hipMemcpyDeviceToHost, s->stream));
  if (gw) HIP_TRY(hipMemcpyAsync(s->h_pin + L.gw, s->d_stage, sizeof(double) * s->n, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (obs) std::memcpy(obs, s->h_pin + L.obs, sizeof(double) * s->n * ow);
  if (info) std::memcpy(info, s->h_pin + L.info, (size_t)s->n * 8);
  if (gw) std::memcpy(gw, s->h_pin + L.gw, sizeof(double) * s->n);
  return RCSH_OK;
}

int rcsh_env_step(rcsh_sim* s, const double* action, const float* gripper, double* obs, uint8_t* info, double* gw, int32_t* substeps) {
  REQUIRE_SIM(s);
  if (!action) return fail(RCSH_ERR_ARG, "null action");
  const int aw = rcsh_env_action_width(s), ow = 14 + s->narm;
  const PinLayout L = pin_layout(s, aw, ow);
  int rc = pin_reserve(s, L.total);
  if (rc) return rc;
  double* d_action = s->d_stage + (size_t)s->n;  // d_stage[0..n) carries gripper widths
  std::memcpy(s->h_pin + L.action, action, sizeof(double) * s->n * aw);
  HIP_TRY(hipMemcpyAsync(d_action, s->h_pin + L.action, sizeof(double) * s->n * aw, hipMemcpyHostToDevice, s->stream));
  if (gripper) {
    std::memcpy(s->h_pin + L.gripper, gripper, sizeof(float) * s->n);
    HIP_TRY(hipMemcpyAsync(s->d_floats, s->h_pin + L.gripper, sizeof(float) * s->n, hipMemcpyHostToDevice, s->stream));
  }
  rc = rcsh_env_step_dev(s, d_action, gripper ? s->d_floats : nullptr, s->d_stage2, s->d_bytes, s->d_stage, s->d_ints);
  if (rc) return rc;
  if (obs) HIP_TRY(hipMemcpyAsync(s->h_pin + L.obs, s->d_stage2, sizeof(double) * s->n * ow, hipMemcpyDeviceToHost, s->stream));
  if (info) HIP_TRY(hipMemcpyAsync(s->h_pin + L.info, s->d_bytes, (size_t)s->n * 8, hipMemcpyDeviceToHost, s->stream));
  if (gw) HIP_TRY(hipMemcpyAsync(s->h_pin + L.gw, s->d_stage, sizeof(double) * s->n, hipMemcpyDeviceToHost, s->stream));
  if (substeps) HIP_TRY(hipMemcpyAsync(s->h_pin + L.sub, s->d_ints, sizeof(int32_t) * s->n, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (obs) std::memcpy(obs, s->h_pin + L.obs, sizeof(double) * s->n * ow);
  if (info) std::memcpy(info, s->h_pin + L.info, (size_t)s->n * 8);
  if (gw) std::memcpy(gw, s->h_pin + L.gw, sizeof(double) * s->n);
  if (substeps) std::memcpy(substeps, s->h_pin + L.sub, sizeof(int32_t) * s->n);
  return RCSH_OK;
}

// ---- depth renderer
int rcsh_sim_set_render_scene(rcsh_sim* s, const rcsh_render_scene_desc* d) {
  REQUIRE_SIM(s);
  if (!d || d->nshape < 1 || d->nshape > kMaxShapes) return fail(RCSH_ERR_ARG, "render scene: between 1 and 32 shapes");
  if (!(d->znear > 0) || !(d->zfar > d->znear)) return fail(RCSH_ERR_ARG, "render scene: need 0 < znear < zfar");
  std::vector<RenderShape> sh(d->nshape);
  for (int i = 0; i < d->nshape; ++i) {
    RenderShape& r = sh[i];
    r.shape = d->shape[i]; r.link = d->link[i]; r.plane_adr = d->plane_adr[i]; r.plane_num = d->plane_num[i];
    if (r.shape < kShapePlane || r.shape > kShapeCapsule) return fail(RCSH_ERR_ARG, "render scene: unknown shape type");
    if (r.link < kLinkFreeBody || r.link >= s->nl) return fail(RCSH_ERR_ARG, "render scene: link index out of range");
    if (r.link == kLinkFreeBody && !s->box.present) return fail(RCSH_ERR_STATE, "render scene: no free box attached");
    if (r.shape == kShapeHull && (r.plane_adr < 0 || r.plane_num < 4 || r.plane_adr + r.plane_num > d->nplanes))
      return fail(RCSH_ERR_ARG, "render scene: hull plane range out of bounds");
    for (int k = 0; k < 3; ++k) { r.pos[k] = d->pos[3 * i + k]; r.size[k] = d->size[3 * i + k]; }
    for (int k = 0; k < 9; ++k) r.rot[k] = d->rot[9 * i + k];
    for (int k = 0; k < 4; ++k) r.sphere[k] = d->sphere[4 * i + k];
  }
  // The outline method (render.h: k_hull_views): each hull's polytope edges, worked out here from its planes, and room for one view
  // record per environment and hull.  RCSH_RENDER_OUTLINE=0 keeps the plane-by-plane walk (measurements); a hull whose planes
  // do not give a clean polytope is walked plane by plane as well.
  std::vector<int32_t> edge_planes;
  std::vector<double> edge_verts;
  int64_t view_stride = 0;
  {
    const char* env = std::getenv("RCSH_RENDER_OUTLINE");
    const bool outline = !(env && env[0] == '0');
    for (int i = 0; i < d->nshape && outline; ++i) {
      RenderShape& r = sh[i];
      if (r.shape != kShapeHull) continue;
      std::vector<HullEdge> edges;
      if (!build_hull_edges(d->planes + 4 * (size_t)r.plane_adr, r.plane_num, edges, r.centre)) continue;
      r.edge_adr = (int32_t)(edge_planes.size() / 2);
      r.edge_num = (int32_t)edges.size();
      r.view_adr = view_stride;
      view_stride += hull_view_doubles(r.plane_num);
      for (const HullEdge& e : edges) {
        edge_planes.push_back(e.a); edge_planes.push_back(e.b);
        edge_verts.insert(edge_verts.end(), e.v1, e.v1 + 3);
        edge_verts.insert(edge_verts.end(), e.v2, e.v2 + 3);
      }
    }
  }
  HIP_TRY(hipSetDevice(s->device));
  const bool first_scene = s->d_frames == nullptr;
  hipFree(s->d_rshapes); hipFree(s->d_rplanes); hipFree(s->d_frames); hipFree(s->d_wframes); hipFree(s->d_rcolours);
  hipFree(s->d_redge_planes); hipFree(s->d_redge_verts); hipFree(s->d_rviews);
  s->d_rshapes = nullptr; s->d_rplanes = nullptr; s->d_frames = nullptr; s->d_wframes = nullptr; s->d_rcolours = nullptr;
  s->d_redge_planes = nullptr; s->d_redge_verts = nullptr; s->d_rviews = nullptr;
  s->rscene.colours = nullptr;
  s->rscene.edge_planes = nullptr; s->rscene.edge_verts = nullptr; s->rscene.views = nullptr; s->rscene.view_stride = 0;
  const int np = d->nplanes > 0 ? d->nplanes : 1;
  HIP_TRY(hipMalloc(&s->d_rshapes, sizeof(RenderShape) * d->nshape));
  HIP_TRY(hipMalloc(&s->d_rplanes, sizeof(double) * 4 * np));
  HIP_TRY(hipMalloc(&s->d_frames, sizeof(double) * 12 * (size_t)(s->nl + 1) * s->n));
  HIP_TRY(hipMalloc(&s->d_wframes, sizeof(double) * kShapeFrameDoubles * (size_t)(d->nshape + 1) * s->n));
  if (view_stride > 0) {
    HIP_TRY(hipMalloc(&s->d_redge_planes, sizeof(int32_t) * edge_planes.size()));
    HIP_TRY(hipMalloc(&s->d_redge_verts, sizeof(double) * edge_verts.size()));
    HIP_TRY(hipMalloc(&s->d_rviews, sizeof(double) * (size_t)view_stride * s->n));
    HIP_TRY(hipMemcpyAsync(s->d_redge_planes, edge_planes.data(), sizeof(int32_t) * edge_planes.size(), hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipMemcpyAsync(s->d_redge_verts, edge_verts.data(), sizeof(double) * edge_verts.size(), hipMemcpyHostToDevice, s->stream));
  }
  HIP_TRY(hipMemcpyAsync(s->d_rshapes, sh.data(), sizeof(RenderShape) * d->nshape, hipMemcpyHostToDevice, s->stream));
  if (d->nplanes > 0) HIP_TRY(hipMemcpyAsync(s->d_rplanes, d->planes, sizeof(double) * 4 * d->nplanes, hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  // The kernels keep "the qpos the last position stage saw" (what mjData.xpos / geom_xpos derive from) only while a render
  // scene is attached.  A scene attached after some stepping starts from the current qpos instead of whatever the field held:
  // one substep's motion off for the first frame, never a stale pose.
  if (first_scene) with_layout(s, [&](auto topo) {
    using L = Lay<decltype(topo)>;
    const size_t n = (size_t)s->n;
    (void)hipMemcpyAsync(s->S + (size_t)L::QPRE * n, s->S + (size_t)L::QPOS * n, sizeof(double) * n * s->nl, hipMemcpyDeviceToDevice, s->stream);
    if (s->box.present)
      (void)hipMemcpyAsync(s->S + (size_t)(L::BOX + kBoxPre) * n, s->S + (size_t)(L::BOX + kBoxQ) * n, sizeof(double) * n * 7, hipMemcpyDeviceToDevice, s->stream);
    return 0;
  });
  HIP_TRY(hipStreamSynchronize(s->stream));
  s->rscene.nshape = d->nshape; s->rscene.nframes = s->nl + 1;
  s->rscene.znear = d->znear; s->rscene.zfar = d->zfar;
  s->rscene.inv_near = 1.0 / d->znear; s->rscene.inv_span = 1.0 / (1.0 / d->znear - 1.0 / d->zfar);
  s->rscene.shapes = s->d_rshapes; s->rscene.planes = s->d_rplanes;
  s->rscene.edge_planes = s->d_redge_planes; s->rscene.edge_verts = s->d_redge_verts; s->rscene.views = s->d_rviews; s->rscene.view_stride = view_stride;
  return RCSH_OK;
}

int rcsh_hull_edges(const double* planes, int32_t nplanes, int32_t capacity, int32_t* edge_planes, double* edge_verts, int32_t* nedges, double* centre) {
  if (!planes || !nedges || !centre || nplanes < 0 || capacity < 0) return fail(RCSH_ERR_ARG, "hull edges: null argument");
  std::vector<HullEdge> edges;
  *nedges = 0;
  if (!build_hull_edges(planes, nplanes, edges, centre)) return RCSH_OK;  // (no clean polytope: zero edges, the hull is walked plane by plane)
  *nedges = (int32_t)edges.size();
  if (!edge_planes || !edge_verts) return RCSH_OK;
  if ((int32_t)edges.size() > capacity) return fail(RCSH_ERR_ARG, "hull edges: capacity too small");
  for (size_t k = 0; k < edges.size(); ++k) {
    edge_planes[2 * k] = edges[k].a; edge_planes[2 * k + 1] = edges[k].b;
    for (int t = 0; t < 3; ++t) { edge_verts[6 * k + t] = edges[k].v1[t]; edge_verts[6 * k + 3 + t] = edges[k].v2[t]; }
  }
  return RCSH_OK;
}

int rcsh_sim_add_camera(rcsh_sim* s, const rcsh_camera_desc* c, int32_t* cam_id) {
  REQUIRE_SIM(s);
  if (!c || !cam_id) return fail(RCSH_ERR_ARG, "null camera description");
  if (c->width < 1 || c->height < 1 || !(c->fovy_deg > 0 && c->fovy_deg < 180)) return fail(RCSH_ERR_ARG, "camera: bad resolution or field of view");
  if (c->link < kLinkFreeBody || c->link >= s->nl) return fail(RCSH_ERR_ARG, "camera: link index out of range");
  RenderCam rc{};
  rc.link = c->link; rc.width = c->width; rc.height = c->height;
  for (int k = 0; k < 3; ++k) rc.pos[k] = c->pos[k];
  for (int k = 0; k < 9; ++k) rc.rot[k] = c->rot[k];
  rc.tan_half_fovy = std::tan(c->fovy_deg * 3.14159265358979323846 / 360.0);
  rc.tx = rc.tan_half_fovy * (double)c->width / (double)c->height;
  rc.two_over_w = 2.0 / c->width; rc.two_over_h = 2.0 / c->height;
  s->cams.push_back(rc);
  *cam_id = (int32_t)s->cams.size() - 1;
  return RCSH_OK;
}

int rcsh_sim_set_render_colours(rcsh_sim* s, const rcsh_render_colours* c) {
  REQUIRE_SIM(s);
  if (!s->d_frames) return fail(RCSH_ERR_STATE, "no render scene: call rcsh_sim_set_render_scene first");
  if (!c || !c->colour) return fail(RCSH_ERR_ARG, "null colour table");
  const int ns = s->rscene.nshape;
  std::vector<RenderColour> col(ns);
  for (int i = 0; i < ns; ++i) {
    const double* w = c->colour + 8 * (size_t)i;
    for (int k = 0; k < 3; ++k) { col[i].rgb[k] = w[k]; col[i].rgb2[k] = w[3 + k]; }
    col[i].square = w[6]; col[i].checker = w[7];
    if (col[i].checker != 0.0 && !(col[i].square > 0)) return fail(RCSH_ERR_ARG, "render colours: checker squares need a positive edge length");
  }
  HIP_TRY(hipSetDevice(s->device));
  if (!s->d_rcolours) HIP_TRY(hipMalloc(&s->d_rcolours, sizeof(RenderColour) * ns));
  HIP_TRY(hipMemcpyAsync(s->d_rcolours, col.data(), sizeof(RenderColour) * ns, hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  s->rscene.colours = s->d_rcolours;
  RenderShade& L = s->rscene.shade;
  for (int k = 0; k < 3; ++k) {
    L.ambient[k] = c->headlight_ambient[k]; L.head_diffuse[k] = c->headlight_diffuse[k];
    L.light_dir[k] = c->light_dir[k]; L.light_diffuse[k] = c->light_diffuse[k];
    L.sky1[k] = c->sky_rgb1[k]; L.sky2[k] = c->sky_rgb2[k];
  }
  const double dn = std::sqrt(L.light_dir[0] * L.light_dir[0] + L.light_dir[1] * L.light_dir[1] + L.light_dir[2] * L.light_dir[2]);
  if (dn > 0) for (int k = 0; k < 3; ++k) L.light_dir[k] /= dn;
  return RCSH_OK;
}

int rcsh_sim_set_render_schedule(rcsh_sim* s, const int32_t* cam_ids, const double* seconds_between_calls, int32_t ncam, int32_t capacity) {
  REQUIRE_SIM(s);
  if (!s->d_frames) return fail(RCSH_ERR_STATE, "no render scene: call rcsh_sim_set_render_scene first");
  if (ncam < 0 || ncam > kMaxRateCams) return fail(RCSH_ERR_ARG, "render schedule: at most 4 cameras with a frame rate");
  if (ncam > 0 && (!cam_ids || !seconds_between_calls || capacity < 1 || capacity > 256)) return fail(RCSH_ERR_ARG, "render schedule: bad arguments");
  for (int c = 0; c < ncam; ++c) {
    if (cam_ids[c] < 0 || cam_ids[c] >= (int)s->cams.size()) return fail(RCSH_ERR_ARG, "render schedule: unknown camera id");
    if (!(seconds_between_calls[c] > 0)) return fail(RCSH_ERR_ARG, "render schedule: the period must be positive");
  }
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  const size_t n = (size_t)s->n, nf = (size_t)s->nl + 9;
  // The same cameras with the same periods and a larger capacity: the schedule GROWS -- the cameras' clocks and what the last
  // launch recorded stay (a host that is about to run a longer launch than the schedule was sized for -- Sim.step(k) with a
  // large k, a raised max_convergence_steps -- calls this first; re-registering must not make every camera due again).
  bool same = ncam > 0 && ncam == s->rend.ncam && capacity >= s->rend.capacity;
  for (int c = 0; same && c < ncam; ++c) same = s->rend_cam_id[c] == cam_ids[c] && s->rend.period[c] == seconds_between_calls[c];
  if (same) {
    if (capacity == s->rend.capacity) return RCSH_OK;
    double* snap = nullptr;
    HIP_TRY(hipMalloc(&snap, sizeof(double) * (size_t)capacity * nf * n));
    HIP_TRY(hipMemcpy(snap, s->rend.snap, sizeof(double) * (size_t)s->rend.capacity * nf * n, hipMemcpyDeviceToDevice));
    hipFree(s->rend.snap);
    s->rend.snap = snap;
    s->rend.capacity = capacity;
    return RCSH_OK;
  }
  hipFree(s->rend.last); hipFree(s->rend.snap); hipFree(s->rend.count);
  s->rend = RendCfg{};
  s->rend_dropped = 0;
  if (ncam == 0) return RCSH_OK;
  HIP_TRY(hipMalloc(&s->rend.last, sizeof(double) * kMaxRateCams * n));
  HIP_TRY(hipMalloc(&s->rend.snap, sizeof(double) * (size_t)capacity * nf * n));
  HIP_TRY(hipMalloc(&s->rend.count, sizeof(int32_t) * n));
  HIP_TRY(hipMemsetAsync(s->rend.count, 0, sizeof(int32_t) * n, s->stream));
  // register_rendering_callback (sim.cpp:160-173): last_call_timestamp = -1 / frame_rate, "so that we will directly render"
  std::vector<double> last(kMaxRateCams * n, 0.0);
  for (int c = 0; c < ncam; ++c) {
    s->rend.period[c] = seconds_between_calls[c];
    s->rend_cam_id[c] = cam_ids[c];
    for (size_t e = 0; e < n; ++e) last[c * n + e] = -seconds_between_calls[c];
  }
  HIP_TRY(hipMemcpyAsync(s->rend.last, last.data(), sizeof(double) * last.size(), hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  s->rend.ncam = ncam;
  s->rend.capacity = capacity;
  return RCSH_OK;
}

int rcsh_render_pending(rcsh_sim* s, int32_t* count) {
  REQUIRE_SIM(s);
  if (!count) return fail(RCSH_ERR_ARG, "null output");
  if (s->rend.ncam == 0) return fail(RCSH_ERR_STATE, "no render schedule: call rcsh_sim_set_render_schedule first");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipMemcpyAsync(count, s->rend.count, sizeof(int32_t) * s->n, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  // more frames due in one launch than the schedule holds: the records beyond its capacity were not written (the newest are
  // lost); the stepping itself is unaffected, so this is a count for the host to warn about, not an error after the fact
  bool clamped = false;
  for (int e = 0; e < s->n; ++e)
    if (count[e] > s->rend.capacity) {
      s->rend_dropped += count[e] - s->rend.capacity;
      count[e] = s->rend.capacity;
      clamped = true;
    }
  if (clamped) {
    // the device-side counters outlive this call (observation-only launches keep them, and so does a second pending / collect
    // without a stepping launch in between): write the clamped counts back, so that an overflow is counted ONCE (advisor, round 3)
    HIP_TRY(hipMemcpyAsync(s->rend.count, count, sizeof(int32_t) * s->n, hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
  }
  return RCSH_OK;
}

int rcsh_render_dropped(rcsh_sim* s, int64_t* dropped) {
  REQUIRE_SIM(s);
  if (dropped) *dropped = s->rend_dropped;
  return RCSH_OK;
}

int rcsh_camera_render_snapshot(rcsh_sim* s, int32_t cam_id, int32_t slot, uint8_t* rgb, float* depth_gl, uint16_t* depth_mm, double* cam_pose,
                                double* timestamp, uint8_t* due) {
  REQUIRE_SIM(s);
  if (s->rend.ncam == 0) return fail(RCSH_ERR_STATE, "no render schedule: call rcsh_sim_set_render_schedule first");
  if (slot < 0 || slot >= s->rend.capacity) return fail(RCSH_ERR_ARG, "snapshot slot out of range");
  int which = -1;
  for (int c = 0; c < s->rend.ncam; ++c) which = s->rend_cam_id[c] == cam_id ? c : which;
  if (which < 0) return fail(RCSH_ERR_ARG, "camera is not part of the render schedule");
  HIP_TRY(hipSetDevice(s->device));
  const size_t n = (size_t)s->n, nf = (size_t)s->nl + 9;
  // the frames kernel reads "the qpos the last position stage saw" and the box's pre-step pose through field offsets: point
  // it at the record instead of the state
  s->frames_src = s->rend.snap + (size_t)slot * nf * n;
  int rc = rcsh_camera_render_rgb(s, cam_id, rgb, depth_gl, depth_mm, cam_pose);
  s->frames_src = nullptr;
  if (rc) return rc;
  std::vector<double> tm(2 * n);
  HIP_TRY(hipMemcpyAsync(tm.data(), s->frames_src_base(slot) + (size_t)(s->nl + 7) * n, sizeof(double) * 2 * n, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  std::vector<int32_t> cnt(n);
  HIP_TRY(hipMemcpy(cnt.data(), s->rend.count, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
  for (size_t e = 0; e < n; ++e) {
    const bool have = slot < cnt[e] && (((uint32_t)tm[n + e] >> which) & 1u);
    if (timestamp) timestamp[e] = tm[e];
    if (due) due[e] = have ? 1 : 0;
  }
  return RCSH_OK;
}

int rcsh_camera_render_rgb_dev(rcsh_sim* s, int32_t cam_id, uint8_t* rgb, float* depth_gl, uint16_t* depth_mm, double* cam_pose) {
  REQUIRE_SIM(s);
  if (!s->d_frames) return fail(RCSH_ERR_STATE, "no render scene: call rcsh_sim_set_render_scene first");
  if (rgb && !s->rscene.colours) return fail(RCSH_ERR_STATE, "no colours: call rcsh_sim_set_render_colours first");
  if (cam_id < 0 || cam_id >= (int)s->cams.size()) return fail(RCSH_ERR_ARG, "unknown camera id");
  HIP_TRY(hipSetDevice(s->device));
  const RenderCam& cam = s->cams[cam_id];
  hipError_t err = hipSuccess;
  bool ok = dispatch_topology(s->narm, s->grip, [&](auto topo) {
    using T = decltype(topo);
    if (s->frames_src)  // a record of the render schedule: qpre at field 0, the box's pre-step pose behind it
      hipLaunchKernelGGL((k_link_frames<T>), dim3(grid_for(s->n)), dim3(kBlock), 0, s->stream, s->d_model, s->frames_src, s->n, 0,
                         T::NL - kBoxPre, (int)s->box.present, s->d_frames);
    else
      hipLaunchKernelGGL((k_link_frames<T>), dim3(grid_for(s->n)), dim3(kBlock), 0, s->stream, s->d_model, s->S, s->n,
                         (int)Lay<T>::QPRE, (int)Lay<T>::BOX, (int)s->box.present, s->d_frames);
    err = hipGetLastError();
  });
  if (!ok) return fail(RCSH_ERR_MODEL, "no kernel instantiated for this archetype");
  if (err != hipSuccess) return fail(RCSH_ERR_DEVICE, std::string("k_link_frames launch: ") + hipGetErrorString(err));
  hipLaunchKernelGGL(k_shape_frames, dim3(grid_for(s->n * (s->rscene.nshape + 1))), dim3(kBlock), 0, s->stream, s->rscene, cam, s->d_frames, s->n,
                     s->d_wframes);
  // (the rays' arithmetic type: float unless rcsh_sim_set_render_f64 asked for the instantiation that equals the restatement bit for bit)
  const dim3 grid((unsigned)(((size_t)render_wgs_per_env(cam.width, cam.height) * (size_t)s->n + 7) / 8 * 8));  // (a multiple of 8: k_render_depth numbers its workgroups per XCD)
  auto cast = [&](auto zero) {
    using F = decltype(zero);
    if (s->rscene.views)
      hipLaunchKernelGGL(k_hull_views<F>, dim3((unsigned)s->n * (unsigned)s->rscene.nshape), dim3(64), 0, s->stream, s->rscene, cam, s->d_wframes, s->n);
    if (rgb)
      hipLaunchKernelGGL((k_render_depth<true, F>), grid, dim3(256), 0, s->stream, s->rscene, cam, s->d_wframes, s->n, depth_gl, depth_mm, cam_pose, rgb);
    else
      hipLaunchKernelGGL((k_render_depth<false, F>), grid, dim3(256), 0, s->stream, s->rscene, cam, s->d_wframes, s->n, depth_gl, depth_mm, cam_pose,
                         (uint8_t*)nullptr);
  };
  if (s->render_f64) cast(0.0); else cast(0.0f);
  err = hipGetLastError();
  if (err != hipSuccess) return fail(RCSH_ERR_DEVICE, std::string("k_render_depth launch: ") + hipGetErrorString(err));
  return RCSH_OK;
}

int rcsh_sim_set_render_f64(rcsh_sim* s, int32_t on) {
  REQUIRE_SIM(s);
  s->render_f64 = on != 0;
  return RCSH_OK;
}

int rcsh_camera_render_dev(rcsh_sim* s, int32_t cam_id, float* depth_gl, uint16_t* depth_mm, double* cam_pose) {
  return rcsh_camera_render_rgb_dev(s, cam_id, nullptr, depth_gl, depth_mm, cam_pose);
}

int rcsh_camera_render_rgb(rcsh_sim* s, int32_t cam_id, uint8_t* rgb, float* depth_gl, uint16_t* depth_mm, double* cam_pose) {
  REQUIRE_SIM(s);
  if (cam_id < 0 || cam_id >= (int)s->cams.size()) return fail(RCSH_ERR_ARG, "unknown camera id");
  HIP_TRY(hipSetDevice(s->device));
  const size_t px = (size_t)s->n * s->cams[cam_id].width * s->cams[cam_id].height;
  // device staging: f32 depth, u16 depth, camera poses (8-byte aligned), rgb
  const size_t off_mm = px * sizeof(float), off_pose = ((off_mm + px * sizeof(uint16_t) + 7) / 8) * 8, off_rgb = off_pose + sizeof(double) * 12 * s->n;
  const size_t need = off_rgb + 3 * px;
  if (need > s->image_cap) {
    hipFree(s->d_image);
    s->d_image = nullptr; s->image_cap = 0;
    HIP_TRY(hipMalloc(&s->d_image, need));
    s->image_cap = need;
  }
  char* base = static_cast<char*>(s->d_image);
  float* dgl = reinterpret_cast<float*>(base);
  uint16_t* dmm = reinterpret_cast<uint16_t*>(base + off_mm);
  double* dpose = reinterpret_cast<double*>(base + off_pose);
  uint8_t* drgb = reinterpret_cast<uint8_t*>(base + off_rgb);
  int rc = rcsh_camera_render_rgb_dev(s, cam_id, rgb ? drgb : nullptr, depth_gl ? dgl : nullptr, depth_mm ? dmm : nullptr, cam_pose ? dpose : nullptr);
  if (rc) return rc;
  if (rgb) HIP_TRY(hipMemcpyAsync(rgb, drgb, 3 * px, hipMemcpyDeviceToHost, s->stream));
  if (depth_gl) HIP_TRY(hipMemcpyAsync(depth_gl, dgl, px * sizeof(float), hipMemcpyDeviceToHost, s->stream));
  if (depth_mm) HIP_TRY(hipMemcpyAsync(depth_mm, dmm, px * sizeof(uint16_t), hipMemcpyDeviceToHost, s->stream));
  if (cam_pose) HIP_TRY(hipMemcpyAsync(cam_pose, dpose, sizeof(double) * 12 * s->n, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RCSH_OK;
}

int rcsh_camera_render(rcsh_sim* s, int32_t cam_id, float* depth_gl, uint16_t* depth_mm, double* cam_pose) {
  return rcsh_camera_render_rgb(s, cam_id, nullptr, depth_gl, depth_mm, cam_pose);
}

// ---- The all-gather's second carrier: copy engines instead of a collective kernel.
// RCCL's all-gather is a KERNEL (261-280 registers a lane in this image's librccl.so): next to a stepping wavefront of 424 registers it
// finds no SIMD to run on, so on a full batch the gather starts when the env-step ends instead of overlapping it.  Here every rank
// WRITES its block into the receive buffer of each peer (IPC-mapped) with plain asynchronous copies -- SDMA engines over xGMI, one
// stream per peer so the seven links work in parallel -- and says so with an 8-byte copy of the step's sequence number into a flag
// word of the peer.  The only thing that runs on a CU is one single wavefront of 16 registers per gather that waits for the flag words
// (k_wait_flags): it fits beside a stepping wavefront.  Same slot protocol, same entry points for posting and waiting
// (rcsh_comm_allgather_dev / rcsh_comm_wait); the receive buffers belong to the carrier (they have to be exported), see rcs_hip.h.
//   flags[kind][slot][rank]: kind 0 "ack" -- rank says its receive buffer of `slot` is free for sequence number q (its consumer is done
//   with what the buffer held); kind 1 "got" -- rank's block of sequence q has arrived in this rank's receive buffer of `slot`.
namespace {
constexpr int kCopyMaxWorld = 16;
constexpr int kCopySeqRing = 256;
struct CopyBlob {  // what a rank exports (RCSH_COMM_COPY_BLOB_BYTES >= sizeof)
  uint32_t magic, rank;
  uint64_t bytes_per_rank;
  int32_t device, pid;
  hipIpcMemHandle_t recv[2], flags;
};
static_assert(sizeof(CopyBlob) <= RCSH_COMM_COPY_BLOB_BYTES, "the blob fits what the header promises");
}  // namespace
struct CopyCarrier {
  int rank = 0, world = 1;
  size_t bytes = 0;
  void* recv[2] = {nullptr, nullptr};   // [world * bytes] each, this rank's
  uint64_t* flags = nullptr;            // [2][2][kCopyMaxWorld], this rank's (fine-grained: peers write it, a waiting wavefront reads it)
  void* peer_recv[kCopyMaxWorld][2] = {};
  uint64_t* peer_flags[kCopyMaxWorld] = {};
  bool opened[kCopyMaxWorld] = {};
  hipStream_t cs[kCopyMaxWorld] = {};   // one copy stream per peer (own rank: the local block)
  hipEvent_t cs_done[kCopyMaxWorld] = {}, acked = nullptr;
  uint64_t* seq_ring = nullptr;         // pinned: the sequence numbers the flag copies read
  uint64_t seq[2] = {0, 0};
  int ring_pos = 0;
  uint32_t* timeout_flag = nullptr;     // pinned: a wait gave up (a peer died)
  bool connected = false;
  // how the flag words are written and waited for: stream memory operations (hipStreamWriteValue64 / hipStreamWaitValue64: the
  // command processor does both, nothing runs on a CU and nothing goes through a copy engine) with RCSH_COPY_CARRIER_FLAGS=value where
  // the device has them; by default an 8-byte copy and the waiting wavefront.
  bool stream_values = false;
  // The whole gather of a slot as ONE captured graph (the default): per post the host records an event, makes the communicator's stream
  // wait for it and launches the graph -- four calls instead of ~5 W + 4.  The sequence number then lives on the device (qdev[slot],
  // bumped by the graph's first node); the flag words are 8-byte device-to-device copies of it, the waiting wavefronts read it.
  bool use_graph = true;
  hipGraphExec_t gexec[2] = {nullptr, nullptr};
  hipGraph_t graph[2] = {nullptr, nullptr};
  const void* gsend[2] = {nullptr, nullptr};
  uint64_t* qdev = nullptr;             // [2] the slots' sequence numbers (graph form)
};
namespace {
// one wavefront: waits until the `n` words at `w` (skipping index `skip`) have all reached `q`; gives up after ~20 s
__global__ void __launch_bounds__(64) k_wait_flags(const uint64_t* w, int n, int skip, uint64_t q, uint32_t* timeout_flag) {
  const int lane = threadIdx.x;
  const unsigned long long t0 = wall_clock64();
  for (;;) {
    bool ok = true;
    if (lane < n && lane != skip) ok = __hip_atomic_load(w + lane, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= q;
    if (__ballot(!ok) == 0) break;
    __builtin_amdgcn_s_sleep(32);
    if (wall_clock64() - t0 > 2000000000ull) {  // (100 MHz constant clock)
      if (lane == 0) __hip_atomic_store(timeout_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      break;
    }
  }
}
// (graph form) the sequence number is read from device memory; the slot's number is bumped by the graph's first node
__global__ void __launch_bounds__(64) k_wait_flags_at(const uint64_t* w, int n, int skip, const uint64_t* qptr, uint32_t* timeout_flag) {
  const int lane = threadIdx.x;
  const uint64_t q = *qptr;
  const unsigned long long t0 = wall_clock64();
  for (;;) {
    bool ok = true;
    if (lane < n && lane != skip) ok = __hip_atomic_load(w + lane, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= q;
    if (__ballot(!ok) == 0) break;
    __builtin_amdgcn_s_sleep(32);
    if (wall_clock64() - t0 > 2000000000ull) {
      if (lane == 0) __hip_atomic_store(timeout_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      break;
    }
  }
}
__global__ void k_bump(uint64_t* q) { *q += 1; }
void copy_carrier_free(rcsh_sim* s) {
  CopyCarrier* c = s->copy;
  if (!c) return;
  for (int k = 0; k < 2; ++k) {
    if (c->gexec[k]) hipGraphExecDestroy(c->gexec[k]);
    if (c->graph[k]) hipGraphDestroy(c->graph[k]);
  }
  if (c->qdev) hipFree(c->qdev);
  for (int p = 0; p < c->world; ++p) {
    if (c->cs[p]) hipStreamSynchronize(c->cs[p]);
  }
  for (int p = 0; p < c->world; ++p) {
    if (c->opened[p]) {
      for (int k = 0; k < 2; ++k) if (c->peer_recv[p][k]) hipIpcCloseMemHandle(c->peer_recv[p][k]);
      if (c->peer_flags[p]) hipIpcCloseMemHandle(c->peer_flags[p]);
    }
    if (c->cs[p]) hipStreamDestroy(c->cs[p]);
    if (c->cs_done[p]) hipEventDestroy(c->cs_done[p]);
  }
  if (c->acked) hipEventDestroy(c->acked);
  for (int k = 0; k < 2; ++k) if (c->recv[k]) hipFree(c->recv[k]);
  if (c->flags) hipFree(c->flags);
  if (c->seq_ring) hipHostFree(c->seq_ring);
  if (c->timeout_flag) hipHostFree(c->timeout_flag);
  delete c;
  s->copy = nullptr;
}
}  // namespace

int rcsh_comm_copy_create(rcsh_sim* s, int32_t rank, int32_t world, size_t bytes_per_rank, uint8_t blob[RCSH_COMM_COPY_BLOB_BYTES]) {
  REQUIRE_SIM(s);
  if (!blob || world < 1 || world > kCopyMaxWorld || rank < 0 || rank >= world || bytes_per_rank == 0 || bytes_per_rank % 8)
    return fail(RCSH_ERR_ARG, "copy carrier: need 0 <= rank < world <= 16 and a block size that is a multiple of 8 bytes");
  if (s->comm || s->copy) return fail(RCSH_ERR_STATE, "a communicator is already attached to this sim");
  CopyCarrier* c = new CopyCarrier;
  s->copy = c;
  c->rank = rank; c->world = world; c->bytes = bytes_per_rank;
  hipError_t he = hipSuccess;
  auto ok = [&](hipError_t e) { if (he == hipSuccess) he = e; return he == hipSuccess; };
  for (int k = 0; k < 2 && he == hipSuccess; ++k) ok(hipMalloc(&c->recv[k], bytes_per_rank * world));
  // (fine-grained: the words are written by copies other processes start and polled by a wavefront of this one)
  void* fl = nullptr;
  if (ok(hipExtMallocWithFlags(&fl, sizeof(uint64_t) * 2 * 2 * kCopyMaxWorld, hipDeviceMallocFinegrained))) {
    c->flags = (uint64_t*)fl;
    ok(hipMemset(c->flags, 0, sizeof(uint64_t) * 2 * 2 * kCopyMaxWorld));
  }
  if (he == hipSuccess && ok(hipMalloc((void**)&c->qdev, 2 * sizeof(uint64_t)))) ok(hipMemset(c->qdev, 0, 2 * sizeof(uint64_t)));
  if (const char* e = std::getenv("RCSH_COPY_CARRIER_GRAPH")) c->use_graph = std::atoi(e) != 0;
  if (he == hipSuccess) ok(hipHostMalloc((void**)&c->seq_ring, sizeof(uint64_t) * kCopySeqRing, hipHostMallocDefault));
  if (he == hipSuccess) ok(hipHostMalloc((void**)&c->timeout_flag, sizeof(uint32_t), hipHostMallocDefault));
  if (he == hipSuccess) *c->timeout_flag = 0;
  if (he == hipSuccess) ok(hipStreamCreateWithFlags(&s->comm_stream, hipStreamNonBlocking));
  if (he == hipSuccess) ok(hipEventCreateWithFlags(&s->comm_ready, hipEventDisableTiming));
  for (int k = 0; k < 2 && he == hipSuccess; ++k) ok(hipEventCreateWithFlags(&s->comm_done[k], hipEventDisableTiming));
  if (he == hipSuccess) ok(hipEventCreateWithFlags(&c->acked, hipEventDisableTiming));
  for (int p = 0; p < world && he == hipSuccess; ++p) {
    ok(hipStreamCreateWithFlags(&c->cs[p], hipStreamNonBlocking));
    if (he == hipSuccess) ok(hipEventCreateWithFlags(&c->cs_done[p], hipEventDisableTiming));
  }
  {
    // (default: the 8-byte copies and ONE waiting wavefront per wait -- a wait on W - 1 words is W - 1 stream operations, and at 8 ranks the
    // host's calls per gather are what limits this carrier; both ways pass the two-process test and measure the same on one device)
    int can = 0;
    if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, s->device) != hipSuccess) can = 0;
    c->stream_values = false;
    if (const char* e = std::getenv("RCSH_COPY_CARRIER_FLAGS")) c->stream_values = can != 0 && std::string(e) == "value";
  }
  CopyBlob b{};
  b.magic = 0x52435348u; b.rank = (uint32_t)rank; b.bytes_per_rank = bytes_per_rank; b.device = s->device; b.pid = (int32_t)getpid();
  for (int k = 0; k < 2 && he == hipSuccess; ++k) ok(hipIpcGetMemHandle(&b.recv[k], c->recv[k]));
  if (he == hipSuccess) ok(hipIpcGetMemHandle(&b.flags, c->flags));
  if (he != hipSuccess) {
    if (s->comm_stream) hipStreamDestroy(s->comm_stream);
    if (s->comm_ready) hipEventDestroy(s->comm_ready);
    for (int k = 0; k < 2; ++k) if (s->comm_done[k]) hipEventDestroy(s->comm_done[k]);
    s->comm_stream = nullptr; s->comm_ready = s->comm_done[0] = s->comm_done[1] = nullptr;
    copy_carrier_free(s);
    return fail(RCSH_ERR_DEVICE, std::string("copy carrier: ") + hipGetErrorString(he));
  }
  std::memset(blob, 0, RCSH_COMM_COPY_BLOB_BYTES);
  std::memcpy(blob, &b, sizeof(b));
  s->comm_pending[0] = s->comm_pending[1] = false;
  s->comm_rank = rank; s->comm_world = world;
  return RCSH_OK;
}

int rcsh_comm_copy_connect(rcsh_sim* s, const uint8_t* blobs) {
  REQUIRE_SIM(s);
  CopyCarrier* c = s->copy;
  if (!c) return fail(RCSH_ERR_STATE, "no copy carrier: call rcsh_comm_copy_create first");
  if (c->connected) return fail(RCSH_ERR_STATE, "the copy carrier is connected already");
  if (!blobs) return fail(RCSH_ERR_ARG, "null blobs");
  for (int p = 0; p < c->world; ++p) {
    CopyBlob b;
    std::memcpy(&b, blobs + (size_t)p * RCSH_COMM_COPY_BLOB_BYTES, sizeof(b));
    if (b.magic != 0x52435348u || (int)b.rank != p || b.bytes_per_rank != c->bytes)
      return fail(RCSH_ERR_ARG, "copy carrier: blob " + std::to_string(p) + " is not rank " + std::to_string(p) + "'s, or the ranks disagree on the block size");
    if (p == c->rank) {
      c->peer_recv[p][0] = c->recv[0]; c->peer_recv[p][1] = c->recv[1]; c->peer_flags[p] = c->flags;
      continue;
    }
    if (b.pid == (int32_t)getpid()) return fail(RCSH_ERR_ARG, "copy carrier: two ranks in one process (IPC handles open in another process only)");
    for (int k = 0; k < 2; ++k) HIP_TRY(hipIpcOpenMemHandle(&c->peer_recv[p][k], b.recv[k], hipIpcMemLazyEnablePeerAccess));
    void* fl = nullptr;
    HIP_TRY(hipIpcOpenMemHandle(&fl, b.flags, hipIpcMemLazyEnablePeerAccess));
    c->peer_flags[p] = (uint64_t*)fl;
    c->opened[p] = true;
  }
  c->connected = true;
  return RCSH_OK;
}

int rcsh_comm_copy_recv_buffer(rcsh_sim* s, int32_t slot, void** recv_dev) {
  REQUIRE_SIM(s);
  if (!s->copy) return fail(RCSH_ERR_STATE, "no copy carrier: call rcsh_comm_copy_create first");
  if (slot < 0 || slot > 1 || !recv_dev) return fail(RCSH_ERR_ARG, "exchange slot is 0 or 1");
  *recv_dev = s->copy->recv[slot];
  return RCSH_OK;
}

namespace {
int copy_allgather(rcsh_sim* s, int32_t slot, const void* send_dev, void* recv_dev, size_t bytes_per_rank) {
  CopyCarrier* c = s->copy;
  if (!c->connected) return fail(RCSH_ERR_STATE, "copy carrier: call rcsh_comm_copy_connect first");
  if (recv_dev != c->recv[slot] || bytes_per_rank != c->bytes)
    return fail(RCSH_ERR_ARG, "copy carrier: the receive buffer of a slot is the carrier's (rcsh_comm_copy_recv_buffer), the block size the one it was created with");
  if (*c->timeout_flag) return fail(RCSH_ERR_DEVICE, "copy carrier: an earlier gather gave up waiting for a peer");
  const int me = c->rank, W = c->world;
  auto flag = [&](uint64_t* base, int kind, int sl, int r) { return base + ((size_t)kind * 2 + sl) * kCopyMaxWorld + r; };
  // after what the handle's stream holds so far: the env-step that wrote the send buffer, the consumer of this slot's last gather
  HIP_TRY(hipEventRecord(s->comm_ready, s->stream));
  HIP_TRY(hipStreamWaitEvent(s->comm_stream, s->comm_ready, 0));
  if (c->use_graph && !c->stream_values) {
    if (c->gexec[slot] && c->gsend[slot] != send_dev) {  // (another send buffer: the copies' source is part of the graph)
      hipGraphExecDestroy(c->gexec[slot]); hipGraphDestroy(c->graph[slot]);
      c->gexec[slot] = nullptr; c->graph[slot] = nullptr;
    }
    if (!c->gexec[slot]) {
      uint64_t* qd = c->qdev + slot;
      hipError_t ge = hipStreamBeginCapture(s->comm_stream, hipStreamCaptureModeThreadLocal);
      auto g = [&](hipError_t e_) { if (ge == hipSuccess) ge = e_; };
      if (ge == hipSuccess) {
        hipLaunchKernelGGL(k_bump, dim3(1), dim3(1), 0, s->comm_stream, qd);
        g(hipGetLastError());
        for (int p = 0; p < W; ++p)
          if (p != me) g(hipMemcpyAsync(flag(c->peer_flags[p], 0, slot, me), qd, sizeof(uint64_t), hipMemcpyDeviceToDevice, s->comm_stream));
        if (W > 1) {
          hipLaunchKernelGGL(k_wait_flags_at, dim3(1), dim3(64), 0, s->comm_stream, flag(c->flags, 0, slot, 0), W, me, qd, c->timeout_flag);
          g(hipGetLastError());
        }
        g(hipEventRecord(c->acked, s->comm_stream));
        for (int p = 0; p < W; ++p) {
          g(hipStreamWaitEvent(c->cs[p], c->acked, 0));
          g(hipMemcpyAsync((char*)c->peer_recv[p][slot] + (size_t)me * c->bytes, send_dev, c->bytes, hipMemcpyDeviceToDevice, c->cs[p]));
          if (p != me) g(hipMemcpyAsync(flag(c->peer_flags[p], 1, slot, me), qd, sizeof(uint64_t), hipMemcpyDeviceToDevice, c->cs[p]));
          g(hipEventRecord(c->cs_done[p], c->cs[p]));
          g(hipStreamWaitEvent(s->comm_stream, c->cs_done[p], 0));
        }
        if (W > 1) {
          hipLaunchKernelGGL(k_wait_flags_at, dim3(1), dim3(64), 0, s->comm_stream, flag(c->flags, 1, slot, 0), W, me, qd, c->timeout_flag);
          g(hipGetLastError());
        }
        hipGraph_t gr = nullptr;
        const hipError_t ee = hipStreamEndCapture(s->comm_stream, &gr);
        if (ge == hipSuccess) ge = ee;
        if (ge == hipSuccess) ge = hipGraphInstantiate(&c->gexec[slot], gr, nullptr, nullptr, 0);
        if (ge == hipSuccess) { c->graph[slot] = gr; c->gsend[slot] = send_dev; }
        else if (gr) hipGraphDestroy(gr);
      }
      if (ge != hipSuccess) {  // (no graph on this runtime: the calls one by one, below)
        (void)hipGetLastError();
        c->use_graph = false;
        c->gexec[slot] = nullptr;
        if (c->seq[0] || c->seq[1]) return fail(RCSH_ERR_DEVICE, std::string("copy carrier: graph capture failed after gathers had run as graphs: ") + hipGetErrorString(ge));
      }
    }
    if (c->gexec[slot]) {
      ++c->seq[slot];
      HIP_TRY(hipGraphLaunch(c->gexec[slot], s->comm_stream));
      HIP_TRY(hipEventRecord(s->comm_done[slot], s->comm_stream));
      s->comm_pending[slot] = true;
      return RCSH_OK;
    }
  }
  const uint64_t q = ++c->seq[slot];
  uint64_t* qsrc = c->seq_ring + (c->ring_pos++ % kCopySeqRing);
  *qsrc = q;
  auto write_word = [&](hipStream_t st, uint64_t* dst) -> hipError_t {
    if (c->stream_values) return hipStreamWriteValue64(st, dst, q, 0);
    return hipMemcpyAsync(dst, qsrc, sizeof(uint64_t), hipMemcpyHostToDevice, st);
  };
  auto wait_words = [&](hipStream_t st, uint64_t* base) -> hipError_t {  // every peer's word of this rank's flags has reached q
    if (W < 2) return hipSuccess;
    if (c->stream_values) {
      for (int p = 0; p < W; ++p) {
        if (p == me) continue;
        const hipError_t e = hipStreamWaitValue64(st, base + p, q, hipStreamWaitValueGte, ~0ull);
        if (e != hipSuccess) return e;
      }
      return hipSuccess;
    }
    hipLaunchKernelGGL(k_wait_flags, dim3(1), dim3(64), 0, st, base, W, me, q, c->timeout_flag);
    return hipGetLastError();
  };
  // 1. tell every peer that this rank's receive buffer of the slot is free for q; wait until every peer has said so
  for (int p = 0; p < W; ++p)
    if (p != me) HIP_TRY(write_word(s->comm_stream, flag(c->peer_flags[p], 0, slot, me)));
  HIP_TRY(wait_words(s->comm_stream, flag(c->flags, 0, slot, 0)));
  HIP_TRY(hipEventRecord(c->acked, s->comm_stream));
  // 2. the block to every peer, each over its own stream (its own link), followed by the word that says it has arrived
  for (int p = 0; p < W; ++p) {
    HIP_TRY(hipStreamWaitEvent(c->cs[p], c->acked, 0));
    HIP_TRY(hipMemcpyAsync((char*)c->peer_recv[p][slot] + (size_t)me * c->bytes, send_dev, c->bytes, hipMemcpyDeviceToDevice, c->cs[p]));
    if (p != me) HIP_TRY(write_word(c->cs[p], flag(c->peer_flags[p], 1, slot, me)));
    HIP_TRY(hipEventRecord(c->cs_done[p], c->cs[p]));
    HIP_TRY(hipStreamWaitEvent(s->comm_stream, c->cs_done[p], 0));  // (the send buffer is free once these have run)
  }
  // 3. the slot is gathered when every peer's block has arrived here
  HIP_TRY(wait_words(s->comm_stream, flag(c->flags, 1, slot, 0)));
  HIP_TRY(hipEventRecord(s->comm_done[slot], s->comm_stream));
  s->comm_pending[slot] = true;
  return RCSH_OK;
}
}  // namespace


// ---- RCCL behind the C-ABI.  The library is dlopen'ed so that single-GPU users neither link nor load it.
namespace {
struct Rccl {
  typedef struct { char internal[RCSH_COMM_ID_BYTES]; } UniqueId;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  void* lib = nullptr;
  std::string why;
};
static Rccl& rccl() {
  static Rccl r;
  if (r.lib || !r.why.empty()) return r;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (r.lib) break;
  }
  if (!r.lib) { r.why = std::string("librccl.so not loadable: ") + dlerror(); return r; }
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
  r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
  if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy || !r.GetErrorString) r.why = "librccl.so lacks the collective entry points";
  return r;
}
#define RCCL_TRY(expr)                                                                                             \
  do {                                                                                                             \
    const int _e = (expr);                                                                                         \
    if (_e != 0) return fail(RCSH_ERR_DEVICE, std::string(#expr) + ": " + rccl().GetErrorString(_e));             \
  } while (0)
}  // namespace

int rcsh_comm_get_unique_id(uint8_t id[RCSH_COMM_ID_BYTES]) {
  if (!id) return fail(RCSH_ERR_ARG, "null id");
  Rccl& r = rccl();
  if (!r.why.empty()) return fail(RCSH_ERR_DEVICE, r.why);
  Rccl::UniqueId u;
  RCCL_TRY(r.GetUniqueId(&u));
  std::memcpy(id, u.internal, RCSH_COMM_ID_BYTES);
  return RCSH_OK;
}

int rcsh_comm_init(rcsh_sim* s, const uint8_t id[RCSH_COMM_ID_BYTES], int32_t rank, int32_t world) {
  REQUIRE_SIM(s);
  if (!id || world < 1 || rank < 0 || rank >= world) return fail(RCSH_ERR_ARG, "communicator: need an id and 0 <= rank < world");
  if (s->comm) return fail(RCSH_ERR_STATE, "a communicator is already attached to this sim");
  Rccl& r = rccl();
  if (!r.why.empty()) return fail(RCSH_ERR_DEVICE, r.why);
  Rccl::UniqueId u;
  std::memcpy(u.internal, id, RCSH_COMM_ID_BYTES);
  // the communicator, its stream and its events belong to THIS handle's device: a host with one process per GPU that never
  // called hipSetDevice itself (a plain C host has no reason to) must not end up with every rank on device 0
  HIP_TRY(hipSetDevice(s->device));
  // stream and events first: they cannot fail collectively, ncclCommInitRank can only be entered by all ranks or none
  hipError_t he = hipStreamCreateWithFlags(&s->comm_stream, hipStreamNonBlocking);
  if (he == hipSuccess) he = hipEventCreateWithFlags(&s->comm_ready, hipEventDisableTiming);
  for (int k = 0; k < 2 && he == hipSuccess; ++k) he = hipEventCreateWithFlags(&s->comm_done[k], hipEventDisableTiming);
  int nrc = 0;
  if (he == hipSuccess) nrc = r.CommInitRank(&s->comm, world, u, rank);
  if (he != hipSuccess || nrc != 0) {
    if (s->comm_stream) hipStreamDestroy(s->comm_stream);
    if (s->comm_ready) hipEventDestroy(s->comm_ready);
    for (int k = 0; k < 2; ++k) if (s->comm_done[k]) hipEventDestroy(s->comm_done[k]);
    s->comm = nullptr; s->comm_stream = nullptr; s->comm_ready = s->comm_done[0] = s->comm_done[1] = nullptr;
    if (he != hipSuccess) return fail(RCSH_ERR_DEVICE, std::string("communicator stream / events: ") + hipGetErrorString(he));
    return fail(RCSH_ERR_DEVICE, std::string("ncclCommInitRank: ") + (r.GetErrorString ? r.GetErrorString(nrc) : "error ") + " (" + std::to_string(nrc) + ")");
  }
  s->comm_pending[0] = s->comm_pending[1] = false;
  s->comm_rank = rank;
  s->comm_world = world;
  return RCSH_OK;
}

int rcsh_comm_rank(const rcsh_sim* s, int32_t* rank, int32_t* world) {
  if (!s) return fail(RCSH_ERR_ARG, "null sim handle");
  if (rank) *rank = s->comm_rank;
  if (world) *world = s->comm_world;
  return RCSH_OK;
}

int rcsh_comm_allgather_dev(rcsh_sim* s, int32_t slot, const void* send_dev, void* recv_dev, size_t bytes_per_rank) {
  REQUIRE_SIM(s);
  if (!s->comm && !s->copy) return fail(RCSH_ERR_STATE, "no communicator: call rcsh_comm_init (or rcsh_comm_copy_create) first");
  if (!send_dev || !recv_dev) return fail(RCSH_ERR_ARG, "null buffer");
  if (slot < 0 || slot > 1) return fail(RCSH_ERR_ARG, "exchange slot is 0 or 1");
  if (s->copy) return copy_allgather(s, slot, send_dev, recv_dev, bytes_per_rank);
  // after what the handle's stream holds so far (the env-step that wrote the observations), on the communicator's stream
  HIP_TRY(hipEventRecord(s->comm_ready, s->stream));
  HIP_TRY(hipStreamWaitEvent(s->comm_stream, s->comm_ready, 0));
  RCCL_TRY(rccl().AllGather(send_dev, recv_dev, bytes_per_rank, /* ncclInt8 */ 0, s->comm, s->comm_stream));
  HIP_TRY(hipEventRecord(s->comm_done[slot], s->comm_stream));
  s->comm_pending[slot] = true;
  return RCSH_OK;
}

int rcsh_env_allgather_obs_dev(rcsh_sim* s, int32_t slot, const double* local_obs_dev, double* all_obs_dev) {
  REQUIRE_SIM(s);
  return rcsh_comm_allgather_dev(s, slot, local_obs_dev, all_obs_dev, sizeof(double) * (size_t)s->n * (14 + s->narm));
}

int rcsh_comm_wait(rcsh_sim* s, int32_t slot, int32_t block_host) {
  REQUIRE_SIM(s);
  if (!s->comm && !s->copy) return fail(RCSH_ERR_STATE, "no communicator: call rcsh_comm_init (or rcsh_comm_copy_create) first");
  if (slot < 0 || slot > 1) return fail(RCSH_ERR_ARG, "exchange slot is 0 or 1");
  if (!s->comm_pending[slot]) return RCSH_OK;
  if (block_host) {
    HIP_TRY(hipEventSynchronize(s->comm_done[slot]));
    if (s->copy && *s->copy->timeout_flag) return fail(RCSH_ERR_DEVICE, "copy carrier: gave up waiting for a peer's block (a rank died?)");
    s->comm_pending[slot] = false;  // (a stream-side wait leaves it set: a later host-side wait must still see the event)
  } else {
    // (a stream-side wait cannot see a gather give up while it is in flight; it refuses to order consumers behind a carrier that HAS
    // given up on a peer -- the pinned flag, host-readable at any time: every later wait and post fails until the carrier is rebuilt;
    // advisor, round 5)
    if (s->copy && *s->copy->timeout_flag) return fail(RCSH_ERR_DEVICE, "copy carrier: an earlier gather gave up waiting for a peer's block (a rank died?)");
    HIP_TRY(hipStreamWaitEvent(s->stream, s->comm_done[slot], 0));
  }
  return RCSH_OK;
}

int rcsh_comm_destroy(rcsh_sim* s) {
  REQUIRE_SIM(s);
  if (!s->comm && !s->copy) return RCSH_OK;
  hipStreamSynchronize(s->comm_stream);
  if (s->copy) copy_carrier_free(s);
  else rccl().CommDestroy(s->comm);
  hipEventDestroy(s->comm_ready); hipEventDestroy(s->comm_done[0]); hipEventDestroy(s->comm_done[1]);
  hipStreamDestroy(s->comm_stream);
  s->comm = nullptr; s->comm_stream = nullptr; s->comm_ready = s->comm_done[0] = s->comm_done[1] = nullptr; s->comm_pending[0] = s->comm_pending[1] = false;
  s->comm_rank = 0; s->comm_world = 1;
  return RCSH_OK;
}


int rcsh_dev_alloc(rcsh_sim* s, size_t bytes, void** ptr) {
  REQUIRE_SIM(s);
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipMalloc(ptr, bytes));
  return RCSH_OK;
}
int rcsh_dev_free(rcsh_sim* s, void* ptr) {
  REQUIRE_SIM(s);
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipFree(ptr));
  return RCSH_OK;
}
int rcsh_dev_upload(rcsh_sim* s, void* dst, const void* src, size_t bytes) {
  REQUIRE_SIM(s);
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RCSH_OK;
}
int rcsh_dev_download(rcsh_sim* s, void* dst, const void* src, size_t bytes) {
  REQUIRE_SIM(s);
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RCSH_OK;
}

#ifdef RCSH_CHECK_TAIL
extern "C" int rcsh_debug_check_tail(unsigned long long* out16, int clear) {
  hipDeviceSynchronize();
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(rcsh::g_chk_tail), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
  if (clear) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(rcsh::g_chk_tail), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
#endif
#ifdef RCSH_WAVE_TIMES
extern "C" int rcsh_debug_wave_times(unsigned long long* out4x4096) {
  hipDeviceSynchronize();
  return hipMemcpyFromSymbol(out4x4096, HIP_SYMBOL(rcsh::g_wave_times), sizeof(unsigned long long) * 4 * 4096) == hipSuccess ? 0 : 1;
}
#endif
#ifdef RCSH_CHECK_TAIL
extern "C" int rcsh_debug_check_hist(unsigned long long* out64, int clear) {
  hipDeviceSynchronize();
  if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(rcsh::g_chk_hist), sizeof(unsigned long long) * 64) != hipSuccess) return 1;
  if (clear) { unsigned long long z[64] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(rcsh::g_chk_hist), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
#endif
#ifdef RCSH_CHECK_DEBUG
extern "C" int rcsh_debug_check(int* out64, int clear) {
  if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(rcsh::g_chk_dbg), sizeof(int) * 64) != hipSuccess) return 1;
  if (clear) { int z[64] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(rcsh::g_chk_dbg), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
extern "C" int rcsh_debug_check_f(double* out128) {
  return hipMemcpyFromSymbol(out128, HIP_SYMBOL(rcsh::g_chk_dbgf), sizeof(double) * 128) != hipSuccess;
}
extern "C" int rcsh_debug_check_cycles(unsigned long long* out16, int clear) {
  hipDeviceSynchronize();
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(rcsh::g_chk_cyc), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
  if (clear) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(rcsh::g_chk_cyc), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
extern "C" int rcsh_debug_check_pairs(rcsh_sim* s, int32_t* g0g1 /* [cap][2] */, int32_t cap, int32_t* n, int32_t* nb) {
  *n = (int)s->chk_pairs.size(); *nb = 0;
  for (int i = 0; i < *n && i < cap; ++i) { g0g1[2 * i] = s->cgeoms[s->chk_pairs[i].g0].geom_id; g0g1[2 * i + 1] = s->cgeoms[s->chk_pairs[i].g1].geom_id; }
  return 0;
}
#endif
#ifdef RCSH_PHASE_TIMING
extern "C" int rcsh_debug_slack(double* out16) {
  hipDeviceSynchronize();
  return hipMemcpyFromSymbol(out16, HIP_SYMBOL(rcsh::g_slack_dbg), sizeof(double) * 16) != hipSuccess;
}
extern "C" int rcsh_debug_team_cycles(unsigned long long* out16 /* 24 slots */) {
  hipDeviceSynchronize();
  return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_team_cycles), sizeof(unsigned long long) * 24) == hipSuccess ? 0 : 1;
}
extern "C" int rcsh_debug_team_cycles48(unsigned long long* out48) {
  hipDeviceSynchronize();
  return hipMemcpyFromSymbol(out48, HIP_SYMBOL(g_team_cycles), sizeof(unsigned long long) * 48) == hipSuccess ? 0 : 1;
}
extern "C" int rcsh_debug_team_cycles96(unsigned long long* out96, int clear_max) {
  hipDeviceSynchronize();
  if (hipMemcpyFromSymbol(out96, HIP_SYMBOL(g_team_cycles), sizeof(unsigned long long) * 96) != hipSuccess) return 1;
  if (clear_max) {  // (the maxima are per window)
    unsigned long long z = 0;
    hipMemcpyToSymbol(HIP_SYMBOL(g_team_cycles), &z, sizeof(z), sizeof(z) * 66);
    hipMemcpyToSymbol(HIP_SYMBOL(g_team_cycles), &z, sizeof(z), sizeof(z) * 68);
  }
  return 0;
}
extern "C" int rcsh_debug_team_cycles64(unsigned long long* out64) {
  hipDeviceSynchronize();
  return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_team_cycles), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : 1;
}
#endif

int rcsh_debug_dump_model(rcsh_sim* s, void* buf, size_t cap, size_t* size) {
  REQUIRE_SIM(s);
  if (size) *size = sizeof(DevModel);
  if (buf && cap >= sizeof(DevModel)) std::memcpy(buf, &s->dm, sizeof(DevModel));
  return RCSH_OK;
}

int rcsh_prof_enable(rcsh_sim* s, int32_t enable) {
  REQUIRE_SIM(s);
  HIP_TRY(hipSetDevice(s->device));
  if (enable && s->ev_start.empty()) {
    s->ev_start.resize(kProfRing);
    s->ev_stop.resize(kProfRing);
    for (int i = 0; i < kProfRing; ++i) {
      HIP_TRY(hipEventCreate(&s->ev_start[i]));
      HIP_TRY(hipEventCreate(&s->ev_stop[i]));
    }
  }
  s->prof = enable != 0;
  s->prof_region = enable < 0;
  s->prof_region_launches = 0;
  s->prof_every = enable > 1 ? enable : 1;
  s->prof_seen = 0;
  s->prof_pending = 0; s->prof_ms = 0; s->prof_launches = 0;
  return RCSH_OK;
}
int rcsh_prof_read(rcsh_sim* s, double* total_ms, int64_t* launches) {
  REQUIRE_SIM(s);
  if (s->prof_region) {
    float ms = 0.f;
    if (s->prof_region_launches > 0) {
      HIP_TRY(hipEventRecord(s->ev_stop[0], s->stream));
      HIP_TRY(hipEventSynchronize(s->ev_stop[0]));
      HIP_TRY(hipEventElapsedTime(&ms, s->ev_start[0], s->ev_stop[0]));
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = s->prof_region_launches;
    s->prof_region_launches = 0;
    return RCSH_OK;
  }
  int rc = prof_flush(s);
  if (rc) return rc;
  if (total_ms) *total_ms = s->prof_ms;
  if (launches) *launches = s->prof_launches;
  return RCSH_OK;
}

}  // extern "C"
