// render.h -- depth images of the batched scenes by ray casting: the counterpart of the reference's SimCameraSet
// pixel path (src/sim/camera.cpp:86-140: mjv_updateScene + mjr_render + mjr_readPixels on MuJoCo's OpenGL context).
//
// What is kept from the reference: the pinhole camera of MuJoCo (vertical field of view `fovy`, looking down -z, +y
// up; pixel centres on the viewport's grid), the frames of the LAST position stage (mjv_updateScene reads mjData.xpos /
// geom_xpos / cam_xpos, i.e. kinematics of the qpos the last mj_step1 saw), the OpenGL depth encoding
// d = (1/near - 1/z) / (1/near - 1/far) in [0, 1] as float32 with near / far = vis.map.znear / zfar times
// stat.extent, the bottom-up row order of glReadPixels, and the conversion the Python layer applies
// (python/rcs/camera/sim.py:57-86: row flip, z = near / (1 - d (1 - near / far)) in float32, x 1000, uint16).
// What differs: the pixels come from one ray per pixel against analytic shapes -- the floor plane, boxes, the convex
// hulls of the robot's collision meshes (rcs_amd/render.py) -- not from a rasteriser over the visual meshes, and there
// is no colour image.
//
// Two kernels: k_link_frames (one thread per environment: forward kinematics of the stored pre-step qpos, frames of
// all links and of the free box to HBM, 12 doubles each) and k_render_depth (one thread per pixel, 256 pixels of one
// environment per workgroup; the environment's shape frames are composed once per workgroup into LDS; per ray a
// bounding-sphere test per shape, then slabs / hull planes).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "box_team.h"
#include "dyn.h"
#include "model.h"

namespace rcsh {

constexpr int kMaxShapes = 32;
enum : int32_t { kShapePlane = 0, kShapeBox = 1, kShapeHull = 2, kShapeCapsule = 3 };
enum : int32_t { kLinkWorld = -1, kLinkFreeBody = -2 };

struct RenderShape {
  int32_t shape, link, plane_adr, plane_num;
  double pos[3], rot[9];  // shape frame in its link's frame
  double size[3];         // box: half extents; hull / capsule: half extents of the bounding box (capsule, axis z: radius = size[0], half length = size[2] - size[0])
  double sphere[4];       // bounding sphere: centre (shape frame), radius (< 0: unbounded)
  // hulls, the outline method (k_hull_views): the polytope's edges [edge_adr, edge_adr + edge_num) of RenderScene::edge_*,
  // a point inside it, and where this hull's view record starts within an environment's block of RenderScene::views
  int32_t edge_adr, edge_num;
  int64_t view_adr;       // in doubles
  double centre[3];
};
struct RenderCam {
  int32_t link, width, height, pad;
  double pos[3], rot[9];  // camera frame in its link's frame
  double tan_half_fovy;
  double tx, two_over_w, two_over_h;  // host-computed: tan_half_fovy * W / H, 2 / W, 2 / H (the kernels divide by nothing that is the same for every ray)
};
// colour of a shape: rgb; a checkered plane alternates rgb / rgb2 in squares of edge `square` (shape frame x, y)
struct RenderColour {
  double rgb[3], rgb2[3], square, checker;
};
// lighting of the colour image (fixed-function style, no shadows, no specular): the headlight at the camera and one
// directional light of the scene; rays that leave the scene see the sky gradient (zenith sky1, nadir sky2)
struct RenderShade {
  double ambient[3], head_diffuse[3], light_dir[3], light_diffuse[3], sky1[3], sky2[3];
};
struct RenderScene {
  int32_t nshape, nframes;  // nframes = links + 1 (the last entry is the free box, identity if the scene has none)
  double znear, zfar;
  double inv_near, inv_span;  // host-computed: 1 / znear, 1 / (1 / znear - 1 / zfar)
  const RenderShape* shapes;
  const double* planes;  // [.][4] n . x <= d
  const RenderColour* colours;  // [nshape], null until rcsh_sim_set_render_colours
  RenderShade shade;
  // the outline method (null / 0: every hull is walked plane by plane)
  const int32_t* edge_planes;  // [.][2] the two planes (indices within the hull) that meet in the edge
  const double* edge_verts;    // [.][6] its end points, shape frame
  double* views;               // [n][view_stride] per environment and hull: what k_hull_views found for the camera being rendered
  int64_t view_stride;         // in doubles
};
// A hull's view record: header (int32 nfront, int32 noutline (-1: not usable, walk the planes), the rest unused), then room for
// plane_num + kViewPad rows of the planes facing the camera, then kMaxOutline + kViewPad rows (mx, my, mz, 0) of the outline.
constexpr int kMaxOutline = 64;
constexpr int kViewHeaderDoubles = 8;  // (64 bytes: the rows behind it start on the boundary the 16-dword scalar loads like)
// Both lists of rows are padded with copies of their last row to a multiple of four (a repeated row changes no minimum): the ray
// caster reads four rows at a time with ONE scalar load and no clamping of indices.  The outline's rows start kViewPad rows
// behind the room for plane_num front rows.
constexpr int kViewPad = 4;
constexpr int64_t hull_view_doubles(int plane_num) { return (kViewHeaderDoubles + 4 * (int64_t)(plane_num + kViewPad + kMaxOutline + kViewPad) + 7) / 8 * 8; }  // (a multiple of 64 bytes)

#if defined(__HIP__)

// frames[e][i] = R(9) p(3) of link i for the qpos the last position stage saw; entry nl: the free box
template <class T>
__global__ void k_link_frames(const DevModel* gm, const double* S, int n, int qpre_field, int box_field, int has_box, double* frames) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const DevModel& m = *gm;
  double* out = frames + (size_t)e * (T::NL + 1) * 12;
  double Ra[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pa[3] = {0, 0, 0};  // frame of the arm link processed last
  for (int i = 0; i < T::NL; ++i) {
    // parent: the previous arm link; both fingers hang off the last arm link (Ra / pa stop advancing there)
    const double q = S[(size_t)(qpre_field + i) * n + e] - m.qpos0[i];
    double o[3], R0[9], R[9], p[3];
    mulmv(Ra, m.pos0[i], o);
    o[0] += pa[0]; o[1] += pa[1]; o[2] += pa[2];
    mulmm(Ra, m.rot0[i], R0);
    if (m.jtype[i] == kHinge) {
      double s, c;
      fast_sincos(q, &s, &c);
      const double* u = m.axis[i];
      const double t = 1.0 - c;
      const double Q[9] = {c + t * u[0] * u[0],        t * u[0] * u[1] - s * u[2], t * u[0] * u[2] + s * u[1],
                           t * u[0] * u[1] + s * u[2], c + t * u[1] * u[1],        t * u[1] * u[2] - s * u[0],
                           t * u[0] * u[2] - s * u[1], t * u[1] * u[2] + s * u[0], c + t * u[2] * u[2]};
      double an[3], rj[3];
      mulmv(R0, m.jpos[i], an);
      mulmm(R0, Q, R);
      mulmv(R, m.jpos[i], rj);
      for (int k = 0; k < 3; ++k) p[k] = an[k] + o[k] - rj[k];
    } else {
      double a[3];
      mulmv(R0, m.axis[i], a);
      for (int k = 0; k < 9; ++k) R[k] = R0[k];
      for (int k = 0; k < 3; ++k) p[k] = o[k] + q * a[k];
    }
    for (int k = 0; k < 9; ++k) out[12 * i + k] = R[k];
    for (int k = 0; k < 3; ++k) out[12 * i + 9 + k] = p[k];
    if (i < T::NARM) {
      for (int k = 0; k < 9; ++k) Ra[k] = R[k];
      for (int k = 0; k < 3; ++k) pa[k] = p[k];
    }
  }
  double* b = out + 12 * T::NL;
  if (has_box) {
    const double w = S[(size_t)(box_field + kBoxPre + 3) * n + e], x = S[(size_t)(box_field + kBoxPre + 4) * n + e],
                 y = S[(size_t)(box_field + kBoxPre + 5) * n + e], z = S[(size_t)(box_field + kBoxPre + 6) * n + e];
    b[0] = w * w + x * x - y * y - z * z; b[4] = w * w - x * x + y * y - z * z; b[8] = w * w - x * x - y * y + z * z;
    b[1] = 2 * (x * y - w * z); b[2] = 2 * (x * z + w * y); b[3] = 2 * (x * y + w * z);
    b[5] = 2 * (y * z - w * x); b[6] = 2 * (x * z - w * y); b[7] = 2 * (y * z + w * x);
    for (int k = 0; k < 3; ++k) b[9 + k] = S[(size_t)(box_field + kBoxPre + k) * n + e];
  } else {
    for (int k = 0; k < 12; ++k) b[k] = (k == 0 || k == 4 || k == 8) ? 1.0 : 0.0;
  }
}

// world frame of something given in a link's frame
__device__ __forceinline__ void in_world(const double* frames, int link, int nframes, const double* pos, const double* rot, double* R, double* p) {
  if (link == kLinkWorld) {
    for (int k = 0; k < 9; ++k) R[k] = rot[k];
    for (int k = 0; k < 3; ++k) p[k] = pos[k];
    return;
  }
  const double* f = frames + 12 * (link == kLinkFreeBody ? nframes - 1 : link);
  mulmm(f, rot, R);
  mulmv(f, pos, p);
  p[0] += f[9]; p[1] += f[10]; p[2] += f[11];
}

// Per environment and camera, one row per shape g < nshape and one for the camera (entry nshape).  One thread per (environment, entry).
//   shape row: [0..8] R, [9..11] p (world; k_hull_views), [12..14] bounding sphere's centre (world), [15] its radius (< 0: unbounded),
//     [16..18] half extents, [19..21] centre of the slabs in the shape frame (box: 0; hull / capsule: the bounding sphere's),
//     and what a RAY of this camera needs -- none of it depends on the pixel, so no ray computes it:
//     [22..30] M = R' cR with the third column negated: a ray (x, y, -1) of the camera frame has direction
//              ld_i = M[3i] x + M[3i+1] y + M[3i+2] in the shape frame (two multiply-adds a component),
//     [31..33] lo = R' (o - p): the camera's position in the shape frame (every ray starts there),
//     [34..36] the sphere's centre in the camera frame, [37..39] R' light_dir (colour: the directional light in the shape frame),
//     [40] |centre - o|^2 - radius^2 (the ray-sphere test's constant), [41..43] unused;
//   camera row: [0..8] cR, [9..11] cp, zeros.
// The ray caster stages an environment's rows in LDS in ITS arithmetic type and reads nothing else per shape but its planes.
constexpr int kShapeFrameDoubles = 44;
constexpr int kRowSize = 16, kRowCen = 19, kRowM = 22, kRowLo = 31, kRowSc = 34, kRowLight = 37, kRowK = 40;
__global__ void k_shape_frames(RenderScene sc, RenderCam cam, const double* frames, int n, double* wf) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int per_env = sc.nshape + 1;
  if (idx >= n * per_env) return;
  const int e = idx / per_env, g = idx % per_env;
  const double* fe = frames + (size_t)e * sc.nframes * 12;
  double* out = wf + (size_t)idx * kShapeFrameDoubles;
  double cR[9], cp[3];
  in_world(fe, cam.link, sc.nframes, cam.pos, cam.rot, cR, cp);
  if (g == sc.nshape) {
    for (int k = 0; k < 9; ++k) out[k] = cR[k];
    for (int k = 0; k < 3; ++k) out[9 + k] = cp[k];
    for (int k = 12; k < kShapeFrameDoubles; ++k) out[k] = 0.0;
    return;
  }
  const RenderShape& sh = sc.shapes[g];
  double R[9], p[3];
  in_world(fe, sh.link, sc.nframes, sh.pos, sh.rot, R, p);
  double c[3];
  mulmv(R, sh.sphere, c);
  for (int k = 0; k < 3; ++k) c[k] += p[k];
  for (int k = 0; k < 9; ++k) out[k] = R[k];
  for (int k = 0; k < 3; ++k) { out[9 + k] = p[k]; out[12 + k] = c[k]; }
  out[15] = sh.sphere[3];
  // slabs: of the box -- or, for a hull / capsule, of its bounding box (centre = the bounding sphere's, half extents in `size`)
  for (int k = 0; k < 3; ++k) { out[kRowSize + k] = sh.size[k]; out[kRowCen + k] = sh.shape != kShapeBox ? sh.sphere[k] : 0.0; }
  const double om[3] = {cp[0] - p[0], cp[1] - p[1], cp[2] - p[2]}, q[3] = {c[0] - cp[0], c[1] - cp[1], c[2] - cp[2]};
  double k2 = -sh.sphere[3] * sh.sphere[3];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      const double m = R[i] * cR[j] + R[3 + i] * cR[3 + j] + R[6 + i] * cR[6 + j];
      out[kRowM + 3 * i + j] = j == 2 ? -m : m;
    }
    out[kRowLo + i] = R[i] * om[0] + R[3 + i] * om[1] + R[6 + i] * om[2];
    const double sc_i = cR[i] * q[0] + cR[3 + i] * q[1] + cR[6 + i] * q[2];
    out[kRowSc + i] = sc_i;
    k2 += sc_i * sc_i;
    out[kRowLight + i] = R[i] * sc.shade.light_dir[0] + R[3 + i] * sc.shade.light_dir[1] + R[6 + i] * sc.shade.light_dir[2];
  }
  out[kRowK] = k2;
  out[41] = out[42] = out[43] = 0.0;
}

// The ray caster's arithmetic type.  F = float is the product's (the reference's depth image IS a float32 z-buffer read back
// and quantised to uint16 millimetres, python/rcs/camera/sim.py:57-86: double precision buys nothing the reference has);
// F = double is kept as the instantiation whose pixels equal the numpy restatement's bit for bit (rcsh_sim_set_render_f64).
template <class F> struct RenderNum;
template <> struct RenderNum<double> {
  static __device__ __forceinline__ double rcp(double x) { return fast_rcp(x); }
  static __device__ __forceinline__ double rsqrt(double x) { return 1.0 / sqrt(x); }
};
template <> struct RenderNum<float> {
  static __device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
  static __device__ __forceinline__ float rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
};

// The outline method.  All rays of a camera start at one point o.  Seen from o a convex polytope has FRONT faces (o outside
// their plane: n . o > d) and back faces; a ray can only ENTER through a front face, and it hits the polytope at all exactly
// when its direction lies inside the cone from o over the polytope's outline -- the closed chain of edges where a front face
// meets a back face.  So per (environment, camera, hull), once: classify the ~110 face planes (keeping d - n . o, which every
// ray needs and which is the same for all of them), find the ~40 outline edges among the polytope's ~330, and turn each into
// the plane through o that contains it, normal pointing at the hull.  A ray then costs one dot product per outline edge (inside
// all of them = hit) and one per FRONT plane (the entry depth is the largest no / nd) -- ~85 dot products instead of the two per
// plane, ~220, of walking every face plane for both ends of the ray's interval.  What a ray sees is the same set of points;
// rays that graze the outline may fall on the other side of it by round-off (as between any two ways of writing the test).
// One wavefront per (environment, hull); lanes take planes, then edges; ballots compact the survivors into the view record.
// The record's rows are written in the ray caster's type F (worked out in double here: d - n . o is a difference of metres).
// A front plane's row is m = n / (d - n . o): the INVERSE of the depth at which a ray of direction ld meets the plane is m . ld,
// and the entry depth of a ray inside the cone -- the largest of the planes' depths -- is 1 / min (m . ld): three multiply-adds
// and a minimum per plane and ray, one division per ray (round 3 kept (n, d - n . o) and spent two comparisons, a multiplication
// and now and then a division per plane).
__device__ __forceinline__ double plane_no(const double* q, const double* lo) {
  return fma(-q[2], lo[2], fma(-q[1], lo[1], fma(-q[0], lo[0], q[3])));  // d - n . o, one rounding order everywhere it is needed
}
template <class F>
__global__ void __launch_bounds__(64) k_hull_views(RenderScene sc, RenderCam cam, const double* wf, int n) {
  const int e = blockIdx.x / sc.nshape, g = blockIdx.x % sc.nshape;
  if (e >= n) return;
  const RenderShape& sh = sc.shapes[g];
  if (sh.shape != kShapeHull || sh.edge_num <= 0) return;
  const int lane = threadIdx.x;
  const double* w = wf + ((size_t)e * (sc.nshape + 1) + g) * kShapeFrameDoubles;           // the hull's frame: R (9) p (3)
  const double* cw = wf + ((size_t)e * (sc.nshape + 1) + sc.nshape) * kShapeFrameDoubles;  // the camera's
  {
    // a hull the camera cannot see (its bounding sphere outside the image's pyramid of rays or behind the near plane) is never
    // visited by a ray -- k_render_depth culls it per tile with the same test -- so its view is not needed: the wrist camera
    // sees the fingers and little else of the arm
    const double q[3] = {w[12] - cw[9], w[13] - cw[10], w[14] - cw[11]}, r = w[15];
    const double x = cw[0] * q[0] + cw[3] * q[1] + cw[6] * q[2], y = cw[1] * q[0] + cw[4] * q[1] + cw[7] * q[2], z = cw[2] * q[0] + cw[5] * q[1] + cw[8] * q[2];
    const double tx = cam.tx, ty = cam.tan_half_fovy, r2 = r * r;
    const double sl = x - tx * z, sr = -x - tx * z, sb = y - ty * z, st = -y - ty * z;  // (the image spans x in [-tx, tx] (-z), y in [-ty, ty] (-z))
    const bool visible = -z + r >= sc.znear && (sl >= 0 || sl * sl <= r2 * (1 + tx * tx)) && (sr >= 0 || sr * sr <= r2 * (1 + tx * tx)) &&
                         (sb >= 0 || sb * sb <= r2 * (1 + ty * ty)) && (st >= 0 || st * st <= r2 * (1 + ty * ty));
    if (!visible) {
      // (should a tile's own cull let it through after all -- the two tests are not the same inequality -- the rays walk its planes)
      if (lane == 0) { int32_t* hdr = (int32_t*)(sc.views + (size_t)e * sc.view_stride + sh.view_adr); hdr[0] = 0; hdr[1] = -1; }
      return;
    }
  }
  const double om[3] = {cw[9] - w[9], cw[10] - w[10], cw[11] - w[11]};
  const double lo[3] = {w[0] * om[0] + w[3] * om[1] + w[6] * om[2], w[1] * om[0] + w[4] * om[1] + w[7] * om[2], w[2] * om[0] + w[5] * om[1] + w[8] * om[2]};
  double* out = sc.views + (size_t)e * sc.view_stride + sh.view_adr;
  F* rows = (F*)(out + kViewHeaderDoubles);
  const double* planes = sc.planes + 4 * (size_t)sh.plane_adr;
  const uint64_t below = (1ull << lane) - 1ull;
  int nfront = 0;
  double lastf[4] = {0, 0, 0, 0};  // the last front row written (every lane keeps a copy: the padding)
  for (int base = 0; base < sh.plane_num; base += 64) {
    const int i = base + lane;
    bool front = false;
    double q[4] = {0, 0, 0, 0}, no = 0, row[4] = {0, 0, 0, 0};
    if (i < sh.plane_num) {
      for (int k = 0; k < 4; ++k) q[k] = planes[4 * i + k];
      no = plane_no(q, lo);
      front = no < 0;
    }
    const uint64_t m = __ballot(front);
    if (front) {
      F* r = rows + 4 * (size_t)(nfront + __popcll(m & below));
      const double inv = 1.0 / (no < -1e-30 ? no : -1e-30);
      row[0] = q[0] * inv; row[1] = q[1] * inv; row[2] = q[2] * inv; row[3] = no;
      r[0] = (F)row[0]; r[1] = (F)row[1]; r[2] = (F)row[2]; r[3] = (F)row[3];
    }
    if (m) {
      const int src = 63 - __clzll((long long)m);
      for (int k = 0; k < 4; ++k) lastf[k] = __shfl(row[k], src);
    }
    nfront += __popcll(m);
  }
  if (lane < kViewPad - 1) {
    F* r = rows + 4 * (size_t)(nfront + lane);
    for (int k = 0; k < 4; ++k) r[k] = (F)lastf[k];
  }
  F* outline = rows + 4 * (size_t)(sh.plane_num + kViewPad);
  const double ci[3] = {sh.centre[0] - lo[0], sh.centre[1] - lo[1], sh.centre[2] - lo[2]};
  int nout = 0;
  double lasto[3] = {0, 0, 0};
  for (int base = 0; base < sh.edge_num; base += 64) {
    const int k = base + lane;
    bool sil = false;
    double mm[3] = {0, 0, 0};
    if (k < sh.edge_num) {
      const int32_t* ep = sc.edge_planes + 2 * (size_t)(sh.edge_adr + k);
      sil = (plane_no(planes + 4 * ep[0], lo) < 0) != (plane_no(planes + 4 * ep[1], lo) < 0);
      if (sil) {
        const double* ev = sc.edge_verts + 6 * (size_t)(sh.edge_adr + k);
        const double a[3] = {ev[0] - lo[0], ev[1] - lo[1], ev[2] - lo[2]}, b[3] = {ev[3] - lo[0], ev[4] - lo[1], ev[5] - lo[2]};
        mm[0] = a[1] * b[2] - a[2] * b[1]; mm[1] = a[2] * b[0] - a[0] * b[2]; mm[2] = a[0] * b[1] - a[1] * b[0];
        if (mm[0] * ci[0] + mm[1] * ci[1] + mm[2] * ci[2] < 0) { mm[0] = -mm[0]; mm[1] = -mm[1]; mm[2] = -mm[2]; }
        if (sizeof(F) == 4) {
          // (float rows: the cross product of two ~metre vectors spans many orders of magnitude; only its direction matters)
          const double s = 1.0 / sqrt(mm[0] * mm[0] + mm[1] * mm[1] + mm[2] * mm[2] + 1e-300);
          mm[0] *= s; mm[1] *= s; mm[2] *= s;
        }
      }
    }
    const uint64_t m = __ballot(sil);
    const int r = nout + __popcll(m & below);
    if (sil && r < kMaxOutline) {
      F* o = outline + 4 * (size_t)r;
      o[0] = (F)mm[0]; o[1] = (F)mm[1]; o[2] = (F)mm[2]; o[3] = 0;
    }
    if (m) {
      const int src = 63 - __clzll((long long)m);
      for (int c = 0; c < 3; ++c) lasto[c] = __shfl(mm[c], src);
    }
    nout += __popcll(m);
  }
  if (lane < kViewPad - 1 && nout <= kMaxOutline) {
    F* o = outline + 4 * (size_t)(nout + lane);
    o[0] = (F)lasto[0]; o[1] = (F)lasto[1]; o[2] = (F)lasto[2]; o[3] = 0;
  }
  if (lane == 0) {
    int32_t* hdr = (int32_t*)out;
    hdr[0] = nfront;
    hdr[1] = nout <= kMaxOutline ? nout : -1;
  }
}

// depth_gl: [n][H][W] float32 in [0, 1], rows bottom-up (mjr_readPixels); depth_mm: [n][H][W] uint16, rows top-down,
// millimetres (SimCameraSet with physical_units); cam_pose: [n][12] world rotation (9) and position (3) of the camera
// (mjData.cam_xmat / cam_xpos).  Any of the three may be null.
// COLOR: also rgb [n][H][W][3] uint8, rows bottom-up like the depth buffer: the colour of the shape the ray enters first,
// lit by the headlight and the scene's directional light on the entry face's normal (flat shading; sc.colours).
//
// Work decomposition (round 4).  The kernel is bound by the number of instructions it issues -- vector AND scalar (rocprofv3: before
// the round's changes 2132 scalar against 1542 vector instructions per wavefront, no memory stall to speak of) -- so the design is
// about instructions per ray.  A wavefront renders a 16 x 16 pixel tile in two passes of 16 x 8 rays, two rays a lane, and
// kTilesPerWave tiles one after the other; a workgroup's four wavefronts share nothing but
// the environment's shape rows in LDS, staged ONCE per workgroup.  Per tile, the wavefront's own business, no barrier: lane g culls
// shape g against the tile's pyramid of rays, a ballot gives the tile's shapes, ranks come from lane reads, the next shape to visit
// from a ballot -- ~100 instructions per 256 rays (round 3: a workgroup per tile, the cull by one wavefront while three waited at
// one of three barriers; an intermediate version with the cull per 8 x 8 sub-tile spent a quarter of its instructions there).  Per
// ray and shape nothing is computed that does not depend on the pixel: the camera's position and the ray's direction in the shape
// frame come from k_shape_frames' row (two multiply-adds per component instead of two rotations), the sphere test runs in the
// camera frame on three numbers, a hull's front planes cost three multiply-adds and a minimum each (k_hull_views).
constexpr int kTilesPerWave = 2;
__host__ __device__ inline int render_wgs_per_env(int W, int H) { return (((W + 15) / 16) * ((H + 15) / 16) + 4 * kTilesPerWave - 1) / (4 * kTilesPerWave); }

// four rows of a view record (4 x (x, y, z, w) of F) through the constant address space: one 16-dword scalar load for floats, two for doubles
template <class F> struct RowLoad;
template <> struct RowLoad<float> {
  typedef float V __attribute__((ext_vector_type(16)));
  static __device__ __forceinline__ void load4(const float __attribute__((address_space(4)))* p, float (&q)[4][4]) {
    const V v = *(const V __attribute__((address_space(4)))*)p;
#pragma unroll
    for (int j = 0; j < 16; ++j) q[j / 4][j % 4] = v[j];
  }
};
template <> struct RowLoad<double> {
  typedef double V __attribute__((ext_vector_type(8)));
  static __device__ __forceinline__ void load4(const double __attribute__((address_space(4)))* p, double (&q)[4][4]) {
    const V a = *(const V __attribute__((address_space(4)))*)p, b = *((const V __attribute__((address_space(4)))*)p + 1);
#pragma unroll
    for (int j = 0; j < 8; ++j) { q[j / 4][j % 4] = a[j]; q[2 + j / 4][j % 4] = b[j]; }
  }
};
template <class F> __device__ __forceinline__ F render_min(F a, F b);
template <> __device__ __forceinline__ float render_min<float>(float a, float b) { return __builtin_fminf(a, b); }
template <> __device__ __forceinline__ double render_min<double>(double a, double b) { return __builtin_fmin(a, b); }
template <class F> __device__ __forceinline__ F render_max(F a, F b);
template <> __device__ __forceinline__ float render_max<float>(float a, float b) { return __builtin_fmaxf(a, b); }
template <> __device__ __forceinline__ double render_max<double>(double a, double b) { return __builtin_fmax(a, b); }

template <bool COLOR, class F>
__global__ void __launch_bounds__(256) k_render_depth(RenderScene sc, RenderCam cam, const double* wf, int n, float* depth_gl,
                                                      uint16_t* depth_mm, double* cam_pose, uint8_t* rgb) {
  using Num = RenderNum<F>;
  // A lane casts TWO rays, through horizontally adjacent pixels, as the halves of two-component vectors: in float the compiler
  // turns their multiply-adds into packed instructions (v_pk_fma_f32: two rays' worth per issue), and whatever is the wavefront's
  // or the lane's own -- loop control, scalar loads of rows, the shape's constants from LDS -- is paid once per 128 rays.
  typedef F V2 __attribute__((ext_vector_type(2)));
  __shared__ F lw[(kMaxShapes + 1) * kShapeFrameDoubles];  // this environment's rows of wf (k_shape_frames)
  const int W = cam.width, H = cam.height;
  const int tiles_x = (W + 15) / 16, ntile = tiles_x * ((H + 15) / 16);
  const int wgs_per_env = render_wgs_per_env(W, H);
  // Workgroups go to the chip's 8 XCDs round robin by their index.  The workgroups of one environment read the same shape frames
  // and hull views; numbered so that they follow each other on ONE XCD they find them in that XCD's L2, instead of all eight L2s
  // fetching every environment's rows -- and every XCD gets whole environments, i.e. an even share of the rays that meet a hull.
  // (The grid is rounded up to a multiple of 8.)
  const int nblocks = n * wgs_per_env, per_xcd = (nblocks + 7) / 8;
#ifdef RCSH_NO_XCD_REMAP
  const int block = (int)blockIdx.x;
#else
  const int block = (int)(blockIdx.x % 8) * per_xcd + (int)(blockIdx.x / 8);
#endif
  if (block >= nblocks) return;
  const int e = block / wgs_per_env, part = block % wgs_per_env;
  const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
  const F ty = (F)cam.tan_half_fovy, tx = (F)cam.tx, two_over_w = (F)cam.two_over_w, two_over_h = (F)cam.two_over_h;
  const F znear = (F)sc.znear, zfar = (F)sc.zfar, inv_near = (F)sc.inv_near;
  {
    const int words = (sc.nshape + 1) * kShapeFrameDoubles;
    const double* src = wf + (size_t)e * words;
    for (int k = threadIdx.x; k < words; k += 256) lw[k] = (F)src[k];
    if (cam_pose && part == 0 && threadIdx.x < 12) cam_pose[(size_t)e * 12 + threadIdx.x] = src[sc.nshape * kShapeFrameDoubles + threadIdx.x];
  }
  __syncthreads();
  // lane g < nshape: shape g as the camera sees it -- its bounding sphere's centre in the camera frame, its radius, and the
  // key the wavefront's visits are ordered by (below) -- the same for every tile
  const bool isshape = lane < sc.nshape;
  F sx = 0, sy = 0, sz = 0, sr = -1, key = (F)INFINITY;
  if (isshape) {
    const F* w = lw + lane * kShapeFrameDoubles;
    sr = w[15];
    sx = w[kRowSc]; sy = w[kRowSc + 1]; sz = w[kRowSc + 2];
    // Front to back.  A ray only needs the NEAREST entry point, and the slab test below starts from t1 = best: a shape whose box
    // begins behind the nearest hit so far costs three slabs instead of its outline and front planes.  Seen from above an arm is a
    // stack of links, each ray's pyramid crossing most of their boxes.  The shapes are ranked by the view depth of their box's
    // nearest point (planes first: one cheap test that bounds `best`).  Which shape is hit does not depend on the order (ties
    // between two shapes' entry depths aside), the depth never does.
    if (sc.shapes[lane].shape == kShapePlane) key = -(F)INFINITY;
    else {
      // view depth = -(z of the camera frame); the camera's z axis in the shape frame is the third column of R' cR (stored negated)
      key = -sz;
#pragma unroll
      for (int k = 0; k < 3; ++k) key -= w[kRowSize + k] * fabs(w[kRowM + 3 * k + 2]);
    }
  }
  const F* cR = lw + sc.nshape * kShapeFrameDoubles;
  auto min2 = [](V2 a, V2 b) { return V2{render_min<F>(a.x, b.x), render_min<F>(a.y, b.y)}; };
  auto max2 = [](V2 a, V2 b) { return V2{render_max<F>(a.x, b.x), render_max<F>(a.y, b.y)}; };
  for (int ti = 0; ti < kTilesPerWave; ++ti) {
    const int tile = (part * 4 + wave) * kTilesPerWave + ti;  // (the wavefront's: every test on it is a scalar branch)
    if (tile >= ntile) break;
    const int c0 = (tile % tiles_x) * 16, r0 = (tile / tiles_x) * 16;
    // does shape g's bounding sphere reach into the pyramid of this tile's rays (four planes through the camera, and the near
    // plane)?  The tile spans x in [xl, xr] (-z), y in [yb, yt] (-z)
    bool visible = isshape;
    if (isshape && sr >= 0) {
      const F xl = (c0 * two_over_w - 1) * tx, xr = ((c0 + 16) * two_over_w - 1) * tx;
      const F yb = (r0 * two_over_h - 1) * ty, yt = ((r0 + 16) * two_over_h - 1) * ty;
      // signed distance of the centre to each side plane of the pyramid, times that plane's normal's length: on the inner side,
      // or no further out than the radius (squares: no square root)
      const F r2 = sr * sr;
      const F sl = sx + xl * sz, srr = -sx - xr * sz, sb = sy + yb * sz, st = -sy - yt * sz;
      visible = -sz + sr >= znear && (sl >= 0 || sl * sl <= r2 * (1 + xl * xl)) && (srr >= 0 || srr * srr <= r2 * (1 + xr * xr)) &&
                (sb >= 0 || sb * sb <= r2 * (1 + yb * yb)) && (st >= 0 || st * st <= r2 * (1 + yt * yt));
    }
    const uint32_t wave_shapes = (uint32_t)__ballot(visible);
    const int nvisit = __popc(wave_shapes);
    // (few shapes -- the wrist camera's usual view: floor, cube, finger pads -- are visited in index order: nothing to gain)
    int rank = __popc(wave_shapes & ((1u << (lane & 31)) - 1u));
    if (nvisit > 4) {
      rank = 0;
      for (uint32_t m = wave_shapes; m; m &= m - 1) {
        const int g = __ffs(m) - 1;
        const F kg = __shfl(key, g);
        rank += (kg < key || (kg == key && g < lane)) ? 1 : 0;
      }
    }
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      // 16 x 8 pixels a pass: lane (i, j) of the 8 x 8 takes the pixels (c0 + 2 i, c0 + 2 i + 1) of row r0 + 8 pass + j
      if (r0 + pass * 8 >= H) continue;  // (the wavefront's)
      const int col = c0 + 2 * (lane & 7);
      const int row = r0 + pass * 8 + (lane >> 3);  // row 0 = bottom of the image (OpenGL window coordinates)
      const bool in_image[2] = {col < W && row < H, col + 1 < W && row < H};
      // rays through the pixel centres, camera frame: (x, y, -1), so that the ray parameter IS the view depth z
      const V2 x = {((col + (F)0.5) * two_over_w - 1) * tx, ((col + (F)1.5) * two_over_w - 1) * tx};
      const F y = ((row + (F)0.5) * two_over_h - 1) * ty;
      const V2 dd = x * x + (y * y + 1);
      V2 best = {zfar, zfar};
      bool hit[2] = {false, false};
      // COLOR: the shape entered first and where -- a plane index of a hull, axis (0..2) and side of a box
      int hit_g[2] = {-1, -1}, hit_face[2] = {0, 0};
      bool hit_outline[2] = {false, false};  // hit_face counts the hull's FRONT planes (its view record), not its planes
      for (int vi = 0; vi < nvisit; ++vi) {
        // (the visit list is the wavefront's: the shape's constants and its rows come through scalar loads)
        const int g = __ffsll((unsigned long long)__ballot(visible && rank == vi)) - 1;
        const RenderShape& sh = sc.shapes[g];
        const F* w = lw + g * kShapeFrameDoubles;
        bool ok[2] = {true, true};
        if (w[15] >= 0) {
          // bounding sphere: closest approach of the ray to the centre, in the camera frame
          const V2 b = w[kRowSc] * x + (w[kRowSc + 1] * y - w[kRowSc + 2]);
          const V2 lhs = w[kRowK] * dd, rhs = b * b;
          ok[0] = !(lhs.x > rhs.x); ok[1] = !(lhs.y > rhs.y);
          if (!ok[0] && !ok[1]) continue;
        }
        // rays in the shape's frame (they share the origin, and whatever of the direction the row of pixels fixes)
        const F lo[3] = {w[kRowLo], w[kRowLo + 1], w[kRowLo + 2]};
        const V2 ld[3] = {w[kRowM] * x + (w[kRowM + 1] * y + w[kRowM + 2]), w[kRowM + 3] * x + (w[kRowM + 4] * y + w[kRowM + 5]),
                          w[kRowM + 6] * x + (w[kRowM + 7] * y + w[kRowM + 8])};
        const int shape = __builtin_amdgcn_readfirstlane(sh.shape);
        if (shape == kShapePlane) {
          // the plane z = 0 of the shape frame, seen from above (MuJoCo draws planes one-sided)
          if (!(lo[2] > 0)) continue;
          const V2 t = -lo[2] * V2{Num::rcp(ld[2].x), Num::rcp(ld[2].y)};
#pragma unroll
          for (int c = 0; c < 2; ++c)
            if (ok[c] && ld[2][c] < 0 && t[c] >= znear && t[c] < best[c]) { best[c] = t[c]; hit[c] = true; if (COLOR) { hit_g[c] = g; hit_face[c] = 0; } }
          continue;
        }
        V2 t0 = {znear, znear}, t1 = best;
        // slabs of the box -- or, for a hull, of its bounding box first (centre = the bounding sphere's, half extents in
        // `size`): most rays that pass the sphere of an elongated link miss the link
        int face[2] = {0, 0};
        {
          // (no branch: an axis the ray runs along -- ld_k = 0 -- gives infinite or undefined bounds, which the minima and maxima
          // pass over exactly as the test |lk| <= size would)
          V2 b0 = t0, b1 = t1;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const F lk = lo[k] - w[kRowCen + k], sz_k = w[kRowSize + k];
            const V2 inv = {Num::rcp(ld[k].x), Num::rcp(ld[k].y)};
            const V2 tp = (-sz_k - lk) * inv, tq = (sz_k - lk) * inv;
            const V2 ta = min2(tp, tq), tb = max2(tp, tq);
            if (COLOR) {
#pragma unroll
              for (int c = 0; c < 2; ++c)
                if (ta[c] > b0[c]) face[c] = ld[k][c] > 0 ? 2 * k : 2 * k + 1;  // entered through the -k (even) or the +k (odd) face
            }
            b0 = max2(b0, ta);
            b1 = min2(b1, tb);
          }
          ok[0] = ok[0] && b0.x <= b1.x; ok[1] = ok[1] && b0.y <= b1.y;
          if (shape == kShapeBox) { t0 = b0; t1 = b1; }
        }
        if ((ok[0] || ok[1]) && shape == kShapeCapsule) {
          // capsule about the shape frame's z axis: the ray's first point on the cylinder's wall between the caps, or on the outer
          // half of a cap sphere -- the smallest of the (at most three) candidates, the surface being convex
          const F r = w[kRowSize], hl = w[kRowSize + 2] - w[kRowSize];
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const F l0 = ld[0][c], l1 = ld[1][c], l2 = ld[2][c];
            F te = (F)INFINITY;
            const F a = l0 * l0 + l1 * l1, bq = lo[0] * l0 + lo[1] * l1, cq = lo[0] * lo[0] + lo[1] * lo[1] - r * r;
            const F disc = bq * bq - a * cq;
            if (a > 0 && disc >= 0) {
              const F t = (-bq - sqrt(disc)) / a;
              if (fabs(lo[2] + t * l2) <= hl) te = t;
            }
            const F A = a + l2 * l2;
#pragma unroll
            for (int side = 0; side < 2; ++side) {
              const F zc = side ? hl : -hl, oz = lo[2] - zc;
              const F B = bq + oz * l2, Cq = cq + oz * oz;
              const F ds = B * B - A * Cq;
              if (ds >= 0) {
                const F t = (-B - sqrt(ds)) / A;
                const F zr = oz + t * l2;  // of the point, from the cap's centre
                if ((side ? zr >= 0 : zr <= 0) && t < te) te = t;
              }
            }
            bool okc = ok[c] && te < (F)INFINITY;
            t0[c] = okc && te > t0[c] ? te : t0[c];
            ok[c] = okc && te >= znear && t0[c] <= t1[c];
          }
        }
#ifdef RCSH_RENDER_NOWALK
        if (shape == kShapeHull) ok[0] = ok[1] = false;  // (measurement: everything but the hulls' own tests)
#endif
#ifdef RCSH_RENDER_FLOORONLY
        ok[0] = ok[1] = false;  // (measurement: the floor and the bookkeeping)
#endif
        bool by_outline = false;
        typedef const double __attribute__((address_space(4))) kdouble;
        typedef const F __attribute__((address_space(4))) kF;
        const int plane_num = __builtin_amdgcn_readfirstlane(sh.plane_num);
        if ((ok[0] || ok[1]) && shape == kShapeHull && sc.views != nullptr && sh.edge_num > 0) {
          // the outline method (k_hull_views): this environment's record of the hull as the camera sees it.  The address is the
          // wavefront's (e is the workgroup's, g the wavefront's): header and rows come through scalar loads.
          kdouble* vw = (kdouble*)(sc.views + (size_t)e * sc.view_stride + sh.view_adr);
          typedef const int32_t __attribute__((address_space(4))) kint;
          const int nfront = __builtin_amdgcn_readfirstlane(((kint*)vw)[0]), nout = __builtin_amdgcn_readfirstlane(((kint*)vw)[1]);
          if (nout >= 0) {
            by_outline = true;
            kF* fr = (kF*)(vw + kViewHeaderDoubles);
            kF* ol = fr + 4 * (size_t)(plane_num + kViewPad);
            // inside the cone over the outline?  Four rows a round: ONE scalar load (the lists are padded to multiples of four with
            // copies of their last row), the smallest of the four products decides -- the vector unit takes the minimum, the scalar
            // unit, which this kernel keeps as busy as the vector unit, is asked once per round
            if (!(nfront > 0 && nout > 0)) ok[0] = ok[1] = false;
            for (int k = 0; k < nout && (ok[0] || ok[1]); k += 4) {
              F q[4][4];
              RowLoad<F>::load4(ol + 4 * (size_t)k, q);
              V2 smin = q[0][0] * ld[0] + q[0][1] * ld[1] + q[0][2] * ld[2];
#pragma unroll
              for (int j = 1; j < 4; ++j) smin = min2(smin, q[j][0] * ld[0] + q[j][1] * ld[1] + q[j][2] * ld[2]);
              ok[0] = ok[0] && smin.x >= 0; ok[1] = ok[1] && smin.y >= 0;
            }
            // entry depth: 1 / the smallest m . ld over the front planes (every one of them faces a ray inside the cone: m . ld > 0;
            // a ray that grazes the outline may find one that does not -- it passes for a miss)
            V2 umin = {inv_near, inv_near};
            for (int k = 0; k < nfront && (ok[0] || ok[1]); k += 4) {
              F q[4][4];
              RowLoad<F>::load4(fr + 4 * (size_t)k, q);
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const V2 u = q[j][0] * ld[0] + q[j][1] * ld[1] + q[j][2] * ld[2];
                if (COLOR) {
#pragma unroll
                  for (int c = 0; c < 2; ++c)
                    if (u[c] < umin[c]) face[c] = k + j < nfront ? k + j : nfront - 1;
                }
                umin = min2(umin, u);
              }
            }
            t0 = V2{Num::rcp(umin.x), Num::rcp(umin.y)};  // (no plane nearer than the near plane: t0 = znear, and the test below says no)
            ok[0] = ok[0] && umin.x > 0 && t0.x <= t1.x; ok[1] = ok[1] && umin.y > 0 && t0.y <= t1.y;
          }
        }
        if ((ok[0] || ok[1]) && shape == kShapeHull && !by_outline) {
          // four planes per round: their loads go out together (a plane a round would wait for L1 every time); the tail
          // round repeats the last plane, which changes nothing
          // (the planes are read through the constant address space: the address is the wavefront's -- g is -- so they arrive by
          // scalar loads, 32 bytes per plane per WAVEFRONT instead of per lane, and feed the multiply-adds from scalar registers)
          kdouble* pl = (kdouble*)(sc.planes + 4 * (size_t)sh.plane_adr);
          for (int k = 0; k < plane_num && (ok[0] || ok[1]); k += 4) {
            F q4[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              kdouble* src = pl + 4 * (size_t)(k + j < plane_num ? k + j : plane_num - 1);
              q4[j][0] = (F)src[0]; q4[j][1] = (F)src[1]; q4[j][2] = (F)src[2]; q4[j][3] = (F)src[3];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const V2 nd = q4[j][0] * ld[0] + q4[j][1] * ld[1] + q4[j][2] * ld[2];
              const F no = q4[j][3] - (q4[j][0] * lo[0] + q4[j][1] * lo[1] + q4[j][2] * lo[2]);  // >= 0: origin inside this half space
#pragma unroll
              for (int c = 0; c < 2; ++c) {
                if (nd[c] == 0) { ok[c] = ok[c] && no >= 0; continue; }
                // Does this plane move the interval at all?  Entering planes (nd < 0) matter when t = no / nd > t0, leaving ones when
                // t < t1 -- either way  no < bound * nd  -- and after the first few planes of a walk almost none does: the division
                // is spent only on the planes that pass
                const bool front = nd[c] < 0;
                const F bound = front ? t0[c] : t1[c];
                if (no < bound * nd[c]) {
                  const F t = no * Num::rcp(nd[c]);
                  if (COLOR && front && t > t0[c]) face[c] = k + j < plane_num ? k + j : plane_num - 1;
                  if (front) t0[c] = t > t0[c] ? t : t0[c]; else t1[c] = t < t1[c] ? t : t1[c];
                }
              }
            }
            ok[0] = ok[0] && t0.x <= t1.x; ok[1] = ok[1] && t0.y <= t1.y;
          }
        }
        // a camera inside a shape sees its inside faces culled (back faces): only entry points count
#pragma unroll
        for (int c = 0; c < 2; ++c)
          if (ok[c] && t0[c] > znear && t0[c] < best[c]) { best[c] = t0[c]; hit[c] = true; if (COLOR) { hit_g[c] = g; hit_face[c] = face[c]; hit_outline[c] = by_outline; } }
      }
      const size_t img = (size_t)e * W * H;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if (!in_image[c]) continue;
        const int colc = col + c;
        const F xc = x[c], bestc = best[c];
        // (1/near - 1/z) / (1/near - 1/far): a difference of reciprocals of very different sizes -- the encoding (a few instructions
        // a ray) stays in double whatever the rays' type
        const float dgl = hit[c] ? (float)((sc.inv_near - fast_rcp((double)bestc)) * sc.inv_span) : 1.0f;
        if (depth_gl) depth_gl[img + (size_t)row * W + colc] = dgl;
        if (depth_mm) {
          // python/rcs/camera/sim.py:74-86 in float32: z = near / (1 - d (1 - near / far)); uint16(z * 1000)
#pragma clang fp contract(off)  // numpy rounds the product before the subtraction: no fused multiply-add here
          const float nearf = (float)sc.znear;
          const float k1 = (float)(1.0 - sc.znear / sc.zfar);
          const float prod = dgl * k1;
          const float z = nearf / (1.0f - prod);
          const float mm = z * 1000.0f;
          depth_mm[img + (size_t)(H - 1 - row) * W + colc] = (uint16_t)mm;
        }
        if constexpr (COLOR) {
          if (!rgb) continue;
          const RenderShade& L = sc.shade;
          const F inv_len = Num::rsqrt(dd[c]);
          F out[3];
          if (!hit[c]) {
            const F dz = cR[6] * xc + cR[7] * y - cR[8];  // the ray's world z
            const F f = (F)0.5 * (dz * inv_len + 1);
#pragma unroll
            for (int k = 0; k < 3; ++k) out[k] = (F)L.sky2[k] + f * ((F)L.sky1[k] - (F)L.sky2[k]);
          } else {
            // shading in the SHAPE's frame: the face normal is given there, the ray's direction and the light's are in the row
            const int hg = hit_g[c], hf = hit_face[c];
            const RenderShape& sh = sc.shapes[hg];
            const RenderColour& col_g = sc.colours[hg];
            const F* w = lw + hg * kShapeFrameDoubles;
            const F ld[3] = {w[kRowM] * xc + w[kRowM + 1] * y + w[kRowM + 2], w[kRowM + 3] * xc + w[kRowM + 4] * y + w[kRowM + 5],
                             w[kRowM + 6] * xc + w[kRowM + 7] * y + w[kRowM + 8]};
            const F hp[3] = {w[kRowLo] + bestc * ld[0], w[kRowLo + 1] + bestc * ld[1], w[kRowLo + 2] + bestc * ld[2]};  // the hit point
            F nl[3] = {0, 0, 1};
            if (sh.shape == kShapeBox) {
              const int ax = hf >> 1;
              const F sgn = (hf & 1) ? 1 : -1;
              nl[0] = ax == 0 ? sgn : 0; nl[1] = ax == 1 ? sgn : 0; nl[2] = ax == 2 ? sgn : 0;
            } else if (sh.shape == kShapeCapsule) {
              // the hit point from the nearest point of the axis segment
              const F hl = w[kRowSize + 2] - w[kRowSize];
              nl[0] = hp[0]; nl[1] = hp[1];
              nl[2] = hp[2] - (hp[2] > hl ? hl : (hp[2] < -hl ? -hl : hp[2]));
            } else if (sh.shape == kShapeHull) {
              if (hit_outline[c]) {  // (a front row is n / (d - n . o) with d - n . o < 0: the normal points the other way)
                const F* q = (const F*)(sc.views + (size_t)e * sc.view_stride + sh.view_adr + kViewHeaderDoubles) + 4 * (size_t)hf;
                nl[0] = -q[0]; nl[1] = -q[1]; nl[2] = -q[2];
              } else {
                const double* q = sc.planes + 4 * (size_t)(sh.plane_adr + hf);
                nl[0] = (F)q[0]; nl[1] = (F)q[1]; nl[2] = (F)q[2];
              }
            }
            const F nn = Num::rsqrt(nl[0] * nl[0] + nl[1] * nl[1] + nl[2] * nl[2]);
            const F ndv = -(nl[0] * ld[0] + nl[1] * ld[1] + nl[2] * ld[2]) * nn * inv_len;                                  // towards the camera
            const F ndl = -(nl[0] * w[kRowLight] + nl[1] * w[kRowLight + 1] + nl[2] * w[kRowLight + 2]) * nn;  // towards the light
            const F kv = ndv > 0 ? ndv : 0, kl = ndl > 0 ? ndl : 0;
            bool second = false;
            if (col_g.checker != 0.0) {
              const long long ix = (long long)floor(hp[0] / (F)col_g.square), iy = (long long)floor(hp[1] / (F)col_g.square);
              second = ((ix + iy) & 1) != 0;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              const F base = (F)(second ? col_g.rgb2[k] : col_g.rgb[k]);
              out[k] = base * ((F)L.ambient[k] + (F)L.head_diffuse[k] * kv + (F)L.light_diffuse[k] * kl);
            }
          }
          uint8_t* px = rgb + 3 * (img + (size_t)row * W + colc);
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const F v = out[k] < 0 ? 0 : (out[k] > 1 ? 1 : out[k]);
            px[k] = (uint8_t)(v * 255 + (F)0.5);
          }
        }
      }
    }
  }
}

#endif  // __HIP__

}  // namespace rcsh
