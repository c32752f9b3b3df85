"""``SimCameraSet`` for N environments: depth images by ray casting (reference src/sim/camera.cpp,
python/rcs/camera/sim.py, python/rcs/camera/interface.py).

Kept from the reference: the configuration type, the frame-set buffer with its timestamp rule (frames rendered at the
same simulation time share one frame set; ``clear_buffer`` forgets the last timestamp), render-on-demand,
the conversion of the depth buffer (row flip, metres with ``physical_units``, ``DEPTH_SCALE``, uint16), the colour
frame's layout ([H, W, 3] uint8, rows flipped like the depth), intrinsics and extrinsics, the four camera types
(``fixed``, ``default_free``, ``free`` -- an untouched mjvCamera: it looks at the origin from 2 m -- and ``tracking``, which
fails as it does in the reference: SimCameraSet never sets a body to track).  Different: the pixels are ray-cast against
analytic shapes (``rcs_amd/render.py``) -- flat-shaded colours of the collision shapes, no textures but the floor's
checker, no shadows: not OpenGL's pixels -- and the buffer keeps the last ``max_framesets`` frame sets instead of growing
until the next reset (one frame set of 4096 environments at 256 x 256 is 0.5 GB).
"""

from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from enum import IntEnum
from typing import Any

import numpy as np

from . import _lib, common, render
from . import sim as _sim


class CameraType(IntEnum):  # src/sim/camera.h:19-24
    free = 0
    tracking = 1
    fixed = 2
    default_free = 3


@dataclass(kw_only=True)
class SimCameraConfig:  # src/sim/camera.h:26-34, common::BaseCameraConfig
    identifier: str
    frame_rate: int = 0
    resolution_width: int = 256
    resolution_height: int = 256
    type: CameraType = CameraType.fixed


@dataclass(kw_only=True)
class DataFrame:  # python/rcs/camera/interface.py:14-20
    data: Any
    timestamp: Any = None
    intrinsics: np.ndarray | None = None
    extrinsics: np.ndarray | None = None


@dataclass(kw_only=True)
class CameraFrame:
    color: DataFrame
    ir: DataFrame | None = None
    depth: DataFrame | None = None
    temperature: float | None = None


@dataclass(kw_only=True)
class Frame:
    camera: CameraFrame
    imu: Any = None
    avg_timestamp: Any = None


@dataclass(kw_only=True)
class FrameSet:
    frames: dict[str, Frame]
    avg_timestamp: Any


class SimCameraSet:
    DEPTH_SCALE: int = 1000  # BaseCameraSet.DEPTH_SCALE

    def __init__(self, simulation: _sim.Sim, cameras: dict[str, SimCameraConfig], physical_units: bool = False,
                 render_on_demand: bool = True, max_framesets: int = 2):
        self._sim = simulation
        self.cameras = cameras
        self.physical_units = physical_units
        self.render_on_demand = render_on_demand
        self.max_framesets = max_framesets
        self._L = simulation._L
        cm = simulation.model
        self._scene = render.build_render_scene(cm, os.path.dirname(simulation.scene_path))
        rs = self._scene
        d = _lib.RenderSceneDesc()
        d.nshape, d.nplanes, d.znear, d.zfar = len(rs.shape), len(rs.planes), rs.znear, rs.zfar
        self._keep = []
        for name in ("shape", "link", "pos", "rot", "size", "plane_adr", "plane_num", "sphere", "planes"):
            a = np.ascontiguousarray(getattr(rs, name)).reshape(-1)
            self._keep.append(a)
            setattr(d, name, a.ctypes.data_as(_lib._I32P if a.dtype == np.int32 else _lib._F64P))
        _lib.check(self._L.rcsh_sim_set_render_scene(simulation._h, C.byref(d)))
        rc = _lib.RenderColours()
        col = np.ascontiguousarray(rs.colour, dtype=np.float64).reshape(-1)
        self._keep.append(col)
        rc.colour = col.ctypes.data_as(_lib._F64P)
        for name in ("headlight_ambient", "headlight_diffuse", "light_dir", "light_diffuse", "sky_rgb1", "sky_rgb2"):
            getattr(rc, name)[:] = [float(x) for x in getattr(rs, name)]
        _lib.check(self._L.rcsh_sim_set_render_colours(simulation._h, C.byref(rc)))
        self._ids: dict[str, int] = {}
        self._fovy: dict[str, float] = {}
        for name, cfg in cameras.items():
            if cfg.type == CameraType.default_free:  # camera.cpp:41-42: mjv_defaultFreeCamera
                link, pos, rot, fovy = render.default_free_camera(cm)
            elif cfg.type == CameraType.fixed:
                link, pos, rot, fovy = render.camera_in_link(cm, cfg.identifier)
            elif cfg.type == CameraType.free:  # camera.cpp:36-47: a default mjvCamera with type = mjCAMERA_FREE
                link, pos, rot, fovy = render.free_camera(cm)
            else:
                # camera.cpp:44-46 sets type and fixedcamid only; trackbodyid stays -1 and mjv_updateCamera refuses it
                raise RuntimeError("track body id is outside valid range")
            c = _lib.CameraDesc()
            c.link, c.width, c.height, c.fovy_deg = link, cfg.resolution_width, cfg.resolution_height, fovy
            c.pos[:] = [float(x) for x in pos]
            c.rot[:] = [float(x) for x in rot]
            cid = C.c_int32(-1)
            _lib.check(self._L.rcsh_sim_add_camera(simulation._h, C.byref(c), C.byref(cid)))
            self._ids[name] = cid.value
            self._fovy[name] = fovy
        self._buffer: list[dict] = []
        self._last_ts = None
        # Rendering callbacks (camera.cpp:23-47 -> Sim::register_rendering_callback): cameras with a frame rate are rendered
        # from inside the stepping at that rate when render_on_demand is off.  The kernels record what the renderer would
        # have seen at the moments a frame became due; `collect()` (called by Sim / the environments after every launch)
        # renders the records.  Every environment has its own clock, so a frame set of the batch is "the environments whose
        # cameras were due in that record", with a timestamp per environment; `_latest` keeps, per camera and environment,
        # the newest frame.
        self._rated = [name for name, cfg in cameras.items() if cfg.frame_rate != 0]
        self._latest: dict | None = None
        if not render_on_demand and self._rated:
            if len(self._rated) > 4:
                raise ValueError("at most 4 cameras with a frame rate")
            if getattr(simulation, "_rate_camera_sets", None):
                # the kernels keep ONE render schedule per Sim; a second rate-driven set would silently replace the first one's
                raise RuntimeError("this Sim already has a SimCameraSet with render_on_demand=False: put all rate-driven cameras into one set")
            self._sched_ids = np.array([self._ids[nm] for nm in self._rated], dtype=np.int32)
            self._sched_periods = np.array([1.0 / cameras[nm].frame_rate for nm in self._rated], dtype=np.float64)
            self._timestep = float(cm.timestep)
            self._capacity = 0
            self._dropped = 0
            # records per launch: sized for the longest launch the present SimConfig can produce (step_until_convergence's cap);
            # Sim.step(k) / Sim.set_config call ensure_capacity again before a longer one
            cap = simulation.get_config().max_convergence_steps
            self.ensure_capacity(cap if cap > 0 else 2000)
            simulation._rate_camera_sets = [self]

    def ensure_capacity(self, substeps: int) -> None:
        """Make the render schedule hold every record a launch of `substeps` substeps can produce (records of different cameras
        need not coincide).  Growing keeps the cameras' clocks and pending records (rcsh_sim_set_render_schedule with the same
        cameras); the device buffer is capped at 256 records per environment and launch -- beyond that the newest records of a
        launch are lost, which `collect` reports with a warning."""
        if self.render_on_demand or not self._rated:
            return
        want = int(min(256, np.ceil(substeps * self._timestep / self._sched_periods).sum() + 2))
        if want > self._capacity:
            _lib.check(self._L.rcsh_sim_set_render_schedule(self._sim._h, _lib.ptr(self._sched_ids), _lib.ptr(self._sched_periods),
                                                            len(self._sched_ids), want))
            self._capacity = want

    # ---- SimCameraSet (src/sim/camera.cpp:54-83)
    def buffer_size(self) -> int:
        return len(self._buffer)

    def clear_buffer(self) -> None:
        self._last_ts = None  # "when we clear the buffer, there is no last image timestep"
        self._buffer.clear()
        self._latest = None

    def set_double_precision(self, on: bool) -> None:
        """The ray caster's arithmetic: float32 by default (the reference's depth image is a float32 z-buffer); `on` selects the
        double-precision instantiation, whose pixels equal the tests' numpy restatement bit for bit (about 1.4x the time)."""
        _lib.check(self._L.rcsh_sim_set_render_f64(self._sim._h, 1 if on else 0))

    def render_raw(self, name: str):
        """(depth buffer [N,H,W] f32 rows bottom-up as mjr_readPixels returns it, cam_xmat [N,3,3], cam_xpos [N,3])."""
        cfg = self.cameras[name]
        n = self._sim.n_envs
        depth = np.zeros((n, cfg.resolution_height, cfg.resolution_width), dtype=np.float32)
        pose = np.zeros((n, 12))
        _lib.check(self._L.rcsh_camera_render(self._sim._h, self._ids[name], _lib.ptr(depth), None, _lib.ptr(pose)))
        return depth, pose[:, :9].reshape(n, 3, 3), pose[:, 9:]

    def render_raw_rgbd(self, name: str):
        """(rgb [N,H,W,3] uint8 and depth [N,H,W] f32, both rows bottom-up as mjr_readPixels returns them, cam_xmat, cam_xpos)."""
        cfg = self.cameras[name]
        n = self._sim.n_envs
        rgb = np.zeros((n, cfg.resolution_height, cfg.resolution_width, 3), dtype=np.uint8)
        depth = np.zeros((n, cfg.resolution_height, cfg.resolution_width), dtype=np.float32)
        pose = np.zeros((n, 12))
        _lib.check(self._L.rcsh_camera_render_rgb(self._sim._h, self._ids[name], _lib.ptr(rgb), _lib.ptr(depth), None, _lib.ptr(pose)))
        return rgb, depth, pose[:, :9].reshape(n, 3, 3), pose[:, 9:]

    def render_rgb_dev(self, name: str, out_ptr: int) -> None:
        """The colour frame alone into device memory ([N,H,W,3] uint8, rows bottom-up)."""
        _lib.check(self._L.rcsh_camera_render_rgb_dev(self._sim._h, self._ids[name], C.c_void_p(out_ptr), None, None, None))

    def render_depth_mm(self, name: str) -> np.ndarray:
        """The fused device path: [N,H,W] uint16 millimetres, rows top-down (physical units)."""
        cfg = self.cameras[name]
        out = np.zeros((self._sim.n_envs, cfg.resolution_height, cfg.resolution_width), dtype=np.uint16)
        _lib.check(self._L.rcsh_camera_render(self._sim._h, self._ids[name], None, _lib.ptr(out), None))
        return out

    def render_depth_mm_dev(self, name: str, out_ptr: int) -> None:
        _lib.check(self._L.rcsh_camera_render_dev(self._sim._h, self._ids[name], None, C.c_void_p(out_ptr), None))

    def _render_all(self) -> None:  # render_all + render_single (camera.cpp:86-140)
        ts = self._sim.time
        same = self._last_ts is not None and np.array_equal(ts, self._last_ts)
        if not same:
            self._buffer.append({"timestamp": ts, "depth": {}, "color": {}, "pose": {}})
            del self._buffer[: max(0, len(self._buffer) - self.max_framesets)]
            self._last_ts = ts
        fs = self._buffer[-1]
        for name in self.cameras:
            rgb, depth, xmat, xpos = self.render_raw_rgbd(name)
            fs["depth"][name] = depth
            fs["color"][name] = rgb
            fs["pose"][name] = (xmat, xpos)

    def collect(self) -> None:
        """Render the records of the launch that just ran (render_single for every due camera, camera.cpp:103-140)."""
        if self.render_on_demand or not self._rated:
            return
        n = self._sim.n_envs
        count = np.zeros(n, dtype=np.int32)
        _lib.check(self._L.rcsh_render_pending(self._sim._h, _lib.ptr(count)))
        dropped = C.c_int64(0)
        _lib.check(self._L.rcsh_render_dropped(self._sim._h, C.byref(dropped)))
        if dropped.value > self._dropped:
            import warnings

            warnings.warn(f"SimCameraSet: {dropped.value - self._dropped} frames became due in one launch beyond the render schedule's "
                          f"capacity ({self._capacity} records per environment) and were not rendered", RuntimeWarning, stacklevel=2)
            self._dropped = dropped.value
        for slot in range(int(count.max(initial=0))):
            event = {"timestamp": np.full(n, np.nan), "depth": {}, "color": {}, "pose": {}, "have": {}}
            for name in self._rated:
                cfg = self.cameras[name]
                rgb = np.zeros((n, cfg.resolution_height, cfg.resolution_width, 3), dtype=np.uint8)
                depth = np.zeros((n, cfg.resolution_height, cfg.resolution_width), dtype=np.float32)
                pose = np.zeros((n, 12))
                ts = np.zeros(n)
                due = np.zeros(n, dtype=np.uint8)
                _lib.check(self._L.rcsh_camera_render_snapshot(self._sim._h, self._ids[name], slot, _lib.ptr(rgb), _lib.ptr(depth), None, _lib.ptr(pose),
                                                               _lib.ptr(ts), _lib.ptr(due)))
                have = due.astype(bool)
                if not have.any():
                    continue
                event["timestamp"][have] = ts[have]
                event["depth"][name], event["color"][name] = depth, rgb
                event["pose"][name] = (pose[:, :9].reshape(n, 3, 3), pose[:, 9:])
                event["have"][name] = have
            if not event["have"]:
                continue
            self._buffer.append(event)
            del self._buffer[: max(0, len(self._buffer) - self.max_framesets)]
            self._merge_latest(event)

    def _merge_latest(self, event: dict) -> None:
        n = self._sim.n_envs
        if self._latest is None:
            self._latest = {"timestamp": np.full(n, np.nan), "depth": {}, "color": {}, "pose": {}, "cam_timestamp": {}}
        lt = self._latest
        for name, have in event["have"].items():
            if name not in lt["depth"]:
                lt["depth"][name] = np.ones_like(event["depth"][name])
                lt["color"][name] = np.zeros_like(event["color"][name])
                lt["pose"][name] = (np.tile(np.eye(3), (n, 1, 1)), np.zeros((n, 3)))
                lt["cam_timestamp"][name] = np.full(n, np.nan)
            lt["depth"][name][have] = event["depth"][name][have]
            lt["color"][name][have] = event["color"][name][have]
            lt["pose"][name][0][have] = event["pose"][name][0][have]
            lt["pose"][name][1][have] = event["pose"][name][1][have]
            lt["cam_timestamp"][name][have] = event["timestamp"][have]
            lt["timestamp"][have] = event["timestamp"][have]

    def get_latest_frames(self) -> FrameSet | None:
        if self.render_on_demand:
            self._render_all()
            if not self._buffer:
                return None
            return self._to_frames(self._buffer[-1])
        # rate-driven: the newest frame of every camera and environment (an environment's entry of `avg_timestamp` is NaN
        # until its first frame; DataFrame.timestamp is the camera's own)
        if self._latest is None:
            return None
        return self._to_frames(self._latest)

    def get_timestamp_frames(self, ts) -> FrameSet | None:
        for fs in reversed(self._buffer):
            if np.array_equal(fs["timestamp"], ts):
                return self._to_frames(fs)
        return None

    # ---- python/rcs/camera/sim.py:45-115
    def _to_frames(self, fs: dict) -> FrameSet:
        frames: dict[str, Frame] = {}
        for name, raw in fs["depth"].items():
            depth = raw[:, ::-1, :, None]  # glReadPixels rows are bottom-up
            if self.physical_units:
                near, far = self._scene.znear, self._scene.zfar
                depth = near / (1 - depth * (1 - near / far))
            xmat, xpos = fs["pose"][name]
            cam_ts = fs.get("cam_timestamp", {}).get(name, fs["timestamp"])
            frames[name] = Frame(
                camera=CameraFrame(
                    color=DataFrame(data=fs["color"][name][:, ::-1], timestamp=cam_ts, intrinsics=self._intrinsics(name),
                                    extrinsics=self._extrinsics(xmat, xpos)),
                    depth=DataFrame(data=(depth * self.DEPTH_SCALE).astype(np.uint16), timestamp=cam_ts,
                                    intrinsics=self._intrinsics(name), extrinsics=self._extrinsics(xmat, xpos)),
                ),
                avg_timestamp=fs["timestamp"],
            )
        return FrameSet(frames=frames, avg_timestamp=fs["timestamp"])

    def _intrinsics(self, camera_name: str) -> np.ndarray:
        cfg = self.cameras[camera_name]
        fx = fy = 0.5 * cfg.resolution_height / np.tan(self._fovy[camera_name] * np.pi / 360)
        return np.array([[fx, 0, (cfg.resolution_width - 1) / 2, 0], [0, fy, (cfg.resolution_height - 1) / 2, 0], [0, 0, 1, 0]])

    @staticmethod
    def _extrinsics(xmat: np.ndarray, xpos: np.ndarray) -> np.ndarray:
        """[N,4,4]: inverse of (camera pose * a half turn about x that puts the z axis in front)."""
        flip = common.Pose(rpy_vector=np.array([np.pi, 0, 0]), translation=np.zeros(3))
        out = np.zeros((len(xpos), 4, 4))
        for e in range(len(xpos)):
            cam = common.Pose(rotation=xmat[e], translation=xpos[e]) * flip
            out[e] = cam.inverse().pose_matrix()
        return out

    def calibrate(self) -> bool:
        return True

    def config(self, camera_name: str) -> SimCameraConfig:
        return self.cameras[camera_name]

    def close(self) -> None:
        pass

    @property
    def camera_names(self) -> list[str]:
        return list(self.cameras.keys())

    @property
    def name_to_identifier(self) -> dict[str, str]:
        return {name: cfg.identifier for name, cfg in self.cameras.items()}
