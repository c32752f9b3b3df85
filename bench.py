#!/usr/bin/env python
"""Headline benchmark: env-steps/s, fr3_empty_world, JOINTS mode, 4096 environments per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one Gymnasium env.step() over the whole batch (BASELINE.json configs[1]: 4096 x fr3_empty_world,
ControlMode.JOINTS, relative +-5 deg actions, gripper commanded every step, no contacts, IK off), in the fixed-work
mode the reference uses for rollouts (SimConfig async_control=True, 30 Hz => 17 physics substeps per env-step;
reference python/rcs/envs/sim.py:52-53).  Inputs (the pre-generated action tensor) are resident in HBM before the
timed region; each step is one fused kernel launch through the C-ABI (rcsh_env_step_dev).  With N > 1 every rank
owns 4096 environments on its own GPU (weak scaling) and the observation tensor is all-gathered over RCCL once
per step, inside the timed region.  No PyTorch on the data path: device buffers are rcsh_dev_alloc'ed, actions drawn with
numpy, streams ordered by the library's own handles; with N > 1 the ranks (launched by torch.distributed.run, or by this script
itself: `python bench.py --gpus N`) meet over a Unix-domain socket -- RCCL id, barriers, max-over-ranks of the clock -- and the
exchange is RCCL behind the C-ABI: no rank imports torch.

Prints ONE JSON line (rank 0).  Extra objects: `roofline` (algorithmic HBM bytes of the fused launch / measured
kernel time, HIP events on the launch stream) and `cpu_baseline` (the CPU oracle timed on this host's cores on a
bounded sample of the same workload; N = 1 only).
"""

from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, "robot-control-stack_amd")]

N_ENVS = 4096
SUBSTEPS = 17
# action 60 B + physics state r/w 576 B + RCS per-env state r/w 480 B + obs 168 B + flags 8 B
ALGO_BYTES_PER_ENV_STEP = 1292
ALGO_FLOP_PER_SUBSTEP = 1.0e4  # SURVEY 8(d) estimate of MuJoCo's pipeline for this scene
HBM_PEAK_GBS = 8000.0
FP64_VECTOR_PEAK_TFLOPS = 78.6


def cpu_baseline_worker(n_envs: int, n_steps: int, seed0: int) -> None:
    """Child process: steps `n_envs` oracle environments; prints the elapsed seconds of two legs."""
    # (the checker lives under oracle/ and tests/; only this CPU-baseline leg of the benchmark touches either)
    sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    import numpy as np
    from parity_util import make_oracle_envs, synthetic_actions

    envs = make_oracle_envs(n_envs, async_control=True)
    joints, grip = synthetic_actions(n_envs, n_steps, seed0)
    for e in envs:
        e.reset()
    # leg 1, the compiled path alone: SimRobot::set_joint_position + SimGripper::set_normalized_width + Sim::step(17) per
    # env-step, in C (what the reference spends in its C++ library per RobotEnv.step; no Python wrapper logic)
    q0 = [np.array(e.sim.get_joint_position()) for e in envs]
    t0 = time.perf_counter()
    for t in range(n_steps):
        for i, e in enumerate(envs):
            e.sim.set_joint_position(q0[i] + joints[t, i])
            e.sim.gripper_set_normalized_width(float(grip[t, i]))
            e.sim.step(SUBSTEPS)
    dt_c = time.perf_counter() - t0
    # leg 2, through the restated Gymnasium wrapper stack (Python), a tenth of the steps
    for e in envs:
        e.reset()
    n_py = max(n_steps // 10, 1)
    t0 = time.perf_counter()
    for t in range(n_py):
        for i, e in enumerate(envs):
            e.step({"joints": joints[t, i], "gripper": grip[t, i]})
    dt_py = time.perf_counter() - t0
    print(json.dumps({"seconds_c": dt_c, "env_steps_c": n_envs * n_steps, "seconds_py": dt_py, "env_steps_py": n_envs * n_py}))


def usable_cores() -> int:
    """Cores this process may actually use: affinity mask, capped by the cgroup CPU quota (a container can see 256
    CPUs and be allowed 16; oversubscribing the quota makes every worker crawl)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())  # cgroup v1
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return n


def run_cpu_baseline() -> dict:
    """Oracle env-steps/s on all usable host cores: one process per core, each a bounded sample of the workload."""
    cores = usable_cores()
    per_proc_envs, steps = 64, 2000  # 128k env-steps of C per process (~10 s at ~70 us each) + a tenth through Python
    t0 = time.perf_counter()
    procs = [
        subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(per_proc_envs), str(steps), str(1000 * p)],
                         stdout=subprocess.PIPE, text=True)
        for p in range(cores)
    ]
    outs = [p.communicate()[0] for p in procs]
    wall = time.perf_counter() - t0
    recs = [json.loads(o.strip().splitlines()[-1]) for o in outs if o.strip()]
    slow_c = max(r["seconds_c"] for r in recs)
    slow_py = max(r["seconds_py"] for r in recs)
    rate_c = sum(r["env_steps_c"] for r in recs) / slow_c
    return {
        "value": rate_c,
        "unit": "env-steps/s",
        "cores": cores,
        "cores_visible": os.cpu_count(),
        "kind": "port",
        "sample": f"{cores} processes x {per_proc_envs} envs x {steps} env-steps (set_joint_position + gripper command + "
                  f"Sim.step(17) per env-step, C restatement called once per call from Python), all processes concurrent, "
                  f"slowest {slow_c:.2f}s; launch-to-finish incl. the wrapper leg {wall:.1f}s",
        "physics_substeps_per_s_per_core": rate_c * SUBSTEPS / cores,
        "through_python_wrapper_stack": sum(r["env_steps_py"] for r in recs) / slow_py,
        "note": "CPU restatement (oracle/), not MuJoCo: MuJoCo is not installable here (SURVEY F2). `value` is the compiled "
                "path alone; `through_python_wrapper_stack` adds the restated Gymnasium wrappers (Python), as the reference's env does",
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--envs", type=int, default=N_ENVS, help="environments per GPU (headline: 4096)")
    ap.add_argument("--mode", choices=["async", "convergence"], default="async")
    ap.add_argument("--control", choices=["joints", "cartesian"], default="joints",
                    help="cartesian = BASELINE configs[2]: relative TRPY actions -> CLIK -> joint targets (not the headline)")
    ap.add_argument("--episode-length", type=int, default=None,
                    help="env.reset() every this many steps (inside the timed region; resets are not counted as env-steps). "
                         "Default: none for joints; 10 for cartesian, as the reference's examples loop (reset + 10 steps) -- a longer "
                         "random walk of Cartesian targets leaves the workspace and the CLIK then runs to its 1000-iteration cap")
    ap.add_argument("--robot", choices=["fr3", "xarm7", "xarm7_box", "xarm7_pick", "arm6", "ur5e", "so101", "mixed"], default="fr3",
                    help="xarm7: 7-dof arm with dry joint friction, no gripper; xarm7_box: the same next to a free cube with floor contacts "
                         "(builder-authored scene xarm7_box_world, camera side_cam); xarm7_pick: BASELINE configs[3] as written -- the xArm7 with a two-finger "
                         "gripper next to the cube, gripper / cube / floor contacts resolved together with the dry-friction rows (builder-authored "
                         "scene xarm7_pick_world, camera side_cam); arm6 / ur5e: builder-authored 6-dof arms (Topo<6,false>); "
                         "so101: builder-authored 5-dof arm + two-finger gripper (Topo<5,true>); mixed: FR3 / xArm7 / UR5e / SO101 sharded by robot "
                         "type -- rank r runs type r mod 4 with one specialised kernel per GPU, as BASELINE configs[4] asks; with fewer than 4 "
                         "ranks a rank hosts several types as sub-batches on streams of their own (throughput only, not the headline)")
    ap.add_argument("--task", choices=["none", "pick_up"], default="none",
                    help="pick_up = the registered gym task rcs/FR3SimplePickUpSim-v0 (fr3_simple_pick_up scene: free cube on the floor with "
                         "elliptic-cone contacts + noslip, RandomCubePos on reset, PickCubeSuccessWrapper reward; relative TRPY control, 30 Hz); "
                         "not the headline")
    ap.add_argument("--cameras", default="", help="comma-separated MJCF camera names: one depth frame (uint16 mm, ray-cast) per camera per "
                                                 "env-step, inside the timed region (SimCameraSet, render on demand); not the headline")
    ap.add_argument("--resolution", default="256x256", help="WxH of the depth frames (FR3SimplePickUpSimEnvCreator default 256x256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the figures taken after the clock stops (host API, round-4 configuration, BASELINE.md's 1000-step rollout, step_until_convergence): profiling runs")
    ap.add_argument("--clock-warmup-ms", type=float, default=25.0,
                    help="before the W warmup steps: this many milliseconds of the same env-step launches, then a reset -- the device needs "
                         "~12 ms of work to reach its operating clocks (profiles/README.md: a launch takes 128 us at first, 112.5 us from "
                         "there on), W = 5 steps are 0.6 ms.  0: off")
    ap.add_argument("--contact-check-every", type=int, default=1,
                    help="cadence of the lean launches' end-of-launch collision check (csrc/check_team.h): every this many env-steps.  1 (default, "
                         "the library's default and what the parity tests run): every env-step.  Where robot contacts are resolved environment by "
                         "environment (--contacts resolve, the default) the check IS what sends an environment to the contact-resolving kernel and "
                         "the library runs it in every launch whatever this says; a larger cadence only applies to --contacts flag")
    ap.add_argument("--contacts", default="resolve", choices=("resolve", "flag"),
                    help="resolve (default, round 5): every contact of the robot's geoms -- floor and robot <-> robot -- is resolved, environment by "
                         "environment (Sim(resolve_robot_contacts=None)); flag: the round-4 configuration -- lean kernels only, an environment in "
                         "contact gets the sticky info.contact_unresolved and steps on unresolved")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL behind the C-ABI; the measured configuration) or host (the observation blocks through the rendezvous socket and host memory: "
                                                               "exercises the N > 1 code path on a box with fewer GPUs than ranks)")
    ap.add_argument("--cpu-baseline-worker", nargs=3, metavar=("ENVS", "STEPS", "SEED"))
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        cpu_baseline_worker(*(int(x) for x in args.cpu_baseline_worker))
        return

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher of N ranks, one per GPU (what the driver's
        # `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` does)
        # (no launcher framework either: N child processes with the environment torch.distributed.run would give them)
        import socket

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        procs = []
        for r in range(args.gpus):
            env_r = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env_r))
        raise SystemExit(max(p.wait() for p in procs))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or run `python bench.py --gpus N`, which spawns them)")

    cpu_base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_base = run_cpu_baseline()  # before any GPU context exists in this process

    import ctypes as C

    import numpy as np

    from rcs_amd import _lib
    from rcs_amd.envs import MAX_JOINT_MOV, make_vec_env

    # No PyTorch on this path (BASELINE north star): device memory comes from the C-ABI (rcsh_dev_alloc), the synthetic actions
    # from numpy, streams and their ordering from the library's own handles; with N > 1 ranks the launcher-side rendezvous is a
    # Unix-domain socket (rcs_amd.envs.sharding.SocketRendezvous) and the one exchange of the data path, the all-gather of the
    # observation block, is RCCL behind the C-ABI.
    if os.environ.get("RCSH_LIB"):
        _lib.LIB_PATH = os.environ["RCSH_LIB"]  # (a development build of the library, for A/B measurements)
    L0 = _lib.load()
    n_dev = int(L0.rcsh_device_count())
    if n_dev < 1:
        raise SystemExit("bench.py needs a GPU: the batched backend has no CPU execution path")
    # One rank per GPU: LOCAL_RANK names the device -- unless the launcher masked the devices so that every rank sees only its own
    # (then that is device 0), or there are fewer GPUs than ranks (a functional check: ranks share a GPU, RCCL refuses two ranks on
    # one device and the exchange falls back to the rendezvous group through host memory, see below).
    local_rank %= n_dev
    from rcs_amd.envs.sharding import SocketRendezvous

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):
            # one node: RCCL's bootstrap (ncclGetUniqueId / ncclCommInitRank) over the loopback interface -- the container's
            # hostname may not resolve, and the interface RCCL would pick by default may not be routable between ranks
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        # the one message is 688 KB per rank (4096 x 21 f64): one or two channels -- each channel is a 256-thread workgroup of
        # ~270 registers per lane that must find a SIMD next to (or in turn with) the stepping wavefronts; RCCL's default for a
        # node is many more, sized for bandwidth on megabyte messages
        os.environ.setdefault("NCCL_MIN_NCHANNELS", "1")
        os.environ.setdefault("NCCL_MAX_NCHANNELS", "2")
    # the launcher's rendezvous: a Unix-domain socket (RCCL id, barriers, max over ranks of the clock); no torch.distributed
    rdv = SocketRendezvous(rank, world)

    n = args.envs
    T = args.steps + args.warmup
    mixed = args.robot == "mixed"
    MIXED_TYPES = ("fr3", "xarm7", "ur5e", "so101")
    hosted = [args.robot]
    if mixed:
        if args.task != "none" or args.control != "joints" or args.cameras:
            raise SystemExit("bench.py: --robot mixed is a JOINTS-mode throughput configuration (no task, no cameras)")
        hosted = [MIXED_TYPES[rank % 4]] if world >= 4 else [MIXED_TYPES[i] for i in range(4) if i % world == rank]
        if n % len(hosted):
            raise SystemExit(f"bench.py: --envs {n} does not split over the {len(hosted)} robot types this rank hosts")
        args.robot = hosted[0]
    if args.task == "pick_up":
        from rcs_amd.envs import FR3SimplePickUpSimEnvCreator

        args.control, args.mode = "cartesian", "async"
        env = FR3SimplePickUpSimEnvCreator()(n_envs=n, device=local_rank)
    elif args.control == "cartesian":
        from rcs_amd.envs import ControlMode

        env = make_vec_env(n, async_control=(args.mode == "async"), gripper=True, relative=True, device=local_rank,
                           control_mode=ControlMode.CARTESIAN_TRPY, max_relative_movement=(0.2, float(np.deg2rad(45))), robot=args.robot)
    else:
        env = make_vec_env(n // len(hosted), async_control=(args.mode == "async"), gripper=True, relative=True, device=local_rank, robot=args.robot,
                           resolve_robot_contacts=None if args.contacts == "resolve" else False)
    # a rank that hosts several robot types (mixed, fewer than 4 ranks): one sub-batch per type, each on its handle's own
    # stream, so that the sub-batches' launches (each too small to fill the chip) run side by side
    envs = [env] + [make_vec_env(n // len(hosted), async_control=(args.mode == "async"), gripper=True, relative=True, device=local_rank, robot=r,
                                 resolve_robot_contacts=None if args.contacts == "resolve" else False)
                    for r in hosted[1:]]
    L, h = env._L, env.sim._h
    for e_ in envs:
        e_.sim.set_contact_check(args.contact_check_every)

    class DevBuf:
        """[rows, ...] array in HBM (rcsh_dev_alloc), optionally filled from a host array; `at(t)` = address of row t."""

        def __init__(self, shape, dtype, host=None):
            self.shape, self.dtype = tuple(shape), np.dtype(dtype)
            self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
            self.row_bytes = self.nbytes // self.shape[0]
            p = C.c_void_p()
            _lib.check(L.rcsh_dev_alloc(h, max(self.nbytes, 8), C.byref(p)))
            self.ptr = p.value
            a = np.zeros(self.shape, dtype=self.dtype) if host is None else np.ascontiguousarray(host, dtype=self.dtype)
            assert a.nbytes == self.nbytes
            _lib.check(L.rcsh_dev_upload(h, C.c_void_p(self.ptr), a.ctypes.data_as(C.c_void_p), self.nbytes))

        def at(self, t: int, offset_rows_bytes: int = 0) -> int:
            return self.ptr + t * self.row_bytes + offset_rows_bytes

        def download(self) -> np.ndarray:
            out = np.zeros(self.shape, dtype=self.dtype)
            _lib.check(L.rcsh_dev_download(h, out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), self.nbytes))
            return out

    def device_sync() -> None:
        for e_ in envs:
            e_.sim.synchronize()

    # synthetic actions, resident in HBM (joints ~ U(+-5 deg)^7 f64, gripper ~ U(0,1) f32), drawn on the host
    rng = np.random.default_rng(1234 + rank)
    if args.control == "cartesian":  # xyz ~ U(+-0.05 m)^3, rpy ~ U(+-0.1 rad)^3
        scale = np.array([0.05, 0.05, 0.05, 0.1, 0.1, 0.1])
        joints = DevBuf((T, n, 6), np.float64, (rng.random((T, n, 6)) * 2 - 1) * scale)
    else:
        joints = DevBuf((T, env.n_envs, env.dof), np.float64, (rng.random((T, env.n_envs, env.dof)) * 2 - 1) * MAX_JOINT_MOV)
    grip = DevBuf((T, n), np.float32, rng.random((T, n), dtype=np.float32))
    # (sub-batches of other robot types: their own action tensors; info / width / substep outputs are slices of the rank's)
    sub_joints = [joints] + [DevBuf((T, e_.n_envs, e_.dof), np.float64, (rng.random((T, e_.n_envs, e_.dof)) * 2 - 1) * MAX_JOINT_MOV) for e_ in envs[1:]]
    ow = max(e_.obs_width for e_ in envs) if not mixed else 21  # mixed: every rank's block is n x 21 doubles, narrower rows packed at its start
    obs = DevBuf((n, ow), np.float64)
    info = DevBuf((n, 8), np.uint8)
    gw = DevBuf((n,), np.float64)
    sub = DevBuf((n,), np.int32)

    class HostStagedExchange:
        """The exchange protocol of RcclObservationExchange carried by the launcher's rendezvous socket instead (through host
        memory, not overlapped): `--dist-backend host` -- ranks sharing a GPU, where RCCL refuses to form a communicator -- and
        the fallback when the C-ABI communicator cannot be created on some rank.  Functional, not the measured configuration."""

        def __init__(self):
            self._local = [DevBuf((n, ow), np.float64) for _ in range(2)]
            self._all = [np.zeros((world * n, ow)) for _ in range(2)]

        def local_ptr(self, t: int) -> int:
            return self._local[t & 1].ptr

        def post(self, t: int) -> None:
            self._all[t & 1] = np.concatenate(rdv.gather(self._local[t & 1].download()))

        def gathered(self, t: int) -> np.ndarray:
            return self._all[t & 1]

        def drain(self) -> None:
            pass

        def close(self) -> None:
            pass

    # the one exchange of the sharded rollout: all-gather of the observation block, RCCL behind the C-ABI
    exchange = None
    if world > 1 and args.dist_backend == "nccl":
        from rcs_amd.envs.sharding import RcclObservationExchange, comm_unique_id

        # ncclCommInitRank is collective: a rank that cannot even load RCCL must not leave the others waiting inside it.  Every rank
        # first asks the library for an id of its own (that loads librccl and resolves its entry points); only if all of them could
        # does rank 0's id go round and the communicator get created.
        comm_error, my_id = "", None
        try:
            my_id = comm_unique_id()
        except RuntimeError as exc:
            comm_error = str(exc)
        all_ok = rdv.reduce(0 if comm_error else 1, min)
        if all_ok == 1:
            root_id = rdv.broadcast(my_id if rank == 0 else None)
            try:
                exchange = RcclObservationExchange(env.sim, bytes(root_id), rank, world, n_rows=n, width=ow)
            except RuntimeError as exc:  # e.g. a communicator RCCL refuses on this topology
                comm_error = str(exc)
            all_ok = rdv.reduce(0 if comm_error else 1, min)  # every rank takes the same carrier
        if all_ok == 0:
            if exchange is not None:
                exchange.close()
            exchange = HostStagedExchange()
            args.dist_backend = "the rendezvous socket through host memory (the C-ABI RCCL communicator could not be created: " + (comm_error or "on another rank") + ")"
    elif world > 1 and args.dist_backend == "sdma":
        # the same slot protocol over copy engines (rcsh_comm_copy_*): blocks written into the peers' IPC-mapped receive buffers, no
        # collective kernel (rcs_amd.envs.sharding.CopyObservationExchange).  One rank per process; ranks may share a device.
        from rcs_amd.envs.sharding import CopyObservationExchange

        exchange = CopyObservationExchange(env.sim, rank, world, lambda blob: [bytes(b) for b in rdv.gather(blob)], n_rows=n, width=ow)
    elif world > 1:
        exchange = HostStagedExchange()

    episode = args.episode_length if args.episode_length is not None else (10 if args.control == "cartesian" else 0)

    task_out = box_pose = None
    if args.task == "pick_up":
        # RandomCubePos placements of every reset, resident: x, y ~ iso_cube +- 0.1 m, z = 14.4 mm, quaternion (w ~ U(-1, 1), 0, 0, 1)
        n_resets = T // max(episode, 1) + 2
        bp = np.zeros((n_resets, n, 7))
        u = rng.random((n_resets, n, 3))
        bp[..., 0] = 0.498 + u[..., 0] * 0.2 - 0.1
        bp[..., 1] = u[..., 1] * 0.2 - 0.1
        bp[..., 2] = 0.0288 / 2
        bp[..., 3] = 2 * u[..., 2] - 1
        bp[..., 6] = 1.0
        box_pose = DevBuf((n_resets, n, 7), np.float64, bp)
        task_out = DevBuf((n, 9), np.float64)

    cam_set, cam_out = None, {}
    if args.cameras:
        from rcs_amd.camera import SimCameraConfig, SimCameraSet

        rw, rh = (int(x) for x in args.resolution.split("x"))
        cam_set = SimCameraSet(env.sim, {c: SimCameraConfig(identifier=c, resolution_width=rw, resolution_height=rh) for c in args.cameras.split(",")},
                               physical_units=True)
        cam_out = {c: DevBuf((n, rh, rw), np.uint16) for c in cam_set.camera_names}

    def do_reset(t: int) -> None:
        if box_pose is not None:
            env.reset_task_dev(box_pose.at(t // max(episode, 1)), obs.ptr, info.ptr, gw.ptr)
        else:
            row = 0
            for e_ in envs:
                e_.reset_dev(obs.ptr + 8 * row * ow, info.ptr + 8 * row, gw.ptr + 8 * row)
                row += e_.n_envs
            device_sync()

    def one_step(t: int) -> None:
        if episode and t % episode == 0:
            do_reset(t)
        optr = exchange.local_ptr(t) if exchange else obs.ptr
        for e_ in envs[1:]:
            # (the slot of the exchange this step writes was released on the first handle's stream -- and the previous step's
            # render / gather read the block from there: rcsh_sim_wait_for orders the sub-batches' streams behind that point,
            # BEFORE the first sub-batch's own launch goes out, so that all sub-batches run side by side)
            _lib.check(L.rcsh_sim_wait_for(e_.sim._h, h))
        if task_out is not None:
            env.step_task_dev(joints.at(t), grip.at(t), optr, info.ptr, gw.ptr, sub.ptr, task_out.ptr)
        else:
            env.step_dev(joints.at(t), grip.at(t), optr, info.ptr, gw.ptr, sub.ptr)
        row = env.n_envs
        for e_, j_ in zip(envs[1:], sub_joints[1:]):
            e_.step_dev(j_.at(t), grip.at(t, 4 * row), optr + 8 * row * ow, info.ptr + 8 * row, gw.ptr + 8 * row, sub.ptr + 4 * row)
            row += e_.n_envs
        for e_ in envs[1:]:
            _lib.check(L.rcsh_sim_wait_for(h, e_.sim._h))  # the gather (and the clock) see all sub-batches
        for c, buf in cam_out.items():
            cam_set.render_depth_mm_dev(c, buf.ptr)
        if exchange:
            exchange.post(t)  # overlaps with the next env-step

    do_reset(0)
    # Clocks first.  A device that sat idle through the set-up starts its first launches below its operating frequency and needs
    # ~12 ms of work to get there: stepping launches (the measured ones, exchange off) for --clock-warmup-ms, then the environments
    # are reset and the contract's W warmup steps and K timed steps run as specified -- at the clocks a rollout of any length sees.
    clock_warmup_launches = 0
    if args.clock_warmup_ms > 0:
        saved_exchange, exchange = exchange, None
        t_end = time.perf_counter() + args.clock_warmup_ms * 1e-3
        while time.perf_counter() < t_end:
            for t in range(8):
                one_step(clock_warmup_launches % T if not episode else 1 + clock_warmup_launches % max(min(episode, T) - 1, 1))
                clock_warmup_launches += 1
            device_sync()
        exchange = saved_exchange
        do_reset(0)
    for t in range(args.warmup):
        one_step(t)

    # HIP events on the launch stream: ONE pair around the whole timed region (rcsh_prof_enable(h, -1)) -- the stream time of
    # the region's stepping launches, dispatch gaps included, divided by their number.  (A pair around every launch puts ~8 us
    # of dispatch gap into each step of the run it is measuring; pairs around every 8th launch, rounds 1-2, left 3 samples at
    # the driver's --steps 20 and a per-launch figure above the step time it is part of.)
    _lib.check(L.rcsh_prof_enable(h, -1))
    if exchange:
        exchange.drain()
    device_sync()
    rdv.barrier()
    t0 = time.perf_counter()
    for t in range(args.warmup, T):
        one_step(t)
    # the region's closing event goes onto the launch stream right behind the last step (reading it waits for that stream)
    ms, launches = C.c_double(0), C.c_int64(0)
    _lib.check(L.rcsh_prof_read(h, C.byref(ms), C.byref(launches)))
    if exchange:
        exchange.drain()
    device_sync()
    rdv.barrier()
    elapsed = time.perf_counter() - t0
    obs_host = obs.download()
    if exchange:  # (after the clock stopped) this rank's rows of the last gathered tensor: finiteness check below
        obs_host = np.asarray(exchange.gathered(T - 1))[rank * n:(rank + 1) * n]
    elapsed = rdv.reduce(elapsed, max)

    kernel_ms = ms.value / max(launches.value, 1)
    # N > 1: the same K steps once more WITHOUT the exchange (after the clock stopped; reported next to the measured value, never
    # as it) -- what the all-gather costs a step on this node, i.e. how much of a scaling loss is the exchange's and how much the shards'
    no_exchange_value = exchange_ms = None
    if world > 1 and exchange is not None and not episode:
        saved_exchange, exchange = exchange, None
        device_sync()
        rdv.barrier()
        t1 = time.perf_counter()
        for t in range(args.warmup, T):
            one_step(t)
        device_sync()
        rdv.barrier()
        no_exchange_value = world * n * args.steps / rdv.reduce(time.perf_counter() - t1, max)
        exchange = saved_exchange
        # ... and the gather ALONE, back to back on an otherwise idle device: what one exchange costs when nothing hides it
        exchange.drain()
        rdv.barrier()
        t2 = time.perf_counter()
        for t in range(args.steps):
            exchange.local_ptr(t)  # (retires the gather that last used the slot)
            exchange.post(t)
        exchange.drain()
        rdv.barrier()
        exchange_ms = rdv.reduce(time.perf_counter() - t2, max) / args.steps * 1e3
    mean_sub = float(sub.download().astype(np.float64).mean())
    # Environments found in a contact the running configuration does not resolve, by the end of the timed region (the sticky flag
    # of the end-of-launch check, csrc/check_team.h; cleared by the reset before the W warmup steps).  The words "no contacts" in
    # config.workload are DERIVED from this count being zero on every rank.
    contacts_seen = contacts_resolved = escalated_now = contacts_unresolved = 0
    for e_ in envs:
        unres = e_.sim.contact_unresolved()
        now_, ever_ = e_.sim.contact_escalated()
        contacts_unresolved += int(unres.sum())
        contacts_resolved += int(ever_.sum())
        escalated_now += int(now_.sum())
        contacts_seen += int((unres | ever_ | now_).sum())
    contacts_seen = rdv.reduce(contacts_seen, sum)
    contacts_resolved = rdv.reduce(contacts_resolved, sum)
    contacts_unresolved = rdv.reduce(contacts_unresolved, sum)
    escalated_now = rdv.reduce(escalated_now, sum)
    resolving = bool(env.sim.resolve_robot_contacts)
    # What a Gymnasium user calls: the same env-step through the host-array interface (rcsh_env_step: numpy in, numpy out, two PCIe
    # copies and a stream synchronise per step) -- after the clock stopped, reported next to `value`, never as it.
    value_host_api = None
    headline_cfg = (args.mode == "async" and args.task == "none" and args.robot == "fr3" and args.control == "joints" and not mixed and not cam_out and world == 1)
    headline_cfg = headline_cfg and not args.no_extras
    if headline_cfg and rank == 0:
        a_host = (np.random.default_rng(7).random((32, n, env.dof)) * 2 - 1) * MAX_JOINT_MOV
        g_host = np.random.default_rng(8).random((32, n)).astype(np.float32)
        for t_ in range(4):
            env.step({"joints": a_host[t_], "gripper": g_host[t_]})
        th = time.perf_counter()
        for t_ in range(4, 32):
            env.step({"joints": a_host[t_], "gripper": g_host[t_]})
        value_host_api = n * 28 / (time.perf_counter() - th)
    # ... and the round-4 configuration of the same workload (lean kernels only, contacts flagged, the check every 16th env-step): what
    # resolving contacts environment by environment costs a rollout in which nothing touches anything
    value_flag_only = None
    if headline_cfg and rank == 0 and resolving:
        env4 = make_vec_env(n, async_control=True, gripper=True, relative=True, device=local_rank, robot=args.robot, resolve_robot_contacts=False)
        env4.sim.set_contact_check(16)
        env4.reset_dev(obs.ptr, info.ptr, gw.ptr)
        for t_ in range(min(T, 64)):
            env4.step_dev(joints.at(t_), grip.at(t_), obs.ptr, info.ptr, gw.ptr, sub.ptr)
        env4.reset_dev(obs.ptr, info.ptr, gw.ptr)
        env4.sim.synchronize()
        t4 = time.perf_counter()
        for t_ in range(args.warmup, T):
            env4.step_dev(joints.at(t_), grip.at(t_), obs.ptr, info.ptr, gw.ptr, sub.ptr)
        env4.sim.synchronize()
        value_flag_only = n * args.steps / (time.perf_counter() - t4)
        env4.close()
    # ... and the two figures BASELINE.md / SURVEY 8(d) quote next to the driver's K steps, on the same workload (after the clock stopped):
    # * BASELINE.md section 3's rollout length -- T = 1000 env-steps after 50 warm-up, no resets -- by which a quarter of the batch has
    #   folded onto the floor or itself and every step carries the contact-resolving launch of those environments;
    # * the reference's default mode, Sim.step_until_convergence (cap 500), 30 env-steps after 5.
    value_rollout = value_until_conv = None
    rollout_info = {}
    if headline_cfg and rank == 0 and resolving and args.contacts == "resolve":
        def extra_run(envx, steps_, warm_):
            acts = DevBuf((steps_ + warm_, n, envx.dof), np.float64, (np.random.default_rng(4321).random((steps_ + warm_, n, envx.dof)) * 2 - 1) * MAX_JOINT_MOV)
            grp = DevBuf((steps_ + warm_, n), np.float32, np.random.default_rng(4322).random((steps_ + warm_, n), dtype=np.float32))
            envx.reset_dev(obs.ptr, info.ptr, gw.ptr)
            for t_ in range(warm_):
                envx.step_dev(acts.at(t_), grp.at(t_), obs.ptr, info.ptr, gw.ptr, sub.ptr)
            envx.sim.synchronize()
            tx = time.perf_counter()
            for t_ in range(warm_, warm_ + steps_):
                envx.step_dev(acts.at(t_), grp.at(t_), obs.ptr, info.ptr, gw.ptr, sub.ptr)
            envx.sim.synchronize()
            return n * steps_ / (time.perf_counter() - tx)

        if (args.steps, args.warmup) == (1000, 50):
            value_rollout = None  # (this run IS that rollout: `value`)
        else:
            value_rollout = extra_run(env, 1000, 50)
            now_, ever_ = env.sim.contact_escalated()
            rollout_info = {"contacts_resolved": int(ever_.sum()), "escalated_at_end": int(now_.sum()), "contacts_unresolved": int(env.sim.contact_unresolved().sum())}
        envc = make_vec_env(n, async_control=False, gripper=True, relative=True, device=local_rank, robot=args.robot)
        value_until_conv = extra_run(envc, 30, 5)
        envc.close()
    finite = bool(np.isfinite(obs_host).all())
    if task_out is not None:
        finite = finite and bool(np.isfinite(task_out.download()).all())
    rccl_exchange = exchange is not None and args.dist_backend in ("nccl", "sdma")  # (device-side carriers: posted and waited for on streams)

    SCENE_OF = {"fr3": "fr3_empty_world", "xarm7": "xarm7_empty_world", "xarm7_box": "xarm7_box_world", "xarm7_pick": "xarm7_pick_world", "arm6": "arm6_empty_world",
                "ur5e": "ur5e_empty_world", "so101": "so101_empty_world"}
    SCENE_LABEL = {**SCENE_OF, "xarm7_box": "xarm7_box_world (free cube, elliptic-cone floor contacts)",
                   "xarm7_pick": "xarm7_pick_world (xArm7 + two-finger gripper + free cube: gripper / cube / floor contacts resolved, dry-friction rows in the coupled solve)", "arm6": "arm6_empty_world (builder-authored 6-dof arm)",
                   "ur5e": "ur5e_empty_world (builder-authored, UR5e proportions)", "so101": "so101_empty_world (builder-authored 5-dof arm + two-finger gripper)"}
    mixed_label = ("fr3 / xarm7 / ur5e (builder-authored) / so101 (builder-authored) _empty_world, robot type = rank mod 4" if world >= 4 else
                   f"fr3 / xarm7 / ur5e (builder-authored) / so101 (builder-authored) _empty_world, {len(hosted)} types per rank as sub-batches of {n // len(hosted)} on streams of their own")
    if rank == 0:
        total_env_steps = world * n * args.steps
        value = total_env_steps / elapsed
        # SURVEY 8(d)'s per-env-step figure for the configuration actually run (1292 B for the FR3 + hand in JOINTS mode): action in,
        # physics state (qpos, qvel, warm start per dof + ctrl per actuator + time) and RCS state (target / previous / previous action per
        # arm joint, 6 callback timestamps, 2 gripper widths, 8 flag bytes) read and written, observation and info out
        nl_, nu_, narm_ = int(env.sim.model.nv), int(env.sim.model.nu), int(env.dof)
        act_b = (6 if args.control == "cartesian" else narm_) * 8 + 4
        phys_b = 2 * 8 * (3 * nl_ + nu_ + 1)
        rcs_b = 2 * (8 * (3 * narm_ + 8) + 8)
        per_env = act_b + phys_b + rcs_b + 8 * (14 + narm_) + 8
        if args.control == "cartesian":
            per_env += 2 * 8 * 20  # k_cartesian_team: ~20 f64 of state read and written
        if getattr(env.sim.model, "free_bodies", []):
            per_env += 2 * 8 * 41 + (9 * 8 if task_out is not None else 0)  # the free body's state r/w, the task's outputs
        assert not (args.robot == "fr3" and args.control == "joints" and args.task == "none") or per_env == ALGO_BYTES_PER_ENV_STEP
        algo_bytes = per_env * env.n_envs  # (the timed kernel is the first sub-batch's when a rank hosts several)
        if cam_out:  # (the depth frames the timed region writes: 2 bytes a pixel -- what the ray caster's roofline is about)
            rw_, rh_ = (int(x) for x in args.resolution.split("x"))
            algo_bytes += len(cam_out) * env.n_envs * rw_ * rh_ * 2
        achieved_gbs = algo_bytes / (kernel_ms * 1e-3) / 1e9
        substeps_per_launch = mean_sub * env.n_envs
        traffic = None
        tj_extra = {}
        import glob

        tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*_traffic.json")), key=lambda f: int(os.path.basename(f)[1:].split("_")[0]))
        tpath = tfiles[-1] if tfiles else os.path.join(ROOT, "profiles", "r4_traffic.json")  # (the latest round's PMC passes)
        headline = args.mode == "async" and n == N_ENVS and args.task == "none" and args.robot == "fr3" and args.control == "joints" and not mixed
        if os.path.exists(tpath) and headline:  # the PMC passes profiled exactly this workload (profiles/run_profile.sh)
            tj = json.load(open(tpath))  # PMC pass of this same command (profiles/run_profile.sh), bytes per launch
            tj_extra = tj
            # (counter units are KiB; corrected as calibrated on this access pattern: profiles/r2_hbm_calib, tools/hbm_calib.hip)
            traffic = (tj["fetch_size_kb_per_dispatch"] * tj.get("fetch_correction", 1.0) + tj["write_size_kb_per_dispatch"] * tj.get("write_correction", 1.0)) * 1024
        out = {
            "metric": "env-steps/sec (whole node), " + (
                "fr3_simple_pick_up task (CARTESIAN_TRPY)" if args.task != "none" else
                ("fr3 / xarm7 / ur5e / so101 _empty_world by rank" if mixed else SCENE_OF[args.robot])
                + (" JOINTS mode" if args.control == "joints" else " CARTESIAN_TRPY mode"))
                + (", step_until_convergence" if args.mode == "convergence" else "") + f", {n} envs per GPU",
            "value": value,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": (f"{n}x fr3_empty_world batched JOINTS per GPU, relative +-5deg actions, gripper commanded, "
                             + ("robot contacts resolved, IK off" if args.robot == "xarm7_pick" else
                                ("no contacts (checked every env-step: 0 environments touched the floor or themselves), IK off" if contacts_seen == 0 else
                                 (f"{contacts_seen} environments touched the floor or themselves: resolved, environment by environment "
                                  f"({contacts_resolved} resolved, {escalated_now} on the contact-resolving kernel at the end), IK off" if resolving else
                                  f"{contacts_seen} environments ran into a contact the lean kernels do not resolve (flagged: info.contact_unresolved), IK off")))
                             if args.control == "joints" else
                             f"{n}x fr3_empty_world batched CARTESIAN_TRPY per GPU, relative +-5cm / +-0.1rad actions -> CLIK, gripper commanded"
                             ).replace("fr3_empty_world", (mixed_label if mixed else SCENE_LABEL[args.robot]) if args.task == "none" else
                                       "fr3_simple_pick_up (free cube: plane-box contacts, elliptic cones, noslip; RandomCubePos + PickCubeSuccessWrapper)"),
                "mode": "async_control 30Hz (17 substeps/env-step)" if args.mode == "async" else "step_until_convergence (cap 500)",
                "envs_per_gpu": n,
                "substeps_per_env_step": mean_sub,
                "physics_substeps_per_s": value * mean_sub,
                "episode_length": episode or None,
                "depth_frames": (f"{args.cameras} at {args.resolution}, one ray-cast uint16 frame per camera per env-step "
                                 f"({len(cam_out) * n * int(args.resolution.split('x')[0]) * int(args.resolution.split('x')[1]) / (elapsed / args.steps) / 1e9:.2f} G rays/s incl. the physics)") if cam_out else None,
                "exchange": ((("copy engines behind the C-ABI (rcsh_comm_copy_*: every rank writes its block into the peers' IPC-mapped receive buffers, one wavefront waits for the flag words)" if args.dist_backend == "sdma" else "RCCL ncclAllGather behind the C-ABI (rcsh_env_allgather_obs_dev)") if rccl_exchange else f"all-gather over {args.dist_backend}")
                             + f" of obs [N,{ow}] f64 per step, double-buffered, overlapped with the next env-step") if world > 1 else "none (1 GPU)",
                "contacts_seen": contacts_seen,
                "contacts_resolved": contacts_resolved,
                "contacts_unresolved": contacts_unresolved,
                "escalated_at_end": escalated_now,
                "contacts": ("resolved environment by environment (lean launch + contact-resolving launch per env-step; an environment found in contact at "
                             "the end of a lean launch has that launch redone with its contacts resolved; csrc/sim_kernels.h RunOp::esc_role)" if resolving and not env.sim.model.free_bodies
                             else ("resolved by the whole batch on the contact-resolving kernels" if resolving else "detected only (sticky info.contact_unresolved)")),
                "contact_check": ("exact collision check of every lean-kernel environment at the end of EVERY env-step, inside the timed region (csrc/check_team.h)" if resolving and not env.sim.model.free_bodies else
                                  (f"exact collision check of every environment every {args.contact_check_every} env-steps inside the timed region"
                                   if args.contact_check_every > 0 else "no check inside the timed region") + " + once on the final state (sticky flags, csrc/check_team.h)"),
                "value_host_api": value_host_api,  # the same env-step through rcsh_env_step (numpy in / out): what a Gymnasium user calls
                "value_flag_only_check_every_16": value_flag_only,  # round 4's configuration of this workload (contacts flagged, not resolved)
                "value_rollout_1000_50": value_rollout,  # BASELINE.md section 3's rollout of this workload: T = 1000 after 50 warm-up, no resets
                "rollout_1000_50": rollout_info or None,  # ... environments whose contacts were resolved / on the contact-resolving launch at its end
                "value_until_convergence_30": value_until_conv,  # the reference's default mode (step_until_convergence, cap 500): 30 env-steps after 5
                "obs_finite": finite,
                "value_without_exchange": no_exchange_value,
                "exchange_ms": exchange_ms,  # the gather alone, back to back (after the clock stopped)
                "rccl_channels": (os.environ.get("NCCL_MIN_NCHANNELS"), os.environ.get("NCCL_MAX_NCHANNELS")) if world > 1 else None,
                "rendezvous": "Unix-domain socket (rcs_amd.envs.sharding.SocketRendezvous): no torch.distributed in any rank" if world > 1 else None,
                "clock_warmup": (f"{clock_warmup_launches} untimed launches over {args.clock_warmup_ms:g} ms before the W warmup steps, then a reset "
                                 "(the device reaches its operating clocks after ~12 ms of work)") if clock_warmup_launches else None,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved_gbs,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": (f"rocprofv3 FETCH_SIZE x 2 (gfx950: counts half the bytes of this access pattern, calibrated in profiles/r2_hbm_calib) + WRITE_SIZE, "
                                   f"separate PMC passes (profiles/{os.path.basename(tpath)}: {tj_extra.get('source', 'profiles/run_profile.sh')})") if traffic else None,
                "kernel": (("k_render_depth<false, float> (the ray caster: most of this region at this resolution, profiles/r4_render) + " if cam_out and
                            int(args.resolution.split("x")[0]) * int(args.resolution.split("x")[1]) >= 128 * 128 else "")
                           + "k_run_team" + f"<Topo<{env.dof},{'true' if env.gripper is not None else 'false'}>> (fused env-step)"
                           + (" + free box" if args.task != "none" or args.robot in ("xarm7_box", "xarm7_pick") else "")
                           + (" + k_cartesian_team (CLIK)" if args.control == "cartesian" else "")
                           + (" [lean launch + contact-resolving launch over the escalated environments]" if resolving and not getattr(env.sim.model, "free_bodies", []) and args.robot == "fr3" else "")),
                "algorithmic_bytes_per_env_step": per_env,
                # what the region's event pair brackets: every launch of a step on the handle's stream, not the stepping kernel alone
                "timed_region": "k_run_team" + (" + k_cartesian_team" if args.control == "cartesian" else "") + (" + the depth frames' kernels (k_link_frames, k_shape_frames, k_hull_views, k_render_depth)" if cam_out else "")
                                + (" + the reset launches" if episode else "") + ", dispatch gaps included",
                "kernel_ms_avg": kernel_ms,
                "launches_timed": int(launches.value),
                "algorithmic_bytes_per_launch": algo_bytes,
                "fp64_algorithmic_tflops": substeps_per_launch * ALGO_FLOP_PER_SUBSTEP / (kernel_ms * 1e-3) / 1e12,
                "fp64_vector_peak_tflops": FP64_VECTOR_PEAK_TFLOPS,
                "valu_busy_frac": tj_extra.get("valu_busy_frac"),
                "wait_any_frac": tj_extra.get("wait_any_frac"),
                "note": "path is FP64 VALU-issue bound, not HBM bound (SURVEY F7; profiles/README.md: every SIMD holds one wavefront that issues ~1 VALU instruction per 5 cycles): the HBM fraction is reported as the contract asks",
            },
        }
        if cpu_base is not None:
            out["cpu_baseline"] = cpu_base
        # (librccl prints its version banner through C stdio, which is flushed at exit -- behind this line; the JSON line is the LAST line)
        try:
            C.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if exchange:
        exchange.close()
    rdv.close()


if __name__ == "__main__":
    main()
