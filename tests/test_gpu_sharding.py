"""GPU: the sharded rollout's one exchange -- the all-gather of the observation block over RCCL behind the C-ABI
(rcs_amd.envs.sharding.RcclObservationExchange: rcsh_comm_init / rcsh_comm_allgather_dev / rcsh_comm_wait, two slots, its own
stream) -- and BASELINE configs[4]'s shards at their per-GPU size.

On a box with ONE GPU (the development box) RCCL forms a communicator of one rank only ("Duplicate GPU detected" otherwise):
the slot protocol then runs with world = 1 here, with world = 2 over gloo in tests/test_distributed_cpu.py, and the multi-rank
test below skips.  On a multi-GPU box it spawns one process per GPU and checks rank order, overlap slots and contents."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, "..", "robot-control-stack_amd"), HERE]

pytestmark = pytest.mark.gpu


def _dev_upload(L, h, arr):
    from rcs_amd import _lib

    p = C.c_void_p()
    _lib.check(L.rcsh_dev_alloc(h, arr.nbytes, C.byref(p)))
    _lib.check(L.rcsh_dev_upload(h, p, arr.ctypes.data_as(C.c_void_p), arr.nbytes))
    return p


def _exchange_worker(rank, world, uid_q, out_q, n_steps, carrier="rccl", same_device=False, fold=False):
    """One rank of the slot protocol: step, post slot t & 1, keep stepping into the other slot, read the gathered block two steps
    later (the overlap bench.py relies on).  Every rank's rows must be that rank's observations, in rank order."""
    try:
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")  # one node: RCCL's bootstrap over the loopback interface
        from rcs_amd.envs import make_vec_env
        from rcs_amd.envs.sharding import RcclObservationExchange, comm_unique_id

        n = 96
        env = make_vec_env(n, True, device=0 if same_device else rank)
        L, h = env._L, env.sim._h
        if carrier == "copy":
            from rcs_amd.envs.sharding import CopyObservationExchange

            def gather_blobs(blob):
                # (the test's side channel: every rank puts its blob on its peers' queues -- uid_q is a list of per-rank queues here)
                for r in range(world):
                    if r != rank:
                        uid_q[r].put((rank, blob))
                blobs = {rank: blob}
                while len(blobs) < world:
                    r, b = uid_q[rank].get(timeout=120)
                    blobs[r] = b
                return [blobs[r] for r in range(world)]

            make = lambda: CopyObservationExchange(env.sim, rank, world, gather_blobs)  # noqa: E731
        else:
            if rank == 0:
                uid = comm_unique_id()
                for _ in range(world - 1):
                    uid_q.put(uid)
            else:
                uid = uid_q.get(timeout=120)
            make = lambda: RcclObservationExchange(env.sim, uid, rank, world)  # noqa: E731
        with make() as ex:
            env.reset()
            rng = np.random.default_rng(100 + rank)
            acts = rng.uniform(-0.05, 0.05, (n_steps, n, env.dof))
            if fold:  # half of the environments lean forward until hand and wrist lie on the floor (from step ~95): escalation on every rank
                acts[:, : n // 2, 1] = 0.08
                acts[:, : n // 2, 3] = 0.08
            dact, dgrip = _dev_upload(L, h, acts), _dev_upload(L, h, np.ones(n, dtype=np.float32))
            ok, own = True, {}
            for t in range(n_steps):
                env.step_dev(dact.value + t * n * env.dof * 8, dgrip.value, ex.local_ptr(t))
                ex.post(t)
                own[t] = env.sim.qpos[:, :7].copy()
                if t >= 1:  # the gather of the previous step has had a whole env-step to finish
                    g = ex.gathered(t - 1)
                    ok = ok and g.shape == (world * n, ex.width) and bool(np.isfinite(g).all())
                    ok = ok and float(np.abs(g[rank * n:(rank + 1) * n, 7:14] - own[t - 1]).max()) < 1e-12
                    if world > 1:  # ranks draw different actions: another rank's rows differ from this rank's
                        other = (rank + 1) % world
                        ok = ok and float(np.abs(g[other * n:(other + 1) * n] - g[rank * n:(rank + 1) * n]).max()) > 1e-6
            ex.drain()
            if fold:  # (both launches of a step ran, both slots were reused, while environments went to the contact-resolving launch and came back)
                now, ever = env.sim.contact_escalated()
                ok = ok and int(ever.sum()) >= n // 4 and int(now.sum()) >= n // 4 and not env.sim.contact_unresolved().any()
        env.close()
        out_q.put((rank, bool(ok), ""))
    except Exception as exc:  # noqa: BLE001
        out_q.put((rank, False, repr(exc)))


def _run_exchange(world, carrier="rccl", same_device=False, n_steps=6, fold=False):
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    uid_q, out_q = (ctx.Queue() if carrier == "rccl" else [ctx.Queue() for _ in range(world)]), ctx.Queue()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, uid_q, out_q, n_steps, carrier, same_device, fold)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(out_q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
    assert all(r[1] for r in res), res


def test_rccl_exchange_slot_protocol_single_rank():
    _run_exchange(1)


def test_rccl_exchange_slot_protocol_one_rank_per_gpu():
    from rcs_amd import _lib

    ndev = int(_lib.load().rcsh_device_count())
    if ndev < 2:
        pytest.skip("one GPU: RCCL refuses two ranks on one device (covered over gloo in tests/test_distributed_cpu.py)")
    _run_exchange(2)  # (two ranks exercise everything the protocol has; the full node is the driver's scaling run)


def test_copy_engine_exchange_single_rank():
    """The copy-engine carrier of the all-gather (rcsh_comm_copy_*) with one rank: local block only, the slot protocol unchanged."""
    _run_exchange(1, carrier="copy")


def test_copy_engine_exchange_two_processes_on_one_gpu():
    """Two ranks in two processes on ONE device: every rank's block arrives in the other's IPC-mapped receive buffer, in rank
    order, with the two-slot overlap (a gather is read while the next env-step writes the other slot); 12 steps, so that every
    flag word is reused several times.  (The same code path a node's 8 processes take; there the copies run over xGMI.)"""
    _run_exchange(2, carrier="copy", same_device=True, n_steps=12)


def test_copy_engine_exchange_two_processes_with_escalating_environments():
    """Verdict r5, next 9: the copy carrier across 300 env-steps in which half of every rank's environments fold onto the floor -- every
    step is a lean launch AND a contact-resolving launch on the handle's stream, both slots of the exchange are reused 150 times, the
    gathered rows stay every rank's own observations of the step before."""
    _run_exchange(2, carrier="copy", same_device=True, n_steps=300, fold=True)


def test_copy_engine_exchange_one_rank_per_gpu():
    from rcs_amd import _lib

    if int(_lib.load().rcsh_device_count()) < 2:
        pytest.skip("one GPU (the two-process test above covers the protocol)")
    _run_exchange(2, carrier="copy")


def test_mixed_robot_shards_of_baseline_config_4():
    """BASELINE configs[4] -- "32768x mixed FR3 / xArm7 / UR5e / SO101 scenes sharded over 8 GPUs, RCCL all-gather of the
    observations" -- is 4096 environments of ONE robot type per GPU.  Here, on one GPU: each of the four shards at that size,
    one after the other, stepping into the exchange's block (n x 21 doubles per rank, narrower rows packed at its start) and
    all-gathering it through RCCL (world = 1); 16 distinct action streams tiled 256x: every copy equals the first bit for bit."""
    from rcs_amd import _lib
    from rcs_amd.envs import MAX_JOINT_MOV, make_vec_env
    from rcs_amd.envs.sharding import RcclObservationExchange, comm_unique_id

    n, base, steps = 4096, 16, 5
    for robot in ("fr3", "xarm7", "ur5e", "so101"):
        env = make_vec_env(n, True, robot=robot)
        L, h = env._L, env.sim._h
        with RcclObservationExchange(env.sim, comm_unique_id(), 0, 1, n_rows=n, width=21) as ex:
            rng = np.random.default_rng(3)
            acts = np.tile((rng.random((steps, base, env.dof)) * 2 - 1) * MAX_JOINT_MOV, (1, n // base, 1))
            grip = np.tile(rng.random((steps, base)).astype(np.float32), (1, n // base))
            dact, dgrip = _dev_upload(L, h, acts), _dev_upload(L, h, grip)
            obs0 = np.zeros((n, env.obs_width))
            dobs = _dev_upload(L, h, obs0)
            env.reset_dev(dobs.value)
            for t in range(steps):
                env.step_dev(dact.value + t * n * env.dof * 8, dgrip.value + t * n * 4, ex.local_ptr(t))
                ex.post(t)
            g = ex.gathered(steps - 1)
            rows = np.ascontiguousarray(g.reshape(-1)[: n * env.obs_width]).reshape(n, env.obs_width)  # (rows of obs_width doubles, packed)
            a = rows.reshape(n // base, base, -1)
            assert np.isfinite(rows).all() and np.array_equal(a, np.broadcast_to(a[0], a.shape)), robot
            assert np.abs(rows[:, 7:7 + env.dof] - env.sim.qpos[:, : env.dof]).max() < 1e-12, robot
            for p in (dact, dgrip, dobs):
                _lib.check(L.rcsh_dev_free(h, p))
        env.close()
