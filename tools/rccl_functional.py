"""Functional run of the RCCL exchange behind the C-ABI (rcsh_comm_*), no torch on the data path.

    python tools/rccl_functional.py [world]      # spawns `world` processes (default 2) that share whatever GPUs exist

Each rank builds a small FR3 batch, steps it, all-gathers the observation rows and checks that every rank's rows arrived
in rank order.  The unique id travels from rank 0 to the others through a multiprocessing pipe.
"""
import multiprocessing as mp
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "robot-control-stack_amd"))


def worker(rank, world, uid_q, out_q, ndev):
    import ctypes as C

    import numpy as np

    from rcs_amd import _lib
    from rcs_amd.envs import make_vec_env
    from rcs_amd.envs.sharding import RcclObservationExchange, comm_unique_id

    try:
        n = 64
        env = make_vec_env(n, True, device=rank % ndev)
        if rank == 0:
            uid = comm_unique_id()
            for _ in range(world - 1):
                uid_q.put(uid)
        else:
            uid = uid_q.get(timeout=120)
        ex = RcclObservationExchange(env.sim, uid, rank, world)
        L, h = env._L, env.sim._h
        env.reset()
        rng = np.random.default_rng(100 + rank)
        act = np.zeros((n, env.dof)); grip = np.ones(n, dtype=np.float32)
        dact, dgrip = C.c_void_p(), C.c_void_p()
        _lib.check(L.rcsh_dev_alloc(h, act.nbytes, C.byref(dact))); _lib.check(L.rcsh_dev_alloc(h, grip.nbytes, C.byref(dgrip)))
        ok = True
        mine = None
        for t in range(6):
            act[:] = rng.uniform(-0.05, 0.05, act.shape)
            _lib.check(L.rcsh_dev_upload(h, dact, act.ctypes.data_as(C.c_void_p), act.nbytes))
            _lib.check(L.rcsh_dev_upload(h, dgrip, grip.ctypes.data_as(C.c_void_p), grip.nbytes))
            env.step_dev(dact.value, dgrip.value, ex.local_ptr(t))
            ex.post(t)
            g = ex.gathered(t)
            mine = g[rank * n:(rank + 1) * n]
            ok = ok and np.isfinite(g).all() and g.shape == (world * n, ex.width)
            # ranks run different actions: rows of different ranks differ, rows of this rank equal its own observation
            q = env.sim.qpos[:, :7]
            ok = ok and np.abs(mine[:, 7:14] - q).max() < 1e-12
            if world > 1:
                other = g[((rank + 1) % world) * n:((rank + 1) % world + 1) * n]
                ok = ok and np.abs(other - mine).max() > 1e-6
        ex.close()
        env.close()
        out_q.put((rank, bool(ok), float(mine[0, 7])))
    except Exception as exc:  # noqa: BLE001
        out_q.put((rank, False, repr(exc)))


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    import ctypes as C

    from rcs_amd import _lib

    ndev = max(int(_lib.load().rcsh_device_count()), 1)
    ctx = mp.get_context("spawn")
    uid_q, out_q = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, uid_q, out_q, ndev)) for r in range(world)]
    for p in procs:
        p.start()
    res = [out_q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(60)
    print(f"world={world} devices={ndev} results={sorted(res)}")
    sys.exit(0 if all(r[1] for r in res) else 1)


if __name__ == "__main__":
    main()
