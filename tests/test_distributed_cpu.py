"""CPU, world_size 2, gloo: the N > 1 path of the rollout -- contiguous env shards per rank, one all-gather of the
observation tensor per env-step, max-over-ranks timing reduction (what bench.py does over RCCL)."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rcs_amd.envs.sharding import ObservationExchange, gather_observations, shard_range


def test_shard_range_partitions_exactly():
    for n, w in [(4096, 1), (4096, 2), (4096, 8), (32768, 8), (10, 4), (3, 4)]:
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def _worker(rank, world, port, n_total, width, steps, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b = shard_range(n_total, rank, world)
    obs_all = None
    ok = True
    for t in range(steps):
        ids = torch.arange(a, b, dtype=torch.float64).unsqueeze(1)
        obs_local = ids * 100.0 + torch.arange(width, dtype=torch.float64).unsqueeze(0) + 0.001 * t  # "observation" of env id
        obs_all = gather_observations(obs_local, obs_all)
        expect = torch.arange(n_total, dtype=torch.float64).unsqueeze(1) * 100.0 + torch.arange(width, dtype=torch.float64).unsqueeze(0) + 0.001 * t
        ok = ok and bool(torch.equal(obs_all, expect))
    # the overlapped form bench.py uses: double-buffered, gather of step t retired when step t+2 needs the buffer
    ex = ObservationExchange(b - a, width, torch.float64, "cpu")
    ids = torch.arange(a, b, dtype=torch.float64).unsqueeze(1)
    all_ids = torch.arange(n_total, dtype=torch.float64).unsqueeze(1)
    cols = torch.arange(width, dtype=torch.float64).unsqueeze(0)
    for t in range(steps + 2):
        ex.local(t).copy_(ids * 100.0 + cols + 0.001 * t)
        ex.post(t)
        if t >= 1:
            ok = ok and bool(torch.equal(ex.gathered(t - 1), all_ids * 100.0 + cols + 0.001 * (t - 1)))
    ex.drain()
    ok = ok and bool(torch.equal(ex.gathered(steps + 1), all_ids * 100.0 + cols + 0.001 * (steps + 1)))
    elapsed = torch.tensor([0.5 + rank], dtype=torch.float64)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    ok = ok and float(elapsed) == 0.5 + world - 1
    dist.barrier()
    out.put((rank, ok))
    dist.destroy_process_group()


def test_two_rank_observation_all_gather():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 64, 21, 3, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    results = dict(out.get(timeout=5) for _ in range(2))
    assert results == {0: True, 1: True}
    _ = np


def _socket_worker(rank, world, name, out):
    from rcs_amd.envs.sharding import SocketRendezvous

    rdv = SocketRendezvous(rank, world, name=name, timeout=30)
    ok = rdv.gather(["r", rank]) == [["r", r] for r in range(world)]  # (lists: the wire format is JSON, a tuple comes back as a list)
    ok = ok and rdv.broadcast(b"id-from-rank-0" if rank == 0 else None) == b"id-from-rank-0"
    ok = ok and rdv.reduce(0.5 + rank, max) == 0.5 + world - 1 and rdv.reduce(rank, sum) == sum(range(world)) and rdv.reduce(1 if rank else 0, min) == 0
    # an all-gather of array blocks through the rendezvous (the functional fallback of the exchange when RCCL cannot form a communicator)
    a, b = shard_range(10, rank, world)
    blocks = rdv.gather(np.arange(a, b, dtype=np.float64))
    ok = ok and np.array_equal(np.concatenate(blocks), np.arange(10.0))
    rdv.barrier()
    rdv.close()
    out.put((rank, bool(ok)))


def test_socket_rendezvous_three_ranks_without_torch():
    """bench.py's N > 1 launcher rendezvous (RCCL id broadcast, barriers, max over ranks of the clock) over a Unix-domain socket:
    no torch.distributed process group involved."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    name = f"rcs_amd_test_{os.getpid()}"
    procs = [ctx.Process(target=_socket_worker, args=(r, 3, name, out)) for r in range(3)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert dict(out.get(timeout=5) for _ in range(3)) == {0: True, 1: True, 2: True}
