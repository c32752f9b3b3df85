"""Environment sharding across the GPUs of one node.

Environments are independent (the reference holds exactly one mjModel/mjData pair per Sim, reference
src/sim/sim.h:75-77), so the batch is split into contiguous env-id ranges, one range per rank, and nothing is
exchanged on the data path except the observation tensor: one all-gather per env-step (RCCL over xGMI on GPUs,
gloo in the CPU tests).
"""

from __future__ import annotations


def shard_range(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """[start, stop) of the env ids rank owns; the first n_total % world ranks get one extra env."""
    if not 0 <= rank < world:
        raise ValueError("rank outside world")
    base, extra = divmod(n_total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def gather_observations(obs_local, obs_all=None, group=None):
    """All-gather the per-rank observation tensor [n_local, width] into [world * n_local, width] (equal shards)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if obs_all is None:
        obs_all = torch.empty((world * obs_local.shape[0],) + tuple(obs_local.shape[1:]), dtype=obs_local.dtype, device=obs_local.device)
    dist.all_gather_into_tensor(obs_all, obs_local.contiguous(), group=group)
    return obs_all
