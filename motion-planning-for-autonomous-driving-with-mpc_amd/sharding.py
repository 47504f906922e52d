"""
Multi-GPU sharding of independent MPC instances.

The instances of a batch have no coupling term anywhere in the NLP (each ego has its own x0, p, obstacle), so the
path shards embarrassingly: rank r of G solves the contiguous block of rows [lo_r, hi_r) with its own handle on
its own GPU and there is NO collective on the data path.  The only exchange is the final gather of the result
rows (`all_gather_results`, RCCL all-gather over xGMI on GPUs, gloo in the CPU tests) and the timing reduction
of bench.py (max over ranks).
"""
from __future__ import annotations

import numpy as np


def shard_bounds(B: int, rank: int, world: int):
    """contiguous, balanced partition of range(B): first (B % world) ranks get one extra row"""
    base, rem = divmod(int(B), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def shard_rows(arr, rank: int, world: int):
    lo, hi = shard_bounds(arr.shape[0], rank, world)
    return arr[lo:hi]


def all_gather_results(x_local, status_local, iters_local, B: int, group=None):
    """final gather of [B_r, n_w] float64 rows + int32 status/iters from every rank; returns full-batch tensors on
    every rank.  Ragged shards are padded to the largest shard for the collective and trimmed afterwards."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    sizes = [shard_bounds(B, r, world)[1] - shard_bounds(B, r, world)[0] for r in range(world)]
    mx = max(sizes)

    def gather(t):
        t = t if torch.is_tensor(t) else torch.as_tensor(np.ascontiguousarray(t))
        pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        outs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(outs, pad, group=group)
        return torch.cat([o[:n] for o, n in zip(outs, sizes)], dim=0)

    return gather(x_local), gather(status_local), gather(iters_local)


def max_over_ranks(value: float, device=None, group=None) -> float:
    """bench.py timing contract: the slowest rank defines the step time"""
    import torch
    import torch.distributed as dist

    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def sum_over_ranks(value: float, device=None, group=None) -> float:
    import torch
    import torch.distributed as dist

    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t.item())
