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


# ---- the gather payload of a step: result rows with their status and iteration count in ONE block ----------------
def packed_width(n_w: int) -> int:
    """columns of a packed block: n_w result values, then status and iteration count (int32 values, exact as float64)"""
    return int(n_w) + 2


def pack_rows(x, status, iters, out=None):
    """[B, n_w] float64 rows + int32 status / iters -> one [B, n_w + 2] float64 block (torch tensors, any device; `out` is reused)"""
    import torch

    B, n_w = x.shape
    if out is None:
        out = torch.empty((B, n_w + 2), dtype=torch.float64, device=x.device)
    out[:, :n_w].copy_(x)
    out[:, n_w].copy_(status)
    out[:, n_w + 1].copy_(iters)
    return out


def unpack_rows(block, n_w: int):
    """inverse of pack_rows: (x [B, n_w] float64, status [B] int32, iters [B] int32)"""
    import torch

    return block[:, :n_w], block[:, n_w].round().to(torch.int32), block[:, n_w + 1].round().to(torch.int32)


def gather_packed(block, out, group=None, async_op=False):
    """the one collective of a step: every rank's packed block -> `out` [world * B, n_w + 2] on every rank (all_gather_into_tensor:
    RCCL on GPUs, gloo in the CPU tests).  Returns the work handle when async_op, else None."""
    import torch.distributed as dist

    return dist.all_gather_into_tensor(out, block, group=group, async_op=async_op)


def solve_stats_over_ranks(status, iters, device=None, group=None):
    """converged fraction / mean / max of the iteration counts over ALL ranks' rows (each rank passes its own status and iters)"""
    import torch
    import torch.distributed as dist

    st = torch.as_tensor(status)
    it = torch.as_tensor(iters)
    sums = torch.tensor([float((st == 1).sum()), float(st.numel()), float(it.sum())], dtype=torch.float64, device=device)
    mx = torch.tensor([float(it.max()) if it.numel() else 0.0], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    n = max(1.0, float(sums[1]))
    return dict(converged_frac=float(sums[0]) / n, mean_iters=float(sums[2]) / n, max_iters=int(mx.item()), rows=int(sums[1]))
