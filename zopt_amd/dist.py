"""Batch sharding across the GPUs of one node.

Trajectories / MPC instances / iLQR problems are independent (no cross-trajectory term anywhere in
lqrUtils.py:167-172, ilqrUtils.py:305-322, mpcUtils.py:76-81), so the batch axis is split contiguously over
ranks -- one process per GPU, `torch.distributed` (backend "nccl" is RCCL on ROCm; "gloo" for CPU rehearsals) --
and there is NO collective on the data path.  The only exchange is an optional all-gather of results.
"""
from __future__ import annotations

from typing import Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(batch: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous split of `batch` items: rank r owns [lo, hi).  The first `batch % world_size` ranks get one extra."""
    if world_size < 1 or not (0 <= rank < world_size) or batch < 0:
        raise ValueError("bad (batch, world_size, rank)")
    base, rem = divmod(batch, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(arrays: Sequence, world_size: int = None, rank: int = None):
    """Slices every array's leading (batch) axis to this rank's shard."""
    world_size = dist.get_world_size() if world_size is None else world_size
    rank = dist.get_rank() if rank is None else rank
    lo, hi = shard_bounds(int(arrays[0].shape[0]), world_size, rank)
    return [a[lo:hi] for a in arrays]


def allgather_results(local: torch.Tensor, batch: int, group=None, out: torch.Tensor = None) -> torch.Tensor:
    """All-gather of per-rank result shards (leading axis = local batch) into the full (batch, ...) tensor on every
    rank.  Equal shards use one `all_gather_into_tensor` (a single RCCL all-gather over xGMI); ragged shards are
    padded to the largest shard first.  `out` (equal shards only): a caller-allocated (batch, ...) receive buffer, so that
    the one step that can fail on a single rank -- the allocation -- happens before any rank enters the collective."""
    world = dist.get_world_size(group)
    sizes = [shard_bounds(batch, world, r) for r in range(world)]
    counts = [hi - lo for lo, hi in sizes]
    local = local.contiguous()
    if local.shape[0] != counts[dist.get_rank(group)]:
        raise ValueError(f"local shard has {local.shape[0]} items, expected {counts[dist.get_rank(group)]}")
    mx = max(counts)
    if min(counts) == mx:
        shape = (batch,) + tuple(local.shape[1:])
        if out is None:
            out = torch.empty(shape, dtype=local.dtype, device=local.device)
        elif tuple(out.shape) != shape or out.dtype != local.dtype or not out.is_contiguous():
            raise ValueError(f"out must be a contiguous {shape} tensor of {local.dtype}")
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    if out is not None:
        raise ValueError("out= is only supported for equal shards")
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    buf = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    return torch.cat([buf[r * mx: r * mx + counts[r]] for r in range(world)], dim=0)
