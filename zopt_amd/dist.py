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


def allgather_results(local: torch.Tensor, batch: int, group=None, out: torch.Tensor = None,
                      counts: Sequence[int] = None) -> torch.Tensor:
    """All-gather of per-rank result shards (leading axis = local batch) into the full (batch, ...) tensor on every
    rank.  Equal shards use one `all_gather_into_tensor` (a single RCCL all-gather over xGMI); ragged shards are
    padded to the largest shard first.  `out` (equal shards only): a caller-allocated (batch, ...) receive buffer, so that
    the one step that can fail on a single rank -- the allocation -- happens before any rank enters the collective.
    `counts`: the shard sizes rank by rank when they are not `shard_bounds(batch, world, r)` (e.g. a shard capped by memory)."""
    world = dist.get_world_size(group)
    if counts is None:
        counts = [hi - lo for lo, hi in (shard_bounds(batch, world, r) for r in range(world))]
    counts = [int(c) for c in counts]
    if len(counts) != world or sum(counts) != batch:
        raise ValueError(f"counts {counts} do not describe {batch} items on {world} ranks")
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


def shard_counts(local_items: int, group=None):
    """The shard sizes of every rank, rank by rank (one small object all-gather)."""
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, int(local_items), group=group)
    return out


# ------------------------------------------------------------------------------------------------------------------------
# Result tuples (iLQR / DDP / MPC): ONE collective for all fields
# ------------------------------------------------------------------------------------------------------------------------
def pack_results(fields: Sequence[torch.Tensor]) -> torch.Tensor:
    """Packs per-item result fields (each `(b, ...)`; bool / int fields are carried as 0.0 / 1.0 ...) into one contiguous
    `(b, width)` float64 buffer, fields side by side: the message of ONE all-gather instead of one per field (SURVEY 8e:
    configs[3] xTraj + uTraj + L + J + flag = 6 414 doubles per problem at T = 100, 52.5 MB per rank of 1 024 problems)."""
    b = int(fields[0].shape[0])
    flat = []
    for f in fields:
        if int(f.shape[0]) != b:
            raise ValueError("every field must have the same leading (batch) size")
        flat.append(f.reshape(b, -1).to(torch.float64))
    return torch.cat(flat, dim=1).contiguous()


def unpack_results(buf: torch.Tensor, shapes: Sequence[Sequence[int]], dtypes: Sequence[torch.dtype] = None):
    """Inverse of `pack_results`: `shapes` are the per-item shapes of the fields (without the batch axis)."""
    out, col = [], 0
    for i, s in enumerate(shapes):
        w = 1
        for d in s:
            w *= int(d)
        f = buf[:, col:col + w].reshape((buf.shape[0],) + tuple(int(d) for d in s))
        if dtypes is not None and dtypes[i] != torch.float64:
            f = f.to(dtypes[i])
        out.append(f)
        col += w
    if col != buf.shape[1]:
        raise ValueError(f"packed width {buf.shape[1]} does not match the shapes (sum {col})")
    return out


def allgather_tuple(fields: Sequence[torch.Tensor], batch: int, group=None, counts: Sequence[int] = None):
    """All-gather of a result tuple (e.g. `(xTraj, uTraj, L, J, converged)` of this rank's iLQR shard) as ONE collective over
    the packed buffer; returns the fields of the whole batch, in batch order, on every rank (ragged shards allowed)."""
    shapes = [tuple(f.shape[1:]) for f in fields]
    dtypes = [f.dtype for f in fields]
    full = allgather_results(pack_results(fields), batch, group=group, counts=counts)
    return unpack_results(full, shapes, dtypes)


# ------------------------------------------------------------------------------------------------------------------------
# Chunked gather overlapped with the sweep (configs[4]: 1.68 GB of gains per rank cost as much as the compute)
# ------------------------------------------------------------------------------------------------------------------------
class ChunkedGather:
    """All-gather of a rank's result shard in `nchunks` pieces, each issued as soon as the kernel that produced it has finished,
    so that the exchange of chunk c runs over xGMI while the sweep of chunk c + 1 computes.

    Zero-copy layout: the receive buffer is `(nchunks, world, chunk, ...)` -- every piece is the contiguous output of one
    `all_gather_into_tensor` -- and `global_view()` is the strided view `(world, nchunks, chunk, ...)`, i.e. `[r, c, i]` is item
    `r * local + c * chunk + i` of the whole batch (equal shards; `local` = nchunks * chunk items per rank).

    On a GPU the collectives are enqueued from a side stream that waits on the producing kernel's event (RCCL then runs them
    on its own stream behind it); on CPU tensors (gloo rehearsal) they are plain asynchronous collectives."""

    def __init__(self, local_shape: Sequence[int], nchunks: int, dtype, device, group=None):
        local = int(local_shape[0])
        if nchunks < 1 or local % nchunks:
            raise ValueError(f"{local} items per rank do not split into {nchunks} equal chunks")
        self.group, self.nchunks, self.chunk = group, nchunks, local // nchunks
        self.world = dist.get_world_size(group)
        self.item = tuple(int(d) for d in local_shape[1:])
        self.out = torch.empty((nchunks, self.world, self.chunk) + self.item, dtype=dtype, device=device)
        self.cuda = self.out.is_cuda
        self.side = torch.cuda.Stream(device=device) if self.cuda else None
        self.pending = []

    def chunk_slice(self, c: int) -> slice:
        return slice(c * self.chunk, (c + 1) * self.chunk)

    def _flat(self, c: int) -> torch.Tensor:
        """Receive buffer of piece c in the concatenated form (world * chunk, ...) every backend accepts."""
        return self.out[c].view((self.world * self.chunk,) + self.item)

    def issue(self, c: int, local_chunk: torch.Tensor, producer_stream=None):
        """Gathers chunk `c` (this rank's `(chunk, ...)` piece; contiguous).  `producer_stream`: the stream whose work so far
        produces it (default: the current stream)."""
        if tuple(local_chunk.shape) != (self.chunk,) + self.item or not local_chunk.is_contiguous():
            raise ValueError("chunk must be a contiguous (chunk, ...) piece of the local result")
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(producer_stream if producer_stream is not None else torch.cuda.current_stream(self.out.device))
            self.side.wait_event(ev)
            with torch.cuda.stream(self.side):
                w = dist.all_gather_into_tensor(self._flat(c), local_chunk, group=self.group, async_op=True)
        else:
            w = dist.all_gather_into_tensor(self._flat(c), local_chunk, group=self.group, async_op=True)
        self.pending.append(w)

    def wait(self):
        """Makes the current stream (GPU) / the caller (CPU) wait for every issued piece."""
        for w in self.pending:
            w.wait()
        self.pending = []
        if self.cuda:
            torch.cuda.current_stream(self.out.device).wait_stream(self.side)

    def global_view(self) -> torch.Tensor:
        """`(world, nchunks, chunk, ...)` strided view in batch order (no copy)."""
        return self.out.transpose(0, 1)
