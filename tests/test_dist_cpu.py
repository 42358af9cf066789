"""World-size-2 rehearsal of the multi-GPU path on CPU (gloo): contiguous batch sharding + result all-gather.

The per-rank "solve" here is a stand-in elementwise function (the HIP kernels need a GPU); what is tested is that
sharding + gather reproduce the single-process result for equal and ragged shards, exactly as bench.py / a user
would drive one process per GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from zopt_amd import dist as zdist


def test_shard_bounds_cover_batch_exactly():
    for batch in (0, 1, 7, 8, 4096, 8191):
        for world in (1, 2, 3, 8):
            spans = [zdist.shard_bounds(batch, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == batch
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        zdist.shard_bounds(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, batch, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        full = torch.randn(batch, 5, 4, 12, dtype=torch.float64, generator=g)     # e.g. gains (batch, T, m, n)
        (mine,) = zdist.shard_batch([full])
        local_result = mine * 2.0 + 1.0                                            # stand-in for the per-rank solve
        gathered = zdist.allgather_results(local_result, batch)
        ok = torch.equal(gathered, full * 2.0 + 1.0)
        q.put((rank, bool(ok), tuple(gathered.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("batch", [8, 7])   # equal shards / ragged shards
def test_two_rank_shard_and_allgather(batch):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), res
    assert all(r[2] == (batch, 5, 4, 12) for r in res)
