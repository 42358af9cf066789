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


# ------------------------------------------------------------------------------------------------------------------------
# REAL solver outputs through the sharded path: committed fixture inputs, the oracle as the per-rank solver (the HIP kernels
# need a GPU; tests/test_*_gpu.py hold them to the same oracle), gather, compare with the unsharded solve.
# ------------------------------------------------------------------------------------------------------------------------
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lqr_config2_first8.npz")


def _ilqr_problems(batch):
    rng = np.random.default_rng(11)
    A = np.array([[1.0, 0.1], [0.0, 1.0]])
    B = np.array([[0.005], [0.1]])
    x0 = rng.uniform(-2, 2, (batch, 2))
    return A, B, x0


def _solve_ilqr_shard(x0s):
    """(xTraj, uTraj, L, J, converged) of each start in `x0s`, by the oracle's iLQR loop (reference ilqrUtils.py:290-327)."""
    from oracle import zopt_oracle as zo
    A, B, _ = _ilqr_problems(1)
    T = 6
    outs = [zo.iterativeLqr(lambda x, u: A @ x + B @ u, np.eye(2), np.eye(1), 10 * np.eye(2), x0, np.zeros((T, 1))) for x0 in x0s]
    return (torch.as_tensor(np.stack([o[0].xTraj for o in outs])), torch.as_tensor(np.stack([o[0].uTraj for o in outs])),
            torch.as_tensor(np.stack([o[1] for o in outs])), torch.as_tensor(np.array([o[2] for o in outs])),
            torch.as_tensor(np.array([o[3] for o in outs])))


def _real_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import zopt_oracle as zo
        from tests import problems
        g = np.load(GOLD)
        T = int(g["T"])
        A, B, Q, R = problems.tile_over_horizon(g["A"], g["B"], g["Q"], g["R"], T)          # the 8 committed configs[1] systems
        batch = A.shape[0]
        # (1) discreteFiniteHorizonLqr: shard -> solve -> ONE all-gather == the unsharded solve == the committed gains
        mine = zdist.shard_batch([A, B, Q, R])
        L_local = torch.as_tensor(zo.discreteFiniteHorizonLqr(*mine, T))
        L_all = zdist.allgather_results(L_local, batch)
        ok_lqr = np.array_equal(L_all.numpy(), zo.discreteFiniteHorizonLqr(A, B, Q, R, T)) and \
            np.allclose(L_all.numpy(), g["L"], rtol=0, atol=1e-12)
        # (2) iterativeLqr: the result TUPLE (xTraj, uTraj, L, J, converged) as one collective over the packed buffer
        nprob = 7                                                                           # 4 + 3 on two ranks, 3 + 2 + 2 on three
        _, _, x0 = _ilqr_problems(nprob)
        lo, hi = zdist.shard_bounds(nprob, world, rank)
        fields = _solve_ilqr_shard(x0[lo:hi])
        got = zdist.allgather_tuple(fields, nprob)
        ref = _solve_ilqr_shard(x0)
        ok_ilqr = all(torch.equal(a, b) and a.dtype == b.dtype for a, b in zip(got, ref)) and got[4].dtype == torch.bool
        # (3) the chunked gather: two pieces per rank, zero-copy layout, strided global view == the unsharded gains
        if batch % (2 * world) == 0:
            cg = zdist.ChunkedGather(tuple(L_local.shape), 2, L_local.dtype, L_local.device)
            for c in range(2):
                cg.issue(c, L_local[cg.chunk_slice(c)])
            cg.wait()
            ok_chunk = torch.equal(cg.global_view().reshape(L_all.shape), L_all)
        else:
            ok_chunk = True
        q.put((rank, bool(ok_lqr), bool(ok_ilqr), bool(ok_chunk)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_real_solves_equal_the_unsharded_ones(world):
    """Sharding + gather of REAL solver outputs: the committed configs[1] systems through discreteFiniteHorizonLqr (8 systems: equal
    shards on two ranks, 3 + 3 + 2 on three) and 7 iLQR problems' result tuples (always ragged), solved per shard by the oracle and
    gathered, must equal the unsharded solve bit for bit on every rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_real_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    assert all(r[1] and r[2] and r[3] for r in res), res


def test_pack_unpack_round_trip():
    g = torch.Generator().manual_seed(1)
    fields = [torch.randn(5, 7, 3, dtype=torch.float64, generator=g), torch.randn(5, dtype=torch.float64, generator=g),
              torch.tensor([True, False, True, True, False]), torch.arange(5, dtype=torch.int32)]
    buf = zdist.pack_results(fields)
    assert buf.shape == (5, 21 + 1 + 1 + 1) and buf.dtype == torch.float64
    back = zdist.unpack_results(buf, [tuple(f.shape[1:]) for f in fields], [f.dtype for f in fields])
    assert all(torch.equal(a, b) and a.dtype == b.dtype for a, b in zip(fields, back))
    with pytest.raises(ValueError):
        zdist.unpack_results(buf, [(7, 3)])
