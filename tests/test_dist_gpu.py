"""Sharded solves on REAL kernels, two rank processes on one MI355X (this pool gives a builder one GPU; RCCL refuses two ranks on one
device, so the collectives run on gloo through host memory -- the transport is not what is tested here).  Each rank solves its
`dist.shard_bounds` slice of a batch through the HIP path, the result tuple is gathered as ONE packed collective, and every rank must
hold exactly what a single process computes for the whole batch (the solvers are batch-composition independent, bit for bit).
The N-rank harness of bench.py is exercised the same way (`--share-gpu --backend gloo`: real sharded workloads of every config)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem(batch, T):
    from zopt_amd import models
    rng = np.random.default_rng(5)
    x0 = np.zeros((batch, 12))
    x0[:, 9:12] = rng.uniform(-10, 10, (batch, 3))
    ug = np.tile(models.QuadcopterEuler.uTrim, (batch, T, 1))
    return models.QuadcopterEuler(0.1), models.QuadraticCost(np.eye(12), np.eye(4), 10 * np.eye(12)), x0, ug


def _solve(x0, ug):
    from zopt_amd import ilqrUtils
    model, cost, _, _ = _problem(1, ug.shape[1])
    traj, L, J, conv = ilqrUtils.iterativeLqr(model, cost, cost, torch.as_tensor(x0, device="cuda"), torch.as_tensor(ug, device="cuda"),
                                              maxIter=30)
    return [t.cpu() for t in (traj.xTraj, traj.uTraj, L, J, conv)]


def _worker(rank, world, port, batch, T, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from zopt_amd import dist as zdist
        torch.cuda.set_device(0)                       # both ranks share the one GPU
        _, _, x0, ug = _problem(batch, T)
        lo, hi = zdist.shard_bounds(batch, world, rank)
        mine = _solve(x0[lo:hi], ug[lo:hi])            # HIP: zm_ilqr_solve_f64 on this rank's slice
        got = zdist.allgather_tuple(mine, batch)       # ONE collective over the packed tuple (host-staged: gloo)
        ref = _solve(x0, ug)                           # the whole batch in this one process
        same = [bool(torch.equal(a.contiguous().view(torch.uint8), b.contiguous().view(torch.uint8))) if a.dtype != torch.bool
                else bool(torch.equal(a, b)) for a, b in zip(got, ref)]
        q.put((rank, same, int(ref[4].sum()), [str(t.dtype) for t in got]))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_ilqr_equals_the_unsharded_solve_on_one_gpu():
    world, batch, T = 2, 37, 20                        # ragged shards: 19 + 18
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, batch, T, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, same, nconv, dtypes in res:
        assert all(same), (rank, same)                 # xTraj, uTraj, L, J, converged: bit for bit the unsharded solve
        assert nconv > 0 and dtypes[4] == "torch.bool"


def test_bench_harness_with_real_sharded_workloads_on_one_gpu():
    """`bench.py --gpus 2 --share-gpu --backend gloo`: the N-rank launcher, barrier / max-over-ranks timing and result gathers with the
    REAL HIP workloads of configs[3] (strong-sharded 64 problems); the gather's own check (every rank finds its shard unchanged in
    the gathered job) runs inside bench.py and fails the run otherwise."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--backend", "gloo", "--workload", "ilqr",
                          "--scaling", "strong", "--batch", "64", "--steps", "2", "--warmup", "1"], env=env, capture_output=True, text=True,
                         timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["rccl_ranks"] == 2 and res["scaling"] == "strong" and "stub" not in res
    assert res["allgather"]["items"] == 64 and res["allgather"]["fields"] == 5 and res["allgather"]["collectives"] == 1
    assert res["config"]["job_items"] == 64 and res["value"] > 0
