#!/usr/bin/env python3
"""The workloads `bench.py --workload ...` can time, one class per BASELINE config, each shardable over ranks.

    lqr   configs[1]  discreteFiniteHorizonLqr, n=12 m=4 T=50 fp64 (the headline; total 4096 systems)
    mpc   configs[2]  lqrMpc quadcopter, N=30, demo tolerance (total 1024 instances)
    ilqr  configs[3]  iterativeLqr quadcopter, T=100 (total 8192 problems: "batch sharded 8 x MI355X")
    ddp   configs[3]  differentialDynamicProgramming, same problems with the DDP demo's R = 0.2 I
    n64   configs[4]  discreteFiniteHorizonLqr n=64 m=16 T=200 fp32 (total 16384 systems, MFMA P-update)

A workload owns this rank's shard only (contiguous split of the batch axis, nothing replicated but the < 3 kB of
shared MPC matrices, no data-path collective).  `--scaling weak` gives every rank `per_gpu` items, `--scaling strong`
gives rank r the slice `dist.shard_bounds(total, world, r)` of the config's total.  A "step" is one pass of the path over
the shard: one sweep launch (lqr, n64) or one whole batched solve through the C ABI (mpc, ilqr, ddp).  `results()` is what
the result all-gather carries.  `Stub*` twins run the same harness on CPU tensors (gloo rehearsal, never a measurement).
"""
from __future__ import annotations

import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

FP32_MATRIX_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
HBM_PEAK_GBS = 8000.0

# name -> (config index, total items of the config, items per GPU under weak scaling, default timed steps, default warm-up)
SPECS = {
    "lqr": {"config": 1, "total": 4096, "per_gpu": 4096, "steps": 300, "warmup": 50, "T": 50, "n": 12, "m": 4},
    "mpc": {"config": 2, "total": 1024, "per_gpu": 1024, "steps": 10, "warmup": 2, "T": 30, "n": 12, "m": 4},
    "ilqr": {"config": 3, "total": 8192, "per_gpu": 1024, "steps": 5, "warmup": 1, "T": 100, "n": 12, "m": 4},
    "ddp": {"config": 3, "total": 8192, "per_gpu": 1024, "steps": 3, "warmup": 1, "T": 100, "n": 12, "m": 4},
    "n64": {"config": 4, "total": 16384, "per_gpu": 2048, "steps": 10, "warmup": 3, "T": 200, "n": 64, "m": 16},
}


def local_range(spec_name, scaling, world, rank, batch=None):
    """[lo, hi) of the items this rank owns and the size of the whole job.  `batch` overrides the per-GPU count (weak) or the
    total (strong)."""
    from zopt_amd import dist as zdist
    spec = SPECS[spec_name]
    if scaling == "weak":
        per = int(batch) if batch else spec["per_gpu"]
        return rank * per, (rank + 1) * per, per * world
    total = int(batch) if batch else spec["total"]
    lo, hi = zdist.shard_bounds(total, world, rank)
    return lo, hi, total


class _HipBase:
    stub = False

    def _init_device(self, local_rank):
        import torch
        from zopt_amd import _lib
        self.torch, self._lib = torch, _lib
        self.dev = torch.device("cuda", local_rank)
        torch.cuda.set_device(self.dev)
        self.lib = _lib.lib()
        self.stream = torch.cuda.current_stream(self.dev)

    def device_name(self):
        p = self.torch.cuda.get_device_properties(self.dev)
        return f"{self.dev} {p.name} {getattr(p, 'gcnArchName', '')}".strip()

    def sync(self):
        self.torch.cuda.synchronize()

    def event(self):
        return self.torch.cuda.Event(enable_timing=True)

    def record(self, ev):
        ev.record(self.stream)      # HIP event on the stream the kernels are launched on

    def extras(self):
        return {}


def make_lti_inputs(batch, T, n, m, seed, device):
    import torch
    from tests import problems
    A1, B1, Q1, R1 = problems.random_lti_systems(batch, n, m, seed=seed)
    out = []
    for X in (A1, B1, Q1, R1):
        t = torch.as_tensor(X, device=device)
        out.append(t[:, None].expand(-1, T, -1, -1).contiguous())  # materialised (b,T,.,.) as the reference API takes
    return out


class HipLqrWorkload(_HipBase):
    """configs[1]: zm_lqr_backward_f64 through the C ABI on this rank's GPU."""
    name, unit, dtype = "lqr", "horizon-steps/s", "f64"

    def __init__(self, lo, hi, total, rank, local_rank, T=50, n=12, m=4):
        self._init_device(local_rank)
        torch = self.torch
        b = hi - lo
        self.shape = (b, T, n, m)
        # two distinct resident input sets, alternated per step, so that no step can be served from the 256 MiB L3
        self.sets = [make_lti_inputs(b, T, n, m, seed=2 * rank + i, device=self.dev) for i in range(2)]
        self.L = torch.empty((b, T, m, n), dtype=torch.float64, device=self.dev)
        self.last_set = 0
        self.units_per_step = b * T

    def step(self, i):
        b, T, n, m = self.shape
        A, B, Q, R = self.sets[i & 1]
        self.last_set = i & 1
        rc = self.lib.zm_lqr_backward_f64(A.data_ptr(), B.data_ptr(), Q.data_ptr(), R.data_ptr(), self.L.data_ptr(), b, T, n, m,
                                          ctypes.c_void_p(self.stream.cuda_stream))
        self._lib.check(rc, "zm_lqr_backward_f64")

    def results(self):
        return [self.L]

    def result(self):
        return self.L


class HipIlqrWorkload(_HipBase):
    """configs[3]: the whole iterativeLqr / differentialDynamicProgramming solve of this rank's problems (one C-ABI call,
    zm_ilqr_solve_f64, behind zopt_amd.ilqrUtils)."""
    unit, dtype = "problems/s", "f64"

    def __init__(self, lo, hi, total, rank, local_rank, ddp=False, T=100):
        self._init_device(local_rank)
        from tools import secondary_bench
        from zopt_amd import ilqrUtils
        self.name = "ddp" if ddp else "ilqr"
        self.ddp, self.T = ddp, T
        # the problem set is defined for the WHOLE job (seed 2); a rank takes its slice -- under weak scaling the job
        # grows with the world, and every rank still sees the same distribution of starts
        self.model, self.cost, x0, ug = secondary_bench.config3_problem(max(total, hi), T, ddp)
        self.x0 = self.torch.as_tensor(x0[lo:hi], device=self.dev)
        self.ug = self.torch.as_tensor(ug[lo:hi], device=self.dev)
        self.solve = ilqrUtils.differentialDynamicProgramming if ddp else ilqrUtils.iterativeLqr
        self.units_per_step = hi - lo
        self.out = None

    def step(self, i):
        self.out = self.solve(self.model, self.cost, self.cost, self.x0, self.ug)

    def results(self):
        traj, L, J, conv = self.out
        return [traj.xTraj, traj.uTraj, L, J, conv]

    def extras(self):
        traj, L, J, conv = self.out
        fin = self.torch.isfinite(J)
        return {"converged_frac_rank0": float(conv.double().mean().item()),
                "J_nonfinite_frac_rank0": float((~fin).double().mean().item())}


class HipMpcWorkload(_HipBase):
    """configs[2]: one cold-started batched lqrMpc solve of this rank's instances at the demo tolerance."""
    name, unit, dtype = "mpc", "instance-solves/s", "f64"

    def __init__(self, lo, hi, total, rank, local_rank, N=30, eps=1e-2):
        self._init_device(local_rank)
        from tools import secondary_bench
        self.prob, x_ub = secondary_bench.mpc_problem(N)
        self.x0 = self.torch.as_tensor(secondary_bench.mpc_x0(max(total, hi), x_ub)[lo:hi], device=self.dev)
        self.eps, self.N = eps, N
        self.units_per_step = hi - lo
        self.out = None
        self.prob.solve(self.x0[:64], eps_abs=eps, eps_rel=eps)      # Riccati tables (setup is not part of a solve)

    def step(self, i):
        self.out = self.prob.solve(self.x0, solver="OSQP", eps_abs=self.eps, eps_rel=self.eps, max_iter=100000, warm_start=False)

    def results(self):
        u0, traj, status = self.out
        code = self.torch.as_tensor((np.asarray(status) == "optimal").astype(np.float64), device=self.dev)
        return [u0, traj.xTraj, traj.uTraj, code]

    def extras(self):
        its = self.prob.last_iterations
        return {"optimal_frac_rank0": float(np.mean(np.asarray(self.out[2]) == "optimal")),
                "admm_iters_mean_rank0": float(its.mean()), "admm_iters_max_rank0": int(its.max()),
                "parity": "unpinned (reference arithmetic is OSQP's, absent here)"}


def tiled_mfma_per_step(n):
    nt = (n + 15) // 16
    return 4 * nt * nt + 4 * nt * (nt + 1) + 4 * nt ** 3 + 4 * nt * nt + 4 * nt + 4 * nt * nt + 4 * nt * nt + 4 * nt ** 3


class HipN64Workload(_HipBase):
    """configs[4]: zm_lqr_backward_f32 at n=64, m=16, T=200 on this rank's shard, inputs generated on the device.  The shard is
    capped by what fits the GPU (`mem_frac` of the free memory); the sweep can run in `nchunks` launches over contiguous pieces
    of the shard so that a chunked result gather can overlap it (zopt_amd.dist.ChunkedGather)."""
    name, unit, dtype = "n64", "horizon-steps/s", "f32"

    def __init__(self, lo, hi, total, rank, local_rank, T=200, n=64, m=16, nchunks=None, mem_frac=0.6):
        self._init_device(local_rank)
        torch = self.torch
        want = hi - lo
        bytes_per_traj = 4 * T * (2 * n * n + 2 * n * m + m * m)      # inputs + gains; the gathered copy is accounted below
        free, _ = torch.cuda.mem_get_info(self.dev)
        cap = int(mem_frac * free // bytes_per_traj)
        b = min(want, cap)
        if nchunks is None:
            # a chunk must still fill the chip: one trajectory per wave, 1024 SIMDs -- 2048 trajectories in four launches of 512 took
            # 12.1 ms against 6.1 ms in one (measured); so at least 1024 trajectories per chunk
            nchunks = max(1, min(8, b // 1024))
        b -= b % nchunks
        self.capped = b < want
        self.wanted = want
        self.nchunks = nchunks
        g = torch.Generator(device=self.dev).manual_seed(3 + rank)
        rn = lambda *s: torch.randn(*s, device=self.dev, dtype=torch.float32, generator=g)   # noqa: E731
        A1 = rn(b, n, n) * (0.9 / n ** 0.5)
        B1 = rn(b, n, m)
        M, N = rn(b, n, n), rn(b, m, m)
        eye = lambda k: torch.eye(k, device=self.dev, dtype=torch.float32)   # noqa: E731
        Q1 = M @ M.transpose(-1, -2) / n + eye(n)
        R1 = N @ N.transpose(-1, -2) / m + eye(m)
        del M, N
        self.A, self.B, self.Q, self.R = (X[:, None].expand(b, T, *X.shape[1:]).contiguous() for X in (A1, B1, Q1, R1))
        self.L = torch.empty((b, T, m, n), device=self.dev, dtype=torch.float32)
        self.shape = (b, T, n, m)
        self.units_per_step = b * T

    def _launch(self, lo, hi):
        b, T, n, m = self.shape
        rc = self.lib.zm_lqr_backward_f32(self.A[lo:hi].data_ptr(), self.B[lo:hi].data_ptr(), self.Q[lo:hi].data_ptr(),
                                          self.R[lo:hi].data_ptr(), self.L[lo:hi].data_ptr(), hi - lo, T, n, m,
                                          ctypes.c_void_p(self.stream.cuda_stream))
        self._lib.check(rc, "zm_lqr_backward_f32")

    def step(self, i):
        self._launch(0, self.shape[0])

    def step_chunk(self, c):
        ch = self.shape[0] // self.nchunks
        self._launch(c * ch, (c + 1) * ch)
        return self.L[c * ch:(c + 1) * ch]

    def results(self):
        return [self.L]

    def extras(self):
        return {"shard_capped_by_memory": self.capped, "shard_wanted": self.wanted, "shard_run": self.shape[0]}


# ------------------------------------------------------------------------------------------------------------------------
class StubWorkload:
    """CPU stand-in with the result tuple of the named workload (tiny shapes): the gloo rehearsal of the multi-rank harness."""
    stub = True

    FIELDS = {   # per-item result shapes (T, n, m are the stub's small sizes)
        "lqr": lambda T, n, m: [(T, m, n)],
        "n64": lambda T, n, m: [(T, m, n)],
        "ilqr": lambda T, n, m: [(T + 1, n), (T, m), (T, m, n), (), ()],
        "ddp": lambda T, n, m: [(T + 1, n), (T, m), (T, m, n), (), ()],
        "mpc": lambda T, n, m: [(m,), (T + 1, n), (T, m), ()],
    }

    def __init__(self, name, lo, hi, total, rank, T=5, n=12, m=4, fail_rank=-1, nchunks=2):
        import torch
        self.torch, self.name, self.rank, self.fail_rank = torch, name, rank, fail_rank
        self.dev = torch.device("cpu")
        self.unit = {"lqr": "horizon-steps/s", "n64": "horizon-steps/s", "mpc": "instance-solves/s"}.get(name, "problems/s")
        self.dtype = "f32" if name == "n64" else "f64"
        b = hi - lo
        if name == "n64":
            b -= b % nchunks
        self.nchunks = nchunks
        self.lo, self.b = lo, b
        self.shape = (b, T, n, m)
        self.units_per_step = b * T if name in ("lqr", "n64") else b
        dt = torch.float32 if name == "n64" else torch.float64
        # item g of the job gets values that depend on g only: a gathered result can be checked against the unsharded one
        ids = torch.arange(lo, lo + b, dtype=torch.float64)
        self.src = []
        for k, s in enumerate(self.FIELDS[name](T, n, m)):
            w = int(np.prod(s)) if s else 1
            f = (ids[:, None] * 1000.0 + torch.arange(w, dtype=torch.float64)[None, :] + 0.25 * k).reshape((b,) + tuple(s))
            self.src.append(f.to(dt))
        if name in ("ilqr", "ddp"):
            self.src[-1] = (ids % 2 == 0)                         # the `converged` flags are bool in the real tuple
        self.out = [torch.empty_like(f) for f in self.src]
        self.L = self.out[0]

    def device_name(self):
        return f"cpu (stub) pid {os.getpid()}"

    def step(self, i):
        if self.fail_rank == self.rank:
            raise RuntimeError("stub failure requested (--stub-fail-rank)")
        for o, s in zip(self.out, self.src):
            o.copy_(s)

    def step_chunk(self, c):
        ch = self.b // self.nchunks
        self.out[0][c * ch:(c + 1) * ch].copy_(self.src[0][c * ch:(c + 1) * ch])
        return self.out[0][c * ch:(c + 1) * ch]

    def sync(self):
        pass

    def event(self):
        return [0.0]

    def record(self, ev):
        ev[0] = time.perf_counter()

    def results(self):
        return list(self.out)

    def result(self):
        return self.out[0]

    def extras(self):
        return {}

    @staticmethod
    def expected(name, total, T=5, n=12, m=4):
        """The unsharded result tuple of a `total`-item stub job (what a correct gather must reproduce on every rank)."""
        w = StubWorkload(name, 0, total, total, rank=0, T=T, n=n, m=m, nchunks=1)
        return w.src


def make(name, scaling, world, rank, local_rank, batch=None, stub=False, stub_fail_rank=-1, T=None, n=None, m=None):
    lo, hi, total = local_range(name, scaling, world, rank, batch)
    if stub:
        return StubWorkload(name, lo, hi, total, rank, T=T or 5, n=n or 12, m=m or 4, fail_rank=stub_fail_rank), total
    spec = SPECS[name]
    if name == "lqr":
        return HipLqrWorkload(lo, hi, total, rank, local_rank, T or spec["T"], n or spec["n"], m or spec["m"]), total
    if name in ("ilqr", "ddp"):
        return HipIlqrWorkload(lo, hi, total, rank, local_rank, ddp=(name == "ddp"), T=T or spec["T"]), total
    if name == "mpc":
        return HipMpcWorkload(lo, hi, total, rank, local_rank, N=T or spec["T"]), total
    if name == "n64":
        return HipN64Workload(lo, hi, total, rank, local_rank, T or spec["T"], n or spec["n"], m or spec["m"]), total
    raise ValueError(name)
