#!/usr/bin/env python3
"""bench.py -- batched finite-horizon LQR backward Riccati solves on MI355X (BASELINE.json configs[1]).

One "step" = one pass of the hot path over one batch: `batch` independent trajectories x T horizon steps of
`discreteFiniteHorizonLqr` (n=12, m=4, T=50, fp64), inputs already resident in HBM.  Weak scaling: every rank
(one process per GPU) owns its own `batch` trajectories; there is no data-path collective.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (see README / DESIGN.md "Measurement").

Defaults: 300 timed steps after 50 warm-up steps (56 ms of GPU time).  A step is 0.15 ms, and this GPU needs tens of
milliseconds under load to leave its idle clock state (measured: 10 warm-up + 50 timed steps -> 174 us per launch,
50 + 300 -> 147 us), so short runs mostly time the clock ramp.  To make the line independent of the K / W a caller
picks, an untimed device warm-up (--prewarm-ms, default 60 ms of launches) precedes the W warm-up steps; the timed
region is still exactly K steps between barrier + synchronize.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def bytes_per_step(n, m, elt=8):
    """Algorithmic HBM bytes per horizon step: read A_k, B_k, Q_k, R_k once, write L_k once."""
    return elt * (2 * n * n + 2 * n * m + m * m)


def make_inputs(batch, T, n, m, seed, device):
    import torch
    from tests import problems
    A1, B1, Q1, R1 = problems.random_lti_systems(batch, n, m, seed=seed)
    out = []
    for X in (A1, B1, Q1, R1):
        t = torch.as_tensor(X, device=device)
        out.append(t[:, None].expand(-1, T, -1, -1).contiguous())  # materialised (b,T,.,.) as the reference API takes
    return out


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota (GPU boxes give a
    one-GPU job a share of the host, e.g. 16 of 256 hardware threads)."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    return n


def cpu_baseline(batch, T, n, m, target_seconds=10.0):
    """Times the plain-C oracle (a port of lqrUtils.py:167-172) on all host cores over the same workload."""
    from oracle import c_oracle
    from tests import problems
    A1, B1, Q1, R1 = problems.random_lti_systems(batch, n, m, seed=0)
    A, B, Q, R = problems.tile_over_horizon(A1, B1, Q1, R1, T)
    cores = min(c_oracle.num_threads(), usable_cores())
    c_oracle.lqr_backward(A[:64], B[:64], Q[:64], R[:64], nthreads=cores)  # warm-up
    c_oracle.lqr_backward(A, B, Q, R, nthreads=cores)      # warm-up of the full-size call (thread pool, page faults)
    reps, elapsed = 0, 0.0
    t0 = time.perf_counter()
    while True:
        c_oracle.lqr_backward(A, B, Q, R, nthreads=cores)
        reps += 1
        elapsed = time.perf_counter() - t0
        # bounded sample: ~target_seconds of CPU work summed over cores, but at least 1 s of wall time
        if (elapsed * cores >= target_seconds and elapsed >= 1.0) or elapsed >= 20.0:
            break
    return {"value": batch * T * reps / elapsed, "unit": "horizon-steps/s", "cores": cores, "kind": "port",
            "sample": f"full workload ({batch} trajectories x T={T}, n={n}, m={m}, fp64) x {reps} reps, "
                      f"oracle/riccati_oracle.c with OpenMP over the batch"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=4096, help="trajectories per GPU (weak scaling)")
    ap.add_argument("--T", type=int, default=50)
    ap.add_argument("--n", type=int, default=12)
    ap.add_argument("--m", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prewarm-ms", type=float, default=60.0,
                    help="untimed device warm-up before the W warm-up steps: launches until this much time has passed, so that a "
                         "run with small --warmup/--steps is not a measurement of the clock ramp (0 disables)")
    ap.add_argument("--gather", action="store_true", help="(default for N > 1) also time an RCCL all-gather of the gains, reported separately")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the all-gather of the results after the timed region")
    args = ap.parse_args()

    import ctypes
    import torch
    import torch.distributed as dist
    from zopt_amd import _lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    batch, T, n, m = args.batch, args.T, args.n, args.m
    lib = _lib.lib()
    # two distinct resident input sets, alternated per step, so that no step can be served from the 256 MiB L3
    sets = [make_inputs(batch, T, n, m, seed=2 * rank + i, device=dev) for i in range(2)]
    L = torch.empty((batch, T, m, n), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev)

    def step(i):
        A, B, Q, R = sets[i & 1]
        rc = lib.zm_lqr_backward_f64(A.data_ptr(), B.data_ptr(), Q.data_ptr(), R.data_ptr(), L.data_ptr(), batch, T, n,
                                     m, ctypes.c_void_p(stream.cuda_stream))
        _lib.check(rc, "zm_lqr_backward_f64")

    def barrier():
        if world > 1:
            dist.barrier()

    if args.prewarm_ms > 0:      # leave the idle clock state (docstring); not part of the W warm-up steps or of the timed region
        p0 = time.perf_counter()
        while (time.perf_counter() - p0) * 1e3 < args.prewarm_ms:
            for i in range(16):
                step(i)
            torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        evs[i][0].record(stream)
        step(i)
        evs[i][1].record(stream)
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))  # HIP events on the launch stream

    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # After (outside) the timed region: the one exchange the path has -- every rank collects all gains (north star: "RCCL
    # all-gather of results over xGMI").  Reported separately; it is never part of `value`.
    gather_ms, gather_err = None, None
    if world > 1 and not args.no_gather:
        try:
            from zopt_amd import dist as zdist
            zdist.allgather_results(L, world * batch)      # warm-up (RCCL communicator setup)
            torch.cuda.synchronize()
            barrier()
            g0 = time.perf_counter()
            full = zdist.allgather_results(L, world * batch)
            torch.cuda.synchronize()
            gather_ms = (time.perf_counter() - g0) * 1e3
            assert full.shape[0] == world * batch
            del full
        except Exception as e:  # noqa: BLE001 -- the throughput line must survive a failed optional exchange
            gather_err = f"{type(e).__name__}: {e}"

    if rank == 0:
        steps_per_launch = batch * T
        bps = bytes_per_step(n, m)
        achieved = bps * steps_per_launch / (kern_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath) and (batch, T, n, m) == (4096, 50, 12, 4):
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        res = {
            "metric": "LQR horizon-steps/sec (batch x T) at n=12,m=4,T=50",
            "value": world * steps_per_launch * args.steps / elapsed,
            "unit": "horizon-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"discreteFiniteHorizonLqr: {batch} random LTI systems per GPU, n={n} m={m} T={T} "
                                   f"fp64, A/B/Q/R materialised (b,T,.,.), BASELINE configs[1]",
                       "batch_per_gpu": batch, "T": T, "n": n, "m": m, "parallelism": f"batch-sharded x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "lqr_backward_dma_f64<12,4,3>" if (n, m) == (12, 4) else "lqr_backward", "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": bps * steps_per_launch},
        }
        if (n, m) == (12, 4):
            # Informational: the fp64 matrix pipe is the resource this kernel actually saturates.  Per horizon step the wave
            # issues 9 v_mfma_f64_16x16x4 (2048 flop) + 3 v_mfma_f64_4x4x4_4b (512 flop); the sustained fp64 MFMA rate of the
            # chip, measured at steady state (profiles/r01_ubench_mfma_f64_steady.txt), is 47.2 TFLOP/s (nominal 78.6).
            issued = (9 * 2048 + 3 * 512) * steps_per_launch / (kern_ms * 1e-3) / 1e12
            res["fp64_matrix_pipe"] = {"issued_mfma_tflops": issued, "sustained_peak_tflops": 47.2, "nominal_peak_tflops": 78.6,
                                       "frac_of_sustained": issued / 47.2}
        if gather_ms is not None:
            nbytes = world * batch * T * m * n * 8
            res["allgather"] = {"ms": gather_ms, "bytes_per_rank": nbytes, "GBps_per_rank": nbytes / (gather_ms * 1e-3) / 1e9}
        if gather_err is not None:
            res["allgather"] = {"error": gather_err}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(batch, T, n, m)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
