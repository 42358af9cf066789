#!/usr/bin/env python3
"""bench.py -- batched finite-horizon LQR backward Riccati solves on MI355X (BASELINE.json configs[1]).

One "step" = one pass of the hot path over one batch: `batch` independent trajectories x T horizon steps of
`discreteFiniteHorizonLqr` (n=12, m=4, T=50, fp64), inputs already resident in HBM.  Weak scaling: every rank
(one process per GPU) owns its own `batch` trajectories; there is no data-path collective.

    python bench.py --gpus N --steps K --warmup W          # N > 1: starts its N ranks itself (see below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W              # ranks started by the caller

Rank 0 prints ONE JSON line (see README / DESIGN.md "Measurement").

Multi-GPU.  With --gpus N > 1 and no WORLD_SIZE in the environment this process is only a LAUNCHER: before any GPU call
(it never imports torch) it starts N fresh rank processes of this same file with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR=127.0.0.1 / MASTER_PORT set, relays rank 0's JSON line and exits non-zero if any rank fails or the run times
out.  With WORLD_SIZE set (torch.distributed.run, or our own launcher) it is one rank.  The line carries `rccl_ranks`
(= dist.get_world_size() observed after init) and `devices` (one entry per rank) so that the N ranks can be seen to exist.

Defaults: 300 timed steps after 50 warm-up steps (56 ms of GPU time).  A step is 0.15 ms, and this GPU needs tens of
milliseconds under load to leave its idle clock state (measured: 10 warm-up + 50 timed steps -> 174 us per launch,
50 + 300 -> 147 us), so short runs mostly time the clock ramp.  To make the line independent of the K / W a caller
picks, an untimed device warm-up (--prewarm-ms, default 60 ms of launches) precedes the W warm-up steps; the timed
region is still exactly K steps between barrier + synchronize.

After -- and outside -- the timed region, rank 0 at N = 1 also reports: `parity_rel_err` (one trajectory of the timed kernel's
output against the C oracle), `cpu_baseline` (C port, all granted cores) and `cpu_baseline_numpy` (single-process NumPy
batched restatement), and `secondary` (bounded runs of BASELINE configs[2], [3], [4]; tools/secondary_bench.py).

--backend gloo --stub-step replaces the HIP launch by a CPU tensor op: the CPU rehearsal of the launcher / barrier /
max-over-ranks / all-gather code path (tests/test_bench_launcher_cpu.py).  Its line is marked "stub": true and is not a
measurement.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
K1_SOURCES = ("zopt_amd/csrc/lqr_backward_dma.hip", "zopt_amd/csrc/dma_ring.h", "zopt_amd/csrc/tile16_f64.h")


def bytes_per_step(n, m, elt=8):
    """Algorithmic HBM bytes per horizon step: read A_k, B_k, Q_k, R_k once, write L_k once."""
    return elt * (2 * n * n + 2 * n * m + m * m)


def k1_source_sha():
    """Fingerprint of the headline kernel's sources: a PMC traffic figure is only quoted for the kernel it was measured on."""
    h = hashlib.sha256()
    for rel in K1_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


# ------------------------------------------------------------------------------------------------------------------------
# launcher (parent process; no torch, no GPU)
# ------------------------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv, timeout_s):
    """Starts n rank processes of this file, relays rank 0's stdout, returns the exit code (0 only if every rank exited 0)."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on these hosts (RCCL needs it)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    deadline = time.monotonic() + timeout_s
    rc, out0 = 0, b""
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if r == 0:
                    out0 = procs[0].stdout.read()
                if code != 0:
                    rc = code if code > 0 else 1
                    print(f"bench.py launcher: rank {r} exited with code {code}", file=sys.stderr)
            if rc != 0 or not pending:
                break
            if time.monotonic() > deadline:
                rc = 124
                print(f"bench.py launcher: timeout after {timeout_s:.0f} s", file=sys.stderr)
                break
            time.sleep(0.05)
    finally:
        for p in procs:                 # exact PIDs we started; never a pattern
            if p.poll() is None:
                p.kill()
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                pass
    if not out0 and procs[0].stdout is not None:
        try:
            out0 = procs[0].stdout.read()
        except Exception:  # noqa: BLE001
            out0 = b""
    # rank 0's stdout: the result line goes to our stdout, anything else a library printed there (e.g. gloo's connection
    # notice) to stderr -- the contract is ONE JSON line
    result_lines = 0
    for ln in out0.decode("utf-8", "replace").splitlines():
        if ln.startswith("{"):
            result_lines += 1
            print(ln, flush=True)
        elif ln.strip():
            print(ln, file=sys.stderr)
    if rc == 0 and result_lines != 1:
        print(f"bench.py launcher: rank 0 printed {result_lines} result lines, expected 1", file=sys.stderr)
        rc = 1
    return rc


# ------------------------------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------------------------------
def make_inputs(batch, T, n, m, seed, device):
    import torch
    from tests import problems
    A1, B1, Q1, R1 = problems.random_lti_systems(batch, n, m, seed=seed)
    out = []
    for X in (A1, B1, Q1, R1):
        t = torch.as_tensor(X, device=device)
        out.append(t[:, None].expand(-1, T, -1, -1).contiguous())  # materialised (b,T,.,.) as the reference API takes
    return out


class HipLqrWorkload:
    """The product path: zm_lqr_backward_f64 through the C ABI on this rank's GPU."""
    stub = False

    def __init__(self, args, rank, local_rank):
        import ctypes
        import torch
        from zopt_amd import _lib
        self.torch, self._lib, self.ctypes = torch, _lib, ctypes
        self.dev = torch.device("cuda", local_rank)
        torch.cuda.set_device(self.dev)
        self.lib = _lib.lib()
        b, T, n, m = args.batch, args.T, args.n, args.m
        self.shape = (b, T, n, m)
        # two distinct resident input sets, alternated per step, so that no step can be served from the 256 MiB L3
        self.sets = [make_inputs(b, T, n, m, seed=2 * rank + i, device=self.dev) for i in range(2)]
        self.L = torch.empty((b, T, m, n), dtype=torch.float64, device=self.dev)
        self.stream = torch.cuda.current_stream(self.dev)
        self.last_set = 0

    def device_name(self):
        p = self.torch.cuda.get_device_properties(self.dev)
        return f"{self.dev} {p.name} {getattr(p, 'gcnArchName', '')}".strip()

    def step(self, i):
        b, T, n, m = self.shape
        A, B, Q, R = self.sets[i & 1]
        self.last_set = i & 1
        rc = self.lib.zm_lqr_backward_f64(A.data_ptr(), B.data_ptr(), Q.data_ptr(), R.data_ptr(), self.L.data_ptr(), b, T, n, m,
                                          self.ctypes.c_void_p(self.stream.cuda_stream))
        self._lib.check(rc, "zm_lqr_backward_f64")

    def sync(self):
        self.torch.cuda.synchronize()

    def event(self):
        return self.torch.cuda.Event(enable_timing=True)

    def record(self, ev):
        ev.record(self.stream)      # HIP event on the stream the kernel is launched on

    def result(self):
        return self.L


class StubWorkload:
    """CPU stand-in for the launch (gloo rehearsal of the multi-rank harness; never a measurement)."""
    stub = True

    def __init__(self, args, rank, local_rank):
        import torch
        self.torch = torch
        self.dev = torch.device("cpu")
        b, T, n, m = args.batch, args.T, args.n, args.m
        self.shape = (b, T, n, m)
        g = torch.Generator().manual_seed(rank)
        self.src = torch.randn((b, T, m, n), dtype=torch.float64, generator=g)
        self.L = torch.empty_like(self.src)
        self.rank = rank
        self.fail_rank = args.stub_fail_rank

    def device_name(self):
        return f"cpu (stub) pid {os.getpid()}"

    def step(self, i):
        if self.fail_rank == self.rank:
            raise RuntimeError("stub failure requested (--stub-fail-rank)")
        self.torch.add(self.src, float(self.rank), out=self.L)

    def sync(self):
        pass

    def event(self):
        return [0.0]

    def record(self, ev):
        ev[0] = time.perf_counter()

    def result(self):
        return self.L


def elapsed_ms(work, a, b):
    return (b[0] - a[0]) * 1e3 if work.stub else a.elapsed_time(b)


# ------------------------------------------------------------------------------------------------------------------------
# CPU legs (rank 0, N = 1): baselines and the parity check of the timed kernel's output
# ------------------------------------------------------------------------------------------------------------------------
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
        except Exception:  # noqa: BLE001
            continue
    return n


def host_cpu():
    """Model name and logical CPU count of the host (printed next to the CPU baselines), from /proc/cpuinfo"""
    model = None
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    import numpy
    return {"model": model, "logical_cpus": os.cpu_count(), "numpy": numpy.__version__}


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
                      f"oracle/riccati_oracle.c with OpenMP over the batch", "host": host_cpu()}


def cpu_baseline_numpy(batch, T, n, m, target_seconds=4.0):
    """BASELINE.md section 4 row 1 / SURVEY 8(d)(i): the single-process NumPy batched restatement (Python loop over T,
    batched matmul + batched LAPACK solve) on a bounded sample of the same workload."""
    from oracle import zopt_oracle as zo
    from tests import problems
    sample = min(batch, 512)
    A1, B1, Q1, R1 = problems.random_lti_systems(sample, n, m, seed=0)
    A, B, Q, R = problems.tile_over_horizon(A1, B1, Q1, R1, T)
    zo.discreteFiniteHorizonLqr(A[:8], B[:8], Q[:8], R[:8], T)   # warm-up
    reps, t0 = 0, time.perf_counter()
    while True:
        zo.discreteFiniteHorizonLqr(A, B, Q, R, T)
        reps += 1
        elapsed = time.perf_counter() - t0
        if elapsed >= target_seconds:
            break
    return {"value": sample * T * reps / elapsed, "unit": "horizon-steps/s", "cores": 1, "kind": "port",
            "sample": f"{sample} of the {batch} trajectories x T={T} x {reps} reps, oracle/zopt_oracle.py (NumPy batched matmul + "
                      f"numpy.linalg.solve, Python loop over T), one process"}


def parity_of_timed_output(work, ntraj=4):
    """Checker leg: the first `ntraj` trajectories of the LAST timed launch's gains against oracle/riccati_oracle.c."""
    import numpy as np
    from oracle import c_oracle
    A, B, Q, R = (x[:ntraj].cpu().numpy() for x in work.sets[work.last_set])
    got = work.L[:ntraj].cpu().numpy()
    ref = c_oracle.lqr_backward(A, B, Q, R, nthreads=1)
    return float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))


# ------------------------------------------------------------------------------------------------------------------------
def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=4096, help="trajectories per GPU (weak scaling)")
    ap.add_argument("--T", type=int, default=50)
    ap.add_argument("--n", type=int, default=12)
    ap.add_argument("--m", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the post-timing check of the kernel's output against the C oracle")
    ap.add_argument("--no-secondary", action="store_true", help="skip the bounded runs of BASELINE configs[2..4] after the timed region")
    ap.add_argument("--secondary-budget-s", type=float, default=25.0)
    ap.add_argument("--prewarm-ms", type=float, default=60.0,
                    help="untimed device warm-up before the W warm-up steps: launches until this much time has passed, so that a "
                         "run with small --warmup/--steps is not a measurement of the clock ramp (0 disables)")
    ap.add_argument("--gather", action="store_true", help="(default for N > 1) also time an RCCL all-gather of the gains, reported separately")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the all-gather of the results after the timed region")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend (nccl = RCCL; gloo with --stub-step for the CPU rehearsal)")
    ap.add_argument("--stub-step", action="store_true", help="CPU stand-in for the HIP launch (harness rehearsal; not a measurement)")
    ap.add_argument("--stub-fail-rank", type=int, default=-1, help="(stub only) this rank raises in its first step: failure-propagation test")
    ap.add_argument("--launch-timeout-s", type=float, default=1500.0, help="launcher: kill the ranks and fail after this long")
    return ap.parse_args(argv)


def run_rank(args):
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if args.stub_step and args.backend != "gloo":
        raise SystemExit("bench.py: --stub-step needs --backend gloo")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    formed = dist.get_world_size() if (world > 1 and dist.is_initialized()) else 1

    work = (StubWorkload if args.stub_step else HipLqrWorkload)(args, rank, local_rank)
    batch, T, n, m = args.batch, args.T, args.n, args.m

    def barrier():
        if world > 1:
            dist.barrier()

    if args.prewarm_ms > 0 and not work.stub:   # leave the idle clock state (docstring); not part of W or of the timed region
        p0 = time.perf_counter()
        while (time.perf_counter() - p0) * 1e3 < args.prewarm_ms:
            for i in range(16):
                work.step(i)
            work.sync()
    for i in range(args.warmup):
        work.step(i)
    work.sync()
    # HIP events around single launches, on every `stride`-th step only: an event is a marker packet between two kernels, and a pair
    # around every launch costs the timed region ~3 us per step of its own (`value` is wall-clock over all K steps either way)
    stride = 8 if args.steps >= 32 else (4 if args.steps >= 8 else 1)
    evs = {i: (work.event(), work.event()) for i in range(0, args.steps, stride)}
    barrier()
    work.sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev = evs.get(i)
        if ev is not None:
            work.record(ev[0])
        work.step(i)
        if ev is not None:
            work.record(ev[1])
    work.sync()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    kern_ms = float(np.mean([elapsed_ms(work, a, b) for a, b in evs.values()]))

    devices = [work.device_name()]
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=work.dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)       # max over ranks
        elapsed = float(tmax.item())
        names = [None] * world
        dist.all_gather_object(names, devices[0])
        devices = names

    # After (outside) the timed region: the one exchange the path has -- every rank collects all gains (north star: "RCCL
    # all-gather of results over xGMI").  Reported separately; it is never part of `value`.  What can fail on ONE rank (the
    # receive buffer of world x 78.6 MB) is done first and the ranks agree on it with an all-reduce, so that no rank skips a
    # collective the others enter; a failure inside the collective itself is fatal for the run (non-zero exit).
    gather_ms, gather_err = None, None
    if world > 1 and not args.no_gather:
        from zopt_amd import dist as zdist
        L = work.result()
        ok, full = 1, None
        try:
            full = torch.empty((world * batch,) + tuple(L.shape[1:]), dtype=L.dtype, device=L.device)
        except Exception as e:  # noqa: BLE001
            ok, gather_err = 0, f"{type(e).__name__}: {e}"
        flag = torch.tensor([ok], dtype=torch.int32, device=work.dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            zdist.allgather_results(L, world * batch, out=full)      # warm-up (RCCL communicator setup)
            work.sync()
            barrier()
            g0 = time.perf_counter()
            zdist.allgather_results(L, world * batch, out=full)
            work.sync()
            gather_ms = (time.perf_counter() - g0) * 1e3
            lo = rank * batch
            if not torch.equal(full[lo:lo + batch], L):
                raise SystemExit("bench.py: all-gather returned a different shard than this rank contributed")
        elif gather_err is None:
            gather_err = "another rank could not allocate the receive buffer"
        del full

    if rank == 0:
        steps_per_launch = batch * T
        bps = bytes_per_step(n, m)
        achieved = bps * steps_per_launch / (kern_ms * 1e-3) / 1e9
        traffic, traffic_source = None, "none"
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath) and (batch, T, n, m) == (4096, 50, 12, 4) and not work.stub:
            tj = json.load(open(tpath))
            if tj.get("k1_source_sha") == k1_source_sha():
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_source = (f"profiles/traffic_latest.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of THIS kernel source "
                                  f"(sha {tj['k1_source_sha']}), {tj.get('source', '')}; not measured in this run")
            else:
                traffic_source = (f"profiles/traffic_latest.json was measured on another version of the kernel source "
                                  f"(sha {tj.get('k1_source_sha')}, current {k1_source_sha()}): not quoted")
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
            "rccl_ranks": formed, "backend": args.backend if world > 1 else None, "devices": devices,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "lqr_backward_dma_f64<12,4,3>" if (n, m) == (12, 4) else "lqr_backward", "kernel_ms": kern_ms,
                         "kernel_ms_samples": len(evs),
                         "algorithmic_bytes_per_launch": bps * steps_per_launch},
        }
        if work.stub:
            res["stub"] = True
            res["roofline"]["kernel"] = "stub (CPU tensor op): harness rehearsal, not a measurement"
        if (n, m) == (12, 4) and not work.stub:
            # Informational: what the kernel issues on the fp64 matrix pipe.  Per horizon step a wave issues 9 v_mfma_f64_16x16x4
            # (2048 flop) + 3 v_mfma_f64_4x4x4_4b (512 flop).  The pipe's peak is the nominal 78.6 TFLOP/s: the chip holds 2.35-2.39 GHz
            # under pure fp64 MFMA load and issues one 16x16x4 per 64 cycles and SIMD (profiles/r02_ubench_clock_f64.txt; round 1's
            # "47.2 TFLOP/s sustained" was an artifact of its microbenchmark).  This kernel is bound by the memory system, not by this pipe.
            issued = (9 * 2048 + 3 * 512) * steps_per_launch / (kern_ms * 1e-3) / 1e12
            res["fp64_matrix_pipe"] = {"issued_mfma_tflops": issued, "peak_tflops": 78.6, "frac": issued / 78.6}
        if gather_ms is not None:
            nbytes = world * batch * T * m * n * 8
            res["allgather"] = {"ms": gather_ms, "bytes_per_rank": nbytes, "GBps_per_rank": nbytes / (gather_ms * 1e-3) / 1e9}
        if gather_err is not None:
            res["allgather"] = {"error": gather_err}
        if not work.stub and not args.no_parity:
            res["parity_rel_err"] = parity_of_timed_output(work)      # checker leg (oracle/riccati_oracle.c), after the timing
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(batch, T, n, m)
            res["cpu_baseline_numpy"] = cpu_baseline_numpy(batch, T, n, m)
        if world == 1 and not work.stub and not args.no_secondary:
            del work.sets                      # free the 1.15 GB of headline inputs before the secondary workloads
            torch.cuda.empty_cache()
            from tools import secondary_bench
            res["secondary"] = secondary_bench.run_all(args.secondary_budget_s)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, argv, args.launch_timeout_s))
    run_rank(args)


if __name__ == "__main__":
    main()
