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
    """Starts n rank processes of this file, relays rank 0's result line, returns the exit code (0 only if every rank exited 0).
    Rank 0's stdout goes to a temporary FILE (a pipe read only at exit would block the rank once a library has written more than
    the pipe holds), the other ranks' stdout to our stderr (what a failing rank prints there must not be lost)."""
    import tempfile
    port = free_port()
    procs = []
    out0_file = tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on these hosts (RCCL needs it)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=out0_file if r == 0 else sys.stderr))
    deadline = time.monotonic() + timeout_s
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
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
    out0_file.seek(0)
    out0 = out0_file.read()
    out0_file.close()
    # rank 0's stdout: the result line goes to our stdout, anything else a library printed there (e.g. gloo's connection
    # notice) to stderr -- the contract is ONE JSON line
    result_lines = 0
    for ln in out0.decode("utf-8", "replace").splitlines():
        if ln.startswith("{"):
            result_lines += 1
            if rc == 0:
                print(ln, flush=True)
        elif ln.strip():
            print(ln, file=sys.stderr)
    if rc == 0 and result_lines != 1:
        print(f"bench.py launcher: rank 0 printed {result_lines} result lines, expected 1", file=sys.stderr)
        rc = 1
    return rc


# ------------------------------------------------------------------------------------------------------------------------
# workloads: tools/workloads.py (one class per BASELINE config, each shardable over ranks; Stub twins for the gloo rehearsal)
# ------------------------------------------------------------------------------------------------------------------------
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
    ap.add_argument("--workload", default="lqr", choices=["lqr", "mpc", "ilqr", "ddp", "n64"],
                    help="which BASELINE config one step is a pass of: lqr = configs[1] (the headline, default), mpc = configs[2], "
                         "ilqr / ddp = configs[3], n64 = configs[4] (tools/workloads.py)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank owns --batch items (default: the config's per-GPU share); strong: the config's total "
                         "(--batch overrides it) is split contiguously over the ranks")
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default per workload: lqr 300, mpc 10, ilqr 5, ddp 3, n64 10)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default per workload: lqr 50, mpc 2, ilqr 1, ddp 1, n64 3)")
    ap.add_argument("--batch", type=int, default=None, help="items per GPU (weak) or in the whole job (strong); default: the config's")
    ap.add_argument("--T", type=int, default=None)
    ap.add_argument("--n", type=int, default=None)
    ap.add_argument("--m", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the post-timing check of the kernel's output against the C oracle")
    ap.add_argument("--no-secondary", action="store_true", help="skip the bounded runs of BASELINE configs[2..4] after the timed region")
    ap.add_argument("--secondary-budget-s", type=float, default=25.0)
    ap.add_argument("--prewarm-ms", type=float, default=60.0,
                    help="untimed device warm-up before the W warm-up steps: launches until this much time has passed, so that a "
                         "run with small --warmup/--steps is not a measurement of the clock ramp (0 disables)")
    ap.add_argument("--gather", action="store_true", help="(default for N > 1) also time an RCCL all-gather of the results, reported separately")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the all-gather of the results after the timed region")
    ap.add_argument("--force-dist", action="store_true",
                    help="N = 1: initialise a one-rank process group all the same, so that the gather legs (RCCL all-gather, chunked "
                         "overlapped gather) run on the one GPU a builder has")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a box with fewer GPUs than ranks: rank r runs its HIP workload on GPU r %% device_count (several "
                         "processes share a device; RCCL refuses that, so use --backend gloo: collectives are staged through host memory). "
                         "Exercises the REAL sharded kernels and gathers; its timings are not measurements")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend (nccl = RCCL; gloo with --stub-step for the CPU rehearsal)")
    ap.add_argument("--stub-step", action="store_true", help="CPU stand-in for the HIP launch (harness rehearsal; not a measurement)")
    ap.add_argument("--stub-fail-rank", type=int, default=-1, help="(stub only) this rank raises in its first step: failure-propagation test")
    ap.add_argument("--stub-stall-rank", type=int, default=-1,
                    help="(stub only) this rank never enters the legs after the headline: test of --optional-budget-s")
    ap.add_argument("--launch-timeout-s", type=float, default=1500.0, help="launcher: kill the ranks and fail after this long")
    ap.add_argument("--optional-budget-s", type=float, default=420.0,
                    help="N > 1: the legs after the headline (result gathers, strong-sharded configs[2..4]) get this long; "
                         "after that rank 0 prints the headline line with an error entry for them and every rank exits")
    return ap.parse_args(argv)


class Comm:
    """The few collectives the harness needs, no-ops on one rank without a process group."""

    def __init__(self, dist, world, dev):
        self.dist, self.world, self.dev = dist, world, dev
        self.on = dist.is_initialized()
        self.host_staged = False

    def barrier(self):
        if self.on:
            self.dist.barrier()

    def reduce(self, x, op):
        import torch
        if not self.on:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=self.dev)
        self.dist.all_reduce(t, op={"max": self.dist.ReduceOp.MAX, "min": self.dist.ReduceOp.MIN, "sum": self.dist.ReduceOp.SUM}[op])
        return float(t.item())

    def names(self, mine):
        if not self.on:
            return [mine]
        out = [None] * self.dist.get_world_size()
        self.dist.all_gather_object(out, mine)
        return out


def timed_steps(work, comm, steps, warmup, prewarm_ms):
    """W warm-up steps, then EXACTLY `steps` steps between barrier + synchronize on both sides; returns (max-over-ranks seconds,
    this rank's mean HIP-event ms per step, number of event samples)."""
    import numpy as np
    if prewarm_ms > 0 and not work.stub:   # leave the idle clock state (docstring); not part of W or of the timed region
        p0 = time.perf_counter()
        while (time.perf_counter() - p0) * 1e3 < prewarm_ms:
            for i in range(16 if work.name == "lqr" else 1):
                work.step(i)
            work.sync()
    for i in range(warmup):
        work.step(i)
    work.sync()
    # HIP events around single launches, on every `stride`-th step only: an event is a marker packet between two kernels, and a pair
    # around every launch costs the timed region ~3 us per step of its own (`value` is wall-clock over all K steps either way)
    stride = 8 if steps >= 32 else (4 if steps >= 8 else 1)
    evs = {i: (work.event(), work.event()) for i in range(0, steps, stride)}
    comm.barrier()
    work.sync()
    t0 = time.perf_counter()
    for i in range(steps):
        ev = evs.get(i)
        if ev is not None:
            work.record(ev[0])
        work.step(i)
        if ev is not None:
            work.record(ev[1])
    work.sync()
    comm.barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = float(np.mean([elapsed_ms(work, a, b) for a, b in evs.values()]))
    return comm.reduce(elapsed, "max"), kern_ms, len(evs)


def same_bits(a, b):
    """Bitwise equality of two tensors (NaN == NaN: a diverged iLQR start's results are NaN, and they must travel unchanged too)."""
    import torch
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    if a.dtype in (torch.float64, torch.float32):
        it = torch.int64 if a.dtype == torch.float64 else torch.int32
        return bool(torch.equal(a.contiguous().view(it), b.contiguous().view(it)))
    return bool(torch.equal(a, b))


def gather_leg(work, comm, rank):
    """After (outside) the timed region: the one exchange the path has -- every rank collects all results (north star: "RCCL
    all-gather of results over xGMI") as ONE collective over the packed result tuple.  What can fail on ONE rank (packing, i.e.
    allocating) is done first and the ranks agree on it with an all-reduce, so that no rank skips a collective the others enter; a
    failure inside the collective itself is fatal for the run (non-zero exit)."""
    import torch
    from zopt_amd import dist as zdist
    ok, err, packed, fields = 1.0, None, None, work.results()
    try:
        packed = fields[0].contiguous() if len(fields) == 1 else zdist.pack_results(fields)
        if comm.host_staged:
            packed = packed.cpu()
    except Exception as e:  # noqa: BLE001
        ok, err = 0.0, f"{type(e).__name__}: {e}"
    if comm.reduce(ok, "min") < 1.0:
        return {"error": err or "another rank could not pack its results"}
    counts = zdist.shard_counts(packed.shape[0])          # rank by rank (a shard may have been capped by memory)
    total = sum(counts)
    full = zdist.allgather_results(packed, total, counts=counts)      # warm-up (communicator setup)
    work.sync()
    comm.barrier()
    g0 = time.perf_counter()
    full = zdist.allgather_results(packed, total, counts=counts)
    work.sync()
    ms = (time.perf_counter() - g0) * 1e3
    lo = sum(counts[:rank])
    if not same_bits(full[lo:lo + packed.shape[0]], packed):
        raise SystemExit("bench.py: all-gather returned a different shard than this rank contributed")
    if work.stub and work.name != "n64" and getattr(work, "lo", 0) == lo:
        # the stub's item g carries values that depend on g only: the gathered job must be the unsharded job, on every rank
        from tools import workloads as wl
        b, T, n, m = work.shape
        exp = wl.StubWorkload.expected(work.name, total, T=T, n=n, m=m)
        exp = exp[0] if len(exp) == 1 else zdist.pack_results(exp)
        if not same_bits(full, exp):
            raise SystemExit("bench.py: the gathered stub job differs from the unsharded one")
    nbytes = full.numel() * full.element_size()
    return {"ms": ms, "bytes_per_rank": nbytes, "GBps_per_rank": nbytes / (ms * 1e-3) / 1e9, "fields": len(fields),
            "collectives": 1, "items": int(full.shape[0])}


def overlapped_gather_leg(work, comm, reps=3):
    """configs[4]: the rank's sweep in `nchunks` launches with the all-gather of every finished chunk's gains overlapped with
    the next chunk's sweep (zopt_amd.dist.ChunkedGather), against the same launches followed by ONE gather of the whole shard."""
    import torch
    from zopt_amd import dist as zdist
    if comm.host_staged:
        return {"skipped": "host-staged collectives (gloo rehearsal of HIP workloads): nothing to overlap"}
    L = work.results()[0]
    nch = work.nchunks
    counts = zdist.shard_counts(L.shape[0])
    if min(counts) != max(counts):        # every rank sees the same list: all skip together
        return {"skipped": f"unequal shards {counts}: the chunked gather needs equal ones"}
    cg = zdist.ChunkedGather(tuple(L.shape), nch, L.dtype, L.device)
    whole = torch.empty((comm.world,) + tuple(L.shape), dtype=L.dtype, device=L.device)

    def sequential():
        for c in range(nch):
            work.step_chunk(c)
        comm.dist.all_gather_into_tensor(whole.view((-1,) + tuple(L.shape[1:])), L)

    def overlapped():
        for c in range(nch):
            cg.issue(c, work.step_chunk(c), None if work.stub else work.stream)
        cg.wait()

    def sweep_only():
        for c in range(nch):
            work.step_chunk(c)

    out = {}
    for key, fn in (("sweep_chunked_ms", sweep_only), ("sweep_then_gather_ms", sequential), ("sweep_overlapped_gather_ms", overlapped)):
        fn()
        work.sync()
        ts = []
        for _ in range(reps):
            comm.barrier()
            t0 = time.perf_counter()
            fn()
            work.sync()
            ts.append(comm.reduce(time.perf_counter() - t0, "max") * 1e3)
        out[key] = min(ts)
    r = comm.dist.get_rank()
    if not same_bits(cg.global_view()[r].reshape(L.shape), L) or not same_bits(whole[r], L):
        raise SystemExit("bench.py: chunked gather returned a different shard than this rank contributed")
    if not same_bits(cg.global_view().reshape(whole.shape), whole):
        raise SystemExit("bench.py: chunked and whole-shard gathers disagree")
    out.update({"nchunks": nch, "bytes_per_rank": whole.numel() * whole.element_size()})
    return out


def workload_line(work):
    """Metric name, workload description and (batch, T, n, m) of the result line (rank 0)."""
    from tools import workloads as wl
    b, T, n, m = getattr(work, "shape", (work.units_per_step, None, None, None))
    spec = wl.SPECS[work.name]
    T = T or spec["T"]
    n, m = n or spec["n"], m or spec["m"]
    what = {"lqr": "discreteFiniteHorizonLqr: {b} random LTI systems per GPU, n={n} m={m} T={T} fp64, A/B/Q/R materialised (b,T,.,.), BASELINE configs[1]",
            "mpc": "lqrMpc quadcopter n=12 m=4 N={T}, {b} instances per GPU, eps_abs=eps_rel=1e-2, cold start, BASELINE configs[2]",
            "ilqr": "iterativeLqr quadcopter n=12 m=4 T={T}, {b} problems per GPU, fp64, R=1*I, maxIter=100, tol=1e-3, BASELINE configs[3]",
            "ddp": "differentialDynamicProgramming quadcopter n=12 m=4 T={T}, {b} problems per GPU, fp64, R=0.2*I, maxIter=100, tol=1e-3, BASELINE configs[3]",
            "n64": "discreteFiniteHorizonLqr: {b} random LTI systems per GPU, n={n} m={m} T={T} fp32, inputs generated on the device, BASELINE configs[4]"}
    metric = {"lqr": f"LQR horizon-steps/sec (batch x T) at n={n},m={m},T={T}",
              "mpc": f"lqrMpc instance-solves/sec (quadcopter n=12,m=4,N={T}, demo tolerance)",
              "ilqr": f"iterativeLqr problems solved/sec (quadcopter n=12,m=4,T={T})",
              "ddp": f"differentialDynamicProgramming problems solved/sec (quadcopter n=12,m=4,T={T})",
              "n64": f"LQR horizon-steps/sec (batch x T) at n={n},m={m},T={T} fp32"}[work.name]
    return metric, what[work.name].format(b=b, T=T, n=n, m=m), (b, T, n, m)


class OptionalLegsWatchdog:
    """The legs that follow the headline at N > 1 (result all-gathers, the strong-sharded configs[2..4]) are collectives over every
    rank: a rank that fails inside one leaves the others waiting.  That must not cost the run its headline, which is complete before
    they start -- after `seconds` a timer thread lets rank 0 print the headline line with an error entry in place of the missing legs,
    and ends the process (`os._exit`: the main thread may sit in a collective that never returns).  HIP synchronisation and c10d
    collectives release the GIL, so the timer thread does run."""

    def __init__(self, seconds, emit):
        import threading
        self.seconds, self.emit, self.stage = seconds, emit, "start"
        self.timer = threading.Timer(seconds, self.fire)
        self.timer.daemon = True

    def start(self):
        self.timer.start()

    def cancel(self):
        self.timer.cancel()

    def fire(self):
        msg = f"not finished {self.seconds:.0f} s after the headline (--optional-budget-s), last stage entered: {self.stage}"
        print(f"bench.py: optional legs {msg}", file=sys.stderr, flush=True)
        try:
            self.emit(msg)
        finally:
            os._exit(0)


def run_rank(args):
    # ONE JSON line on stdout, nothing else: RCCL prints a version banner to stdout when a communicator is created, gloo a connection
    # notice -- everything this process or its libraries print goes to stderr, the result line alone to the saved stdout
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from tools import workloads as wl

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if args.stub_step and args.backend != "gloo":
        raise SystemExit("bench.py: --stub-step needs --backend gloo")
    if world > 1 or args.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    formed = dist.get_world_size() if dist.is_initialized() else 1

    if args.share_gpu:
        if args.backend != "gloo" and world > 1:
            raise SystemExit("bench.py: --share-gpu needs --backend gloo (RCCL refuses two ranks on one device)")
        local_rank = local_rank % max(1, torch.cuda.device_count())
    spec = wl.SPECS[args.workload]
    steps = args.steps if args.steps is not None else spec["steps"]
    warmup = args.warmup if args.warmup is not None else spec["warmup"]
    work, total = wl.make(args.workload, args.scaling, world, rank, local_rank, batch=args.batch, stub=args.stub_step,
                          stub_fail_rank=args.stub_fail_rank, T=args.T, n=args.n, m=args.m)
    # gloo moves host memory: with real (HIP) workloads under --backend gloo the harness' scalars and the gathered results are staged
    # through the CPU (rehearsal); RCCL takes device tensors directly
    host_staged = (args.backend == "gloo") and not work.stub
    comm = Comm(dist, world, torch.device("cpu") if host_staged else work.dev)
    comm.host_staged = host_staged
    elapsed, kern_ms, nev = timed_steps(work, comm, steps, warmup, args.prewarm_ms)
    job_units = comm.reduce(float(work.units_per_step), "sum")      # ragged shards: the job's units are the sum over ranks
    devices = comm.names(work.device_name())

    legs = {"gather": None, "sharded": None, "parity": None}

    def emit(optional_error=None):
        """rank 0: the ONE result line, from what has been measured so far"""
        gather, sharded, parity = legs["gather"], legs["sharded"], legs["parity"]
        metric, what, (batch, T, n, m) = workload_line(work)
        res = {
            "metric": metric,
            "value": job_units * steps / elapsed,
            "unit": work.unit,
            "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": work.dtype, "data": "synthetic",
            "config": {"workload": what, "batch_per_gpu": batch, "T": T, "n": n, "m": m, "parallelism": f"batch-sharded x{world}"},
            "rccl_ranks": formed, "backend": args.backend if dist.is_initialized() else None, "devices": devices,
        }
        if args.scaling == "strong":
            res["config"]["job_items"] = total
        if args.workload == "lqr":
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
            res["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                               "kernel": "lqr_backward_dma_f64<12,4,3>" if (n, m) == (12, 4) else "lqr_backward", "kernel_ms": kern_ms,
                               "kernel_ms_samples": nev,
                               "algorithmic_bytes_per_launch": bps * steps_per_launch}
            if work.stub:
                res["roofline"]["kernel"] = "stub (CPU tensor op): harness rehearsal, not a measurement"
            if (n, m) == (12, 4) and not work.stub:
                # Informational: what the kernel issues on the fp64 matrix pipe.  Per horizon step a wave issues 9 v_mfma_f64_16x16x4
                # (2048 flop) + 3 v_mfma_f64_4x4x4_4b (512 flop).  The pipe's peak is the nominal 78.6 TFLOP/s: the chip holds 2.35-2.39 GHz
                # under pure fp64 MFMA load and issues one 16x16x4 per 64 cycles and SIMD (profiles/r02_ubench_clock_f64.txt; round 1's
                # "47.2 TFLOP/s sustained" was an artifact of its microbenchmark).  This kernel is bound by the memory system, not by this pipe.
                issued = (9 * 2048 + 3 * 512) * steps_per_launch / (kern_ms * 1e-3) / 1e12
                res["fp64_matrix_pipe"] = {"issued_mfma_tflops": issued, "peak_tflops": 78.6, "frac": issued / 78.6}
        elif args.workload == "n64" and not work.stub:
            tflops = batch * T * wl.tiled_mfma_per_step(n) * 2048 / (kern_ms * 1e-3) / 1e12
            res["roofline"] = {"bound": "mfma", "achieved": tflops, "peak": wl.FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": tflops / wl.FP32_MATRIX_PEAK_TFLOPS, "traffic": None, "kernel": "lqr_backward_tiled_f32",
                               "kernel_ms": kern_ms, "kernel_ms_samples": nev,
                               "note": "issued v_mfma_f32_16x16x4 flops of one sweep launch over this rank's shard / its HIP-event time"}
        else:
            res["roofline"] = None      # latency-bound solves (tiny HBM footprint): the yardstick is ms per solve, DESIGN section 4.2
            res["solve_ms_rank0"] = kern_ms
        if work.stub:
            res["stub"] = True
        res.update(work.extras())
        if gather is not None:
            res["allgather"] = gather
        if sharded is not None:
            res["secondary_sharded"] = sharded
        if parity is not None:
            res["parity_rel_err"] = parity
        if world == 1 and args.workload == "lqr" and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(batch, T, n, m)
            res["cpu_baseline_numpy"] = cpu_baseline_numpy(batch, T, n, m)
        if world == 1 and args.workload == "lqr" and not work.stub and not args.no_secondary:
            del work.sets                      # free the 1.15 GB of headline inputs before the secondary workloads
            torch.cuda.empty_cache()
            from tools import secondary_bench
            res["secondary"] = secondary_bench.run_all(args.secondary_budget_s)
        if optional_error is not None:
            res["optional_legs_error"] = optional_error
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(res) + "\n").encode())

    dog = None
    if comm.on and args.optional_budget_s > 0:
        dog = OptionalLegsWatchdog(args.optional_budget_s, emit if rank == 0 else (lambda msg: None))
        dog.start()
    if args.stub_step and rank == args.stub_stall_rank:
        time.sleep(3600.0)                    # the watchdog ends this rank too
    if comm.on and not args.no_gather:
        if dog:
            dog.stage = "result all-gather of the headline workload"
        legs["gather"] = gather_leg(work, comm, rank)
        if args.workload == "n64" and "error" not in legs["gather"]:
            legs["gather"]["overlap"] = overlapped_gather_leg(work, comm)

    if rank == 0 and args.workload == "lqr" and not work.stub and not args.no_parity:
        legs["parity"] = parity_of_timed_output(work)      # checker leg (oracle/riccati_oracle.c), after the timing; rank-local, no collective

    # N > 1, headline workload: the configs BASELINE defines as multi-GPU, strong-sharded over the same ranks (every rank takes part)
    if world > 1 and args.workload == "lqr" and not args.no_secondary:
        legs["sharded"] = sharded_secondary(args, comm, world, rank, local_rank, work, dog)
    if dist.is_initialized():
        if dog:
            dog.stage = "closing barrier"
        dist.barrier()
    if dog:
        dog.cancel()
    if rank == 0:
        emit()
    if dist.is_initialized():
        dist.destroy_process_group()



def sharded_secondary(args, comm, world, rank, local_rank, headline_work, dog=None):
    """N > 1: configs[2], [3] (iLQR and DDP) and [4] strong-sharded over the ranks of this run -- every rank solves its
    `dist.shard_bounds` slice of the config's total through the same C-ABI calls as at N = 1, barrier + max-over-ranks timing, then
    the result all-gather (configs[4]: also chunked and overlapped with the sweep).  One driver SCALE pass over N = 1, 2, 4, 8 thus
    yields the strong-scaling curve of every config next to the headline's weak-scaling one (at N = 1 the same workloads are
    `secondary`).  Construction is the one thing that can fail on a single rank; the ranks agree on it before any collective."""
    import torch
    from tools import workloads as wl
    if not headline_work.stub:
        for attr in ("sets", "L"):
            if hasattr(headline_work, attr):
                delattr(headline_work, attr)          # free the headline's buffers
        torch.cuda.empty_cache()
    out = {}
    plan = (("mpc", 3, 1), ("ilqr", 3, 1), ("ddp", 2, 1), ("n64", 5, 2))
    for name, steps, warmup in plan:
        key = f"configs[{wl.SPECS[name]['config']}]_{name}"
        if dog:
            dog.stage = f"strong-sharded {key}"
        work, ok, err = None, 1.0, None
        try:
            work, total = wl.make(name, "strong", world, rank, local_rank, batch=(8 * world if args.stub_step else None),
                                  stub=args.stub_step)
        except Exception as e:  # noqa: BLE001
            ok, err = 0.0, f"{type(e).__name__}: {e}"
        if comm.reduce(ok, "min") < 1.0:
            out[key] = {"error": err or "another rank could not set the workload up"}
            del work
            continue
        elapsed, kern_ms, _ = timed_steps(work, comm, steps, warmup, 0.0)
        units = comm.reduce(float(work.units_per_step), "sum")
        entry = {"scaling": "strong", "job_items": total, "items_rank0": int(work.results()[0].shape[0]), "steps": steps,
                 "ms_per_step": elapsed / steps * 1e3, "value": units * steps / elapsed, "unit": work.unit}
        entry.update(work.extras())
        if not args.no_gather:
            entry["allgather"] = gather_leg(work, comm, rank)
            if name == "n64" and "error" not in entry["allgather"]:
                entry["allgather"]["overlap"] = overlapped_gather_leg(work, comm)
        out[key] = entry
        del work
        if not args.stub_step:
            torch.cuda.empty_cache()
    return out


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, argv, args.launch_timeout_s))
    run_rank(args)


if __name__ == "__main__":
    main()
