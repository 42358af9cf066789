"""CPU rehearsal of bench.py's multi-rank harness (`-m "not gpu"`): the SAME launcher, rank initialisation, barrier,
max-over-ranks reduction and `allgather_results` code the N-GPU run uses, with the gloo backend and a stub step
(`--backend gloo --stub-step`: a CPU tensor op instead of the HIP launch).  Covers the command form the driver uses
(`python bench.py --gpus N` with no WORLD_SIZE in the environment) and the torch.distributed.run form."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
STUB = ["--backend", "gloo", "--stub-step", "--steps", "4", "--warmup", "1", "--batch", "6", "--T", "5", "--no-cpu-baseline"]


def _clean_env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    return env


def _run(cmd, env, timeout=300):
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def _one_json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_self_launch_two_ranks_gloo():
    out = _run([sys.executable, BENCH, "--gpus", "2"] + STUB, _clean_env())
    assert out.returncode == 0, out.stderr[-2000:]
    res = _one_json_line(out.stdout)
    assert res["n_gpus"] == 2 and res["rccl_ranks"] == 2 and res["backend"] == "gloo"
    assert len(res["devices"]) == 2 and len(set(res["devices"])) == 2          # two distinct rank processes
    assert res["steps"] == 4 and res["warmup"] == 1 and res["stub"] is True
    assert res["value"] > 0 and res["ms_per_step"] > 0
    assert res["value"] == pytest.approx(2 * 6 * 5 * 4 / (res["ms_per_step"] * 4e-3), rel=1e-9)   # whole-job units / max-over-ranks time
    assert "error" not in res["allgather"] and res["allgather"]["bytes_per_rank"] == 2 * 6 * 5 * 4 * 12 * 8
    assert "cpu_baseline" not in res                                            # N = 1 only


def test_three_ranks_and_no_gather():
    out = _run([sys.executable, BENCH, "--gpus", "3", "--no-gather"] + STUB, _clean_env())
    assert out.returncode == 0, out.stderr[-2000:]
    res = _one_json_line(out.stdout)
    assert res["n_gpus"] == 3 and res["rccl_ranks"] == 3 and len(res["devices"]) == 3 and "allgather" not in res


def test_failing_rank_fails_the_launcher():
    out = _run([sys.executable, BENCH, "--gpus", "2", "--stub-fail-rank", "1", "--launch-timeout-s", "120"] + STUB, _clean_env())
    assert out.returncode != 0
    assert "rank 1 exited" in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]   # no result line from a failed run


def test_under_torch_distributed_run():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH, "--gpus", "2"] + STUB
    out = _run(cmd, _clean_env())
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["rccl_ranks"] == 2


def test_single_rank_stub_and_world_size_mismatch():
    out = _run([sys.executable, BENCH, "--gpus", "1"] + STUB, _clean_env())
    assert out.returncode == 0, out.stderr[-2000:]
    res = _one_json_line(out.stdout)
    assert res["n_gpus"] == 1 and res["rccl_ranks"] == 1 and res["backend"] is None
    env = _clean_env()
    env.update({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    bad = _run([sys.executable, BENCH, "--gpus", "2"] + STUB, env)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in bad.stderr
