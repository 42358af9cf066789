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


@pytest.mark.parametrize("workload", ["mpc", "ilqr", "ddp", "n64"])
@pytest.mark.parametrize("ranks,scaling,batch", [(2, "weak", 4), (3, "strong", 8)])   # equal shards / ragged shards (3 + 3 + 2)
def test_every_workload_shards_and_gathers(workload, ranks, scaling, batch):
    """`--workload X --scaling S` on 2 and 3 self-launched ranks: each rank takes its `dist.shard_bounds` slice, barrier +
    max-over-ranks timing, ONE all-gather of the packed result tuple (xTraj, uTraj, L, J, converged for the solvers) which must
    reproduce the unsharded stub job on every rank (bench.py checks that itself and exits non-zero otherwise); n64 also runs the
    chunked gather overlapped with its chunked sweep."""
    out = _run([sys.executable, BENCH, "--gpus", str(ranks), "--workload", workload, "--scaling", scaling, "--batch", str(batch),
                "--backend", "gloo", "--stub-step", "--steps", "2", "--warmup", "1"], _clean_env())
    assert out.returncode == 0, out.stderr[-2000:]
    res = _one_json_line(out.stdout)
    assert res["n_gpus"] == ranks and res["rccl_ranks"] == ranks and res["scaling"] == scaling and res["stub"] is True
    assert len(set(res["devices"])) == ranks
    job = batch * ranks if scaling == "weak" else batch
    g = res["allgather"]
    assert "error" not in g and g["collectives"] == 1
    assert g["fields"] == {"mpc": 4, "ilqr": 5, "ddp": 5, "n64": 1}[workload]
    if workload == "n64":
        assert g["items"] in (job, job - 2)          # the stub trims a shard to a whole number of chunks (3 + 3 + 2 -> 2 + 2 + 2)
        ov = g["overlap"]
        assert ov["nchunks"] == 2 and ov["sweep_overlapped_gather_ms"] > 0 and ov["sweep_then_gather_ms"] > 0
        assert res["unit"] == "horizon-steps/s" and res["dtype"] == "f32"
    else:
        assert g["items"] == job
        per_step = job                                # one unit per problem / instance
        assert res["value"] == pytest.approx(per_step * 2 / (res["ms_per_step"] * 2e-3), rel=1e-9)
    assert "secondary_sharded" not in res             # only the headline workload carries the other configs along


def test_headline_at_two_ranks_carries_the_sharded_configs():
    """N > 1 with the default workload: after the headline, configs[2], [3] (iLQR and DDP) and [4] run strong-sharded over the
    same ranks, each with its result gather -- what a driver SCALE pass over N = 1, 2, 4, 8 records for them."""
    out = _run([sys.executable, BENCH, "--gpus", "2"] + STUB, _clean_env())
    assert out.returncode == 0, out.stderr[-2000:]
    res = _one_json_line(out.stdout)
    sh = res["secondary_sharded"]
    assert set(sh) == {"configs[2]_mpc", "configs[3]_ilqr", "configs[3]_ddp", "configs[4]_n64"}
    for key, e in sh.items():
        assert "error" not in e, (key, e)
        assert e["scaling"] == "strong" and e["job_items"] == 16 and e["items_rank0"] == 8 and e["value"] > 0
        assert "error" not in e["allgather"] and e["allgather"]["items"] == 16
    assert "overlap" in sh["configs[4]_n64"]["allgather"]
    out = _run([sys.executable, BENCH, "--gpus", "2", "--no-secondary"] + STUB, _clean_env())
    assert out.returncode == 0 and "secondary_sharded" not in _one_json_line(out.stdout)


def test_strong_scaling_of_the_headline_and_force_dist():
    out = _run([sys.executable, BENCH, "--gpus", "3", "--scaling", "strong", "--batch", "7", "--no-secondary", "--backend", "gloo",
                "--stub-step", "--steps", "2", "--warmup", "1", "--T", "5"], _clean_env())
    assert out.returncode == 0, out.stderr[-2000:]
    res = _one_json_line(out.stdout)
    assert res["scaling"] == "strong" and res["config"]["job_items"] == 7 and res["allgather"]["items"] == 7
    assert res["value"] == pytest.approx(7 * 5 * 2 / (res["ms_per_step"] * 2e-3), rel=1e-9)     # the job's units, not N x rank 0's
    # one rank with a process group all the same: the gather legs run (what a one-GPU box can rehearse with RCCL)
    out = _run([sys.executable, BENCH, "--gpus", "1", "--force-dist", "--workload", "n64", "--batch", "4", "--backend", "gloo",
                "--stub-step", "--steps", "2", "--warmup", "1"], _clean_env())
    assert out.returncode == 0, out.stderr[-2000:]
    res = _one_json_line(out.stdout)          # no launcher in between: the rank itself keeps gloo's notice off stdout
    assert res["rccl_ranks"] == 1 and res["allgather"]["items"] == 4 and "overlap" in res["allgather"]


def test_a_rank_stalling_after_the_headline_does_not_cost_the_line():
    """the legs after the headline (result gathers, strong-sharded configs) are collectives: when one rank never arrives, rank 0 still
    prints the headline line -- with an error entry for the legs -- once --optional-budget-s has passed, and every rank exits"""
    out = _run([sys.executable, BENCH, "--gpus", "2", "--stub-stall-rank", "1", "--optional-budget-s", "4", "--launch-timeout-s", "120"]
               + STUB, _clean_env())
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["value"] > 0 and "allgather" not in res
    assert "optional_legs_error" in res and "all-gather" in res["optional_legs_error"]
