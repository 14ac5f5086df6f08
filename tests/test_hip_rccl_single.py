"""The RCCL path on one GPU: bench.py's distributed branch (process group on backend "nccl", parameter broadcast,
one all-reduce AVG of the flat gradient buffer per step) in a fresh child process with WORLD_SIZE=1, against the same
run without a process group.  One rank averages with itself, so loss and throughput line must agree exactly; what the
test pins is that init_process_group("nccl") and the collectives execute on this stack."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bench(extra_args, extra_env, gpus=1):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", "2", "--batch", "4",
           "--skip-cpu", "--skip-beam", "--skip-opt", "--skip-sweep", "--skip-large"] + extra_args
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


def test_one_rank_nccl_process_group_runs_and_matches_plain_run():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    plain = _bench([], {})
    assert plain["rccl_ranks"] == 0 and plain["gradient_allreduce"] is None
    env = {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())}
    ddp = _bench(["--force-dist"], env)
    assert ddp["rccl_ranks"] == 1
    assert ddp["gradient_allreduce"]["backend"] == "nccl" and ddp["gradient_allreduce"]["bytes"] > 0
    assert ddp["n_gpus"] == 1 and ddp["config"]["parallelism"] == "dp1"
    assert ddp["loss"] == plain["loss"]          # same seed, same dropout stream, mean over one rank = identity
    # the collective is timed on its own (events around it on the step's stream, and host wall time)
    assert ddp["gradient_allreduce"]["ms_per_step"] > 0 and ddp["gradient_allreduce"]["wall_ms_per_step"] > 0
    assert plain["roofline"]["kernel"].startswith("k_edge_bwd") and plain["roofline_fwd"]["kernel"].startswith("k_edge_fwd")


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment: the script starts the two ranks itself (child
    processes created before the parent touches the GPU), rank 0's JSON line comes back on stdout and the exit status is
    0.  Two ranks share the one GPU of the test box, so the process group runs on gloo (GCN_BENCH_BACKEND); what is
    pinned here is the launcher, the rendezvous and the two-rank step, not RCCL."""
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    out = _bench([], {"GCN_BENCH_BACKEND": "gloo"}, gpus=2)
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["config"]["global_batch"] == 8
    assert out["rccl_ranks"] == 0 and out["gradient_allreduce"]["backend"] == "gloo"
    assert out["gradient_allreduce"]["wall_ms_per_step"] > 0
    assert out["value"] > 0 and out["scaling"] == "weak"
