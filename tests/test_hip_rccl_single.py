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


def _bench(extra_args, extra_env):
    env = dict(os.environ)
    env.update(extra_env)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--batch", "4",
           "--skip-cpu", "--skip-beam", "--skip-opt", "--skip-sweep"] + extra_args
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
