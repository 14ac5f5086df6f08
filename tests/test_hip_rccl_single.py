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


def _bench(extra_args, extra_env, gpus=1, batch="4"):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", "2", "--batch", batch,
           "--skip-cpu", "--skip-beam", "--skip-opt", "--skip-sweep", "--skip-large", "--skip-traffic"] + extra_args
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


def test_bench_measures_the_edge_kernels_hbm_traffic_itself():
    """Without --skip-traffic the run starts two rocprofv3 counter passes of its own and reports their result as
    roofline.traffic: at least the algorithmic bytes (every edge tensor of the model is read or written once), and not
    a multiple of them (re-reads would show here first)."""
    import shutil
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    if shutil.which("rocprofv3") is None:
        pytest.skip("no rocprofv3 on this box")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "20",
                        "--skip-cpu", "--skip-beam", "--skip-opt", "--skip-sweep", "--skip-large"], cwd=ROOT,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")},
                       capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    for key in ("roofline", "roofline_fwd"):
        rf = out[key]
        assert rf["traffic_source"].startswith("two rocprofv3 --pmc passes"), rf
        assert rf["traffic"] > 0 and rf["traffic_stale"] is False
        assert 0.9 <= rf["traffic_over_algorithmic"] <= 3.0, rf


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


def test_bucketed_gradient_allreduce_runs_behind_the_layer_events_on_one_nccl_rank():
    """Gradients of 8 MB or more are averaged in buckets that start while the backward pass still runs: each bucket's
    all-reduce is queued on a side stream behind the completion event of its last layer (gcn_model_bwd_ev).  One RCCL
    rank at H = 300 with the bucket size lowered to 1 MB so that four layers make several buckets; averaging over one
    rank is the identity, so the loss must equal the plain run's (a bucket reduced before its layer was final, or a
    range computed wrongly, would change the gradients only - those are compared in test_hip_dist_trainer - but the
    collectives, events and stream waits all execute here)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    cfgargs = ["--config", "synthetic_100n_50c_h300_l30", "--layers", "4"]
    env = {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())}
    plain = _bench(cfgargs + ["--no-graph"], {}, batch="1")
    ddp = _bench(cfgargs + ["--force-dist", "--bucket-mb", "1"], env, batch="1")
    ar = ddp["gradient_allreduce"]
    assert ddp["rccl_ranks"] == 1 and ar["backend"] == "nccl"
    assert ar["buckets"] is not None and len(ar["buckets"]) >= 3
    assert ar["buckets"][-1][0] == 0 and sum(hi - lo for lo, hi in ar["buckets"]) == ar["bytes"]      # a partition of flat_grad
    assert "eager" in ddp["config"]["workload"]
    assert ddp["loss"] == plain["loss"]
