"""Data-parallel plumbing on CPU with two gloo ranks: batch sharding and the single flat-gradient all-reduce."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG  # noqa: F401  (puts the package on sys.path)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    import sys
    sys.path.insert(0, PKG)
    from src.common.synthetic import make_batch
    from src.gcn.hip import dist as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        batch = make_batch(5, 14, 3, seed=11)                 # same global batch on every rank
        shard = D.shard_batch(batch)
        lo, hi = D.shard_bounds(5, rank, world)
        assert shard["edges_capacity"].shape[0] == hi - lo
        assert np.array_equal(shard["commodities"], batch["commodities"][lo:hi])
        assert np.array_equal(shard["edge_cw"], batch["edge_cw"])          # class weights stay global
        # parameters: rank 0's values everywhere
        p = torch.nn.Parameter(torch.full((7,), float(rank + 1)))
        D.broadcast_parameters([p])
        assert torch.equal(p.data, torch.ones(7))
        # gradients: one all-reduce over the flat buffer, mean over ranks
        flat = torch.arange(10, dtype=torch.float32) * (rank + 1)
        views = [flat[:4].view(2, 2), flat[4:]]
        D.allreduce_mean_(flat)
        want = torch.arange(10, dtype=torch.float32) * (1 + 2) / 2
        assert torch.allclose(flat, want)
        assert torch.allclose(views[0], want[:4].view(2, 2))                # p.grad views see the reduced values
        # per-parameter gradients of an autograd step: averaged through one concatenated message
        q = [torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(2, 2)), torch.nn.Parameter(torch.zeros(1))]
        q[0].grad = torch.full((3,), float(rank))
        q[1].grad = torch.full((2, 2), 2.0 * rank + 1.0)
        if rank == 1:
            q[2].grad = torch.full((1,), 4.0)     # a parameter only one rank differentiated: the message still has one length
        D.allreduce_mean_grads(q)
        assert torch.allclose(q[0].grad, torch.full((3,), 0.5)) and torch.allclose(q[1].grad, torch.full((2, 2), 2.0))
        assert q[2].grad is not None and torch.allclose(q[2].grad, torch.full((1,), 2.0))      # zeros stood in on rank 0
        out.put((rank, "ok"))
    except Exception as ex:  # pragma: no cover
        out.put((rank, repr(ex)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharding_and_allreduce():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_shard_bounds_cover_everything():
    from src.gcn.hip import dist as D
    for n in (1, 5, 20, 33):
        for w in (1, 2, 3, 8):
            spans = [D.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
