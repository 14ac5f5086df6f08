"""Data-parallel helpers: one process per GPU, the batch of graph instances is sharded over ranks, the
flat gradient buffer is averaged with ONE all-reduce per step (RCCL over xGMI with backend "nccl";
"gloo" on CPU for tests).  BatchNorm statistics stay per replica, which is what the reference's
``nn.DataParallel`` does (``src/gcn/train/trainer.py:54``)."""
from __future__ import annotations

import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard_bounds(n, rank_, world_):
    """Contiguous shard [lo, hi) of n items for rank_ of world_ (sizes differ by at most one)."""
    base, rem = divmod(n, world_)
    lo = rank_ * base + min(rank_, rem)
    return lo, lo + base + (1 if rank_ < rem else 0)


def shard_batch(batch, rank_=None, world_=None):
    """Slice every array of a batch dict along dim 0; 'edge_cw' (global class weights) is left whole."""
    rank_ = rank() if rank_ is None else rank_
    world_ = world() if world_ is None else world_
    n = len(batch["edges_capacity"])
    lo, hi = shard_bounds(n, rank_, world_)
    return {k: (v if k == "edge_cw" else v[lo:hi]) for k, v in batch.items()}


def _staged(t):
    """gloo cannot take device tensors on this stack (rehearsals of the multi-rank path on one GPU, tests): such
    collectives go through a host copy.  The production backend is nccl (= RCCL), which reduces in place on the GPU."""
    return t.is_cuda and dist.get_backend() != "nccl"


def broadcast_(t, src=0, group=None):
    if active() and (world() > 1 or RUN_SINGLE_RANK_COLLECTIVES):
        if _staged(t):
            c = t.detach().cpu()
            dist.broadcast(c, src=src, group=group)
            t.copy_(c)
        else:
            dist.broadcast(t, src=src, group=group)
    return t


def allreduce_sum_(t, group=None):
    if world() > 1:
        if _staged(t):
            c = t.detach().cpu()
            dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def broadcast_parameters(params, src=0):
    for p in params:
        broadcast_(p.data, src=src)


# Set by callers that want the collectives to execute even in a one-rank process group (bench.py --force-dist:
# the RCCL path then runs on a single GPU exactly as it does on eight).
RUN_SINGLE_RANK_COLLECTIVES = False


def active():
    return dist.is_available() and dist.is_initialized()


def rccl_ranks():
    """Ranks of the process group when it runs on RCCL (backend "nccl"), else 0."""
    return dist.get_world_size() if active() and dist.get_backend() == "nccl" else 0


def allreduce_mean_(flat, group=None):
    """In-place mean over ranks of a flat tensor (the whole gradient in one message).  On RCCL the division is
    part of the collective (ReduceOp.AVG): no separate scaling launch.  Staged (gloo) copies divide on the host."""
    if not active():
        return flat
    w = world()
    if w == 1 and not RUN_SINGLE_RANK_COLLECTIVES:
        return flat
    if flat.is_cuda and dist.get_backend() == "nccl":
        dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=group)
    elif _staged(flat):
        c = flat.detach().cpu()
        dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(c.mul_(1.0 / w))
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.mul_(1.0 / w)
    return flat


def allreduce_max_(t, group=None):
    if active() and (world() > 1 or RUN_SINGLE_RANK_COLLECTIVES):
        if _staged(t):
            c = t.detach().cpu()
            dist.all_reduce(c, op=dist.ReduceOp.MAX, group=group)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t


def allreduce_mean_grads(params, group=None):
    """Mean over ranks of the .grad tensors of params, as one all-reduce of their concatenation (for steps that
    go through autograd and therefore have no flat gradient buffer)."""
    if world() == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    allreduce_mean_(flat, group)
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()
