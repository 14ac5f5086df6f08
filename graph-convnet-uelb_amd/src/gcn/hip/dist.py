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
    if world() > 1:
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
    if world() > 1:
        for p in params:
            broadcast_(p.data, src=src)


def allreduce_mean_(flat, group=None):
    """In-place mean over ranks of a flat tensor (the whole gradient in one message)."""
    w = world()
    if w > 1:
        allreduce_sum_(flat, group)
        flat.mul_(1.0 / w)
    return flat


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
