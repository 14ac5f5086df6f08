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
    go through autograd and therefore have no flat gradient buffer).  Every parameter takes part on every rank, zeros
    standing in for a missing .grad (which then becomes the mean of the others'): the message has the same length on all
    ranks whatever each of them happened to differentiate."""
    if world() == 1:
        return
    params = list(params)
    if not params:
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    allreduce_mean_(flat, group)
    off = 0
    for p in params:
        piece = flat[off:off + p.numel()].view_as(p)
        if p.grad is None:
            p.grad = piece.clone()
        else:
            p.grad.copy_(piece)
        off += p.numel()

class BucketedGradAllreduce:
    """Gradient averaging that starts while the backward pass still runs (the counterpart of DistributedDataParallel's
    buckets for HipTrainStep; the reference's nn.DataParallel, src/gcn/train/trainer.py:54, reduces after backward()).

    The backward pass finishes the layers from the top down and records an event per finished part
    (HipTrainStep.set_done_events / gcn_model_bwd_ev).  Consecutive parts are grouped into buckets of at least
    `bucket_bytes`; `after_step()` queues, per bucket, "wait for the bucket's last part, all-reduce its range of
    flat_grad" on a side stream, so a bucket travels over the links while the layers below it are still being
    differentiated; the last bucket (layer 0 and the embeddings) follows the end of the step.  `wait()` makes the
    current stream wait for all of them.  Below `min_total_bytes` of gradients one message after the step is cheaper
    (latency bound) and `active` is False: call allreduce_mean_(step.flat_grad) instead."""

    def __init__(self, step, bucket_bytes=8 << 20, min_total_bytes=8 << 20, group=None):
        self.step, self.group = step, group
        self.active = bool(active() and step.flat_grad.numel() * 4 >= min_total_bytes and not step.use_graph)
        self.buckets = []           # (lo, hi, index of the event to wait for or None = end of the step)
        if not self.active:
            return
        nl = step.model.num_layers
        ranges = step.layer_grad_ranges()
        order = [nl] + list(range(nl - 1, 0, -1))             # parts in the order they finish: MLP, layers L-1 .. 1
        # the MLP sits behind the layers in flat_grad, so it is a bucket of its own; the layers below it form runs
        self.buckets.append((ranges[nl][0], ranges[nl][1], nl))
        hi, size = None, 0
        for k in order[1:]:
            lo_k, hi_k = ranges[k]
            if hi is None:
                hi = hi_k
            size += (hi_k - lo_k) * 4
            if size >= bucket_bytes:
                self.buckets.append((lo_k, hi, k))
                hi, size = None, 0
        # what is left, layer 0 and the embeddings: final with the step's last kernel
        self.buckets.append((0, hi if hi is not None else ranges[0][1], None))
        need = {b[2] for b in self.buckets if b[2] is not None}
        self.events = [torch.cuda.Event() if k in need else None for k in range(nl + 1)]
        step.set_done_events(self.events)
        self.side = torch.cuda.Stream(step.device)
        self.end = torch.cuda.Event()

    def after_step(self):
        """Call right after step.step(): queues the bucket all-reduces behind their events."""
        if not self.active:
            return allreduce_mean_(self.step.flat_grad, self.group)
        cur = torch.cuda.current_stream(self.step.device)
        self.end.record(cur)
        with torch.cuda.stream(self.side):
            for lo, hi, k in self.buckets:
                self.side.wait_event(self.events[k] if k is not None else self.end)
                allreduce_mean_(self.step.flat_grad[lo:hi], self.group)
        cur.wait_stream(self.side)
        return self.step.flat_grad
