"""Stochastic path sampling for REINFORCE training, run on the GPU by libgcn_uelb_hip.so.

Keeps the reference's API (``src/gcn/algorithms/path_sampler.py``): ``PathSampler(y_pred_edges,
edges_capacity, commodities, num_samples, temperature, top_p, entropy_epsilon, dtypeFloat, dtypeLong)``,
``sample() -> (batch_paths, log_probs_tensor, is_feasible, batch_stepwise_entropies)`` and the
``reachability`` attribute.  ``log_probs_tensor`` ([B], one value per graph) is attached to
``y_pred_edges`` through ``PathLogProbFn``, so ``(-log_probs * advantages).mean().backward()`` reaches the
model exactly as in the reference (``src/gcn/training/reinforcement_strategy.py:427-428``).

All (graph, commodity) pairs are sampled by one kernel launch.  The draws come from a counter-based generator
keyed by (seed, call counter, graph, commodity, step): the seed is ``torch.initial_seed()`` with the process
rank mixed in and the counter advances by one per ``sample()`` call, so a run is reproducible from
``torch.manual_seed(s)`` followed by ``src.gcn.hip.lib.reseed()`` (which restarts the counter), but it does not
consume torch's generator the way the reference's ``torch.multinomial`` calls do; the distribution of the paths
is the same.
"""
from __future__ import annotations

import torch

from ..hip import lib as L

_calls = {}


def reset_call_counters():
    """Restart the draw streams (called by ``lib.reseed``): the next ``sample()`` uses call counter 0 again."""
    _calls.clear()


L.register_reseed_hook(reset_call_counters)


def _rng_words(dev):
    """{seed, call counter} on the device.  The seed is ``lib.base_seed()`` (torch's initial seed with the rank of
    this process mixed in, so that ranks seeded alike draw differently for their different graphs); the counter
    advances with every call and starts again at 0 after ``lib.reseed()`` - calling ``torch.manual_seed`` alone does
    not restart it."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), L.base_seed())
    n = _calls.get(key, 0)
    _calls[key] = n + 1
    return torch.tensor([L.base_seed(), n], dtype=torch.int64, device=dev)


def _commodity_tensor(commodities, B, dev):
    """[B, C, 3] int64 (src, dst, demand) from a tensor or from one list of tuples shared by all graphs."""
    if isinstance(commodities, torch.Tensor):
        t = commodities.detach()
        if t.dim() == 2:
            t = t.unsqueeze(0).expand(B, -1, -1)
    else:
        t = torch.tensor([[int(s), int(d), int(q)] for s, d, q in commodities]).unsqueeze(0).expand(B, -1, -1)
    return t.to(dev, torch.int64).contiguous()


class PathLogProbFn(torch.autograd.Function):
    """Samples (or replays) paths and returns their per-graph log-probabilities, differentiable in the logits."""

    @staticmethod
    def forward(ctx, y, cap, comm, temperature, top_p, eps, rng, given):
        lib = L.load()
        dev = y.device
        B, V, _, C = y.shape
        PL = max(V, 10)
        if given is None:
            paths = torch.empty((B, C, PL), dtype=torch.int32, device=dev)
            plen = torch.empty((B, C), dtype=torch.int32, device=dev)
        else:
            paths, plen = given
        logp = torch.empty((B, C), dtype=torch.float32, device=dev)
        ent = torch.zeros((B, C, PL), dtype=torch.float32, device=dev)
        nent = torch.empty((B, C), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.gcn_path_sample(B, V, C, temperature, top_p, eps, L.ptr(y), L.ptr(cap), L.ptr(comm), L.ptr(rng),
                                     int(given is not None), L.ptr(paths), L.ptr(plen), L.ptr(logp), L.ptr(ent),
                                     L.ptr(nent), L.stream_ptr(dev))
        L.check(rc, "gcn_path_sample")
        ctx.save_for_backward(y, cap, comm, paths, plen)
        ctx.cfg = (B, V, C, temperature, eps)
        ctx.mark_non_differentiable(paths, plen, ent, nent, logp)
        return logp.sum(1), paths, plen, ent, nent, logp

    @staticmethod
    def backward(ctx, dlogp, *_unused):
        lib = L.load()
        y, cap, comm, paths, plen = ctx.saved_tensors
        B, V, C, temperature, eps = ctx.cfg
        dev = y.device
        dy = torch.zeros_like(y)
        dlogp = dlogp.to(torch.float32).contiguous()
        with torch.cuda.device(dev):
            rc = lib.gcn_path_sample_bwd(B, V, C, temperature, eps, L.ptr(y), L.ptr(cap), L.ptr(comm), L.ptr(paths),
                                         L.ptr(plen), L.ptr(dlogp), L.ptr(dy), L.stream_ptr(dev))
        L.check(rc, "gcn_path_sample_bwd")
        return dy, None, None, None, None, None, None, None


class PathSampler:
    """Samples one path per commodity from the edge scores, respecting connectivity but not capacities."""

    def __init__(self, y_pred_edges, edges_capacity, commodities, num_samples=1, temperature=1.0, top_p=0.8,
                 entropy_epsilon: float = 0.0, dtypeFloat=torch.float, dtypeLong=torch.long):
        L.load()
        L.require_cuda(y_pred_edges, "y_pred_edges")
        if y_pred_edges.dim() != 4:
            raise ValueError(f"y_pred_edges must be [batch, nodes, nodes, commodities], got {tuple(y_pred_edges.shape)}")
        self.y_pred_edges = y_pred_edges
        self.edges_capacity = edges_capacity
        self.commodities = commodities
        self.num_samples = num_samples
        self.temperature = temperature
        self.top_p = top_p
        self.entropy_epsilon = float(entropy_epsilon) if entropy_epsilon is not None else 0.0
        self.dtypeFloat = dtypeFloat
        self.dtypeLong = dtypeLong
        self.batch_size = y_pred_edges.shape[0]
        self.num_nodes = y_pred_edges.shape[1]
        self.device = y_pred_edges.device
        self._cap = edges_capacity.detach().to(self.device, torch.float32).contiguous()
        self._comm = _commodity_tensor(commodities, self.batch_size, self.device)
        if self._comm.shape[1] != y_pred_edges.shape[3]:
            raise ValueError(f"{self._comm.shape[1]} commodities for predictions with {y_pred_edges.shape[3]}")
        self._reach = None
        self.last_rng = None
        # results of the last sample(), kept on the device
        self.paths_tensor = None        # int32 [B, C, max(V, 10)], -1 padded
        self.path_len = None            # int32 [B, C]
        self.log_probs_per_commodity = None
        self.max_load_factor = None     # float [B]
        self.complete = None            # bool  [B, C]: path ends at its destination

    @property
    def reachability(self):
        """bool [B, V, V]: reachability[b, i, j] is True if j can be reached from i over edges with capacity."""
        if self._reach is None:
            lib = L.load()
            B, V = self.batch_size, self.num_nodes
            r = torch.empty((B, V, V), dtype=torch.uint8, device=self.device)
            with torch.cuda.device(self.device):
                L.check(lib.gcn_reachability(B, V, L.ptr(self._cap), L.ptr(r), L.stream_ptr(self.device)), "gcn_reachability")
            self._reach = r.bool()
        return self._reach

    def _run(self, given=None):
        y = self.y_pred_edges
        if y.dtype != torch.float32 or not y.is_contiguous():
            y = y.float().contiguous()
        rng = None if given is not None else _rng_words(self.device)
        self.last_rng = None if rng is None else tuple(int(v) for v in rng.tolist())   # (seed, call counter)
        return PathLogProbFn.apply(y, self._cap, self._comm, float(self.temperature), float(self.top_p),
                                   float(self.entropy_epsilon), rng, given)

    def _finish(self, logp_b, paths, plen, ent, nent, logp):
        lib = L.load()
        dev = self.device
        B, V, C = self.batch_size, self.num_nodes, self._comm.shape[1]
        self.paths_tensor, self.path_len, self.log_probs_per_commodity = paths, plen, logp
        usage = torch.empty(B * V * V, dtype=torch.int32, device=dev)
        max_lf = torch.empty(B, dtype=torch.float32, device=dev)
        complete = torch.empty(B * C, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = lib.gcn_load_factor(B, V, C, paths.shape[2], L.ptr(paths), L.ptr(plen), L.ptr(self._cap), L.ptr(self._comm),
                                     L.ptr(usage), L.ptr(max_lf), L.ptr(complete), L.stream_ptr(dev))
        L.check(rc, "gcn_load_factor")
        self.max_load_factor, self.complete = max_lf, complete.view(B, C).bool()
        p, n = paths.cpu().tolist(), plen.cpu().tolist()
        batch_paths = [[row[:n[b][c]] for c, row in enumerate(graph)] for b, graph in enumerate(p)]
        e, ne = ent.cpu().tolist(), nent.cpu().tolist()
        entropies = [[row[:ne[b][c]] for c, row in enumerate(graph)] for b, graph in enumerate(e)]
        # total edge usage within capacity on every graph (reference :428-446): demands and usage are integers
        is_feasible = bool((max_lf <= 1.0).all().item())
        return batch_paths, logp_b, is_feasible, entropies

    def sample(self):
        """Returns (batch_paths[b][c], log_probs_tensor [B], is_feasible, batch_stepwise_entropies[b][c][step])."""
        return self._finish(*self._run())

    def evaluate(self, paths_tensor, path_len):
        """Log-probabilities and entropies of given paths (int32 [B,C,max(V,10)] / [B,C] on the device), in the
        format of sample().  Not in the reference; used to replay its sampled paths."""
        given = (paths_tensor.to(self.device, torch.int32).contiguous(), path_len.to(self.device, torch.int32).contiguous())
        return self._finish(*self._run(given))
