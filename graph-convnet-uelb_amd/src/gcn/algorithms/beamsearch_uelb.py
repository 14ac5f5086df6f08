"""Capacity-aware beam search, decoded on the GPU by libgcn_uelb_hip.so.

Keeps the reference's API (``src/gcn/algorithms/beamsearch_uelb.py``): the four algorithm classes,
``BeamSearchFactory.create_algorithm(name, **kwargs)``, ``search() -> (paths, is_feasible)``,
``get_performance_info()``, the attributes ``y``, ``execution_time``, ``algorithm_name`` and the
``_search_single_batch(batch)`` hook for subclasses, plus the alias ``BeamsearchUELB``.

All graphs of the batch are decoded by one kernel launch (one wavefront per graph; per
(graph, commodity) for the unconstrained algorithm).  Given the same log-probabilities ``y`` and the
same permutations, the paths are bit-identical to the reference's.

``standard`` and ``greedy`` draw ``torch.randperm(C)`` from the global CPU generator once per attempt,
graph after graph (reference :99, :299), so graph b's first permutation sits at stream offset
sum(attempts of the graphs before it).  To stay exact while running graphs in parallel the
permutations are pre-drawn from a clone of the generator, every graph is run assuming one attempt
each, and only graphs whose offset turns out to differ are re-run until the offsets are consistent;
the global generator is then advanced by exactly the number of permutations the reference would have
drawn.
"""
from __future__ import annotations

import time
from abc import ABC
from typing import List, Tuple

import torch

from ..hip import lib as L

ALGO_ID = {"standard": 0, "deterministic": 1, "greedy": 2, "unconstrained": 3}


def _device_of(t):
    if t.is_cuda:
        return t.device
    if not torch.cuda.is_available():
        raise L.GcnHipError("beam search needs a ROCm GPU (inputs are on CPU and no GPU is available); "
                            "there is no CPU fallback in this package")
    return torch.device("cuda", torch.cuda.current_device())


class BeamSearchAlgorithm(ABC):
    """Base class: prepares the scores and runs the batched GPU search."""

    algorithm_key = "standard"

    def __init__(self, y_pred_edges, beam_size, batch_size, edges_capacity, commodities,
                 dtypeFloat, dtypeLong, mode_strict=False, max_iter=5):
        lib = L.load()
        dev = _device_of(y_pred_edges)
        self.device = dev
        logits = y_pred_edges.detach().to(dev, torch.float32).contiguous()
        if logits.dim() == 5:
            if logits.shape[4] != 2:
                raise ValueError("5-D predictions must have 2 classes in the last dim")
            B, V, _, C, _ = logits.shape
            five_d = 1
        elif logits.dim() == 4:
            B, V, _, C = logits.shape
            five_d = 0
        else:
            raise ValueError(f"y_pred_edges must be 4-D or 5-D, got {logits.dim()}-D")
        y = torch.empty((B, V, V, C), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            L.check(lib.gcn_beam_scores(B, V, C, five_d, L.ptr(logits), L.ptr(y), L.stream_ptr(dev)), "gcn_beam_scores")
        self.y = y
        self.beam_size = beam_size
        self.batch_size = batch_size
        self.edges_capacity = edges_capacity
        self.commodities = commodities
        self.dtypeFloat = dtypeFloat
        self.dtypeLong = dtypeLong
        self.mode_strict = mode_strict
        self.max_iter = max_iter
        self.execution_time = 0.0
        self.algorithm_name = self.__class__.__name__
        # results of the last search(), kept on the device
        self.paths_tensor = None        # int32 [B, C, max(V, 10)], -1 padded
        self.path_len = None            # int32 [B, C]
        self.feasible_per_graph = None  # bool  [B]
        self.attempts = None            # int32 [B]

    # ------------------------------------------------------------------ GPU driver
    def _run(self, graphs=None):
        """Decode graphs (default: the first batch_size) -> (paths, path_len, feasible, attempts) on device."""
        lib = L.load()
        dev = self.device
        y = self.y
        Btot, V, _, C = y.shape
        sel = list(range(self.batch_size)) if graphs is None else list(graphs)
        idx = torch.as_tensor(sel, dtype=torch.long, device=dev)
        full = len(sel) == Btot and sel == list(range(Btot))
        yb = y if full else y.index_select(0, idx).contiguous()
        cap = self.edges_capacity.detach().to(dev, torch.float32)
        cap = (cap if full else cap.index_select(0, idx)).contiguous()
        comm = self.commodities.detach().to(dev, torch.int64)
        comm = (comm if full else comm.index_select(0, idx)).contiguous()
        B = len(sel)
        algo = ALGO_ID[self.algorithm_key]
        K = int(self.beam_size)
        PL = max(V, 10)
        paths = torch.empty((B, C, PL), dtype=torch.int32, device=dev)
        plen = torch.empty((B, C), dtype=torch.int32, device=dev)
        feas = torch.empty(B, dtype=torch.int32, device=dev)
        att = torch.empty(B, dtype=torch.int32, device=dev)
        ws_bytes = lib.gcn_beam_ws_bytes(B, V, C, K)
        ws = L.workspace(dev, ws_bytes)
        needs_perms = algo in (0, 2)

        def launch(perms, n_perms, start, mask):
            with torch.cuda.device(dev):
                rc = lib.gcn_beam_search(B, V, C, K, algo, int(bool(self.mode_strict)), int(self.max_iter), L.ptr(yb),
                                         L.ptr(cap), L.ptr(comm), L.ptr(perms), n_perms, L.ptr(start), L.ptr(mask),
                                         L.ptr(paths), L.ptr(plen), L.ptr(feas), L.ptr(att), L.ptr(ws), ws.numel(),
                                         L.stream_ptr(dev))
            L.check(rc, "gcn_beam_search")

        if not needs_perms:
            launch(None, 0, None, None)
            return paths, plen, feas, att
        # pre-draw what torch.randperm(C) would return, without disturbing the global generator yet
        gen = torch.Generator()
        gen.set_state(torch.get_rng_state())
        n_perms = B * int(self.max_iter)
        perms_cpu = torch.stack([torch.randperm(C, generator=gen) for _ in range(n_perms)]).to(torch.int32)
        perms = perms_cpu.to(dev)
        start_cpu = torch.arange(B, dtype=torch.int32)
        mask_cpu = torch.ones(B, dtype=torch.uint8)
        while True:
            launch(perms, n_perms, start_cpu.to(dev), mask_cpu.to(dev))
            att_cpu = att.cpu()
            new_start = torch.zeros(B, dtype=torch.int32)
            new_start[1:] = torch.cumsum(att_cpu, 0)[:-1].to(torch.int32)
            changed = new_start != start_cpu
            if not bool(changed.any()):
                break
            start_cpu = new_start
            mask_cpu = changed.to(torch.uint8)
        # consume exactly the permutations the reference would have drawn
        for _ in range(int(att_cpu.sum().item())):
            torch.randperm(C)
        return paths, plen, feas, att

    @staticmethod
    def _to_lists(paths, plen):
        p, n = paths.cpu().tolist(), plen.cpu().tolist()
        return [[row[:n[b][c]] for c, row in enumerate(graph)] for b, graph in enumerate(p)]

    # ------------------------------------------------------------------ reference API
    def search(self) -> Tuple[List[List[List[int]]], bool]:
        """Decode every graph of the batch.  Returns (paths[B][C], is_feasible of the LAST graph), as the
        reference does (:49-55); per-graph feasibility is in ``feasible_per_graph``."""
        start = time.time()
        if type(self)._search_single_batch is not BeamSearchAlgorithm._search_single_batch:
            # a subclass supplied its own per-graph routine (reference hook): keep the reference's loop
            all_paths, feas, is_feasible = [], [], True
            for b in range(self.batch_size):
                _, p, is_feasible = self._search_single_batch(b)
                all_paths.append(p)
                feas.append(bool(is_feasible))
            self.feasible_per_graph = torch.tensor(feas)
            self.execution_time = time.time() - start
            return all_paths, is_feasible
        paths, plen, feas, att = self._run()
        self.paths_tensor, self.path_len, self.attempts = paths, plen, att
        self.feasible_per_graph = feas.bool()
        out = self._to_lists(paths, plen)
        is_feasible = bool(feas[-1].item())
        self.execution_time = time.time() - start
        return out, is_feasible

    def _search_single_batch(self, batch: int):
        """Decode one graph -> (node_orders, commodity_paths, is_feasible)."""
        paths, plen, feas, _ = self._run([batch])
        commodity_paths = self._to_lists(paths, plen)[0]
        V = self.y.shape[1]
        node_orders = []
        for p in commodity_paths:
            order = [0] * V
            if bool(feas[0].item()) or self.algorithm_key == "unconstrained":
                for i, node in enumerate(p):
                    order[node] = i + 1
            node_orders.append(order)
        return node_orders, commodity_paths, bool(feas[0].item())

    def get_performance_info(self) -> dict:
        return {
            'algorithm_name': self.algorithm_name,
            'execution_time': self.execution_time,
            'beam_size': self.beam_size,
            'batch_size': self.batch_size,
            'max_iter': self.max_iter,
        }


class StandardBeamSearch(BeamSearchAlgorithm):
    """Random commodity order per attempt, up to max_iter attempts (reference :86-181)."""
    algorithm_key = "standard"


class DeterministicBeamSearch(BeamSearchAlgorithm):
    """Commodities in their given order (reference :184-278)."""
    algorithm_key = "deterministic"


class GreedyBeamSearch(BeamSearchAlgorithm):
    """Standard search with the beam size forced to 1 (reference :281-381)."""
    algorithm_key = "greedy"

    def __init__(self, y_pred_edges, beam_size, batch_size, edges_capacity, commodities,
                 dtypeFloat, dtypeLong, mode_strict=False, max_iter=5):
        super().__init__(y_pred_edges, 1, batch_size, edges_capacity, commodities,
                         dtypeFloat, dtypeLong, mode_strict, max_iter)


class UnconstrainedBeamSearch(BeamSearchAlgorithm):
    """No demand check, no capacity update, always reported feasible (reference :384-457)."""
    algorithm_key = "unconstrained"


class BeamSearchFactory:
    """name -> algorithm class (reference :460-495)."""

    ALGORITHMS = {
        'standard': StandardBeamSearch,
        'deterministic': DeterministicBeamSearch,
        'greedy': GreedyBeamSearch,
        'unconstrained': UnconstrainedBeamSearch,
    }

    @classmethod
    def create_algorithm(cls, algorithm_name: str, **kwargs) -> BeamSearchAlgorithm:
        if algorithm_name not in cls.ALGORITHMS:
            available = ', '.join(cls.ALGORITHMS.keys())
            raise ValueError(f"invalid algorithm name: {algorithm_name}. available: {available}")
        return cls.ALGORITHMS[algorithm_name](**kwargs)

    @classmethod
    def get_available_algorithms(cls) -> List[str]:
        return list(cls.ALGORITHMS.keys())


class BeamsearchUELB(StandardBeamSearch):
    """Backward-compatible alias used by the evaluator (reference :499-501)."""
    pass
