"""CPU oracle for the evaluation metrics computed right after decoding (TEST INFRASTRUCTURE, never the product path).

Plain Python / numpy restatement of
  * ``mean_feasible_load_factor`` (reference ``src/common/graph/graph_utils.py:90-192``) and
  * ``Evaluator._compute_completion_and_approx`` (reference ``src/gcn/train/evaluator.py:20-79``).

Parity status.  The load factor is PINNED: the same quantity is part of the reference's REINFORCE golden vectors
(``tests/golden/rl_*.npz``: ``load_factor_individual``, ``complete_paths_rate``), which the GPU path reproduces.
The completion / approximation rates are PARITY UNPINNED: the reference's ``evaluator.py`` cannot be imported in the
build container (it imports ``fastprogress``, which is not installed, at module level) and the reference ships no
fixture for these numbers, so this file is a line-by-line restatement checked by reading only.
"""
from __future__ import annotations

import numpy as np


def edge_usage(paths, demands, V):
    """Demand summed along every hop of every path of one graph (evaluator.py:56-63 / graph_utils.py:119-145).
    Hops with a node outside [0, V) are skipped, as evaluator.py:62 does."""
    usage = np.zeros((V, V), dtype=np.float64)
    for path, d in zip(paths, demands):
        for u, v in zip(path[:-1], path[1:]):
            if 0 <= u < V and 0 <= v < V:
                usage[u, v] += float(d)
    return usage


def max_load_factor(paths, commodities, cap):
    """Per-graph value of mean_feasible_load_factor (graph_utils.py:160-174): usage / capacity, +inf where a
    zero-capacity edge carries demand, 0 where an unused edge has zero capacity; maximum over the edges."""
    cap = np.asarray(cap, dtype=np.float32)
    V = cap.shape[0]
    usage = edge_usage(paths, commodities[:, 2], V).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        lf = np.where(cap == 0, np.where(usage == 0, np.float32(0), np.float32(np.inf)), usage / cap)
    return float(lf.max())


def completion_and_approx(pred_paths, batch_commodities, gt_load_factors, edge_capacity):
    """Evaluator._compute_completion_and_approx (evaluator.py:20-79).

    Comp% = commodities whose path ends on their destination; CompSample% = graphs all of whose commodities do;
    Approx% = mean over those graphs of gt_lf / model_lf * 100, where model_lf is the maximum of usage / capacity
    over the edges with capacity > 0 ONLY (a used zero-capacity edge does not make it infinite, :64-65), graphs
    with model_lf <= 1e-8 or gt_lf <= 0 left out (:73-76); 0.0 when no graph qualifies (:77)."""
    B = len(pred_paths)
    C = batch_commodities.shape[1]
    total = done = 0
    complete_mask, model_lf = [], []
    for b in range(B):
        all_complete = True
        for c in range(C):
            total += 1
            path = pred_paths[b][c] if c < len(pred_paths[b]) else []
            if len(path) > 0 and path[-1] == int(batch_commodities[b, c, 1]):
                done += 1
            else:
                all_complete = False
        complete_mask.append(all_complete)
        cap_b = np.asarray(edge_capacity[b], dtype=np.float32)
        usage = edge_usage([pred_paths[b][c] if c < len(pred_paths[b]) else [] for c in range(C)],
                           batch_commodities[b, :, 2], cap_b.shape[0]).astype(np.float32)
        valid = cap_b > 0
        model_lf.append(float((usage[valid] / cap_b[valid]).max()) if valid.any() else 0.0)
    comp = done / total * 100 if total > 0 else 0.0
    comp_sample = sum(complete_mask) / B * 100 if B > 0 else 0.0
    ratios = []
    for b in range(B):
        if complete_mask[b] and model_lf[b] > 1e-8:
            gt = float(gt_load_factors[b])
            if gt > 0:
                ratios.append(gt / model_lf[b] * 100)
    return comp, comp_sample, (float(np.mean(ratios)) if ratios else 0.0)
