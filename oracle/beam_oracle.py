"""CPU oracle for the capacity-aware beam search (TEST INFRASTRUCTURE, never the product path).

Pure-Python / numpy-float32 restatement of src/gcn/algorithms/beamsearch_uelb.py of the
reference.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg
may import it.  A C restatement of the same algorithm, used for timing and for larger cases,
lives in ``oracle/beam_oracle.c``; ``tests/test_oracle_golden.py`` pins both against vectors
produced by the reference itself (``oracle/gen_golden.py``).

The one deliberate difference in mechanics (not in results): the reference clones a V x V
remaining-capacity matrix per surviving candidate and checks ``demand <= remaining[cur, n]``
(beamsearch_uelb.py:161-163).  Paths are simple, so edge (cur, n) cannot have been used earlier
on the same path and ``remaining[cur, n]`` always equals the capacity the commodity started
with.  The restatement therefore checks the entry capacity and subtracts the demand along
the chosen path afterwards.
"""
from __future__ import annotations

import numpy as np

FALLBACK_PATH = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9]  # beamsearch_uelb.py:179-181 (hard-coded 10 nodes)
ALGORITHMS = ("standard", "deterministic", "greedy", "unconstrained")


def _stable_desc(items):
    # sorted(..., key=score, reverse=True) keeps equal scores in insertion order
    # (beamsearch_uelb.py:144,173)
    return sorted(items, key=lambda it: it[2], reverse=True)


def beam_search_commodity(cap, y, source, target, demand, beam_size, check_demand=True):
    """_beam_search_for_commodity (beamsearch_uelb.py:137-177) and its unconstrained twin
    (:417-457).  cap: [V,V] float32 (capacity at the start of this commodity), y: [V,V]
    float32 log-probabilities.  Returns the best path as a list of ints, [] if none."""
    V = cap.shape[0]
    queue = [(source, [source], np.float32(0.0))]
    best = []
    while queue:
        queue = _stable_desc(queue)[:beam_size]
        nxt = []
        for node, path, score in queue:
            if node == target:
                best.append((node, path, score))
                continue
            for n in range(V):
                if n in path:
                    continue
                if cap[node, n] == 0:
                    continue
                new_score = np.float32(score + y[node, n])   # fp32 add, left to right along the path
                if (not check_demand) or demand <= cap[node, n]:
                    nxt.append((n, path + [n], new_score))
        queue = nxt
    if not best:
        return []
    return _stable_desc(best)[0][1]


def search_graph(y, cap, commodities, beam_size, algorithm="standard", mode_strict=False,
                 max_iter=5, perms=None):
    """_search_single_batch of the four algorithms (beamsearch_uelb.py:89-135, 187-232,
    289-335, 387-415) for one graph.

    y: [V,V,C] float32, cap: [V,V] float32, commodities: [C,3] ints (src, dst, demand).
    perms: iterator yielding one permutation of range(C) per attempt (``standard`` and
    ``greedy`` draw ``torch.randperm(C)`` once per attempt, :99, :299).
    Returns (paths[C], is_feasible, attempts_used).
    """
    C = commodities.shape[0]
    if algorithm == "unconstrained":
        paths = [beam_search_commodity(cap, y[:, :, c], int(commodities[c, 0]), int(commodities[c, 1]),
                                       0, beam_size, check_demand=False) for c in range(C)]
        return paths, True, 0
    if algorithm == "greedy":
        beam_size = 1                                        # beamsearch_uelb.py:286-287
    count, attempts = 0, 0
    is_feasible, result = True, None
    while count < max_iter:
        if algorithm == "deterministic":
            order = list(range(C))
        else:
            order = [int(v) for v in next(perms)]
        attempts += 1
        cur = cap.copy()
        found = []
        is_feasible = True
        for c in order:
            s, t, d = (int(v) for v in commodities[c])
            path = beam_search_commodity(cur, y[:, :, c], s, t, d, beam_size)
            if not path:
                break
            if mode_strict:                                  # :123-124 keep the chosen path's remaining capacity
                for u, v in zip(path[:-1], path[1:]):
                    cur[u, v] = np.float32(cur[u, v] - np.float32(d))
            found.append(path)
        if len(found) == C:
            count = max_iter
            result = [None] * C
            for pos, c in enumerate(order):                  # un-permute, :102,129
                result[c] = found[pos]
        else:
            is_feasible = False
            result = [list(FALLBACK_PATH) for _ in range(C)]
            count += 1
    return result, is_feasible, attempts


def search(y, cap, commodities, beam_size, algorithm="standard", mode_strict=False, max_iter=5,
           perms=None):
    """BeamSearchAlgorithm.search (beamsearch_uelb.py:37-55) over a batch.

    y: [B,V,V,C], cap: [B,V,V], commodities: [B,C,3].  Returns (paths[B][C], feasible[B]);
    the reference's own return value is ``feasible[-1]`` (:50,55).
    """
    y = np.asarray(y, dtype=np.float32)
    cap = np.asarray(cap, dtype=np.float32)
    commodities = np.asarray(commodities)
    it = iter(perms) if perms is not None else None
    all_paths, feas = [], []
    for b in range(y.shape[0]):
        p, f, _ = search_graph(y[b], cap[b], commodities[b], beam_size, algorithm, mode_strict, max_iter, it)
        all_paths.append(p)
        feas.append(f)
    return all_paths, feas
