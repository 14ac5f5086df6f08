"""Seeded synthetic UELB batches with the shapes and value ranges of the reference's configs.

Used by bench.py, the smoke test and the GPU tests; no dataset exists offline.  Topologies:
NSFNet (14 nodes, 21 undirected links; the link list is the public NSFNET T1 backbone as used by the
reference's generator, src/common/graph/graph_making.py:79-92) and a ring with random chords for other
sizes (the reference uses a Newman-Watts-Strogatz graph with k=3, p=0.15, :59-69; same degree
statistics, equal capacity in both directions).  Commodities are distinct ordered node pairs sorted by
demand descending (src/common/graph/graph_utils.py:269-281); node features follow
src/common/data_management/dataset_reader.py:119-123; labels are the edges of a BFS shortest path per
commodity (exact ILP labels cannot be produced offline; label values do not change the cost of a step).
"""
from __future__ import annotations

from collections import deque

import numpy as np

NSFNET_LINKS = [(0, 1), (0, 2), (0, 7), (1, 2), (1, 3), (2, 5), (3, 4), (3, 10), (4, 5), (4, 6), (5, 9), (5, 12),
                (6, 7), (7, 8), (8, 9), (8, 11), (8, 13), (10, 11), (10, 13), (11, 12), (12, 13)]


def _links(rng, V, topology):
    if topology == "nsfnet":
        if V != 14:
            raise ValueError("the NSFNet topology has 14 nodes")
        return list(NSFNET_LINKS)
    links = {(i, (i + 1) % V) for i in range(V)} if V > 2 else {(0, 1)} if V == 2 else set()
    for i in range(V):                       # k = 3 ring lattice joins each node to its nearest neighbour only;
        if rng.random() < 0.15 and V > 3:    # with probability p one shortcut per node is added
            j = int(rng.integers(0, V))
            if j != i and (i, j) not in links and (j, i) not in links:
                links.add((i, j))
    return sorted(links)


def _bfs_path(adj, s, t):
    prev = {s: None}
    dq = deque([s])
    while dq:
        u = dq.popleft()
        if u == t:
            break
        for v in adj[u]:
            if v not in prev:
                prev[v] = u
                dq.append(v)
    if t not in prev:
        return []
    path = [t]
    while prev[path[-1]] is not None:
        path.append(prev[path[-1]])
    return path[::-1]


def make_batch(B, V, C, topology="nsfnet", cap_lo=1000, cap_hi=5000, dem_lo=5, dem_hi=500, seed=0):
    """Returns a dict of numpy arrays: edges [B,V,V] i64, edges_capacity [B,V,V] f32, nodes [B,V,C] i64,
    commodities [B,C,3] i64, edges_target [B,V,V,C] i64, edge_cw [2] f32."""
    rng = np.random.default_rng(seed)
    cap = np.zeros((B, V, V), np.float32)
    comm = np.zeros((B, C, 3), np.int64)
    nodes = np.zeros((B, V, C), np.int64)
    target = np.zeros((B, V, V, C), np.int64)
    if C > V * (V - 1):
        raise ValueError(f"{C} commodities exceed the {V * (V - 1)} ordered node pairs")
    for b in range(B):
        adj = [[] for _ in range(V)]
        for (i, j) in _links(rng, V, topology):
            c = float(rng.integers(cap_lo, cap_hi))
            cap[b, i, j] = cap[b, j, i] = c
            adj[i].append(j)
            adj[j].append(i)
        for a in adj:
            a.sort()
        seen, rows = set(), []
        while len(rows) < C:
            s, t = (int(v) for v in rng.integers(0, V, size=2))
            if s == t or (s, t) in seen:
                continue
            seen.add((s, t))
            rows.append([s, t, int(rng.integers(dem_lo, dem_hi + 1))])
        rows.sort(key=lambda r: -r[2])
        comm[b] = np.asarray(rows)
        for c, (s, t, _) in enumerate(rows):
            nodes[b, s, c] = 1
            nodes[b, t, c] = 2
            p = _bfs_path(adj, s, t)
            for u, v in zip(p[:-1], p[1:]):
                target[b, u, v, c] = 1
    n, n1 = target.size, int(target.sum())
    n1 = max(n1, 1)
    edge_cw = np.asarray([n / (2.0 * max(n - n1, 1)), n / (2.0 * n1)], np.float32)   # sklearn "balanced"
    return dict(edges=(cap > 0).astype(np.int64), edges_capacity=cap, nodes=nodes, commodities=comm,
                edges_target=target, edge_cw=edge_cw)


# shapes of the configurations named by the benchmark definition
CONFIGS = {
    "default2": dict(V=40, C=5, H=64, L=10, B=20, topology="random", cap_lo=1000, cap_hi=10000, dem_lo=1, dem_hi=500),
    "nsfnet_10_commodities": dict(V=14, C=10, H=64, L=10, B=20, topology="nsfnet", cap_lo=1000, cap_hi=5000, dem_lo=5, dem_hi=500),
    "nsfnet_20_commodities": dict(V=14, C=20, H=64, L=10, B=20, topology="nsfnet", cap_lo=2000, cap_hi=10000, dem_lo=5, dem_hi=500),
    # the two large synthetic shapes (graphs per GPU chosen to fit one 288 GB device with room to spare)
    "synthetic_100n_50c_h300_l30": dict(V=100, C=50, H=300, L=30, B=2, topology="random", cap_lo=1000, cap_hi=10000, dem_lo=1, dem_hi=500),
    "synthetic_200n_100c": dict(V=200, C=100, H=64, L=10, B=4, topology="random", cap_lo=1000, cap_hi=10000, dem_lo=1, dem_hi=500),
}
