"""Batch iterator yielding the same ``DotDict`` of numpy arrays as the reference's ``DatasetReader``
(``src/common/data_management/dataset_reader.py:25-169``): edges, edges_capacity, edges_target, nodes,
nodes_target, commodities, load_factor.

Two sources:
  * the reference's on-disk layout under ``$GCN_UELB_DATA_ROOT/<dataset>/<mode>_data/`` (per-sample GML graph,
    commodity CSV, node-flow CSV, read with networkx), when it exists;
  * seeded synthetic batches (``config.synthetic_data = true`` or no dataset on disk and
    ``config.allow_synthetic_data = true``), for smoke runs and benchmarks where no dataset can be generated
    (the exact-label ILP solver is not available offline).
"""
from __future__ import annotations

import csv
import os
from pathlib import Path

import numpy as np

from ..synthetic import make_batch

BUCKET_SIZE = 10          # samples per numbered sub-directory (reference src/common/config/paths.py:30)


class DotDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def data_dir_for(config, mode):
    """<data root>/<dataset name>/<mode>_data; the root comes from $GCN_UELB_DATA_ROOT, then config.data_root, then
    ../ml-data/graph-convnet-uelb/datasets (reference src/common/config/paths.py:54-99)."""
    root = os.environ.get("GCN_UELB_DATA_ROOT") or config.get("data_root") or "../ml-data/graph-convnet-uelb/datasets"
    name = config.get("dataset_name") or config.get("expt_name")
    if not name:
        raise ValueError("Config must specify 'dataset_name' (or 'expt_name' as fallback) to locate the dataset directory.")
    return Path(str(root)).expanduser() / str(name) / f"{mode}_data"


def read_exact_solutions(data_dir):
    """First column of exact_solution.csv, one row per sample in index order, or None if the file is absent
    (reference dataset_reader.py:46-57)."""
    f = Path(data_dir) / "exact_solution.csv"
    if not f.exists():
        return None
    with open(f) as fh:
        return [float(row[0]) for row in csv.reader(fh)]


def read_sample(data_dir, i):
    """One sample of the reference's on-disk layout -> (adjacency [V,V] int, capacity [V,V] float64,
    commodities [C][3] int, node order [C][V] int) (reference dataset_reader.py:82-127)."""
    import networkx as nx
    data_dir = Path(data_dir)
    bucket = i - (i % BUCKET_SIZE)
    G = nx.read_gml(str(data_dir / "graph_file" / str(bucket) / f"graph_{i}.gml"), destringizer=int)
    V = len(G.nodes)
    with open(data_dir / "commodity_file" / str(bucket) / f"commodity_data_{i}.csv") as f:
        commodity_list = [[int(r[0]), int(r[1]), int(r[2])] for r in csv.reader(f)]
    cap = np.zeros((V, V))
    for u, v, data in G.edges(data=True):
        cap[u, v] = data.get("capacity", 0)
    with open(data_dir / "node_flow_file" / str(bucket) / f"node_flow_{i}.csv", newline="") as f:
        node_flow = [[int(x) for x in row] for row in csv.reader(f)]
    return nx.adjacency_matrix(G).toarray(), cap, commodity_list, node_flow


def nodes_from_commodities(commodity_list, V):
    """nodes[src, c] = 1, nodes[dst, c] = 2 (reference dataset_reader.py:117-123)."""
    nodes = np.zeros((V, len(commodity_list)), dtype=int)
    for c, (s, t, _) in enumerate(commodity_list):
        nodes[s, c] = 1
        nodes[t, c] = 2
    return nodes


def edges_from_node_order(node_flow, V):
    """target[u, v, c] = 1 for consecutive nodes u, v of commodity c's path, the path being the nodes with a
    positive visiting order sorted by that order (reference dataset_reader.py:132-145)."""
    C = len(node_flow)
    target = np.zeros((V, V, C), dtype=int)
    for c in range(C):
        order = np.asarray(node_flow[c])
        visited = np.nonzero(order > 0)[0]
        seq = visited[np.argsort(order[visited], kind="stable")]
        for u, v in zip(seq[:-1], seq[1:]):
            target[u, v, c] = 1
    return target


class DatasetReader:
    def __init__(self, num_data, batch_size, mode, config, shuffle=None, seed=None):
        self.num_data, self.batch_size, self.mode, self.config = num_data, batch_size, mode, config
        self.max_iter = num_data // batch_size
        self.shuffle = (mode == "train") if shuffle is None else shuffle
        self.seed = seed
        self.synthetic = bool(config.get("synthetic_data", False))
        self.data_dir = None
        if not self.synthetic:
            self.data_dir = data_dir_for(config, mode)
            self.synthetic = not self.data_dir.exists() and bool(config.get("allow_synthetic_data", False))
        if not self.synthetic and not self.data_dir.exists():
            raise FileNotFoundError(f"dataset directory {self.data_dir} not found (set GCN_UELB_DATA_ROOT, or "
                                    f"synthetic_data=true in the config)")
        self._epoch = 0
        self._load_factors = None

    def __iter__(self):
        rng = np.random.default_rng(self.seed)
        indices = np.arange(self.num_data)
        if self.shuffle:
            rng.shuffle(indices)
        for it in range(self.max_iter):
            yield self.process_batch(indices[it * self.batch_size:(it + 1) * self.batch_size])

    # ------------------------------------------------------------------ synthetic
    def _synthetic_batch(self, idx):
        c = self.config
        topo = "nsfnet" if c.get("graph_model", "nsfnet") == "nsfnet" and c.num_nodes == 14 else "random"
        b = make_batch(len(idx), c.num_nodes, c.num_commodities, topo, c.get("capacity_lower", 1000),
                       c.get("capacity_higher", 5000), c.get("demand_lower", 1), c.get("demand_higher", 500),
                       seed=int(idx[0]) * 7919 + {"train": 0, "val": 1, "test": 2}.get(self.mode, 3))
        out = DotDict(edges=b["edges"], edges_capacity=b["edges_capacity"].astype(np.float64), edges_target=b["edges_target"],
                      nodes=b["nodes"], commodities=b["commodities"],
                      nodes_target=np.zeros((len(idx), c.num_commodities, c.num_nodes), dtype=np.int64),
                      load_factor=np.full(len(idx), np.nan))
        return out

    # ------------------------------------------------------------------ on-disk format of the reference
    def process_batch(self, batch_indices):
        if self.synthetic:
            return self._synthetic_batch(batch_indices)
        edges, caps, targets, nodes_l, node_targets, comms = [], [], [], [], [], []
        for i in batch_indices:
            adj, cap, commodity_list, node_flow = read_sample(self.data_dir, int(i))
            V = cap.shape[0]
            edges.append(adj); caps.append(cap); targets.append(edges_from_node_order(node_flow, V))
            nodes_l.append(nodes_from_commodities(commodity_list, V)); node_targets.append(node_flow); comms.append(commodity_list)
        batch = DotDict()
        batch.edges = np.stack(edges)
        batch.edges_capacity = np.stack(caps)
        batch.edges_target = np.stack(targets)
        batch.nodes = np.stack(nodes_l)
        batch.nodes_target = np.stack(node_targets)
        batch.commodities = np.stack(comms)
        if self._load_factors is None:
            self._load_factors = read_exact_solutions(self.data_dir) or []
        batch.load_factor = (np.asarray([self._load_factors[int(i)] for i in batch_indices]) if self._load_factors
                             else np.full(len(batch_indices), np.nan))
        return batch
