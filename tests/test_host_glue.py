"""Host-side pieces that need no GPU: config overlay, dataset reader (synthetic and the reference's on-disk
layout), synthetic generator invariants."""
import csv
import json
import os

import numpy as np
import pytest

from conftest import PKG


def test_config_overlay_and_attribute_access(tmp_path, monkeypatch):
    from src.common.config.config import Settings, get_config
    monkeypatch.chdir(PKG)
    cfg = get_config(os.path.join(PKG, "configs", "gcn", "nsfnet_10_commodities.json"))
    assert cfg.num_nodes == 14 and cfg.batch_size == 20
    assert cfg.voc_nodes_in == 30                      # inherited from configs/gcn/default.json
    assert cfg.get("missing", 7) == 7
    with pytest.raises(AttributeError):
        cfg.not_there
    s = Settings({"a": 1})
    s.b = 2
    assert s["b"] == 2


def test_synthetic_batch_invariants():
    from src.common.synthetic import make_batch
    b = make_batch(6, 14, 10, "nsfnet", 1000, 5000, 5, 500, seed=3)
    cap = b["edges_capacity"]
    assert np.array_equal(cap, cap.transpose(0, 2, 1)) and (np.diagonal(cap, axis1=1, axis2=2) == 0).all()
    assert ((cap > 0).sum(axis=(1, 2)) == 42).all()                     # 21 undirected NSFNet links
    assert ((cap == 0) | ((cap >= 1000) & (cap < 5000))).all()
    comm = b["commodities"]
    assert (comm[:, :, 0] != comm[:, :, 1]).all()
    assert (np.diff(comm[:, :, 2], axis=1) <= 0).all()                  # sorted by demand descending
    assert ((b["nodes"] == 1).sum(axis=1) == 1).all() and ((b["nodes"] == 2).sum(axis=1) == 1).all()
    assert set(np.unique(b["edges_target"])) <= {0, 1}
    # labels only on existing links
    assert (b["edges_target"].sum(axis=3)[cap == 0] == 0).all()


def test_dataset_reader_synthetic_and_on_disk(tmp_path, monkeypatch):
    import networkx as nx

    from src.common.config.config import Settings
    from src.common.data_management.dataset_reader import DatasetReader
    cfg = Settings(dict(num_nodes=14, num_commodities=3, synthetic_data=True, graph_model="nsfnet",
                        capacity_lower=1000, capacity_higher=5000, demand_lower=5, demand_higher=500))
    r = DatasetReader(8, 4, "train", cfg, seed=0)
    batches = list(r)
    assert len(batches) == 2 and batches[0].edges_capacity.shape == (4, 14, 14)
    assert batches[0].edges_target.shape == (4, 14, 14, 3) and batches[0].commodities.shape == (4, 3, 3)
    # on-disk layout of the reference: graph_file/<bucket>/graph_<i>.gml etc.
    root = tmp_path / "data"
    d = root / "mini" / "val_data"
    for sub in ("graph_file/0", "commodity_file/0", "node_flow_file/0"):
        (d / sub).mkdir(parents=True)
    for i in range(2):
        G = nx.DiGraph()
        G.add_nodes_from(range(4))
        for u, v, c in [(0, 1, 100 + i), (1, 2, 200), (2, 3, 300), (3, 0, 400)]:
            G.add_edge(u, v, capacity=c)
            G.add_edge(v, u, capacity=c)
        nx.write_gml(G, str(d / "graph_file/0" / f"graph_{i}.gml"))
        with open(d / "commodity_file/0" / f"commodity_data_{i}.csv", "w", newline="") as f:
            csv.writer(f).writerows([[0, 2, 50], [3, 1, 20]])
        with open(d / "node_flow_file/0" / f"node_flow_{i}.csv", "w", newline="") as f:
            csv.writer(f).writerows([[1, 2, 3, 0], [0, 3, 2, 1]])      # visit order per node, 0 = not on the path
    monkeypatch.setenv("GCN_UELB_DATA_ROOT", str(root))
    cfg2 = Settings(dict(num_nodes=4, num_commodities=2, dataset_name="mini"))
    b = next(iter(DatasetReader(2, 2, "val", cfg2)))
    assert b.edges_capacity.shape == (2, 4, 4) and b.edges_capacity[1, 0, 1] == 101
    assert b.nodes[0, 0, 0] == 1 and b.nodes[0, 2, 0] == 2
    assert b.edges_target[0, 0, 1, 0] == 1 and b.edges_target[0, 1, 2, 0] == 1 and b.edges_target[0, 2, 3, 0] == 0
    assert b.edges_target[0, 3, 2, 1] == 1 and b.edges_target[0, 2, 1, 1] == 1
    with pytest.raises(FileNotFoundError):
        DatasetReader(2, 2, "test", Settings(dict(num_nodes=4, num_commodities=2, dataset_name="absent")))
