"""Dataset reading against what the reference's DatasetReader returned for the fixture datasets under
tests/golden/datasets (written in the reference's on-disk format by its own graph and commodity generators)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden

CASES = ["nsf_small", "rand_small"]
KEYS = ("edges", "edges_capacity", "edges_target", "nodes", "nodes_target", "commodities", "load_factor")


def _cfg(name):
    from src.common.config.config import Settings
    return Settings(dict(data_root=os.path.join(GOLDEN, "datasets"), dataset_name=name))


@pytest.fixture(autouse=True)
def _no_env_root(monkeypatch):
    monkeypatch.delenv("GCN_UELB_DATA_ROOT", raising=False)


@pytest.mark.parametrize("name", CASES)
def test_process_batch_matches_reference(name):
    from src.common.data_management.dataset_reader import DatasetReader
    g = load_golden(os.path.join(GOLDEN, f"dataset_{name}.npz"))
    r = DatasetReader(int(g["cfg_N"]), int(g["cfg_batch"]), "train", _cfg(name), shuffle=False)
    for k in range(2):
        b = r.process_batch(g[f"b{k}.indices"])
        for key in KEYS:
            got, want = np.asarray(b[key]), g[f"b{k}.{key}"]
            assert got.shape == want.shape and got.dtype.kind == want.dtype.kind, key
            np.testing.assert_array_equal(got, want, err_msg=key)
    first = next(iter(r))
    np.testing.assert_array_equal(first.commodities, g["b0.commodities"])       # shuffle=False: batch 0 = samples 0..


@pytest.mark.parametrize("name", CASES)
def test_packed_file_holds_the_same_samples(name, tmp_path):
    from src.common.data_management.packed_dataset import PackedDataset, pack_dataset
    g = load_golden(os.path.join(GOLDEN, f"dataset_{name}.npz"))
    path = pack_dataset(_cfg(name), "train", int(g["cfg_N"]), out_path=tmp_path / "p.gcnpack")
    ds = PackedDataset(path)
    assert (ds.N, ds.V, ds.C) == (int(g["cfg_N"]), int(g["cfg_V"]), int(g["cfg_C"]))
    for k in range(2):
        idx = g[f"b{k}.indices"]
        np.testing.assert_array_equal(ds.cap[idx].astype(np.float64), g[f"b{k}.edges_capacity"])
        np.testing.assert_array_equal(ds.edges[idx], g[f"b{k}.edges"])
        np.testing.assert_array_equal(ds.comm[idx], g[f"b{k}.commodities"])
        np.testing.assert_array_equal(ds.flow[idx], g[f"b{k}.nodes_target"])
        np.testing.assert_array_equal(ds.lf[idx], g[f"b{k}.load_factor"])
    with open(path, "r+b") as f:
        f.write(b"XXXX")
    with pytest.raises(ValueError):
        PackedDataset(path)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_packed_loader_yields_the_reference_tensors(name, tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    from src.common.data_management.packed_dataset import PackedBatchLoader, pack_dataset
    dev = torch.device("cuda:0")
    g = load_golden(os.path.join(GOLDEN, f"dataset_{name}.npz"))
    N, B = int(g["cfg_N"]), int(g["cfg_batch"])
    path = pack_dataset(_cfg(name), "train", N, out_path=tmp_path / "p.gcnpack")
    loader = PackedBatchLoader(path, B, dev, shuffle=False)
    batches = [{k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in b.items()} for b in loader]
    assert len(batches) == N // B == len(loader)
    b0 = batches[0]
    want = {"x_edges": g["b0.edges"], "x_edges_capacity": g["b0.edges_capacity"].astype(np.float32), "x_nodes": g["b0.nodes"],
            "y_edges": g["b0.edges_target"], "batch_commodities": g["b0.commodities"],
            "x_commodities": g["b0.commodities"][:, :, 2].astype(np.float32)}
    for key, w in want.items():
        got = b0[key].cpu().numpy()
        assert got.dtype == (np.float32 if w.dtype == np.float32 else np.int64) and got.shape == w.shape, key
        np.testing.assert_array_equal(got, w, err_msg=key)
    np.testing.assert_array_equal(b0["load_factor"], g["b0.load_factor"])
    # every batch against the slow reader; a shuffled epoch covers each sample at most once; shards split a batch
    from src.common.data_management.dataset_reader import DatasetReader
    slow = DatasetReader(N, B, "train", _cfg(name), shuffle=False)
    for b in batches:
        ref = slow.process_batch(b["indices"])
        np.testing.assert_array_equal(b["y_edges"].cpu().numpy(), ref.edges_target)
        np.testing.assert_array_equal(b["x_nodes"].cpu().numpy(), ref.nodes)
    seen = np.concatenate([b["indices"] for b in PackedBatchLoader(path, B, dev, shuffle=True, seed=3)])
    assert len(set(seen.tolist())) == len(seen) == (N // B) * B
    parts = [next(iter(PackedBatchLoader(path, B, dev, shuffle=False, rank=r, world=2))) for r in range(2)]
    both = torch.cat([p["y_edges"].clone() for p in parts])
    np.testing.assert_array_equal(both.cpu().numpy(), g["b0.edges_target"])


@pytest.mark.gpu
def test_unpack_orders_with_gaps_and_ties():
    """Visiting orders need not be 1..k: gaps sort the same way; equal orders resolve to the lower node first."""
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    from src.common.data_management.dataset_reader import edges_from_node_order, nodes_from_commodities
    from src.gcn.hip import lib as L
    dev = torch.device("cuda:0")
    lib = L.load()
    rng = np.random.default_rng(5)
    B, V, C = 3, 11, 4
    flow = np.zeros((B, C, V), np.int16)
    comm = np.zeros((B, C, 3), np.int32)
    for b in range(B):
        for c in range(C):
            k = rng.integers(0, V + 1)
            nodes = rng.permutation(V)[:k]
            flow[b, c, nodes] = np.sort(rng.choice(np.arange(1, 40), size=k, replace=(b == 2)))   # graph 2: ties
            comm[b, c] = (rng.integers(0, V), rng.integers(0, V), rng.integers(1, 500))
    comm[0, 0, 1] = comm[0, 0, 0]                                                                   # source == destination
    edges = (rng.random((B, V, V)) < 0.3).astype(np.uint8)
    t = lambda a: torch.from_numpy(a).to(dev)
    d_e, d_c, d_f = t(edges), t(comm), t(flow)
    o = dict(x_edges=torch.empty((B, V, V), dtype=torch.int64, device=dev), x_nodes=torch.empty((B, V, C), dtype=torch.int64, device=dev),
             y=torch.empty((B, V, V, C), dtype=torch.int64, device=dev), comm=torch.empty((B, C, 3), dtype=torch.int64, device=dev),
             dem=torch.empty((B, C), dtype=torch.float32, device=dev), succ=torch.empty((B, C, V), dtype=torch.int16, device=dev))
    L.check(lib.gcn_unpack_batch(B, V, C, L.ptr(d_e), L.ptr(d_c), L.ptr(d_f), L.ptr(o["x_edges"]), L.ptr(o["x_nodes"]), L.ptr(o["y"]),
                                 L.ptr(o["comm"]), L.ptr(o["dem"]), L.ptr(o["succ"]), L.stream_ptr(dev)), "gcn_unpack_batch")
    for b in range(B):
        np.testing.assert_array_equal(o["y"][b].cpu().numpy(), edges_from_node_order(flow[b].tolist(), V))
        np.testing.assert_array_equal(o["x_nodes"][b].cpu().numpy(), nodes_from_commodities(comm[b].tolist(), V))
    np.testing.assert_array_equal(o["x_edges"].cpu().numpy(), edges.astype(np.int64))
    np.testing.assert_array_equal(o["comm"].cpu().numpy(), comm.astype(np.int64))
    np.testing.assert_array_equal(o["dem"].cpu().numpy(), comm[:, :, 2].astype(np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("strategy", ["supervised", "reinforcement"])
def test_trainer_runs_from_a_packed_dataset(strategy, tmp_path, monkeypatch):
    """One epoch over the fixture dataset through the packed loader: the pack file is written on first use, the
    supervised step replays its graph on device-resident batches, the REINFORCE step consumes the same dict."""
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    import shutil

    from src.common.config.config import Settings
    from src.gcn.train.trainer import Trainer
    shutil.copytree(os.path.join(GOLDEN, "datasets", "nsf_small"), tmp_path / "nsf_small")
    monkeypatch.setenv("GCN_UELB_MODEL_ROOT", str(tmp_path / "models"))
    cfg = Settings(dict(data_root=str(tmp_path), dataset_name="nsf_small", packed_dataset=True, num_train_data=12, batch_size=4,
                        num_nodes=14, num_commodities=4, voc_edges_in=3, voc_edges_out=2 if strategy == "supervised" else 1,
                        hidden_dim=16, num_layers=2, mlp_layers=2, aggregation="mean", learning_rate=1e-3, max_epochs=1,
                        training_strategy=strategy, beam_size=2, rl_entropy_epsilon=0.05))
    torch.manual_seed(0)
    tr = Trainer(cfg, torch.float32, torch.long)
    before = [p.detach().clone() for p in tr.get_model().parameters()]
    t, loss, err, rl = tr.train_one_epoch()
    assert (tmp_path / "nsf_small" / "train_data" / "packed.gcnpack").exists()
    assert np.isfinite(loss)
    assert any(not torch.equal(a, b.detach()) for a, b in zip(before, tr.get_model().parameters()))
    if strategy == "reinforcement":
        assert "reward" in rl and np.isfinite(rl["reward"]) and 0.0 <= rl["complete_paths_rate"] <= 100.0
    else:
        assert 0.0 <= err <= 100.0
        # the class weights came from the labels on the device: n / (2 * count_k)
        from src.common.data_management.dataset_reader import DatasetReader
        w = tr.strategy.edge_cw.cpu().numpy()
        assert w.shape == (2,) and w[1] > w[0] > 0.5
