"""CPU-side checks: the C-ABI library loads and exports every symbol the header declares, and the
drop-in Python classes keep the reference's names, state_dict keys and error behaviour."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, golden_files, load_golden


def _header_functions():
    text = open(os.path.join(ROOT, "include", "gcn_uelb_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gcn_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from src.gcn.hip import lib as L
    lib = L.load()
    names = _header_functions()
    assert "gcn_model_fwd" in names and "gcn_layer_bwd" in names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/gcn_uelb_hip.h but not exported"
    assert lib.gcn_abi_version() == L.ABI_VERSION


def test_size_queries_and_argument_errors_without_gpu():
    from src.gcn.hip import lib as L
    lib = L.load()
    assert lib.gcn_layer_saved_floats(2, 5, 3, 8) == 2 * 5 * 5 * 3 * 8 + 4 * 2 * 5 * 3 * 8 + 32
    assert lib.gcn_layer_ws_bytes(2, 5, 3, 8) > 0
    assert lib.gcn_layer_saved_floats(2, 5, 3, 6) == 0            # H % 4 != 0
    assert b"multiple of 4" in lib.gcn_last_error()
    d = L.ModelDesc(B=2, V=5, C=3, H=8, L=2, mlp_layers=9, voc_out=2, n_emb=9, aggregation=1)
    assert lib.gcn_model_num_params(ctypes.byref(d)) == -1
    assert b"mlp_layers" in lib.gcn_last_error()
    d.mlp_layers = 2
    assert lib.gcn_model_num_params(ctypes.byref(d)) == 3 + 12 * 2 + 4


def test_plan_and_saved_layout_queries_without_gpu():
    from src.gcn.hip import lib as L
    lib = L.load()
    plan = L.layer_plan(20, 14, 10, 64)
    assert plan["j_tiles"] == 1 and not plan["wide"] and 1 <= plan["row_chunk"] <= 14
    assert plan["n_chunks"] == -(-14 // plan["row_chunk"]) and plan["grid_fwd"] > 0 and plan["grid_bwd"] > 0
    assert L.layer_plan(1, 100, 50, 300)["wide"]
    d = L.ModelDesc(B=2, V=5, C=3, H=8, L=3, mlp_layers=2, voc_out=2, n_emb=9, aggregation=1)
    total = lib.gcn_model_saved_floats(ctypes.byref(d))
    Re, Rn, H = 2 * 5 * 5 * 3, 2 * 5 * 3, 8
    seen = set()
    for layer in range(3):
        off = (ctypes.c_size_t * 8)()
        assert lib.gcn_model_saved_layout(ctypes.byref(d), layer, off) == 0
        spans = [(off[0], Re * H), (off[1], Rn * H), (off[2], 4 * H), (off[3], Re * H), (off[4], Rn * H)]
        for lo, n in spans:
            assert lo + n <= total and lo not in seen       # inside the buffer, no two tensors at one offset
            seen.add(lo)
    assert lib.gcn_model_saved_layout(ctypes.byref(d), 3, off) != 0 and b"layer" in lib.gcn_last_error()
    assert lib.gcn_load_factor_static(2, 1024, 1, 1024, None, None, None, None, None, None, None) == -1   # null pointers first


class _Cfg(dict):
    __getattr__ = dict.__getitem__


def test_model_state_dict_keys_match_reference():
    from src.gcn.models.gcn_model import ResidualGatedGCNModel
    g = load_golden([p for p in golden_files("model") if p.endswith("model_h12_mlp3.npz")][0])
    cfg = _Cfg(num_nodes=6, num_commodities=4, voc_edges_in=3, voc_edges_out=2, hidden_dim=12, num_layers=2,
               mlp_layers=3, aggregation="mean")
    model = ResidualGatedGCNModel(cfg, torch.float32, torch.long)
    ref_keys = [k[len("param."):] for k in g if k.startswith("param.")]
    assert list(model.state_dict().keys()) == ref_keys
    for k, v in model.state_dict().items():
        assert tuple(v.shape) == g["param." + k].shape
    assert [p.data_ptr() for p in model.hip_parameters()] == [v.data_ptr() for v in model.state_dict().values()]
    for attr in ("num_nodes", "num_commodities", "hidden_dim", "num_layers", "mlp_layers", "aggregation",
                 "voc_edges_out", "dropout_rate"):
        assert hasattr(model, attr)


def test_model_errors_match_reference_behaviour():
    from src.gcn.hip.lib import GcnHipError
    from src.gcn.models.gcn_layers import MLP, ResidualGatedGCNLayer
    from src.gcn.models.gcn_model import ResidualGatedGCNModel
    cfg = _Cfg(num_nodes=5, num_commodities=3, voc_edges_in=3, voc_edges_out=2, hidden_dim=8, num_layers=1,
               mlp_layers=2, aggregation="mean")
    model = ResidualGatedGCNModel(cfg, torch.float32, torch.long)
    cap = torch.zeros(2, 5, 5)
    with pytest.raises(ValueError):        # gcn_model.py:115-116
        model(cap.long(), torch.zeros(2, 3), cap, torch.zeros(2, 5, 3, dtype=torch.long))
    with pytest.raises(GcnHipError):       # no silent CPU fallback
        model(cap.long(), torch.zeros(2, 3), cap, torch.zeros(2, 5, 3, dtype=torch.long), compute_loss=False)
    with pytest.raises(ValueError):        # gcn_layers.py:153-154
        MLP(8, 2, num_layers=3, hidden_dims=[8])
    layer = ResidualGatedGCNLayer(8, "mean", dropout_rate=0.0)
    assert len(layer.hip_parameters()) == 12
    with pytest.raises(GcnHipError):
        layer(torch.zeros(1, 2, 1, 8), torch.zeros(1, 2, 2, 1, 8))


def test_launch_plan_is_sane_over_a_grid_of_shapes():
    """gcn_layer_plan is host code: row chunks of equal length that cover V, grids within the partial-row buffers
    (2048 rows), a multiple of 8 forward workgroups on the strip path (the forward kernels deal their units to the 8
    XCDs), and the two-block kernels exactly from 300,000 edge rows on at 32 / 64 channels."""
    from src.gcn.hip import lib as L
    L.load()
    for H in (16, 32, 48, 64, 128, 300):
        for (B, V, C) in [(1, 5, 1), (2, 14, 4), (20, 14, 10), (20, 14, 20), (20, 40, 5), (40, 40, 5), (80, 14, 10),
                          (160, 14, 10), (320, 14, 10), (1, 100, 50), (2, 100, 50), (1, 200, 100), (4, 200, 100), (3, 33, 7)]:
            plan = L.layer_plan(B, V, C, H)
            it, n = plan["row_chunk"], plan["n_chunks"]
            assert 1 <= it <= V and n == -(-V // it), (B, V, C, H, plan)
            assert n * it - V < it                                  # the last chunk is not empty
            assert -(-V // n) == it or H > 64                       # equal chunks (the wide path rounds up to fours)
            assert plan["j_tiles"] == -(-V // 16)
            assert 1 <= plan["grid_fwd"] <= 2048 and (0 if plan["packed_last_tile"] else 1) <= plan["grid_bwd"] <= 2048
            assert plan["packed_last_tile"] == (H > 64 and V - 16 * (plan["j_tiles"] - 1) <= 8)
            if H > 64:
                assert it % 4 == 0 or it == V                       # the wide kernels walk a chunk four rows at a time
            assert plan["wide"] == (H > 64)
            if H <= 64:
                assert plan["grid_fwd"] % 8 == 0
            assert plan["two_blocks"] == (H in (32, 64) and B * V * V * C >= 300000), (B, V, C, H, plan)


def test_metrics_logger_books_what_the_trainer_reports(tmp_path):
    """The logger takes the calls the reference's trainer makes (trainer.py:514, 541, 553, 571) and reports the
    reference's final-metric keys (metrics.py:104-117)."""
    import json
    from src.gcn.train.metrics import GCNMetricsLogger, metrics_to_str
    lg = GCNMetricsLogger(str(tmp_path))
    for epoch in range(3):
        lg.log_train_metrics(1.0 / (epoch + 1), 10.0 - epoch, 2.0)
        lg.log_rl_metrics(epoch, {"reward": -1.5 + epoch, "entropy": 0.3, "load_factor": 1.2, "complete_paths_rate": 90.0})
        lg.log_val_metrics(80.0 + epoch, 0.5, epoch=epoch, comp_rate=95.0, comp_sample_rate=60.0)
    lg.log_test_metrics(85.0, 0.25, epoch=2, comp_rate=97.0, comp_sample_rate=70.0)
    m = lg.get_final_metrics()
    assert set(m) == {"final_train_loss", "final_edge_error", "final_val_approximation_rate", "final_test_approximation_rate",
                      "best_val_approximation_rate", "best_test_approximation_rate", "total_train_time", "total_test_time",
                      "avg_train_time_per_epoch", "avg_test_time_per_epoch"}
    assert m["final_train_loss"] == 1.0 / 3 and m["final_edge_error"] == 8.0 and m["best_val_approximation_rate"] == 82.0
    assert m["total_train_time"] == 6.0 and m["avg_train_time_per_epoch"] == 2.0 and m["total_test_time"] == 0.25
    assert lg.val_comp_rate_list == [95.0] * 3 and lg.test_comp_sample_rate_list == [70.0] and lg.rl_reward_list == [-1.5, -0.5, 0.5]
    (csv,) = [f for f in os.listdir(tmp_path) if f.startswith("rl_gcn_metrics_")]
    rows = open(os.path.join(tmp_path, csv)).read().strip().split("\n")
    assert rows[0].startswith("epoch,reward,reward_std,advantage,advantage_std,entropy,load_factor,load_factor_std,baseline,")
    assert len(rows) == 4 and rows[3].startswith("2,0.500000,0.000000,")
    jpath, tpath = lg.save_results({"num_train_data": 10, "num_test_data": 4, "config_file": "x.json"})
    booked = json.load(open(jpath))
    assert booked["train_loss_list"] == lg.train_loss_list and booked["time_per_data"]["total_train_samples"] == 30
    assert "final_train_loss" in open(tpath).read()
    line = metrics_to_str(3, 1.5, 1e-3, 0.25, 1.1, 1.0, 90.0, 0.0, comp_rate=99.0)
    assert line.startswith("epoch:03\texecution time:1.50000000s\tlr:1.00e-03\tloss:0.2500\t") and line.endswith("comp_rate:99.00%")


def test_wide_launch_plan_row_chunks_and_ragged_tile():
    """Host-side plan of the wide backward pass (no GPU needed): row chunks are multiples of four rows chosen for the
    fewest steps on the busiest of 256 workgroups, and a last column tile with at most 8 of 16 columns in use gets the
    packed-tile launch."""
    from src.gcn.hip import lib as L
    p = L.layer_plan(2, 100, 50, 300)          # 7 column tiles, the last with 4 columns
    assert p["wide"] and p["packed_last_tile"] and (p["row_chunk"], p["n_chunks"], p["j_tiles"]) == (20, 5, 7)
    assert p["grid_bwd"] == 250                # 3,000 units of the six full tiles: 12 per workgroup
    p = L.layer_plan(2, 96, 50, 300)           # six full tiles
    assert p["wide"] and not p["packed_last_tile"] and p["row_chunk"] % 4 == 0
    p = L.layer_plan(1, 7, 256, 80)            # one tile of 7 columns: only the packed-tile launch runs
    assert p["packed_last_tile"] and p["grid_bwd"] == 0 and (p["row_chunk"], p["n_chunks"]) == (7, 1)
    p = L.layer_plan(1, 40, 40, 80)            # 16 + 16 + 8 columns
    assert p["packed_last_tile"] and p["j_tiles"] == 3
    p = L.layer_plan(1, 44, 40, 80)            # 16 + 16 + 12 columns: walked like a full tile
    assert not p["packed_last_tile"]
    p = L.layer_plan(20, 14, 10, 64)           # the strip path has no such launch
    assert not p["wide"] and not p["packed_last_tile"]

