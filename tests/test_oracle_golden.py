"""The CPU oracles against vectors produced by the reference itself (oracle/gen_golden.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import golden_files, load_golden
from oracle import beam_oracle, beam_oracle_c, gcn_oracle


def _params(g, requires_grad=True):
    P = {}
    for k, v in g.items():
        if k.startswith("param."):
            t = torch.from_numpy(v.copy())
            t.requires_grad_(requires_grad)
            P[k[len("param."):]] = t
    return P


def _cfg(g):
    return dict(num_layers=int(g["cfg_L"]), mlp_layers=int(g["cfg_mlp_layers"]),
                aggregation=str(g["cfg_aggregation"]), voc_edges_out=int(g["cfg_voc"]))


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("path", golden_files("model"), ids=os.path.basename)
def test_model_oracle_matches_reference(path):
    g = load_golden(path)
    P = _params(g)
    cfg = _cfg(g)
    cap = torch.from_numpy(g["in_cap"])
    comm = torch.from_numpy(g["in_comm"])
    x_nodes = torch.from_numpy(g["in_x_nodes"])
    if cfg["voc_edges_out"] == 2:
        y, loss, layers = gcn_oracle.model_forward(
            P, cfg, cap, comm, x_nodes, torch.from_numpy(g["in_y_edges"]), g["in_edge_cw"], return_layers=True)
        loss.backward()
        assert abs(loss.item() - float(g["out_loss"])) <= 1e-5 * abs(float(g["out_loss"]))
        err, err_idx = gcn_oracle.edge_error(y, torch.from_numpy(g["in_y_edges"]))
        assert err == pytest.approx(float(g["out_edge_error"]), nan_ok=True)
        assert np.array_equal(err_idx.numpy(), g["out_err_idx"])
    else:
        y, loss, layers = gcn_oracle.model_forward(
            P, cfg, cap, comm, x_nodes, compute_loss=False, mask_invalid_edges=bool(g["cfg_mask"]),
            return_layers=True)
        assert loss is None
        (y * torch.from_numpy(g["in_dy_pred"])).sum().backward()
    assert rel_err(y.detach().numpy(), g["out_y_pred"]) < 1e-5
    assert rel_err(layers[-1][0].detach().numpy(), g["out_x_last"]) < 1e-5
    for l, (x_l, e_l, _) in enumerate(layers):
        if f"layer{l}.x" in g:
            assert rel_err(x_l.detach().numpy(), g[f"layer{l}.x"]) < 1e-5
            assert rel_err(e_l.detach().numpy(), g[f"layer{l}.e"]) < 1e-5
    for k, p in P.items():
        ref = g["grad." + k]
        got = np.zeros_like(ref) if p.grad is None else p.grad.numpy()
        scale = max(np.abs(ref).max(), 1e-30)
        assert np.abs(got - ref).max() <= 2e-4 * scale + 1e-7, k


def _unpack(paths, plen):
    return [[[int(v) for v in paths[b, c, :plen[b, c]]] for c in range(paths.shape[1])]
            for b in range(paths.shape[0])]


@pytest.mark.parametrize("path", golden_files("beam"), ids=os.path.basename)
def test_beam_oracles_match_reference(path):
    g = load_golden(path)
    algo, K = str(g["cfg_algorithm"]), int(g["cfg_K"])
    strict, max_iter = bool(g["cfg_strict"]), int(g["cfg_max_iter"])
    want = _unpack(g["out_paths"], g["out_path_len"])
    # C restatement, bit-exact
    paths, plen, feas, att, used = beam_oracle_c.search(g["in_y"], g["in_cap"], g["in_comm"], K, algo, strict,
                                                        max_iter, g["in_perms"])
    assert _unpack(paths, plen) == want
    assert np.array_equal(feas, g["out_feasible"])
    assert used == g["in_perms"].shape[0]
    # pure-Python restatement on the smaller cases
    if int(g["cfg_V"]) <= 14:
        p2, f2 = beam_oracle.search(g["in_y"], g["in_cap"], g["in_comm"], K, algo, strict, max_iter,
                                    perms=list(g["in_perms"]))
        assert p2 == want
        assert [int(v) for v in f2] == [int(v) for v in g["out_feasible"]]


def test_beam_scores_restatement():
    g = load_golden([p for p in golden_files("beam") if p.endswith("beam_det_strict.npz")][0])
    y = gcn_oracle.beam_scores(torch.from_numpy(g["in_logits"])).numpy()
    assert np.array_equal(y, g["in_y"])
    g = load_golden([p for p in golden_files("beam") if p.endswith("beam_scores4d.npz")][0])
    y = gcn_oracle.beam_scores(torch.from_numpy(g["in_logits"])).numpy()
    assert np.array_equal(y, g["in_y"])
