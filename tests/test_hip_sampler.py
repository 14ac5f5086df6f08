"""GPU path sampler against the reference's golden vectors and the CPU restatement.

Sampled paths depend on the random stream, so parity has three legs: (1) replaying the reference's own sampled
paths must reproduce its log-probabilities, step entropies and autograd gradients (1e-4 relative);
(2) with the same counter-based uniforms the GPU and the CPU restatement must walk the same paths, except
where a draw lands within 1e-4 of a boundary of the cumulative distribution (different rounding of exp / sums
may then pick the neighbour); (3) first-hop frequencies must follow the exact nucleus distribution."""
import os

import numpy as np
import pytest
import torch

from conftest import golden_files, load_golden
from oracle import sampler_oracle as so

pytestmark = pytest.mark.gpu
CASES = [p for p in golden_files("sampler") if not p.endswith("sampler_freq.npz")]


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    return torch.device("cuda:0")


def _sampler(y, cap, comm, T, top_p, eps, dev):
    from src.gcn.algorithms.path_sampler import PathSampler
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    y = y if isinstance(y, torch.Tensor) else t(y)
    return PathSampler(y_pred_edges=y, edges_capacity=t(cap), commodities=t(comm), num_samples=1, temperature=T,
                       top_p=top_p, entropy_epsilon=eps)


def unpack(paths, plen):
    paths, plen = np.asarray(paths), np.asarray(plen)
    return [[[int(v) for v in paths[b, c, :plen[b, c]]] for c in range(paths.shape[1])] for b in range(paths.shape[0])]


@pytest.mark.parametrize("path", CASES, ids=os.path.basename)
def test_replayed_reference_paths(path):
    dev = _dev()
    g = load_golden(path)
    y = torch.from_numpy(g["in_logits"].copy()).to(dev).requires_grad_(True)
    s = _sampler(y, g["in_cap"], g["in_comm"], float(g["cfg_temperature"]), float(g["cfg_top_p"]), float(g["cfg_eps"]), dev)
    paths, logp, feasible, ents = s.evaluate(torch.from_numpy(g["out_paths"]), torch.from_numpy(g["out_path_len"]))
    assert paths == unpack(g["out_paths"], g["out_path_len"])
    np.testing.assert_allclose(logp.detach().cpu().numpy(), g["out_logp"], rtol=1e-4, atol=1e-4)
    assert feasible == bool(g["out_feasible"])
    for b, graph in enumerate(ents):
        for c, e in enumerate(graph):
            assert len(e) == int(g["out_n_entropy"][b, c])
            np.testing.assert_allclose(e, g["out_entropy"][b, c, :len(e)], rtol=1e-4, atol=1e-5)
    (logp * torch.from_numpy(g["in_w"]).to(dev)).sum().backward()
    scale = np.abs(g["out_dlogits"]).max()
    np.testing.assert_allclose(y.grad.cpu().numpy(), g["out_dlogits"], rtol=1e-4, atol=1e-4 * scale)
    np.testing.assert_array_equal(s.reachability.cpu().numpy().astype(np.uint8), g["out_reach"])


def _compare_with_oracle(y, cap, comm, T, top_p, eps, seed, dev, max_mismatch_frac=0.05):
    torch.manual_seed(seed)
    yt = torch.from_numpy(y.copy()).to(dev).requires_grad_(True)
    s = _sampler(yt, cap, comm, T, top_p, eps, dev)
    paths, logp, feasible, ents = s.sample()
    sd, call = s.last_rng
    margins = []
    yo = torch.from_numpy(y.copy()).requires_grad_(True)
    opaths, ologp, ofeas, oents, ologp_bc = so.sample(yo, cap, comm, T, top_p, eps, uniform=so.counter_uniform(sd, call),
                                                      margin_out=margins)
    B, C = len(paths), len(paths[0])
    close = {(b, c) for b, c, _, m in margins if m < 1e-4}
    same = np.zeros((B, C), dtype=bool)
    for b in range(B):
        for c in range(C):
            same[b, c] = paths[b][c] == opaths[b][c]
            assert same[b, c] or (b, c) in close, (b, c, paths[b][c], opaths[b][c])
            assert paths[b][c][0] == int(comm[b, c, 0])
    assert (~same).mean() <= max_mismatch_frac
    got = s.log_probs_per_commodity.cpu().numpy()
    np.testing.assert_allclose(got[same], ologp_bc.detach().numpy()[same], rtol=1e-4, atol=1e-4)
    for b in range(B):
        for c in range(C):
            if same[b, c]:
                np.testing.assert_allclose(ents[b][c], oents[b][c], rtol=1e-4, atol=1e-5)
    if same.all():
        assert feasible == ofeas
        w = torch.linspace(-1.0, 1.5, B)
        (logp * w.to(dev)).sum().backward()
        (ologp * w).sum().backward()
        scale = float(yo.grad.abs().max()) + 1e-30
        np.testing.assert_allclose(yt.grad.cpu().numpy(), yo.grad.numpy(), rtol=1e-4, atol=1e-4 * scale)
    return same


@pytest.mark.parametrize("name", ["nsf", "spike", "noeps_stuck", "v70"])
@pytest.mark.parametrize("seed", [5, 6])
def test_sampled_paths_match_the_cpu_restatement(golden_dir, name, seed):
    dev = _dev()
    g = load_golden(os.path.join(golden_dir, f"sampler_{name}.npz"))
    _compare_with_oracle(g["in_logits"], g["in_cap"], g["in_comm"], float(g["cfg_temperature"]), float(g["cfg_top_p"]),
                         float(g["cfg_eps"]), seed, dev)


def _random_case(rng, B, V, C, extra=0.1, cap_rng=(1000, 5000), dem_rng=(5, 500)):
    from oracle.gen_golden import make_graph_batch
    cap, comm = make_graph_batch(rng, B, V, C, cap_rng[0], cap_rng[1], dem_rng[0], dem_rng[1], extra)
    y = (rng.standard_normal((B, V, V, C)) * 2.0).astype(np.float32)
    y[np.broadcast_to((cap == 0)[..., None], y.shape)] = -1e9
    return y, cap, comm


@pytest.mark.parametrize("V,C,T,top_p,eps", [(100, 3, 1.0, 0.9, 0.05), (130, 2, 0.7, 1.0, 0.0), (65, 4, 2.0, 0.5, 0.2),
                                             (14, 10, 1.5, 0.95, 0.05)])
def test_larger_graphs_match_the_cpu_restatement(V, C, T, top_p, eps):
    dev = _dev()
    rng = np.random.default_rng(V * 7 + C)
    y, cap, comm = _random_case(rng, 2, V, C, extra=0.03 if V > 20 else 0.2)
    _compare_with_oracle(y, cap, comm, T, top_p, eps, 11, dev)


def test_source_equals_destination_and_tight_capacities():
    dev = _dev()
    rng = np.random.default_rng(3)
    y, cap, comm = _random_case(rng, 3, 8, 5, extra=0.3, cap_rng=(300, 700), dem_rng=(100, 500))
    comm[1, 2, 1] = comm[1, 2, 0]                                   # a commodity that is already at its destination
    same = _compare_with_oracle(y, cap, comm, 1.0, 0.9, 0.05, 21, dev, max_mismatch_frac=0.0)
    assert same.all()
    s = _sampler(y, cap, comm, 1.0, 0.9, 0.05, dev)
    paths, logp, feasible, ents = s.sample()
    assert paths[1][2] == [int(comm[1, 2, 0])] and ents[1][2] == []
    assert float(s.log_probs_per_commodity[1, 2]) == 0.0
    # feasibility flag against a direct count of edge usage
    ok = True
    for b in range(3):
        use = np.zeros((8, 8))
        for c in range(5):
            for u, v in zip(paths[b][c][:-1], paths[b][c][1:]):
                use[u, v] += comm[b, c, 2]
        ok = ok and not (use > cap[b] + 1e-6).any()
    assert feasible == ok


def test_first_hop_frequencies(golden_dir):
    dev = _dev()
    g = load_golden(os.path.join(golden_dir, "sampler_freq.npz"))
    T, top_p, eps = float(g["cfg_temperature"]), float(g["cfg_top_p"]), float(g["cfg_eps"])
    dist = so.first_hop_distribution(g["in_logits"], g["in_cap"], g["in_comm"], 0, 0, T, top_p, eps)
    reps = 1000
    y = np.repeat(g["in_logits"], reps, 0)
    cap = np.repeat(g["in_cap"], reps, 0)
    comm = np.repeat(g["in_comm"], reps, 0)
    torch.manual_seed(77)
    s = _sampler(y, cap, comm, T, top_p, eps, dev)
    counts = np.zeros_like(dist)
    for _ in range(8):
        s.sample()
        first = s.paths_tensor[:, 0, 1].cpu().numpy()
        counts += np.bincount(first, minlength=len(dist))
    assert (counts[dist == 0] == 0).all()
    sel = dist > 0
    n = counts.sum()
    stat = float((((counts[sel] - n * dist[sel]) ** 2) / (n * dist[sel])).sum())
    assert stat < 16.3 + 5 * (int(sel.sum()) - 1), (stat, counts, n * dist)
    # successive calls use fresh draws
    a = s.sample()[0]
    b = s.sample()[0]
    assert a != b


def test_sampler_rejects_what_it_cannot_run():
    dev = _dev()
    from src.gcn.hip.lib import GcnHipError
    rng = np.random.default_rng(0)
    y, cap, comm = _random_case(rng, 1, 6, 2)
    with pytest.raises(GcnHipError):
        _sampler(torch.from_numpy(y), cap, comm, 1.0, 0.9, 0.0, dev)          # predictions on the CPU
    with pytest.raises(GcnHipError):
        _sampler(y, cap, comm, 0.0, 0.9, 0.0, dev).sample()                   # temperature 0
    big = torch.zeros((1, 300, 300, 1), device=dev)
    with pytest.raises(GcnHipError):
        _sampler(big, np.ones((1, 300, 300), np.float32), np.array([[[0, 1, 5]]]), 1.0, 0.9, 0.0, dev).sample()
