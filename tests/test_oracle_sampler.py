"""The CPU restatement of the path sampler against vectors produced by the reference's PathSampler.

The sampled paths themselves depend on torch's generator, so parity is pinned on everything that is a
deterministic function of a path (log-probabilities, step entropies, gradients, reachability) by replaying the
reference's own paths, on a nucleus of one entry (top_p ~ 0, deterministic walk), and on first-hop
frequencies."""
import os

import numpy as np
import pytest
import torch

from conftest import golden_files, load_golden
from oracle import sampler_oracle as so

CASES = [p for p in golden_files("sampler") if not p.endswith("sampler_freq.npz")]


def unpack(paths, plen):
    return [[[int(v) for v in paths[b, c, :plen[b, c]]] for c in range(paths.shape[1])] for b in range(paths.shape[0])]


@pytest.mark.parametrize("path", CASES, ids=os.path.basename)
def test_replayed_reference_paths(path):
    g = load_golden(path)
    y = torch.from_numpy(g["in_logits"].copy()).requires_grad_(True)
    want_paths = unpack(g["out_paths"], g["out_path_len"])
    paths, logp, feasible, ents, _ = so.sample(y, g["in_cap"], g["in_comm"], float(g["cfg_temperature"]),
                                               float(g["cfg_top_p"]), float(g["cfg_eps"]), forced_paths=want_paths)
    assert paths == want_paths
    np.testing.assert_allclose(logp.detach().numpy(), g["out_logp"], rtol=1e-5, atol=1e-5)
    assert feasible == bool(g["out_feasible"])
    for b, graph in enumerate(ents):
        for c, e in enumerate(graph):
            assert len(e) == int(g["out_n_entropy"][b, c])
            np.testing.assert_allclose(e, g["out_entropy"][b, c, :len(e)], rtol=1e-5, atol=1e-6)
    (logp * torch.from_numpy(g["in_w"])).sum().backward()
    scale = np.abs(g["out_dlogits"]).max()
    np.testing.assert_allclose(y.grad.numpy(), g["out_dlogits"], rtol=1e-4, atol=1e-5 * scale)
    np.testing.assert_array_equal(so.reachability(g["in_cap"]).astype(np.uint8), g["out_reach"])


def test_nucleus_of_one_is_deterministic(golden_dir):
    g = load_golden(os.path.join(golden_dir, "sampler_greedy.npz"))
    want = unpack(g["out_paths"], g["out_path_len"])
    for call in (0, 7):
        paths, logp, _, _, _ = so.sample(torch.from_numpy(g["in_logits"]), g["in_cap"], g["in_comm"],
                                         float(g["cfg_temperature"]), float(g["cfg_top_p"]), float(g["cfg_eps"]),
                                         uniform=so.counter_uniform(123, call))
        assert paths == want
        np.testing.assert_allclose(logp.numpy(), g["out_logp"], rtol=1e-5, atol=1e-5)


def test_stuck_and_fallback_cases_are_covered(golden_dir):
    """The fixtures must exercise the stuck exit and the uniform fallback, else they pin nothing there."""
    g = load_golden(os.path.join(golden_dir, "sampler_noeps_stuck.npz"))
    assert int(g["out_path_len"][0, 0]) == 1                       # isolated destination: stuck at the source
    g = load_golden(os.path.join(golden_dir, "sampler_spike.npz"))
    steps = g["out_path_len"] - 1
    assert (g["out_n_entropy"] < steps).any()                      # some step went through the fallback


def chi2(counts, dist):
    n = counts.sum()
    assert (counts[dist == 0] == 0).all(), "draw outside the nucleus"
    sel = dist > 0
    return float((((counts[sel] - n * dist[sel]) ** 2) / (n * dist[sel])).sum()), int(sel.sum()) - 1


def test_first_hop_frequencies(golden_dir):
    g = load_golden(os.path.join(golden_dir, "sampler_freq.npz"))
    T, top_p, eps = float(g["cfg_temperature"]), float(g["cfg_top_p"]), float(g["cfg_eps"])
    dist = so.first_hop_distribution(g["in_logits"], g["in_cap"], g["in_comm"], 0, 0, T, top_p, eps)
    stat, dof = chi2(g["out_first_hop_counts"].astype(np.float64), dist)   # the reference's own draws
    assert stat < 16.3 + 5 * dof, (stat, dof)                               # far beyond the 0.1 % quantile for dof <= 4
    counts = np.zeros_like(dist)
    y = torch.from_numpy(g["in_logits"])
    for call in range(1500):                                                # the counter-based draws
        paths, *_ = so.sample(y, g["in_cap"], g["in_comm"], T, top_p, eps, uniform=so.counter_uniform(99, call))
        counts[paths[0][0][1]] += 1
    stat, dof = chi2(counts, dist)
    assert stat < 16.3 + 5 * dof, (stat, dof)
