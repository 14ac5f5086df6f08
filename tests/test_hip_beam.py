"""GPU beam search against the reference's golden paths and the C oracle (bit-exact)."""
import os

import numpy as np
import pytest
import torch

from conftest import golden_files, load_golden
from oracle import beam_oracle_c

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    return torch.device("cuda:0")


def _make(algo, y_or_logits, cap, comm, K, strict, max_iter, dev, use_y=True):
    from src.gcn.algorithms.beamsearch_uelb import BeamSearchFactory
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    B, V = cap.shape[0], cap.shape[1]
    C = comm.shape[1]
    dummy = torch.zeros((B, V, V, C), device=dev)
    a = BeamSearchFactory.create_algorithm(algo, y_pred_edges=dummy if use_y else t(y_or_logits), beam_size=K,
                                           batch_size=B, edges_capacity=t(cap), commodities=t(comm),
                                           dtypeFloat=torch.float32, dtypeLong=torch.long, mode_strict=strict,
                                           max_iter=max_iter)
    if use_y:
        a.y = t(y_or_logits)          # parity is defined on identical log-probabilities
    return a


def _unpack(paths, plen):
    paths, plen = np.asarray(paths), np.asarray(plen)
    return [[[int(v) for v in paths[b, c, :plen[b, c]]] for c in range(paths.shape[1])] for b in range(paths.shape[0])]


@pytest.mark.parametrize("path", golden_files("beam"), ids=os.path.basename)
def test_beam_matches_reference_golden(path, monkeypatch):
    dev = _dev()
    g = load_golden(path)
    algo, K = str(g["cfg_algorithm"]), int(g["cfg_K"])
    strict, max_iter = bool(g["cfg_strict"]), int(g["cfg_max_iter"])
    a = _make(algo, g["in_y"], g["in_cap"], g["in_comm"], K, strict, max_iter, dev)
    # feed the permutations the reference drew, in order; anything beyond them is filler that an exact
    # implementation never consumes
    golden = [torch.from_numpy(p.astype(np.int64)) for p in g["in_perms"]]
    calls = {"n": 0}
    C = int(g["cfg_C"])

    def fake_randperm(n, *args, **kw):
        i = calls["n"]
        calls["n"] += 1
        return golden[i].clone() if i < len(golden) else torch.arange(n - 1, -1, -1)

    monkeypatch.setattr(torch, "randperm", fake_randperm)
    paths, is_feasible = a.search()
    want = _unpack(g["out_paths"], g["out_path_len"])
    assert paths == want
    assert [int(v) for v in a.feasible_per_graph.cpu()] == [int(v) for v in g["out_feasible"]]
    assert is_feasible == bool(g["out_feasible"][-1])
    if algo in ("standard", "greedy"):
        assert int(a.attempts.sum()) == len(golden)
        # pre-draw of B*max_iter plus exactly the reference's number of draws afterwards
        assert calls["n"] == int(g["cfg_B"]) * max_iter + len(golden)


def test_seeded_standard_search_consumes_generator_like_reference():
    """With torch.manual_seed the class draws the same permutations as the reference did and leaves the
    global generator in the state the reference would leave it in."""
    dev = _dev()
    g = load_golden([p for p in golden_files("beam") if p.endswith("beam_std_tight.npz")][0])
    a = _make("standard", g["in_y"], g["in_cap"], g["in_comm"], int(g["cfg_K"]), True, int(g["cfg_max_iter"]), dev)
    torch.manual_seed(14)             # the seed gen_golden.py used for this case
    paths, _ = a.search()
    after = torch.rand(1).item()
    assert paths == _unpack(g["out_paths"], g["out_path_len"])
    torch.manual_seed(14)
    for _ in range(len(g["in_perms"])):
        torch.randperm(int(g["cfg_C"]))
    assert torch.rand(1).item() == after


def test_scores_from_logits_close_to_torch():
    dev = _dev()
    for name in ("beam_det_strict.npz", "beam_scores4d.npz"):
        g = load_golden([p for p in golden_files("beam") if p.endswith(name)][0])
        a = _make("deterministic", g["in_logits"], g["in_cap"], g["in_comm"], 2, True, 5, dev, use_y=False)
        got = a.y.cpu().numpy()
        assert got.shape == g["in_y"].shape
        assert np.abs(got - g["in_y"]).max() <= 1e-5 * max(1.0, np.abs(g["in_y"]).max())
        assert not (got == 0).any()


@pytest.mark.parametrize("algo,strict,V,C,K,B", [("deterministic", True, 14, 20, 10, 40), ("standard", True, 14, 10, 10, 60),
                                                  ("unconstrained", False, 14, 20, 10, 20), ("greedy", True, 30, 6, 5, 16),
                                                  ("deterministic", False, 40, 5, 10, 20), ("deterministic", True, 100, 4, 10, 4),
                                                  ("standard", True, 6, 4, 3, 33), ("deterministic", True, 5, 3, 50, 3),
                                                  # more beams than lanes: no adjacency lists (dense grid in LDS at
                                                  # V = 5, in registers at V = 3)
                                                  ("deterministic", True, 5, 3, 70, 3), ("deterministic", True, 3, 2, 80, 3)])
def test_beam_matches_c_oracle_at_benchmark_sizes(algo, strict, V, C, K, B):
    """Random instances at and beyond the configs' sizes (V = 100 exercises the capacity matrix in
    global scratch instead of LDS; K = 50 > candidates exercises short queues)."""
    from src.common.synthetic import make_batch
    dev = _dev()
    rng = np.random.default_rng(V * 100 + C)
    b = make_batch(B, V, C, "nsfnet" if V == 14 else "random", 300 if algo == "standard" else 1000, 5000, 5, 500, seed=V + C)
    y = np.log(rng.random((B, V, V, C)).astype(np.float32) * 0.98 + 0.01).astype(np.float32)
    y[rng.random(y.shape) < 0.2] = np.float32(-0.5)      # ties
    max_iter = 5
    perms = np.stack([rng.permutation(C) for _ in range(B * max_iter)]).astype(np.int32)
    rp, rl, rf, ratt, used = beam_oracle_c.search(y, b["edges_capacity"], b["commodities"], K, algo, strict, max_iter, perms)
    a = _make(algo, y, b["edges_capacity"], b["commodities"], K, strict, max_iter, dev)
    it = iter([torch.from_numpy(p.astype(np.int64)) for p in perms])
    orig = torch.randperm
    torch.randperm = lambda n, *args, **kw: next(it, torch.arange(n))
    try:
        a.search()
    finally:
        torch.randperm = orig
    assert np.array_equal(a.paths_tensor.cpu().numpy(), rp)
    assert np.array_equal(a.path_len.cpu().numpy(), rl)
    assert np.array_equal(a.feasible_per_graph.cpu().numpy().astype(np.int32), rf)
    if algo in ("standard", "greedy"):
        assert int(a.attempts.sum()) == used


@pytest.mark.parametrize("algo,strict,V,C,K,B,shift", [("deterministic", True, 14, 12, 10, 12, 1.5), ("deterministic", True, 60, 5, 10, 4, 0.7),
                                                        ("unconstrained", False, 30, 6, 8, 5, 2.0), ("deterministic", True, 30, 6, 4, 6, 0.0)])
def test_beam_scores_above_zero_are_searched_to_the_end(algo, strict, V, C, K, B, shift):
    """With log-probabilities (all scores <= 0) a search stops once no beam can beat the best result; scores above
    zero can rise along a path, so such inputs must take the reference's run-until-empty walk.  shift moves part of the
    scores above zero (0.0: the largest score is exactly zero, still non-positive)."""
    from src.common.synthetic import make_batch
    dev = _dev()
    rng = np.random.default_rng(V * 7 + C)
    b = make_batch(B, V, C, "nsfnet" if V == 14 else "random", 1000, 5000, 5, 500, seed=3 * V + C)
    y = (np.log(rng.random((B, V, V, C)).astype(np.float32) * 0.98 + 0.01) + np.float32(shift)).astype(np.float32)
    if shift == 0.0:
        y[rng.random(y.shape) < 0.1] = np.float32(0.0)
    else:
        assert (y > 0).any() and (y < 0).any()
    rp, rl, rf, _, _ = beam_oracle_c.search(y, b["edges_capacity"], b["commodities"], K, algo, strict)
    a = _make(algo, y, b["edges_capacity"], b["commodities"], K, strict, 5, dev)
    a.search()
    assert np.array_equal(a.paths_tensor.cpu().numpy(), rp)
    assert np.array_equal(a.path_len.cpu().numpy(), rl)
    assert np.array_equal(a.feasible_per_graph.cpu().numpy().astype(np.int32), rf)


@pytest.mark.parametrize("V,density,K,C,B,strict", [(40, 0.8, 10, 6, 6, True), (60, 0.33, 16, 5, 4, True),
                                                      (60, 0.33, 16, 5, 4, False), (48, 1.0, 8, 4, 3, True)])
def test_beam_dense_graphs_match_c_oracle(V, density, K, C, B, strict):
    """Graphs that are not sparse.  The kernels examine the beams' out-edges through adjacency lists when the dense
    K x V candidate grid exceeds a wavefront's registers; these cases cover what lies beside that path: graphs with more
    edges than the lists hold (density 0.8 and 1.0: the dense grid in LDS), and graphs whose lists are in use but whose
    beams together have more than 256 out-edges in some steps and fewer in others (density 0.33 at K = 16), so that
    one search alternates between the two."""
    dev = _dev()
    rng = np.random.default_rng(V * 7 + K)
    cap = np.zeros((B, V, V), np.float32)
    for b in range(B):
        m = np.triu(rng.random((V, V)) < density, 1)
        m = m | m.T
        ring = np.arange(V)
        m[ring, (ring + 1) % V] = True
        m[(ring + 1) % V, ring] = True
        c = np.triu(rng.integers(300, 3000, size=(V, V)).astype(np.float32), 1)
        cap[b] = (c + c.T) * m
    comm = np.zeros((B, C, 3), np.int64)
    for b in range(B):
        pairs = set()
        while len(pairs) < C:
            s_, t_ = (int(v) for v in rng.integers(0, V, size=2))
            if s_ != t_:
                pairs.add((s_, t_))
        comm[b] = np.asarray(sorted((s_, t_, int(rng.integers(5, 500))) for s_, t_ in pairs))
    y = np.log(rng.random((B, V, V, C)).astype(np.float32) * 0.98 + 0.01).astype(np.float32)
    y[rng.random(y.shape) < 0.2] = np.float32(-0.5)      # ties
    rp, rl, rf, _, _ = beam_oracle_c.search(y, cap, comm, K, "deterministic", strict, 5)
    a = _make("deterministic", y, cap, comm, K, strict, 5, dev)
    a.search()
    assert np.array_equal(a.paths_tensor.cpu().numpy(), rp)
    assert np.array_equal(a.path_len.cpu().numpy(), rl)
    assert np.array_equal(a.feasible_per_graph.cpu().numpy().astype(np.int32), rf)


@pytest.mark.parametrize("algo,V,C,K,B,seed", [("deterministic", 14, 10, 10, 30, 1), ("standard", 14, 20, 10, 30, 2),
                                               ("greedy", 14, 10, 10, 30, 3), ("deterministic", 40, 12, 5, 10, 4),
                                               ("standard", 100, 6, 4, 4, 5)])
def test_beam_tight_capacities_match_c_oracle(algo, V, C, K, B, seed):
    """Strict mode with capacities of the order of the demands: earlier commodities' paths change which edges later
    ones may use, attempts fail and are retried.  This exercises both branches of the ordered pass (keeping the result
    of the parallel first pass, and searching again on the updated capacities) against the sequential C oracle."""
    from src.common.synthetic import make_batch
    dev = _dev()
    rng = np.random.default_rng(1000 + seed)
    b = make_batch(B, V, C, "nsfnet" if V == 14 else "random", 100, 700, 5, 500, seed=seed)
    y = np.log(rng.random((B, V, V, C)).astype(np.float32) * 0.98 + 0.01).astype(np.float32)
    y[rng.random(y.shape) < 0.2] = np.float32(-0.5)
    max_iter = 5
    perms = np.stack([rng.permutation(C) for _ in range(B * max_iter)]).astype(np.int32)
    rp, rl, rf, ratt, used = beam_oracle_c.search(y, b["edges_capacity"], b["commodities"], K, algo, True, max_iter, perms)
    a = _make(algo, y, b["edges_capacity"], b["commodities"], K, True, max_iter, dev)
    it = iter([torch.from_numpy(p.astype(np.int64)) for p in perms])
    orig = torch.randperm
    torch.randperm = lambda n, *args, **kw: next(it, torch.arange(n))
    try:
        a.search()
    finally:
        torch.randperm = orig
    assert np.array_equal(a.paths_tensor.cpu().numpy(), rp)
    assert np.array_equal(a.path_len.cpu().numpy(), rl)
    assert np.array_equal(a.feasible_per_graph.cpu().numpy().astype(np.int32), rf)
    if algo in ("standard", "greedy"):
        assert int(a.attempts.sum()) == used


@pytest.mark.parametrize("algo,cap_lo,cap_hi", [("deterministic", 1000, 5000), ("standard", 1000, 5000), ("deterministic", 300, 1500)])
def test_beam_at_200_nodes_100_commodities_matches_c_oracle(algo, cap_lo, cap_hi):
    """The largest benchmark configuration (V = 200, C = 100, K = 10, strict): 160 KB capacity matrices live in the
    workspace instead of LDS, paths of up to 200 nodes, 100 sequentially dependent commodities per graph.  Scores
    favour each commodity's labelled path (as a trained model's would) with random decoys tied with the label edges,
    so that some graphs decode completely and others run into retries and the fallback; bit-exact against the
    sequential C oracle."""
    from src.common.synthetic import make_batch
    dev = _dev()
    B, V, C, K, max_iter = 2, 200, 100, 10, 5
    b = make_batch(B, V, C, "random", cap_lo, cap_hi, 5, 500, seed=300)
    rng = np.random.default_rng(V * 100 + C)
    p = rng.random((B, V, V, C)).astype(np.float32) * 0.3 + 0.01
    p[b["edges_target"] == 1] = 0.9
    p[rng.random(p.shape) < 0.02] = 0.9
    y = np.log(p).astype(np.float32)
    perms = np.stack([rng.permutation(C) for _ in range(B * max_iter)]).astype(np.int32)
    rp, rl, rf, ratt, used = beam_oracle_c.search(y, b["edges_capacity"], b["commodities"], K, algo, True, max_iter, perms)
    a = _make(algo, y, b["edges_capacity"], b["commodities"], K, True, max_iter, dev)
    it = iter([torch.from_numpy(q.astype(np.int64)) for q in perms])
    orig = torch.randperm
    torch.randperm = lambda n, *args, **kw: next(it, torch.arange(n))
    try:
        a.search()
    finally:
        torch.randperm = orig
    assert np.array_equal(a.path_len.cpu().numpy(), rl)
    assert np.array_equal(a.paths_tensor.cpu().numpy(), rp)
    assert np.array_equal(a.feasible_per_graph.cpu().numpy().astype(np.int32), rf)
    if algo == "standard":
        assert int(a.attempts.sum()) == used
    if cap_lo == 1000:
        assert rf.max() == 1 and rl.max() > 10          # at least one graph decoded completely, with long paths


@pytest.mark.parametrize("algo,V,C,K,B,cap_hi,seed", [("standard", 120, 24, 6, 3, 700, 11), ("greedy", 120, 32, 4, 3, 700, 12),
                                                      ("deterministic", 96, 40, 8, 3, 1200, 13), ("standard", 128, 17, 10, 2, 400, 14)])
def test_beam_ordered_walk_in_rounds_matches_c_oracle(algo, V, C, K, B, cap_hi, seed):
    """Graphs whose capacity matrix does not fit LDS, 16 commodities or more, strict mode: the ordered walk runs in
    rounds (walk until a commodity's edge predicate differs from the snapshot, search all remaining commodities in
    parallel against the capacities reached, walk on).  Capacities of the order of the demands make predicates change
    often, attempts fail and restart from the initial capacities in the middle of the rounds, and some graphs end in
    the fallback; paths, lengths, feasibility and the number of attempts must equal the sequential C oracle's."""
    from src.common.synthetic import make_batch
    dev = _dev()
    rng = np.random.default_rng(2000 + seed)
    b = make_batch(B, V, C, "random", 100, cap_hi, 5, 500, seed=seed)
    p = rng.random((B, V, V, C)).astype(np.float32) * 0.5 + 0.01
    p[b["edges_target"] == 1] = 0.9
    y = np.log(p).astype(np.float32)
    max_iter = 5
    perms = np.stack([rng.permutation(C) for _ in range(B * max_iter)]).astype(np.int32)
    rp, rl, rf, ratt, used = beam_oracle_c.search(y, b["edges_capacity"], b["commodities"], K, algo, True, max_iter, perms)
    a = _make(algo, y, b["edges_capacity"], b["commodities"], K, True, max_iter, dev)
    it = iter([torch.from_numpy(q.astype(np.int64)) for q in perms])
    orig = torch.randperm
    torch.randperm = lambda n, *args, **kw: next(it, torch.arange(n))
    try:
        a.search()
    finally:
        torch.randperm = orig
    assert np.array_equal(a.path_len.cpu().numpy(), rl)
    assert np.array_equal(a.paths_tensor.cpu().numpy(), rp)
    assert np.array_equal(a.feasible_per_graph.cpu().numpy().astype(np.int32), rf)
    if algo in ("standard", "greedy"):
        assert int(a.attempts.sum()) == used
