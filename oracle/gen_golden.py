#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Run in the build container only (the reference checkout is not present on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [--ref /root/reference]

It imports the reference's model, beam search and path sampler unchanged, feeds them seeded synthetic
inputs with the value ranges of the reference's configs, and stores inputs, weights and the
reference's outputs as .npz files.  Nothing of the reference's source is stored, only data.
Model cases run in eval mode with autograd on (dropout off; BatchNorm still uses batch
statistics because track_running_stats=False, src/gcn/models/gcn_layers.py:10,30).
"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")


# ----------------------------------------------------------------------------- synthetic inputs
def make_graph_batch(rng, B, V, C, cap_lo, cap_hi, dem_lo, dem_hi, extra_edges=0.15):
    """Symmetric capacitated graphs (ring + random chords, equal capacity both ways, zero
    diagonal, 0 = no edge) and commodities (distinct src != dst, sorted by demand descending,
    src/common/graph/graph_utils.py:269-281)."""
    cap = np.zeros((B, V, V), dtype=np.float32)
    comm = np.zeros((B, C, 3), dtype=np.int64)
    for b in range(B):
        for i in range(V):
            j = (i + 1) % V
            c = rng.integers(cap_lo, cap_hi)
            cap[b, i, j] = cap[b, j, i] = c
        for i in range(V):
            for j in range(i + 2, V):
                if rng.random() < extra_edges and cap[b, i, j] == 0:
                    c = rng.integers(cap_lo, cap_hi)
                    cap[b, i, j] = cap[b, j, i] = c
        pairs = set()
        rows = []
        while len(rows) < C:
            s, t = rng.integers(0, V, size=2)
            if s == t or (s, t) in pairs:
                continue
            pairs.add((s, t))
            rows.append([s, t, rng.integers(dem_lo, dem_hi + 1)])
        rows.sort(key=lambda r: -r[2])
        comm[b] = np.asarray(rows)
    return cap, comm


def nodes_from_commodities(comm, V):
    """x_nodes[b, src, c] = 1, x_nodes[b, dst, c] = 2 (src/common/data_management/dataset_reader.py:119-123)."""
    B, C, _ = comm.shape
    x_nodes = np.zeros((B, V, C), dtype=np.int64)
    for b in range(B):
        for c in range(C):
            x_nodes[b, comm[b, c, 0], c] = 1
            x_nodes[b, comm[b, c, 1], c] = 2
    return x_nodes


# ----------------------------------------------------------------------------- model cases
def run_model_case(name, B, V, C, H, L, mlp_layers=2, aggregation="mean", voc=2, mask=False,
                   seed=0, keep_layers=True, cap_rng=(1000, 10000), dem_rng=(1, 500)):
    from src.gcn.models.gcn_model import ResidualGatedGCNModel
    from src.gcn.models.model_utils import edge_error
    from src.common.config.config import Settings

    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    cfg = dict(num_nodes=V, num_commodities=C, voc_edges_in=3, voc_edges_out=voc, hidden_dim=H,
               num_layers=L, mlp_layers=mlp_layers, aggregation=aggregation)
    model = ResidualGatedGCNModel(Settings(cfg), torch.float32, torch.long)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "batch_norm.weight" in n:
                p.add_(0.2 * torch.randn_like(p))
            if "batch_norm.bias" in n:
                p.add_(0.1 * torch.randn_like(p))
    model.eval()
    cap, comm = make_graph_batch(rng, B, V, C, cap_rng[0], cap_rng[1], dem_rng[0], dem_rng[1])
    x_nodes = nodes_from_commodities(comm, V)
    y_edges = (rng.random((B, V, V, C)) < 0.12).astype(np.int64)
    n1 = y_edges.sum()
    n = y_edges.size
    edge_cw = np.asarray([n / (2.0 * (n - n1)), n / (2.0 * n1)], dtype=np.float32)  # sklearn "balanced"
    x_edges = (cap > 0).astype(np.int64)

    t = lambda a: torch.from_numpy(a)
    out = {"cfg_B": B, "cfg_V": V, "cfg_C": C, "cfg_H": H, "cfg_L": L, "cfg_mlp_layers": mlp_layers,
           "cfg_aggregation": aggregation, "cfg_voc": voc, "cfg_mask": int(mask),
           "in_cap": cap, "in_comm": comm, "in_x_nodes": x_nodes, "in_y_edges": y_edges, "in_edge_cw": edge_cw}
    for k, v in model.state_dict().items():
        out["param." + k] = v.detach().numpy().copy()

    # hooks to capture per-layer outputs of the reference
    layer_out = []
    hooks = [lyr.register_forward_hook(lambda m, i, o: layer_out.append(o)) for lyr in model.gcn_layers]
    demand = t(comm[:, :, 2].astype(np.float32))
    if voc == 2:
        y_pred, loss = model(t(x_edges), demand, t(cap), t(x_nodes), t(y_edges), edge_cw)
        loss.backward()
        err, err_idx = edge_error(y_pred, t(y_edges), t(x_edges))
        out["out_loss"] = np.float32(loss.item())
        out["out_edge_error"] = np.float64(err)
        out["out_err_idx"] = err_idx.numpy()
    else:
        # score head (voc_edges_out=1): no loss inside the model; backprop a fixed linear functional
        y_pred, loss = model(t(x_edges), demand, t(cap), t(x_nodes), compute_loss=False, mask_invalid_edges=mask)
        assert loss is None
        R = rng.standard_normal(tuple(y_pred.shape)).astype(np.float32)
        out["in_dy_pred"] = R
        (y_pred * t(R)).sum().backward()
    for h in hooks:
        h.remove()
    out["out_y_pred"] = y_pred.detach().numpy()
    for k, p in model.named_parameters():
        out["grad." + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
    out["out_x_last"] = layer_out[-1][0].detach().numpy()
    if keep_layers:
        for l, (x_l, e_l) in enumerate(layer_out):
            out[f"layer{l}.x"] = x_l.detach().numpy()
            out[f"layer{l}.e"] = e_l.detach().numpy()
    else:
        out["out_e_last_slice"] = layer_out[-1][1].detach().numpy()[0, :3]  # [3,V,C,H] slice of graph 0
    path = os.path.join(OUT, f"model_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}: loss={out.get('out_loss')} -> {path} ({os.path.getsize(path) / 1e6:.2f} MB)")


# ----------------------------------------------------------------------------- beam-search cases
def run_beam_case(name, B, V, C, K, algorithm, strict, seed, cap_rng=(1000, 5000), dem_rng=(5, 500),
                  quantize=None, voc=2, max_iter=5, extra_edges=0.2):
    from src.gcn.algorithms.beamsearch_uelb import BeamSearchFactory

    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    cap, comm = make_graph_batch(rng, B, V, C, cap_rng[0], cap_rng[1], dem_rng[0], dem_rng[1], extra_edges)
    if voc == 2:
        logits = rng.standard_normal((B, V, V, C, 2)).astype(np.float32) * 2.0
    else:
        logits = rng.standard_normal((B, V, V, C)).astype(np.float32) * 2.0
    if quantize:
        logits = np.round(logits * quantize) / quantize          # few distinct values -> tied scores
    algo = BeamSearchFactory.create_algorithm(
        algorithm, y_pred_edges=torch.from_numpy(logits.copy()), beam_size=K, batch_size=B,
        edges_capacity=torch.from_numpy(cap), commodities=torch.from_numpy(comm),
        dtypeFloat=torch.float32, dtypeLong=torch.long, mode_strict=strict, max_iter=max_iter)
    y = algo.y.numpy().copy()
    if quantize:
        # force exact ties in the scores themselves as well
        y = (np.round(y * 4.0) / 4.0).astype(np.float32)
        y[y == 0] = np.float32(-1e-20)
        algo.y = torch.from_numpy(y.copy())
    drawn = []
    orig_randperm = torch.randperm

    def recording_randperm(n, *a, **kw):
        p = orig_randperm(n, *a, **kw)
        drawn.append(p.numpy().astype(np.int32).copy())
        return p

    torch.randperm = recording_randperm
    try:
        paths, feas = [], []
        for b in range(B):
            _, p, f = algo._search_single_batch(b)
            paths.append(p)
            feas.append(bool(f))
    finally:
        torch.randperm = orig_randperm
    PL = max(V, 10)
    parr = -np.ones((B, C, PL), dtype=np.int32)
    plen = np.zeros((B, C), dtype=np.int32)
    for b in range(B):
        for c in range(C):
            p = [int(v) for v in paths[b][c]]
            parr[b, c, :len(p)] = p
            plen[b, c] = len(p)
    perms = np.stack(drawn) if drawn else np.zeros((0, C), dtype=np.int32)
    out = dict(cfg_B=B, cfg_V=V, cfg_C=C, cfg_K=K, cfg_algorithm=algorithm, cfg_strict=int(strict),
               cfg_max_iter=max_iter, in_logits=logits, in_y=y, in_cap=cap, in_comm=comm, in_perms=perms,
               out_paths=parr, out_path_len=plen, out_feasible=np.asarray(feas, dtype=np.int32))
    path = os.path.join(OUT, f"beam_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}: feasible={feas} perms={len(drawn)} -> {path} ({os.path.getsize(path) / 1e3:.1f} kB)")

# ----------------------------------------------------------------------------- path-sampler cases
def run_sampler_case(name, B, V, C, temperature, top_p, eps, seed, isolate_dst=False, spike=False, extra_edges=0.2):
    """Runs the reference's PathSampler once and stores its sampled paths, log-probabilities, step entropies,
    reachability and the autograd gradient of a fixed linear functional of the log-probabilities."""
    from src.gcn.algorithms.path_sampler import PathSampler

    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    cap, comm = make_graph_batch(rng, B, V, C, 1000, 5000, 5, 500, extra_edges)
    if isolate_dst:
        # graph 0: the destination of commodity 0 loses all its edges (unreachable -> the path gets stuck)
        d = comm[0, 0, 1]
        cap[0, d, :] = 0
        cap[0, :, d] = 0
    logits = rng.standard_normal((B, V, V, C)).astype(np.float32) * 2.0
    if spike:
        # a few huge logits: rows whose whole mass sits on one edge, so that after masking the rest underflows
        for _ in range(3 * B * C):
            b, i, j, c = rng.integers(0, B), rng.integers(0, V), rng.integers(0, V), rng.integers(0, C)
            logits[b, i, j, c] = 300.0
    logits[np.broadcast_to((cap == 0)[..., None], logits.shape)] = -1e9         # what the model's own mask does
    y = torch.from_numpy(logits.copy()).requires_grad_(True)
    sampler = PathSampler(y_pred_edges=y, edges_capacity=torch.from_numpy(cap), commodities=torch.from_numpy(comm),
                          num_samples=1, temperature=temperature, top_p=top_p, entropy_epsilon=eps)
    paths, logp, feasible, ents = sampler.sample()
    w = rng.standard_normal(B).astype(np.float32)
    (logp * torch.from_numpy(w)).sum().backward()
    PL = max(V, 10)
    parr = -np.ones((B, C, PL), dtype=np.int32)
    plen = np.zeros((B, C), dtype=np.int32)
    earr = np.zeros((B, C, PL), dtype=np.float32)
    nent = np.zeros((B, C), dtype=np.int32)
    for b in range(B):
        for c in range(C):
            p = [int(v) for v in paths[b][c]]
            parr[b, c, :len(p)] = p
            plen[b, c] = len(p)
            earr[b, c, :len(ents[b][c])] = ents[b][c]
            nent[b, c] = len(ents[b][c])
    out = dict(cfg_B=B, cfg_V=V, cfg_C=C, cfg_temperature=np.float32(temperature), cfg_top_p=np.float32(top_p),
               cfg_eps=np.float32(eps), in_logits=logits, in_cap=cap, in_comm=comm, in_w=w, out_paths=parr,
               out_path_len=plen, out_logp=logp.detach().numpy(), out_feasible=np.int32(bool(feasible)),
               out_entropy=earr, out_n_entropy=nent, out_dlogits=y.grad.numpy().copy(),
               out_reach=sampler.reachability.numpy().astype(np.uint8))
    path = os.path.join(OUT, f"sampler_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"sampler_{name}: logp={logp.detach().numpy().round(3)} feasible={feasible} -> {path} "
          f"({os.path.getsize(path) / 1e3:.1f} kB)")


def run_sampler_freq_case(name, V, temperature, top_p, eps, seed, draws):
    """First-hop frequencies of the reference's sampler over many draws on one graph with one commodity."""
    from src.gcn.algorithms.path_sampler import PathSampler

    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    cap, comm = make_graph_batch(rng, 1, V, 1, 1000, 5000, 5, 500, 0.5)
    logits = rng.standard_normal((1, V, V, 1)).astype(np.float32)
    logits[np.broadcast_to((cap == 0)[..., None], logits.shape)] = -1e9
    counts = np.zeros(V, dtype=np.int64)
    y = torch.from_numpy(logits.copy())
    for _ in range(draws):
        sampler = PathSampler(y_pred_edges=y, edges_capacity=torch.from_numpy(cap), commodities=torch.from_numpy(comm),
                              num_samples=1, temperature=temperature, top_p=top_p, entropy_epsilon=eps)
        paths, _, _, _ = sampler.sample()
        counts[int(paths[0][0][1])] += 1
    out = dict(cfg_V=V, cfg_temperature=np.float32(temperature), cfg_top_p=np.float32(top_p), cfg_eps=np.float32(eps),
               cfg_draws=draws, in_logits=logits, in_cap=cap, in_comm=comm, out_first_hop_counts=counts)
    path = os.path.join(OUT, f"sampler_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"sampler_{name}: counts={counts.tolist()} -> {path}")

def run_rl_loss_case(name, B, V, C, seed, calls=2, tight=False, **rl):
    """Runs the reference's ReinforcementLearningStrategy.compute_loss on fixed edge scores (a stub model returns
    them, masked like the real model) `calls` times in a row, so that the moving baseline is exercised, and stores
    the sampled paths, rewards, advantages, loss terms and d loss / d scores of every call."""
    from src.gcn.training.reinforcement_strategy import ReinforcementLearningStrategy

    class FixedScores(torch.nn.Module):
        def __init__(self, y):
            super().__init__()
            self.y = torch.nn.Parameter(y)

        def forward(self, x_edges, x_commodities, cap, x_nodes, y_edges=None, edge_cw=None, compute_loss=False,
                    mask_invalid_edges=False):
            y = self.y
            if mask_invalid_edges:
                y = y.masked_fill((cap.unsqueeze(-1) == 0).expand_as(y), -1e9)
            return y, None

    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    cfg = dict(batch_size=B, num_nodes=V, num_commodities=C, rl_debug_logging=False, **rl)
    strat = ReinforcementLearningStrategy(cfg)
    out = dict(cfg_B=B, cfg_V=V, cfg_C=C, cfg_calls=calls, cfg_json=np.asarray(__import__("json").dumps(cfg)))
    PL = max(V, 10)
    for k in range(calls):
        if tight:
            cap, comm = make_graph_batch(rng, B, V, C, 300, 700, 100, 500, 0.3)
        else:
            cap, comm = make_graph_batch(rng, B, V, C, 1000, 5000, 5, 500, 0.2)
        logits = rng.standard_normal((B, V, V, C)).astype(np.float32) * 1.5
        model = FixedScores(torch.from_numpy(logits.copy()))
        model.train()
        strat.set_epoch(k)
        data = dict(x_edges=torch.from_numpy((cap > 0).astype(np.int64)), x_commodities=torch.from_numpy(comm[:, :, 2].astype(np.float32)),
                    x_edges_capacity=torch.from_numpy(cap), x_nodes=torch.from_numpy(nodes_from_commodities(comm, V)),
                    batch_commodities=torch.from_numpy(comm))
        loss, m = strat.compute_loss(model, data)
        loss.backward()
        parr = -np.ones((B, C, PL), dtype=np.int32)
        plen = np.zeros((B, C), dtype=np.int32)
        for b in range(B):
            for c in range(C):
                p = [int(v) for v in m['pred_paths'][b][c]]
                parr[b, c, :len(p)] = p
                plen[b, c] = len(p)
        pre = f"call{k}."
        out.update({pre + "in_logits": logits, pre + "in_cap": cap, pre + "in_comm": comm, pre + "paths": parr,
                    pre + "path_len": plen, pre + "loss": np.float32(loss.item()), pre + "dlogits": model.y.grad.numpy().copy()})
        for key in ("reward_individual", "advantage_individual", "load_factor_individual"):
            out[pre + key] = np.asarray(m[key], dtype=np.float32)
        for key in ("entropy", "traj_entropy", "entropy_used", "policy_loss", "entropy_bonus", "baseline_before_update",
                    "baseline_after_update", "complete_paths_rate", "finite_solution_rate", "avg_finite_load_factor",
                    "avg_path_length", "capacity_violation_rate", "temperature", "reward", "reward_std", "advantage_std"):
            out[pre + key] = np.float64(m[key])
        out[pre + "fallback"] = np.int32(bool(m["reinforce_fallback_used"]))
        print(f"rl_{name} call {k}: loss={loss.item():.5f} rewards={np.round(m['reward_individual'], 3).tolist()} "
              f"entropy={m['entropy']:.4f} T={m['temperature']}")
    path = os.path.join(OUT, f"rl_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"rl_{name} -> {path} ({os.path.getsize(path) / 1e3:.1f} kB)")

# ----------------------------------------------------------------------------- dataset-reader cases
def run_dataset_case(name, graph_model, V, C, N, seed, batch):
    """Writes a small dataset in the reference's on-disk format with the reference's own graph and commodity
    generators (node flows: hop-shortest paths instead of ILP solutions, which need a solver that is not
    installed), then stores what the reference's DatasetReader.process_batch returns for two batches."""
    import csv
    import random
    import shutil

    import importlib.util

    import networkx as nx
    from src.common.graph.graph_making import Graphs
    from src.common.graph.graph_utils import generate_commodity

    # the reader module alone: its package's __init__ pulls in the ILP solver bindings (python-mip), which are
    # not installed here and which reading a dataset does not need
    spec = importlib.util.spec_from_file_location("ref_dataset_reader", os.path.join("src", "common", "data_management", "dataset_reader.py"))
    ref_reader = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_reader)
    DatasetReader = ref_reader.DatasetReader

    random.seed(seed)
    np.random.seed(seed)
    root = os.path.join(OUT, "datasets")
    mode_dir = os.path.join(root, name, "train_data")
    shutil.rmtree(os.path.join(root, name), ignore_errors=True)
    for i in range(N):
        bucket = i - i % 10
        for d in ("graph_file", "commodity_file", "node_flow_file"):
            os.makedirs(os.path.join(mode_dir, d, str(bucket)), exist_ok=True)
        G = Graphs(C)
        if graph_model == "nsfnet":
            G.nsfnet(G, 1000, 5000)
        else:
            G.randomGraph(G, 3, V, 1000, 10000)
        comm = generate_commodity(G, 5, 500, C)
        nx.write_gml(G, os.path.join(mode_dir, "graph_file", str(bucket), f"graph_{i}.gml"))
        with open(os.path.join(mode_dir, "commodity_file", str(bucket), f"commodity_data_{i}.csv"), "w", newline="") as f:
            csv.writer(f).writerows(comm)
        nodes = list(G.nodes)
        flow = []
        for s, t, _ in comm:
            p = nx.shortest_path(G, s, t)
            row = [0] * len(nodes)
            for k, n in enumerate(p):
                row[nodes.index(n)] = k + 1
            flow.append(row)
        with open(os.path.join(mode_dir, "node_flow_file", str(bucket), f"node_flow_{i}.csv"), "w", newline="") as f:
            csv.writer(f).writerows(flow)
        with open(os.path.join(mode_dir, "exact_solution.csv"), "a", newline="") as f:
            csv.writer(f).writerow([round(random.uniform(0.2, 0.95), 6), round(random.uniform(0.01, 2.0), 4), None])
    cfg = dict(data_root=root, dataset_name=name)
    env = os.environ.pop("GCN_UELB_DATA_ROOT", None)
    try:
        reader = DatasetReader(N, batch, "train", cfg, shuffle=False)
        out = dict(cfg_N=N, cfg_V=V, cfg_C=C, cfg_batch=batch)
        for k, idx in enumerate([np.arange(batch), np.arange(N - batch, N)[::-1].copy()]):
            b = reader.process_batch(idx)
            out[f"b{k}.indices"] = idx
            for key in ("edges", "edges_capacity", "edges_target", "nodes", "nodes_target", "commodities", "load_factor"):
                out[f"b{k}.{key}"] = np.asarray(b[key])
    finally:
        if env is not None:
            os.environ["GCN_UELB_DATA_ROOT"] = env
    path = os.path.join(OUT, f"dataset_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"dataset_{name}: {N} samples -> {mode_dir}; reader output -> {path} ({os.path.getsize(path) / 1e3:.1f} kB)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", default="", help="comma-separated groups to regenerate: model, beam, sampler, rl, dataset (default all)")
    args = ap.parse_args()
    groups = set(args.only.split(",")) if args.only else {"model", "beam", "sampler", "rl", "dataset"}
    sys.dont_write_bytecode = True
    sys.path.insert(0, args.ref)
    os.chdir(args.ref)
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)

    if "sampler" in groups:
        run_sampler_case("greedy", 3, 9, 4, 1.0, 1e-6, 0.05, seed=31)
        run_sampler_case("nsf", 3, 14, 5, 1.3, 0.9, 0.05, seed=32)
        run_sampler_case("noeps_stuck", 2, 10, 4, 0.8, 0.95, 0.0, seed=33, isolate_dst=True)
        run_sampler_case("spike", 3, 8, 4, 1.0, 0.9, 0.05, seed=34, spike=True, extra_edges=0.5)
        run_sampler_case("v70", 1, 70, 3, 1.5, 0.95, 0.05, seed=35, extra_edges=0.05)
        run_sampler_freq_case("freq", 8, 1.2, 0.85, 0.05, seed=36, draws=4000)
    if "rl" in groups:
        run_rl_loss_case("default", 4, 9, 4, seed=41, rl_sampling_temperature=1.5, rl_min_temperature=0.8,
                         rl_temperature_decay=0.035, rl_sampling_top_p=0.95, rl_entropy_weight=0.1, rl_entropy_epsilon=0.05,
                         rl_baseline_momentum=0.95)
        run_rl_loss_case("tight", 5, 8, 6, seed=42, calls=3, tight=True, rl_sampling_top_p=0.9, rl_entropy_weight=0.01,
                         rl_entropy_epsilon=0.0, rl_normalize_advantages=False)
    if "dataset" in groups:
        run_dataset_case("nsf_small", "nsfnet", 14, 4, 12, seed=51, batch=5)
        run_dataset_case("rand_small", "random", 9, 3, 6, seed=52, batch=3)
    if "model" in groups:
        run_model_cases()
    if "beam" in groups:
        run_beam_cases()


def run_model_cases():
    run_model_case("tiny", 2, 5, 3, 8, 2, seed=1)
    run_model_case("h12_mlp3", 2, 6, 4, 12, 2, mlp_layers=3, seed=2)
    run_model_case("sum_agg", 2, 5, 3, 8, 2, aggregation="sum", seed=3)
    run_model_case("voc1_mask", 2, 5, 3, 8, 2, voc=1, mask=True, seed=4)
    run_model_case("nsf", 3, 14, 10, 64, 2, seed=5, keep_layers=False, cap_rng=(1000, 5000), dem_rng=(5, 500))
    run_model_case("wide", 2, 40, 5, 64, 1, seed=6, keep_layers=False)
    run_model_case("h32_l3", 2, 7, 2, 32, 3, seed=7)


def run_beam_cases():
    run_beam_case("det_strict", 4, 14, 10, 10, "deterministic", True, seed=11)
    run_beam_case("det_loose", 3, 10, 5, 3, "deterministic", False, seed=12)
    run_beam_case("std_strict", 4, 14, 10, 5, "standard", True, seed=13)
    run_beam_case("std_tight", 4, 8, 6, 3, "standard", True, seed=14, cap_rng=(300, 700), dem_rng=(100, 500))
    run_beam_case("greedy", 4, 12, 6, 10, "greedy", True, seed=15)
    run_beam_case("greedy_tight", 3, 8, 6, 4, "greedy", True, seed=16, cap_rng=(300, 700), dem_rng=(100, 500))
    run_beam_case("unconstrained", 3, 14, 10, 10, "unconstrained", False, seed=17)
    run_beam_case("ties", 4, 9, 5, 4, "deterministic", True, seed=18, quantize=2)
    run_beam_case("ties_std", 3, 9, 5, 2, "standard", True, seed=19, quantize=2)
    run_beam_case("scores4d", 3, 10, 4, 5, "deterministic", True, seed=20, voc=1)
    run_beam_case("det_tight", 3, 8, 6, 3, "deterministic", True, seed=21, cap_rng=(300, 700), dem_rng=(100, 500))
    run_beam_case("v40", 2, 40, 5, 10, "deterministic", True, seed=22, extra_edges=0.05)


if __name__ == "__main__":
    main()
