"""``mean_feasible_load_factor`` with the reference's signature (``src/common/graph/graph_utils.py:90-192``),
computed on the GPU by ``gcn_load_factor``."""
from __future__ import annotations

import torch

from ...gcn.hip import lib as L


def load_factor_from_tensors(paths, path_len, edges_capacity, commodities, return_usage=False):
    """paths int32 [B,C,PL] (-1 padded), path_len int32 [B,C] on the GPU -> (max_lf [B] float, complete [B,C] bool)
    and, with return_usage, the int32 [B,V,V] demand carried by every edge (the kernel's scratch, exact sums)."""
    lib = L.load()
    dev = paths.device
    B, C, PL = paths.shape
    cap = edges_capacity.to(dev, torch.float32).contiguous()
    comm = commodities.to(dev, torch.int64).contiguous()
    V = cap.shape[1]
    usage = torch.empty((B, V, V), dtype=torch.int32, device=dev)
    max_lf = torch.empty(B, dtype=torch.float32, device=dev)
    complete = torch.empty((B, C), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        L.check(lib.gcn_load_factor(B, V, C, PL, L.ptr(paths.contiguous()), L.ptr(path_len.contiguous()), L.ptr(cap),
                                    L.ptr(comm), L.ptr(usage), L.ptr(max_lf), L.ptr(complete), L.stream_ptr(dev)),
                "gcn_load_factor")
    if return_usage:
        return max_lf, complete.bool(), usage
    return max_lf, complete.bool()


def paths_to_tensors(pred_paths, num_batch, num_flow, num_node, device):
    """list [B][C] of node lists -> (paths int32 [B,C,PL] padded with -1, path_len int32 [B,C]) on `device`."""
    PL = max(num_node, 10, max((len(p) for g in pred_paths for p in g), default=0))
    paths = torch.full((num_batch, num_flow, PL), -1, dtype=torch.int32)
    plen = torch.zeros((num_batch, num_flow), dtype=torch.int32)
    for b, graph in enumerate(pred_paths[:num_batch]):
        for c, p in enumerate(graph[:num_flow]):
            if len(p):
                paths[b, c, :len(p)] = torch.as_tensor(p, dtype=torch.int32)
            plen[b, c] = len(p)
    return paths.to(device), plen.to(device)


def mean_feasible_load_factor(num_batch, num_flow, num_node, pred_paths, edges_capacity, commodities):
    """pred_paths: list [B][C] of node lists.  Returns (mean of the per-graph maximum load factor, per-graph maxima)."""
    dev = commodities.device if commodities.is_cuda else torch.device("cuda", torch.cuda.current_device())
    paths, plen = paths_to_tensors(pred_paths, num_batch, num_flow, num_node, dev)
    max_lf, _ = load_factor_from_tensors(paths, plen, edges_capacity, commodities)
    return max_lf.mean(), max_lf
