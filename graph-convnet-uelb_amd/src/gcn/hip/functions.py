"""torch.autograd.Function wrappers around the C ABI (plumbing: buffers, pointers, streams)."""
from __future__ import annotations

import ctypes

import torch

from . import lib as L

AGG_ID = {"sum": 0, "mean": 1}


def _agg_id(aggregation):
    if aggregation not in AGG_ID:
        raise ValueError(f"unknown aggregation {aggregation!r} (expected 'sum' or 'mean')")
    return AGG_ID[aggregation]


class GCNLayerFn(torch.autograd.Function):
    """ResidualGatedGCNLayer.forward / backward through gcn_layer_fwd / gcn_layer_bwd."""

    @staticmethod
    def forward(ctx, x, e, aggregation, p_drop, training, stream_id, *params):
        lib = L.load()
        x = L.f32c(x, "x")
        e = L.f32c(e, "e")
        if x.dim() != 4 or e.dim() != 5:
            raise ValueError(f"expected x [B,V,C,H] and e [B,V,V,C,H], got {tuple(x.shape)} and {tuple(e.shape)}")
        B, V, C, H = x.shape
        if tuple(e.shape) != (B, V, V, C, H):
            raise ValueError(f"e shape {tuple(e.shape)} does not match x shape {tuple(x.shape)}")
        params = [L.f32c(p.detach(), "parameter") for p in params]
        dev = x.device
        n_saved = lib.gcn_layer_saved_floats(B, V, C, H)
        ws_bytes = lib.gcn_layer_ws_bytes(B, V, C, H)
        if n_saved == 0 or ws_bytes == 0:
            L.check(-1, "gcn_layer_saved_floats")
        saved = torch.empty(n_saved, dtype=torch.float32, device=dev)
        ws = L.workspace(dev, ws_bytes)
        x_out = torch.empty_like(x)
        e_out = torch.empty_like(e)
        use_drop = bool(training) and p_drop > 0
        rng = L.next_rng_snapshot(dev) if use_drop else None
        with torch.cuda.device(dev):
            rc = lib.gcn_layer_fwd(B, V, C, H, _agg_id(aggregation), float(p_drop), int(bool(training)), L.ptr(rng),
                                   int(stream_id), L.ptr(x), L.ptr(e), L.ptr_array(params), L.ptr(x_out), L.ptr(e_out),
                                   L.ptr(saved), L.ptr(ws), ws.numel(), L.stream_ptr(dev))
        L.check(rc, "gcn_layer_fwd")
        ctx.save_for_backward(x, e, saved, *params)
        ctx.cfg = (B, V, C, H, aggregation, float(p_drop), bool(training), int(stream_id))
        ctx.rng = rng
        return x_out, e_out

    @staticmethod
    def backward(ctx, dx_out, de_out):
        lib = L.load()
        x, e, saved, *params = ctx.saved_tensors
        B, V, C, H, aggregation, p_drop, training, stream_id = ctx.cfg
        dev = x.device
        dx = torch.zeros_like(x) if dx_out is None else dx_out.to(torch.float32).contiguous().clone()
        de = torch.zeros_like(e) if de_out is None else de_out.to(torch.float32).contiguous().clone()
        grads = [torch.empty_like(p) for p in params]
        ws = L.workspace(dev, lib.gcn_layer_ws_bytes(B, V, C, H))
        with torch.cuda.device(dev):
            rc = lib.gcn_layer_bwd(B, V, C, H, _agg_id(aggregation), p_drop, int(training), L.ptr(ctx.rng), stream_id,
                                   L.ptr(x), L.ptr(e), L.ptr_array(params), L.ptr(saved), L.ptr(dx), L.ptr(de),
                                   L.ptr_array(grads), L.ptr(ws), ws.numel(), L.stream_ptr(dev))
        L.check(rc, "gcn_layer_bwd")
        return (dx, de, None, None, None, None, *grads)


class GCNModelFn(torch.autograd.Function):
    """ResidualGatedGCNModel.forward / backward through gcn_model_fwd / gcn_model_bwd.

    Returns (y_pred, loss_out, err_flags); loss_out = [loss, sum of class weights, edge error %, 0].
    """

    @staticmethod
    def forward(ctx, desc_kwargs, cap, demand, x_nodes, y_edges, edge_cw, *params):
        lib = L.load()
        ctx.set_materialize_grads(False)
        cap = L.f32c(cap, "x_edges_capacity")
        demand = L.f32c(demand, "x_commodities")
        L.require_cuda(x_nodes, "x_nodes")
        x_nodes = x_nodes.to(torch.int64).contiguous()
        dev = cap.device
        desc = L.ModelDesc(**desc_kwargs)
        has_labels = bool(desc.has_labels)
        if has_labels:
            L.require_cuda(y_edges, "y_edges")
            y_edges = y_edges.to(torch.int64).contiguous()
            edge_cw = torch.as_tensor(edge_cw, dtype=torch.float32).to(dev).contiguous()
            if edge_cw.numel() != 2:
                raise ValueError("edge_cw must hold one weight per class (2)")
        else:
            y_edges, edge_cw = None, None
        params = [L.f32c(p.detach(), "parameter") for p in params]
        nparams = lib.gcn_model_num_params(ctypes.byref(desc))
        if nparams < 0:
            L.check(nparams, "gcn_model_num_params")
        if nparams != len(params):
            raise ValueError(f"expected {nparams} parameter tensors, got {len(params)}")
        B, V, C, voc = desc.B, desc.V, desc.C, desc.voc_out
        n_saved = lib.gcn_model_saved_floats(ctypes.byref(desc))
        ws_bytes = lib.gcn_model_ws_bytes(ctypes.byref(desc))
        saved = torch.empty(n_saved, dtype=torch.float32, device=dev)
        ws = L.workspace(dev, ws_bytes)
        y_pred = torch.empty((B, V, V, C, voc), dtype=torch.float32, device=dev)
        loss_out = torch.zeros(4, dtype=torch.float32, device=dev)
        counts = torch.empty(2 * B * C, dtype=torch.int32, device=dev) if has_labels else None
        flags = torch.empty(B * C, dtype=torch.uint8, device=dev) if has_labels else None
        use_drop = bool(desc.training) and (desc.p_layer > 0 or desc.p_mlp > 0)
        rng = L.next_rng_snapshot(dev) if use_drop else None
        with torch.cuda.device(dev):
            rc = lib.gcn_model_fwd(ctypes.byref(desc), L.ptr(cap), L.ptr(demand), L.ptr(x_nodes), L.ptr(y_edges),
                                   L.ptr(edge_cw), L.ptr_array(params), L.ptr(rng), L.ptr(y_pred), L.ptr(loss_out),
                                   L.ptr(counts), L.ptr(flags), L.ptr(saved), L.ptr(ws), ws.numel(), L.stream_ptr(dev))
        L.check(rc, "gcn_model_fwd")
        ctx.desc_kwargs = dict(desc_kwargs)
        ctx.rng = rng
        ctx.inputs = (cap, demand, x_nodes, y_edges, edge_cw)
        ctx.save_for_backward(saved, loss_out, *params)
        if flags is None:
            flags = torch.zeros(0, dtype=torch.uint8, device=dev)
        ctx.mark_non_differentiable(flags)
        return y_pred, loss_out, flags

    @staticmethod
    def backward(ctx, dy_pred, dloss_out, _dflags):
        lib = L.load()
        saved, loss_out, *params = ctx.saved_tensors
        cap, demand, x_nodes, y_edges, edge_cw = ctx.inputs
        desc = L.ModelDesc(**ctx.desc_kwargs)
        dev = cap.device
        dloss = None
        if dloss_out is not None and desc.has_labels:
            dloss = dloss_out.to(torch.float32).contiguous()          # element 0 is d/d loss
        if dy_pred is not None:
            dy_pred = dy_pred.to(torch.float32).contiguous()
        grads = [torch.empty_like(p) for p in params]
        if dloss is None and desc.has_labels:
            # loss unused downstream: its upstream gradient is zero
            dloss = torch.zeros(1, dtype=torch.float32, device=dev)
        ws = L.workspace(dev, lib.gcn_model_ws_bytes(ctypes.byref(desc)))
        with torch.cuda.device(dev):
            rc = lib.gcn_model_bwd(ctypes.byref(desc), L.ptr(cap), L.ptr(demand), L.ptr(x_nodes), L.ptr(y_edges),
                                   L.ptr(edge_cw), L.ptr_array(params), L.ptr(ctx.rng), L.ptr(saved), L.ptr(loss_out),
                                   L.ptr(dloss), L.ptr(dy_pred), L.ptr_array(grads), L.ptr(ws), ws.numel(),
                                   L.stream_ptr(dev))
        L.check(rc, "gcn_model_bwd")
        return (None, None, None, None, None, None, *grads)
