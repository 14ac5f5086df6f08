"""Loss and metric helpers with the reference's names and signatures
(``src/gcn/models/model_utils.py``): ``loss_edges``, ``edge_error``, ``update_learning_rate``,
``create_edge_class_weights``.  The tensor work runs in libgcn_uelb_hip.so.

The reference's ``beamsearch_uelb_flows_nodes`` / ``beamsearch_tour_nodes*`` helpers are dead code there
(they call ``Beamsearch`` with argument lists its constructor does not accept) and are not provided.
"""
from __future__ import annotations

import numpy as np
import torch

from ..hip import lib as L


def create_edge_class_weights(y_edges):
    """sklearn's ``compute_class_weight("balanced")`` for the labels present: n / (n_classes * count_k)
    (reference :9-23)."""
    labels = y_edges.detach().cpu().numpy().ravel()
    classes, counts = np.unique(labels, return_counts=True)
    return torch.tensor(labels.size / (len(classes) * counts.astype(np.float64)), dtype=torch.float32)


class _LossEdgesFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y_pred_edges, y_edges, edge_cw):
        lib = L.load()
        logits = L.f32c(y_pred_edges, "y_pred_edges")
        if logits.dim() != 5 or logits.shape[4] != 2:
            raise ValueError("loss_edges expects logits of shape [B, V, V, C, 2]")
        B, V, _, C, _ = logits.shape
        dev = logits.device
        y = y_edges.to(dev, torch.int64).contiguous()
        cw = torch.as_tensor(edge_cw, dtype=torch.float32).to(dev).contiguous()
        loss_out = torch.zeros(4, dtype=torch.float32, device=dev)
        flags = torch.empty(B * C, dtype=torch.uint8, device=dev)
        ws = L.workspace(dev, lib.gcn_loss_ws_bytes(B, V, C))
        with torch.cuda.device(dev):
            L.check(lib.gcn_loss_edges_fwd(B, V, C, L.ptr(logits), L.ptr(y), L.ptr(cw), L.ptr(loss_out), L.ptr(flags),
                                           L.ptr(ws), ws.numel(), L.stream_ptr(dev)), "gcn_loss_edges_fwd")
        ctx.save_for_backward(logits, y, cw, loss_out)
        return loss_out[0].clone()

    @staticmethod
    def backward(ctx, dloss):
        lib = L.load()
        logits, y, cw, loss_out = ctx.saved_tensors
        B, V, _, C, _ = logits.shape
        dev = logits.device
        dl = dloss.to(torch.float32).reshape(1).contiguous()
        dlogits = torch.empty_like(logits)
        with torch.cuda.device(dev):
            L.check(lib.gcn_loss_edges_bwd(B, V, C, L.ptr(logits), L.ptr(y), L.ptr(cw), L.ptr(loss_out), L.ptr(dl),
                                           L.ptr(dlogits), L.stream_ptr(dev)), "gcn_loss_edges_bwd")
        return dlogits, None, None


def loss_edges(y_pred_edges, y_edges, edge_cw):
    """Class-weighted NLL of log_softmax over the two classes, mean-reduced by the weight sum
    (reference :51-78).  Differentiable with respect to ``y_pred_edges``."""
    return _LossEdgesFn.apply(y_pred_edges, y_edges, edge_cw)


def edge_error(y_pred, y_target, x_edges=None):
    """(error in percent, bool [B, C] of imperfect slices) as the reference's ``edge_error`` (:255-279);
    ``x_edges`` is accepted and unused there as well."""
    lib = L.load()
    logits = L.f32c(y_pred.detach(), "y_pred")
    B, V, _, C, _ = logits.shape
    dev = logits.device
    y = y_target.to(dev, torch.int64).contiguous()
    loss_out = torch.zeros(4, dtype=torch.float32, device=dev)
    flags = torch.empty(B * C, dtype=torch.uint8, device=dev)
    ws = L.workspace(dev, lib.gcn_loss_ws_bytes(B, V, C))
    with torch.cuda.device(dev):
        L.check(lib.gcn_loss_edges_fwd(B, V, C, L.ptr(logits), L.ptr(y), None, L.ptr(loss_out), L.ptr(flags), L.ptr(ws),
                                       ws.numel(), L.stream_ptr(dev)), "gcn_loss_edges_fwd")
    return float(loss_out[2].item()), flags.view(B, C).bool()


def update_learning_rate(optimizer, lr):
    """Set the learning rate of every parameter group (reference :225-239)."""
    for group in optimizer.param_groups:
        group['lr'] = lr
    return optimizer
