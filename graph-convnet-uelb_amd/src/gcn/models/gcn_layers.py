"""Drop-in layer classes for the residual gated graph-conv stack, backed by gfx950 HIP kernels.

Class names, constructor signatures, forward signatures and state_dict keys follow the reference's
``src/gcn/models/gcn_layers.py`` so that checkpoints, ``model_converter`` and the encoders of the
other methods (which import ``ResidualGatedGCNLayer`` and ``MLP``) keep working.  The layer's forward
and backward run in ``libgcn_uelb_hip.so`` (one ``torch.autograd.Function``); the small building
blocks below it are parameter containers whose own ``forward`` is kept only so that code using them
in isolation still runs -- the layer never calls them.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..hip.functions import GCNLayerFn


class BatchNormNode(nn.Module):
    """Per-channel batch statistics over all (batch, node, commodity) rows, in train and eval mode
    (reference: gcn_layers.py:5-22, ``BatchNorm2d(H, track_running_stats=False)``)."""

    def __init__(self, hidden_dim):
        super().__init__()
        self.batch_norm = nn.BatchNorm2d(hidden_dim, track_running_stats=False)

    def forward(self, x):
        return _rows_batch_norm(x, self.batch_norm)


class BatchNormEdge(nn.Module):
    """Same for edge rows (reference: gcn_layers.py:25-43, ``BatchNorm3d``)."""

    def __init__(self, hidden_dim):
        super().__init__()
        self.batch_norm = nn.BatchNorm3d(hidden_dim, track_running_stats=False)

    def forward(self, e):
        return _rows_batch_norm(e, self.batch_norm)


def _rows_batch_norm(t, bn):
    dims = tuple(range(t.dim() - 1))
    mean = t.mean(dim=dims, keepdim=True)
    var = t.var(dim=dims, unbiased=False, keepdim=True)
    return (t - mean) * torch.rsqrt(var + bn.eps) * bn.weight + bn.bias


class NodeFeatures(nn.Module):
    """Holds ``U`` and ``V`` of the node update x_i <- U x_i + agg_j gate_ij * V x_j
    (reference: gcn_layers.py:46-70)."""

    def __init__(self, hidden_dim, aggregation="mean"):
        super().__init__()
        self.aggregation = aggregation
        self.U = nn.Linear(hidden_dim, hidden_dim)
        self.V = nn.Linear(hidden_dim, hidden_dim)

    def forward(self, x, edge_gate):
        num = (edge_gate * self.V(x).unsqueeze(1)).sum(dim=2)
        if self.aggregation == "mean":
            num = num / (1e-20 + edge_gate.sum(dim=2))
        return self.U(x) + num


class EdgeFeatures(nn.Module):
    """Holds ``U`` and ``V`` of the edge update e_ij <- U e_ij + V x_i + V x_j
    (reference: gcn_layers.py:73-94)."""

    def __init__(self, hidden_dim):
        super().__init__()
        self.U = nn.Linear(hidden_dim, hidden_dim)
        self.V = nn.Linear(hidden_dim, hidden_dim)

    def forward(self, x, e):
        vx = self.V(x)
        return self.U(e) + vx.unsqueeze(2) + vx.unsqueeze(1)


class ResidualGatedGCNLayer(nn.Module):
    """One gated graph-conv layer with BatchNorm, ReLU, residual and dropout
    (reference: gcn_layers.py:97-129).

    forward(x [B,V,C,H], e [B,V,V,C,H]) -> (x_new, e_new), fp32, on a ROCm device.
    """

    _next_stream_id = 0

    def __init__(self, hidden_dim, aggregation="sum", dropout_rate=0.3):
        super().__init__()
        self.node_feat = NodeFeatures(hidden_dim, aggregation)
        self.edge_feat = EdgeFeatures(hidden_dim)
        self.bn_node = BatchNormNode(hidden_dim)
        self.bn_edge = BatchNormEdge(hidden_dim)
        self.relu = nn.ReLU(inplace=True)
        self.dropout = nn.Dropout(dropout_rate)
        self.hidden_dim = hidden_dim
        self.aggregation = aggregation
        # distinct dropout streams for distinct layer objects
        self._stream_id = 4096 + ResidualGatedGCNLayer._next_stream_id
        ResidualGatedGCNLayer._next_stream_id += 1

    def hip_parameters(self):
        """The 12 tensors in the order the C ABI expects (= state_dict order)."""
        return [self.node_feat.U.weight, self.node_feat.U.bias, self.node_feat.V.weight, self.node_feat.V.bias,
                self.edge_feat.U.weight, self.edge_feat.U.bias, self.edge_feat.V.weight, self.edge_feat.V.bias,
                self.bn_node.batch_norm.weight, self.bn_node.batch_norm.bias,
                self.bn_edge.batch_norm.weight, self.bn_edge.batch_norm.bias]

    def forward(self, x, e):
        return GCNLayerFn.apply(x, e, self.node_feat.aggregation, float(self.dropout.p), self.training,
                                self._stream_id, *self.hip_parameters())


class MLP(nn.Module):
    """Edge classifier head: (num_layers - 1) x [Linear, ReLU, Dropout] then Linear
    (reference: gcn_layers.py:132-176).  Inside ``ResidualGatedGCNModel`` the head runs fused in the HIP
    library; this ``forward`` serves code that uses an MLP on its own (other methods' heads)."""

    def __init__(self, input_dim, output_dim, num_layers=2, hidden_dims=None, dropout_rate=0.2):
        super().__init__()
        if hidden_dims is None:
            hidden_dims = [input_dim] * (num_layers - 1)
        if len(hidden_dims) != num_layers - 1:
            raise ValueError(f"hidden_dims length ({len(hidden_dims)}) must be num_layers - 1 ({num_layers - 1})")
        blocks, prev = [], input_dim
        for dim in hidden_dims:
            blocks += [nn.Linear(prev, dim), nn.ReLU(inplace=True), nn.Dropout(dropout_rate)]
            prev = dim
        self.layers = nn.Sequential(*blocks)
        self.output_layer = nn.Linear(prev, output_dim)
        self.dropout_rate = dropout_rate
        self.uniform_hidden = all(d == input_dim for d in hidden_dims)

    def hip_parameters(self):
        out = []
        for mod in self.layers:
            if isinstance(mod, nn.Linear):
                out += [mod.weight, mod.bias]
        return out + [self.output_layer.weight, self.output_layer.bias]

    def forward(self, x):
        return self.output_layer(self.layers(x))
