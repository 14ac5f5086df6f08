"""Drop-in ``ResidualGatedGCNModel`` whose forward and backward run in libgcn_uelb_hip.so.

Constructor, ``forward`` signature, attributes and state_dict keys follow the reference's
``src/gcn/models/gcn_model.py:8-123``.  One ``torch.autograd.Function`` covers embedding, the L layers,
the edge MLP, masking, the loss and the edge-error counters, so a training step is two library calls
(forward, backward) that can be captured in a HIP graph.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ..hip import lib as L
from ..hip.functions import AGG_ID, GCNModelFn
from .gcn_layers import MLP, ResidualGatedGCNLayer


class ResidualGatedGCNModel(nn.Module):
    """Residual gated GCN predicting, per commodity, which edges its path uses."""

    def __init__(self, config, dtypeFloat, dtypeLong):
        super().__init__()
        self.dtypeFloat = dtypeFloat
        self.dtypeLong = dtypeLong
        self.num_nodes = config.num_nodes
        self.num_commodities = config.num_commodities
        self.voc_nodes_in = config.num_commodities * 3
        self.voc_nodes_out = config['num_nodes']
        self.voc_edges_in = config['voc_edges_in']
        self.voc_edges_out = config['voc_edges_out']
        self.hidden_dim = config['hidden_dim']
        self.num_layers = config['num_layers']
        self.mlp_layers = config['mlp_layers']
        self.aggregation = config['aggregation']
        # read but not forwarded, exactly as the reference does (gcn_model.py:31,39,42): the layers keep
        # their constructor default 0.3 and the MLP its 0.2
        self.dropout_rate = config.get('dropout_rate', 0.2)
        if self.aggregation not in AGG_ID:
            raise ValueError(f"unknown aggregation {self.aggregation!r}")
        self.nodes_embedding = nn.Embedding(self.voc_nodes_in, self.hidden_dim // 2)
        self.commodities_embedding = nn.Linear(1, self.hidden_dim // 2, bias=False)
        self.edges_values_embedding = nn.Linear(1, self.hidden_dim, bias=False)
        self.gcn_layers = nn.ModuleList(
            [ResidualGatedGCNLayer(self.hidden_dim, self.aggregation) for _ in range(self.num_layers)])
        self.mlp_edges = MLP(self.hidden_dim, self.voc_edges_out, self.mlp_layers)
        # filled by every forward that computes the loss: python-side handles to device scalars
        self.last_edge_error = None      # 0-dim tensor, edge error in percent (model_utils.edge_error)
        self.last_err_idx = None         # bool [B, C]

    def hip_parameters(self):
        """All parameter tensors in the order of the C ABI (= state_dict order)."""
        out = [self.nodes_embedding.weight, self.commodities_embedding.weight, self.edges_values_embedding.weight]
        for layer in self.gcn_layers:
            out += layer.hip_parameters()
        return out + self.mlp_edges.hip_parameters()

    def forward(self, x_edges, x_commodities, x_edges_capacity, x_nodes, y_edges=None, edge_cw=None,
                compute_loss=True, mask_invalid_edges=False):
        """Same contract as the reference's forward (gcn_model.py:45-123).

        x_edges is accepted and unused (the convolution is dense over all V x V pairs).
        x_commodities: [B, C] demands or [B, C, 3] (src, dst, demand).
        Returns (y_pred_edges, loss): [B,V,V,C,voc_edges_out] (last dim squeezed if 1) and a 0-dim loss
        tensor or None.
        """
        if compute_loss and (y_edges is None or edge_cw is None):
            raise ValueError("y_edges and edge_cw required when compute_loss=True")
        if x_commodities.dim() == 3:
            x_commodities = x_commodities[:, :, 2]
        L.require_cuda(x_edges_capacity, "x_edges_capacity")
        B, V = x_edges_capacity.shape[0], x_edges_capacity.shape[1]
        C = x_commodities.shape[1]
        if V != self.num_nodes or C != self.num_commodities:
            raise ValueError(f"input has V={V}, C={C}; the model was built for V={self.num_nodes}, C={self.num_commodities}")
        if compute_loss and self.voc_edges_out != 2:
            # loss_edges takes log_softmax over dim 4 of 5-D logits (model_utils.py:69)
            raise ValueError("compute_loss=True needs voc_edges_out == 2")
        layer_p = float(self.gcn_layers[0].dropout.p) if self.num_layers else 0.0
        desc = dict(B=B, V=V, C=C, H=self.hidden_dim, L=self.num_layers, mlp_layers=self.mlp_layers,
                    voc_out=self.voc_edges_out, n_emb=self.nodes_embedding.num_embeddings,
                    aggregation=AGG_ID[self.aggregation], training=int(self.training),
                    mask_invalid_edges=int(bool(mask_invalid_edges)), has_labels=int(bool(compute_loss)),
                    p_layer=layer_p, p_mlp=float(self.mlp_edges.dropout_rate))
        y_pred, loss_out, flags = GCNModelFn.apply(desc, x_edges_capacity, x_commodities, x_nodes,
                                                   y_edges if compute_loss else None,
                                                   edge_cw if compute_loss else None, *self.hip_parameters())
        if self.voc_edges_out == 1:
            y_pred = y_pred.squeeze(-1)
        loss = None
        if compute_loss:
            loss = loss_out[0]
            self.last_edge_error = loss_out[2].detach()
            self.last_err_idx = flags.view(B, C).bool()
        return y_pred, loss
