"""CPU oracle for the GCN hot path (TEST INFRASTRUCTURE, never the product path).

Plain PyTorch fp32 restatement of the reference's residual gated graph-conv model,
its loss and its edge-error metric.  It is used only as the checker: by ``tests/``,
by ``__graft_entry__.smoke()`` and by the ``cpu_baseline`` leg of ``bench.py``.
The HIP path never imports or calls it.

Parity status: PINNED.  ``oracle/gen_golden.py`` imports the reference itself and
stores its outputs under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this
restatement against those vectors.

All ``file:line`` citations are relative to the reference checkout.
Parameters are passed as a dict keyed by the reference's ``state_dict`` names
(``gcn_layers.{l}.node_feat.U.weight`` ...), so a reference checkpoint can be used as is.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # torch.nn.BatchNorm{2,3}d default, src/gcn/models/gcn_layers.py:10,30


def batch_norm_rows(t: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor) -> torch.Tensor:
    """BatchNorm over every dim but the last, batch statistics always, biased variance.

    Follows BatchNormNode / BatchNormEdge (src/gcn/models/gcn_layers.py:5-43): both wrap
    ``BatchNorm{2,3}d(H, track_running_stats=False)`` around a transpose that puts the
    hidden dim on the channel axis, so statistics are per hidden channel over all rows,
    in train and in eval mode alike.
    """
    dims = tuple(range(t.dim() - 1))
    mean = t.mean(dim=dims, keepdim=True)
    var = t.var(dim=dims, unbiased=False, keepdim=True)
    return (t - mean) / torch.sqrt(var + BN_EPS) * gamma + beta


def gcn_layer(x, e, P, prefix, aggregation="mean", dropout_p=0.0, training=False, relu_mask=None):
    """One ResidualGatedGCNLayer (src/gcn/models/gcn_layers.py:97-129).

    x: [B,V,C,H], e: [B,V,V,C,H].  Returns (x_new, e_new, aux) where aux holds the
    intermediates the HIP tests compare (e_tmp, gate sums, x_tmp).

    relu_mask = (edge mask, node mask), 0/1 tensors of the shapes of e and x: the two ReLUs are replaced by a
    product with the given pattern.  With the pattern relu itself would produce this changes nothing; tests pass
    the pattern the HIP path decided on, so that BatchNorm outputs within float32 rounding of zero (which two
    correct implementations may put on either side) take the same branch in both.
    """
    g = lambda n: P[prefix + n]
    # EdgeFeatures.forward, gcn_layers.py:81-94
    Ue = F.linear(e, g("edge_feat.U.weight"), g("edge_feat.U.bias"))
    Ve = F.linear(x, g("edge_feat.V.weight"), g("edge_feat.V.bias"))
    e_tmp = Ue + Ve.unsqueeze(2) + Ve.unsqueeze(1)          # [b,i,j,c] gets Ve[b,i,c] + Ve[b,j,c]
    gate = torch.sigmoid(e_tmp)                              # gcn_layers.py:120
    # NodeFeatures.forward, gcn_layers.py:55-70
    Ux = F.linear(x, g("node_feat.U.weight"), g("node_feat.U.bias"))
    Vn = F.linear(x, g("node_feat.V.weight"), g("node_feat.V.bias"))
    num = (gate * Vn.unsqueeze(1)).sum(dim=2)                # sum over j of gate[b,i,j,c] * Vn[b,j,c]
    if aggregation == "mean":
        x_tmp = Ux + num / (1e-20 + gate.sum(dim=2))
    elif aggregation == "sum":
        x_tmp = Ux + num
    else:
        raise ValueError(f"unknown aggregation {aggregation!r}")
    # BatchNorm, ReLU, residual, dropout: gcn_layers.py:122-128
    e_bn = batch_norm_rows(e_tmp, g("bn_edge.batch_norm.weight"), g("bn_edge.batch_norm.bias"))
    x_bn = batch_norm_rows(x_tmp, g("bn_node.batch_norm.weight"), g("bn_node.batch_norm.bias"))
    if relu_mask is None:
        e_act, x_act = F.relu(e_bn), F.relu(x_bn)
    else:
        e_act, x_act = e_bn * relu_mask[0].to(e_bn.dtype), x_bn * relu_mask[1].to(x_bn.dtype)
    e_new = F.dropout(e_act + e, dropout_p, training)
    x_new = F.dropout(x_act + x, dropout_p, training)
    return x_new, e_new, {"e_tmp": e_tmp, "x_tmp": x_tmp, "e_bn": e_bn, "x_bn": x_bn}


def embed(P, x_edges_capacity, x_commodities, x_nodes):
    """Embedding stage of ResidualGatedGCNModel.forward (src/gcn/models/gcn_model.py:68-84)."""
    B, V, _ = x_edges_capacity.shape
    if x_commodities.dim() == 3:                             # gcn_model.py:73-74
        x_commodities = x_commodities[:, :, 2]
    C = x_commodities.shape[1]
    emb = F.embedding(x_nodes, P["nodes_embedding.weight"])  # [B,V,C,H/2]
    c = x_commodities.to(emb.dtype).view(B, 1, C, 1) * P["commodities_embedding.weight"].view(1, 1, 1, -1)
    c = c.expand(B, V, C, -1)
    e = x_edges_capacity.view(B, V, V, 1, 1) * P["edges_values_embedding.weight"].view(1, 1, 1, 1, -1)
    e = e.expand(B, V, V, C, -1)
    x_agg = emb.mean(dim=2, keepdim=True).expand(-1, -1, C, -1)
    x = torch.cat((x_agg, c), dim=3)
    return x.contiguous(), e.contiguous()


def mlp(t, P, prefix, mlp_layers, dropout_p=0.0, training=False):
    """MLP.forward (src/gcn/models/gcn_layers.py:132-176) with hidden_dims=None."""
    for k in range(mlp_layers - 1):
        t = F.linear(t, P[f"{prefix}layers.{3 * k}.weight"], P[f"{prefix}layers.{3 * k}.bias"])
        t = F.dropout(F.relu(t), dropout_p, training)
    return F.linear(t, P[prefix + "output_layer.weight"], P[prefix + "output_layer.bias"])


def loss_edges(y_pred_edges, y_edges, edge_cw):
    """loss_edges (src/gcn/models/model_utils.py:51-78): log_softmax over the class dim,
    class-weighted NLL with mean reduction = sum(w[y] * -logp[y]) / sum(w[y])."""
    logp = F.log_softmax(y_pred_edges, dim=4).permute(0, 4, 1, 2, 3).contiguous()
    return F.nll_loss(logp, y_edges.long(), weight=edge_cw)


def edge_error(y_pred, y_target):
    """edge_error/_edge_error (src/gcn/models/model_utils.py:255-308).

    Returns (err_percent, err_idx[B,C]).  argmax ties go to class 0; a (b,c) slice with an
    empty mask gives 0/0 = NaN exactly as the reference does.
    """
    y = F.softmax(y_pred, dim=4).argmax(dim=4)
    mask = ((y_target + y) > 0).long()
    acc = ((y == y_target).long() * mask)
    acc = acc.sum(dim=1).sum(dim=1).float() / mask.sum(dim=1).sum(dim=1).float()
    err_idx = acc < 1.0
    mean_acc = acc.sum().float().item() / acc.numel()
    return 100.0 * (1.0 - mean_acc), err_idx


def model_forward(P, cfg, x_edges_capacity, x_commodities, x_nodes, y_edges=None, edge_cw=None,
                  compute_loss=True, mask_invalid_edges=False, training=False,
                  layer_dropout=0.3, mlp_dropout=0.2, return_layers=False, relu_masks=None):
    """ResidualGatedGCNModel.forward (src/gcn/models/gcn_model.py:45-123).

    cfg: dict with num_layers, mlp_layers, aggregation, voc_edges_out.  The layer dropout is
    the constructor default 0.3 and the MLP's 0.2 whatever the config's ``dropout_rate`` says
    (gcn_model.py:39,42).
    """
    x, e = embed(P, x_edges_capacity, x_commodities, x_nodes)
    layers = []
    for l in range(cfg["num_layers"]):
        x, e, aux = gcn_layer(x, e, P, f"gcn_layers.{l}.", cfg["aggregation"], layer_dropout, training,
                              None if relu_masks is None else relu_masks[l])
        if return_layers:
            layers.append((x, e, aux))
    y = mlp(e, P, "mlp_edges.", cfg["mlp_layers"], mlp_dropout, training)
    voc = cfg["voc_edges_out"]
    if voc == 1:                                             # gcn_model.py:93-94
        y = y.squeeze(-1)
    if mask_invalid_edges:                                   # gcn_model.py:97-110
        inv = x_edges_capacity == 0
        inv = inv.unsqueeze(-1) if voc == 1 else inv.unsqueeze(-1).unsqueeze(-1)
        y = y.masked_fill(inv.expand_as(y), -1e9)
    loss = None
    if compute_loss:
        if y_edges is None or edge_cw is None:
            raise ValueError("y_edges and edge_cw required when compute_loss=True")
        loss = loss_edges(y, y_edges, torch.as_tensor(edge_cw).to(device=y.device, dtype=y.dtype))
    if return_layers:
        return y, loss, layers
    return y, loss


def model_loss_and_grads_checkpointed(P, cfg, x_edges_capacity, x_commodities, x_nodes, y_edges, edge_cw,
                                      relu_masks=None, dy_pred=None):
    """The same function as ``model_forward`` followed by ``loss.backward()`` (eval mode), arranged so that the
    autograd graph of only one layer is alive at a time: the forward keeps each layer's inputs, the backward re-runs
    one layer at a time under autograd and chains the gradients by hand.  For the large configurations (a float64
    edge tensor of config 4 is 1.2 GB; the plain version keeps about seven of them per layer).

    relu_masks: None, or a callable ``l -> (edge mask, node mask)`` (see ``gcn_layer``), called once in the forward
    and once in the backward of layer l.  With ``y_edges`` None the scalar is sum(y * dy_pred) instead of the loss.
    Returns (y, loss or None, grads) with grads keyed like P (zeros for parameters the result does not depend on).
    """
    L = cfg["num_layers"]
    mask_of = (lambda l: None) if relu_masks is None else relu_masks
    xs, es = [], []
    with torch.no_grad():
        x, e = embed(P, x_edges_capacity, x_commodities, x_nodes)
        for l in range(L):
            xs.append(x)
            es.append(e)
            x, e, _ = gcn_layer(x, e, P, f"gcn_layers.{l}.", cfg["aggregation"], relu_mask=mask_of(l))
    grads = {k: torch.zeros_like(v) for k, v in P.items()}

    def take(names, outs, ins, gouts):
        params = [P[n] for n in names]
        g = torch.autograd.grad(outs, ins + params, gouts, allow_unused=True)
        for n, gv in zip(names, g[len(ins):]):
            if gv is not None:
                grads[n] = gv
        return g[:len(ins)]

    # head
    eL = e.detach().requires_grad_(True)
    y = mlp(eL, P, "mlp_edges.", cfg["mlp_layers"])
    if cfg["voc_edges_out"] == 1:
        y = y.squeeze(-1)
    loss = None
    if y_edges is not None:
        loss = loss_edges(y, y_edges, torch.as_tensor(edge_cw).to(device=y.device, dtype=y.dtype))
        scalar = loss
    else:
        scalar = (y * dy_pred).sum()
    (de,) = take([k for k in P if k.startswith("mlp_edges.")], [scalar], [eL], None)
    y, loss = y.detach(), (None if loss is None else loss.detach())
    dx = None
    del eL, e, x
    for l in reversed(range(L)):
        xi, ei = xs.pop().requires_grad_(True), es.pop().requires_grad_(True)
        xo, eo, _ = gcn_layer(xi, ei, P, f"gcn_layers.{l}.", cfg["aggregation"], relu_mask=mask_of(l))
        outs, gouts = ([eo], [de]) if dx is None else ([xo, eo], [dx, de])
        dx, de = take([k for k in P if k.startswith(f"gcn_layers.{l}.")], outs, [xi, ei], gouts)
        if dx is None:
            dx = torch.zeros_like(xi)
        del xi, ei, xo, eo
    x0, e0 = embed(P, x_edges_capacity, x_commodities, x_nodes)
    take(["nodes_embedding.weight", "commodities_embedding.weight", "edges_values_embedding.weight"], [x0, e0], [], [dx, de])
    return y, loss, grads


def beam_scores(y_pred_edges):
    """BeamSearchAlgorithm.__init__ (src/gcn/algorithms/beamsearch_uelb.py:14-21):
    logits -> log-probabilities used as additive path scores."""
    if y_pred_edges.dim() == 5:
        y = F.log_softmax(y_pred_edges, dim=4)[:, :, :, :, 1]
    else:
        y = F.log_softmax(y_pred_edges, dim=2)
    y = y.clone()
    y[y == 0] = -1e-20
    return y
