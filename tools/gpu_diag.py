#!/usr/bin/env python3
"""Diagnostic run on a GPU box: HIP path vs the CPU oracle, every tensor, no early exit.
Usage: python tools/gpu_diag.py [layer] [model]  (default both).  Writes a report to stdout."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "graph-convnet-uelb_amd"))

from oracle import gcn_oracle  # noqa: E402
from src.gcn.models.gcn_layers import ResidualGatedGCNLayer  # noqa: E402
from src.gcn.models.gcn_model import ResidualGatedGCNModel  # noqa: E402

DEV = torch.device("cuda:0")


def rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def report(tag, a, b, tol=1e-4):
    r = rel(a, b)
    print(f"  {'OK ' if r <= tol else 'BAD'} {tag:<44s} rel={r:.3e}  |ref|max={float(b.abs().max()):.3e}")
    return r <= tol


def layer_case(B, V, C, H, agg, seed):
    print(f"layer B={B} V={V} C={C} H={H} agg={agg}")
    torch.manual_seed(seed)
    layer = ResidualGatedGCNLayer(H, agg)
    with torch.no_grad():
        for n, p in layer.named_parameters():
            if "batch_norm.weight" in n:
                p.add_(0.2 * torch.randn_like(p))
            if "batch_norm.bias" in n:
                p.add_(0.1 * torch.randn_like(p))
    layer.eval()
    x = torch.randn(B, V, C, H)
    e = torch.randn(B, V, V, C, H) * 3.0
    gx = torch.randn(B, V, C, H)
    ge = torch.randn(B, V, V, C, H)
    # oracle
    P = {k: v.detach().clone().requires_grad_(True) for k, v in layer.state_dict().items()}
    xo = x.clone().requires_grad_(True)
    eo = e.clone().requires_grad_(True)
    x1, e1, aux = gcn_oracle.gcn_layer(xo, eo, P, "", agg)
    ((x1 * gx).sum() + (e1 * ge).sum()).backward()
    # HIP
    layer = layer.to(DEV)
    xh = x.to(DEV).requires_grad_(True)
    eh = e.to(DEV).requires_grad_(True)
    x2, e2 = layer(xh, eh)
    ((x2 * gx.to(DEV)).sum() + (e2 * ge.to(DEV)).sum()).backward()
    torch.cuda.synchronize()
    ok = report("x_new", x2, x1)
    ok &= report("e_new", e2, e1)
    ok &= report("dx_in", xh.grad, xo.grad)
    ok &= report("de_in", eh.grad, eo.grad)
    for k, p in layer.named_parameters():
        ok &= report("grad " + k, p.grad, P[k].grad)
    return ok


def model_case(path):
    g = dict(np.load(path))
    B, V, C, H, Lr = (int(g[k]) for k in ("cfg_B", "cfg_V", "cfg_C", "cfg_H", "cfg_L"))
    voc, mask = int(g["cfg_voc"]), bool(g["cfg_mask"])
    print(f"model {os.path.basename(path)} B={B} V={V} C={C} H={H} L={Lr} voc={voc} mlp={int(g['cfg_mlp_layers'])}")

    class Cfg(dict):
        __getattr__ = dict.__getitem__
    cfg = Cfg(num_nodes=V, num_commodities=C, voc_edges_in=3, voc_edges_out=voc, hidden_dim=H, num_layers=Lr,
              mlp_layers=int(g["cfg_mlp_layers"]), aggregation=str(g["cfg_aggregation"]))
    model = ResidualGatedGCNModel(cfg, torch.float32, torch.long)
    sd = {k[len("param."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param.")}
    model.load_state_dict(sd)
    model = model.to(DEV).eval()
    t = lambda a: torch.from_numpy(a).to(DEV)
    cap, comm, x_nodes = t(g["in_cap"]), t(g["in_comm"]), t(g["in_x_nodes"])
    x_edges = (cap > 0).long()
    ok = True
    if voc == 2:
        y, loss = model(x_edges, comm[:, :, 2].float(), cap, x_nodes, t(g["in_y_edges"]), g["in_edge_cw"])
        loss.backward()
        torch.cuda.synchronize()
        ok &= report("loss", loss, torch.tensor(float(g["out_loss"])))
        ok &= report("edge_error", model.last_edge_error, torch.tensor(float(g["out_edge_error"])), tol=1e-5)
        same = bool((model.last_err_idx.cpu().numpy() == g["out_err_idx"]).all())
        print(f"  {'OK ' if same else 'BAD'} err_idx flags")
        ok &= same
    else:
        y, loss = model(x_edges, comm, cap, x_nodes, compute_loss=False, mask_invalid_edges=mask)
        (y * t(g["in_dy_pred"])).sum().backward()
        torch.cuda.synchronize()
    ok &= report("y_pred", y, torch.from_numpy(g["out_y_pred"]))
    for k, p in model.named_parameters():
        ok &= report("grad " + k, p.grad, torch.from_numpy(g["grad." + k]))
    return ok


def main():
    what = sys.argv[1:] or ["layer", "model"]
    t0 = time.time()
    allok = True
    if "layer" in what:
        for (B, V, C, H, agg, seed) in [(2, 5, 3, 8, "mean", 1), (2, 6, 4, 12, "mean", 2), (3, 14, 10, 64, "mean", 3),
                                        (2, 20, 3, 32, "mean", 4), (2, 5, 3, 16, "sum", 5), (1, 40, 2, 64, "mean", 6),
                                        (2, 7, 2, 48, "mean", 7)]:
            try:
                allok &= layer_case(B, V, C, H, agg, seed)
            except Exception as ex:  # keep going: one report for all cases
                allok = False
                print("  EXC", repr(ex))
    if "model" in what:
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "model_*.npz"))):
            try:
                allok &= model_case(path)
            except Exception as ex:
                allok = False
                print("  EXC", repr(ex))
    print(f"ALL OK: {allok}  ({time.time() - t0:.1f}s)")
    return 0 if allok else 1


if __name__ == "__main__":
    sys.exit(main())
