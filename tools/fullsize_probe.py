#!/usr/bin/env python3
"""Per-tensor gradient error of the HIP path at a full-size configuration, next to the error of the same oracle run in
float32 (plain PyTorch on the GPU), both against the float64 oracle with the HIP run's ReLU pattern (measurement aid for the
bounds in tests/test_hip_fullsize.py).  Usage: python tools/fullsize_probe.py <config> [B]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-convnet-uelb_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
from oracle import gcn_oracle  # noqa: E402
from test_hip_fullsize import _HipRun  # noqa: E402
from test_hip_parity import _Cfg  # noqa: E402
from src.common.synthetic import CONFIGS, make_batch  # noqa: E402
from src.gcn.models.gcn_model import ResidualGatedGCNModel  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "synthetic_100n_50c_h300_l30"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
s = CONFIGS[name]
V, C, H, Lr = s["V"], s["C"], s["H"], s["L"]
batch = make_batch(B, V, C, s["topology"], s["cap_lo"], s["cap_hi"], s["dem_lo"], s["dem_hi"], seed=99)
cfg = _Cfg(num_nodes=V, num_commodities=C, voc_edges_in=3, voc_edges_out=2, hidden_dim=H, num_layers=Lr, mlp_layers=2, aggregation="mean")
torch.manual_seed(5)
model = ResidualGatedGCNModel(cfg, torch.float32, torch.long).to(dev).eval()
hip = _HipRun(model, batch, dev, B)
hip.run()
t = lambda a: torch.from_numpy(a).to(dev)
cap, comm, nodes, y_edges = t(batch["edges_capacity"]), t(batch["commodities"]), t(batch["nodes"]), t(batch["edges_target"])
ocfg = dict(num_layers=Lr, mlp_layers=2, aggregation="mean", voc_edges_out=2)
res = {}
for dt in (torch.float64, torch.float32):
    P = {k: v.detach().to(dt).clone().requires_grad_(True) for k, v in model.state_dict().items()}
    y, loss, g = gcn_oracle.model_loss_and_grads_checkpointed(P, ocfg, cap.to(dt), comm[:, :, 2].to(dt), nodes, y_edges, batch["edge_cw"],
                                                              relu_masks=hip.relu_masks)
    res[dt] = (y.double(), float(loss), {k: v.double() for k, v in g.items()})
    del P, g
    torch.cuda.empty_cache()
y64, l64, g64 = res[torch.float64]
y32, l32, g32 = res[torch.float32]
print(f"{name} B={B}: loss hip {float(hip.step.loss_out[0]):.7f} oracle64 {l64:.7f} oracle32 {l32:.7f}")
ys = float(y64.abs().max())
print(f"y_pred: hip err {float((hip.step.y_pred.double() - y64).abs().max()) / ys:.2e}, float32 oracle err {float((y32 - y64).abs().max()) / ys:.2e}")
gmax = max(float(v.abs().max()) for v in g64.values())
rows = []
for (k, p), gv in zip(model.named_parameters(), hip.step.grad_views):
    ref = g64[k]
    sc = float(ref.abs().max())
    eh = float((gv.double() - ref).abs().max())
    e32 = float((g32[k] - ref).abs().max())
    rows.append((eh / max(sc, 1e-300), k, eh, e32, sc))
zero = [r for r in rows if r[4] < 1e-9 * gmax]
rows = [r for r in rows if r[4] >= 1e-9 * gmax]
print(f"{len(zero)} tensors are zero in exact arithmetic (own max below 1e-9 of the largest gradient); their largest |hip| error "
      f"{max(r[2] for r in zero) / gmax:.2e} and float32-oracle error {max(r[3] for r in zero) / gmax:.2e} of the largest gradient")
rows.sort(reverse=True)
print(f"largest gradient of the model {gmax:.3e}; tensors by hip error / own max (float32 oracle's error beside it):")
for r, k, eh, e32, sc in rows[:25]:
    print(f"  {k:50s} hip {eh / max(sc, 1e-300):9.2e}  f32-oracle {e32 / max(sc, 1e-300):9.2e}  scale {sc:.3e} ({sc / gmax:.1e} of the largest)")
