import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-convnet-uelb_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import gcn_oracle
from src.common.synthetic import CONFIGS, make_batch
from src.gcn.models.gcn_model import ResidualGatedGCNModel
class Cfg(dict):
    __getattr__ = dict.__getitem__
name = sys.argv[1] if len(sys.argv) > 1 else "nsfnet_10_commodities"
dbl = len(sys.argv) > 2
s = CONFIGS[name]
B, V, C, H, L = s["B"], s["V"], s["C"], s["H"], s["L"]
batch = make_batch(B, V, C, s["topology"], s["cap_lo"], s["cap_hi"], s["dem_lo"], s["dem_hi"], seed=99)
cfg = Cfg(num_nodes=V, num_commodities=C, voc_edges_in=3, voc_edges_out=2, hidden_dim=H, num_layers=L, mlp_layers=2, aggregation="mean")
torch.manual_seed(5)
model = ResidualGatedGCNModel(cfg, torch.float32, torch.long)
cap = torch.from_numpy(batch["edges_capacity"]); comm = torch.from_numpy(batch["commodities"])
nodes = torch.from_numpy(batch["nodes"]); y_edges = torch.from_numpy(batch["edges_target"]); cw = batch["edge_cw"]
ocfg = dict(num_layers=L, mlp_layers=2, aggregation="mean", voc_edges_out=2)
def run_oracle(dt):
    P = {k: v.detach().clone().to(dt).requires_grad_(True) for k, v in model.state_dict().items()}
    y, loss = gcn_oracle.model_forward(P, ocfg, cap.to(dt), comm[:, :, 2].to(dt), nodes, y_edges, torch.as_tensor(cw).to(dt))
    loss.backward()
    return {k: (P[k].grad if P[k].grad is not None else torch.zeros_like(P[k])).double() for k in P}, y.detach().double()
try:
    g64, y64 = run_oracle(torch.float64)
except Exception as ex:
    print("fp64 oracle failed:", ex); g64 = None
g32, y32 = run_oracle(torch.float32)
dev = torch.device("cuda:0")
m = model.to(dev).eval()
y, loss = m((cap > 0).long().to(dev), comm.to(dev), cap.to(dev), nodes.to(dev), y_edges.to(dev), cw)
loss.backward()
print("y: hip vs o32 %.2e" % float((y.double().cpu() - y32).abs().max() / y32.abs().max()), ("hip vs o64 %.2e  o32 vs o64 %.2e" % (float((y.double().cpu() - y64).abs().max() / y64.abs().max()), float((y32 - y64).abs().max() / y64.abs().max()))) if g64 else "")
for k, p in m.named_parameters():
    g = p.grad.double().cpu()
    sc = float(g32[k].abs().max()) + 1e-30
    line = "%-44s hip-o32 %.2e" % (k, float((g - g32[k]).abs().max()) / sc)
    if g64:
        sc64 = float(g64[k].abs().max()) + 1e-30
        line += "  hip-o64 %.2e  o32-o64 %.2e" % (float((g - g64[k]).abs().max()) / sc64, float((g32[k] - g64[k]).abs().max()) / sc64)
    print(line)
