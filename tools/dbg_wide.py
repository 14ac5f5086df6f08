import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-convnet-uelb_amd")); sys.path.insert(0, ROOT)
import torch
from oracle import gcn_oracle
from src.common.synthetic import make_batch
from src.gcn.models.gcn_model import ResidualGatedGCNModel
class Cfg(dict):
    __getattr__ = dict.__getitem__
H, L = int(sys.argv[1]), int(sys.argv[2])
B, V, C = 2, 7, 3
batch = make_batch(B, V, C, "random", 1000, 5000, 5, 500, seed=3)
cfg = Cfg(num_nodes=V, num_commodities=C, voc_edges_in=3, voc_edges_out=2, hidden_dim=H, num_layers=L, mlp_layers=2, aggregation="mean")
torch.manual_seed(H + L)
model = ResidualGatedGCNModel(cfg, torch.float32, torch.long)
with torch.no_grad():
    for n, p in model.named_parameters():
        if "batch_norm.weight" in n: p.add_(0.2 * torch.randn_like(p))
        if "batch_norm.bias" in n: p.add_(0.1 * torch.randn_like(p))
P = {k: v.detach().clone().double().requires_grad_(True) for k, v in model.state_dict().items()}
cap = torch.from_numpy(batch["edges_capacity"]); comm = torch.from_numpy(batch["commodities"])
nodes = torch.from_numpy(batch["nodes"]); y_edges = torch.from_numpy(batch["edges_target"]); cw = batch["edge_cw"]
ocfg = dict(num_layers=L, mlp_layers=2, aggregation="mean", voc_edges_out=2)
y_ref, loss_ref = gcn_oracle.model_forward(P, ocfg, cap.double(), comm[:, :, 2].double(), nodes, y_edges, cw)
loss_ref.backward()
dev = torch.device("cuda:0")
m = model.to(dev).eval()
y, loss = m((cap > 0).long().to(dev), comm.to(dev), cap.to(dev), nodes.to(dev), y_edges.to(dev), cw)
loss.backward()
print("y %.2e loss %.2e" % (float((y.double().cpu() - y_ref).abs().max() / y_ref.abs().max()), abs(float(loss) - float(loss_ref))))
for k, p in m.named_parameters():
    r = P[k].grad if P[k].grad is not None else torch.zeros_like(P[k])
    g = p.grad.double().cpu()
    d = (g - r).abs()
    idx = [int(v) for v in torch.unravel_index(d.argmax(), d.shape)]
    print("%-44s %.2e (max ref %.2e) at %s" % (k, float(d.max()), float(r.abs().max()), idx))
