import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-convnet-uelb_amd")); sys.path.insert(0, ROOT)
import torch
from oracle import gcn_oracle
from src.gcn.models.gcn_layers import ResidualGatedGCNLayer
dev = torch.device("cuda:0")
def rel(a, b):
    a = a.detach().cpu(); b = b.detach().cpu()
    d = (a - b).abs()
    i = int(d.argmax())
    return float(d.max() / b.abs().max()), list(torch.unravel_index(torch.tensor(i), a.shape))
for p in (0.0, 0.3):
    torch.manual_seed(0)
    B, V, C, H = 3, 14, 4, 64
    layer = ResidualGatedGCNLayer(H, "mean", dropout_rate=p)
    x = torch.randn(B, V, C, H) + 4.0
    e = torch.randn(B, V, V, C, H) + 4.0
    gx, ge = torch.randn(B, V, C, H), torch.randn(B, V, V, C, H)
    P = {k: v.detach().clone().requires_grad_(True) for k, v in layer.state_dict().items()}
    layer = layer.to(dev).train()
    xh, eh = x.to(dev).requires_grad_(True), e.to(dev).requires_grad_(True)
    x1, e1 = layer(xh, eh)
    ((x1 * gx.to(dev)).sum() + (e1 * ge.to(dev)).sum()).backward()
    mx, me = (x1 != 0).float().cpu(), (e1 != 0).float().cpu()
    xo, eo = x.clone().requires_grad_(True), e.clone().requires_grad_(True)
    xp, ep, _ = gcn_oracle.gcn_layer(xo, eo, P, "", "mean")
    xe, ee = xp * mx / (1 - p), ep * me / (1 - p)
    ((xe * gx).sum() + (ee * ge).sum()).backward()
    print("p", p, "x", rel(x1, xe), "e", rel(e1, ee))
    print("  dx", rel(xh.grad, xo.grad), "de", rel(eh.grad, eo.grad))
    for k, q in layer.named_parameters():
        print("  ", k, rel(q.grad, P[k].grad))
