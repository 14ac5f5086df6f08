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
    d[torch.isnan(d)] = float("inf")
    i = int(d.argmax())
    return "%.2e@%s" % (float(d.max() / b.abs().max()), [int(v) for v in torch.unravel_index(torch.tensor(i), a.shape)])
torch.manual_seed(0)
B, V, C, H = 3, 14, 4, 64
layer = ResidualGatedGCNLayer(H, "mean", dropout_rate=0.0)
x = torch.randn(B, V, C, H) + 4.0
e = torch.randn(B, V, V, C, H) + 4.0
gx, ge = torch.randn(B, V, C, H), torch.randn(B, V, V, C, H)
P = {k: v.detach().clone().requires_grad_(True) for k, v in layer.state_dict().items()}
xo, eo = x.clone().requires_grad_(True), e.clone().requires_grad_(True)
xp, ep, _ = gcn_oracle.gcn_layer(xo, eo, P, "", "mean")
((xp * gx).sum() + (ep * ge).sum()).backward()
layer = layer.to(dev).train()
for it in range(4):
    for val in (float("nan"), 1e30):
        junk = torch.full((256 << 20,), val, device=dev); del junk     # 1 GiB of poison for torch.empty to hand back
        xh, eh = x.to(dev).requires_grad_(True), e.to(dev).requires_grad_(True)
        layer.zero_grad()
        x1, e1 = layer(xh, eh)
        ((x1 * gx.to(dev)).sum() + (e1 * ge.to(dev)).sum()).backward()
        print(it, val, "x", rel(x1, xp), "e", rel(e1, ep), "dx", rel(xh.grad, xo.grad), "de", rel(eh.grad, eo.grad))
        print("     ", " ".join("%s %s" % (k.replace("batch_norm.", ""), rel(q.grad, P[k].grad)) for k, q in layer.named_parameters() if "weight" in k or "bn" in k))
