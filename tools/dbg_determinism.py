import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-convnet-uelb_amd")); sys.path.insert(0, ROOT)
import torch
from src.gcn.models.gcn_layers import ResidualGatedGCNLayer
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, V, C, H = 3, 14, 4, 64
layer = ResidualGatedGCNLayer(H, "mean", dropout_rate=0.0).to(dev).train()
x = (torch.randn(B, V, C, H) + 4.0).to(dev)
e = (torch.randn(B, V, V, C, H) + 4.0).to(dev)
gx, ge = torch.randn(B, V, C, H).to(dev), torch.randn(B, V, V, C, H).to(dev)
ref = None
for it in range(8):
    xh, eh = x.clone().requires_grad_(True), e.clone().requires_grad_(True)
    layer.zero_grad()
    x1, e1 = layer(xh, eh)
    ((x1 * gx).sum() + (e1 * ge).sum()).backward()
    torch.cuda.synchronize()
    cur = {"x1": x1.detach().clone(), "e1": e1.detach().clone(), "dx": xh.grad.clone(), "de": eh.grad.clone()}
    for k, q in layer.named_parameters():
        cur[k] = q.grad.clone()
    if ref is None:
        ref = cur
        continue
    for k in cur:
        d = (cur[k] != ref[k])
        n = int(d.sum())
        if n:
            idx = d.nonzero()
            print(it, k, "differs in", n, "of", d.numel(), "first", idx[0].tolist(), "last", idx[-1].tolist(),
                  "maxabs", float((cur[k] - ref[k]).abs().max()))
print("done")
