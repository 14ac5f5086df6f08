"""One ResidualGatedGCNLayer through gcn_layer_fwd / gcn_layer_bwd (called directly through the C ABI) against the
float64 oracle on the same device, at any size: the oracle is given the ReLU pattern of the HIP run (read from the
`saved` buffer with gcn_bn_relu_pattern: pre-BatchNorm edge rows first, then the node rows, A, D, Vn, statistics last),
so BatchNorm outputs within float32 rounding of zero take the same branch in both, and every output and gradient
entry is held to the 1e-4 bound.  Where the two patterns differ the oracle's BatchNorm output must be zero at the
parity tolerance."""
import torch

from conftest import report
from oracle import gcn_oracle

RTOL = 1e-4


def _rel_err(got, ref):
    ref = ref.double()
    return float((got.double() - ref).abs().max()), float(ref.abs().max())


def check_layer_against_oracle(B, V, C, H, agg, seed, e_scale=3.0, tag=None):
    from src.gcn.hip import lib as L
    from src.gcn.models.gcn_layers import ResidualGatedGCNLayer
    dev = torch.device("cuda:0")
    lib = L.load()
    Re, Rn = B * V * V * C, B * V * C
    torch.manual_seed(seed)
    layer = ResidualGatedGCNLayer(H, agg)
    with torch.no_grad():
        for n, p in layer.named_parameters():
            if "batch_norm.weight" in n:
                p.add_(0.2 * torch.randn_like(p))
            if "batch_norm.bias" in n:
                p.add_(0.1 * torch.randn_like(p))
    layer = layer.to(dev).eval()
    x = torch.randn(B, V, C, H, device=dev)
    e = torch.randn(B, V, V, C, H, device=dev) * e_scale
    gx, ge = torch.randn(B, V, C, H, device=dev), torch.randn(B, V, V, C, H, device=dev)
    params = [p.detach().contiguous() for p in layer.hip_parameters()]
    saved = torch.empty(lib.gcn_layer_saved_floats(B, V, C, H), dtype=torch.float32, device=dev)
    ws = torch.empty(lib.gcn_layer_ws_bytes(B, V, C, H), dtype=torch.uint8, device=dev)
    x2, e2 = torch.empty_like(x), torch.empty_like(e)
    st = L.stream_ptr(dev)
    aid = 1 if agg == "mean" else 0
    L.check(lib.gcn_layer_fwd(B, V, C, H, aid, 0.0, 0, None, 0, L.ptr(x), L.ptr(e), L.ptr_array(params), L.ptr(x2), L.ptr(e2),
                              L.ptr(saved), L.ptr(ws), ws.numel(), st), "gcn_layer_fwd")
    dx, de = gx.clone(), ge.clone()
    grads = [torch.empty_like(p) for p in params]
    L.check(lib.gcn_layer_bwd(B, V, C, H, aid, 0.0, 0, None, 0, L.ptr(x), L.ptr(e), L.ptr_array(params), L.ptr(saved), L.ptr(dx),
                              L.ptr(de), L.ptr_array(grads), L.ptr(ws), ws.numel(), st), "gcn_layer_bwd")
    stats = saved[Re * H + 4 * Rn * H:]
    masks = []
    for rows, tmp, s0, bn in ((Re, saved[:Re * H], 0, layer.bn_edge.batch_norm),
                              (Rn, saved[Re * H:Re * H + Rn * H], 2, layer.bn_node.batch_norm)):
        m = torch.empty(rows * H, dtype=torch.uint8, device=dev)
        L.check(lib.gcn_bn_relu_pattern(rows, H, L.ptr(tmp), L.ptr(stats[s0 * H:]), L.ptr(stats[(s0 + 1) * H:]),
                                        L.ptr(bn.weight.detach()), L.ptr(bn.bias.detach()), L.ptr(m), st), "gcn_bn_relu_pattern")
        masks.append(m)
    me, mx = masks[0].view(B, V, V, C, H), masks[1].view(B, V, C, H)
    torch.cuda.synchronize()
    P = {k: v.detach().double().clone().requires_grad_(True) for k, v in layer.state_dict().items()}
    xo, eo = x.double().requires_grad_(True), e.double().requires_grad_(True)
    x1, e1, aux = gcn_oracle.gcn_layer(xo, eo, P, "", agg, relu_mask=(me, mx))
    ties = []
    for bn, m in ((aux["e_bn"].detach(), me), (aux["x_bn"].detach(), mx)):
        differ = (bn > 0) != m.bool()
        ties.append(int(differ.sum()))
        if ties[-1]:
            assert float(bn[differ].abs().max()) <= RTOL * float(bn.abs().max()), "ReLU pattern differs away from a tie"
    ((x1 * gx.double()).sum() + (e1 * ge.double()).sum()).backward()
    worst = []
    for got, ref, what in ((x2, x1, "x_new"), (e2, e1, "e_new"), (dx, xo.grad, "dx_in"), (de, eo.grad, "de_in")):
        errv, scale = _rel_err(got.detach(), ref.detach())
        worst.append(f"{what} {errv / scale:.2e}")
        assert errv <= RTOL * scale, f"{what}: max err {errv:.3e} vs scale {scale:.3e}"
    def grads32():
        # the same oracle in float32 (plain PyTorch, same ReLU pattern): the yardstick where float32 cannot reach RTOL
        P32 = {k: v.detach().float().clone().requires_grad_(True) for k, v in layer.state_dict().items()}
        a, b, _ = gcn_oracle.gcn_layer(x, e, P32, "", agg, relu_mask=(me, mx))
        ((a * gx).sum() + (b * ge).sum()).backward()
        return {k: P32[k].grad for k in P32}

    from test_hip_parity import assert_grads_close
    gmax = max(float(P[k].grad.abs().max()) for k in P)
    for (k, _), g in zip(layer.named_parameters(), grads):
        errv, scale = _rel_err(g, P[k].grad)
        worst.append(f"{k} {errv / max(scale, 1e-7 * gmax):.2e}")
    beyond = assert_grads_close(((k, g) for (k, _), g in zip(layer.named_parameters(), grads)), {k: P[k].grad for k in P}, grads32)
    if beyond:
        worst.append("held to the float32 oracle's own error (tensor, HIP, float32 oracle): " + ", ".join(f"{k} {a:.2e} {b:.2e}" for k, a, b in beyond))
    if tag:
        report(f"[{tag}] B={B} V={V} C={C} H={H} {agg}: ReLU decisions differing from the float64 oracle's, edge/node: "
               f"{ties}; errors: " + ", ".join(worst))
