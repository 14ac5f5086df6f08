"""One reseed restarts every counter-based generator of the library: the dropout streams of the autograd path and of
live HipTrainStep objects, and the path sampler's call counter.  ``torch.manual_seed`` alone does not (the step and call
counters keep counting), which is what the docstrings say."""
import numpy as np
import pytest
import torch

from test_hip_parity import _Cfg, _dev

pytestmark = pytest.mark.gpu


def _two_train_steps(step):
    out = []
    for _ in range(2):
        o = step.step()
        torch.cuda.synchronize()
        out.append((float(o[0]), step.flat_grad.clone()))
    return out


def test_reseed_restarts_the_dropout_stream_of_a_live_train_step():
    from src.common.synthetic import make_batch
    from src.gcn.hip import lib as L
    from src.gcn.hip.train_step import HipTrainStep
    from src.gcn.models.gcn_model import ResidualGatedGCNModel
    dev = _dev()
    B, V, C = 4, 14, 5
    batch = make_batch(B, V, C, "nsfnet", seed=3)
    cfg = _Cfg(num_nodes=V, num_commodities=C, voc_edges_in=3, voc_edges_out=2, hidden_dim=32, num_layers=3, mlp_layers=2,
               aggregation="mean", dropout_rate=0.2)
    torch.manual_seed(1)
    model = ResidualGatedGCNModel(cfg, torch.float32, torch.long).to(dev).train()
    step = HipTrainStep(model, B, dev, use_graph=False)
    step.load_batch(batch["edges_capacity"], batch["commodities"][:, :, 2].astype(np.float32), batch["nodes"],
                    batch["edges_target"], batch["edge_cw"])
    torch.manual_seed(9)
    L.reseed(device=torch.device("cuda"))            # an index-less device means the current one
    a = _two_train_steps(step)
    assert a[0][0] != a[1][0]                        # new masks every step
    torch.manual_seed(9)
    c = _two_train_steps(step)                       # manual_seed alone: the step counter keeps counting
    assert c[0][0] != a[0][0]
    torch.manual_seed(9)
    L.reseed()
    b = _two_train_steps(step)
    for (la, ga), (lb, gb) in zip(a, b):
        assert la == lb and torch.equal(ga, gb)
    L.reseed(seed=1234)
    d = _two_train_steps(step)
    assert d[0][0] != a[0][0]
    L.reseed(seed=1234)
    e = _two_train_steps(step)
    assert d[0][0] == e[0][0] and torch.equal(d[1][1], e[1][1])
    L.reseed()


def test_reseed_restarts_the_path_sampler():
    from src.common.synthetic import make_batch
    from src.gcn.hip import lib as L
    from test_hip_sampler import _sampler
    dev = _dev()
    B, V, C = 3, 14, 6
    batch = make_batch(B, V, C, "nsfnet", seed=4)
    y = np.random.default_rng(0).normal(size=(B, V, V, C)).astype(np.float32)

    def draw():
        s = _sampler(y, batch["edges_capacity"], batch["commodities"], 1.5, 0.95, 0.0, dev)
        paths = s.sample()[0]
        return paths, s.last_rng

    torch.manual_seed(21)
    L.reseed()
    p0, r0 = draw()
    p1, r1 = draw()
    assert r0[1] == 0 and r1[1] == 1 and r0[0] == L.base_seed()
    torch.manual_seed(21)
    p2, r2 = draw()
    assert r2[1] == 2                                # not restarted by manual_seed alone
    L.reseed()
    q0, s0 = draw()
    q1, s1 = draw()
    assert (s0, s1) == (r0, r1) and q0 == p0 and q1 == p1
