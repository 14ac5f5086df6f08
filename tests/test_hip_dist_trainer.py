"""The trainer's multi-rank path end to end with two processes that share cuda:0 (gloo with host-staged collectives;
the production backend is nccl): parameters broadcast from rank 0, every global batch split between the ranks, class
weights from the global label counts, gradients averaged, so both ranks hold identical parameters after an epoch."""
import os
import shutil
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import GOLDEN, PKG

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, root, strategy, packed, out):
    import sys
    sys.path.insert(0, PKG)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["GCN_UELB_MODEL_ROOT"] = os.path.join(root, "models")
    os.environ.pop("GCN_UELB_DATA_ROOT", None)
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        from src.common.config.config import Settings
        from src.gcn.train.trainer import Trainer
        cfg = Settings(dict(data_root=root, dataset_name="nsf_small", packed_dataset=packed, num_train_data=12, batch_size=4,
                            num_nodes=14, num_commodities=4, voc_edges_in=3, voc_edges_out=2 if strategy == "supervised" else 1,
                            hidden_dim=16, num_layers=2, mlp_layers=2, aggregation="mean", learning_rate=1e-3, max_epochs=1,
                            training_strategy=strategy, beam_size=2, seed=5))
        torch.manual_seed(100 + rank)               # different initial weights: the broadcast must make them equal
        tr = Trainer(cfg, torch.float32, torch.long)
        from src.gcn.hip import lib as L
        torch.manual_seed(7)                        # both ranks seeded alike from here on: the rank still separates
        L.reseed()                                  # their dropout and sampling streams
        t, loss, err, rl = tr.train_one_epoch()
        flat = torch.cat([p.detach().reshape(-1) for p in tr.get_model().parameters()]).cpu()
        if strategy == "supervised":
            cw = tr.strategy.edge_cw.cpu().numpy().tolist()
        else:
            cw = dict(baseline=tr.strategy.reward_baseline, seed=L.base_seed(), grad_norm=rl.get("grad_norm") if isinstance(rl, dict) else None)
        out.put((rank, "ok", float(loss), flat.numpy(), cw))
    except Exception as ex:  # pragma: no cover
        import traceback
        out.put((rank, "error: " + repr(ex) + traceback.format_exc()[-1500:], 0.0, None, None))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("strategy,packed", [("supervised", True), ("supervised", False), ("reinforcement", True)])
def test_two_ranks_end_with_identical_parameters(strategy, packed, tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    shutil.copytree(os.path.join(GOLDEN, "datasets", "nsf_small"), tmp_path / "nsf_small")
    if packed:      # written once up front so that the two ranks only read it
        import sys
        sys.path.insert(0, PKG)
        from src.common.config.config import Settings
        from src.common.data_management.packed_dataset import pack_dataset
        os.environ.pop("GCN_UELB_DATA_ROOT", None)
        pack_dataset(Settings(dict(data_root=str(tmp_path), dataset_name="nsf_small")), "train", 12)
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), strategy, packed, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=240) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    assert [r[1] for r in res] == ["ok", "ok"], res
    assert np.isfinite(res[0][2]) and np.isfinite(res[1][2])
    np.testing.assert_array_equal(res[0][3], res[1][3])          # same parameters on both ranks, bit for bit
    if strategy == "supervised":
        assert res[0][4] == res[1][4]                            # class weights from the global label counts
    else:
        # the moving reward baseline follows the GLOBAL batch mean, so the ranks agree on it exactly; their sampler
        # and dropout streams differ (rank mixed into the seed) although both were seeded alike by the config
        assert res[0][4]["baseline"] == res[1][4]["baseline"] and np.isfinite(res[0][4]["baseline"])
        assert res[0][4]["seed"] != res[1][4]["seed"]


def _grad_worker(rank, world, port, out):
    import sys
    sys.path.insert(0, PKG)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        dev = torch.device("cuda:0")
        from src.common.synthetic import make_batch
        from src.gcn.hip import dist as D
        from src.gcn.hip.train_step import HipTrainStep
        model = _grad_model().to(dev).eval()            # dropout off: the oracle can follow; BatchNorm per replica
        batch = D.shard_batch(make_batch(*_GRAD_SHAPE, "nsfnet", seed=21))
        step = HipTrainStep(model, len(batch["edges_capacity"]), dev, use_graph=False)
        step.load_batch(batch["edges_capacity"], batch["commodities"][:, :, 2].astype(np.float32), batch["nodes"],
                        batch["edges_target"], batch["edge_cw"])
        loss = float(step.step()[0].item())
        D.allreduce_mean_(step.flat_grad)
        torch.cuda.synchronize()
        out.put((rank, "ok", loss, [g.detach().cpu().numpy() for g in step.grad_views]))
    except Exception as ex:  # pragma: no cover
        import traceback
        out.put((rank, "error: " + repr(ex) + traceback.format_exc()[-1500:], 0.0, None))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


_GRAD_SHAPE = (6, 14, 4)
_GRAD_CFG = dict(num_nodes=14, num_commodities=4, voc_edges_in=3, voc_edges_out=2, hidden_dim=32, num_layers=3, mlp_layers=2,
                 aggregation="mean")


def _grad_model():
    import sys
    sys.path.insert(0, PKG)
    from src.common.config.config import Settings
    from src.gcn.models.gcn_model import ResidualGatedGCNModel
    torch.manual_seed(77)
    model = ResidualGatedGCNModel(Settings(dict(_GRAD_CFG)), torch.float32, torch.long)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "batch_norm.bias" in n:
                p.add_(0.1 * torch.randn_like(p))
    return model


def test_reduced_gradient_is_the_mean_of_the_per_shard_reference_gradients():
    """Parity at N ranks: each rank runs its shard of a global batch (per-replica BatchNorm statistics, global class
    weights), the flat gradient buffer is averaged over the ranks, and the result on every rank must equal the mean of
    the oracle's gradients computed shard by shard (float64 on the CPU), to the 1e-4 bar of the single-GPU tests.  A
    reduction that is symmetric but wrong (sum instead of mean, a rank's gradient counted twice) fails here."""
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    import sys
    sys.path.insert(0, PKG)
    from oracle import gcn_oracle
    from src.common.synthetic import make_batch
    from src.gcn.hip import dist as D
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=240) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    assert [r[1] for r in res] == ["ok"] * world, res
    model = _grad_model()
    names = [k for k, _ in model.named_parameters()]
    glob = make_batch(*_GRAD_SHAPE, "nsfnet", seed=21)
    ocfg = dict(num_layers=_GRAD_CFG["num_layers"], mlp_layers=2, aggregation="mean", voc_edges_out=2)
    mean_ref, losses = None, []
    for r in range(world):
        sh = D.shard_batch(glob, r, world)
        P = {k: v.detach().double().clone().requires_grad_(True) for k, v in model.state_dict().items()}
        t = torch.from_numpy
        _, loss = gcn_oracle.model_forward(P, ocfg, t(sh["edges_capacity"]).double(), t(sh["commodities"])[:, :, 2].double(),
                                           t(sh["nodes"]), t(sh["edges_target"]), sh["edge_cw"])
        loss.backward()
        losses.append(float(loss.detach()))
        g = [(P[k].grad if P[k].grad is not None else torch.zeros_like(P[k])).numpy() for k in names]
        mean_ref = g if mean_ref is None else [a + b for a, b in zip(mean_ref, g)]
    mean_ref = [a / world for a in mean_ref]
    gmax = max(float(np.abs(a).max()) for a in mean_ref)
    for r in range(world):
        assert abs(res[r][2] - losses[r]) <= 1e-4 * abs(losses[r])           # each rank's loss is its shard's loss
        for k, got, want in zip(names, res[r][3], mean_ref):
            err = float(np.abs(got.astype(np.float64) - want).max())
            assert err <= 1e-4 * float(np.abs(want).max()) + 1e-6 * gmax, (r, k, err)
    for a, b in zip(res[0][3], res[1][3]):
        np.testing.assert_array_equal(a, b)
