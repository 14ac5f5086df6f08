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
        t, loss, err, rl = tr.train_one_epoch()
        flat = torch.cat([p.detach().reshape(-1) for p in tr.get_model().parameters()]).cpu()
        cw = None if strategy != "supervised" else tr.strategy.edge_cw.cpu().numpy().tolist()
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
