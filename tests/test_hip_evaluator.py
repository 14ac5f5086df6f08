"""Evaluation metrics behind the decode (completion, approximation, load factor) against their CPU restatement
(oracle/eval_oracle.py), and the evaluator's two branches: with a strategy (what train_gcn.py passes) and without.

Parity status of the approximation / completion rates: UNPINNED.  The reference's evaluator.py cannot be imported in
the build container (it imports fastprogress at module level) and the reference holds no fixture for these numbers;
the oracle is a line-by-line restatement of evaluator.py:20-79.  The load factor itself is pinned through the
reference's REINFORCE golden vectors (tests/test_hip_rl.py)."""
import os
import shutil

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import eval_oracle

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    return torch.device("cuda:0")


def test_completion_and_approximation_match_the_oracle():
    """Paths that cover every branch of evaluator.py:42-77: complete graphs, the beam search's fallback [0..9] (node
    ids beyond a 7-node graph are skipped by the usage sum), a complete graph with a hop over a zero-capacity edge
    (infinite for mean_feasible_load_factor, finite and counted for the approximation rate), an empty path, and a
    complete graph whose paths carry no demand over positive-capacity edges (load factor 0: left out)."""
    dev = _dev()
    from src.common.config.config import Settings
    from src.common.graph.graph_utils import mean_feasible_load_factor
    from src.gcn.train.evaluator import Evaluator
    rng = np.random.default_rng(3)
    B, V, C = 6, 7, 3
    cap = rng.integers(100, 900, size=(B, V, V)).astype(np.float32)
    for b in range(B):
        np.fill_diagonal(cap[b], 0.0)
    cap[2, 1, 4] = 0.0                       # graph 2 routes over this edge
    cap[5] = 0.0
    cap[5, 6, 0] = 50.0                      # graph 5: the only edge with capacity is unused
    comm = np.zeros((B, C, 3), dtype=np.int64)
    comm[:, :, 2] = rng.integers(5, 200, size=(B, C))
    paths = [[[] for _ in range(C)] for _ in range(B)]
    ends = [(0, 3), (2, 5), (6, 1)]
    for b in range(B):
        for c, (s, d) in enumerate(ends):
            comm[b, c, 0], comm[b, c, 1] = s, d
    paths[0] = [[0, 1, 3], [2, 4, 5], [6, 0, 1]]                     # all complete
    paths[1] = [list(range(10)) for _ in range(C)]                    # fallback: none ends on its destination
    paths[2] = [[0, 1, 4, 3], [2, 1, 4, 5], [6, 1]]                   # complete, two commodities over the zero-capacity edge
    paths[3] = [[0, 3], [], [6, 5, 1]]                                # one commodity without a path
    paths[4] = [[0, 2, 3], [2, 3, 5], [6, 2, 1]]                      # complete
    paths[5] = [[0, 3], [2, 5], [6, 1]]                               # complete, but only zero-capacity edges are used
    gt = np.array([0.4, 0.3, 0.2, 0.5, 0.0, 0.1])                     # graph 4: ground truth 0 -> left out
    capt, commt = torch.from_numpy(cap).to(dev), torch.from_numpy(comm).to(dev)
    ev = Evaluator(Settings(dict(batch_size=B)), torch.float32, torch.long)
    got = ev._compute_completion_and_approx(paths, commt, gt, capt)
    want = eval_oracle.completion_and_approx(paths, comm, gt, cap)
    assert got == pytest.approx(want, rel=1e-6, abs=1e-9), (got, want)
    assert want[0] == pytest.approx(100.0 * 14 / 18) and want[1] == pytest.approx(100.0 * 4 / 6)
    # the approximation rate averages graphs 0 and 2 only (1 incomplete, 3 incomplete, 4 has gt 0, 5 has load factor 0)
    assert want[2] > 0
    # mean_feasible_load_factor of the same paths: +inf for graph 2 (demand on a zero-capacity edge) and graph 5
    mean_lf, per_graph = mean_feasible_load_factor(B, C, V, [[[n for n in p if n < V] for p in g] for g in paths], capt, commt)
    ref = [eval_oracle.max_load_factor([[n for n in p if n < V] for p in paths[b]], comm[b], cap[b]) for b in range(B)]
    np.testing.assert_allclose(per_graph.cpu().numpy(), np.array(ref, dtype=np.float32), rtol=1e-6)
    assert np.isinf(ref[2]) and np.isinf(ref[5]) and np.isfinite(ref[0])


def _cfg(tmp_path, strategy, **kw):
    from src.common.config.config import Settings
    base = dict(data_root=str(tmp_path), dataset_name="nsf_small", num_train_data=12, num_val_data=8, num_test_data=8, batch_size=4,
                num_nodes=14, num_commodities=4, voc_edges_in=3, voc_edges_out=2 if strategy == "supervised" else 1,
                hidden_dim=16, num_layers=2, mlp_layers=2, aggregation="mean", learning_rate=1e-3, max_epochs=1, val_every=1,
                training_strategy=strategy, beam_size=2, rl_entropy_epsilon=0.05, rl_beam_search_type="standard")
    base.update(kw)
    return Settings(base)


def _dataset(tmp_path):
    shutil.copytree(os.path.join(GOLDEN, "datasets", "nsf_small"), tmp_path / "nsf_small")
    for mode in ("val_data", "test_data"):
        shutil.copytree(tmp_path / "nsf_small" / "train_data", tmp_path / "nsf_small" / mode)


@pytest.mark.parametrize("strategy", ["reinforcement", "supervised"])
def test_training_with_validation_through_the_strategy(strategy, tmp_path, monkeypatch):
    """One training epoch followed by a validation epoch with Evaluator(strategy=trainer.strategy), as train_gcn.py
    builds it.  REINFORCE (edge-score head, voc_edges_out = 1): the strategy decodes with beam search in evaluation
    mode and the evaluator reports completion from its paths.  Supervised: the strategy returns no paths, so
    approximation and completion stay 0 and every batch counts as infeasible - the reference's behaviour."""
    dev = _dev()
    from src.gcn.train.evaluator import Evaluator
    from src.gcn.train.trainer import Trainer
    _dataset(tmp_path)
    monkeypatch.setenv("GCN_UELB_MODEL_ROOT", str(tmp_path / "models"))
    monkeypatch.delenv("GCN_UELB_DATA_ROOT", raising=False)
    cfg = _cfg(tmp_path, strategy)
    torch.manual_seed(0)
    tr = Trainer(cfg, torch.float32, torch.long)
    ev = Evaluator(cfg, torch.float32, torch.long, strategy=tr.strategy)
    tr.train(ev, None)
    res = ev.evaluate(tr.get_model(), None, mode='val')
    assert len(res) == 8 and np.isfinite(res[1])
    if strategy == "reinforcement":
        assert 0.0 <= res[6] <= 100.0 and 0.0 <= res[7] <= 100.0 and res[4] >= 0.0
    else:
        assert res[4] == 0.0 and res[6] == 0.0 and res[7] == 0.0 and res[5] == 100.0 and res[2] == 0.0


def test_evaluator_without_strategy_decodes_with_beam_search(tmp_path, monkeypatch):
    """strategy=None: model forward, strict standard beam search and load factors inside the evaluator; the metrics
    equal the oracle's on the paths the search returned."""
    dev = _dev()
    from src.gcn.models.gcn_model import ResidualGatedGCNModel
    from src.gcn.train.evaluator import Evaluator
    _dataset(tmp_path)
    monkeypatch.delenv("GCN_UELB_DATA_ROOT", raising=False)
    cfg = _cfg(tmp_path, "supervised")
    torch.manual_seed(1)
    net = ResidualGatedGCNModel(cfg, torch.float32, torch.long).to(dev)
    res = Evaluator(cfg, torch.float32, torch.long, strategy=None).evaluate(net, None, mode='val')
    assert len(res) == 8 and np.isfinite(res[1])
    assert 0.0 <= res[6] <= 100.0 and 0.0 <= res[7] <= 100.0 and 0.0 <= res[5] <= 100.0
    assert res[3] > 0.0                                   # ground-truth load factors of the dataset


def test_trainer_schedule_checkpoints_and_pretrained_head_conversion(tmp_path, monkeypatch):
    """The reference's epoch schedule (validation at epoch % val_every == 0 and on the last epoch, learning-rate decay at
    epoch % val_every == 0 except epoch 0, evaluator or not), its checkpoint file names (md5 over its hash keys),
    keep-last-5 cleanup, and loading a supervised checkpoint into the REINFORCE model through the logit difference."""
    dev = _dev()
    import hashlib
    import json
    from src.gcn.train.trainer import Trainer
    _dataset(tmp_path)
    monkeypatch.setenv("GCN_UELB_MODEL_ROOT", str(tmp_path / "models"))
    monkeypatch.delenv("GCN_UELB_DATA_ROOT", raising=False)
    cfg = _cfg(tmp_path, "supervised", max_epochs=3, val_every=2, decay_rate=2.0, save_every_epoch=True, expt_name="t")
    torch.manual_seed(0)
    tr = Trainer(cfg, torch.float32, torch.long)
    calls = []

    class Ev:
        def evaluate(self, net, bar, mode):
            calls.append((len(tr.history), mode))
            return (0.0, 1.0, 0.0, 0.0, 0.0, 100.0, 0.0, 0.0)

    net = tr.train(Ev(), None)
    assert net is tr.get_model()
    assert calls == [(0, 'val'), (0, 'test'), (2, 'val'), (2, 'test')]          # epochs 0 and 2 (= last); test_every defaults to the run length
    assert tr.optimizer.param_groups[0]['lr'] == pytest.approx(1e-3 / 2.0)     # one decay, at epoch 2
    keys = ['hidden_dim', 'num_layers', 'mlp_layers', 'voc_edges_in', 'voc_edges_out', 'aggregation', 'beam_size', 'num_commodities',
            'num_nodes', 'learning_rate', 'batch_size', 'max_epochs', 'decay_rate', 'training_strategy', 'expt_name']
    want = hashlib.md5(json.dumps({k: cfg[k] for k in keys}, sort_keys=True).encode()).hexdigest()[:8]
    assert tr._get_config_hash() == want
    names = sorted(os.listdir(tmp_path / "models"))
    assert names == [f"model_{want}_epoch_{e}.pt" for e in range(3)] + [f"model_{want}_latest.pt"]
    # a supervised checkpoint with DataParallel key names -> REINFORCE model with a one-score head
    sup = {("module." + k): v.detach().cpu().clone() for k, v in tr.get_model().state_dict().items()}
    path = tmp_path / "sup.pt"
    torch.save({'epoch': 7, 'model_state_dict': sup}, path)
    cfg_rl = _cfg(tmp_path, "reinforcement", load_pretrained_model=True, pretrained_model_path=str(path), convert_supervised_to_rl=True)
    rl = Trainer(cfg_rl, torch.float32, torch.long)
    w2, b2 = sup["module.mlp_edges.output_layer.weight"], sup["module.mlp_edges.output_layer.bias"]
    sd = rl.get_model().state_dict()
    assert torch.equal(sd["mlp_edges.output_layer.weight"].cpu(), w2[1:2] - w2[0:1])
    assert torch.equal(sd["mlp_edges.output_layer.bias"].cpu(), b2[1:2] - b2[0:1])
    assert torch.equal(sd["gcn_layers.0.edge_feat.U.weight"].cpu(), sup["module.gcn_layers.0.edge_feat.U.weight"])
    # checkpoint_module_prefix: the file carries the reference's key names (trainer.py:382 saves an nn.DataParallel
    # wrapper), loads strictly into such a wrapper, and reads back here
    cfg_p = _cfg(tmp_path, "supervised", max_epochs=3, val_every=2, decay_rate=2.0, expt_name="t", checkpoint_module_prefix=True)
    tr_p = Trainer(cfg_p, torch.float32, torch.long)
    saved = tr_p.save_model(0, 1.0)
    ck = torch.load(saved, map_location="cpu", weights_only=True)
    assert all(k.startswith("module.") for k in ck["model_state_dict"])
    from src.gcn.models.gcn_model import ResidualGatedGCNModel
    wrapper = torch.nn.DataParallel(ResidualGatedGCNModel(cfg_p, torch.float32, torch.long))
    wrapper.load_state_dict(ck["model_state_dict"], strict=True)
    other = Trainer(cfg_p, torch.float32, torch.long)
    with torch.no_grad():
        for q in other.get_model().parameters():
            q.add_(1.0)
    assert other.load_checkpoint(saved) == 0
    for (k, a), b in zip(tr_p.get_model().state_dict().items(), other.get_model().state_dict().values()):
        assert torch.equal(a, b), k
