"""REINFORCE strategy on the GPU against vectors produced by the reference's ReinforcementLearningStrategy.

The reference was run on fixed edge scores (a stub model that returns them) for several consecutive batches; the
sampled paths it drew are replayed here, which makes everything else a deterministic function: load factors,
rewards, moving baseline, advantages, policy loss, entropy bonus and the gradient with respect to the scores."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import golden_files, load_golden

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    return torch.device("cuda:0")


class FixedScores(torch.nn.Module):
    def __init__(self, y):
        super().__init__()
        self.y = torch.nn.Parameter(y)

    def forward(self, x_edges, x_commodities, cap, x_nodes, y_edges=None, edge_cw=None, compute_loss=False,
                mask_invalid_edges=False):
        y = self.y
        if mask_invalid_edges:
            y = y.masked_fill((cap.unsqueeze(-1) == 0).expand_as(y), -1e9)
        return y, None


@pytest.mark.parametrize("path", golden_files("rl"), ids=os.path.basename)
def test_rl_loss_matches_reference(path):
    dev = _dev()
    from src.gcn.training.reinforcement_strategy import ReinforcementLearningStrategy
    g = load_golden(path)
    cfg = json.loads(str(g["cfg_json"]))
    strat = ReinforcementLearningStrategy(cfg)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    for k in range(int(g["cfg_calls"])):
        pre = f"call{k}."
        model = FixedScores(t(g[pre + "in_logits"])).train()
        comm = g[pre + "in_comm"]
        data = dict(x_edges=None, x_commodities=t(comm[:, :, 2].astype(np.float32)), x_edges_capacity=t(g[pre + "in_cap"]),
                    x_nodes=None, batch_commodities=t(comm))
        strat.set_epoch(k)
        strat.replay_paths = (t(g[pre + "paths"]), t(g[pre + "path_len"]))
        loss, m = strat.compute_loss(model, data)
        loss.backward()
        assert not m["reinforce_fallback_used"] and not bool(g[pre + "fallback"])
        assert abs(m["temperature"] - float(g[pre + "temperature"])) < 1e-12
        np.testing.assert_allclose(m["load_factor_individual"], g[pre + "load_factor_individual"], rtol=1e-6)
        np.testing.assert_allclose(m["reward_individual"], g[pre + "reward_individual"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(m["advantage_individual"], g[pre + "advantage_individual"], rtol=1e-4, atol=1e-5)
        for key in ("entropy", "traj_entropy", "entropy_used", "policy_loss", "entropy_bonus", "baseline_before_update",
                    "baseline_after_update", "complete_paths_rate", "finite_solution_rate", "avg_finite_load_factor",
                    "avg_path_length", "capacity_violation_rate", "reward", "reward_std", "advantage_std"):
            np.testing.assert_allclose(m[key], float(g[pre + key]), rtol=1e-4, atol=1e-5, err_msg=key)
        np.testing.assert_allclose(loss.item(), float(g[pre + "loss"]), rtol=1e-4, atol=1e-5)
        want = g[pre + "dlogits"]
        np.testing.assert_allclose(model.y.grad.cpu().numpy(), want, rtol=1e-4, atol=1e-4 * np.abs(want).max())


def test_entropy_grid_without_multi_choice_rows():
    """Rows with a single admissible next node only: the mean runs over all rows (reference :473-476)."""
    dev = _dev()
    from src.gcn.training.reinforcement_strategy import EntropyGridFn
    y = torch.full((2, 5, 5, 3), -1e9, device=dev)
    idx = torch.arange(5, device=dev)
    y[:, idx, (idx + 1) % 5, :] = 0.3
    y.requires_grad_(True)
    e = EntropyGridFn.apply(y, 1.3)
    e.backward()
    p = torch.softmax(y.detach() / 1.3, dim=2)
    want = -(p * torch.log_softmax(y.detach() / 1.3, dim=2)).sum(2).mean()
    assert abs(e.item() - want.item()) < 1e-6
    assert torch.isfinite(y.grad).all()


def test_rl_training_steps_with_the_hip_model():
    """Two REINFORCE steps through the real model (edge-score head): finite loss, finite clipped gradients,
    parameters move, evaluation mode falls back to beam search."""
    dev = _dev()
    from src.common.config.config import Settings
    from src.common.synthetic import make_batch
    from src.gcn.hip.fused_adam import FusedAdam
    from src.gcn.models.gcn_model import ResidualGatedGCNModel
    from src.gcn.training.reinforcement_strategy import ReinforcementLearningStrategy
    B, V, C = 6, 14, 5
    cfg = Settings(dict(num_nodes=V, num_commodities=C, voc_edges_in=3, voc_edges_out=1, hidden_dim=32, num_layers=2,
                        mlp_layers=2, aggregation="mean", batch_size=B, beam_size=3, rl_sampling_temperature=1.5,
                        rl_min_temperature=0.8, rl_sampling_top_p=0.95, rl_entropy_weight=0.1, rl_entropy_epsilon=0.05,
                        rl_beam_search_type="unconstrained"))
    torch.manual_seed(3)
    net = ResidualGatedGCNModel(cfg, torch.float32, torch.long).to(dev).train()
    opt = FusedAdam(net.parameters(), lr=1e-3)
    strat = ReinforcementLearningStrategy(cfg)
    from src.common.data_management.dataset_reader import DotDict
    batch = DotDict(make_batch(B, V, C, "nsfnet", seed=9))
    data = strat.prepare_batch_data(batch, dev)
    before = [p.detach().clone() for p in net.parameters()]
    for step in range(2):
        loss, m = strat.compute_loss(net, data, dev)
        assert np.isfinite(loss.item()) and not m["reinforce_fallback_used"]
        assert len(m["pred_paths"]) == B and all(len(g) == C for g in m["pred_paths"])
        assert strat.backward_step(loss, opt, 1, step)
    assert np.isfinite(strat.grad_norm_history).all()
    assert any(not torch.equal(a, b.detach()) for a, b in zip(before, net.parameters()))
    net.eval()
    with torch.no_grad():
        loss, m = strat.compute_loss(net, data, dev)
    assert m["reinforce_fallback_used"] and np.isfinite(loss.item())
