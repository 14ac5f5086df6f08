"""REINFORCE strategy (reference: ``src/gcn/training/reinforcement_strategy.py:19-785``): the model's edge
scores are a policy, one path per commodity is sampled from it, the reward is the maximum load factor of
the sampled routing, and the loss is ``mean(-log pi(paths) * advantage) - w * entropy``.

Same constructor keys (``rl_*``), ``compute_loss`` / ``backward_step`` / ``set_epoch`` /
``get_current_temperature`` and metric names as the reference.  On the GPU: model forward, path sampling with
its log-probabilities, load factors and the entropy term are library kernels and stay on the device; the
per-graph rewards (a handful of scalars) are formed on the host as in the reference.  The debug print-outs of
the reference are not reproduced.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from ..hip import dist as D
from ...common.graph.graph_utils import load_factor_from_tensors
from ..algorithms.beamsearch_uelb import BeamSearchFactory
from ..algorithms.path_sampler import PathSampler
from ..hip import lib as L
from .base_strategy import BaseTrainingStrategy


class EntropyGridFn(torch.autograd.Function):
    """Mean entropy of softmax(y / T over the next node) over the rows with at least two choices."""

    @staticmethod
    def forward(ctx, y, temperature):
        lib = L.load()
        y = L.f32c(y, "y_preds")
        B, V, _, C = y.shape
        dev = y.device
        rows = torch.empty(2 * B * V * C, dtype=torch.float32, device=dev)
        out = torch.empty(4, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            L.check(lib.gcn_entropy_grid_fwd(B, V, C, float(temperature), L.ptr(y), L.ptr(rows), L.ptr(out),
                                             L.stream_ptr(dev)), "gcn_entropy_grid_fwd")
        ctx.save_for_backward(y, rows, out)
        ctx.temperature = float(temperature)
        return out[0].clone()

    @staticmethod
    def backward(ctx, dout):
        lib = L.load()
        y, rows, out = ctx.saved_tensors
        B, V, _, C = y.shape
        dev = y.device
        dy = torch.zeros_like(y)
        d = dout.to(torch.float32).reshape(1).contiguous()
        with torch.cuda.device(dev):
            L.check(lib.gcn_entropy_grid_bwd(B, V, C, ctx.temperature, L.ptr(y), L.ptr(rows), L.ptr(out), L.ptr(d),
                                             L.ptr(dy), L.stream_ptr(dev)), "gcn_entropy_grid_bwd")
        return dy, None


def _stats(t):
    if t.numel() == 0:
        return {'mean': 0.0, 'std': 0.0, 'min': 0.0, 'max': 0.0}
    t = t.detach()
    return {'mean': t.mean().item(), 'std': t.std(unbiased=False).item() if t.numel() > 1 else 0.0,
            'min': t.min().item(), 'max': t.max().item()}


class ReinforcementLearningStrategy(BaseTrainingStrategy):
    def __init__(self, config):
        super().__init__(config)
        g = config.get
        self.beam_size = g('beam_size', 1280)
        self.batch_size = g('batch_size', 1)
        self.num_nodes = g('num_nodes', 14)
        self.num_commodities = g('num_commodities', 5)
        self.reward_type = g('rl_reward_type', 'load_factor')
        self.use_baseline = g('rl_use_baseline', True)
        self.baseline_momentum = g('rl_baseline_momentum', 0.9)
        self.entropy_weight = g('rl_entropy_weight', 0.01)
        self.beam_search_type = g('rl_beam_search_type', 'standard')
        self.use_sampling = g('rl_use_sampling', True)
        self.sampling_temperature = g('rl_sampling_temperature', 1.0)
        self.sampling_top_p = g('rl_sampling_top_p', 0.9)
        self.normalize_advantages = g('rl_normalize_advantages', True)
        self.use_smooth_penalty = g('rl_use_smooth_penalty', True)
        self.penalty_lambda = g('rl_penalty_lambda', 5.0)
        self.mask_invalid_edges = g('rl_mask_invalid_edges', True)
        self.reward_baseline = None
        self.initial_temperature = g('rl_sampling_temperature', 1.0)
        self.min_temperature = g('rl_min_temperature', 0.5)
        self.temperature_decay = g('rl_temperature_decay', 0.05)
        self.current_epoch = 0
        self.max_epochs = g('max_epochs', 10)
        self.entropy_epsilon = g('rl_entropy_epsilon', 0.05)
        self.use_trajectory_entropy = g('rl_use_trajectory_entropy', True)
        self.detect_grad_explosion = g('rl_detect_grad_explosion', True)
        self.grad_explosion_threshold = g('rl_grad_explosion_threshold', 10.0)
        self.grad_norm_history = []
        self.grad_norm_history_size = g('rl_grad_norm_history_size', 100)
        self.grad_clip_max_norm = g('rl_grad_clip_max_norm', 1.0)
        self.last_grad_explosion = False
        # test hook: (paths int32 [B,C,PL], path_len int32 [B,C]) to evaluate instead of sampling
        self.replay_paths = None

    def set_epoch(self, epoch):
        self.current_epoch = epoch

    def get_current_temperature(self):
        """Linear decay per epoch down to the minimum (reference :104-111)."""
        return max(self.min_temperature, self.initial_temperature - self.current_epoch * self.temperature_decay)

    # ------------------------------------------------------------------ rewards (reference :322-377)
    def _reward(self, load_factor, incomplete):
        if incomplete > 0:
            return -0.5 if incomplete <= 1 else (-1.0 if incomplete <= 2 else -2.0)
        if np.isinf(load_factor):
            return -1.0
        if self.reward_type != 'load_factor':
            raise ValueError(f"Unknown reward type: {self.reward_type}")
        return max(-2.0, min(2.0, 2.0 - 2.0 * load_factor))

    def compute_loss(self, model, batch_data, device=None):
        cap = batch_data['x_edges_capacity']
        comm = batch_data['batch_commodities']
        y_preds, _ = model.forward(batch_data['x_edges'], batch_data['x_commodities'], cap, batch_data['x_nodes'],
                                   y_edges=None, edge_cw=None, compute_loss=False,
                                   mask_invalid_edges=self.mask_invalid_edges)
        dev = y_preds.device
        B = y_preds.shape[0]            # the graphs of this rank's shard (self.batch_size is the global batch)
        sampling = bool(self.use_sampling and model.training)
        path_log_probs = stepwise = None
        if sampling:
            if y_preds.dim() != 4:
                raise ValueError("path sampling needs edge scores [B,V,V,C] (voc_edges_out == 1)")
            sampler = PathSampler(y_pred_edges=y_preds, edges_capacity=cap, commodities=comm, num_samples=1,
                                  temperature=self.get_current_temperature(), top_p=self.sampling_top_p,
                                  entropy_epsilon=self.entropy_epsilon)
            if self.replay_paths is not None:
                pred_paths, path_log_probs, is_feasible, stepwise = sampler.evaluate(*self.replay_paths)
            else:
                pred_paths, path_log_probs, is_feasible, stepwise = sampler.sample()
            max_lf, complete, plen = sampler.max_load_factor, sampler.complete, sampler.path_len
        else:
            bs = BeamSearchFactory.create_algorithm(self.beam_search_type, y_pred_edges=y_preds, beam_size=self.beam_size,
                                                    batch_size=B, edges_capacity=cap, commodities=comm,
                                                    dtypeFloat=torch.float, dtypeLong=torch.long, mode_strict=True)
            pred_paths, is_feasible = bs.search()
            plen = bs.path_len
            max_lf, complete = load_factor_from_tensors(bs.paths_tensor, plen, cap, comm)
        mean_lf = max_lf.mean()
        lf = max_lf.cpu().numpy().astype(np.float64)
        done = complete.cpu().numpy()
        edges_per_path = np.maximum(plen.cpu().numpy() - 1, 0)

        # path-quality metrics (reference :236-308)
        finite = [v for v in lf if not np.isinf(v) and v > 0]
        n_comm = done.size
        metrics_quality = {
            'complete_paths_rate': 100.0 * done.sum() / n_comm if n_comm else 0.0,
            'finite_solution_rate': 100.0 * len(finite) / B if B else 0.0,
            'avg_finite_load_factor': float(np.mean(finite)) if finite else 0.0,
            'avg_path_length': float(edges_per_path.mean()) if n_comm else 0.0,
            'commodity_success_rate': 100.0 * done.sum() / n_comm if n_comm else 0.0,
            'capacity_violation_rate': 100.0 * sum(v > 1.0 for v in finite) / len(finite) if finite else 0.0,
        }
        rewards = torch.tensor([self._reward(lf[i], int((~done[i]).sum())) for i in range(B)], dtype=torch.float32, device=dev)

        # baseline and advantages (reference :383-416).  Under several ranks the reward mean, and the mean and standard
        # deviation used to normalise the advantages, are those of the GLOBAL batch (sums all-reduced over the ranks), so
        # that every rank keeps the same moving baseline and N ranks do what one process does on the whole batch.
        n_glob, r_mean = float(B), rewards.mean().item()
        if D.world() > 1:
            acc = torch.stack([rewards.double().sum(), torch.tensor(float(B), dtype=torch.float64, device=dev)])
            D.allreduce_sum_(acc)
            n_glob, r_mean = acc[1].item(), (acc[0] / acc[1]).item()
        if self.reward_baseline is None:
            self.reward_baseline = r_mean
        baseline_before = self.reward_baseline
        if self.use_baseline:
            advantages = rewards - self.reward_baseline
            self.reward_baseline = (self.baseline_momentum * self.reward_baseline
                                    + (1 - self.baseline_momentum) * r_mean)
        else:
            advantages = rewards
        adv_pre = advantages.detach().clone()
        if self.normalize_advantages and n_glob > 1:
            if D.world() > 1:
                acc = torch.stack([advantages.double().sum(), (advantages.double() ** 2).sum()])
                D.allreduce_sum_(acc)
                a_mean = acc[0] / n_glob
                std = torch.sqrt(torch.clamp(acc[1] - n_glob * a_mean * a_mean, min=0.0) / (n_glob - 1)).float()
                a_mean = a_mean.float()
            else:
                std, a_mean = advantages.std(), advantages.mean()
            if not std < 1e-8:
                advantages = (advantages - a_mean) / (std + 1e-8)
        adv_post = advantages.detach().clone()

        # policy loss (reference :418-453)
        fallback = not (path_log_probs is not None and model.training and path_log_probs.requires_grad)
        if not fallback:
            policy_loss = (-path_log_probs * advantages).mean()
        else:
            # evaluation / beam-search mode: surrogate on the raw scores, as the reference does
            policy_loss = -F.log_softmax(y_preds, dim=-1).mean() * advantages.mean()

        # entropy bonus over the next-node distributions (reference :455-502)
        temp = self.get_current_temperature() if (model.training and self.use_sampling) else 1.0
        if y_preds.dim() == 4:
            entropy_grid = EntropyGridFn.apply(y_preds, temp)
        else:
            p = F.softmax(y_preds / temp, dim=2)
            row = -(p * F.log_softmax(y_preds / temp, dim=2)).sum(dim=2)
            multi = (p > 1e-8).sum(dim=2) >= 2
            entropy_grid = row[multi].mean() if bool(multi.any()) else row.mean()
        flat = [e for g_ in (stepwise or []) for c_ in g_ for e in c_]
        traj_entropy = torch.tensor(float(np.mean(flat)) if flat else 0.0, device=dev)
        entropy_bonus = -self.entropy_weight * entropy_grid
        entropy_used = traj_entropy if (self.use_trajectory_entropy and stepwise is not None and flat) else entropy_grid
        total_loss = policy_loss + entropy_bonus

        rs, ap, aq = _stats(rewards), _stats(adv_pre), _stats(adv_post)
        mean_lf_val = mean_lf.item()
        metrics = {
            'mean_load_factor': mean_lf, 'reward': rs['mean'], 'reward_std': rs['std'], 'advantage': aq['mean'],
            'advantage_std': aq['std'], 'entropy': entropy_grid.item(), 'traj_entropy': traj_entropy.item(),
            'entropy_used': entropy_used.item(), 'baseline': self.reward_baseline if self.reward_baseline else 0.0,
            'is_feasible': 1 if 0 < mean_lf_val <= 1 else 0, 'policy_loss': policy_loss.item(),
            'entropy_bonus': entropy_bonus.item(), 'baseline_before_update': baseline_before,
            'baseline_after_update': self.reward_baseline, 'advantage_pre_norm_std': ap['std'],
            'advantage_post_norm_std': aq['std'], 'advantage_pre_norm_min': ap['min'], 'advantage_pre_norm_max': ap['max'],
            'reward_min': rs['min'], 'reward_max': rs['max'],
            'normalization_applied': not torch.allclose(adv_pre, adv_post, atol=1e-8, rtol=1e-5),
            'reinforce_fallback_used': fallback, 'reward_individual': rewards.cpu().tolist(),
            'advantage_individual': advantages.detach().cpu().tolist(), 'load_factor_individual': max_lf.cpu().tolist(),
            'pred_paths': pred_paths, 'temperature': self.get_current_temperature(),
        }
        metrics.update(metrics_quality)
        self.metrics = metrics
        return total_loss, metrics

    # ------------------------------------------------------------------ backward (reference :628-777)
    # This strategy averages its gradients over the ranks itself (before clipping): the trainer must not do it again.
    syncs_gradients = True

    def backward_step(self, loss, optimizer, accumulation_steps=1, batch_num=0):
        (loss / accumulation_steps).backward()
        every = [p for group in optimizer.param_groups for p in group['params']]
        last_micro_step = (batch_num + 1) % accumulation_steps == 0
        if last_micro_step:
            # mean over the ranks first, then norm and clip: clip(mean(g_r)), what one process on the whole batch computes.
            # Over the fixed parameter list (zeros for a missing .grad), so that every rank sends the same message
            D.allreduce_mean_grads(every)
        params = [p for p in every if p.grad is not None]
        with torch.no_grad():
            total = torch.linalg.vector_norm(torch.stack([p.grad.norm() for p in params])).item() if params else 0.0
        # the history the explosion detector looks at holds averaged gradients only: under several ranks the norm of
        # a micro-step before the all-reduce is a local one and would differ from rank to rank
        if last_micro_step or not D.active() or D.world() == 1:
            self.grad_norm_history.append(total)
        if len(self.grad_norm_history) > self.grad_norm_history_size:
            self.grad_norm_history.pop(0)
        self.last_grad_explosion = False
        if self.detect_grad_explosion and len(self.grad_norm_history) >= 10:
            recent = sum(self.grad_norm_history[-10:]) / 10
            if total > self.grad_explosion_threshold * recent and total > 1.0:
                self.last_grad_explosion = True
                print(f"gradient explosion: norm {total:.4e}, recent mean {recent:.4e}", flush=True)
        if not np.isfinite(total):
            print(f"warning: non-finite gradient norm {total}", flush=True)
        if (batch_num + 1) % accumulation_steps == 0:
            if self.grad_clip_max_norm > 0 and params:
                torch.nn.utils.clip_grad_norm_(params, self.grad_clip_max_norm)
            optimizer.step()
            optimizer.zero_grad()
            return True
        return False
