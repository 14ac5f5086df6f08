"""Evaluator with the reference's interface (``src/gcn/train/evaluator.py:11-226``):
``Evaluator(config, dtypeFloat, dtypeLong, strategy=None).evaluate(net, bar, mode)`` returns the 8-tuple
(time, loss, mean max load factor, mean ground-truth load factor, approximation %, infeasible %,
completion %, complete-sample %).

As in the reference, a strategy given to the constructor decides how a batch is evaluated (``evaluator.py:124-128``):
its ``compute_loss`` supplies the loss, ``mean_load_factor`` and ``pred_paths``.  The REINFORCE strategy decodes with
beam search in evaluation mode; the supervised strategy returns no paths, so with the stock ``train_gcn.py`` (which
passes ``trainer.strategy``) a supervised validation reports approximation 0 and counts every batch as infeasible,
exactly as the reference does.  Without a strategy the evaluator runs the model, the strict standard beam search and
``mean_feasible_load_factor`` itself (``:129-141``).  Forward, beam search and load factors stay on the GPU; only
per-graph scalars come back.
"""
from __future__ import annotations

import time

import numpy as np
import torch

from ...common.data_management.dataset_reader import DatasetReader
from ...common.graph.graph_utils import load_factor_from_tensors, paths_to_tensors
from ..algorithms.beamsearch_uelb import BeamsearchUELB
from ..training.base_strategy import BaseTrainingStrategy


class _Prep(BaseTrainingStrategy):
    def compute_loss(self, *a, **k): raise NotImplementedError
    def backward_step(self, *a, **k): raise NotImplementedError


def completion_and_approx_from_tensors(paths, path_len, batch_commodities, gt_load_factors, edge_capacity):
    """``_compute_completion_and_approx`` (``evaluator.py:20-79``) on decoded paths that are already on the GPU.
    The per-graph load factor of this metric is the maximum of usage / capacity over the edges with capacity > 0
    only (``:64-65``): a used zero-capacity edge does not make it infinite, unlike ``mean_feasible_load_factor``."""
    cap = edge_capacity.to(paths.device, torch.float32)
    _, complete, usage = load_factor_from_tensors(paths, path_len, cap, batch_commodities, return_usage=True)
    valid = cap > 0
    ratio = torch.where(valid, usage.to(torch.float32) / torch.where(valid, cap, torch.ones_like(cap)),
                        torch.full_like(cap, -1.0))
    model_lf = ratio.flatten(1).amax(dim=1)
    model_lf = torch.where(valid.flatten(1).any(dim=1), model_lf, torch.zeros_like(model_lf)).cpu().numpy().astype(np.float64)
    cm = complete.cpu().numpy()
    B = cm.shape[0]
    comp_rate = 100.0 * cm.mean() if cm.size else 0.0
    all_done = cm.all(axis=1)
    comp_sample_rate = 100.0 * all_done.mean() if B else 0.0
    gt = np.asarray(gt_load_factors, dtype=np.float64)
    ok = all_done & (model_lf > 1e-8) & (gt > 0)
    approx = float(np.mean(gt[ok] / model_lf[ok] * 100.0)) if ok.any() else 0.0
    return float(comp_rate), float(comp_sample_rate), approx


class Evaluator:
    def __init__(self, config, dtypeFloat, dtypeLong, strategy=None):
        self.config, self.dtypeFloat, self.dtypeLong, self.strategy = config, dtypeFloat, dtypeLong, strategy

    def _compute_completion_and_approx(self, pred_paths, batch_commodities, gt_load_factors_per_sample, edge_capacity):
        """Same arguments as the reference's method: pred_paths is a list [B][C] of node lists."""
        B, C = len(pred_paths), batch_commodities.shape[1]
        dev = edge_capacity.device if edge_capacity.is_cuda else torch.device("cuda", torch.cuda.current_device())
        paths, plen = paths_to_tensors(pred_paths, B, C, edge_capacity.shape[1], dev)
        return completion_and_approx_from_tensors(paths, plen, batch_commodities[:B], gt_load_factors_per_sample,
                                                  edge_capacity[:B])

    def evaluate(self, net, master_bar=None, mode='test'):
        net.eval()
        cfg = self.config
        batch_size = cfg.batch_size
        dataset = DatasetReader(getattr(cfg, f'num_{mode}_data'), batch_size, mode, cfg, shuffle=False)
        device = next(net.parameters()).device
        prep = _Prep(cfg)
        acc_steps = cfg.get('accumulation_steps', 1)
        edge_cw = None
        tot_loss = tot_lf = tot_gt = 0.0
        nb = nb_data = feasible = infeasible = 0
        comp, comp_sample, approx = [], [], []
        start = time.time()
        with torch.no_grad():
            for batch in dataset:
                paths_t = None
                if self.strategy is not None:                                   # evaluator.py:124-128
                    # one host-to-device copy per batch: the strategy's dict carries the two tensors used below
                    d = batch_data = self.strategy.prepare_batch_data(batch, device)
                    loss, metrics = self.strategy.compute_loss(net, batch_data, device)
                    mean_lf = float(metrics.get('mean_load_factor', 0.0))
                    pred_paths = metrics.get('pred_paths', None)
                    if pred_paths is not None:
                        paths_t = paths_to_tensors(pred_paths, len(pred_paths), d['batch_commodities'].shape[1],
                                                   d['x_edges_capacity'].shape[1], device)
                else:                                                           # evaluator.py:129-141
                    d = prep.prepare_batch_data(batch, device)
                    if edge_cw is None:
                        labels = batch.edges_target.ravel()
                        classes, counts = np.unique(labels, return_counts=True)
                        edge_cw = torch.tensor(labels.size / (len(classes) * counts.astype(np.float64)), dtype=torch.float32)
                    y_preds, loss = net.forward(d['x_edges'], d['x_commodities'], d['x_edges_capacity'], d['x_nodes'],
                                                d['y_edges'], edge_cw)
                    bs = BeamsearchUELB(y_preds, cfg.beam_size, batch_size, d['x_edges_capacity'], d['batch_commodities'],
                                        torch.float, torch.long, mode_strict=True)
                    bs.search()
                    paths_t = (bs.paths_tensor, bs.path_len)
                    max_lf, _ = load_factor_from_tensors(bs.paths_tensor, bs.path_len, d['x_edges_capacity'],
                                                         d['batch_commodities'])
                    mean_lf = float(max_lf.mean().item())
                if mean_lf > 1 or mean_lf == 0:                                 # evaluator.py:145-148
                    infeasible += 1
                else:
                    feasible += 1
                if paths_t is not None:                                         # evaluator.py:151-158
                    nbp = paths_t[0].shape[0]
                    c_rate, cs_rate, a_rate = completion_and_approx_from_tensors(
                        paths_t[0], paths_t[1], d['batch_commodities'][:nbp], np.asarray(batch.load_factor)[:nbp],
                        d['x_edges_capacity'][:nbp])
                    comp.append(c_rate)
                    comp_sample.append(cs_rate)
                    if a_rate > 0:
                        approx.append(a_rate)
                tot_loss += batch_size * float(loss.mean().item()) * acc_steps   # evaluator.py:161
                tot_lf += mean_lf
                tot_gt += float(np.mean(batch.load_factor))
                nb += 1
                nb_data += batch_size
        mean = lambda v: float(np.mean(v)) if v else 0.0
        infeasible_rate = 100.0 * infeasible / max(feasible + infeasible, 1)
        return (time.time() - start, tot_loss / max(nb_data, 1), tot_lf / max(nb, 1), tot_gt / max(nb, 1), mean(approx),
                infeasible_rate, mean(comp), mean(comp_sample))
