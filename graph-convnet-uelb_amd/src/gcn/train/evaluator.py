"""Evaluator with the reference's interface (``src/gcn/train/evaluator.py:11-226``):
``Evaluator(config, dtypeFloat, dtypeLong, strategy=None).evaluate(net, bar, mode)`` returns the 8-tuple
(time, loss, mean max load factor, mean ground-truth load factor, approximation %, infeasible %,
completion %, complete-sample %).  Forward, beam search and load factors all stay on the GPU; only the
per-graph scalars come back."""
from __future__ import annotations

import time

import numpy as np
import torch

from ...common.data_management.dataset_reader import DatasetReader
from ...common.graph.graph_utils import load_factor_from_tensors
from ..algorithms.beamsearch_uelb import BeamsearchUELB
from ..training.base_strategy import BaseTrainingStrategy


class _Prep(BaseTrainingStrategy):
    def compute_loss(self, *a, **k): raise NotImplementedError
    def backward_step(self, *a, **k): raise NotImplementedError


class Evaluator:
    def __init__(self, config, dtypeFloat, dtypeLong, strategy=None):
        self.config, self.dtypeFloat, self.dtypeLong, self.strategy = config, dtypeFloat, dtypeLong, strategy

    def evaluate(self, net, master_bar=None, mode='test'):
        net.eval()
        cfg = self.config
        batch_size = cfg.batch_size
        dataset = DatasetReader(getattr(cfg, f'num_{mode}_data'), batch_size, mode, cfg, shuffle=False)
        device = next(net.parameters()).device
        prep = _Prep(cfg)
        edge_cw = None
        tot_loss = tot_lf = tot_gt = 0.0
        nb = nb_data = feasible = infeasible = 0
        comp, comp_sample, approx = [], [], []
        start = time.time()
        with torch.no_grad():
            for batch in dataset:
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
                max_lf, complete = load_factor_from_tensors(bs.paths_tensor, bs.path_len, d['x_edges_capacity'],
                                                            d['batch_commodities'])
                mean_lf = float(max_lf.mean().item())
                if mean_lf > 1 or mean_lf == 0:
                    infeasible += 1
                else:
                    feasible += 1
                cm = complete.cpu().numpy()
                comp.append(100.0 * cm.mean())
                all_done = cm.all(axis=1)
                comp_sample.append(100.0 * all_done.mean())
                lf = max_lf.cpu().numpy()
                gt = np.asarray(batch.load_factor, dtype=np.float64)
                ok = all_done & (lf > 1e-8) & np.isfinite(lf) & (gt > 0)
                if ok.any():
                    approx.append(float(np.mean(gt[ok] / lf[ok] * 100.0)))
                tot_loss += batch_size * float(loss.item())
                tot_lf += mean_lf
                tot_gt += float(np.mean(batch.load_factor))
                nb += 1
                nb_data += batch_size
        mean = lambda v: float(np.mean(v)) if v else 0.0
        infeasible_rate = 100.0 * infeasible / max(feasible + infeasible, 1)
        return (time.time() - start, tot_loss / max(nb_data, 1), tot_lf / max(nb, 1), tot_gt / max(nb, 1), mean(approx),
                infeasible_rate, mean(comp), mean(comp_sample))
