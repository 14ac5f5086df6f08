"""Supervised strategy (reference: ``src/gcn/training/supervised_strategy.py:17-108``): class-weighted NLL
against the solver's edge labels.  The class weights are computed on the first batch of each epoch and
reused, as the reference does; the edge error comes out of the fused head kernel instead of a second pass
over the logits."""
import numpy as np
import torch

from .base_strategy import BaseTrainingStrategy


class SupervisedLearningStrategy(BaseTrainingStrategy):
    def __init__(self, config):
        super().__init__(config)
        self.edge_cw = None

    def compute_loss(self, model, batch_data, device=None):
        y_edges = batch_data['y_edges']
        if self.edge_cw is None or not isinstance(self.edge_cw, torch.Tensor):
            labels = y_edges.detach().cpu().numpy().ravel()
            classes, counts = np.unique(labels, return_counts=True)
            self.edge_cw = torch.tensor(labels.size / (len(classes) * counts.astype(np.float64)), dtype=torch.float32)
            if device:
                self.edge_cw = self.edge_cw.to(device)
        y_preds, loss = model.forward(batch_data['x_edges'], batch_data['x_commodities'], batch_data['x_edges_capacity'],
                                      batch_data['x_nodes'], y_edges, self.edge_cw)
        net = model.module if hasattr(model, "module") else model
        metrics = {'edge_error': float(net.last_edge_error.item())}
        return loss, metrics

    def backward_step(self, loss, optimizer, accumulation_steps=1, batch_num=0):
        loss = loss.mean() / accumulation_steps
        loss.backward()
        if (batch_num + 1) % accumulation_steps == 0:
            optimizer.step()
            optimizer.zero_grad()
            return True
        return False

    def reset_metrics(self):
        super().reset_metrics()
        self.edge_cw = None
