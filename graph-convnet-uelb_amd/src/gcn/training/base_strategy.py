"""Training-strategy plug point, same interface as the reference's ``BaseTrainingStrategy``
(``src/gcn/training/base_strategy.py:13-112``)."""
from abc import ABC, abstractmethod

import torch


class BaseTrainingStrategy(ABC):
    def __init__(self, config):
        self.config = config
        self.metrics = {}

    @abstractmethod
    def compute_loss(self, model, batch_data, device=None):
        """-> (loss tensor, metrics dict)"""

    @abstractmethod
    def backward_step(self, loss, optimizer, accumulation_steps=1, batch_num=0):
        """-> True if the optimizer stepped"""

    def get_metrics(self):
        return self.metrics.copy()

    def reset_metrics(self):
        self.metrics = {}

    def prepare_batch_data(self, batch, device):
        """numpy batch -> dict of device tensors (one upload per array)."""
        t = lambda a, dt: torch.as_tensor(a).to(dt).contiguous().to(device, non_blocking=True)
        batch_commodities = t(batch.commodities, torch.long)
        return {
            'x_edges': t(batch.edges, torch.long),
            'x_edges_capacity': t(batch.edges_capacity, torch.float),
            'x_nodes': t(batch.nodes, torch.long),
            'y_edges': t(batch.edges_target, torch.long),
            'batch_commodities': batch_commodities,
            'x_commodities': batch_commodities[:, :, 2].to(torch.float),
            'raw_batch': batch,
        }
