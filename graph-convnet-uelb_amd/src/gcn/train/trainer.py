"""Trainer with the reference's interface (``src/gcn/train/trainer.py:20-584``): ``Trainer(config, dtypeFloat,
dtypeLong)``, ``train_one_epoch(bar) -> (time, loss, err_edges, rl_metrics)``, ``train(evaluator, logger)``,
``get_model()``, ``.strategy``, ``save_model(epoch, loss)``, ``_get_config_hash()``.

Differences that follow from the platform: one process per GPU with the batch sharded over ranks and one
RCCL all-reduce of the flat gradient per step instead of ``nn.DataParallel``; with the supervised strategy
and ``accumulation_steps == 1`` the step runs through ``HipTrainStep`` (two library calls replayed from a
HIP graph) instead of the autograd engine.  Checkpoints keep the reference's file names and dictionary
keys; state-dict keys carry no ``module.`` prefix (``model_converter`` strips it anyway).
"""
from __future__ import annotations

import hashlib
import json
import os
import time
from pathlib import Path

import numpy as np
import torch

from ...common.data_management.dataset_reader import DatasetReader
from ..hip import dist as D
from ..hip.fused_adam import FusedAdam
from ..hip.train_step import HipTrainStep
from ..models.gcn_model import ResidualGatedGCNModel
from ..models.model_utils import update_learning_rate
from ..training.reinforcement_strategy import ReinforcementLearningStrategy
from ..training.supervised_strategy import SupervisedLearningStrategy


class Trainer:
    def __init__(self, config, dtypeFloat, dtypeLong):
        self.config = config
        self.dtypeFloat, self.dtypeLong = dtypeFloat, dtypeLong
        root = config.get('models_dir') or os.environ.get("GCN_UELB_MODEL_ROOT", "./saved_models")
        self.models_dir = str(Path(root).expanduser())
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.net = ResidualGatedGCNModel(config, dtypeFloat, dtypeLong).to(self.device)
        D.broadcast_parameters(self.net.parameters())
        # the reference builds torch.optim.Adam(lr) here (trainer.py:95); same arithmetic and state layout, one launch per step
        self.optimizer = FusedAdam(self.net.parameters(), lr=config.learning_rate)
        strategy_type = config.get('training_strategy', 'supervised')
        if strategy_type == 'supervised':
            self.strategy = SupervisedLearningStrategy(config)
        elif strategy_type == 'reinforcement':
            self.strategy = ReinforcementLearningStrategy(config)
        else:
            raise ValueError(f"Unknown training strategy: {strategy_type}")
        self._fast = None
        self._packed = None
        self._epochs_run = 0
        # steps that go through autograd (accumulation, REINFORCE) average their gradients over the ranks right
        # before the optimizer step; the graph-replay step reduces its flat gradient buffer itself
        self._sync_grads = False
        params = list(self.net.parameters())
        self.optimizer.register_step_pre_hook(
            lambda opt, args, kwargs: D.allreduce_mean_grads(params) if self._sync_grads else None)

    # ------------------------------------------------------------------ one epoch
    def train_one_epoch(self, master_bar=None):
        self.net.train()
        cfg = self.config
        batch_size, acc = cfg.batch_size, cfg.get('accumulation_steps', 1)
        world, rank = D.world(), D.rank()
        packed = bool(cfg.get('packed_dataset', False)) and not cfg.get('synthetic_data', False)
        if packed:
            dataset = self._packed_train_loader()
        else:
            # every rank reads the same global batches and takes its shard: the shuffle must be the same on all of them
            dataset = DatasetReader(cfg.get('num_train_data'), batch_size, "train", cfg, seed=int(cfg.get('seed', 0)) + self._epochs_run)
        self._epochs_run += 1
        self.strategy.reset_metrics()
        start = time.time()
        running_loss = running_err = 0.0
        running_nb = 0
        use_fast = acc == 1 and isinstance(self.strategy, SupervisedLearningStrategy)
        rl_sums, rl_n = {}, 0
        for batch_num, batch in enumerate(dataset):
            lo, hi = D.shard_bounds(batch_size, rank, world)
            if use_fast:
                if self._fast is None or self._fast.B != hi - lo:
                    self._fast = HipTrainStep(self.net, hi - lo, self.device)
                if packed:
                    # the loader already holds this rank's shard as device tensors
                    if self.strategy.edge_cw is None:
                        self.strategy.edge_cw = self._class_weights_on_device(batch['y_edges'])
                    self._fast.load_batch(batch['x_edges_capacity'], batch['x_commodities'], batch['x_nodes'],
                                          batch['y_edges'], self.strategy.edge_cw)
                else:
                    if self.strategy.edge_cw is None:       # global class weights from the first batch of the epoch
                        labels = batch.edges_target.ravel()
                        classes, counts = np.unique(labels, return_counts=True)
                        self.strategy.edge_cw = torch.tensor(labels.size / (len(classes) * counts.astype(np.float64)),
                                                             dtype=torch.float32)
                    self._fast.load_batch(batch.edges_capacity[lo:hi], batch.commodities[lo:hi, :, 2].astype(np.float32),
                                          batch.nodes[lo:hi], batch.edges_target[lo:hi], self.strategy.edge_cw)
                out = self._fast.step()
                D.allreduce_mean_(self._fast.flat_grad)
                self.optimizer.step()
                loss_val, err_val = (float(v) for v in out[[0, 2]].tolist())   # one sync per step, as the reference's .item()
            else:
                if packed:
                    data = batch
                    if isinstance(self.strategy, SupervisedLearningStrategy) and self.strategy.edge_cw is None:
                        self.strategy.edge_cw = self._class_weights_on_device(batch['y_edges'])
                else:
                    sub = type(batch)({k: (v[lo:hi] if hasattr(v, "__len__") and len(v) == batch_size else v) for k, v in batch.items()})
                    data = self.strategy.prepare_batch_data(sub, self.device)
                loss, metrics = self.strategy.compute_loss(self.net, data, self.device)
                self._sync_grads = True
                self.strategy.backward_step(loss, self.optimizer, acc, batch_num)
                self._sync_grads = False
                loss_val, err_val = float(loss.item()), metrics.get('edge_error', 0.0)
                if 'reward' in metrics:                  # epoch means of the scalar REINFORCE metrics
                    rl_n += 1
                    for k, v in metrics.items():
                        if isinstance(v, (int, float)) and not isinstance(v, bool):
                            rl_sums[k] = rl_sums.get(k, 0.0) + float(v)
            running_nb += batch_size
            running_loss += batch_size * loss_val
            running_err += batch_size * err_val
        loss = running_loss / max(running_nb, 1)
        err = running_err / max(running_nb, 1)
        rl_metrics = {k: v / rl_n for k, v in rl_sums.items()} if rl_n else {}
        if rl_n:
            rl_metrics['load_factor'] = rl_metrics.get('avg_finite_load_factor', 0.0)
            rl_metrics['baseline'] = self.strategy.reward_baseline or 0.0
        return time.time() - start, loss, err, rl_metrics

    # ------------------------------------------------------------------ packed dataset
    def _packed_train_loader(self):
        """Binary pre-packed training set (written once by rank 0), uploaded in compact form and expanded on the GPU."""
        from ...common.data_management.packed_dataset import PackedBatchLoader, pack_dataset, pack_file_for
        if self._packed is None:
            cfg = self.config
            path = pack_file_for(cfg, "train")
            if not path.exists() and D.rank() == 0:
                pack_dataset(cfg, "train", cfg.get('num_train_data'))
            if D.world() > 1:
                torch.distributed.barrier()
            self._packed = PackedBatchLoader(path, cfg.batch_size, self.device, shuffle=True, seed=cfg.get('seed', 0),
                                             num_data=cfg.get('num_train_data'), rank=D.rank(), world=D.world())
        return self._packed

    def _class_weights_on_device(self, y_edges):
        """n / (2 * count_k) over the whole global batch: label counts of the shards are summed over the ranks."""
        n1 = y_edges.sum().to(torch.float64)
        counts = torch.stack([y_edges.numel() - n1, n1])
        D.allreduce_sum_(counts)
        counts = counts.cpu()
        k = int((counts > 0).sum())
        w = torch.where(counts > 0, counts.sum() / (k * counts.clamp(min=1.0)), torch.ones(2, dtype=torch.float64))
        return w.to(torch.float32)

    # ------------------------------------------------------------------ accessors / checkpoints
    def get_model(self):
        return self.net

    def get_optimizer(self):
        return self.optimizer

    def update_learning_rate(self, new_lr):
        self.optimizer = update_learning_rate(self.optimizer, new_lr)

    def _get_config_hash(self):
        keys = ['num_nodes', 'num_commodities', 'hidden_dim', 'num_layers', 'mlp_layers', 'aggregation', 'voc_edges_out',
                'learning_rate', 'batch_size']
        blob = json.dumps({k: self.config.get(k) for k in keys}, sort_keys=True)
        return hashlib.md5(blob.encode()).hexdigest()[:8]

    def _get_model_filename(self, epoch=None):
        tag = "latest" if epoch is None else f"epoch_{epoch}"
        return f"model_{self._get_config_hash()}_{tag}.pt"

    def save_model(self, epoch, loss):
        if D.rank() != 0:
            return None
        os.makedirs(self.models_dir, exist_ok=True)
        ckpt = {'epoch': epoch, 'model_state_dict': self.net.state_dict(), 'optimizer_state_dict': self.optimizer.state_dict(),
                'loss': loss, 'config_hash': self._get_config_hash(), 'config': dict(self.config)}
        path = os.path.join(self.models_dir, self._get_model_filename())
        torch.save(ckpt, path)
        if self.config.get('save_every_epoch', False):
            torch.save(ckpt, os.path.join(self.models_dir, self._get_model_filename(epoch)))
        return path

    # ------------------------------------------------------------------ whole run
    def train(self, evaluator=None, metrics_logger=None):
        cfg = self.config
        lr = cfg.learning_rate
        history = []
        for epoch in range(cfg.max_epochs):
            if hasattr(self.strategy, 'set_epoch'):
                self.strategy.set_epoch(epoch)
            t, loss, err, rl = self.train_one_epoch()
            line = f"epoch {epoch + 1}/{cfg.max_epochs}  time {t:.2f}s  lr {lr:.2e}  loss {loss:.4f}  err_edges {err:.2f}"
            if evaluator is not None and (epoch + 1) % cfg.get('val_every', 1) == 0:
                res = evaluator.evaluate(self.net, None, mode='val')
                line += f"  val_loss {res[1]:.4f}  val_load_factor {res[2]:.4f}  infeasible% {res[5]:.1f}"
                lr /= cfg.get('decay_rate', 1.0)
                self.update_learning_rate(lr)
            if rl:
                line += (f"  reward {rl['reward']:.4f}  entropy {rl.get('entropy', 0):.4f}  complete% {rl.get('complete_paths_rate', 0):.1f}"
                         f"  T {rl.get('temperature', 0):.2f}")
            if D.rank() == 0:
                print(line, flush=True)
            history.append((epoch + 1, t, loss, err))
            if cfg.get('save_model', False):
                self.save_model(epoch + 1, loss)
        return history
