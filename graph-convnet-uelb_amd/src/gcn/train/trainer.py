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
        # pretrained weights, optionally a supervised two-class head converted to the REINFORCE score head
        # (reference trainer.py:68-92; its load_saved_model branch is commented out there, :47-52, and has no effect)
        if config.get('load_pretrained_model', False) and config.get('pretrained_model_path'):
            self.load_checkpoint(str(Path(config.get('pretrained_model_path')).expanduser()),
                                 convert_supervised_to_rl=bool(config.get('convert_supervised_to_rl', False)))
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
                self._sync_grads = not getattr(self.strategy, 'syncs_gradients', False)
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
        """md5 over the same configuration keys as the reference (trainer.py:271-309), so that checkpoint file names
        agree for the same configuration."""
        hash_keys = ['hidden_dim', 'num_layers', 'mlp_layers', 'node_dim', 'voc_nodes_out', 'voc_edges_in', 'voc_edges_out',
                     'aggregation', 'dropout_rate', 'beam_size',
                     'num_commodities', 'capacity_lower', 'capacity_higher', 'demand_lower', 'demand_higher', 'num_nodes',
                     'sample_size',
                     'learning_rate', 'batch_size', 'max_epochs', 'decay_rate',
                     'training_strategy', 'reward_function_version',
                     'solver_type', 'graph_model', 'expt_name']
        cfg = dict(self.config)
        picked = {k: cfg[k] for k in hash_keys
                  if k in cfg and (isinstance(cfg[k], (int, float, str, bool)) or cfg[k] is None)}
        return hashlib.md5(json.dumps(picked, sort_keys=True).encode()).hexdigest()[:8]

    def _get_model_filename(self, epoch=None):
        tag = "latest" if epoch is None else f"epoch_{epoch}"
        return f"model_{self._get_config_hash()}_{tag}.pt"

    def save_model(self, epoch, loss, save_latest=True):
        """Checkpoint with the reference's file names and top-level keys (trainer.py:375-412); rank 0 writes.
        The reference saves the state dict of an nn.DataParallel wrapper, whose parameter names start with ``module.``
        (trainer.py:382), and loads it back into the wrapper with strict key matching.  The model here is not wrapped,
        so its names carry no prefix; with ``checkpoint_module_prefix: true`` in the config the prefix is written, and
        the file then loads in the reference as it is.  load_checkpoint accepts both forms."""
        if D.rank() != 0:
            return None
        os.makedirs(self.models_dir, exist_ok=True)
        cfg = self.config
        state = self.net.state_dict()
        if cfg.get('checkpoint_module_prefix', False):
            state = {'module.' + k: v for k, v in state.items()}
        ckpt = {'epoch': epoch, 'model_state_dict': state, 'optimizer_state_dict': self.optimizer.state_dict(),
                'loss': loss, 'config_hash': self._get_config_hash(),
                'config': {'hidden_dim': cfg.hidden_dim, 'num_layers': cfg.num_layers, 'mlp_layers': cfg.mlp_layers,
                           'node_dim': cfg.get('node_dim'), 'voc_nodes_in': cfg.num_commodities * 3,
                           'voc_nodes_out': cfg.get('voc_nodes_out'), 'voc_edges_in': cfg.voc_edges_in,
                           'voc_edges_out': cfg.voc_edges_out, 'aggregation': cfg.aggregation,
                           'dropout_rate': cfg.get('dropout_rate', 0.0)}}
        path = None
        if cfg.get('save_every_epoch', False):
            torch.save(ckpt, os.path.join(self.models_dir, self._get_model_filename(epoch)))
        if save_latest:
            path = os.path.join(self.models_dir, self._get_model_filename())
            torch.save(ckpt, path)
        return path

    def cleanup_old_models(self, keep_last_n=5):
        """Keep the newest keep_last_n per-epoch checkpoints of this configuration (trainer.py:414-441)."""
        if D.rank() != 0 or not os.path.exists(self.models_dir):
            return
        pattern = f"model_{self._get_config_hash()}_epoch_"
        found = []
        for name in os.listdir(self.models_dir):
            if name.startswith(pattern) and name.endswith('.pt'):
                try:
                    found.append((int(name[len(pattern):-3]), name))
                except ValueError:
                    continue
        for _, name in sorted(found)[:-keep_last_n] if len(found) > keep_last_n else []:
            os.remove(os.path.join(self.models_dir, name))

    def load_checkpoint(self, path, strict=True, convert_supervised_to_rl=False):
        """Weights-only load of a checkpoint written by save_model or by the reference (whose state-dict keys carry the
        ``module.`` prefix of nn.DataParallel, stripped here).  With convert_supervised_to_rl the two-class output layer
        of a supervised model becomes the one-score layer of the REINFORCE model by the logit difference
        (W1 - W0, b1 - b0), as the reference's model_converter.py:12-80 does.  Returns the checkpoint's epoch."""
        ckpt = torch.load(path, map_location=self.device, weights_only=True)
        sd = ckpt
        if isinstance(ckpt, dict):
            sd = ckpt.get('model_state_dict', ckpt.get('state_dict', ckpt))
        out = {}
        for k, v in sd.items():
            k = k[len('module.'):] if k.startswith('module.') else k
            if convert_supervised_to_rl and 'mlp_edges' in k and ('output_layer' in k or 'fc_out' in k) and v.shape[0] == 2:
                v = v[1:2] - v[0:1]
            out[k] = v
        self.net.load_state_dict(out, strict=strict and not convert_supervised_to_rl)
        D.broadcast_parameters(self.net.parameters())
        return ckpt.get('epoch') if isinstance(ckpt, dict) else None

    # ------------------------------------------------------------------ whole run
    def train(self, evaluator=None, metrics_logger=None):
        """The reference's schedule (trainer.py:482-584), epochs counted from 0: validation when
        ``epoch % val_every == 0`` or on the last epoch, the learning rate divided by ``decay_rate`` when
        ``epoch % val_every == 0 and epoch != 0`` (whether or not an evaluator is given), test when
        ``epoch % test_every == 0`` or on the last epoch, a checkpoint after every epoch unless ``save_model`` is false.
        Returns the trained network, as the reference does; the per-epoch numbers are kept in ``self.history``."""
        cfg = self.config
        max_epochs = cfg.max_epochs
        val_every, test_every = cfg.get('val_every', 1), cfg.get('test_every', max(max_epochs, 1))
        lr, decay_rate = cfg.learning_rate, cfg.get('decay_rate', 1.0)
        self.history = []
        for epoch in range(max_epochs):
            if hasattr(self.strategy, 'set_epoch'):
                self.strategy.set_epoch(epoch)
            t, loss, err, rl = self.train_one_epoch()
            if metrics_logger is not None:
                metrics_logger.log_train_metrics(loss, err, t)
                if rl:
                    metrics_logger.log_rl_metrics(epoch, rl)
            line = f"epoch {epoch + 1}/{max_epochs}  time {t:.2f}s  lr {lr:.2e}  loss {loss:.4f}  err_edges {err:.2f}"
            if rl:
                line += (f"  reward {rl['reward']:.4f}  entropy {rl.get('entropy', 0):.4f}  complete% {rl.get('complete_paths_rate', 0):.1f}"
                         f"  T {rl.get('temperature', 0):.2f}")
            last = epoch == max_epochs - 1
            if evaluator is not None and (epoch % val_every == 0 or last):
                self.net.eval()
                res = evaluator.evaluate(self.net, None, mode='val')
                if metrics_logger is not None:
                    metrics_logger.log_val_metrics(res[4], res[0], epoch=epoch, comp_rate=res[6], comp_sample_rate=res[7])
                line += f"  val_loss {res[1]:.4f}  val_load_factor {res[2]:.4f}  approx% {res[4]:.1f}  infeasible% {res[5]:.1f}"
            if epoch % val_every == 0 and epoch != 0:
                lr /= decay_rate
                self.update_learning_rate(lr)
            if evaluator is not None and cfg.get(f'num_test_data', 0) and (epoch % test_every == 0 or last):
                self.net.eval()
                res = evaluator.evaluate(self.net, None, mode='test')
                if metrics_logger is not None:
                    metrics_logger.log_test_metrics(res[4], res[0], epoch=epoch, comp_rate=res[6], comp_sample_rate=res[7])
                line += f"  test_loss {res[1]:.4f}  test_load_factor {res[2]:.4f}"
            if D.rank() == 0:
                print(line, flush=True)
            self.history.append((epoch + 1, t, loss, err))
            if cfg.get('save_model', True):
                self.save_model(epoch, loss)
                if cfg.get('cleanup_old_models', True):
                    self.cleanup_old_models()
        return self.net
