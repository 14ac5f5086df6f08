#!/usr/bin/env python3
"""Train the GCN with the MI355X HIP path: ``python scripts/gcn/train_gcn.py --config configs/gcn/<name>.json``
(same entry point and flag as the reference's ``scripts/gcn/train_gcn.py``).  For several GPUs launch it with
``python -m torch.distributed.run --nproc-per-node N scripts/gcn/train_gcn.py --config ...``."""
import argparse
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))

import torch  # noqa: E402

from src.common.config.config_manager import ConfigManager  # noqa: E402
from src.gcn.train.evaluator import Evaluator  # noqa: E402
from src.gcn.train.metrics import GCNMetricsLogger  # noqa: E402
from src.gcn.train.trainer import Trainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="configs/gcn/default2.json")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
    manager = ConfigManager(args.config)
    config = manager.get_config()
    f, l = manager.get_dtypes()
    trainer = Trainer(config, f, l)
    evaluator = Evaluator(config, f, l, strategy=trainer.strategy)      # as the reference (train_gcn.py:46)
    rank0 = int(os.environ.get("RANK", "0")) == 0
    # the reference builds its logger unconditionally (train_gcn.py:47); here rank 0 keeps the books
    metrics_logger = GCNMetricsLogger(os.environ.get('GCN_UELB_LOG_DIR', config.get('log_dir', 'logs'))) if rank0 and config.get('metrics_logger', True) else None
    trainer.train(evaluator, metrics_logger)
    info = {'config_file': args.config, 'mode': 'training', 'max_epochs': config.get('max_epochs', 0),
            'num_train_data': config.get('num_train_data', 0), 'num_test_data': config.get('num_test_data', 0),
            'num_val_data': config.get('num_val_data', 0)}
    if config.get('num_test_data', 0) >= config.batch_size:
        res = evaluator.evaluate(trainer.get_model(), None, mode='test')
        if rank0:
            print(f"test: loss {res[1]:.4f}  mean max load factor {res[2]:.4f}  infeasible% {res[5]:.1f}  "
                  f"completion% {res[6]:.1f}", flush=True)
    if metrics_logger is not None:
        metrics_logger.print_summary(info)
        paths = metrics_logger.save_results(info)
        print(f"results written to {paths[0]}", flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
