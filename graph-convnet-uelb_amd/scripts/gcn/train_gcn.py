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
    trainer.train(evaluator, None)
    if config.get('num_test_data', 0) >= config.batch_size:
        res = evaluator.evaluate(trainer.get_model(), None, mode='test')
        if int(os.environ.get("RANK", "0")) == 0:
            print(f"test: loss {res[1]:.4f}  mean max load factor {res[2]:.4f}  infeasible% {res[5]:.1f}  "
                  f"completion% {res[6]:.1f}", flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
