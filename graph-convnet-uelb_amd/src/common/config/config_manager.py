"""``ConfigManager(path).get_config() / get_dtypes()`` as in the reference
(``src/common/config/config_manager.py:5-43``).  On ROCm the device is chosen with
``torch.cuda.set_device`` (one process per GPU) instead of rewriting CUDA_VISIBLE_DEVICES."""
import os

import torch

from .config import get_config


class ConfigManager:
    def __init__(self, config_path: str = "configs/gcn/default2.json"):
        self.config_path = config_path
        self.config = get_config(config_path)
        self.dtypeFloat, self.dtypeLong = torch.float32, torch.long
        if getattr(self.config, 'use_gpu', True) and torch.cuda.is_available():
            local_rank = int(os.environ.get("LOCAL_RANK", str(self.config.get('gpu_id', 0)).split(',')[0] or 0))
            torch.cuda.set_device(local_rank % torch.cuda.device_count())

    def get_config(self):
        return self.config

    def get_dtypes(self):
        return self.dtypeFloat, self.dtypeLong
