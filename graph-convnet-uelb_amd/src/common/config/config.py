"""Experiment settings: a dict with attribute access, loaded from JSON over ``configs/gcn/default.json``
(same behaviour as the reference's ``src/common/config/config.py:4-85``)."""
import json
import os

_PKG_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))


class Settings(dict):
    def __init__(self, config_dict=None):
        super().__init__()
        for key, value in (config_dict or {}).items():
            self[key] = value

    def __getattr__(self, attr):
        try:
            return self[attr]
        except KeyError:
            raise AttributeError(attr)

    def __setattr__(self, key, value):
        self[key] = value

    __delattr__ = dict.__delitem__


def get_default_config():
    for path in ("./configs/gcn/default.json", "./configs/default.json",
                 os.path.join(_PKG_ROOT, "configs", "gcn", "default.json")):
        if os.path.exists(path):
            with open(path) as f:
                return Settings(json.load(f))
    return Settings({})


def get_config(filepath):
    config = get_default_config()
    with open(filepath) as f:
        config.update(Settings(json.load(f)))
    return config
