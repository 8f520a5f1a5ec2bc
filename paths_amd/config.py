"""config.json surface of the reference (reference config.py:13-131), kept field-for-field so both shipped
JSON files (models/sample/config.json, models/brca_paths_0/config.json) load unchanged.

Out of scope here: dataset construction (``get_dataset``) and the preprocess
directory check — ``Config.load`` only validates the directory when ``test_mode`` is False AND the directory
is given, it never touches WSI files.
"""
from __future__ import annotations

import json
import os
from dataclasses import field, make_dataclass
from typing import List, Optional


class ModelConfig:
    """Base class of per-model configuration blocks (reference config.py:13-15)."""


def _spec(rows):
    return [(name, typ, field(default=default)) if default is not ... else (name, typ) for name, typ, default in rows]


# The field surface of config.json.  (name, type, default) — names and defaults are the reference's (config.py:19-37 and
# :41-79); `...` marks a required key.
_MODEL_FIELDS = [
    ("hierarchical_ctx", bool, True), ("slide_ctx_mode", str, "residual"),          # residual / concat / none
    ("patch_embed_dim", int, 1024), ("dropout", float, 0.0), ("patch_size", int, 256),
    ("importance_mode", str, "mul"),                                                # mul / none
    ("trans_dim", int, 192), ("trans_heads", int, 4), ("trans_layers", int, 2),
    ("pos_encoding_mode", str, "1d"),                                               # 1d / 2d
    ("importance_mlp_hidden_dim", int, 128), ("hierarchical_ctx_mlp_hidden_dim", int, 256), ("lstm", bool, True),
]
_TRAIN_FIELDS = [
    ("model_config", ModelConfig, ...),
    ("base_power", float, ...), ("magnification_factor", int, ...), ("num_levels", int, ...), ("num_epochs", int, ...),
    ("top_k_patches", List[int], ...), ("model_type", str, ...), ("wsi_dir", str, ...), ("csv_path", str, ...),
    ("nbins", int, 4), ("loss", str, "nll"), ("task", str, "survival"), ("filter_to_subtypes", Optional[List[str]], None),
    ("preprocess_dir", Optional[str], None), ("batch_size", int, 32), ("save_epochs", int, 10), ("eval_epochs", int, 1),
    ("lr", float, 2e-5), ("lr_decay_per_epoch", float, 0.99), ("seed", int, 0), ("early_stopping", bool, False),
    ("weight_decay", float, 1e-2), ("min_epochs", int, 0), ("root_name", str, ""), ("hipt_splits", bool, False),
    ("hipt_val_proportion", float, 0),
]

PATHSProcessorConfig = make_dataclass("PATHSProcessorConfig", _spec(_MODEL_FIELDS), bases=(ModelConfig,))
PATHSProcessorConfig.__module__ = __name__


class _ConfigMethods:
    @staticmethod
    def from_dict(data: dict) -> "Config":
        data = dict(data)
        levels = data["num_levels"]
        # normalisations of reference config.py:93-100
        if isinstance(data["top_k_patches"], int):
            data["top_k_patches"] = [data["top_k_patches"]] * (levels - 1)
        if isinstance(data["num_epochs"], list):
            data["num_epochs"] = data["num_epochs"][0]
        if isinstance(data.get("batch_size", 32), int):
            data["batch_size"] = [data.get("batch_size", 32)] * levels
        if data["model_type"] != "PATHS":
            raise NotImplementedError(f"Unknown model type '{data['model_type']}'")
        mc = data["model_config"]
        if isinstance(mc, dict):
            mc = PATHSProcessorConfig(**mc)
        assert not mc.lstm or mc.hierarchical_ctx, "If LSTM mode is enabled, hierarchical context must be enabled."
        data["model_config"] = mc
        return Config(**data)

    @staticmethod
    def load(root_path: str, test_mode: bool = False) -> "Config":
        """Read ``<root_path>/config.json`` (reference config.py:82-115)."""
        jsonpath = os.path.join(root_path, "config.json")
        assert os.path.isdir(root_path), f"Model directory '{root_path}' not found!"
        assert os.path.isfile(jsonpath), f"config.json not found in directory '{root_path}'."
        with open(jsonpath, "r") as fh:
            cfg = Config.from_dict(json.load(fh))
        if not test_mode and cfg.preprocess_dir is not None:
            assert os.path.isdir(cfg.preprocess_dir), f"Preprocessing root directory '{cfg.preprocess_dir}' not found!"
        return cfg

    def power_levels(self):
        return [self.base_power * self.magnification_factor ** i for i in range(self.num_levels)]

    def get_model(self):
        from .model.interface import RecursiveModel
        from .model.paths import PATHSProcessor
        if self.model_type != "PATHS":
            raise NotImplementedError(f"Unknown model '{self.model_type}'.")
        return RecursiveModel(PATHSProcessor, self.model_config, train_config=self)

    def get_lr_scheduler(self, optimizer):
        from torch.optim.lr_scheduler import ExponentialLR
        return ExponentialLR(optimizer, self.lr_decay_per_epoch)

    def get_dataset(self, *a, **k):
        raise NotImplementedError("dataset / split loading is outside the hot-path scope (SURVEY.md §2 row 12)")


Config = make_dataclass("Config", _spec(_TRAIN_FIELDS), bases=(_ConfigMethods,))
Config.__module__ = __name__
