"""Drop-in counterpart of the reference's vae/base.py (BaseVAE / BaseEncoder /
BaseDecoder, /root/reference/vae/base.py:8-149): same constructor contract,
same checkpoint dictionary, same helper names.  Only what the CVAEMemRand hot
path uses is provided.
"""
from __future__ import annotations

import os
import random

import numpy as np
import torch
import torch.nn as nn


class _ConfiguredModule(nn.Module):
    """Keeps the (shared, mutable) config dict and the device string, as the
    reference's BaseEncoder/BaseDecoder do (base.py:9-12, 18-21)."""

    def __init__(self, config: dict):
        super().__init__()
        self.config = config
        self.device = config["device"]

    def forward(self, x):
        raise NotImplementedError()


class BaseEncoder(_ConfiguredModule):
    pass


class BaseDecoder(_ConfiguredModule):
    pass


class BaseVAE(nn.Module):
    def __init__(self, config: dict):
        super().__init__()
        self.config = config
        # read BEFORE any defaulting, like the reference (base.py:38-39): a config
        # without these keys raises KeyError there too
        self.kl_weight = config["kl_weight"]
        self.device = config["device"]
        self.encoder = None
        self.decoder = None

    def count_parameters(self):
        """Per-parameter element counts + total as a DataFrame (base.py:64-78)."""
        import pandas as pd
        rows = [(n, p.numel()) for n, p in self.named_parameters() if p.requires_grad]
        rows.append(("Total", sum(c for _, c in rows)))
        return pd.DataFrame({"module": [r[0] for r in rows], "num_params": [r[1] for r in rows]})

    def save_weights(self, optimizer: torch.optim.Optimizer, model_dir, filename):
        """Checkpoint format of base.py:126-137: {model, optimizer, model_config,
        system_rng, numpy_rng, torch_rng} -> {model_dir}/{filename}.pt"""
        os.makedirs(model_dir, exist_ok=True)
        torch.save({
            "model": self.state_dict(),
            "optimizer": optimizer.state_dict(),
            "model_config": self.config,
            "system_rng": random.getstate(),
            "numpy_rng": np.random.get_state(),
            "torch_rng": torch.random.get_rng_state(),
        }, f"{model_dir}/{filename}.pt")

    def load_weights(self, f: str = None, dict_to_load: dict = None):
        """base.py:139-149: restores only the "model" entry."""
        assert (f is not None) or (dict_to_load is not None), "One of file path or dict_to_load must not be None"
        if f is not None:
            dict_to_load = torch.load(f, weights_only=False)
        self.load_state_dict(dict_to_load["model"])
