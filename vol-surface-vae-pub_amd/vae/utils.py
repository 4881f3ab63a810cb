"""Training / evaluation loops: drop-in for the loops of the reference's vae/utils.py
(set_seeds :15, model_eval :23, train :45, train_with_scheduled_sampling :102,
test :262).  Same signatures, same checkpointing (best validation loss ->
model.save_weights) and the same log-file format; the model underneath is the
MI355X CVAEMemRand.  Plot helpers (:272-418) and autoregressive *training*
(:426-669) are outside this build's scope.
"""
from __future__ import annotations

import json
import os
import random
import time
from collections import defaultdict

import numpy as np
import torch
import torch.optim as opt
from torch.utils.data import DataLoader

try:
    from tqdm import tqdm
except Exception:                                           # pragma: no cover
    def tqdm(it, **kw):
        return it


def set_seeds(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)


def _to_float(v):
    return v if isinstance(v, float) else v.item()


def model_eval(model, dataloader):
    """Average of test_step's loss terms over a dataloader (reference :23-42)."""
    model.eval()
    eval_losses = defaultdict(float)
    num_batches = 0
    for batch in tqdm(dataloader, desc="eval"):
        losses = model.test_step(batch)
        for k, v in losses.items():
            eval_losses[k] += _to_float(v)
        num_batches += 1
    for k in eval_losses:
        eval_losses[k] /= max(num_batches, 1)
    return eval_losses


def _open_log(model_dir, file_name, epochs):
    os.makedirs(model_dir, exist_ok=True)
    prefix = file_name.split(".")[0] if "." in file_name else file_name
    return prefix, open(f"{model_dir}/{prefix}-{epochs}-log.txt", "w", encoding="utf-8")


def _json_config(cfg):
    return json.dumps({k: (str(v) if isinstance(v, torch.device) else v) for k, v in cfg.items()}, indent=True,
                      default=str)


def _fmt(d):
    return ", ".join(f"{k}: {v:.3f}" for k, v in d.items())


def train(model, train_dataloader: DataLoader, valid_dataloader: DataLoader, lr=1e-5, epochs=100,
          model_dir="./", file_name="vanilla.pt"):
    """Reference :45-100: AdamW(lr), train_step per batch, model_eval per epoch, best
    validation loss checkpointed with save_weights."""
    model.train()
    optimizer = opt.AdamW(model.parameters(), lr)
    best_dev_loss = np.inf
    prefix, log_file = _open_log(model_dir, file_name, epochs)
    print("Model config: ", file=log_file)
    print(_json_config(model.config), file=log_file)
    print(f"LR: {lr}", file=log_file)
    print(f"Epochs: {epochs}", file=log_file)
    print("", file=log_file)
    log_file.flush()
    start_time = time.time()
    train_losses, dev_losses = {}, {}
    for epoch in range(epochs):
        t0 = time.time()
        model.train()
        train_losses = defaultdict(float)
        num_batches = 0
        for batch in tqdm(train_dataloader, desc=f"train-{epoch}"):
            losses = model.train_step(batch, optimizer)
            for k, v in losses.items():
                train_losses[k] += _to_float(v)
            num_batches += 1
        for k in train_losses:
            train_losses[k] /= max(num_batches, 1)
        dev_losses = model_eval(model, valid_dataloader)
        if dev_losses["loss"] < best_dev_loss:
            best_dev_loss = dev_losses["loss"]
            model.save_weights(optimizer, model_dir, prefix)
        msg = (f"epoch {epoch}: \ntrain loss :: {_fmt(train_losses)}, \ndev loss :: {_fmt(dev_losses)}, "
               f"\ntime elapsed :: {time.time() - t0}")
        print(msg)
        print(msg, file=log_file)
    print(f"training finished, total time :: {time.time() - start_time}")
    print(f"training finished, total time :: {time.time() - start_time}", file=log_file)
    log_file.close()
    return train_losses, dev_losses


def train_with_scheduled_sampling(model, train_dataloader: DataLoader, valid_dataloader: DataLoader, lr=1e-5,
                                  epochs=400, model_dir="./", file_name="backfill_model.pt",
                                  teacher_forcing_epochs=200, horizons=[1, 7, 14, 30]):
    """Reference :102-260: phase 1 (epoch < teacher_forcing_epochs) train_step, phase 2
    train_step_multihorizon; validation is always single-step."""
    model.train()
    optimizer = opt.AdamW(model.parameters(), lr)
    best_dev_loss = np.inf
    prefix, log_file = _open_log(model_dir, file_name, epochs)
    for line in ("=" * 80, "SCHEDULED SAMPLING TRAINING", "=" * 80, "Model config: ", _json_config(model.config),
                 f"LR: {lr}", f"Total Epochs: {epochs}", f"Teacher Forcing Epochs: {teacher_forcing_epochs} (Phase 1)",
                 f"Multi-Horizon Epochs: {epochs - teacher_forcing_epochs} (Phase 2)",
                 f"Multi-Horizon Targets: {horizons}", "=" * 80, ""):
        print(line, file=log_file)
    log_file.flush()
    start_time = time.time()
    train_losses, dev_losses = {}, {}
    for epoch in range(epochs):
        t0 = time.time()
        model.train()
        train_losses = defaultdict(float)
        num_batches = 0
        phase1 = epoch < teacher_forcing_epochs
        mode_str = (f"Phase 1 - Teacher Forcing (epoch {epoch + 1}/{teacher_forcing_epochs})" if phase1 else
                    f"Phase 2 - Multi-Horizon {horizons} (epoch {epoch + 1 - teacher_forcing_epochs}/"
                    f"{epochs - teacher_forcing_epochs})")
        print(f"\n{mode_str}")
        for batch in tqdm(train_dataloader, desc=f"train-{epoch}"):
            if phase1:
                losses = model.train_step(batch, optimizer)
            else:
                losses = model.train_step_multihorizon(batch, optimizer, horizons=horizons)
            for k, v in losses.items():
                if isinstance(v, dict):
                    for h, lv in v.items():
                        train_losses[f"{k}_h{h}"] += lv
                else:
                    train_losses[k] += _to_float(v)
            num_batches += 1
        for k in train_losses:
            train_losses[k] /= max(num_batches, 1)
        dev_losses = model_eval(model, valid_dataloader)
        marker = ""
        if dev_losses["loss"] < best_dev_loss:
            best_dev_loss = dev_losses["loss"]
            model.save_weights(optimizer, model_dir, prefix)
            marker = "BEST"
        dt = time.time() - t0
        print(mode_str)
        print(f"train loss :: {_fmt(train_losses)}")
        print(f"dev loss :: {_fmt(dev_losses)}")
        print(f"time :: {dt:.1f}s {marker}")
        print(f"epoch {epoch}: \ntrain loss :: {_fmt(train_losses)}, \ndev loss :: {_fmt(dev_losses)}, "
              f"\ntime elapsed :: {dt} {marker}", file=log_file)
        log_file.flush()
    total = time.time() - start_time
    print(f"Training finished! Total time: {total:.1f}s ({total / 60:.1f} min)")
    print(f"Best validation loss: {best_dev_loss:.6f}")
    print(f"\ntraining finished, total time :: {total}", file=log_file)
    print(f"best validation loss :: {best_dev_loss}", file=log_file)
    log_file.close()
    return train_losses, dev_losses


def test(model, valid_dataloader: DataLoader, test_dataloader: DataLoader, model_fn="./vanilla"):
    """Reference :262-270."""
    model.load_weights(f=model_fn)
    dev_losses = model_eval(model, valid_dataloader)
    test_losses = model_eval(model, test_dataloader)
    print(f"dev loss: {_fmt(dev_losses)}, \ntest_loss: {_fmt(test_losses)}")
    return dev_losses, test_losses


def train_autoregressive_step(*args, **kwargs):
    raise NotImplementedError("autoregressive TRAINING (reference vae/utils.py:426-623) is not part of this build "
                              "yet (SURVEY.md 8f item 4); autoregressive GENERATION is (vsv.paths.ar_generate)")


def train_autoregressive_multi_offset(*args, **kwargs):
    raise NotImplementedError("autoregressive TRAINING (reference vae/utils.py:626-669) is not part of this build yet")
