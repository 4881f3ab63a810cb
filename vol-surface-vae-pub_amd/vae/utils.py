"""Training / evaluation loops: drop-in for the loops of the reference's vae/utils.py
(set_seeds :15, model_eval :23, train :45, train_with_scheduled_sampling :102,
test :262).  Same signatures, same checkpointing (best validation loss ->
model.save_weights) and the same log-file format; the model underneath is the
MI355X CVAEMemRand.  Autoregressive training (train_autoregressive_step :426-623,
train_autoregressive_multi_offset :626-669) is below, as a two-pass rollout over the HIP
kernels.  Plot helpers (:272-418) are outside this build's scope.
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


def _ar_context_plan(step: int, C: int, horizon: int, offset: int):
    """Where every day of the context for AR step `step + 1` comes from, following the reference's
    window assembly (vae/utils.py:537-610) literally: a list of (source, index) with source -1 = real
    calendar day `index`, source j >= 0 = day `index` of step j's prediction.  Overlapping predictions
    (offset < horizon) are concatenated in step order and the LAST C days are kept, duplicates included,
    exactly as the reference does.  Also returns how the reference fills the context's ex_feats:
    ("real", start) = real rows from `start`, padded with the last real row; ("last",) = the last real
    context row repeated."""
    start = (step + 1) * offset
    if start + C <= C:
        return [(-1, i) for i in range(start, start + C)], ("real", start)
    plan = []
    if start < C:
        plan += [(-1, i) for i in range(start, C)]
    for j in range(step + 1):
        g0 = C + j * offset
        lo, hi = max(g0, start), min(g0 + horizon, start + C)
        if lo < hi:
            plan += [(j, i) for i in range(lo - g0, hi - g0)]
    if start < C:
        return plan, ("real", start)
    return plan[-C:], ("last",)


def train_autoregressive_step(model, batch, optimizer, ar_steps, horizon, offset, _eps_steps=None):
    """Reference vae/utils.py:426-623: `ar_steps` teacher-forced forwards of `horizon` days whose context
    slides by `offset` and is re-assembled from real days and the p50 channel of earlier predictions;
    pinball (+ ex-feature) + N(0,1) KL per step, the MEAN over steps back-propagated THROUGH the fed-back
    predictions (no detach, :526-527), optimizer.step() without clipping.  Returns floats.

    On the HIP path the rollout runs twice: a forward pass to obtain the predictions, then, last step
    first, one forward+backward per step (same Philox / dropout seeds, so the same numbers) that also
    returns the gradient w.r.t. its input surfaces; the part that falls on predicted days is added to the
    output gradient of the step that predicted them.  _eps_steps: injected posterior noise per step."""
    import torch
    optimizer.zero_grad()
    surface = batch["surface"].to(model.device)
    B, T = surface.shape[:2]
    C = model.config.get("context_len", 60)
    required_len = C + (ar_steps * offset)
    if T < required_len:
        raise ValueError(f"Sequence too short for offset={offset}, ar_steps={ar_steps}: "
                         f"need {required_len} days, got {T} days")
    has_ex = "ex_feats" in batch
    ex_all = batch["ex_feats"].to(model.device) if has_ex else None
    te = model._train_engine()
    cfg = model._loss_cfg()
    cfg["kl_weight"] = float(model.config["kl_weight"])
    p = float(model.config.get("mem_dropout", 0) or 0) if model.training else 0.0
    seeds = [(model._draw_seed(), model._draw_seed()) for _ in range(ar_steps)]

    def run(k, ctx_s, ctx_e, want_grads, extra=None):
        t0 = C + k * offset
        s_in = torch.cat([ctx_s, surface[:, t0:t0 + horizon]], dim=1)
        e_in = torch.cat([ctx_e, ex_all[:, t0:t0 + horizon]], dim=1) if has_ex else None
        return te.loss_and_grads(s_in, e_in, horizon, cfg, eps=None if _eps_steps is None else _eps_steps[k],
                                 seed=seeds[k][0], want_grads=want_grads, dropout_p=p, dropout_seed=seeds[k][1],
                                 extra_d_pred=extra, want_input_grad=want_grads)

    # ---- pass 1: predictions and contexts
    ctx_s = surface[:, :C]
    ctx_e = ex_all[:, :C] if has_ex else None
    contexts, plans, preds = [(ctx_s, ctx_e)], [[(-1, i) for i in range(C)]], []
    with torch.no_grad():
        for k in range(ar_steps - 1):
            _, _, out = run(k, ctx_s, ctx_e, False)
            preds.append(out["surf"][:, :, 1])                              # p50 channel (B, horizon, 5, 5)
            plan, ex_rule = _ar_context_plan(k, C, horizon, offset)
            ctx_s = torch.stack([surface[:, i] if j < 0 else preds[j][:, i] for j, i in plan], dim=1)
            if has_ex:
                if ex_rule[0] == "real":
                    real = ex_all[:, ex_rule[1]:C]
                    ctx_e = real[:, :C] if real.shape[1] >= C else \
                        torch.cat([real, real[:, -1:].repeat(1, C - real.shape[1], 1)], dim=1)
                else:
                    last = min(C - 1, ex_all.shape[1] - 1)
                    ctx_e = ex_all[:, last:last + 1].repeat(1, C, 1)
            contexts.append((ctx_s, ctx_e))
            plans.append(plan)
    # ---- pass 2: last step first; gradients flow back into the predictions that fed later contexts
    Q = len(model.config["quantiles"])
    extra = [None] * ar_steps
    tot = dict(loss=0.0, recon=0.0, kl=0.0)
    for k in range(ar_steps - 1, -1, -1):
        losses, gp, out = run(k, contexts[k][0], contexts[k][1], True, extra[k])
        model._accumulate_grads(gp, 1.0 / ar_steps)
        tot["loss"] += float(losses["loss"])
        tot["recon"] += float(losses["reconstruction_loss"])
        tot["kl"] += float(losses["kl_loss"])
        d_ctx = out["d_surface"][:, :len(plans[k])]
        for pos, (j, i) in enumerate(plans[k]):
            if j >= 0:
                if extra[j] is None:
                    extra[j] = torch.zeros((B, horizon, Q, 5, 5), dtype=torch.float32, device=d_ctx.device)
                extra[j][:, i, 1] += d_ctx[:, pos]
    optimizer.step()
    return {"loss": tot["loss"] / ar_steps, "reconstruction_loss": tot["recon"] / ar_steps,
            "kl_loss": tot["kl"] / ar_steps, "ar_steps": ar_steps, "offset": offset}


def train_autoregressive_multi_offset(model, batch, optimizer, horizon, offsets, ar_steps=3):
    """Reference vae/utils.py:626-669: one AR training step with an offset drawn from `offsets`."""
    offset = random.choice(offsets)
    return train_autoregressive_step(model=model, batch=batch, optimizer=optimizer, ar_steps=ar_steps,
                                     horizon=horizon, offset=offset)
