"""Drop-in counterpart of the reference's legacy fixed-length model, vae/cvae_with_mem.py (CVAEMem, :466-738):
the origin of the MSE loss and of the one-channel (days, 1000, 5, 5) Monte-Carlo product.

It is the same three stacks as CVAEMemRand -- the module names and the state_dict keys are the same -- with
  * ONE decoder channel (dec_output = Conv2d(c, 1), :383-386): surfaces come back as (B, T-C, 5, 5);
  * a fixed sequence length: config["seq_len"] days of which the first config["ctx_len"] are the context;
  * F.mse_loss on the surfaces (:681, :722) instead of the pinball loss, and no gradient clipping (:691-692);
  * UNIFORM noise in the reparameterisation (torch.rand_like, :157 -- as written in the reference);
  * get_surface_given_conditions that uses the caller's / the prior's z at EVERY position (:547-556): the
    context positions are not overwritten with the posterior mean as CVAEMemRand does.
The kernels are CVAEMemRand's, run with one quantile channel; vsv_loss_grad computes the MSE (surf_l2).
"""
from __future__ import annotations

from typing import Dict

import torch

from .cvae_with_mem_randomized import CVAEMemRand


class CVAEMem(CVAEMemRand):
    def __init__(self, config: dict):
        for req in ("seq_len", "ctx_len"):
            if req not in config:
                raise ValueError(f"config doesn't contain: {req}" if req == "seq_len" else f"config doesn't contain {req}")
        config["quantiles"] = [0.5]                    # one decoder channel (the pinball machinery is not used)
        config["horizon"] = config["seq_len"] - config["ctx_len"]
        super().__init__(config)
        del self.quantile_loss_fn                      # (no pinball loss, and no such buffers in the legacy state_dict)

    # -- helpers ---------------------------------------------------------------
    def _loss_cfg(self):
        # MSE on the one channel; the ex-feature loss is on ALL the features (no ex_loss_on_ret_only switch, :683)
        import torch.nn as nn
        return dict(quantiles=[0.5], quantile_weights=[1.0], surf_l2=True,
                    ex_n=max(int(self.config["ex_feats_dim"]), 1),
                    ex_l2=int(isinstance(self.ex_feats_loss_fn, nn.MSELoss)),
                    ex_weight=float(self.config["re_feat_weight"]), kl_weight=float(self.kl_weight))

    def _fixed(self, x: Dict[str, torch.Tensor]):
        surf, ex = self._batched(x)
        if surf.shape[1] != self.config["seq_len"]:
            raise ValueError(f"sequence length must be seq_len = {self.config['seq_len']}, got {surf.shape[1]}")
        return surf, ex

    # -- forward / generation ----------------------------------------------------
    def forward(self, x: Dict[str, torch.Tensor], _eps: torch.Tensor = None):
        """Reference :603-649 -> (recon (B, T-C, 5, 5)[, ex (B, T-C, n)], z_mean, z_log_var, z).
        _eps (private): the reparameterisation noise (B, T, L) for parity tests; otherwise uniform on [0, 1)
        from torch's generator, like the reference's torch.rand_like."""
        from vsv.autograd import CVAEForward
        from vsv import paths
        surf, ex = self._fixed(x)
        B, T = surf.shape[:2]
        L = self.config["latent_dim"]
        eps = _eps if _eps is not None else torch.rand((B, T, L), device=self.device, dtype=torch.float32)
        dt = self._param_dtype()
        p = float(self.config.get("mem_dropout", 0) or 0) if self.training and self.config["mem_layers"] > 1 else 0.0
        params = self._hip_named_params()
        if (torch.is_grad_enabled() and any(q.requires_grad for _, q in params)) or p > 0:
            self._engine()
            s, e, zm, zl, z = CVAEForward.apply(self._train_engine(), surf, ex, self.horizon, eps, 0, p,
                                                self._draw_seed() if p > 0 else 0, *[q for _, q in params])
        else:
            s, e, zm, zl, z = paths.forward(self._engine(), surf, ex, self.horizon, eps=eps)
        s = s[:, :, 0]
        if ex is not None:
            return s.to(dt), e.to(dt), zm.to(dt), zl.to(dt), z.to(dt)
        return s.to(dt), zm.to(dt), zl.to(dt), z.to(dt)

    def get_surface_given_conditions(self, c: Dict[str, torch.Tensor], z: torch.Tensor = None, mu=0, std=1):
        """Reference :520-562: the decoder over all seq_len positions on cat([padded context embedding, z]),
        z given or drawn N(mu, std) at EVERY position; returns the last seq_len - ctx_len days."""
        from vsv import paths
        surf, ex = self._batched(c)
        C, T, L = self.config["ctx_len"], self.config["seq_len"], self.config["latent_dim"]
        assert surf.shape[1] == C, "context length mismatch"
        if ex is not None:
            assert ex.shape[1] == C, "context length mismatch"
        B = surf.shape[0]
        eng = self._engine()
        if z is not None:
            if z.dim() == 2:
                z = z.unsqueeze(0)
            assert z.shape[1] == T, "latent time dim mismatch"
            assert z.shape[2] == L, "latent dim mismatch"
        else:
            z = mu + eng.philox_normal(self._draw_seed(), 0, B, 1, 0, T).reshape(B, T, L) * std
        ctx = eng.encode(1, surf, ex)
        s, e = paths.decode_given_z(eng, ctx, z, C)
        dt = self._param_dtype()
        s = s[:, :, 0]
        if ex is not None:
            return s.to(dt), e.to(dt)
        return s.to(dt)

    # -- steps -----------------------------------------------------------------------
    def _legacy_losses(self, x, want_grads: bool, _eps=None):
        surf, ex = self._fixed(x)
        B, T = surf.shape[:2]
        eps = _eps if _eps is not None else torch.rand((B, T, self.config["latent_dim"]), device=self.device,
                                                       dtype=torch.float32)
        p = float(self.config.get("mem_dropout", 0) or 0) if self.training else 0.0
        te = self._train_engine()
        losses, gp, _ = te.loss_and_grads(surf, ex, self.horizon, self._loss_cfg(), eps=eps, want_grads=want_grads,
                                          dropout_p=p, dropout_seed=self._draw_seed() if p > 0 else 0)
        if ex is None:
            losses["re_ex_feats"] = torch.zeros(1)
        return losses, gp

    def train_step(self, x, optimizer: torch.optim.Optimizer, _eps=None):
        """Reference :651-700: MSE + ex-feature + KL loss, backward, optimizer.step() -- no clipping."""
        optimizer.zero_grad()
        losses, gp = self._legacy_losses(x, True, _eps)
        flat = self._accumulate_grads(gp)
        if flat is None or not self._adamw_step_(optimizer, flat):
            optimizer.step()
        return losses

    def test_step(self, x, _eps=None):
        """Reference :702-738."""
        return self._legacy_losses(x, False, _eps)[0]

    def train_step_multihorizon(self, *a, **k):
        raise AttributeError("CVAEMem has no train_step_multihorizon (a CVAEMemRand method)")

    def generate_autoregressive_sequence(self, *a, **k):
        raise AttributeError("CVAEMem has no generate_autoregressive_sequence (a CVAEMemRand method)")

    def generate_autoregressive_offset(self, *a, **k):
        raise AttributeError("CVAEMem has no generate_autoregressive_offset (a CVAEMemRand method)")
