"""CPU port of the CVAEMemRand hot path on the torch ops the reference itself
calls (nn.Conv2d / nn.ConvTranspose2d / nn.LSTM / nn.Linear on CPU).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  It exists for two jobs:
  * the `cpu_baseline` leg of bench.py: the reference never travels to the GPU
    box, and this port runs the same ATen CPU kernels in the same call shapes
    (one batched call per day), so its speed is what the reference's would be;
  * an autograd oracle for the training path (losses, gradients) on inputs the
    committed goldens do not cover.
Pinned by tests/test_oracle_vs_golden.py against the reference's goldens.
Written from the module structure in SURVEY.md appendix A, citing the reference
call sites in /root/reference/vae/cvae_with_mem_randomized.py.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class TorchPort(nn.Module):
    def __init__(self, cfg: dict, W: Dict[str, np.ndarray], dtype=torch.float32, dropout: float = 0.0):
        super().__init__()
        self.cfg = cfg
        self.L, self.X = cfg["latent_dim"], cfg["ex_feats_dim"]
        self.nl = cfg["mem_layers"]
        Hm = cfg["mem_hidden"]
        sh = cfg["surface_hidden"]
        self.dense = bool(cfg.get("use_dense_surface", False))
        E = sh[-1] * (1 if self.dense else 25) + self.X                                   # :93, :109, :375-378
        self.Hd = E
        Ec = cfg["ctx_surface_hidden"][-1] * (1 if self.dense else 25) + self.X
        self.p = nn.ParameterDict({k.replace(".", "/"): nn.Parameter(torch.tensor(v, dtype=dtype))
                                   for k, v in W.items() if ".mem." not in k})
        self.mems = nn.ModuleDict()
        Lc = self.L if cfg.get("compress_context", False) else Hm                         # :363-366
        for name, inp, hid in (("encoder", E, Hm), ("ctx_encoder", Ec, Hm), ("decoder", Lc + self.L, self.Hd)):
            kind = {"lstm": nn.LSTM, "gru": nn.GRU}.get(cfg.get("mem_type", "lstm"), nn.RNN)   # :154-159
            m = kind(inp, hid, num_layers=self.nl, batch_first=True, dropout=dropout)     # :155, :307, :476
            m = m.to(dtype)
            with torch.no_grad():
                for l in range(self.nl):
                    for part in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                        getattr(m, f"{part}_l{l}").copy_(torch.tensor(W[f"{name}.mem.{part}_l{l}"], dtype=dtype))
            self.mems[name] = m

    def w(self, name):
        return self.p[name.replace(".", "/")]

    # dropout factors injected by a test: {"encoder"|"ctx_encoder"|"decoder": (B,T,hidden) of 0 or 1/(1-p)}
    drop_masks: Optional[Dict[str, torch.Tensor]] = None

    def _mem(self, name, x):
        """nn.LSTM forward (:155, :307, :476).  With injected masks the two layers run as two
        one-layer LSTMs on the same parameters and layer 0's output is multiplied by the mask in
        between -- what nn.LSTM(dropout=p) does in train mode with its own Bernoulli draw."""
        m = self.mems[name]
        if not self.drop_masks or self.nl == 1:
            return m(x)[0]
        B = x.shape[0]
        h = x
        for l in range(self.nl):
            ws = [getattr(m, f"{p}_l{l}") for p in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
            z = torch.zeros(1, B, m.hidden_size, dtype=x.dtype)
            if isinstance(m, nn.RNN):
                h = torch._VF.rnn_tanh(h, z, ws, True, 1, 0.0, False, False, True)[0]
            elif isinstance(m, nn.GRU):
                h = torch._VF.gru(h, z, ws, True, 1, 0.0, False, False, True)[0]
            else:
                h = torch._VF.lstm(h, (z, z), ws, True, 1, 0.0, False, False, True)[0]
            if l + 1 < self.nl:
                h = h * self.drop_masks[name][:, : h.shape[1]]
        return h

    def named_reference_parameters(self):
        """(reference state_dict name, parameter) pairs."""
        out = {}
        for k, v in self.p.items():
            out[k.replace("/", ".")] = v
        for name, m in self.mems.items():
            for pn, pv in m.named_parameters():
                out[f"{name}.mem.{pn}"] = pv
        return out

    # -- modules ---------------------------------------------------------------
    def _embed(self, surf, prefix, stem, n):
        B, T = surf.shape[:2]
        if self.dense:                                                                # :84-93, :237-246
            x = surf.reshape(B * T, 25)
            stem = stem.replace("_conv", "_dense")
            for i in range(n):
                x = F.relu(F.linear(x, self.w(f"{prefix}.{stem}_{i}.weight"), self.w(f"{prefix}.{stem}_{i}.bias")))
            return x.reshape(B, T, -1)
        x = surf.reshape(B * T, 1, 5, 5)
        for i in range(n):
            x = F.relu(F.conv2d(x, self.w(f"{prefix}.{stem}_{i}.weight"), self.w(f"{prefix}.{stem}_{i}.bias"),
                                padding=1))                                           # :100-103, :252-255
        return x.reshape(B, T, -1)

    def _interact(self, x, prefix, stem):
        """The `interaction` Sequential (:131-142, :283-294, :452-463); Identity for interaction_layers None / 0."""
        n = self.cfg.get("interaction_layers") or 0
        if n <= 0:
            return x
        for i in range(n):
            x = F.relu(F.linear(x, self.w(f"{prefix}.interaction.{stem}_interact_linear_{i}.weight"),
                                self.w(f"{prefix}.interaction.{stem}_interact_linear_{i}.bias")))
        return F.linear(x, self.w(f"{prefix}.interaction.{stem}_interact_linear_final.weight"),
                        self.w(f"{prefix}.interaction.{stem}_interact_linear_final.bias"))

    def encoder(self, surf, ex):
        e = self._embed(surf, "encoder.surface_embedding", "enc_conv", len(self.cfg["surface_hidden"]))
        if ex is not None:
            e = torch.cat([e, ex], dim=-1)                                            # :188
        e = self._interact(e, "encoder", "enc")
        h = self._mem("encoder", e)
        zm = F.linear(h, self.w("encoder.z_mean_layer.weight"), self.w("encoder.z_mean_layer.bias"))
        zl = F.linear(h, self.w("encoder.z_log_var_layer.weight"), self.w("encoder.z_log_var_layer.bias"))
        return zm, zl

    def ctx_encoder(self, surf, ex):
        e = self._embed(surf, "ctx_encoder.ctx_surface_embedding", "ctx_enc_conv",
                        len(self.cfg["ctx_surface_hidden"]))
        if ex is not None:
            e = torch.cat([e, ex], dim=-1)
        e = self._interact(e, "ctx_encoder", "ctx")
        h = self._mem("ctx_encoder", e)
        if not self.cfg.get("compress_context", False):
            return h                                                                  # nn.Identity(), :228
        return F.linear(h, self.w("ctx_encoder.final_compression.weight"),
                        self.w("ctx_encoder.final_compression.bias"))                # :343

    def decoder(self, dec_in):
        B, T = dec_in.shape[:2]
        h = self._mem("decoder", dec_in)                                              # :491
        h = self._interact(h, "decoder", "dec")                                       # :492
        sh = self.cfg["surface_hidden"]
        s = F.linear(h, self.w("decoder.surface_decoder_input.weight"), self.w("decoder.surface_decoder_input.bias"))
        if self.dense:                                                                # :394-407, :495-499
            s = s.reshape(B * T, sh[-1])
            for nm in [f"dec_dense_{i}" for i in range(len(sh) - 1)] + ["dec_final"]:
                s = F.relu(F.linear(s, self.w(f"decoder.surface_decoder.{nm}.weight"),
                                    self.w(f"decoder.surface_decoder.{nm}.bias")))
            s = F.linear(s, self.w("decoder.surface_decoder.dec_output.weight"),
                         self.w("decoder.surface_decoder.dec_output.bias"))
        else:
            s = s.reshape(B * T, sh[-1], 5, 5)                                        # :501
            names = [f"dec_deconv_{i}" for i in range(len(sh) - 1)] + ["dec_final"]
            for nm in names:
                s = F.relu(F.conv_transpose2d(s, self.w(f"decoder.surface_decoder.{nm}.weight"),
                                              self.w(f"decoder.surface_decoder.{nm}.bias"), padding=1))
            s = F.conv2d(s, self.w("decoder.surface_decoder.dec_output.weight"),
                         self.w("decoder.surface_decoder.dec_output.bias"), padding=1)
        s = s.reshape(B, T, -1, 5, 5)
        e = None
        if self.X > 0:
            e = F.linear(h, self.w("decoder.ex_feats_decoder_input.weight"), self.w("decoder.ex_feats_decoder_input.bias"))
            e = F.linear(e, self.w("decoder.ex_feats_decoder.weight"), self.w("decoder.ex_feats_decoder.bias"))
        return s, e

    # -- model-level paths --------------------------------------------------------
    def forward(self, surf, ex, eps, horizon):
        """CVAEMemRand.forward (:1022-1074) with injected eps (B,T,L)."""
        B, T = surf.shape[:2]
        C = T - horizon
        ctx = self.ctx_encoder(surf[:, :C], None if ex is None else ex[:, :C])
        ctx_pad = torch.cat([ctx, torch.zeros(B, T - C, ctx.shape[-1], dtype=ctx.dtype)], dim=1)   # :1061-1062
        zm, zl = self.encoder(surf, ex)
        z = zm + torch.exp(0.5 * zl) * eps                                            # :199
        s, e = self.decoder(torch.cat([ctx_pad, z], dim=-1))                          # :1066
        return s[:, C:], (None if e is None else e[:, C:]), zm, zl, z

    def get_surface_given_conditions(self, surf, ex, z, horizon):
        """:642-715, literal: z (B,C+H,L) is overwritten in place on [:, :C]."""
        B, C = surf.shape[:2]
        T = C + horizon
        with torch.no_grad():
            zm, _ = self.encoder(surf, ex)
            z[:, :C] = zm
            ctx = self.ctx_encoder(surf, ex)
            ctx_pad = torch.cat([ctx, torch.zeros(B, T - C, ctx.shape[-1], dtype=ctx.dtype)], dim=1)
            s, e = self.decoder(torch.cat([ctx_pad, z], dim=-1))
        return s[:, C:], (None if e is None else e[:, C:])

    def losses(self, surf, ex, eps, horizon, quantile_weights=None):
        """Loss terms of train_step / test_step (:1113-1125): pinball (float64
        buffers, :23-26), ex-feature MSE/L1 (ret-only option), KL over all T."""
        cfg = self.cfg
        C = surf.shape[1] - horizon
        s, e, zm, zl, _ = self.forward(surf, ex, eps, horizon)
        q = torch.tensor(cfg["quantiles"], dtype=torch.float64)
        w = torch.ones_like(q) if quantile_weights is None else torch.tensor(quantile_weights, dtype=torch.float64)
        target = surf[:, C:]
        per_q = []
        for i in range(len(q)):
            err = target - s[:, :, i]
            per_q.append(torch.mean(torch.max((q[i] - 1) * err, q[i] * err)))
        re_s = torch.sum(torch.stack(per_q) * w) / torch.sum(w)
        re_e = torch.zeros((), dtype=torch.float64)
        recon = re_s
        if ex is not None:
            pe, te = e, ex[:, C:]
            if cfg.get("ex_loss_on_ret_only", False):
                pe, te = pe[:, :, :1], te[:, :, :1]
            re_e = F.mse_loss(pe, te) if cfg.get("ex_feats_loss_type", "l1") == "l2" else F.l1_loss(pe, te)
            recon = re_s + cfg.get("re_feat_weight", 1.0) * re_e
        kl = -0.5 * (1 + zl - torch.exp(zl) - torch.square(zm))
        kl = torch.mean(torch.sum(kl, dim=-1))
        return {"loss": recon + cfg["kl_weight"] * kl, "re_surface": re_s, "re_ex_feats": re_e,
                "reconstruction_loss": recon, "kl_loss": kl}
