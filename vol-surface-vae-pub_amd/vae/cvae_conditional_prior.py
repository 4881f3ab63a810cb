"""CVAEMemRandConditionalPrior on MI355X: drop-in for /root/reference/vae/cvae_conditional_prior.py:34
(learned conditional prior p(z | context) in place of N(0,1): Gaussian-Gaussian KL in training, prior
sampling from the network at generation, optional sampled context latents and AR(1) latent noise).

Every network evaluation runs on the gfx950 kernels: the prior network occupies the posterior encoder's
slot of a second packed weight blob (see vae/conditional_prior_network.py), so generation and training
reuse vsv_encode / vsv_decode / the BPTT kernels unchanged.  The Gaussian-Gaussian KL itself is a few
element-wise torch ops on (B,C,L) device tensors.
"""
from __future__ import annotations

import math
import weakref
from typing import Dict, Optional

import torch

from vae.conditional_prior_network import ConditionalPriorNetwork, kl_divergence_gaussians
from vae.cvae_with_mem_randomized import CVAEMemRand

_PRIOR_TO_ENCODER = {"compression": None, "prior_mean_layer": None, "prior_logvar_layer": None}


class CVAEMemRandConditionalPrior(CVAEMemRand):
    def __init__(self, config: dict):
        super().__init__(config)
        self.use_conditional_prior = config.get("use_conditional_prior", False)
        if self.use_conditional_prior:
            print("Initializing Conditional Prior Network...")
            self.prior_network = ConditionalPriorNetwork(config)
            self.prior_network.to(self.device)
            self.prior_network.__dict__["_vsv_owner"] = weakref.ref(self)
            print("✓ Conditional Prior Network initialized")
        else:
            self.prior_network = None
            print("Note: use_conditional_prior=False, using standard N(0,1) prior")
        self.__dict__["_vsv_prior_engine"] = None
        self.__dict__["_vsv_prior_key"] = None
        self.__dict__["_vsv_prior_te"] = None

    # -- the main blob holds only the base model's parameters ------------------------------------
    def _hip_named_params(self):
        return [(n, p) for n, p in self.named_parameters() if not n.startswith("prior_network.")]

    # -- prior network in the posterior encoder's slot of a second blob ---------------------------
    def _prior_named_params(self):
        """The base model's parameter list with the encoder.* entries replaced by the prior network:
        conv and LSTM tensors one for one, compression and heads folded into the two head Linears."""
        pn = self.prior_network
        sub = {}
        kind = "dense" if self.config["use_dense_surface"] else "conv"
        for i in range(len(self.config["ctx_surface_hidden"])):
            conv = getattr(pn.surface_embedding, f"prior_enc_{kind}_{i}")
            sub[f"encoder.surface_embedding.enc_{kind}_{i}.weight"] = conv.weight
            sub[f"encoder.surface_embedding.enc_{kind}_{i}.bias"] = conv.bias
        for n, p in pn.interaction.named_parameters():              # prior_interact_linear_* -> enc_interact_linear_*
            sub["encoder.interaction." + n.replace("prior_interact", "enc_interact")] = p
        for n, p in pn.mem.named_parameters():
            sub[f"encoder.mem.{n}"] = p
        for head, name in ((pn.prior_mean_layer, "z_mean_layer"), (pn.prior_logvar_layer, "z_log_var_layer")):
            if self.config["compress_context"]:
                Wc, bc = pn.compression.weight, pn.compression.bias
                sub[f"encoder.{name}.weight"] = head.weight @ Wc
                sub[f"encoder.{name}.bias"] = head.weight @ bc + head.bias
            else:                                                   # compression is nn.Identity()
                sub[f"encoder.{name}.weight"], sub[f"encoder.{name}.bias"] = head.weight, head.bias
        out = []
        for n, p in CVAEMemRandConditionalPrior._hip_named_params(self):
            out.append((n, sub.get(n, p)))
        return out

    def _prior_eng(self):
        from vsv.engine import Engine
        main = self._engine()
        if list(self.config["ctx_surface_hidden"]) != list(self.config["surface_hidden"]):
            raise NotImplementedError("the prior network runs in the encoder's slot: ctx_surface_hidden must equal surface_hidden")
        eng = self.__dict__["_vsv_prior_engine"]
        if eng is None:
            eng = Engine(self.config, None, main.device)
            self.__dict__["_vsv_prior_engine"] = eng
        key = tuple((p.data_ptr(), p._version) for p in self.prior_network.parameters())
        if key != self.__dict__["_vsv_prior_key"]:
            with torch.no_grad():
                eng.set_weights_from_params(self._prior_named_params())
            self.__dict__["_vsv_prior_key"] = key
        return eng

    def _prior_forward(self, context):
        surf, ex = self._batched(context)
        o = self._prior_eng().encode(0, surf, ex)
        L = self.config["latent_dim"]
        dt = self._param_dtype()
        return o[..., :L].contiguous().to(dt), o[..., L:].contiguous().to(dt)

    # -- KL ---------------------------------------------------------------------------------------
    def compute_kl_loss(self, z_mean, z_logvar, context):
        """Reference :65-115: KL(q(z|x)[:, :C] || p(z|context)) with the conditional prior, else the N(0,1) KL."""
        if self.use_conditional_prior and self.prior_network is not None:
            T = context["surface"].shape[1]
            C = T - self.horizon
            ctx_only = {"surface": context["surface"][:, :C]}
            if "ex_feats" in context:
                ctx_only["ex_feats"] = context["ex_feats"][:, :C]
            pm, plv = self._prior_forward(ctx_only)
            return kl_divergence_gaussians(z_mean[:, :C].to(pm.device), z_logvar[:, :C].to(pm.device), pm, plv)
        kl = -0.5 * (1 + z_logvar - torch.exp(z_logvar) - torch.square(z_mean))
        return torch.mean(torch.sum(kl, dim=-1))

    # -- generation -----------------------------------------------------------------------------------
    def get_surface_given_conditions(self, c: Dict[str, torch.Tensor], z: torch.Tensor = None, mu=0, std=1, horizon=None,
                                     prior_mode="standard", sample_context: bool = False, ar_phi: float = 0.0,
                                     _eps_future: Optional[torch.Tensor] = None, _eps_ctx: Optional[torch.Tensor] = None):
        """Reference :113-239.  With the conditional prior and z=None the future latents are drawn from
        p(z | context) at the LAST context step (iid, or AR(1)-correlated with ar_phi > 0); context
        latents are the posterior mean, or sampled when sample_context.  _eps_future (B,H,L) / _eps_ctx
        (B,C,L) inject the standard-normal draws (parity tests); otherwise Philox on device."""
        if z is not None or not (self.use_conditional_prior and self.prior_network is not None):
            if sample_context:
                raise NotImplementedError("sample_context needs the conditional prior path (z=None)")
            return super().get_surface_given_conditions(c, z=z, mu=mu, std=std, horizon=horizon, prior_mode=prior_mode)
        from vsv import paths
        surf, ex = self._batched(c)
        B, C = surf.shape[:2]
        if ex is not None:
            assert ex.shape[1] == C, "context length mismatch"
        if horizon is None:
            horizon = self.horizon
        T, L = C + horizon, self.config["latent_dim"]
        eng, peng = self._engine(), self._prior_eng()
        prior = peng.encode(0, surf, ex)                                   # (B, C, 2L): prior mean | logvar
        zm, zl = paths.encode_both(eng, surf, ex)
        ctx = eng.encode(1, surf, ex)
        zb = torch.empty((B, T, L), dtype=torch.float32, device=eng.device)
        lv = torch.full((B, T, L), -float("inf"), dtype=torch.float32, device=eng.device)   # scale exp(lv/2) = 0
        zb[:, :C] = zm
        zb[:, C:] = prior[:, -1:, :L]
        lv[:, C:] = prior[:, -1:, L:]
        if sample_context:
            lv[:, :C] = zl
        eps = None
        if ar_phi > 0 or _eps_future is not None or _eps_ctx is not None:
            # materialised noise: AR(1) filter of the innovations (reference :180-188), or injected draws
            eps = torch.zeros((B, T, L), dtype=torch.float32, device=eng.device)
            fut = _eps_future.to(eng.device, torch.float32) if _eps_future is not None else \
                eng.philox_normal(self._draw_seed(), 0, B, 1, 0, horizon)
            if ar_phi > 0:
                fut = fut.clone()
                for t in range(1, horizon):
                    fut[:, t] = ar_phi * fut[:, t - 1] + math.sqrt(1 - ar_phi ** 2) * fut[:, t]
            eps[:, C:] = fut
            if sample_context:
                eps[:, :C] = _eps_ctx.to(eng.device, torch.float32) if _eps_ctx is not None else \
                    eng.philox_normal(self._draw_seed(), 0, B, 1, 0, C)
        r = eng.decode(B, T, C, ctx=ctx, zb=zb, lv=lv, eps=eps, seed=self._draw_seed(), S=1)
        dt = self._param_dtype()
        if ex is not None:
            return r["surf"].to(dt), r["ex"].to(dt)
        return r["surf"].to(dt)

    # -- training ---------------------------------------------------------------------------------------
    def _prior_train_engine(self):
        from vsv.train import TrainEngine
        peng = self._prior_eng()
        te = self.__dict__["_vsv_prior_te"]
        if te is None or te.eng is not peng:
            te = TrainEngine(peng, [(n, tuple(p.shape)) for n, p in self._prior_named_params()])
            self.__dict__["_vsv_prior_te"] = te
        return te

    def _cond_loss_terms(self, x, horizon, want_grads, eps=None):
        """Loss terms with the Gaussian-Gaussian KL; with want_grads also the packed-blob gradients of the
        base model and of the prior network's blob."""
        surf, ex = self._batched(x)
        T = surf.shape[1]
        C = T - horizon
        if C < 1:
            raise ValueError(f"Insufficient sequence length: T={T}, horizon={horizon}, resulting in C={C}. Need at least C >= 1.")
        te, pte = self._train_engine(), self._prior_train_engine()
        klw = float(self.kl_weight)
        L = self.config["latent_dim"]
        p = float(self.config.get("mem_dropout", 0) or 0) if self.training else 0.0
        dseed = self._draw_seed() if p > 0 else 0
        stash = {}

        def kl_fn(zm, zl):
            q = torch.cat([zm[:, :C], zl[:, :C]], dim=-1).detach().requires_grad_(want_grads)

            def d_out_fn(out):
                o = out.detach().requires_grad_(True)
                with torch.enable_grad():
                    kl = kl_divergence_gaussians(q[..., :L], q[..., L:], o[..., :L], o[..., L:])
                    g_o, g_q = torch.autograd.grad(klw * kl, [o, q])
                stash["kl"], stash["g_q"] = kl.detach(), g_q
                return g_o

            out, gpp = pte.encoder_pass(surf, ex, C, d_out_fn, which=0, dropout_p=p, dropout_seed=dseed, drop_id=3,
                                        want_grads=want_grads)
            if not want_grads:
                with torch.no_grad():
                    stash["kl"] = kl_divergence_gaussians(q[..., :L], q[..., L:], out[..., :L], out[..., L:])
                return stash["kl"], torch.zeros_like(zm), torch.zeros_like(zl)
            stash["gp_prior"] = gpp
            g_zm, g_zl = torch.zeros_like(zm), torch.zeros_like(zl)
            g_zm[:, :C], g_zl[:, :C] = stash["g_q"][..., :L], stash["g_q"][..., L:]
            return stash["kl"], g_zm, g_zl

        losses, gp, _ = te.loss_and_grads(surf, ex, horizon, self._loss_cfg(), eps=eps, seed=self._draw_seed(),
                                          want_grads=want_grads, dropout_p=p, dropout_seed=dseed, kl_fn=kl_fn)
        if ex is None:
            losses["re_ex_feats"] = torch.zeros(1)
        return losses, gp, stash.get("gp_prior")

    def _accumulate_prior_grads(self, gpp, weight=1.0):
        """Packed-blob gradient of the prior blob -> .grad of the prior network's parameters: conv / LSTM
        one for one, the folded heads unfolded (W_f = W_h W_c, b_f = W_h b_c + b_h)."""
        pte = self.__dict__["_vsv_prior_te"]
        names = [n for n, _ in self._prior_named_params()]
        g = dict(zip(names, pte.param_grads(gpp)))
        pn = self.prior_network
        pairs = []
        for i in range(len(self.config["ctx_surface_hidden"])):
            kind = "dense" if self.config["use_dense_surface"] else "conv"
            conv = getattr(pn.surface_embedding, f"prior_enc_{kind}_{i}")
            pairs += [(conv.weight, g[f"encoder.surface_embedding.enc_{kind}_{i}.weight"]),
                      (conv.bias, g[f"encoder.surface_embedding.enc_{kind}_{i}.bias"])]
        for n, p in pn.interaction.named_parameters():
            pairs.append((p, g["encoder.interaction." + n.replace("prior_interact", "enc_interact")]))
        for n, p in pn.mem.named_parameters():
            pairs.append((p, g[f"encoder.mem.{n}"]))
        if self.config["compress_context"]:
            Wc, bc = pn.compression.weight.detach().float(), pn.compression.bias.detach().float()
            gWc, gbc = torch.zeros_like(Wc), torch.zeros_like(bc)
            for head, name in ((pn.prior_mean_layer, "z_mean_layer"), (pn.prior_logvar_layer, "z_log_var_layer")):
                gWf, gbf = g[f"encoder.{name}.weight"], g[f"encoder.{name}.bias"]
                Wh = head.weight.detach().float()
                pairs += [(head.weight, gWf @ Wc.t() + torch.outer(gbf, bc)), (head.bias, gbf)]
                gWc += Wh.t() @ gWf
                gbc += Wh.t() @ gbf
            pairs += [(pn.compression.weight, gWc), (pn.compression.bias, gbc)]
        else:                                                       # compression is nn.Identity(): heads one for one
            for head, name in ((pn.prior_mean_layer, "z_mean_layer"), (pn.prior_logvar_layer, "z_log_var_layer")):
                pairs += [(head.weight, g[f"encoder.{name}.weight"]), (head.bias, g[f"encoder.{name}.bias"])]
        for p, gr in pairs:
            gr = (gr * weight if weight != 1.0 else gr).to(p.dtype).reshape(p.shape)
            p.grad = gr.clone() if p.grad is None else p.grad + gr

    def _conditional(self):
        return self.use_conditional_prior and self.prior_network is not None

    def test_step(self, x):
        """Reference :460-512."""
        if not self._conditional():
            return super().test_step(x)
        return self._cond_loss_terms(x, self.horizon, False)[0]

    def train_step(self, x, optimizer: torch.optim.Optimizer, scaler=None):
        """Reference :241-318 (float32 here; the reference's CUDA path runs under bf16 autocast)."""
        if not self._conditional():
            return super().train_step(x, optimizer)
        optimizer.zero_grad()
        losses, gp, gpp = self._cond_loss_terms(x, self.horizon, True)
        self._accumulate_grads(gp)
        self._accumulate_prior_grads(gpp)
        torch.nn.utils.clip_grad_norm_(self.parameters(), max_norm=1.0)
        optimizer.step()
        return losses

    def train_step_multihorizon(self, x, optimizer: torch.optim.Optimizer, horizons=[1, 7, 14, 30], scaler=None):
        """Reference :320-458: per horizon reconstruction + kl_weight * KL(q[:, :C] || p(z | context)), fresh
        posterior noise each, the MEAN over horizons back-propagated, clip 1.0.  (The reference's
        cache_encoder_multihorizon only avoids re-running the deterministic posterior encoder; the numbers
        are the same, so both settings take this path.)  Returns loss (tensor), horizon_losses, kl_loss."""
        if not self._conditional():
            return super().train_step_multihorizon(x, optimizer, horizons)
        surface = x["surface"]
        T = surface.shape[0] if surface.dim() == 3 else surface.shape[1]
        optimizer.zero_grad()
        total = None
        horizon_losses, kl = {}, None
        n = len(horizons)
        for h in horizons:
            if T - h < 1:
                continue
            losses, gp, gpp = self._cond_loss_terms(x, h, True)
            self._accumulate_grads(gp, 1.0 / n)
            self._accumulate_prior_grads(gpp, 1.0 / n)
            total = losses["loss"] if total is None else total + losses["loss"]
            horizon_losses[h] = losses["loss"].item()
            kl = losses["kl_loss"]
        total = total / n
        torch.nn.utils.clip_grad_norm_(self.parameters(), max_norm=1.0)
        optimizer.step()
        return {"loss": total, "horizon_losses": horizon_losses, "kl_loss": kl}

    # -- checkpoints (the subclass's own format, reference :514-579) ----------------------------------
    def save_model(self, path: str):
        d = {"model_config": self.config, "encoder_state_dict": self.encoder.state_dict(),
             "decoder_state_dict": self.decoder.state_dict(), "ctx_encoder_state_dict": self.ctx_encoder.state_dict(),
             "use_conditional_prior": bool(self._conditional())}
        if self._conditional():
            d["prior_network_state_dict"] = self.prior_network.state_dict()
        torch.save(d, path)

    def load_weights(self, dict_to_load: dict = None, f: str = None):
        if dict_to_load is None:
            dict_to_load = torch.load(f, weights_only=False)
        if "encoder_state_dict" not in dict_to_load:                     # a base-class checkpoint
            return super().load_weights(dict_to_load=dict_to_load)
        self.encoder.load_state_dict(dict_to_load["encoder_state_dict"])
        self.decoder.load_state_dict(dict_to_load["decoder_state_dict"])
        self.ctx_encoder.load_state_dict(dict_to_load["ctx_encoder_state_dict"])
        if dict_to_load.get("use_conditional_prior", False):
            if "prior_network_state_dict" in dict_to_load:
                if self.prior_network is not None:
                    self.prior_network.load_state_dict(dict_to_load["prior_network_state_dict"])
                    print("✓ Loaded conditional prior network weights")
                else:
                    print("Warning: Checkpoint has prior network but model doesn't. Skipping.")
            else:
                print("Warning: use_conditional_prior=True but no prior_network_state_dict found")
