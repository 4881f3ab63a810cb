"""CVAEMemRand on MI355X: drop-in for the reference class
/root/reference/vae/cvae_with_mem_randomized.py:516 (same constructor/config
contract, same state_dict keys and shapes, same method signatures and return
types), with every forward computation executed by the hand-written gfx950
kernels of libvsv_hip.so through its C ABI (vsv/engine.py).

The torch.nn layers held by the three sub-modules are PARAMETER CONTAINERS: they
give the model the reference's parameter names, shapes, default initialisation and
checkpoint compatibility, but their own forward() is never called.  There is no
CPU fallback: calling a compute method without a GPU raises.

Implemented topology: conv or dense (use_dense_surface) surface embedding, mem_type "lstm" / "gru" / anything
else (nn.RNN), compress_context True or False, interaction_layers None / 1 / 2, ex_feats_hidden=None (any other
value raises the TypeError the reference's check_input raises, :991-994).  Shapes the kernels are not built for
(hidden widths above 128, more than 4 embedding layers) raise NotImplementedError when the engine is built.
"""
from __future__ import annotations

import weakref
from collections import OrderedDict
from typing import Dict, Optional, Tuple, Union

import torch
import torch.nn as nn

from vae.base import BaseDecoder, BaseEncoder, BaseVAE


class QuantileLoss(nn.Module):
    """Pinball loss (reference :9-45): for each quantile q_i,
    mean(max((q_i-1)*e, q_i*e)) with e = target - pred[:, :, i]; weighted mean over
    quantiles.  Buffers are float64 like the reference's, so the result is float64."""

    def __init__(self, quantiles=[0.05, 0.5, 0.95], weights=None):
        super().__init__()
        self.register_buffer("quantiles", torch.tensor(quantiles, dtype=torch.float64))
        w = [1.0] * len(quantiles) if weights is None else weights
        self.register_buffer("weights", torch.tensor(w, dtype=torch.float64))

    def forward(self, preds, target):
        q = self.quantiles.to(preds.device).view(1, 1, -1, 1, 1)
        e = (target.unsqueeze(2) - preds).to(torch.float64)
        per_q = torch.maximum((q - 1) * e, q * e).mean(dim=(0, 1, 3, 4))
        w = self.weights.to(preds.device)
        return torch.sum(per_q * w) / torch.sum(w)


# ---------------------------------------------------------------------------
# parameter containers
# ---------------------------------------------------------------------------

def _conv_stack(prefix: str, channels, padding: int) -> nn.Sequential:
    """{prefix}_conv_i / {prefix}_activation_i / flatten -- the registration names the
    reference's state_dict uses (:96-107, :248-259)."""
    layers = OrderedDict()
    cin = 1
    for i, cout in enumerate(channels):
        layers[f"{prefix}_conv_{i}"] = nn.Conv2d(cin, cout, kernel_size=3, stride=1, padding=padding)
        layers[f"{prefix}_activation_{i}"] = nn.ReLU()
        cin = cout
    layers["flatten"] = nn.Flatten()
    return nn.Sequential(layers)


def _memory(config: dict, input_size: int, hidden_size: int) -> nn.Module:
    kind = {"lstm": nn.LSTM, "gru": nn.GRU}.get(config["mem_type"], nn.RNN)
    return kind(input_size=input_size, hidden_size=hidden_size, num_layers=config["mem_layers"],
                batch_first=True, dropout=config["mem_dropout"])


def _interaction(config: dict, prefix: str, act_prefix: str, width: int) -> nn.Module:
    n = config["interaction_layers"]
    if n is None or n <= 0:
        return nn.Identity()
    layers = OrderedDict()
    for i in range(n):
        layers[f"{prefix}_linear_{i}"] = nn.Linear(width, width)
        layers[f"{act_prefix}_activation_{i}"] = nn.ReLU()
    layers[f"{prefix}_linear_final"] = nn.Linear(width, width)
    return nn.Sequential(layers)


def _dense_stack(prefix: str, sizes, in_feats: int) -> Tuple[nn.Sequential, int]:
    layers = OrderedDict()
    layers["flatten"] = nn.Flatten()
    for i, out in enumerate(sizes):
        layers[f"{prefix}_dense_{i}"] = nn.Linear(in_feats, out)
        layers[f"{prefix}_activation_{i}"] = nn.ReLU()
        in_feats = out
    return nn.Sequential(layers), in_feats


class _HipSubmodule:
    """Mixin: reach the owning CVAEMemRand (and through it the kernel engine)
    without registering it as a child module."""

    def _owner(self) -> "CVAEMemRand":
        ref = self.__dict__.get("_vsv_owner")
        owner = ref() if ref is not None else None
        if owner is None:
            raise RuntimeError("this sub-module is not attached to a CVAEMemRand; the HIP path "
                               "needs the full model (weights are packed per model)")
        return owner


class CVAEMemRandEncoder(BaseEncoder, _HipSubmodule):
    """Posterior encoder (reference :48-200): conv embedding of every day, optional
    ex-features appended, LSTM over time, z_mean / z_log_var heads, z sampled at
    every position."""

    def __init__(self, config: dict):
        super().__init__(config)
        fh, fw = config["feat_dim"]
        hidden = config["surface_hidden"]
        if config["use_dense_surface"]:
            self.surface_embedding, surf_dim = _dense_stack("enc", hidden, fh * fw)
        else:
            self.surface_embedding = _conv_stack("enc", hidden, config["padding"])
            surf_dim = hidden[-1] * fh * fw
        ex_dim = 0
        if config["ex_feats_dim"] > 0:
            if config["ex_feats_hidden"] is not None:
                raise NotImplementedError("ex_feats_hidden must be None (the reference's non-None branch is untested, :121)")
            self.ex_feats_embedding = nn.Identity()
            ex_dim = config["ex_feats_dim"]
        self.interaction = _interaction(config, "enc_interact", "enc_interaction", surf_dim + ex_dim)
        self.mem = _memory(config, surf_dim + ex_dim, config["mem_hidden"])
        self.z_mean_layer = nn.Linear(config["mem_hidden"], config["latent_dim"])
        self.z_log_var_layer = nn.Linear(config["mem_hidden"], config["latent_dim"])

    def forward(self, x):
        """x: {"surface": (B,T,H,W)[, "ex_feats": (B,T,n)]} -> (z_mean, z_log_var, z), each (B,T,L)."""
        return self._owner()._encode_posterior(x)


class CVAECtxMemRandEncoder(BaseEncoder, _HipSubmodule):
    """Context encoder (reference :202-344): same stack, deterministic, followed by
    the final_compression Linear when compress_context is set."""

    def __init__(self, config: dict):
        super().__init__(config)
        fh, fw = config["feat_dim"]
        hidden = config["ctx_surface_hidden"]
        if config["use_dense_surface"]:
            self.ctx_surface_embedding, surf_dim = _dense_stack("ctx_enc", hidden, fh * fw)
        else:
            self.ctx_surface_embedding = _conv_stack("ctx_enc", hidden, config["padding"])
            surf_dim = hidden[-1] * fh * fw
        ex_dim = 0
        if config["ex_feats_dim"] > 0:
            if config["ctx_ex_feats_hidden"] is not None:
                raise NotImplementedError("ctx_ex_feats_hidden must be None")
            self.ctx_ex_feats_embedding = nn.Identity()
            ex_dim = config["ex_feats_dim"]
        self.interaction = _interaction(config, "ctx_interact", "ctx_interaction", surf_dim + ex_dim)
        self.mem = _memory(config, surf_dim + ex_dim, config["mem_hidden"])
        if config["compress_context"]:
            self.final_compression = nn.Linear(config["mem_hidden"], config["latent_dim"])
        else:
            self.final_compression = nn.Identity()

    def forward(self, x):
        """x: {"surface": (B,C,H,W)[, "ex_feats": (B,C,n)]} -> context embedding (B,C,L)."""
        return self._owner()._encode_context(x)


class CVAEMemRandDecoder(BaseDecoder, _HipSubmodule):
    """Decoder (reference :346-514): LSTM on cat([ctx embedding, z]) -> Linear ->
    ConvTranspose+ReLU stack -> Conv to Q quantile channels; ex-features through two
    Linears when ex_feats_dim > 0."""

    def __init__(self, config: dict):
        super().__init__(config)
        fh, fw = config["feat_dim"]
        hidden = config["surface_hidden"]
        L = config["latent_dim"]
        ctx_dim = L if config["compress_context"] else config["mem_hidden"]
        self.surface_final_hidden_size = hidden[-1]
        self.n_info = self.ex_feats_final_hidden_size = config["ex_feats_dim"]
        self.n_surface = hidden[-1] if config["use_dense_surface"] else hidden[-1] * fh * fw
        width = self.n_surface + self.n_info
        self.mem = _memory(config, L + ctx_dim, width)
        self.interaction = _interaction(config, "dec_interact", "dec_interaction", width)
        self.surface_decoder_input = nn.Linear(width, self.n_surface)
        Q = config.get("num_quantiles", 3)
        layers = OrderedDict()
        cin = hidden[-1]
        if config["use_dense_surface"]:
            for i, cout in enumerate(reversed(hidden[:-1])):
                layers[f"dec_dense_{i}"] = nn.Linear(cin, cout)
                layers[f"dec_activation_{i}"] = nn.ReLU()
                cin = cout
            layers["dec_final"] = nn.Linear(cin, fh * fw)
            layers["dec_final_activation"] = nn.ReLU()
            layers["dec_output"] = nn.Linear(fh * fw, fh * fw * Q)
        else:
            pad, opad = config["padding"], config["deconv_output_padding"]
            for i, cout in enumerate(reversed(hidden[:-1])):
                layers[f"dec_deconv_{i}"] = nn.ConvTranspose2d(cin, cout, kernel_size=3, stride=1,
                                                              padding=pad, output_padding=opad)
                layers[f"dec_activation_{i}"] = nn.ReLU()
                cin = cout
            layers["dec_final"] = nn.ConvTranspose2d(cin, cin, kernel_size=3, stride=1,
                                                     padding=pad, output_padding=opad)
            layers["dec_final_activation"] = nn.ReLU()
            layers["dec_output"] = nn.Conv2d(cin, Q, kernel_size=3, padding="same")
        self.surface_decoder = nn.Sequential(layers)
        if self.n_info > 0:
            self.ex_feats_decoder_input = nn.Linear(width, self.n_info)
            self.ex_feats_decoder = nn.Linear(config["ex_feats_dim"], config["ex_feats_dim"])

    def forward(self, x):
        """x: (B,T,ctx_dim+L) = cat([padded ctx embedding, z]) -> (B,T,Q,H,W)[, (B,T,n)]."""
        return self._owner()._decode(x)


# ---------------------------------------------------------------------------
# the model
# ---------------------------------------------------------------------------

class CVAEMemRand(BaseVAE):
    def __init__(self, config: dict):
        super().__init__(config)
        self.check_input(config)
        if not config["use_dense_surface"]:
            # 'same' 3x3 convolution on the grid (reference :545-558): for the 5x5 grid
            # padding = 1 and no output padding
            extra = 3 - 1          # (out-1)*stride + kernel - in, stride 1
            config["padding"] = extra // 2 + extra % 2
            config["deconv_output_padding"] = extra % 2
        self.encoder = CVAEMemRandEncoder(config)
        self.ctx_encoder = CVAECtxMemRandEncoder(config)
        self.decoder = CVAEMemRandDecoder(config)
        self.ex_feats_loss_fn = nn.MSELoss() if config["ex_feats_loss_type"] == "l2" else nn.L1Loss()
        self.quantile_loss_fn = QuantileLoss(quantiles=config["quantiles"],
                                             weights=config.get("quantile_loss_weights"))
        for sub in (self.encoder, self.ctx_encoder, self.decoder):
            sub.__dict__["_vsv_owner"] = weakref.ref(self)
        self.to(self.device)
        self.horizon = config.get("horizon", 1)
        if self.horizon < 1:
            raise ValueError(f"horizon must be >= 1, got {self.horizon}")
        self.__dict__["_vsv_engine"] = None
        self.__dict__["_vsv_key"] = None

    # -- config ------------------------------------------------------------
    def check_input(self, config: dict):
        """Validation and defaulting of the reference (:969-1020); mutates config."""
        for req in ("feat_dim", "latent_dim"):
            if req not in config:
                raise ValueError(f"config doesn't contain: {req}")
        config.setdefault("device", "cuda" if torch.cuda.is_available() else "cpu")
        config.setdefault("kl_weight", 1.0)
        for req in ("surface_hidden", "ex_feats_dim", "mem_hidden", "mem_layers", "ctx_surface_hidden"):
            if req not in config:
                raise ValueError(f"config doesn't contain {req}")
        for key in ("surface_hidden", "ctx_surface_hidden"):
            if isinstance(config[key], int):
                config[key] = [config[key]]
        for key in ("ex_feats_hidden", "ctx_ex_feats_hidden"):
            config.setdefault(key, None)
            if config[key] is not None:
                # the reference calls isinstance() with one argument here (:991/:993)
                raise TypeError("isinstance expected 2 arguments, got 1")
        defaults = {"mem_type": "lstm", "mem_dropout": 0, "re_feat_weight": 1.0, "ex_feats_loss_type": "l1",
                    "ex_loss_on_ret_only": False, "interaction_layers": None, "use_dense_surface": False,
                    "compress_context": False, "quantiles": [0.05, 0.5, 0.95]}
        for k, v in defaults.items():
            config.setdefault(k, v)
        config["num_quantiles"] = len(config["quantiles"])

    # -- engine plumbing -------------------------------------------------------
    def _param_dtype(self):
        return self.encoder.z_mean_layer.weight.dtype

    def _hip_named_params(self):
        # walking the module tree costs ~0.1 ms and a train step asks three times: keep the list (module
        # conversions go through _apply, which drops it; Parameter objects are otherwise never replaced)
        cached = self.__dict__.get("_vsv_named_params")
        if cached is None:
            cached = [(n, p) for n, p in self.named_parameters()]
            self.__dict__["_vsv_named_params"] = cached
        return cached

    def _apply(self, fn, recurse=True):
        self.__dict__.pop("_vsv_named_params", None)
        out = super()._apply(fn, recurse)
        self.__dict__.pop("_vsv_named_params", None)
        return out

    def _engine(self, pack: bool = True):
        """The kernel engine with weights packed from the CURRENT parameters
        (re-packed on device whenever a parameter changed; pack=False leaves the blob alone --
        the graphed training step packs inside its HIP graph)."""
        from vsv.engine import Engine
        dev = torch.device(self.device)
        if dev.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError("CVAEMemRand (MI355X build) computes only on a GPU: config['device'] must be "
                               "a cuda/ROCm device and libvsv_hip.so must be built; there is no CPU fallback")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        eng = self.__dict__["_vsv_engine"]
        if eng is None:
            eng = Engine(self.config, None, dev)
            self.__dict__["_vsv_engine"] = eng
        if not pack:
            return eng
        params = self._hip_named_params()
        key = tuple((p.data_ptr(), p._version) for _, p in params)
        if key != self.__dict__["_vsv_key"]:
            with torch.no_grad():
                eng.set_weights_from_params(params)
            self.__dict__["_vsv_key"] = key
        return eng

    def invalidate_packed_weights(self) -> None:
        """Forces a re-pack of the kernels' weight blob at the next compute call.  The blob is rebuilt whenever a
        parameter's (storage pointer, version counter) changes -- optimizer steps, load_state_dict and .to() all
        bump one of them.  In-place edits through `p.data` (p.data.copy_/mul_/clamp_: EMA swaps, weight clipping,
        manual initialisation) bump NEITHER, so call this after such an edit; otherwise the kernels keep computing
        with the weights packed before it."""
        self.__dict__["_vsv_key"] = None
        if "_vsv_prior_key" in self.__dict__:
            self.__dict__["_vsv_prior_key"] = None

    @staticmethod
    def _draw_seed() -> int:
        """A Philox key drawn from torch's global generator, so set_seeds() makes
        the on-device noise reproducible (the reference uses torch.randn here)."""
        return int(torch.randint(0, 2 ** 62, (1,)).item())

    def _batched(self, x: Dict[str, torch.Tensor]):
        surf = x["surface"].to(self.device)
        if surf.dim() == 3:
            surf = surf.unsqueeze(0)
        ex = None
        if "ex_feats" in x:
            ex = x["ex_feats"].to(self.device)
            if ex.dim() == 2:
                ex = ex.unsqueeze(0)
        return surf, ex

    # -- sub-module entry points (analysis scripts call these directly) --------
    def _encode_posterior(self, x):
        from vsv import paths
        eng = self._engine()
        surf, ex = self._batched(x)
        zm, zl = paths.encode_both(eng, surf, ex)
        eps = eng.philox_normal(self._draw_seed(), 0, zm.shape[0], 1, 0, zm.shape[1])
        z = zm + torch.exp(0.5 * zl) * eps
        dt = self._param_dtype()
        return zm.to(dt), zl.to(dt), z.to(dt)

    def _encode_context(self, x):
        eng = self._engine()
        surf, ex = self._batched(x)
        return eng.encode(1, surf, ex).to(self._param_dtype())

    def _decode(self, dec_in: torch.Tensor):
        eng = self._engine()
        Lc = eng.Lc                               # [context embedding | z], reference :707
        dec_in = dec_in.to(self.device)
        B, T = dec_in.shape[:2]
        r = eng.decode(B, T, 0, ctx=dec_in[..., :Lc].contiguous(), zb=dec_in[..., Lc:].contiguous())
        dt = self._param_dtype()
        if eng.X > 0:
            return r["surf"].to(dt), r["ex"].to(dt)
        return r["surf"].to(dt)

    # -- forward / generation ------------------------------------------------------
    @torch.compiler.disable
    def forward(self, x: Dict[str, torch.Tensor]):
        """Reference :1022-1074.  Returns (recon (B,H,Q,5,5)[, ex (B,H,n)], z_mean, z_log_var, z).
        Under torch.compile (the v3 trainer wraps the model, train_..._v3_conditional_prior.py:298) the whole
        forward is one opaque call: its work is ctypes launches of libvsv_hip.so, nothing for a tracer to fuse."""
        from vsv import paths
        surf, ex = self._batched(x)
        T = surf.shape[1]
        C = T - self.horizon
        if C < 1:
            raise ValueError(f"Insufficient sequence length: T={T}, horizon={self.horizon}, "
                             f"resulting in C={C}. Need at least C >= 1.")
        eng = self._engine()
        dt = self._param_dtype()
        p = float(self.config.get("mem_dropout", 0) or 0) if self.training and self.config["mem_layers"] > 1 else 0.0
        params = self._hip_named_params()
        if (torch.is_grad_enabled() and any(q.requires_grad for _, q in params)) or p > 0:
            # the outputs carry an autograd graph like the reference's: forward with saved activations,
            # the backward kernels run when the caller's loss.backward() reaches them (vsv/autograd.py)
            from vsv.autograd import CVAEForward
            s, e, zm, zl, z = CVAEForward.apply(self._train_engine(), surf, ex, self.horizon, None,
                                                self._draw_seed(), p, self._draw_seed() if p > 0 else 0,
                                                *[q for _, q in params])
        else:
            s, e, zm, zl, z = paths.forward(eng, surf, ex, self.horizon, seed=self._draw_seed())
        if ex is not None:
            return s.to(dt), e.to(dt), zm.to(dt), zl.to(dt), z.to(dt)
        return s.to(dt), zm.to(dt), zl.to(dt), z.to(dt)

    def load_fitted_prior(self, prior_params_path: str):
        """GMM fitted to the aggregated posterior (reference :583-603; file written by
        analysis_code/fit_aggregated_posterior.py:139-155)."""
        params = torch.load(prior_params_path, map_location=self.device)
        fp = {k: params[k].to(self.device).to(torch.float64) for k in ("means", "covariances", "weights")}
        fp["n_components"] = params["n_components"]
        fp["cholesky"] = torch.linalg.cholesky(fp["covariances"])
        self.fitted_prior = fp
        print(f"Loaded fitted prior with {params['n_components']} components from {prior_params_path}")

    def _sample_from_fitted_prior(self, batch_size: int, seq_len: int) -> torch.Tensor:
        """Reference :605-640: component ~ multinomial(weights), z = mean + chol @ eps -- drawn on the device by
        vsv_gmm_sample (Philox keyed by a seed from torch's generator); float64 (B, T, L) like the reference's."""
        fp = getattr(self, "fitted_prior", None)
        if fp is None:
            raise ValueError("Fitted prior not loaded. Call load_fitted_prior() first.")
        n, L = batch_size * seq_len, self.config["latent_dim"]
        z = self._engine().gmm_sample(fp["means"], fp["cholesky"], fp["weights"], n, seed=self._draw_seed())
        return z.to(torch.float64).reshape(batch_size, seq_len, L)

    def get_surface_given_conditions(self, c: Dict[str, torch.Tensor], z: torch.Tensor = None, mu=0, std=1,
                                     horizon=None, prior_mode="standard"):
        """Prior generation (reference :642-715): context positions use the posterior
        mean, future positions z ~ N(mu, std) (or the fitted GMM, or the caller's z,
        whose [:, :C] is overwritten IN PLACE like the reference does at :700).
        Returns (B,horizon,Q,5,5)[, (B,horizon,n)]."""
        from vsv import paths
        surf, ex = self._batched(c)
        B, C = surf.shape[:2]
        if ex is not None:
            assert ex.shape[1] == C, "context length mismatch"
        if horizon is None:
            horizon = self.horizon
        T = C + horizon
        eng = self._engine()
        dt = self._param_dtype()
        if z is None and prior_mode == "fitted":
            z = self._sample_from_fitted_prior(B, T)
        if z is not None:
            if z.dim() == 2:
                z = z.unsqueeze(0)
            zm, _ = paths.encode_both(eng, surf, ex)
            z[:, :C, ...] = zm.to(z.dtype)
            ctx = eng.encode(1, surf, ex)
            s, e = paths.decode_given_z(eng, ctx, z, C)
        else:
            s, e, _ = paths.prior_generate(eng, surf, ex, horizon, 1, mu=float(mu), std=float(std),
                                           seed=self._draw_seed())
        if ex is not None:
            return s.to(dt), e.to(dt)
        return s.to(dt)

    def generate_autoregressive_sequence(self, initial_context: Dict[str, torch.Tensor], horizon: int = 30,
                                         z: Optional[torch.Tensor] = None
                                         ) -> Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
        """Day-by-day rollout (reference :717-807): each step generates one day from the
        prior, the p50 channel (and the predicted ex-features) slide into the context.
        `z` is ignored, as in the reference (:776).  Like the reference's, every step generates self.horizon
        days and all are returned: (B, horizon * self.horizon, Q, 5, 5) (self.horizon is 1 in every config)."""
        if "surface" not in initial_context:
            raise ValueError("initial_context must contain 'surface' key")
        nd = initial_context["surface"].dim()
        if nd not in (3, 4):
            raise ValueError(f"surface must be (C, H, W) or (B, C, H, W), got shape {initial_context['surface'].shape}")
        has_ex = "ex_feats" in initial_context
        if has_ex and initial_context["ex_feats"].dim() not in (2, 3):
            raise ValueError(f"ex_feats must be (C, D) or (B, C, D), got shape {initial_context['ex_feats'].shape}")
        from vsv import paths
        surf, ex = self._batched(initial_context)
        eng = self._engine()
        dt = self._param_dtype()
        if self.horizon == 1:
            s_out, e_out = paths.ar_generate(eng, surf, ex if has_ex else None, horizon, 1, seed=self._draw_seed())
        else:
            # the reference's inner call (:776) generates self.horizon days per step and keeps them all
            s_out, e_out = paths.ar_generate_multi(eng, surf, ex if has_ex else None, horizon, self.horizon,
                                                   seed=self._draw_seed())
        return (s_out.to(dt), e_out.to(dt)) if has_ex else s_out.to(dt)

    def generate_autoregressive_offset(self, initial_context: Dict[str, torch.Tensor], ar_steps: int = 3,
                                       horizon: int = 60, offset: int = 60, z: Optional[torch.Tensor] = None
                                       ) -> Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
        """Chunked rollout (reference :809-923): ar_steps generations of `horizon` days;
        the next context is the last C of the concatenated p50 medians."""
        if "surface" not in initial_context:
            raise ValueError("initial_context must contain 'surface' key")
        C = self.config.get("context_len", 60)
        has_ex = "ex_feats" in initial_context
        if initial_context["surface"].shape[1] != C:
            raise ValueError(f"Context length must be {C}, got {initial_context['surface'].shape[1]}")
        from vsv import paths
        surf, ex = self._batched(initial_context)
        eng = self._engine()
        dt = self._param_dtype()
        s_out, e_out = paths.ar_offset_generate(eng, surf, ex if has_ex else None, ar_steps, horizon,
                                                seed=self._draw_seed())
        return (s_out.to(dt), e_out.to(dt)) if has_ex else s_out.to(dt)

    def _update_context(self, context: Dict[str, torch.Tensor], new_surface: torch.Tensor,
                        new_ex_feat: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """Sliding window: drop the oldest day, append the new one (reference :925-967)."""
        old = context["surface"]
        if old.dim() == 3:
            old = old.unsqueeze(0)
        updated = {"surface": torch.cat([old[:, 1:], new_surface.to(old.device, old.dtype).unsqueeze(1)], dim=1)}
        if "ex_feats" in context and new_ex_feat is not None:
            old_e = context["ex_feats"]
            if old_e.dim() == 2:
                old_e = old_e.unsqueeze(0)
            updated["ex_feats"] = torch.cat([old_e[:, 1:], new_ex_feat.to(old_e.device, old_e.dtype).unsqueeze(1)], dim=1)
        return updated

    # -- losses / steps --------------------------------------------------------------
    def _losses(self, x, outputs, horizon):
        """The loss terms of reference :1113-1125 from forward() outputs."""
        surface = x["surface"]
        if surface.dim() == 3:
            surface = surface.unsqueeze(0)
        C = surface.shape[1] - horizon
        surface_real = surface[:, C:].to(self.device)
        has_ex = "ex_feats" in x
        if has_ex:
            recon, ex_recon, z_mean, z_log_var, _ = outputs
        else:
            recon, z_mean, z_log_var, _ = outputs
        re_surface = self.quantile_loss_fn(recon, surface_real)
        if has_ex:
            ex = x["ex_feats"]
            if ex.dim() == 2:
                ex = ex.unsqueeze(0)
            ex_real = ex[:, C:].to(self.device)
            if self.config["ex_loss_on_ret_only"]:
                ex_recon, ex_real = ex_recon[:, :, :1], ex_real[:, :, :1]
            re_ex = self.ex_feats_loss_fn(ex_recon, ex_real)
            recon_err = re_surface + self.config["re_feat_weight"] * re_ex
        else:
            re_ex = torch.zeros(1)
            recon_err = re_surface
        kl = -0.5 * (1 + z_log_var - torch.exp(z_log_var) - torch.square(z_mean))
        kl = torch.mean(torch.sum(kl, dim=-1))
        return {"loss": recon_err + self.kl_weight * kl, "re_surface": re_surface, "re_ex_feats": re_ex,
                "reconstruction_loss": recon_err, "kl_loss": kl}

    # -- training ------------------------------------------------------------------------
    def _train_engine(self, pack: bool = True):
        """The training engine (forward with saved activations, loss + BPTT kernels,
        parameter-gradient assembly), built once per model."""
        from vsv.train import TrainEngine
        eng = self._engine(pack)
        te = self.__dict__.get("_vsv_train_engine")
        if te is None or te.eng is not eng:
            te = TrainEngine(eng, [(n, tuple(p.shape)) for n, p in self._hip_named_params()])
            self.__dict__["_vsv_train_engine"] = te
        return te

    def _loss_cfg(self):
        # the quantile levels / weights are buffers on the device: read them back once per version,
        # not once per step (each read-back is a blocking copy)
        qf = self.quantile_loss_fn
        key = (qf.quantiles.data_ptr(), qf.quantiles._version, qf.weights.data_ptr(), qf.weights._version)
        cached = self.__dict__.get("_vsv_qcache")
        if cached is None or cached[0] != key:
            cached = (key, qf.quantiles.detach().cpu().tolist(), qf.weights.detach().cpu().tolist())
            self.__dict__["_vsv_qcache"] = cached
        X = self.config["ex_feats_dim"]
        return dict(quantiles=list(cached[1]), quantile_weights=list(cached[2]),
                    ex_n=(1 if self.config["ex_loss_on_ret_only"] else max(X, 1)),
                    ex_l2=int(isinstance(self.ex_feats_loss_fn, nn.MSELoss)),
                    ex_weight=float(self.config["re_feat_weight"]), kl_weight=float(self.kl_weight))

    def _loss_terms(self, x, horizon, want_grads, eps=None):
        """eps (private, parity tests): the posterior noise (B, T, latent_dim) in place of the randn_like of :198."""
        surf, ex = self._batched(x)
        T = surf.shape[1]
        if T - horizon < 1:
            raise ValueError(f"Insufficient sequence length: T={T}, horizon={horizon}, "
                             f"resulting in C={T - horizon}. Need at least C >= 1.")
        # nn.LSTM applies its inter-layer dropout in train mode only (and only with > 1 layer)
        p = float(self.config.get("mem_dropout", 0) or 0) if self.training else 0.0
        if want_grads and self._use_graphs():
            # one HIP-graph replay per step (vsv/graphed.py): weight packing, noise, forward, loss, backward
            from vsv.graphed import GraphCache
            te = self._train_engine(pack=False)
            cache = self.__dict__.setdefault("_vsv_graphs", GraphCache())
            params = self._hip_named_params()
            g = cache.get(te, params, surf, ex, horizon, self._loss_cfg(), p if self.config["mem_layers"] > 1 else 0.0,
                          inject_eps=eps is not None)
            losses, grads = g.run(surf, ex, eps)
            self.__dict__["_vsv_key"] = tuple((q.data_ptr(), q._version) for _, q in params)
            if ex is None:
                losses["re_ex_feats"] = torch.zeros(1)
            return losses, grads
        te = self._train_engine()
        if eps is not None:
            eps = eps.to(surf.device, torch.float32).contiguous()
        losses, gp, _ = te.loss_and_grads(surf, ex, horizon, self._loss_cfg(), eps=eps, seed=self._draw_seed(),
                                          want_grads=want_grads, dropout_p=p,
                                          dropout_seed=self._draw_seed() if p > 0 else 0)
        if ex is None:
            losses["re_ex_feats"] = torch.zeros(1)
        return losses, gp

    def _use_graphs(self) -> bool:
        """HIP-graph replay of the training step (vsv/graphed.py: weight packing, noise, forward, loss, backward
        and the gradient scatter as ONE graph per input shape): on by default, config["hip_graphs"] = False or
        VSV_GRAPHS=0 runs the same launches eagerly.  Eager, a step costs the host 1.65 ms of launch path whatever
        the sequence length, more than the kernels take below T = 10; replayed, the step is bound by the GPU alone
        (bench.py "train": ms_per_step vs ms_per_step_eager).  A new (batch, length) shape is captured on first
        use (tens of ms)."""
        import os
        env = os.environ.get("VSV_GRAPHS")
        if env is not None:
            return env == "1"
        return bool(self.config.get("hip_graphs", True))

    def _accumulate_grads(self, gp, weight=1.0, alias=False):
        """Add weight * gradient to .grad of the parameters.  gp: gradient w.r.t. the packed blob (eager path)
        or the list of parameter gradients the graphed step returned (static graph memory: alias=True lets
        .grad point at it directly -- train_step, one replay per step --, otherwise it is copied out)."""
        te = self.__dict__["_vsv_train_engine"]
        params = [p for _, p in self._hip_named_params()]
        frozen = [not p.requires_grad for p in params]
        graphed = isinstance(gp, (list, tuple))
        grads = gp if graphed else te.param_grads(gp)
        if weight != 1.0:
            grads = torch._foreach_mul(list(grads), weight)
            graphed = False                                  # fresh tensors
        views = weight == 1.0 and (alias or not graphed)        # every .grad set below is a view of te.last_flat
        # a frozen parameter (requires_grad False, user fine-tuning) gets no .grad, as under the reference's autograd:
        # the optimizer then leaves it alone -- and the one-launch clip / AdamW over the flat buffer do not apply
        views = views and not any(frozen)
        for p, g, fz in zip(params, grads, frozen):
            if fz:
                continue
            if g.dtype is not p.dtype:
                g = g.to(p.dtype)
            views = views and p.grad is None and g.dtype is torch.float32
            if p.grad is None:
                # eager-path gradients and scaled copies are fresh tensors; graph outputs are copied out
                # unless aliasing was allowed
                p.grad = g.clone() if (graphed and not alias and g.dtype == torch.float32) else g
            else:
                p.grad = p.grad + g
        return te.last_flat if views else None

    def _clip_flat_(self, flat: torch.Tensor, max_norm: float) -> None:
        """clip_grad_norm_ (reference :1129) on gradients that are all views of one flat buffer: the same total
        2-norm and coefficient in one launch (vsv_clip_grad_norm, from the sums of squares vsv_param_grads left),
        instead of a foreach pass over 56 tensors."""
        te = self.__dict__["_vsv_train_engine"]
        part = te.last_partials
        if part is not None and te.last_flat is flat:
            rc = te.lib.vsv_clip_grad_norm(flat.data_ptr(), flat.numel(), part.data_ptr(), part.numel(), float(max_norm),
                                           None, te._stream())
            if rc != 0:
                raise RuntimeError("vsv_clip_grad_norm: " + te.lib.vsv_last_error().decode())
            return
        total = torch.linalg.vector_norm(flat)
        flat.mul_(torch.clamp(max_norm / (total + 1e-6), max=1.0))

    def _adamw_step_(self, optimizer, flat: torch.Tensor) -> bool:
        """optimizer.step() (reference :1131) as ONE launch (vsv_adamw_step) when the optimizer is a plain
        torch.optim.AdamW over exactly this model's parameters -- what every trainer of the reference builds --
        updating the optimizer's own state tensors and step counts in place, so the optimizer object stays what
        torch's step() would have left (state_dict, schedulers, a later torch step() all see the same state).
        Anything else (other optimizers, amsgrad / maximize / capturable / fused, several parameter groups, hooks,
        the first step, which creates the state) returns False and the caller runs optimizer.step().
        VSV_NATIVE_ADAMW=0 turns it off."""
        import os
        import torch.optim.optimizer as _oo
        if os.environ.get("VSV_NATIVE_ADAMW", "1") == "0" or type(optimizer) is not torch.optim.AdamW:
            return False
        if len(optimizer.param_groups) != 1 or optimizer._optimizer_step_pre_hooks or optimizer._optimizer_step_post_hooks \
                or _oo._global_optimizer_pre_hooks or _oo._global_optimizer_post_hooks:
            return False
        g = optimizer.param_groups[0]
        if g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable") or g.get("fused"):
            return False
        lr, (b1, b2) = g["lr"], g["betas"]
        if not all(isinstance(v, (int, float)) for v in (lr, b1, b2, g["eps"], g["weight_decay"])):
            return False
        params = [p for _, p in self._hip_named_params()]
        for p in params:
            if p.grad is None or not p.requires_grad:
                return False
        gp = g["params"]
        if len(gp) != len(params):
            return False
        for a, b in zip(gp, params):
            if a is not b:
                return False
        st = optimizer.state
        te = self.__dict__["_vsv_train_engine"]
        dev = flat.device
        n = len(params)
        # The per-step host cost of this method used to exceed the step's GPU time at batch 64 (56 parameters x
        # dtype / device / contiguity / step-count checks): what has been validated once is re-checked by IDENTITY --
        # the optimizer's state dicts and their three tensors are the objects validated, and every tensor still
        # lives where the device pointer table says (load_state_dict, .to() and .data swaps all change one of these).
        fast = self.__dict__.get("_vsv_adamw")
        ok = fast is not None and fast["optimizer"] is optimizer and fast["n"] == n and len(st) == n
        if ok:
            for p, sd, m, v, k in zip(params, fast["dicts"], fast["ms"], fast["vs"], fast["steps"]):
                s = st.get(p)
                if s is not sd or s.get("exp_avg") is not m or s.get("exp_avg_sq") is not v or s.get("step") is not k:
                    ok = False
                    break
        if ok:
            ms, vs = fast["ms"], fast["vs"]
            ok = fast["key"] == tuple([t.data_ptr() for t in params] + [t.data_ptr() for t in ms] + [t.data_ptr() for t in vs])
        if ok:
            big = fast["step_block"]
            step = float(big[0])
            ok = step >= 1 and step == int(step) and float(big[n - 1]) == step
        if not ok:
            # full validation; builds the pointer table and moves the n step counters into ONE block of host memory
            # (each state["step"] becomes a 0-d view of it: torch's own step(), state_dict() and LR schedulers see
            # the same values, and one add_ advances them all -- a foreach over 56 host scalars cost 80 us)
            self.__dict__["_vsv_adamw"] = None
            dicts, ms, vs, steps = [], [], [], []
            for p in params:
                s = st.get(p)
                if not s or "exp_avg" not in s:
                    return False
                dicts.append(s); ms.append(s["exp_avg"]); vs.append(s["exp_avg_sq"]); steps.append(s["step"])
            if any(t.dtype != torch.float32 or t.device != dev or not t.is_contiguous() for t in params + ms + vs):
                return False
            if any((not torch.is_tensor(t)) or t.device.type != "cpu" or t.dim() != 0 for t in steps):
                return False
            step = float(steps[0])
            if any(float(t) != step for t in steps[1:]) or step < 1 or step != int(step):
                return False
            if flat.numel() != int(sum(p.numel() for p in params)):
                return False
            big = torch.stack([t.detach().to(torch.float32) for t in steps])
            if steps[0].dtype == torch.float32:
                for i, s in enumerate(dicts):
                    s["step"] = big[i]
                steps = [s["step"] for s in dicts]
            else:
                big = None                                           # (unusual step dtype: leave the counters alone)
            key = tuple([t.data_ptr() for t in params] + [t.data_ptr() for t in ms] + [t.data_ptr() for t in vs])
            offs = [0]
            for p in params:
                offs.append(offs[-1] + p.numel())
            fast = dict(optimizer=optimizer, n=n, dicts=dicts, ms=ms, vs=vs, steps=steps, key=key, step_block=big, total=offs[-1],
                        tab=torch.tensor(key, dtype=torch.int64).to(dev), starts=torch.tensor(offs, dtype=torch.int32).to(dev))
            if big is not None:
                self.__dict__["_vsv_adamw"] = fast
        if flat.numel() != fast["total"]:
            return False
        base = fast["tab"].data_ptr()
        rc = te.lib.vsv_adamw_step(base, base + 8 * n, base + 16 * n, flat.data_ptr(), fast["starts"].data_ptr(), n, flat.numel(),
                                   float(lr), float(b1), float(b2), float(g["eps"]), float(g["weight_decay"]),
                                   int(step) + 1, te._stream())
        if rc != 0:
            raise RuntimeError("vsv_adamw_step: " + te.lib.vsv_last_error().decode())
        if fast["step_block"] is not None:
            fast["step_block"].add_(1)
        else:
            torch._foreach_add_(fast["steps"], 1)
        # the tensors were written through their addresses: bump the version counters an in-place torch op would
        # have bumped (the engine's re-pack test and autograd's saved-tensor checks read the parameters')
        touched = params + fast["ms"] + fast["vs"]
        try:
            torch._C._autograd._unsafe_set_version_counter(touched, [t._version + 1 for t in touched])
        except (AttributeError, TypeError, RuntimeError):
            pass
        self.__dict__["_vsv_key"] = None
        if hasattr(optimizer, "_opt_called"):
            optimizer._opt_called = True                           # (what LRScheduler's wrapper of step() records)
        return True

    def test_step(self, x, _eps=None):
        """Reference :1265-1308: forward + losses, returns a dict of tensors.  _eps (private, like
        train_step_multihorizon's _eps_per_horizon): injected posterior noise for parity tests."""
        return self._loss_terms(x, self.horizon, False, eps=_eps)[0]

    def train_step(self, x, optimizer: torch.optim.Optimizer, _eps=None):
        """Reference :1076-1137: zero_grad, forward, pinball + ex-feature + KL loss, backward,
        clip_grad_norm_(1.0), optimizer.step().  Computed in float32 (the reference's CUDA path
        runs this under bf16 autocast; its CPU path, which the parity goldens pin, in float32).
        Returns a dict of tensors."""
        optimizer.zero_grad()
        losses, gp = self._loss_terms(x, self.horizon, True, eps=_eps)
        flat = self._accumulate_grads(gp, alias=True)
        if flat is not None:
            self._clip_flat_(flat, 1.0)
        else:
            torch.nn.utils.clip_grad_norm_(self.parameters(), max_norm=1.0)
        if flat is None or not self._adamw_step_(optimizer, flat):
            optimizer.step()
        return losses

    def train_step_multihorizon(self, x, optimizer: torch.optim.Optimizer, horizons=[1, 7, 14, 30],
                                _eps_per_horizon=None):
        """Reference :1139-1263: a forward per horizon (fresh noise each), losses weighted
        {1: 1.0, 7: 0.8, 14: 0.6, 30: 0.4} (default 1.0), one backward, no clipping.  Returns floats.
        The horizons run as row groups of ONE forward/backward (TrainEngine.loss_and_grads_multihorizon);
        _eps_per_horizon (private, like vae.utils.train_autoregressive_step's _eps_steps): a sequence of
        (B, T, latent_dim) tensors, the posterior noise of each horizon's forward, for parity tests."""
        weights = {1: 1.0, 7: 0.8, 14: 0.6, 30: 0.4}
        surf, ex = self._batched(x)
        T = surf.shape[1]
        max_horizon = max(horizons)
        if T < 1 + max_horizon:
            raise ValueError(f"Insufficient sequence length: T={T}, max_horizon={max_horizon}, "
                             f"need at least T >= {1 + max_horizon}")
        optimizer.zero_grad()
        p = float(self.config.get("mem_dropout", 0) or 0) if self.training else 0.0
        hw = [weights.get(h, 1.0) for h in horizons]
        if self._use_graphs():
            # one HIP-graph replay per step, as in train_step (vsv/graphed.py): launched eagerly the step's ~70
            # launches and the dozen small torch ops that mix the loss terms leave the GPU waiting for the host
            from vsv.graphed import GraphCache
            te = self._train_engine(pack=False)
            cache = self.__dict__.setdefault("_vsv_graphs", GraphCache())
            params = self._hip_named_params()
            g = cache.get(te, params, surf, ex, 0, self._loss_cfg(), p if self.config["mem_layers"] > 1 else 0.0,
                          inject_eps=_eps_per_horizon is not None, horizons=list(horizons), weights=hw)
            eps = None if _eps_per_horizon is None else torch.stack([e.to(surf.device, torch.float32) for e in _eps_per_horizon])
            losses, gp = g.run(surf, ex, eps)
            self.__dict__["_vsv_key"] = tuple((q.data_ptr(), q._version) for _, q in params)
            flat = self._accumulate_grads(gp, alias=True)
        else:
            te = self._train_engine()
            losses, gp = te.loss_and_grads_multihorizon(
                surf, ex, list(horizons), hw, self._loss_cfg(),
                eps=_eps_per_horizon, seed=self._draw_seed(), dropout_p=p, dropout_seed=self._draw_seed() if p > 0 else 0)
            flat = self._accumulate_grads(gp)
        if flat is None or not self._adamw_step_(optimizer, flat):
            optimizer.step()
        # one read-back for all the reported numbers
        vals = torch.cat([torch.stack([losses["loss"], losses["reconstruction_loss"], losses["kl_loss"]]),
                          losses["horizon_losses"]]).tolist()
        return {"loss": vals[0], "reconstruction_loss": vals[1], "kl_loss": vals[2],
                "horizon_losses": {h: v for h, v in zip(horizons, vals[3:])}}
