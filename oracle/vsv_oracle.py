"""CPU restatement (numpy) of the reference's CVAEMemRand hot path.

TEST INFRASTRUCTURE ONLY -- never imported by the product path (see
oracle/__init__.py).  Written from the equations, not from the reference's
code: the reference delegates all arithmetic to torch.nn ops
(nn.Conv2d / nn.ConvTranspose2d / nn.LSTM / nn.Linear, torch>=2.9.0,
pyproject.toml:18), so each function below restates the published algorithm of
the op and cites the reference call site it stands for.  All citations are
into /root/reference/.

Pinning: ``oracle/make_goldens.py`` imports the reference in the build
container, loads the weights produced by :func:`init_weights`, injects the
noise and stores the reference's outputs under ``tests/golden/``;
``tests/test_oracle_vs_golden.py`` checks every function here against them.

Everything is dtype-generic: pass float32 or float64 arrays.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

Weights = Dict[str, np.ndarray]


# --------------------------------------------------------------------------
# configuration + deterministic weights: shared with the benchmarks (vsv/synth.py holds no model arithmetic)
# --------------------------------------------------------------------------

from vsv.synth import make_config, state_dict_spec, init_weights  # noqa: E402,F401



def prior_state_dict_spec(cfg: dict) -> List[Tuple[str, Tuple[int, ...]]]:
    """Parameter names and shapes of ConditionalPriorNetwork.state_dict() (keys carry the
    "prior_network." prefix they have inside CVAEMemRandConditionalPrior): the context-encoder stack
    under its own names, a compression Linear and the two prior heads
    (/root/reference/vae/conditional_prior_network.py:62-82, 104-112, 159-166); without compress_context the
    compression is nn.Identity() and the heads read the memory's hidden state (:73-78)."""
    L, X, Hm, nl = cfg["latent_dim"], cfg["ex_feats_dim"], cfg["mem_hidden"], cfg["mem_layers"]
    csh = list(cfg["ctx_surface_hidden"])
    dense = bool(cfg.get("use_dense_surface", False))
    E = csh[-1] * (1 if dense else 25) + X
    spec: List[Tuple[str, Tuple[int, ...]]] = []
    cin = 25 if dense else 1
    for i, co in enumerate(csh):
        if dense:                                           # prior_enc_dense_i, conditional_prior_network.py:90-99
            spec.append((f"prior_network.surface_embedding.prior_enc_dense_{i}.weight", (co, cin)))
            spec.append((f"prior_network.surface_embedding.prior_enc_dense_{i}.bias", (co,)))
        else:
            spec.append((f"prior_network.surface_embedding.prior_enc_conv_{i}.weight", (co, cin, 3, 3)))
            spec.append((f"prior_network.surface_embedding.prior_enc_conv_{i}.bias", (co,)))
        cin = co
    n_int = cfg.get("interaction_layers") or 0
    if n_int > 0:
        for nm in [f"prior_interact_linear_{i}" for i in range(n_int)] + ["prior_interact_linear_final"]:
            spec += [(f"prior_network.interaction.{nm}.weight", (E, E)), (f"prior_network.interaction.{nm}.bias", (E,))]
    ng = {"lstm": 4, "gru": 3}.get(cfg.get("mem_type", "lstm"), 1)
    for l in range(nl):
        i = E if l == 0 else Hm
        spec += [(f"prior_network.mem.weight_ih_l{l}", (ng * Hm, i)), (f"prior_network.mem.weight_hh_l{l}", (ng * Hm, Hm)),
                 (f"prior_network.mem.bias_ih_l{l}", (ng * Hm,)), (f"prior_network.mem.bias_hh_l{l}", (ng * Hm,))]
    hid = Hm
    if cfg.get("compress_context", False):
        spec += [("prior_network.compression.weight", (L, Hm)), ("prior_network.compression.bias", (L,))]
        hid = L
    spec += [("prior_network.prior_mean_layer.weight", (L, hid)), ("prior_network.prior_mean_layer.bias", (L,)),
             ("prior_network.prior_logvar_layer.weight", (L, hid)), ("prior_network.prior_logvar_layer.bias", (L,))]
    return spec


def init_prior_weights(cfg: dict, seed: int = 0, scale: float = 1.0, dtype=np.float32) -> Weights:
    """Deterministic prior-network weights, same recipe as init_weights (own PCG64 stream)."""
    rng = np.random.Generator(np.random.PCG64(seed + 777_000))
    spec = prior_state_dict_spec(cfg)
    shapes = dict(spec)
    W: Weights = {}
    for name, shape in spec:
        if ".mem." in name:
            fan = shapes[name.rsplit(".", 1)[0] + ".weight_hh_l" + name[-1]][1]      # hidden size
        else:
            ws = shape if name.endswith(".weight") else shapes[name[:-4] + "weight"]
            fan = ws[1] * (9 if len(ws) == 4 else 1)
        bound = scale / math.sqrt(fan)
        W[name] = rng.uniform(-bound, bound, size=shape).astype(dtype)
    return W


# --------------------------------------------------------------------------
# primitive ops (the torch.nn ops the reference calls)
# --------------------------------------------------------------------------

def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def conv2d_3x3_same(x: np.ndarray, w: np.ndarray, b: np.ndarray) -> np.ndarray:
    """nn.Conv2d(k=3, stride=1, padding=1) cross-correlation.
    x (N,Cin,H,W), w (Cout,Cin,3,3), b (Cout).  Call sites:
    cvae_with_mem_randomized.py:100-103, 252-255, 428-431.
    out[n,o,y,x] = b[o] + sum_{c,ky,kx} xpad[n,c,y+ky,x+kx] * w[o,c,ky,kx],
    evaluated as one matrix product over the (c,ky,kx) patch axis."""
    N, Cin, H, Wd = x.shape
    xp = np.zeros((N, Cin, H + 2, Wd + 2), dtype=x.dtype)
    xp[:, :, 1:-1, 1:-1] = x
    win = np.lib.stride_tricks.sliding_window_view(xp, (3, 3), axis=(2, 3))   # (N,Cin,H,W,3,3)
    patches = win.transpose(0, 2, 3, 1, 4, 5).reshape(N * H * Wd, Cin * 9)
    out = patches @ w.reshape(w.shape[0], Cin * 9).T + b                      # (N*H*W, Cout)
    return np.ascontiguousarray(out.reshape(N, H, Wd, w.shape[0]).transpose(0, 3, 1, 2))


def conv_transpose2d_3x3(x: np.ndarray, w: np.ndarray, b: np.ndarray) -> np.ndarray:
    """nn.ConvTranspose2d(k=3, stride=1, padding=1, output_padding=0).
    x (N,Cin,H,W), w (Cin,Cout,3,3).  Scatter form: every input pixel adds
    in[yi,xi]*w[:, :, ky, kx] to out[yi-1+ky, xi-1+kx].  Call sites:
    cvae_with_mem_randomized.py:413-424."""
    N, Cin, H, Wd = x.shape
    Cout = w.shape[1]
    contrib = x.transpose(0, 2, 3, 1).reshape(N * H * Wd, Cin) @ w.reshape(Cin, Cout * 9)
    contrib = contrib.reshape(N, H, Wd, Cout, 3, 3)
    full = np.zeros((N, Cout, H + 2, Wd + 2), dtype=x.dtype)
    for ky in range(3):
        for kx in range(3):
            full[:, :, ky:ky + H, kx:kx + Wd] += contrib[:, :, :, :, ky, kx].transpose(0, 3, 1, 2)
    return full[:, :, 1:-1, 1:-1] + b[None, :, None, None]


def lstm_forward(x: np.ndarray, W: Weights, prefix: str, n_layers: int,
                 state: Optional[List[Tuple[np.ndarray, np.ndarray]]] = None,
                 return_cache: bool = False):
    """nn.LSTM(batch_first=True), eval mode (no inter-layer dropout).
    Gate row blocks [i; f; g; o]; c' = f*c + i*g; h' = o*tanh(c').
    x (B,T,I) -> (B,T,H), final [(h,c)] per layer.  Call sites:
    cvae_with_mem_randomized.py:155, 193, 307, 342, 476, 491.
    mem_type "gru" (weights with 3H rows, :156-157): nn.GRU, row blocks [r; z; n],
    n = tanh(W_in x + b_in + r * (W_hn h + b_hn)), h' = (1 - z) * n + z * h; the returned state is (h, h)."""
    B, T, _ = x.shape
    inp = x
    finals = []
    caches = []
    for l in range(n_layers):
        w_ih = W[f"{prefix}.weight_ih_l{l}"]
        w_hh = W[f"{prefix}.weight_hh_l{l}"]
        Hh = w_hh.shape[1]
        if w_hh.shape[0] == Hh:
            # nn.RNN (tanh), any mem_type other than "lstm" / "gru" (:158-159): h' = tanh(W_ih x + b_ih + W_hh h + b_hh)
            assert not return_cache
            h = np.zeros((B, Hh), dtype=x.dtype) if state is None else np.broadcast_to(state[l][0], (B, Hh)).copy()
            xproj = inp @ w_ih.T + W[f"{prefix}.bias_ih_l{l}"] + W[f"{prefix}.bias_hh_l{l}"]
            out = np.empty((B, T, Hh), dtype=x.dtype)
            for t in range(T):
                h = np.tanh(xproj[:, t] + h @ w_hh.T)
                out[:, t] = h
            finals.append((h, h))
            inp = out
            continue
        if w_hh.shape[0] == 3 * Hh:
            assert not return_cache
            b_ih, b_hh = W[f"{prefix}.bias_ih_l{l}"], W[f"{prefix}.bias_hh_l{l}"]
            h = np.zeros((B, Hh), dtype=x.dtype) if state is None else np.broadcast_to(state[l][0], (B, Hh)).copy()
            xproj = inp @ w_ih.T + b_ih
            out = np.empty((B, T, Hh), dtype=x.dtype)
            for t in range(T):
                hp = h @ w_hh.T + b_hh
                r = _sigmoid(xproj[:, t, :Hh] + hp[:, :Hh])
                z = _sigmoid(xproj[:, t, Hh:2 * Hh] + hp[:, Hh:2 * Hh])
                n = np.tanh(xproj[:, t, 2 * Hh:] + r * hp[:, 2 * Hh:])
                h = (1 - z) * n + z * h
                out[:, t] = h
            finals.append((h, h))
            inp = out
            continue
        bias = W[f"{prefix}.bias_ih_l{l}"] + W[f"{prefix}.bias_hh_l{l}"]
        if state is None:
            h = np.zeros((B, Hh), dtype=x.dtype)
            c = np.zeros((B, Hh), dtype=x.dtype)
        else:
            h, c = state[l]
            h = np.broadcast_to(h, (B, Hh)).copy()
            c = np.broadcast_to(c, (B, Hh)).copy()
        xproj = inp @ w_ih.T + bias                      # (B,T,4H)
        out = np.empty((B, T, Hh), dtype=x.dtype)
        cache = {"i": [], "f": [], "g": [], "o": [], "c": [], "h_prev": [], "c_prev": []}
        for t in range(T):
            gates = xproj[:, t] + h @ w_hh.T
            i = _sigmoid(gates[:, 0 * Hh:1 * Hh])
            f = _sigmoid(gates[:, 1 * Hh:2 * Hh])
            g = np.tanh(gates[:, 2 * Hh:3 * Hh])
            o = _sigmoid(gates[:, 3 * Hh:4 * Hh])
            if return_cache:
                cache["h_prev"].append(h)
                cache["c_prev"].append(c)
            c = f * c + i * g
            h = o * np.tanh(c)
            out[:, t] = h
            if return_cache:
                for k, v in (("i", i), ("f", f), ("g", g), ("o", o), ("c", c)):
                    cache[k].append(v)
        finals.append((h, c))
        if return_cache:
            cache["x"] = inp
            caches.append(cache)
        inp = out
    if return_cache:
        return inp, finals, caches
    return inp, finals


def linear(x, w, b):
    return x @ w.T + b


# --------------------------------------------------------------------------
# the three modules
# --------------------------------------------------------------------------

def _embed(surface: np.ndarray, W: Weights, prefix: str, names: Sequence[str]) -> np.ndarray:
    """Conv2d+ReLU stack then Flatten: (B,T,5,5) -> (B,T,C*25), index
    ch*25+row*5+col (cvae_with_mem_randomized.py:96-110, 176-178)."""
    B, T = surface.shape[:2]
    if f"{prefix}.{names[0]}.weight" not in W:
        # use_dense_surface: Flatten, then Linear+ReLU layers named *_dense_i (:84-93, :237-246)
        x = surface.reshape(B * T, 25)
        for nm in names:
            nm = nm.replace("_conv_", "_dense_")
            x = np.maximum(linear(x, W[f"{prefix}.{nm}.weight"], W[f"{prefix}.{nm}.bias"]), 0)
        return x.reshape(B, T, -1)
    x = surface.reshape(B * T, 1, 5, 5)
    for nm in names:
        x = np.maximum(conv2d_3x3_same(x, W[f"{prefix}.{nm}.weight"], W[f"{prefix}.{nm}.bias"]), 0)
    return x.reshape(B, T, -1)


def _interact(x: np.ndarray, W: Weights, cfg: dict, prefix: str, stem: str) -> np.ndarray:
    """The `interaction` Sequential (cvae_with_mem_randomized.py:131-142 and twins): n x (Linear, ReLU), Linear;
    Identity when interaction_layers is None / 0."""
    n = cfg.get("interaction_layers") or 0
    if n <= 0:
        return x
    for i in range(n):
        x = np.maximum(linear(x, W[f"{prefix}.interaction.{stem}_interact_linear_{i}.weight"],
                              W[f"{prefix}.interaction.{stem}_interact_linear_{i}.bias"]), 0)
    return linear(x, W[f"{prefix}.interaction.{stem}_interact_linear_final.weight"],
                  W[f"{prefix}.interaction.{stem}_interact_linear_final.bias"])


def encoder_forward(W: Weights, cfg: dict, surface: np.ndarray,
                    ex_feats: Optional[np.ndarray] = None):
    """CVAEMemRandEncoder.forward without the sampling line
    (cvae_with_mem_randomized.py:162-197) -> z_mean, z_log_var (B,T,L)."""
    n = len(cfg["surface_hidden"])
    emb = _embed(surface, W, "encoder.surface_embedding", [f"enc_conv_{i}" for i in range(n)])
    if ex_feats is not None:
        emb = np.concatenate([emb, ex_feats], axis=-1)          # :188
    emb = _interact(emb, W, cfg, "encoder", "enc")              # :190
    h, _ = lstm_forward(emb, W, "encoder.mem", cfg["mem_layers"])
    zm = linear(h, W["encoder.z_mean_layer.weight"], W["encoder.z_mean_layer.bias"])
    zl = linear(h, W["encoder.z_log_var_layer.weight"], W["encoder.z_log_var_layer.bias"])
    return zm, zl


def ctx_encoder_forward(W: Weights, cfg: dict, surface: np.ndarray,
                        ex_feats: Optional[np.ndarray] = None) -> np.ndarray:
    """CVAECtxMemRandEncoder.forward (cvae_with_mem_randomized.py:314-344),
    -> (B,C,Lc): Lc = latent_dim with compress_context, else the memory's hidden state itself
    (final_compression is nn.Identity(), :225-228)."""
    n = len(cfg["ctx_surface_hidden"])
    emb = _embed(surface, W, "ctx_encoder.ctx_surface_embedding",
                 [f"ctx_enc_conv_{i}" for i in range(n)])
    if ex_feats is not None:
        emb = np.concatenate([emb, ex_feats], axis=-1)
    emb = _interact(emb, W, cfg, "ctx_encoder", "ctx")
    h, _ = lstm_forward(emb, W, "ctx_encoder.mem", cfg["mem_layers"])
    if not cfg.get("compress_context", False):
        return h
    return linear(h, W["ctx_encoder.final_compression.weight"],
                  W["ctx_encoder.final_compression.bias"])


def prior_network_forward(Wp: Weights, cfg: dict, surface: np.ndarray, ex_feats: Optional[np.ndarray] = None):
    """ConditionalPriorNetwork.forward (/root/reference/vae/conditional_prior_network.py:168-209):
    conv embedding [+ ex_feats] -> LSTM -> compression -> prior_mean / prior_logvar, each (B,C,L)."""
    n = len(cfg["ctx_surface_hidden"])
    emb = _embed(surface, Wp, "prior_network.surface_embedding", [f"prior_enc_conv_{i}" for i in range(n)])
    if ex_feats is not None:
        emb = np.concatenate([emb, ex_feats], axis=-1)
    emb = _interact(emb, Wp, cfg, "prior_network", "prior")
    h, _ = lstm_forward(emb, Wp, "prior_network.mem", cfg["mem_layers"])
    if cfg.get("compress_context", False):
        h = linear(h, Wp["prior_network.compression.weight"], Wp["prior_network.compression.bias"])
    return (linear(h, Wp["prior_network.prior_mean_layer.weight"], Wp["prior_network.prior_mean_layer.bias"]),
            linear(h, Wp["prior_network.prior_logvar_layer.weight"], Wp["prior_network.prior_logvar_layer.bias"]))


def kl_divergence_gaussians(mu_q, logvar_q, mu_p, logvar_p) -> float:
    """KL(q || p) of diagonal Gaussians, summed over the latent dims, mean over batch and time
    (conditional_prior_network.py:212-233)."""
    kl = 0.5 * (logvar_p - logvar_q + np.exp(logvar_q - logvar_p) + (mu_q - mu_p) ** 2 / np.exp(logvar_p) - 1)
    return float(kl.sum(axis=-1).mean())


def cond_prior_generate(W: Weights, Wp: Weights, cfg: dict, ctx_surface: np.ndarray, ctx_ex: Optional[np.ndarray],
                        eps_future: np.ndarray, eps_ctx: Optional[np.ndarray] = None):
    """CVAEMemRandConditionalPrior.get_surface_given_conditions with z=None
    (/root/reference/vae/cvae_conditional_prior.py:166-239): future latents
    prior_mean[:, -1] + exp(0.5 * prior_logvar[:, -1]) * eps_future (B,H,L) (iid or AR(1) noise, the
    caller's choice), context latents = posterior mean, or sampled with eps_ctx (B,C,L) when given
    (sample_context=True); decode and return the future part."""
    B, C = ctx_surface.shape[:2]
    H = eps_future.shape[1]
    L = cfg["latent_dim"]
    pm, plv = prior_network_forward(Wp, cfg, ctx_surface, ctx_ex)
    zm, zl = encoder_forward(W, cfg, ctx_surface, ctx_ex)
    z = np.zeros((B, C + H, L), dtype=ctx_surface.dtype)
    z[:, C:] = pm[:, -1:] + np.exp(0.5 * plv[:, -1:]) * eps_future
    z[:, :C] = zm if eps_ctx is None else zm + np.exp(0.5 * zl) * eps_ctx
    ctx = ctx_encoder_forward(W, cfg, ctx_surface, ctx_ex)
    ctx_pad = np.zeros((B, C + H, ctx.shape[-1]), dtype=ctx_surface.dtype)
    ctx_pad[:, :C] = ctx
    s, e, _ = decoder_forward(W, cfg, np.concatenate([ctx_pad, z], axis=-1))
    return s[:, C:], (None if e is None else e[:, C:])


def ar1_noise(innov: np.ndarray, phi: float) -> np.ndarray:
    """The AR(1) latent noise of cvae_conditional_prior.py:180-188 from its innovations (B,H,L):
    eps_0 = innov_0, eps_t = phi * eps_{t-1} + sqrt(1 - phi^2) * innov_t."""
    out = np.array(innov, copy=True)
    for t in range(1, out.shape[1]):
        out[:, t] = phi * out[:, t - 1] + math.sqrt(1 - phi ** 2) * innov[:, t]
    return out


def decoder_head(W: Weights, cfg: dict, h: np.ndarray):
    """Everything after the decoder LSTM (cvae_with_mem_randomized.py:494-514):
    Linear -> reshape (.,5,5,5) -> ConvT+ReLU x3 -> Conv -> (B,T,Q,5,5)
    and, if X>0, Linear -> Linear -> (B,T,X)."""
    B, T = h.shape[:2]
    sh = cfg["surface_hidden"]
    Q = len(cfg["quantiles"])
    h = _interact(h, W, cfg, "decoder", "dec")                  # :492
    s = linear(h, W["decoder.surface_decoder_input.weight"], W["decoder.surface_decoder_input.bias"])
    if cfg.get("use_dense_surface", False):
        # Linear+ReLU stack, Linear(-> 25)+ReLU, Linear(25 -> 25 Q) (:394-407, :495-499)
        s = s.reshape(B * T, sh[-1])
        for nm in [f"dec_dense_{i}" for i in range(len(sh) - 1)] + ["dec_final"]:
            p = f"decoder.surface_decoder.{nm}"
            s = np.maximum(linear(s, W[p + ".weight"], W[p + ".bias"]), 0)
        p = "decoder.surface_decoder.dec_output"
        s = linear(s, W[p + ".weight"], W[p + ".bias"]).reshape(B, T, Q, 5, 5)
        if cfg["ex_feats_dim"] > 0:
            e = linear(h, W["decoder.ex_feats_decoder_input.weight"], W["decoder.ex_feats_decoder_input.bias"])
            e = linear(e, W["decoder.ex_feats_decoder.weight"], W["decoder.ex_feats_decoder.bias"])
            return s, e
        return s, None
    s = s.reshape(B * T, sh[-1], 5, 5)
    for i in range(len(sh) - 1):
        p = f"decoder.surface_decoder.dec_deconv_{i}"
        s = np.maximum(conv_transpose2d_3x3(s, W[p + ".weight"], W[p + ".bias"]), 0)
    p = "decoder.surface_decoder.dec_final"
    s = np.maximum(conv_transpose2d_3x3(s, W[p + ".weight"], W[p + ".bias"]), 0)
    p = "decoder.surface_decoder.dec_output"
    s = conv2d_3x3_same(s, W[p + ".weight"], W[p + ".bias"]).reshape(B, T, Q, 5, 5)
    if cfg["ex_feats_dim"] > 0:
        e = linear(h, W["decoder.ex_feats_decoder_input.weight"], W["decoder.ex_feats_decoder_input.bias"])
        e = linear(e, W["decoder.ex_feats_decoder.weight"], W["decoder.ex_feats_decoder.bias"])
        return s, e
    return s, None


def decoder_forward(W: Weights, cfg: dict, dec_in: np.ndarray, state=None):
    """CVAEMemRandDecoder.forward (cvae_with_mem_randomized.py:482-514)."""
    h, finals = lstm_forward(dec_in, W, "decoder.mem", cfg["mem_layers"], state)
    s, e = decoder_head(W, cfg, h)
    return s, e, finals


# --------------------------------------------------------------------------
# model-level paths
# --------------------------------------------------------------------------

def forward(W: Weights, cfg: dict, surface: np.ndarray, ex_feats: Optional[np.ndarray],
            eps: np.ndarray, horizon: int):
    """CVAEMemRand.forward (cvae_with_mem_randomized.py:1022-1074) with the
    randn_like of :198 replaced by the injected ``eps`` (B,T,L).
    Returns recon (B,H,Q,5,5), ex (B,H,X)|None, z_mean, z_log_var, z."""
    B, T = surface.shape[:2]
    C = T - horizon
    if C < 1:
        raise ValueError("Insufficient sequence length")             # :1043
    L = cfg["latent_dim"]
    ctx = ctx_encoder_forward(W, cfg, surface[:, :C], None if ex_feats is None else ex_feats[:, :C])
    ctx_pad = np.zeros((B, T, ctx.shape[-1]), dtype=surface.dtype)               # :1061-1062
    ctx_pad[:, :C] = ctx
    zm, zl = encoder_forward(W, cfg, surface, ex_feats)
    z = zm + np.exp(0.5 * zl) * eps                                   # :199
    s, e, _ = decoder_forward(W, cfg, np.concatenate([ctx_pad, z], axis=-1))   # :1066
    return s[:, C:], (None if e is None else e[:, C:]), zm, zl, z


def get_surface_given_conditions(W: Weights, cfg: dict, ctx_surface: np.ndarray,
                                 ctx_ex: Optional[np.ndarray], z: np.ndarray, horizon: int):
    """CVAEMemRand.get_surface_given_conditions, literal shape
    (cvae_with_mem_randomized.py:642-715): z (B,T,L) is MODIFIED IN PLACE
    (z[:, :C] = encoder z_mean, :700); decodes all T steps, returns [C:]."""
    B, C = ctx_surface.shape[:2]
    T = C + horizon
    L = cfg["latent_dim"]
    zm, _ = encoder_forward(W, cfg, ctx_surface, ctx_ex)
    z[:, :C] = zm
    ctx = ctx_encoder_forward(W, cfg, ctx_surface, ctx_ex)
    ctx_pad = np.zeros((B, T, ctx.shape[-1]), dtype=ctx_surface.dtype)
    ctx_pad[:, :C] = ctx
    s, e, _ = decoder_forward(W, cfg, np.concatenate([ctx_pad, z.astype(ctx_surface.dtype)], axis=-1))
    return s[:, C:], (None if e is None else e[:, C:])


def gmm_sample(means: np.ndarray, covariances: np.ndarray, comp: np.ndarray, eps: np.ndarray) -> np.ndarray:
    """CVAEMemRand._sample_from_fitted_prior (cvae_with_mem_randomized.py:605-640) given the component picks
    (torch.multinomial, :627-630) and the standard normal draws (:637): z = means[c] + chol[c] @ eps, chol the
    lower Cholesky factor (:602).  means (K,L), covariances (K,L,L), comp (n,), eps (n,L) -> (n,L)."""
    chol = np.linalg.cholesky(covariances)
    return means[comp] + np.einsum("bij,bj->bi", chol[comp], eps)


def generate_surfaces_multiday_mle(W: Weights, cfg: dict, vol: np.ndarray, ex: Optional[np.ndarray], start_day: int,
                                   n_days: int, ctx_len: int):
    """generate_surfaces_max_likelihood.py:13-103: for every day one z = 0 generation from the window
    [day - ctx_len, day) (z of shape (1, ctx_len + 1, L): horizon 1; the context positions are overwritten by
    the posterior mean, :700).  -> (p05, p50, p95) each (n_days, 5, 5), ex_feats (n_days, X) or None."""
    L, X = cfg["latent_dim"], cfg["ex_feats_dim"]
    win = np.stack([vol[d - ctx_len:d] for d in range(start_day, start_day + n_days)])
    wex = np.stack([ex[d - ctx_len:d] for d in range(start_day, start_day + n_days)]) if X > 0 else None
    z = np.zeros((n_days, ctx_len + 1, L), dtype=vol.dtype)
    s, e = get_surface_given_conditions(W, cfg, win, wex, z, 1)
    return (s[:, 0, 0], s[:, 0, 1], s[:, 0, 2]), (e[:, 0] if X > 0 else None)


def prior_mc_shared_prefix(W: Weights, cfg: dict, ctx_surface: np.ndarray,
                           ctx_ex: Optional[np.ndarray], z_future: np.ndarray):
    """Same result as get_surface_given_conditions called with the context
    repeated S times, restructured the way the HIP path evaluates it: the
    decoder state after the C context steps depends only on the day, so it is
    computed once per day and broadcast to the S samples.
    ctx_surface (D,C,5,5); z_future (D,S,H,L) -> (D,S,H,Q,5,5)[, (D,S,H,X)]."""
    D, C = ctx_surface.shape[:2]
    _, S, H, L = z_future.shape
    zm, _ = encoder_forward(W, cfg, ctx_surface, ctx_ex)
    ctx = ctx_encoder_forward(W, cfg, ctx_surface, ctx_ex)
    _, finals = lstm_forward(np.concatenate([ctx, zm], axis=-1), W, "decoder.mem", cfg["mem_layers"])
    state = [(np.repeat(h, S, axis=0), np.repeat(c, S, axis=0)) for h, c in finals]
    dec_in = np.concatenate([np.zeros((D * S, H, ctx.shape[-1]), dtype=ctx_surface.dtype),
                             z_future.reshape(D * S, H, L).astype(ctx_surface.dtype)], axis=-1)
    s, e, _ = decoder_forward(W, cfg, dec_in, state)
    s = s.reshape(D, S, H, -1, 5, 5)
    return s, (None if e is None else e.reshape(D, S, H, -1))


def generate_autoregressive_sequence(W: Weights, cfg: dict, ctx_surface: np.ndarray,
                                     ctx_ex: Optional[np.ndarray], z_steps: np.ndarray):
    """CVAEMemRand.generate_autoregressive_sequence
    (cvae_with_mem_randomized.py:717-807, _update_context :925-967).
    z_steps (H,B,C+Hm,L): the randn drawn at :693 in each of the H calls (positions [:C] are overwritten by
    the posterior mean).  Hm is the MODEL's horizon (self.horizon, used because :776 passes no horizon): every
    call generates Hm days, all of them are appended to the output ((B, H*Hm, ...), :801) and only the first
    one's p50 / ex-features are fed back (:786-788).  Hm = 1 in every reference config."""
    surf_ctx = ctx_surface.copy()
    ex_ctx = None if ctx_ex is None else ctx_ex.copy()
    Hm = z_steps.shape[2] - ctx_surface.shape[1]
    outs, exs = [], []
    for step in range(z_steps.shape[0]):
        s, e = get_surface_given_conditions(W, cfg, surf_ctx, ex_ctx, z_steps[step].copy(), Hm)
        outs.append(s)
        new_surface = s[:, 0, 1]                                     # p50 channel, :786/:794
        surf_ctx = np.concatenate([surf_ctx[:, 1:], new_surface[:, None]], axis=1)
        if e is not None:
            exs.append(e)
            ex_ctx = np.concatenate([ex_ctx[:, 1:], e[:, 0][:, None]], axis=1)
    return np.concatenate(outs, axis=1), (np.concatenate(exs, axis=1) if exs else None)


def generate_autoregressive_offset(W: Weights, cfg: dict, ctx_surface: np.ndarray,
                                   ctx_ex: Optional[np.ndarray], z_steps: Sequence[np.ndarray],
                                   horizon: int):
    """CVAEMemRand.generate_autoregressive_offset
    (cvae_with_mem_randomized.py:809-923): len(z_steps) chunks of ``horizon``
    days; next context = last C of the concatenated p50 medians (:907-914)."""
    C = ctx_surface.shape[1]
    surf_ctx, ex_ctx = ctx_surface, ctx_ex
    all_s, all_e, med_s, med_e = [], [], [], []
    for k, z in enumerate(z_steps):
        s, e = get_surface_given_conditions(W, cfg, surf_ctx, ex_ctx, z.copy(), horizon)
        all_s.append(s)
        med_s.append(s[:, :, 1])
        if e is not None:
            all_e.append(e)
            med_e.append(e)
        if k < len(z_steps) - 1:
            surf_ctx = np.concatenate(med_s, axis=1)[:, -C:]
            if e is not None:
                ex_ctx = np.concatenate(med_e, axis=1)[:, -C:]
    return np.concatenate(all_s, axis=1), (np.concatenate(all_e, axis=1) if all_e else None)


# --------------------------------------------------------------------------
# losses (always float64 scalars, like the reference: QuantileLoss buffers are
# float64, cvae_with_mem_randomized.py:23-26)
# --------------------------------------------------------------------------

def quantile_loss(preds: np.ndarray, target: np.ndarray, quantiles, weights=None) -> float:
    """QuantileLoss.forward (cvae_with_mem_randomized.py:28-45)."""
    q = np.asarray(quantiles, dtype=np.float64)
    w = np.ones_like(q) if weights is None else np.asarray(weights, dtype=np.float64)
    losses = []
    for i in range(len(q)):
        err = (target - preds[:, :, i]).astype(np.float64)
        losses.append(np.mean(np.maximum((q[i] - 1) * err, q[i] * err)))
    return float(np.sum(np.asarray(losses) * w) / np.sum(w))


def kl_loss(z_mean: np.ndarray, z_log_var: np.ndarray) -> float:
    """cvae_with_mem_randomized.py:1123-1124: mean over (B,T) of the sum over L."""
    k = -0.5 * (1 + z_log_var - np.exp(z_log_var) - np.square(z_mean))
    return float(np.mean(np.sum(k, axis=-1)))


def ex_feats_loss(pred: np.ndarray, real: np.ndarray, cfg: dict) -> float:
    """cvae_with_mem_randomized.py:1115-1118 (MSE if 'l2' else L1, :564-567)."""
    if cfg.get("ex_loss_on_ret_only", False):
        pred, real = pred[:, :, :1], real[:, :, :1]
    d = pred - real
    return float(np.mean(d * d) if cfg.get("ex_feats_loss_type", "l1") == "l2" else np.mean(np.abs(d)))


def test_step(W: Weights, cfg: dict, surface, ex_feats, eps, horizon) -> Dict[str, float]:
    """CVAEMemRand.test_step (cvae_with_mem_randomized.py:1265-1308)."""
    s, e, zm, zl, _ = forward(W, cfg, surface, ex_feats, eps, horizon)
    C = surface.shape[1] - horizon
    re_s = quantile_loss(s, surface[:, C:], cfg["quantiles"], cfg.get("quantile_loss_weights"))
    re_e = 0.0
    recon = re_s
    if ex_feats is not None:
        re_e = ex_feats_loss(e, ex_feats[:, C:], cfg)
        recon = re_s + cfg.get("re_feat_weight", 1.0) * re_e
    kl = kl_loss(zm, zl)
    return {"loss": recon + cfg["kl_weight"] * kl, "re_surface": re_s, "re_ex_feats": re_e,
            "reconstruction_loss": recon, "kl_loss": kl}


test_step.__test__ = False   # not a pytest test


# --------------------------------------------------------------------------
# MC reduction + counter-based noise
# --------------------------------------------------------------------------

def percentile(samples: np.ndarray, qs=(5, 50, 95), axis: int = 1) -> np.ndarray:
    """np.percentile(..., method='linear') -- the MC->quantile convention of
    experiments/bootstrap_baseline/generate_bootstrap_autoregressive.py:194-199."""
    return np.percentile(samples, list(qs), axis=axis)


_PHILOX_M0 = np.uint64(0xD2511F53)
_PHILOX_M1 = np.uint64(0xCD9E8D57)
_PHILOX_W0 = 0x9E3779B9
_PHILOX_W1 = 0xBB67AE85
_MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0: int, k1: int):
    """Philox4x32-10 (Salmon et al., SC'11).  Counter words are uint32 arrays
    (broadcastable); returns four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & _MASK32 for c in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    for r in range(10):
        p0 = _PHILOX_M0 * c0
        p1 = _PHILOX_M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK32
        kk0 = np.uint64((k0 + r * _PHILOX_W0) & 0xFFFFFFFF)
        kk1 = np.uint64((k1 + r * _PHILOX_W1) & 0xFFFFFFFF)
        c0, c1, c2, c3 = (hi1 ^ c1 ^ kk0), lo1, (hi0 ^ c3 ^ kk1), lo0
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def dropout_mask(p: float, seed: int, lstm_id: int, layer: int, n_rows: int, T: int, hidden: int) -> np.ndarray:
    """The inter-layer dropout factors of the HIP training path (vsv_hip.h, vsv_train_save): unit
    u = 8*tile + 2*jj + hh of (row, t) is kept (factor 1/(1-p)) iff the top 24 bits of word jj of the
    Philox block with counter (row, t, 2*tile + hh, 0x44520000 | id << 8 | layer) are >= p * 2^24.
    Stands in for the Bernoulli draw inside nn.LSTM(dropout=p) (cvae_with_mem_randomized.py:155)."""
    groups = 2 * ((hidden + 7) // 8)
    r, t, g = np.meshgrid(np.arange(n_rows), np.arange(T), np.arange(groups), indexing="ij")
    x = philox4x32_10(r, t, g, np.full_like(g, 0x44520000 | (lstm_id << 8) | layer),
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    out = np.zeros((n_rows, T, 4 * groups), dtype=np.float32)
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    for jj in range(4):
        keep = (x[jj] >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24) >= np.float32(p)
        u = 8 * (g >> 1) + 2 * jj + (g & 1)
        out[r, t, u] = np.where(keep, scale, np.float32(0.0))
    return out[:, :, :hidden]


def philox_normal(seed: int, day0: int, n_days: int, n_samples: int, step0: int,
                  n_steps: int, latent_dim: int) -> np.ndarray:
    """The noise the HIP path draws in on-device RNG mode: element
    (day, sample, step, f) = BoxMuller(Philox(ctr=(sample, day, step, f>>2),
    key=seed))[f&3], float32.  Sharding-invariant by construction."""
    nb = (latent_dim + 3) // 4
    day = (day0 + np.arange(n_days, dtype=np.uint64))[:, None, None, None]
    smp = np.arange(n_samples, dtype=np.uint64)[None, :, None, None]
    stp = (step0 + np.arange(n_steps, dtype=np.uint64))[None, None, :, None]
    blk = np.arange(nb, dtype=np.uint64)[None, None, None, :]
    x = philox4x32_10(smp, day, stp, blk, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    # 23 random bits + 0.5 is exact in float32, so u lies strictly inside (0, 1)
    u = [((xi >> np.uint32(9)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -23) for xi in x]
    out = np.empty(u[0].shape + (4,), dtype=np.float32)
    two_pi = np.float32(6.283185307179586)
    for p in range(2):
        r = np.sqrt(np.float32(-2.0) * np.log(u[2 * p]))
        th = two_pi * u[2 * p + 1]
        out[..., 2 * p] = r * np.cos(th)
        out[..., 2 * p + 1] = r * np.sin(th)
    return out.reshape(n_days, n_samples, n_steps, nb * 4)[..., :latent_dim]
