"""Synthetic model configurations, seeded weights and calendars for benchmarks, smoke tests and
parity tests of the HIP path (no arithmetic of the model lives here).

make_config / state_dict_spec / init_weights describe the reference's CVAEMemRand parameter set
(vae/cvae_with_mem_randomized.py:77-110, 144-160, 230-262, 296-312, 380-450,
969-1020) and fill it from a seeded numpy PCG64 stream, so the GPU box regenerates the very weights the
goldens under tests/golden/ were produced with.  oracle/vsv_oracle.py re-exports them.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import numpy as np

Weights = Dict[str, np.ndarray]


def make_config(latent_dim=5, ex_feats_dim=0, mem_hidden=100, mem_layers=2,
                surface_hidden=(5, 5, 5), quantiles=(0.05, 0.5, 0.95),
                **extra) -> dict:
    """Model config dict with the keys CVAEMemRand.check_input expects
    (vae/cvae_with_mem_randomized.py:969-1020)."""
    cfg = {
        "feat_dim": (5, 5),
        "latent_dim": latent_dim,
        "device": "cpu",
        "kl_weight": 1e-5,
        "re_feat_weight": 1.0,
        "surface_hidden": list(surface_hidden),
        "ex_feats_dim": ex_feats_dim,
        "ex_feats_hidden": None,
        "mem_type": "lstm",
        "mem_hidden": mem_hidden,
        "mem_layers": mem_layers,
        "mem_dropout": 0.0,
        "ctx_surface_hidden": list(surface_hidden),
        "ctx_ex_feats_hidden": None,
        "interaction_layers": None,
        "use_dense_surface": False,
        "compress_context": True,
        "ex_loss_on_ret_only": True,
        "ex_feats_loss_type": "l2",
        "quantiles": list(quantiles),
    }
    cfg.update(extra)
    return cfg


def state_dict_spec(cfg: dict) -> List[Tuple[str, Tuple[int, ...]]]:
    """Parameter names and shapes of CVAEMemRand.state_dict() in module
    registration order (built by
    cvae_with_mem_randomized.py:77-110, 144-160, 230-262, 296-312, 380-450)."""
    L = cfg["latent_dim"]
    X = cfg["ex_feats_dim"]
    Hm = cfg["mem_hidden"]
    nl = cfg["mem_layers"]
    sh = list(cfg["surface_hidden"])
    csh = list(cfg["ctx_surface_hidden"])
    Q = len(cfg["quantiles"])
    dense = bool(cfg.get("use_dense_surface", False))
    per = 1 if dense else 25                     # features per channel (conv) / unit (dense, :84-93, :375-378)
    E = sh[-1] * per + X
    Ec = csh[-1] * per + X
    Hd = sh[-1] * per + X
    spec: List[Tuple[str, Tuple[int, ...]]] = []

    def conv_stack(prefix, names, chans, cin):
        if dense:
            # Flatten + Linear stacks named *_dense_i (:84-93, :237-246)
            cin = 25
            for nm, co in zip(names, chans):
                spec.append((f"{prefix}.{nm.replace('_conv_', '_dense_')}.weight", (co, cin)))
                spec.append((f"{prefix}.{nm.replace('_conv_', '_dense_')}.bias", (co,)))
                cin = co
            return
        for nm, co in zip(names, chans):
            spec.append((f"{prefix}.{nm}.weight", (co, cin, 3, 3)))
            spec.append((f"{prefix}.{nm}.bias", (co,)))
            cin = co

    # gate row blocks: nn.LSTM [i; f; g; o], nn.GRU [r; z; n], nn.RNN (any other mem_type, :158-159) one
    ng = {"lstm": 4, "gru": 3}.get(cfg.get("mem_type", "lstm"), 1)

    def lstm(prefix, inp, hid):
        for l in range(nl):
            i = inp if l == 0 else hid
            spec.append((f"{prefix}.weight_ih_l{l}", (ng * hid, i)))
            spec.append((f"{prefix}.weight_hh_l{l}", (ng * hid, hid)))
            spec.append((f"{prefix}.bias_ih_l{l}", (ng * hid,)))
            spec.append((f"{prefix}.bias_hh_l{l}", (ng * hid,)))

    n_int = cfg.get("interaction_layers") or 0

    def interaction(prefix, stem, width):
        # n Linear+ReLU and a final Linear, width kept (:131-142, :283-294, :452-463)
        if n_int > 0:
            for nm in [f"{stem}_interact_linear_{i}" for i in range(n_int)] + [f"{stem}_interact_linear_final"]:
                spec.append((f"{prefix}.interaction.{nm}.weight", (width, width)))
                spec.append((f"{prefix}.interaction.{nm}.bias", (width,)))

    conv_stack("encoder.surface_embedding", [f"enc_conv_{i}" for i in range(len(sh))], sh, 1)
    interaction("encoder", "enc", E)
    lstm("encoder.mem", E, Hm)
    spec += [("encoder.z_mean_layer.weight", (L, Hm)), ("encoder.z_mean_layer.bias", (L,)),
             ("encoder.z_log_var_layer.weight", (L, Hm)), ("encoder.z_log_var_layer.bias", (L,))]
    conv_stack("ctx_encoder.ctx_surface_embedding", [f"ctx_enc_conv_{i}" for i in range(len(csh))], csh, 1)
    interaction("ctx_encoder", "ctx", Ec)
    lstm("ctx_encoder.mem", Ec, Hm)
    # final_compression is nn.Identity() without compress_context (:225-228); the decoder then takes the
    # context memory's hidden state next to z (:363-366)
    Lc = L if cfg.get("compress_context", False) else Hm
    if cfg.get("compress_context", False):
        spec += [("ctx_encoder.final_compression.weight", (L, Hm)),
                 ("ctx_encoder.final_compression.bias", (L,))]
    lstm("decoder.mem", Lc + L, Hd)
    interaction("decoder", "dec", Hd)
    spec += [("decoder.surface_decoder_input.weight", (sh[-1] * per, Hd)),
             ("decoder.surface_decoder_input.bias", (sh[-1] * per,))]
    cin = sh[-1]
    if dense:
        # Linear stack back to the 25 grid points, then Linear(25 -> 25 Q) (:394-407)
        for i, co in enumerate(reversed(sh[:-1])):
            spec.append((f"decoder.surface_decoder.dec_dense_{i}.weight", (co, cin)))
            spec.append((f"decoder.surface_decoder.dec_dense_{i}.bias", (co,)))
            cin = co
        spec.append(("decoder.surface_decoder.dec_final.weight", (25, cin)))
        spec.append(("decoder.surface_decoder.dec_final.bias", (25,)))
        spec.append(("decoder.surface_decoder.dec_output.weight", (25 * Q, 25)))
        spec.append(("decoder.surface_decoder.dec_output.bias", (25 * Q,)))
    else:
        # ConvTranspose2d weights are (in, out, kH, kW)
        for i, co in enumerate(reversed(sh[:-1])):
            spec.append((f"decoder.surface_decoder.dec_deconv_{i}.weight", (cin, co, 3, 3)))
            spec.append((f"decoder.surface_decoder.dec_deconv_{i}.bias", (co,)))
            cin = co
        spec.append(("decoder.surface_decoder.dec_final.weight", (cin, cin, 3, 3)))
        spec.append(("decoder.surface_decoder.dec_final.bias", (cin,)))
        spec.append(("decoder.surface_decoder.dec_output.weight", (Q, cin, 3, 3)))
        spec.append(("decoder.surface_decoder.dec_output.bias", (Q,)))
    if X > 0:
        spec += [("decoder.ex_feats_decoder_input.weight", (X, Hd)),
                 ("decoder.ex_feats_decoder_input.bias", (X,)),
                 ("decoder.ex_feats_decoder.weight", (X, X)),
                 ("decoder.ex_feats_decoder.bias", (X,))]
    return spec


def init_weights(cfg: dict, seed: int = 0, scale: float = 1.0,
                 dtype=np.float32) -> Weights:
    """Deterministic weights (numpy PCG64) with PyTorch-default-like ranges
    U(-1/sqrt(fan), 1/sqrt(fan)) * scale.  The same function runs on the GPU
    box, so only outputs -- never weights -- are committed as goldens."""
    rng = np.random.Generator(np.random.PCG64(seed))
    W: Weights = {}
    Hm = cfg["mem_hidden"]
    for name, shape in state_dict_spec(cfg):
        if ".mem." in name:
            fan = dict(state_dict_spec(cfg))[name.rsplit(".", 1)[0] + "." + "weight_hh_l" + name[-1]][1]   # hidden size
        elif len(shape) == 4:
            # Conv2d fan_in = Cin*9; ConvTranspose2d (in,out,k,k) fan uses dim 1
            fan = shape[1] * 9
        elif name.endswith(".weight"):
            fan = shape[1]
        else:
            # bias: use the fan of its weight
            wshape = dict(state_dict_spec(cfg))[name[:-4] + "weight"]
            fan = wshape[1] * (9 if len(wshape) == 4 else 1)
        bound = scale / math.sqrt(fan)
        W[name] = rng.uniform(-bound, bound, size=shape).astype(dtype)
    return W


def synthetic_calendar(n_days: int, C: int, X: int, seed: int = 1234):
    """vol (n_days + C, 5, 5), ex (n_days + C, X) or None: the reference's own synthetic test distribution
    (experiments/diagnostics/test_multi_offset_training.py:65-66), float32."""
    rng = np.random.Generator(np.random.PCG64(seed))
    vol = (0.2 + 0.1 * rng.standard_normal((n_days + C, 5, 5))).astype(np.float32)
    ex = (0.01 * rng.standard_normal((n_days + C, X))).astype(np.float32) if X else None
    return vol, ex
