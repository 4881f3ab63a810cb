#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE.

Runs only in the build container (needs /root/reference, which never travels
to the GPU box).  For every case it
  * builds the reference CVAEMemRand on CPU,
  * loads the weights produced by oracle.vsv_oracle.init_weights(seed, scale)
    (regenerated from the seed on the GPU box, so weights are never committed),
  * injects the noise (torch.randn_like / torch.randn patched to pop from a
    queue of stored tensors),
  * stores inputs, noise and the reference's outputs in one small .npz.

Usage:  python oracle/make_goldens.py            (writes tests/golden/*.npz)
"""
from __future__ import annotations

import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")

import torch  # noqa: E402

from oracle import vsv_oracle as vo  # noqa: E402
from vae.cvae_with_mem_randomized import CVAEMemRand, QuantileLoss  # noqa: E402  (the reference)
import vae as _ref_vae  # noqa: E402

assert os.path.abspath(_ref_vae.__path__[0]).startswith("/root/reference"), "goldens must come from the reference's vae package"

OUT = os.path.join(REPO, "tests", "golden")


class NoiseQueue:
    """Patches torch.randn / torch.randn_like to return queued tensors."""

    def __init__(self):
        self.queue = []
        self.used = []

    def push(self, arr):
        self.queue.append(arr)

    def _pop(self, shape, dtype):
        arr = self.queue.pop(0)
        assert tuple(arr.shape) == tuple(shape), (arr.shape, shape)
        self.used.append(arr)
        return torch.from_numpy(np.ascontiguousarray(arr)).to(dtype)

    def __enter__(self):
        self._randn, self._randn_like = torch.randn, torch.randn_like

        def randn(*size, **kw):
            if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
                size = tuple(size[0])
            return self._pop(size, kw.get("dtype") or torch.get_default_dtype())

        def randn_like(t, **kw):
            return self._pop(t.shape, t.dtype)

        torch.randn, torch.randn_like = randn, randn_like
        return self

    def __exit__(self, *a):
        torch.randn, torch.randn_like = self._randn, self._randn_like
        assert not self.queue, "unused injected noise"


def build_reference(cfg: dict, seed: int, scale: float, np_dtype):
    tdtype = torch.float64 if np_dtype == np.float64 else torch.float32
    torch.set_default_dtype(tdtype)
    model = CVAEMemRand(dict(cfg))
    W = vo.init_weights(cfg, seed, scale, np_dtype)
    sd = model.state_dict()
    for k, v in W.items():
        assert tuple(sd[k].shape) == v.shape, k
        sd[k] = torch.from_numpy(v.copy())
    model.load_state_dict(sd, strict=True)
    model.eval()
    return model, W


def synth(rng, B, T, X, np_dtype):
    """The reference's own synthetic test distribution
    (experiments/diagnostics/test_multi_offset_training.py:65-66)."""
    surf = (0.2 + 0.1 * rng.standard_normal((B, T, 5, 5))).astype(np_dtype)
    ex = (0.01 * rng.standard_normal((B, T, X))).astype(np_dtype) if X else None
    return surf, ex


def t2n(t):
    return t.detach().cpu().numpy()


def case_forward(name, cfg, seed, scale, np_dtype, B, T, horizon):
    """G1: model.forward with injected eps; G3: test_step losses."""
    model, _ = build_reference(cfg, seed, scale, np_dtype)
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    X, L = cfg["ex_feats_dim"], cfg["latent_dim"]
    surf, ex = synth(rng, B, T, X, np_dtype)
    eps = rng.standard_normal((B, T, L)).astype(np_dtype)
    x = {"surface": torch.from_numpy(surf)}
    if X:
        x["ex_feats"] = torch.from_numpy(ex)
    model.horizon = horizon
    with NoiseQueue() as nq:
        nq.push(eps)
        out = model.forward(x)
    with NoiseQueue() as nq:
        nq.push(eps)
        losses = model.test_step(x)
    d = dict(surface=surf, eps=eps, horizon=horizon, seed=seed, scale=scale,
             recon=t2n(out[0]), z_mean=t2n(out[-3]), z_log_var=t2n(out[-2]), z=t2n(out[-1]))
    if X:
        d["ex_feats"] = ex
        d["ex_recon"] = t2n(out[1])
    for k, v in losses.items():
        d["loss_" + k] = np.float64(t2n(v).reshape(-1)[0])
    save(name, cfg, d)


def case_prior(name, cfg, seed, scale, np_dtype, B, C, H):
    """G2: get_surface_given_conditions(c, z=z.clone(), horizon=H)."""
    model, _ = build_reference(cfg, seed, scale, np_dtype)
    rng = np.random.Generator(np.random.PCG64(2000 + seed))
    X, L = cfg["ex_feats_dim"], cfg["latent_dim"]
    surf, ex = synth(rng, B, C, X, np_dtype)
    z = rng.standard_normal((B, C + H, L)).astype(np_dtype)
    c = {"surface": torch.from_numpy(surf)}
    if X:
        c["ex_feats"] = torch.from_numpy(ex)
    zt = torch.from_numpy(z.copy())
    out = model.get_surface_given_conditions(c, z=zt, horizon=H)
    d = dict(surface=surf, z=z, horizon=H, seed=seed, scale=scale, z_after=t2n(zt))
    if X:
        d["ex_feats"] = ex
        d["surf_out"], d["ex_out"] = t2n(out[0]), t2n(out[1])
    else:
        d["surf_out"] = t2n(out)
    # submodule outputs (analysis scripts call these directly)
    with NoiseQueue() as nq:
        nq.push(np.zeros((B, C, L), dtype=np_dtype))
        zm, zl, _ = model.encoder(c)
    d["enc_z_mean"], d["enc_z_log_var"] = t2n(zm), t2n(zl)
    d["ctx_embedding"] = t2n(model.ctx_encoder(c))
    save(name, cfg, d)


def case_ar(name, cfg, seed, scale, B, C, H, model_horizon=1):
    """G6: generate_autoregressive_sequence, float64 default dtype (z is
    created float64 at cvae_with_mem_randomized.py:693-694).  model_horizon: the model's own
    self.horizon, which every inner call generates (:776 passes no horizon)."""
    np_dtype = np.float64
    model, _ = build_reference(cfg, seed, scale, np_dtype)
    model.horizon = model_horizon
    rng = np.random.Generator(np.random.PCG64(3000 + seed))
    X, L = cfg["ex_feats_dim"], cfg["latent_dim"]
    surf, ex = synth(rng, B, C, X, np_dtype)
    zs = rng.standard_normal((H, B, C + model_horizon, L))
    c = {"surface": torch.from_numpy(surf)}
    if X:
        c["ex_feats"] = torch.from_numpy(ex)
    with NoiseQueue() as nq:
        for h in range(H):
            nq.push(zs[h])                      # torch.randn at :693
            nq.push(np.zeros((B, C, L)))        # encoder's own randn_like at :198 (result unused)
        out = model.generate_autoregressive_sequence(c, horizon=H)
    d = dict(surface=surf, z_steps=zs, horizon=H, seed=seed, scale=scale, model_horizon=model_horizon)
    if X:
        d["ex_feats"] = ex
        d["surf_out"], d["ex_out"] = t2n(out[0]), t2n(out[1])
    else:
        d["surf_out"] = t2n(out)
    save(name, cfg, d)


def case_ar_offset(name, cfg, seed, scale, B, C, H, steps):
    """G7: generate_autoregressive_offset(ar_steps, horizon=H, offset=H)."""
    np_dtype = np.float64
    cfg = dict(cfg, context_len=C)
    model, _ = build_reference(cfg, seed, scale, np_dtype)
    rng = np.random.Generator(np.random.PCG64(4000 + seed))
    X, L = cfg["ex_feats_dim"], cfg["latent_dim"]
    surf, ex = synth(rng, B, C, X, np_dtype)
    zs = rng.standard_normal((steps, B, C + H, L))
    c = {"surface": torch.from_numpy(surf)}
    if X:
        c["ex_feats"] = torch.from_numpy(ex)
    with NoiseQueue() as nq:
        for k in range(steps):
            nq.push(zs[k])
            nq.push(np.zeros((B, C, L)))
        out = model.generate_autoregressive_offset(c, ar_steps=steps, horizon=H, offset=H)
    d = dict(surface=surf, z_steps=zs, horizon=H, seed=seed, scale=scale)
    if X:
        d["ex_feats"] = ex
        d["surf_out"], d["ex_out"] = t2n(out[0]), t2n(out[1])
    else:
        d["surf_out"] = t2n(out)
    save(name, cfg, d)


def case_grads(name, cfg, seed, scale, B, T, horizon, full=True):
    """G4: gradients of the train_step loss (cvae_with_mem_randomized.py:1113-1127) w.r.t. every
    parameter, float64, dropout 0, injected eps.  test_step builds the same loss graph as
    train_step (:1265-1308 vs :1105-1125); backward() here gives the un-clipped gradients, then
    train_step itself (SGD, lr 0) gives the clipped ones (clip_grad_norm_ 1.0, :1128)."""
    np_dtype = np.float64
    model, _ = build_reference(cfg, seed, scale, np_dtype)
    model.train()                                   # dropout is 0 in cfg, so train == eval numerically
    rng = np.random.Generator(np.random.PCG64(6000 + seed))
    X, L = cfg["ex_feats_dim"], cfg["latent_dim"]
    surf, ex = synth(rng, B, T, X, np_dtype)
    eps = rng.standard_normal((B, T, L)).astype(np_dtype)
    x = {"surface": torch.from_numpy(surf)}
    if X:
        x["ex_feats"] = torch.from_numpy(ex)
    model.horizon = horizon
    model.zero_grad()
    with NoiseQueue() as nq:
        nq.push(eps)
        losses = model.test_step(x)
    losses["loss"].backward()
    d = dict(surface=surf, eps=eps, horizon=horizon, seed=seed, scale=scale)
    if X:
        d["ex_feats"] = ex
    for k, v in losses.items():
        d["loss_" + k] = np.float64(t2n(v).reshape(-1)[0])
    grads = {n: t2n(p.grad).copy() for n, p in model.named_parameters()}
    total = float(np.sqrt(sum(float((g ** 2).sum()) for g in grads.values())))
    d["grad_total_norm"] = np.float64(total)
    for n, g in grads.items():
        d["gnorm/" + n] = np.float64(np.sqrt((g ** 2).sum()))
        if full:
            d["grad/" + n] = g.astype(np.float32)
        else:
            flat = g.ravel()
            idx = np.linspace(0, flat.size - 1, num=min(flat.size, 64)).astype(np.int64)
            d["gidx/" + n] = idx
            d["gval/" + n] = flat[idx]
    # the real train_step: clipped gradients left in .grad (lr = 0 keeps the weights)
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    with NoiseQueue() as nq:
        nq.push(eps)
        out = model.train_step(x, opt)
    d["train_loss"] = np.float64(t2n(out["loss"]).reshape(-1)[0])
    cl = float(np.sqrt(sum(float((t2n(p.grad) ** 2).sum()) for p in model.parameters())))
    d["clipped_total_norm"] = np.float64(cl)
    save(name, cfg, d)


def case_multihorizon(name, cfg, seed, scale, B, T, horizons):
    """G5: train_step_multihorizon (:1139-1263) with one injected eps per horizon."""
    np_dtype = np.float64
    model, _ = build_reference(cfg, seed, scale, np_dtype)
    model.train()
    rng = np.random.Generator(np.random.PCG64(7000 + seed))
    X, L = cfg["ex_feats_dim"], cfg["latent_dim"]
    surf, ex = synth(rng, B, T, X, np_dtype)
    eps = rng.standard_normal((len(horizons), B, T, L)).astype(np_dtype)
    x = {"surface": torch.from_numpy(surf)}
    if X:
        x["ex_feats"] = torch.from_numpy(ex)
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    with NoiseQueue() as nq:
        for i in range(len(horizons)):
            nq.push(eps[i])
        out = model.train_step_multihorizon(x, opt, horizons=list(horizons))
    d = dict(surface=surf, eps=eps, horizons=np.asarray(horizons), seed=seed, scale=scale)
    if X:
        d["ex_feats"] = ex
    d["loss"] = np.float64(out["loss"])
    d["reconstruction_loss"] = np.float64(out["reconstruction_loss"])
    d["kl_loss"] = np.float64(out["kl_loss"])
    d["horizon_losses"] = np.asarray([out["horizon_losses"][h] for h in horizons], dtype=np.float64)
    grads = {n: t2n(p.grad).copy() for n, p in model.named_parameters()}
    d["grad_total_norm"] = np.float64(np.sqrt(sum(float((g ** 2).sum()) for g in grads.values())))
    for n, g in grads.items():
        d["gnorm/" + n] = np.float64(np.sqrt((g ** 2).sum()))
        flat = g.ravel()
        idx = np.linspace(0, flat.size - 1, num=min(flat.size, 64)).astype(np.int64)
        d["gidx/" + n] = idx
        d["gval/" + n] = flat[idx]
    save(name, cfg, d)


def build_reference_cprior(cfg: dict, seed: int, scale: float, np_dtype):
    """The reference's CVAEMemRandConditionalPrior with seeded base + prior-network weights."""
    from vae.cvae_conditional_prior import CVAEMemRandConditionalPrior
    tdtype = torch.float64 if np_dtype == np.float64 else torch.float32
    torch.set_default_dtype(tdtype)
    model = CVAEMemRandConditionalPrior(dict(cfg, use_conditional_prior=True))
    W = dict(vo.init_weights(cfg, seed, scale, np_dtype))
    W.update(vo.init_prior_weights(cfg, seed, scale, np_dtype))
    sd = model.state_dict()
    for k, v in W.items():
        assert tuple(sd[k].shape) == v.shape, k
        sd[k] = torch.from_numpy(v.copy())
    model.load_state_dict(sd, strict=True)
    model.eval()
    return model


def case_cprior_gen(name, cfg, seed, scale, np_dtype, B, C, H, sample_context, ar_phi):
    """G8: CVAEMemRandConditionalPrior.get_surface_given_conditions(z=None) with the draws injected:
    torch.randn((B,H,L)) [or H draws of (B,1,L) for AR(1)], the encoder's own randn_like (unused),
    then randn_like for the sampled context latents (cvae_conditional_prior.py:166-222)."""
    model = build_reference_cprior(cfg, seed, scale, np_dtype)
    rng = np.random.Generator(np.random.PCG64(9000 + seed))
    X, L = cfg["ex_feats_dim"], cfg["latent_dim"]
    surf, ex = synth(rng, B, C, X, np_dtype)
    innov = rng.standard_normal((B, H, L)).astype(np.float32)
    eps_ctx = rng.standard_normal((B, C, L)).astype(np_dtype)
    c = {"surface": torch.from_numpy(surf)}
    if X:
        c["ex_feats"] = torch.from_numpy(ex)
    with NoiseQueue() as nq:
        if ar_phi > 0:
            for t in range(H):
                nq.push(innov[:, t:t + 1])
        else:
            nq.push(innov)
        nq.push(np.zeros((B, C, L), dtype=np_dtype))
        if sample_context:
            nq.push(eps_ctx)
        out = model.get_surface_given_conditions(c, horizon=H, sample_context=sample_context, ar_phi=ar_phi)
    pm, plv = model.prior_network(c)
    d = dict(surface=surf, innov=innov, eps_ctx=eps_ctx, horizon=H, seed=seed, scale=scale,
             sample_context=np.int64(sample_context), ar_phi=np.float64(ar_phi),
             prior_mean=t2n(pm), prior_logvar=t2n(plv))
    if X:
        d["ex_feats"] = ex
        d["surf_out"], d["ex_out"] = t2n(out[0]), t2n(out[1])
    else:
        d["surf_out"] = t2n(out)
    save(name, cfg, d)


def _all_grads(model, d):
    grads = {n: t2n(p.grad).copy() for n, p in model.named_parameters()}
    d["grad_total_norm"] = np.float64(np.sqrt(sum(float((g ** 2).sum()) for g in grads.values())))
    for n, g in grads.items():
        d["gnorm/" + n] = np.float64(np.sqrt((g ** 2).sum()))
        flat = g.ravel()
        idx = np.linspace(0, flat.size - 1, num=min(flat.size, 64)).astype(np.int64)
        d["gidx/" + n] = idx
        d["gval/" + n] = flat[idx]


def case_cprior_train(name, cfg, seed, scale, B, T, horizon, horizons):
    """G8: conditional-prior training.  test_step losses (eval), then train_step and
    train_step_multihorizon (cached-encoder path) with SGD lr 0 and clip_grad_norm_ patched to a no-op,
    so the un-clipped gradients of every parameter -- prior network included -- stay in .grad
    (cvae_conditional_prior.py:241-458).  float64, dropout 0."""
    np_dtype = np.float64
    model = build_reference_cprior(cfg, seed, scale, np_dtype)
    rng = np.random.Generator(np.random.PCG64(9500 + seed))
    X, L = cfg["ex_feats_dim"], cfg["latent_dim"]
    surf, ex = synth(rng, B, T, X, np_dtype)
    eps = rng.standard_normal((B, T, L)).astype(np_dtype)
    eps_h = rng.standard_normal((len(horizons), B, T, L)).astype(np_dtype)
    x = {"surface": torch.from_numpy(surf)}
    if X:
        x["ex_feats"] = torch.from_numpy(ex)
    model.horizon = horizon
    with NoiseQueue() as nq:
        nq.push(eps)
        losses = model.test_step(x)
    d = dict(surface=surf, eps=eps, eps_h=eps_h, horizon=horizon, horizons=np.asarray(horizons), seed=seed, scale=scale)
    if X:
        d["ex_feats"] = ex
    for k, v in losses.items():
        d["loss_" + k] = np.float64(t2n(v).reshape(-1)[0])
    clip = torch.nn.utils.clip_grad_norm_
    torch.nn.utils.clip_grad_norm_ = lambda *a, **k: None
    try:
        model.train()
        opt = torch.optim.SGD(model.parameters(), lr=0.0)
        with NoiseQueue() as nq:
            nq.push(eps)
            out = model.train_step(x, opt)
        d["train_loss"] = np.float64(t2n(out["loss"]).reshape(-1)[0])
        d1 = {}
        _all_grads(model, d1)
        d.update({"ts/" + k: v for k, v in d1.items()})
        with NoiseQueue() as nq:
            nq.push(np.zeros((B, T, L)))                     # the cached encoder call's own (unused) sample
            for i in range(len(horizons)):
                nq.push(eps_h[i])
            out = model.train_step_multihorizon(x, opt, horizons=list(horizons))
        d["mh_loss"] = np.float64(t2n(out["loss"]).reshape(-1)[0])
        d["mh_horizon_losses"] = np.asarray([out["horizon_losses"][h] for h in horizons], dtype=np.float64)
        d["mh_kl_loss"] = np.float64(t2n(out["kl_loss"]).reshape(-1)[0])
        d2 = {}
        _all_grads(model, d2)
        d.update({"mh/" + k: v for k, v in d2.items()})
    finally:
        torch.nn.utils.clip_grad_norm_ = clip
    save(name, cfg, d)


def case_ar_train(name, cfg, seed, scale, B, C, horizon, offset, ar_steps):
    """G9: vae.utils.train_autoregressive_step (reference vae/utils.py:426-623) with SGD lr 0: losses and
    the gradients of every parameter, which flow through the fed-back p50 predictions.  float64, dropout 0."""
    from vae.utils import train_autoregressive_step
    np_dtype = np.float64
    cfg = dict(cfg, context_len=C)
    model, _ = build_reference(cfg, seed, scale, np_dtype)
    model.train()
    rng = np.random.Generator(np.random.PCG64(9800 + seed))
    X, L = cfg["ex_feats_dim"], cfg["latent_dim"]
    T = C + (ar_steps - 1) * offset + horizon
    surf, ex = synth(rng, B, T, X, np_dtype)
    eps = rng.standard_normal((ar_steps, B, C + horizon, L)).astype(np_dtype)
    batch = {"surface": torch.from_numpy(surf)}
    if X:
        batch["ex_feats"] = torch.from_numpy(ex)
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    with NoiseQueue() as nq:
        for k in range(ar_steps):
            nq.push(eps[k])
        out = train_autoregressive_step(model, batch, opt, ar_steps, horizon, offset)
    d = dict(surface=surf, eps=eps, horizon=horizon, offset=offset, ar_steps=ar_steps, context_len=C, seed=seed, scale=scale,
             loss=np.float64(out["loss"]), reconstruction_loss=np.float64(out["reconstruction_loss"]),
             kl_loss=np.float64(out["kl_loss"]))
    if X:
        d["ex_feats"] = ex
    _all_grads(model, d)
    save(name, cfg, d)


def main_ar_train():
    ex5 = vo.make_config(5, 3, quantile_loss_weights=[5.0, 1.0, 5.0], kl_weight=1e-3)
    no_ex = vo.make_config(5, 0)
    case_ar_train("g9_artrain_ex5_C6_H6_off6", ex5, 50, 2.0, 4, 6, 6, 6, 3)
    case_ar_train("g9_artrain_ex5_C6_H6_off3", ex5, 51, 2.0, 4, 6, 6, 3, 3)
    case_ar_train("g9_artrain_noex_C8_H4_off4", no_ex, 52, 2.0, 3, 8, 4, 4, 3)


def main_interaction():
    """interaction_layers > 0 (the v3 conditional-prior trainer's setting, value 2): forward, prior
    generation, train_step gradients, and the conditional-prior variant."""
    f32, f64 = np.float32, np.float64
    ex12i = vo.make_config(12, 3, interaction_layers=2, quantile_loss_weights=[5.0, 1.0, 5.0], kl_weight=1e-3)
    noex1 = vo.make_config(5, 0, interaction_layers=1)
    case_forward("g1_fwd_int2_ex12_T30_h7_f32", ex12i, 60, 2.0, f32, 4, 30, 7)
    case_forward("g1_fwd_int1_noex_T9_h1_f64", noex1, 61, 2.0, f64, 3, 9, 1)
    case_prior("g2_prior_int2_ex12_C20_H7_f32", ex12i, 62, 2.0, f32, 6, 20, 7)
    case_grads("g4_grads_int2_ex12_B4_T20_h3", ex12i, 63, 2.0, 4, 20, 3, full=False)
    case_grads("g4_grads_int1_noex_B5_T9_h2", noex1, 64, 2.0, 5, 9, 2, full=False)
    case_cprior_gen("g8_cprior_gen_int2_ex12_C20_H7_f32", ex12i, 65, 2.0, f32, 4, 20, 7, True, 0.3)
    case_cprior_train("g8_cprior_train_int2_ex12_B4_T24_h3", ex12i, 66, 2.0, 4, 24, 3, [1, 7])


def main_gru():
    """mem_type="gru" (nn.GRU memories, :154-159 / :306-311 / :475-480): forward, prior generation,
    train_step gradients, one-layer, and the conditional-prior variant."""
    f32, f64 = np.float32, np.float64
    ex5 = vo.make_config(5, 3, mem_type="gru")
    noex = vo.make_config(5, 0, mem_type="gru")
    ex12 = vo.make_config(12, 3, mem_type="gru", quantile_loss_weights=[5.0, 1.0, 5.0], kl_weight=1e-3)
    one = vo.make_config(5, 0, mem_layers=1, mem_type="gru")
    case_forward("g1_fwd_gru_ex5_T21_h1_f32", ex5, 70, 2.0, f32, 4, 21, 1)
    case_forward("g1_fwd_gru_noex_T9_h3_f64", noex, 71, 2.0, f64, 3, 9, 3)
    case_forward("g1_fwd_gru_1layer_T6_h1_f32", one, 72, 1.0, f32, 2, 6, 1)
    case_prior("g2_prior_gru_ex12_C20_H7_f32", ex12, 73, 2.0, f32, 6, 20, 7)
    case_prior("g2_prior_gru_noex_C5_H1_f64", noex, 74, 2.0, f64, 8, 5, 1)
    case_grads("g4_grads_gru_ex5_B6_T21_h3", ex5, 75, 2.0, 6, 21, 3, full=False)
    case_grads("g4_grads_gru_ex12_B4_T20_h7", ex12, 76, 2.0, 4, 20, 7, full=False)
    case_grads("g4_grads_gru_1layer_B5_T9_h2", one, 77, 2.0, 5, 9, 2, full=False)
    case_ar("g6_ar_gru_ex5_C20_H30", ex5, 78, 2.0, 4, 20, 30)
    case_cprior_gen("g8_cprior_gen_gru_ex5_C20_H7_f32", ex5, 79, 2.0, f32, 4, 20, 7, True, 0.3)
    case_cprior_train("g8_cprior_train_gru_ex5_B4_T24_h3", ex5, 80, 2.0, 4, 24, 3, [1, 7])


def main_nocompress():
    """compress_context=False, the class default (:1014-1015): the context encoder's final_compression is
    nn.Identity() (:225-228), the decoder memory takes [context memory state (mem_hidden) | z] (:363-366) and the
    prior network's heads read its memory state (conditional_prior_network.py:73-78)."""
    f32, f64 = np.float32, np.float64
    ex5 = vo.make_config(5, 3, compress_context=False)
    noex = vo.make_config(5, 0, compress_context=False)
    ex12 = vo.make_config(12, 3, compress_context=False, quantile_loss_weights=[5.0, 1.0, 5.0], kl_weight=1e-3)
    gru = vo.make_config(5, 3, compress_context=False, mem_type="gru", interaction_layers=1)
    case_forward("g1_fwd_nocomp_ex5_T21_h1_f32", ex5, 90, 2.0, f32, 4, 21, 1)
    case_forward("g1_fwd_nocomp_noex_T9_h3_f64", noex, 91, 2.0, f64, 3, 9, 3)
    case_forward("g1_fwd_nocomp_gru_int1_T12_h2_f32", gru, 92, 2.0, f32, 3, 12, 2)
    case_prior("g2_prior_nocomp_ex12_C20_H7_f32", ex12, 93, 2.0, f32, 6, 20, 7)
    case_prior("g2_prior_nocomp_noex_C5_H1_f64", noex, 94, 2.0, f64, 8, 5, 1)
    case_grads("g4_grads_nocomp_ex5_B6_T21_h3", ex5, 95, 2.0, 6, 21, 3, full=False)
    case_grads("g4_grads_nocomp_ex12_B4_T20_h7", ex12, 96, 2.0, 4, 20, 7, full=False)
    case_grads("g4_grads_nocomp_gru_int1_B5_T9_h2", gru, 97, 2.0, 5, 9, 2, full=False)
    case_ar("g6_ar_nocomp_ex5_C20_H30", ex5, 98, 2.0, 4, 20, 30)
    case_cprior_gen("g8_cprior_gen_nocomp_ex5_C20_H7_f32", ex5, 99, 2.0, f32, 4, 20, 7, True, 0.3)
    case_cprior_train("g8_cprior_train_nocomp_ex5_B4_T24_h3", ex5, 100, 2.0, 4, 24, 3, [1, 7])


def main_dense():
    """use_dense_surface=True: Flatten + Linear+ReLU surface embeddings (:84-93, :237-246), decoder memory of width
    surface_hidden[-1] + ex_feats_dim (:375-380), Linear decoder stack back to the grid (:394-407), the prior
    network's dense embedding (conditional_prior_network.py:90-99).  No reference config turns it on; widths chosen
    to cross tile boundaries (40 > 32 units, 8 < one tile) and to differ between encoder and context encoder."""
    f32, f64 = np.float32, np.float64

    def mk(L, X, sh, csh=None, **kw):
        cfg = vo.make_config(L, X, surface_hidden=sh, use_dense_surface=True, **kw)
        cfg["ctx_surface_hidden"] = list(csh if csh is not None else sh)
        return cfg
    ex5 = mk(5, 3, (40, 16, 8))
    noex = mk(5, 0, (24, 12), csh=(30,))
    ex12 = mk(12, 3, (64, 32, 16), csh=(48, 20), quantile_loss_weights=[5.0, 1.0, 5.0], kl_weight=1e-3)
    gru = mk(5, 3, (33,), mem_type="gru", interaction_layers=1, compress_context=False)
    case_forward("g1_fwd_dense_ex5_T21_h1_f32", ex5, 130, 2.0, f32, 4, 21, 1)
    case_forward("g1_fwd_dense_noex_T9_h3_f64", noex, 131, 2.0, f64, 3, 9, 3)
    case_forward("g1_fwd_dense_gru_int1_T12_h2_f32", gru, 132, 2.0, f32, 3, 12, 2)
    case_prior("g2_prior_dense_ex12_C20_H7_f32", ex12, 133, 2.0, f32, 6, 20, 7)
    case_prior("g2_prior_dense_noex_C5_H1_f64", noex, 134, 2.0, f64, 8, 5, 1)
    case_grads("g4_grads_dense_ex5_B6_T21_h3", ex5, 135, 2.0, 6, 21, 3, full=False)
    case_grads("g4_grads_dense_ex12_B4_T20_h7", ex12, 136, 2.0, 4, 20, 7, full=False)
    case_grads("g4_grads_dense_gru_int1_B5_T9_h2", gru, 137, 2.0, 5, 9, 2, full=False)
    case_ar("g6_ar_dense_ex5_C20_H30", ex5, 138, 2.0, 4, 20, 30)
    case_cprior_gen("g8_cprior_gen_dense_ex5_C20_H7_f32", ex5, 139, 2.0, f32, 4, 20, 7, True, 0.3)
    case_cprior_train("g8_cprior_train_dense_ex5_B4_T24_h3", ex5, 140, 2.0, 4, 24, 3, [1, 7])


def main_rnn():
    """mem_type other than "lstm" / "gru": nn.RNN (tanh) memories (:158-159, :310-311, :479-480)."""
    f32, f64 = np.float32, np.float64
    ex5 = vo.make_config(5, 3, mem_type="rnn")
    noex = vo.make_config(5, 0, mem_type="rnn", compress_context=False)
    ex12 = vo.make_config(12, 3, mem_type="rnn", quantile_loss_weights=[5.0, 1.0, 5.0], kl_weight=1e-3)
    one = vo.make_config(5, 0, mem_layers=1, mem_type="rnn")
    case_forward("g1_fwd_rnn_ex5_T21_h1_f32", ex5, 110, 2.0, f32, 4, 21, 1)
    case_forward("g1_fwd_rnn_nocomp_noex_T9_h3_f64", noex, 111, 2.0, f64, 3, 9, 3)
    case_forward("g1_fwd_rnn_1layer_T6_h1_f32", one, 112, 1.0, f32, 2, 6, 1)
    case_prior("g2_prior_rnn_ex12_C20_H7_f32", ex12, 113, 2.0, f32, 6, 20, 7)
    case_grads("g4_grads_rnn_ex5_B6_T21_h3", ex5, 114, 2.0, 6, 21, 3, full=False)
    case_grads("g4_grads_rnn_ex12_B4_T20_h7", ex12, 115, 2.0, 4, 20, 7, full=False)
    case_grads("g4_grads_rnn_1layer_B5_T9_h2", one, 116, 2.0, 5, 9, 2, full=False)
    case_ar("g6_ar_rnn_ex5_C20_H30", ex5, 117, 2.0, 4, 20, 30)
    case_cprior_train("g8_cprior_train_rnn_ex5_B4_T24_h3", ex5, 118, 2.0, 4, 24, 3, [1, 7])


def main_cprior():
    f32, f64 = np.float32, np.float64
    ex5 = vo.make_config(5, 3)
    no_ex = vo.make_config(5, 0)
    ex12 = vo.make_config(12, 3, quantile_loss_weights=[5.0, 1.0, 5.0], kl_weight=1e-3)
    case_cprior_gen("g8_cprior_gen_ex5_C20_H7_f32", ex5, 40, 2.0, f32, 6, 20, 7, False, 0.0)
    case_cprior_gen("g8_cprior_gen_noex_C5_H1_f64", no_ex, 41, 2.0, f64, 5, 5, 1, False, 0.0)
    case_cprior_gen("g8_cprior_gen_ex12_C60_H30_sctx_ar_f32", ex12, 42, 2.0, f32, 4, 60, 30, True, 0.5)
    case_cprior_train("g8_cprior_train_ex5_B6_T21_h3", ex5, 43, 2.0, 6, 21, 3, [1, 7, 14])
    case_cprior_train("g8_cprior_train_ex12_B4_T40_h7", ex12, 44, 2.0, 4, 40, 7, [1, 7, 14, 30])


def case_losses():
    """G3: QuantileLoss on small tensors incl. the reference's own KATs
    (test_code/test_quantile_decoder.py:58-66, 70-94)."""
    torch.set_default_dtype(torch.float32)
    rng = np.random.Generator(np.random.PCG64(5))
    preds = rng.standard_normal((3, 4, 3, 5, 5)).astype(np.float32)
    target = rng.standard_normal((3, 4, 5, 5)).astype(np.float32)
    d = dict(preds=preds, target=target)
    for tag, w in (("w111", None), ("w515", [5.0, 1.0, 5.0])):
        fn = QuantileLoss([0.05, 0.5, 0.95], w)
        d["loss_" + tag] = np.float64(fn(torch.from_numpy(preds), torch.from_numpy(target)).item())
    fn = QuantileLoss([0.05, 0.5, 0.95])
    d["kat_equal"] = np.float64(fn(torch.ones(2, 1, 3, 5, 5), torch.ones(2, 1, 5, 5)).item())
    yt = torch.ones(1, 1, 2, 2)
    d["kat_under"] = np.float64(fn(torch.zeros(1, 1, 3, 2, 2), yt).item())
    d["kat_over"] = np.float64(fn(torch.full((1, 1, 3, 2, 2), 2.0), yt).item())
    np.savez_compressed(os.path.join(OUT, "g3_quantile_loss.npz"), **d)
    print("wrote g3_quantile_loss")


def save(name, cfg, d):
    d = dict(d)
    d["cfg_latent_dim"] = cfg["latent_dim"]
    d["cfg_ex_feats_dim"] = cfg["ex_feats_dim"]
    d["cfg_mem_layers"] = cfg["mem_layers"]
    d["cfg_interaction_layers"] = int(cfg.get("interaction_layers") or 0)
    d["cfg_mem_type"] = np.asarray(cfg.get("mem_type", "lstm"))
    d["cfg_compress_context"] = np.asarray(bool(cfg.get("compress_context", False)))
    if cfg.get("use_dense_surface", False):
        d["cfg_use_dense_surface"] = np.asarray(True)
        d["cfg_surface_hidden"] = np.asarray(list(cfg["surface_hidden"]))
        d["cfg_ctx_surface_hidden"] = np.asarray(list(cfg["ctx_surface_hidden"]))
    d["cfg_quantile_loss_weights"] = np.asarray(cfg.get("quantile_loss_weights") or [1.0, 1.0, 1.0])
    for k in ("ex_loss_on_ret_only", "ex_feats_loss_type", "kl_weight", "re_feat_weight"):
        d["cfg_" + k] = np.asarray(cfg[k])
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **d)
    print("wrote", name, os.path.getsize(path) // 1024, "KiB")


def main():
    os.makedirs(OUT, exist_ok=True)
    f32, f64 = np.float32, np.float64
    no_ex = vo.make_config(5, 0)
    ex5 = vo.make_config(5, 3)
    ex12 = vo.make_config(12, 3, quantile_loss_weights=[5.0, 1.0, 5.0])
    one_layer = vo.make_config(5, 0, mem_layers=1)

    # G1 / G3: forward + test_step
    case_forward("g1_fwd_noex_T6_h1_f32", no_ex, 0, 1.0, f32, 4, 6, 1)
    case_forward("g1_fwd_noex_T21_h1_f64", no_ex, 1, 2.0, f64, 4, 21, 1)
    case_forward("g1_fwd_ex5_T2_h1_f32", ex5, 2, 1.0, f32, 4, 2, 1)
    case_forward("g1_fwd_ex5_T21_h1_f32", ex5, 3, 2.0, f32, 4, 21, 1)
    case_forward("g1_fwd_ex5_T50_h7_f64", ex5, 4, 2.0, f64, 4, 50, 7)
    case_forward("g1_fwd_ex12_T50_h30_f32", ex12, 5, 2.0, f32, 4, 50, 30)
    case_forward("g1_fwd_ex12_T90_h30_f64", ex12, 6, 3.0, f64, 3, 90, 30)
    case_forward("g1_fwd_1layer_T6_h1_f32", one_layer, 7, 1.0, f32, 2, 6, 1)
    # G2: prior generation
    case_prior("g2_prior_noex_C5_H1_f32", no_ex, 10, 1.0, f32, 8, 5, 1)
    case_prior("g2_prior_noex_C20_H1_f64", no_ex, 11, 2.0, f64, 8, 20, 1)
    case_prior("g2_prior_ex5_C20_H7_f32", ex5, 12, 2.0, f32, 8, 20, 7)
    case_prior("g2_prior_ex12_C60_H30_f32", ex12, 13, 2.0, f32, 8, 60, 30)
    case_prior("g2_prior_ex12_C60_H90_f64", ex12, 14, 3.0, f64, 4, 60, 90)
    # G6 / G7: autoregressive
    case_ar("g6_ar_noex_C20_H30", no_ex, 20, 2.0, 4, 20, 30)
    case_ar("g6_ar_ex5_C20_H30", ex5, 21, 2.0, 4, 20, 30)
    case_ar_offset("g7_aroff_ex12_C60_H60x3", ex12, 22, 2.0, 2, 60, 60, 3)
    case_losses()
    main_training()
    main_cprior()
    main_ar_train()
    main_interaction()
    main_gru()
    main_round2()


def case_sampler():
    """G10: the reference's CustomBatchSampler / VolSurfaceDataSetRand with random.seed(0)."""
    import random
    from vae.datasets_randomized import CustomBatchSampler, VolSurfaceDataSetRand
    rng = np.random.Generator(np.random.PCG64(8000))
    N = 64
    surf = rng.standard_normal((N, 5, 5))
    ex = rng.standard_normal((N, 3))
    ds = VolSurfaceDataSetRand((surf, ex), min_seq_len=4, max_seq_len=10)
    random.seed(0)
    batches = list(CustomBatchSampler(ds, 16, 4))
    flat = np.asarray([i for b in batches for i in b], dtype=np.int64)
    sizes = np.asarray([len(b) for b in batches], dtype=np.int64)
    lens = np.asarray([ds[b[0]]["surface"].shape[0] for b in batches], dtype=np.int64)
    item = ds[3 * N + 60]                      # a tail window: shorter than its nominal length
    np.savez_compressed(os.path.join(OUT, "g10_sampler.npz"), surface=surf, ex_feats=ex, flat=flat, sizes=sizes,
                        lens=lens, n_items=np.int64(len(ds)), tail_len=np.int64(item["surface"].shape[0]),
                        tail_first=item["surface"][0].numpy())
    print("wrote g10_sampler")


def main_training():
    case_sampler()
    """G4 / G5: gradients (float64 reference)."""
    ex5 = vo.make_config(5, 3)
    no_ex = vo.make_config(5, 0)
    ex12w = vo.make_config(12, 3, quantile_loss_weights=[5.0, 1.0, 5.0], ex_loss_on_ret_only=False,
                           ex_feats_loss_type="l1", kl_weight=1e-3, re_feat_weight=0.5)
    case_grads("g4_grads_ex5_B8_T21_h1", ex5, 30, 2.0, 8, 21, 1, full=True)
    case_grads("g4_grads_noex_B5_T9_h3", no_ex, 31, 2.0, 5, 9, 3, full=False)
    case_grads("g4_grads_ex12_B4_T40_h7", ex12w, 32, 2.0, 4, 40, 7, full=False)
    case_multihorizon("g5_multih_ex5_B4_T50", vo.make_config(5, 3, quantile_loss_weights=[5.0, 1.0, 5.0]),
                      33, 2.0, 4, 50, [1, 7, 14, 30])


# ---- round 2: z = 0 "maximum likelihood" generation driver, fitted-GMM prior ------------------------------

def case_mle(name, cfg, seed, scale, n_days, C, check_ex):
    """G11: the reference's generate_surfaces_multiday_mle (generate_surfaces_max_likelihood.py:61-103) on a
    synthetic calendar: one z = 0 generation per day from the window [day - C, day).  Deterministic (the only
    noise the reference draws, the encoder's reparameterisation sample, is discarded by :700)."""
    import generate_surfaces_max_likelihood as ref_mle              # the reference's module
    torch.set_default_dtype(torch.float64)
    model, W = build_reference(cfg, seed, scale, np.float64)
    rng = np.random.Generator(np.random.PCG64(9000 + seed))
    X = cfg["ex_feats_dim"]
    vol = 0.2 + 0.1 * rng.standard_normal((n_days + C, 5, 5))
    ex = 0.01 * rng.standard_normal((n_days + C, max(X, 1)))
    mcfg = dict(cfg)
    model_data = {"model_config": mcfg, "model": model.state_dict()}
    (p05, p50, p95), exo = ref_mle.generate_surfaces_multiday_mle(model_data, ex, vol, C, n_days, 1, model_type=CVAEMemRand,
                                                                  check_ex_feats=check_ex, ctx_len=C)
    d = dict(vol=vol, ex=ex, p05=p05, p50=p50, p95=p95, ex_out=exo, ctx_len=C, start_day=C, n_days=n_days,
             check_ex_feats=int(check_ex), seed=seed, scale=scale)
    save(name, cfg, d)


def case_gmm(name, cfg, seed, scale, B, C, H, K):
    """G12: _sample_from_fitted_prior (:605-640) and get_surface_given_conditions(prior_mode="fitted")
    (:689-691) with torch.multinomial and torch.randn patched to stored draws."""
    import tempfile
    torch.set_default_dtype(torch.float64)
    model, W = build_reference(cfg, seed, scale, np.float64)
    rng = np.random.Generator(np.random.PCG64(9100 + seed))
    L, X = cfg["latent_dim"], cfg["ex_feats_dim"]
    means = rng.standard_normal((K, L))
    A = rng.standard_normal((K, L, L)) * 0.3
    cov = A @ A.transpose(0, 2, 1) + 0.05 * np.eye(L)[None]
    wts = rng.random(K) + 0.2
    wts /= wts.sum()
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "gmm.pt")
        torch.save({"means": torch.from_numpy(means), "covariances": torch.from_numpy(cov),
                    "weights": torch.from_numpy(wts), "n_components": K}, path)
        model.load_fitted_prior(path)
    surf, ex = synth(rng, B, C, X, np.float64)
    T = C + H
    comp = rng.choice(K, size=B * T, p=wts).astype(np.int64)
    eps = rng.standard_normal((B * T, L))
    c = {"surface": torch.from_numpy(surf)}
    if X:
        c["ex_feats"] = torch.from_numpy(ex)
    real_multinomial = torch.multinomial
    torch.multinomial = lambda w, num_samples, **kw: torch.from_numpy(comp)[:, None]
    try:
        with NoiseQueue() as nq:
            nq.push(eps)
            z = model._sample_from_fitted_prior(B, T)
        with NoiseQueue() as nq:
            nq.push(eps)
            nq.push(rng.standard_normal((B, C, L)))          # the encoder's own reparameterisation draw (discarded, :700)
            out = model.get_surface_given_conditions(c, horizon=H, prior_mode="fitted")
    finally:
        torch.multinomial = real_multinomial
    d = dict(surface=surf, means=means, covariances=cov, weights=wts, comp=comp, eps=eps, z=t2n(z), horizon=H,
             seed=seed, scale=scale)
    if X:
        d.update(ex_feats=ex, surf_out=t2n(out[0]), ex_out=t2n(out[1]))
    else:
        d.update(surf_out=t2n(out))
    save(name, cfg, d)


def main_round2():
    case_mle("g11_mle_ex5_C5_N40", vo.make_config(5, 3), 40, 2.0, 40, 5, True)
    case_mle("g11_mle_noex_C20_N33", vo.make_config(5, 0), 41, 2.0, 33, 20, False)
    case_mle("g11_mle_ex12_C60_N35", vo.make_config(12, 3), 42, 2.0, 35, 60, False)
    case_ar("g13_ar_mh3_ex5_C8_H4", vo.make_config(5, 3), 45, 2.0, 3, 8, 4, model_horizon=3)
    case_ar("g13_ar_mh2_noex_C5_H5", vo.make_config(5, 0), 46, 2.0, 4, 5, 5, model_horizon=2)
    case_gmm("g12_gmm_ex5_C8_H4_K3", vo.make_config(5, 3), 43, 2.0, 6, 8, 4, 3)
    case_gmm("g12_gmm_noex12_C20_H7_K5", vo.make_config(12, 0), 44, 2.0, 4, 20, 7, 5)


def case_optimizer_step(name, cfg, seed, scale, B, T, horizon, n_steps=2, lr=1e-5):
    """G15: whole train_step calls of the reference (cvae_with_mem_randomized.py:1101-1131: zero_grad, forward with the
    injected eps, pinball + ex-feature + KL loss, backward, clip_grad_norm_(1.0), optimizer.step()) with the optimizer
    every trainer builds, AdamW(model.parameters(), lr) (vae/utils.py:49), lr 1e-5, dropout 0, in FLOAT32 (the build's
    arithmetic and the reference's CPU path).  Stored per parameter: 64 sampled entries before and after each step, the
    clipped gradient of each step at the same entries (Adam's update is lr * m / (sqrt(v) + eps): entries whose
    gradient is below ~1e-6 are ill-conditioned in float32 and the test skips them), sum and norm after the last."""
    np_dtype = np.float32
    model, _ = build_reference(cfg, seed, scale, np_dtype)
    model.train()                                   # dropout is 0 in cfg
    rng = np.random.Generator(np.random.PCG64(15000 + seed))
    X, L = cfg["ex_feats_dim"], cfg["latent_dim"]
    surf, ex = synth(rng, B, T, X, np_dtype)
    eps = rng.standard_normal((n_steps, B, T, L)).astype(np_dtype)
    x = {"surface": torch.from_numpy(surf)}
    if X:
        x["ex_feats"] = torch.from_numpy(ex)
    model.horizon = horizon
    opt = torch.optim.AdamW(model.parameters(), lr=lr)
    d = dict(surface=surf, eps=eps, horizon=horizon, seed=seed, scale=scale, lr=np.float64(lr), n_steps=n_steps)
    if X:
        d["ex_feats"] = ex
    idx = {}
    for n, p_ in model.named_parameters():
        flat = p_.detach().numpy().ravel()
        idx[n] = np.linspace(0, flat.size - 1, num=min(flat.size, 64)).astype(np.int64)
        d["idx/" + n] = idx[n]
        d["w0/" + n] = flat[idx[n]].copy()
    for k in range(n_steps):
        with NoiseQueue() as nq:
            nq.push(eps[k])
            out = model.train_step(x, opt)
        d[f"loss{k + 1}"] = np.float64(t2n(out["loss"]).reshape(-1)[0])
        for n, p_ in model.named_parameters():
            d[f"w{k + 1}/" + n] = p_.detach().numpy().ravel()[idx[n]].copy()
            d[f"g{k + 1}/" + n] = p_.grad.detach().numpy().ravel()[idx[n]].copy()       # clipped (:1128)
    for n, p_ in model.named_parameters():
        w = p_.detach().numpy().astype(np.float64)
        d["sum/" + n] = np.float64(w.sum())
        d["norm/" + n] = np.float64(np.sqrt((w ** 2).sum()))
    torch.set_default_dtype(torch.float32)
    save(name, cfg, d)


def main_round3():
    case_optimizer_step("g15_optstep_ex5_B8_T21_h1", vo.make_config(5, 3, re_feat_weight=1.0, ex_loss_on_ret_only=True,
                                                                     ex_feats_loss_type="l2"), 60, 2.0, 8, 21, 1)
    case_optimizer_step("g15_optstep_noex12_B4_T30_h7", vo.make_config(12, 0), 61, 2.0, 4, 30, 7)


class UniformQueue:
    """Patches torch.rand_like (the legacy encoder's uniform reparameterisation noise, cvae_with_mem.py:157)."""

    def __init__(self, arr):
        self.arr = arr

    def __enter__(self):
        self._rand_like = torch.rand_like

        def rand_like(t, **kw):
            assert tuple(t.shape) == tuple(self.arr.shape)
            return torch.from_numpy(np.ascontiguousarray(self.arr)).to(t.dtype)

        torch.rand_like = rand_like
        return self

    def __exit__(self, *a):
        torch.rand_like = self._rand_like


def case_legacy(name, cfg, seed, scale, B, seq_len, ctx_len, full=True):
    """G14: the legacy fixed-length CVAEMem (vae/cvae_with_mem.py:466-738): forward with injected UNIFORM
    noise, test_step's MSE / ex / KL losses, their gradients w.r.t. every parameter (float64), and
    get_surface_given_conditions with a given z (context positions NOT replaced by the posterior mean)."""
    from vae.cvae_with_mem import CVAEMem
    np_dtype = np.float64
    torch.set_default_dtype(torch.float64)
    cfg = dict(cfg, seq_len=seq_len, ctx_len=ctx_len, quantiles=[0.5])
    model = CVAEMem(dict(cfg))
    W = vo.init_weights(cfg, seed, scale, np_dtype)
    sd = model.state_dict()
    assert set(sd) == set(W), (set(sd) ^ set(W))
    for k, v in W.items():
        assert tuple(sd[k].shape) == v.shape, k
        sd[k] = torch.from_numpy(v.copy())
    model.load_state_dict(sd, strict=True)
    model.train()
    rng = np.random.Generator(np.random.PCG64(9000 + seed))
    X, L = cfg["ex_feats_dim"], cfg["latent_dim"]
    surf, ex = synth(rng, B, seq_len, X, np_dtype)
    eps = rng.random((B, seq_len, L)).astype(np_dtype)
    x = {"surface": torch.from_numpy(surf)}
    if X:
        x["ex_feats"] = torch.from_numpy(ex)
    with UniformQueue(eps):
        out = model.forward(x)
    model.zero_grad()
    with UniformQueue(eps):
        losses = model.test_step(x)
    losses["loss"].backward()
    d = dict(surface=surf, eps=eps, seq_len=seq_len, ctx_len=ctx_len, seed=seed, scale=scale,
             recon=t2n(out[0]), z_mean=t2n(out[-3]), z_log_var=t2n(out[-2]), z=t2n(out[-1]))
    if X:
        d["ex_feats"] = ex
        d["ex_recon"] = t2n(out[1])
    for k, v in losses.items():
        d["loss_" + k] = np.float64(t2n(v).reshape(-1)[0])
    grads = {n: t2n(p.grad).copy() for n, p in model.named_parameters()}
    d["grad_total_norm"] = np.float64(np.sqrt(sum(float((g ** 2).sum()) for g in grads.values())))
    for n, g in grads.items():
        d["gnorm/" + n] = np.float64(np.sqrt((g ** 2).sum()))
        if full:
            d["grad/" + n] = g.astype(np.float32)
        else:
            flat = g.ravel()
            idx = np.linspace(0, flat.size - 1, num=min(flat.size, 64)).astype(np.int64)
            d["gidx/" + n] = idx
            d["gval/" + n] = flat[idx]
    zg = rng.standard_normal((B, seq_len, L)).astype(np_dtype)
    c = {"surface": torch.from_numpy(surf[:, :ctx_len])}
    if X:
        c["ex_feats"] = torch.from_numpy(ex[:, :ctx_len])
    model.eval()
    with torch.no_grad():
        g = model.get_surface_given_conditions(c, z=torch.from_numpy(zg.copy()))
    d["z_given"] = zg
    if X:
        d["surf_out"], d["ex_out"] = t2n(g[0]), t2n(g[1])
    else:
        d["surf_out"] = t2n(g)
    save(name, cfg, d)


def main_legacy():
    case_legacy("g14_legacy_ex5_T8_C5", vo.make_config(5, 3), 50, 2.0, 6, 8, 5, full=False)
    case_legacy("g14_legacy_noex_T6_C5", vo.make_config(5, 0), 51, 2.0, 5, 6, 5, full=False)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "training":
        os.makedirs(OUT, exist_ok=True)
        main_training()
    elif len(sys.argv) > 1 and sys.argv[1] == "interaction":
        os.makedirs(OUT, exist_ok=True)
        main_interaction()
    elif len(sys.argv) > 1 and sys.argv[1] == "gru":
        os.makedirs(OUT, exist_ok=True)
        main_gru()
    elif len(sys.argv) > 1 and sys.argv[1] == "artrain":
        os.makedirs(OUT, exist_ok=True)
        main_ar_train()
    elif len(sys.argv) > 1 and sys.argv[1] == "round2":
        os.makedirs(OUT, exist_ok=True)
        main_round2()
    elif len(sys.argv) > 1 and sys.argv[1] == "round3":
        os.makedirs(OUT, exist_ok=True)
        main_round3()
    elif len(sys.argv) > 1 and sys.argv[1] == "dense":
        os.makedirs(OUT, exist_ok=True)
        main_dense()
    elif len(sys.argv) > 1 and sys.argv[1] == "nocompress":
        os.makedirs(OUT, exist_ok=True)
        main_nocompress()
    elif len(sys.argv) > 1 and sys.argv[1] == "rnn":
        os.makedirs(OUT, exist_ok=True)
        main_rnn()
    elif len(sys.argv) > 1 and sys.argv[1] == "legacy":
        os.makedirs(OUT, exist_ok=True)
        main_legacy()
    elif len(sys.argv) > 1 and sys.argv[1] == "cprior":
        os.makedirs(OUT, exist_ok=True)
        main_cprior()
    else:
        main()
