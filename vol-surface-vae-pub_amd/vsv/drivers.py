"""The reference's generation DRIVERS on the HIP path: same function names, arguments, return
shapes and .npz products, but every date range is one batched device pass (context prefix for all
days at once, samples as kernel rows) instead of a Python loop of single-day model calls, and the
date range is sharded over the ranks of an initialised torch.distributed group (rank 0 writes).

Reference call sites:
  generate_surfaces.py:13-102            generate_surfaces_random / generate_surfaces_multiday (quantile decoder)
  generate_surfaces_max_likelihood.py:13-103   generate_surfaces_mle / generate_surfaces_multiday_mle (z = 0)
  generate_quantile_surfaces.py:70-132   the stochastic + MLE .npz products per model
  eval_scripts/eval_utils.py:8-70        generate_surfaces / generate_surfaces_multiday (num_vaes MC samples per day)
  experiments/backfill/context20/generate_vae_tf_sequences.py:59-181        generate_and_save_horizon
  experiments/backfill/context60/autoregressive/generate_vae_ar_sequences.py:47-152  generate_and_save_ar_sequences
The .npz key sets are the on-disk contract the analysis scripts read.
"""
from __future__ import annotations

from pathlib import Path
from typing import Optional

import numpy as np
import torch

from . import paths, sharded


def _arrays(model, vol_surface_data, ex_data):
    eng = model._engine()
    vol = torch.as_tensor(np.asarray(vol_surface_data), dtype=torch.float32, device=eng.device)
    ex = None
    if eng.X > 0:
        e = np.asarray(ex_data)
        if e.ndim == 1:
            e = e[:, None]
        ex = torch.as_tensor(e, dtype=torch.float32, device=eng.device)
    return eng, vol, ex


def _rank():
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def _shared_seed(model) -> int:
    """The Philox key of one generation product.  Philox counters are keyed by the GLOBAL day index, so a
    product is the same for any world size -- provided every rank uses the same key: each rank draws from
    its own torch generator (which keeps the generators in step), and rank 0's draw is broadcast."""
    import torch.distributed as dist
    seed = model._draw_seed()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        box = [seed]
        dist.broadcast_object_list(box, src=0)
        seed = int(box[0])
    return seed


def _sharded_pair(fn, n_days):
    """fn(lo, hi) -> (surf, ex|None) on this rank's day range; gathers both over the ranks."""
    has_ex = []

    def first(lo, hi):
        s, e = fn(lo, hi)
        has_ex.append(e)
        return s
    s = sharded.sharded_days(first, n_days)
    e = has_ex[0]
    if e is not None:
        e = sharded.gather_days(e, n_days)
    return s, e


def _model_from(model_data, model_type):
    if isinstance(model_data, torch.nn.Module):
        return model_data
    model = model_type(model_data["model_config"])
    model.load_weights(dict_to_load=model_data)
    return model


# ---- generate_surfaces.py ----------------------------------------------------------------------

def generate_surfaces_random(model, ex_data, vol_surface_data, day, ctx_len, num_vaes, use_ex_feats, check_ex_feats):
    """One day, one prior draw of the quantile surfaces (generate_surfaces.py:13-58):
    -> ((p05, p50, p95) each (5,5), ex_feat (X,) or None).  num_vaes is unused, as in the reference."""
    (p05, p50, p95), ex = generate_surfaces_multiday(model, ex_data, vol_surface_data, day, 1, num_vaes,
                                                     check_ex_feats=check_ex_feats, ctx_len=ctx_len)
    return (p05[0], p50[0], p95[0]), (ex[0] if (use_ex_feats and check_ex_feats) else None)


def generate_surfaces_multiday(model_data, ex_data, vol_surface_data, start_day, days_to_generate, num_vaes,
                               model_type=None, check_ex_feats=False, ctx_len=None):
    """generate_surfaces.py:60-102: for every day in [start_day, start_day + days_to_generate) ONE
    get_surface_given_conditions draw from the window [day-ctx_len, day) ->
    ((p05, p50, p95) each (days, 5, 5) float64, ex_feats (days, X) float64; zeros unless check_ex_feats).
    model_data: a checkpoint dict {"model_config", "model"} (a model is built, as in the reference)
    or an already built CVAEMemRand."""
    if model_type is None:
        from vae.cvae_with_mem_randomized import CVAEMemRand as model_type
    model = _model_from(model_data, model_type)
    if ctx_len is None:
        ctx_len = model.config["ctx_len"]
    eng, vol, ex = _arrays(model, vol_surface_data, ex_data)
    if eng.Q < 3:
        raise ValueError("generate_surfaces_multiday expects the 3-quantile decoder [p05, p50, p95]")
    seed = _shared_seed(model)

    def fn(lo, hi):
        s, e = paths.prior_windows(eng, vol, ex, start_day + lo, hi - lo, ctx_len, model.horizon, 1, seed=seed)
        return s[:, 0, 0], (None if e is None else e[:, 0, 0])
    s, e = _sharded_pair(fn, days_to_generate)
    s = s.double().cpu().numpy()
    n_ex = np.asarray(ex_data).shape[1] if ex_data is not None and np.asarray(ex_data).ndim > 1 else 1
    all_ex = np.zeros((days_to_generate, n_ex))
    if e is not None and check_ex_feats:
        all_ex[...] = e.double().cpu().numpy()
    return (s[:, 0], s[:, 1], s[:, 2]), all_ex


# ---- generate_surfaces_max_likelihood.py / generate_quantile_surfaces.py --------------------------

def generate_surfaces_mle(model, ex_data, vol_surface_data, day, ctx_len, num_vaes, use_ex_feats, check_ex_feats):
    """One day, z = 0 ("maximum likelihood") quantile surfaces (generate_surfaces_max_likelihood.py:13-59):
    -> ((p05, p50, p95) each (5,5), ex_feat (X,) or None).  num_vaes is unused, as in the reference."""
    (p05, p50, p95), ex = generate_surfaces_multiday_mle(model, ex_data, vol_surface_data, day, 1, num_vaes,
                                                         check_ex_feats=check_ex_feats, ctx_len=ctx_len)
    return (p05[0], p50[0], p95[0]), (ex[0] if (use_ex_feats and check_ex_feats) else None)


def generate_surfaces_multiday_mle(model_data, ex_data, vol_surface_data, start_day, days_to_generate, num_vaes,
                                   model_type=None, check_ex_feats=False, ctx_len=None):
    """generate_surfaces_max_likelihood.py:61-103: for every day in [start_day, start_day + days_to_generate)
    one get_surface_given_conditions(z = zeros((1, ctx_len + 1, L))) call on the window [day - ctx_len, day) --
    the future latent is 0, the context latents are the posterior means (:700), the horizon is 1.  Here: ONE
    batched device pass over the date range (no noise is drawn), sharded over the ranks.
    -> ((p05, p50, p95) each (days, 5, 5) float64, ex_feats (days, X) float64; zeros unless check_ex_feats)."""
    if model_type is None:
        from vae.cvae_with_mem_randomized import CVAEMemRand as model_type
    model = _model_from(model_data, model_type)
    if ctx_len is None:
        ctx_len = model.config["ctx_len"]
    eng, vol, ex = _arrays(model, vol_surface_data, ex_data)
    if eng.Q < 3:
        raise ValueError("generate_surfaces_multiday_mle expects the 3-quantile decoder [p05, p50, p95]")

    def fn(lo, hi):
        s, e = paths.prior_windows(eng, vol, ex, start_day + lo, hi - lo, ctx_len, 1, 1, mu=0.0, std=0.0)
        return s[:, 0, 0], (None if e is None else e[:, 0, 0])
    s, e = _sharded_pair(fn, days_to_generate)
    s = s.double().cpu().numpy()
    n_ex = np.asarray(ex_data).shape[1] if ex_data is not None and np.asarray(ex_data).ndim > 1 else 1
    all_ex = np.zeros((days_to_generate, n_ex))
    if e is not None and check_ex_feats:
        all_ex[...] = e.double().cpu().numpy()
    return (s[:, 0], s[:, 1], s[:, 2]), all_ex


def generate_quantile_products(model_data, ex_data, vol_surface_data, start_day, num_days, ctx_len, model_dir,
                               model_name, check_ex_feats=False, model_type=None):
    """One model's pass of generate_quantile_surfaces.py:70-132: the stochastic (z ~ N(0,1)) product
    {model_dir}/{model_name}_quantile_gen{ctx_len}.npz and the z = 0 product ..._quantile_mle_gen{ctx_len}.npz,
    each with surfaces_p05 / surfaces_p50 / surfaces_p95 (days, 5, 5) [+ ex_feats (days, X) when
    check_ex_feats].  Returns the two paths (None, None on ranks > 0)."""
    if model_type is None:
        from vae.cvae_with_mem_randomized import CVAEMemRand as model_type
    model = _model_from(model_data, model_type)
    products = []
    for fn, stem in ((generate_surfaces_multiday, "quantile_gen"), (generate_surfaces_multiday_mle, "quantile_mle_gen")):
        (p05, p50, p95), ex_feats = fn(model, ex_data, vol_surface_data, start_day, num_days, 1,
                                       model_type=model_type, check_ex_feats=check_ex_feats, ctx_len=ctx_len)
        if _rank() != 0:
            products.append(None)
            continue
        out = Path(model_dir)
        out.mkdir(parents=True, exist_ok=True)
        f = out / f"{model_name}_{stem}{ctx_len}.npz"
        if check_ex_feats:
            np.savez(f, surfaces_p05=p05, surfaces_p50=p50, surfaces_p95=p95, ex_feats=ex_feats)
        else:
            np.savez(f, surfaces_p05=p05, surfaces_p50=p50, surfaces_p95=p95)
        products.append(f)
    return tuple(products)


# ---- eval_scripts/eval_utils.py -----------------------------------------------------------------

def mc_generate_surfaces(model, ex_data, vol_surface_data, day, ctx_len, num_vaes, use_ex_feats, check_ex_feats):
    """eval_utils.generate_surfaces (:8-40): num_vaes prior draws for one day -> (num_vaes, 5, 5)
    [(num_vaes, Q, 5, 5) for a Q > 1 decoder, where the reference's reshape((5,5)) would raise],
    ex (num_vaes, X) or None."""
    s, e = mc_generate_surfaces_multiday(model, ex_data, vol_surface_data, day, 1, num_vaes,
                                         check_ex_feats=check_ex_feats, ctx_len=ctx_len)
    return s[0], (e[0] if (use_ex_feats and check_ex_feats) else None)


def mc_generate_surfaces_multiday(model_data, ex_data, vol_surface_data, start_day, days_to_generate, num_vaes,
                                  model_type=None, check_ex_feats=False, ctx_len=None):
    """eval_utils.generate_surfaces_multiday (:42-70): the literal Monte-Carlo product
    all_day_surfaces (days, num_vaes, 5, 5) [(days, num_vaes, Q, 5, 5) when Q > 1], all_day_ex_feats
    (days, num_vaes, X) -- one device pass with num_vaes kernel rows per day instead of
    days * num_vaes model calls with a host sync each."""
    if model_type is None:
        from vae.cvae_with_mem_randomized import CVAEMemRand as model_type
    model = _model_from(model_data, model_type)
    if ctx_len is None:
        ctx_len = model.config["ctx_len"]
    eng, vol, ex = _arrays(model, vol_surface_data, ex_data)
    seed = _shared_seed(model)

    from vae.cvae_with_mem import CVAEMem
    legacy = isinstance(model, CVAEMem)                  # z drawn at every position, seq_len - ctx_len = 1 day ahead

    def fn(lo, hi):
        gen = paths.legacy_prior_windows if legacy else paths.prior_windows
        s, e = gen(eng, vol, ex, start_day + lo, hi - lo, ctx_len, 1, num_vaes, seed=seed)
        return s[:, :, 0], (None if e is None else e[:, :, 0])
    s, e = _sharded_pair(fn, days_to_generate)
    s = s.double().cpu().numpy()
    if eng.Q == 1:
        s = s[:, :, 0]
    n_ex = np.asarray(ex_data).shape[1] if ex_data is not None and np.asarray(ex_data).ndim > 1 else 1
    all_ex = np.zeros((days_to_generate, num_vaes, n_ex))
    if e is not None and check_ex_feats:
        all_ex[...] = e.double().cpu().numpy()
    return s, all_ex


# ---- teacher-forcing / autoregressive sequence products -----------------------------------------

def generate_and_save_horizon(model, vol_surf, ex_data, period_start, period_end, horizon, period_name,
                              sampling_mode, context_len: int = 20, output_dir: Optional[str] = None):
    """generate_vae_tf_sequences.py:59-181.  sampling_mode "oracle": model.forward on context + target,
    z ~ q(z | context, target); "prior": get_surface_given_conditions on the context.  One draw per
    day; writes vae_tf_{period}_h{H}.npz with surfaces (n_days, H, 3, 5, 5), indices, quantiles, horizon,
    period_start, period_end, method, sampling_mode.  Returns the path (None on ranks > 0)."""
    if sampling_mode not in ("oracle", "prior"):
        raise ValueError(f"Unknown sampling_mode: {sampling_mode}")
    min_start = period_start + context_len
    max_start = period_end - horizon + 1
    n = max_start - min_start + 1
    if n <= 0:
        return None
    eng, vol, ex = _arrays(model, vol_surf, ex_data)
    seed = _shared_seed(model)
    gen = paths.oracle_windows if sampling_mode == "oracle" else paths.prior_windows

    def fn(lo, hi):
        s, _ = gen(eng, vol, ex, min_start + lo, hi - lo, context_len, horizon, 1, seed=seed)
        return s[:, 0], None
    s, _ = _sharded_pair(fn, n)
    if _rank() != 0:
        return None
    out = Path(output_dir if output_dir is not None
               else f"results/vae_baseline/predictions/autoregressive/{sampling_mode}")
    out.mkdir(parents=True, exist_ok=True)
    f = out / f"vae_tf_{period_name}_h{horizon}.npz"
    np.savez(f, surfaces=s.double().cpu().numpy(), indices=np.arange(min_start, max_start + 1),
             quantiles=[0.05, 0.50, 0.95], horizon=horizon, period_start=period_start, period_end=period_end,
             method="teacher_forcing", sampling_mode=sampling_mode)
    return f


def generate_and_save_ar_sequences(model, vol_surf, ex_data, period_start, period_end, total_horizon, period_name,
                                   sampling_mode, context_len: int = 60, output_dir: Optional[str] = None):
    """generate_vae_ar_sequences.py:47-152: offset-based autoregressive rollouts (3 chunks of 60 or 90
    days) for every valid date; writes vae_ar_{period}_{total}day.npz with surfaces (n_days, total, 3, 5, 5),
    indices, quantiles, total_horizon, period_start, period_end, method, sampling_mode, context_len."""
    if total_horizon == 180:
        ar_steps, horizon = 3, 60
    elif total_horizon == 270:
        ar_steps, horizon = 3, 90
    else:
        raise ValueError(f"Unsupported total_horizon: {total_horizon}")
    min_start = period_start + context_len
    max_start = period_end - total_horizon + 1
    n = max_start - min_start + 1
    if n <= 0:
        return None
    eng, vol, ex = _arrays(model, vol_surf, ex_data)
    seed = _shared_seed(model)

    def fn(lo, hi):
        s, _ = paths.ar_offset_windows(eng, vol, ex, min_start + lo, hi - lo, context_len, ar_steps, horizon, seed=seed)
        return s, None
    s, _ = _sharded_pair(fn, n)
    if _rank() != 0:
        return None
    out = Path(output_dir if output_dir is not None
               else f"results/context60_baseline/predictions/autoregressive_multi_step/{sampling_mode}")
    out.mkdir(parents=True, exist_ok=True)
    f = out / f"vae_ar_{period_name}_{total_horizon}day.npz"
    np.savez(f, surfaces=s.double().cpu().numpy(), indices=np.arange(min_start, max_start + 1),
             quantiles=[0.05, 0.50, 0.95], total_horizon=total_horizon, period_start=period_start,
             period_end=period_end, method="autoregressive_multistep", sampling_mode=sampling_mode,
             context_len=context_len)
    return f
