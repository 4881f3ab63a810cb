"""Model-level compositions of the C-ABI calls: the reference's forward /
prior generation / MC loops expressed as kernel launches.  Everything here is
launch plumbing on device tensors; no arithmetic happens on the host.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from .engine import Engine


def encode_both(eng: Engine, surf: torch.Tensor, ex: Optional[torch.Tensor]):
    """-> z_mean (B,T,L), z_log_var (B,T,L) of the encoder on `surf`."""
    L = eng.L
    o = eng.encode(0, surf, ex)
    return o[..., :L].contiguous(), o[..., L:].contiguous()


def forward(eng: Engine, surf: torch.Tensor, ex: Optional[torch.Tensor], horizon: int,
            eps: Optional[torch.Tensor] = None, seed: int = 0):
    """CVAEMemRand.forward (cvae_with_mem_randomized.py:1022-1074): posterior
    z at all T positions, decode, return the last `horizon` steps.
    eps (B,T,L) injects the randn_like of :198; None -> Philox."""
    B, T = surf.shape[:2]
    C = T - horizon
    zm, zl = encode_both(eng, surf, ex)
    ctx = eng.encode(1, surf[:, :C], None if ex is None else ex[:, :C])
    r = eng.decode(B, T, C, ctx=ctx, zb=zm, lv=zl, eps=eps, seed=seed, S=1, want_z=True)
    return r["surf"], r["ex"], zm, zl, r["z"]


def decode_given_z(eng: Engine, ctx: torch.Tensor, z: torch.Tensor, C: int):
    """Decoder on cat([ctx padded with zeros, z]) (:704-715); returns steps >= C."""
    B, T = z.shape[:2]
    r = eng.decode(B, T, C, ctx=ctx, zb=z)
    return r["surf"], r["ex"]


def prior_generate(eng: Engine, surf_ctx: torch.Tensor, ex_ctx: Optional[torch.Tensor], horizon: int,
                   n_samples: int = 1, eps: Optional[torch.Tensor] = None, mu: float = 0.0, std: float = 1.0,
                   seed: int = 0, day0: int = 0, step0: int = 0, out_mode: int = 0):
    """get_surface_given_conditions (:642-715) for D context windows x S samples
    with the shared-prefix evaluation: the decoder state after
    the C context steps is computed once per window and broadcast to its samples.
    eps: (D*S, horizon, L) injected future latents (standard normal draws), or None -> Philox."""
    D, C = surf_ctx.shape[:2]
    zm, _ = encode_both(eng, surf_ctx, ex_ctx)
    ctx = eng.encode(1, surf_ctx, ex_ctx)
    pre = eng.decode(D, C, C, ctx=ctx, zb=zm, want_state=True, want_gates=n_samples > 1)
    r = eng.decode(D * n_samples, horizon, 0, state_in=pre["state"], state_div=n_samples, gate_in=pre["gates"],
                   eps=eps, mu0=mu, std0=std, seed=seed, S=n_samples, day0=day0, step0=step0,
                   out_mode=out_mode)
    return r["surf"], r["ex"], zm


def prior_mc_windows(eng: Engine, vol: torch.Tensor, ex: Optional[torch.Tensor], first_day: int, n_days: int,
                     ctx_len: int, horizon: int, n_samples: int, seed: int = 0,
                     qs: Sequence[float] = (5, 50, 95), channel: Optional[int] = 1,
                     chunk_days: int = 256, day_index0: Optional[int] = None, decode_events=None):
    """The Monte-Carlo backfill loop (eval_scripts/eval_utils.py:42-70 /
    generate_surfaces.py:60-102) for days [first_day, first_day+n_days): context
    window vol[day-C:day], S prior samples per day, reduced on device to
    np.percentile-style quantiles.  vol (N,5,5) and ex (N,X) live on the GPU and
    the overlapping windows are read in place (row stride = one day).
    Returns (n_days, horizon, nq, 5, 5) for one decoder channel (default p50),
    or (n_days, horizon, nq, Q, 5, 5) when channel is None.
    Philox counters use the GLOBAL day index (day_index0 + i), so any sharding of
    the day range gives identical numbers.
    decode_events: optional list; a (start, stop) hipevents.Event pair bracketing the
    decoder-LSTM kernel of every chunk is appended (benchmark instrumentation)."""
    assert first_day >= ctx_len and first_day + n_days <= vol.shape[0]
    vol = vol.to(device=eng.device, dtype=torch.float32).contiguous()
    if eng.X > 0:
        ex = ex.to(device=eng.device, dtype=torch.float32).contiguous()
    if day_index0 is None:
        day_index0 = first_day
    L, Q = eng.L, eng.Q
    # context prefix for ALL days in one go (three sequential-in-time kernels whose
    # parallelism is the day count): posterior means, context embedding, decoder state
    w0 = first_day - ctx_len                               # first calendar row of the first window
    sv = vol[w0:]
    se = ex[w0:] if eng.X > 0 else None
    enc = eng.encode_strided(0, sv, 25, 25, se, eng.X, eng.X, n_days, ctx_len)
    zm = enc[..., :L].contiguous()
    ctx = eng.encode_strided(1, sv, 25, 25, se, eng.X, eng.X, n_days, ctx_len)
    pre = eng.decode(n_days, ctx_len, ctx_len, ctx=ctx, zb=zm, want_state=True, want_gates=True)
    state, gates = pre["state"], pre["gates"]
    outs = []
    for d0 in range(0, n_days, chunk_days):
        nd = min(chunk_days, n_days - d0)
        if decode_events is not None:
            from .hipevents import Event
            ev = (Event(), Event())
            eng.lib.vsv_profile_next_decode(ev[0].h, ev[1].h)
            decode_events.append((ev[0], ev[1], nd * n_samples))
        r = eng.decode(nd * n_samples, horizon, 0, state_in=state[d0:d0 + nd], state_div=n_samples,
                       gate_in=gates[d0:d0 + nd], seed=seed, S=n_samples, day0=day_index0 + d0, out_mode=1)
        smp = r["surf"]                                    # (nd, H, Q*25, S)
        if channel is None:
            q = eng.percentile(smp, qs, inner=Q * 25).view(nd, horizon, len(qs), Q, 5, 5)
        else:
            sel = smp.view(nd, horizon, Q, 25, n_samples)[:, :, channel]
            q = eng.percentile(sel, qs, inner=25).view(nd, horizon, len(qs), 5, 5)
        outs.append(q)
    return torch.cat(outs, dim=0)
