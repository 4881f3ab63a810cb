"""Model-level compositions of the C-ABI calls: the reference's forward /
prior generation / MC loops expressed as kernel launches.  Everything here is
launch plumbing on device tensors; no arithmetic happens on the host.
"""
from __future__ import annotations

import os
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


SPLIT_MAX_DAYS = 6000      # beyond: the fused three-segment schedule (see _context_prefix)

# The T-step decode / rollout kernels (dec_seq_kernel, enc_seq_kernel) run ONE wave per 32-row tile and SIMD: 1024 row
# tiles (32768 rows) are resident at a time and a tile lives for the whole sequence, so a launch is quantised in rounds
# of 1024 tiles.  Chunks are sized to fill their last round.
ROUND_ROWS = 1024 * 32


def _arm_decode_events(eng: Engine, decode_events, rows: int):
    """Benchmark instrumentation: a (start, stop) hipevents.Event pair around the NEXT vsv_decode launch."""
    if decode_events is None:
        return
    from .hipevents import Event
    ev = (Event(), Event())
    eng.lib.vsv_profile_next_decode(ev[0].h, ev[1].h)
    decode_events.append((ev[0], ev[1], rows))


def _context_prefix(eng: Engine, sv: torch.Tensor, se: Optional[torch.Tensor], n_days: int, ctx_len: int,
                    n_seg: Optional[int] = None):
    """Decoder state after the C context steps of the windows sv[d : d + C] (d < n_days), and the first
    future step's shared gates: posterior mean and context embedding from the two encoders, then the
    decoder over (ctx_t, z_mean_t).
    Every recurrent launch is a chain of C sequential steps, each bound by the matrix pipe of the one CU a row tile
    sits on.  Two schedules (same products in the same order, bit-equal results):
      * layer stages (the default for two-layer nn.LSTM memories up to SPLIT_MAX_DAYS days): every stack as one launch
        per layer -- layer 1's input product b + W_ih . h0 leaves the time loop for one GEMM over all steps -- and the
        recurrent launches on 16-day tiles (lstm_rec16_kernel: twice the CUs, half the matrix time per step, W_hh in
        registers), issued by the native vsv_context_prefix on two streams (posterior encoder + decoder on the
        caller's, context encoder beside them).  Context 60: 1.33 / 1.73 / 2.28 / 2.70 / 4.30 ms at 500 / 2000 / 3000 /
        4000 / 6000 days;
      * fused stacks (one two-layer launch per stack and time segment) on three streams in three segments, the
        encoders' segment i+1 beside the decoder's segment i: 2.5 - 2.6 ms up to 2000 days, 2.96 ms at 4000, 4.38 at
        6000, 5.05 at 8000 (5.27 in layer stages: every launch then takes several rounds of workgroups either way).
    VSV_PREFIX_LAYERS = 0 / 1 / py forces the fused / layer-stage / layer-stage-driven-from-Python schedule,
    VSV_PREFIX_SEGMENTS the segment count, VSV_PREFIX_STREAMS = 2 / 3 / 4 / 6 the worker streams of the layer stages
    (with HIP's default of four hardware queues more than two busy streams may share a queue and serialise: 1.2 or
    1.9 ms at 500 days from one process to the next with four)."""
    L = eng.L
    mode = os.environ.get("VSV_PREFIX_LAYERS", "")
    two_lstm = eng.dims.mem_layers == 2 and eng.dims.mem_type == 0
    split = eng.dims.mem_layers == 2 and (mode in ("1", "py") or (mode == "" and two_lstm and n_days <= SPLIT_MAX_DAYS))
    auto = split and mode == ""
    if n_seg is None:
        n_seg = int(os.environ.get("VSV_PREFIX_SEGMENTS", "0")) or (1 if auto else (4 if split else 3))
    if (n_seg <= 1 and not split) or n_days > eng.MAX_WINDOWS or ctx_len < 4 * n_seg:
        enc, ctx = eng.encode_both_windows(sv, se, n_days, ctx_len)
        zm = enc[..., :L].contiguous()
        pre = eng.decode(n_days, ctx_len, ctx_len, ctx=ctx, zb=zm, want_state=True, want_gates=True)
        return pre["state"], pre["gates"]
    dev = eng.device
    main = torch.cuda.current_stream(dev)
    if split and mode == "py":
        return _context_prefix_layers(eng, sv, se, n_days, ctx_len, n_seg, main)
    if split:
        return eng.context_prefix(sv, se, n_days, ctx_len, max(n_seg, 1),
                                  int(os.environ.get("VSV_PREFIX_STREAMS", "2" if n_seg <= 1 else "4")))
    if eng._side is None:
        eng._side = torch.cuda.Stream(dev)
    if getattr(eng, "_third", None) is None:
        eng._third = torch.cuda.Stream(dev)
    side, third = eng._side, eng._third
    side.wait_stream(main)
    third.wait_stream(main)
    st_e = st_c = st_d = None
    pre = None
    for i in range(n_seg):
        t0, t1 = (i * ctx_len) // n_seg, ((i + 1) * ctx_len) // n_seg
        last = i == n_seg - 1
        v, e = sv[t0:], (se[t0:] if se is not None else None)
        with torch.cuda.stream(side):
            r = eng.encode_windows(1, v, e, n_days, t1 - t0, state_in=st_c, want_state=not last)
            ctx_i, st_c = r if not last else (r, None)
            ev_side = side.record_event()
        r = eng.encode_windows(0, v, e, n_days, t1 - t0, state_in=st_e, want_state=not last)
        enc_i, st_e = r if not last else (r, None)
        zm_i = enc_i[..., :L].contiguous()
        ev_main = main.record_event()
        third.wait_event(ev_side)
        third.wait_event(ev_main)
        with torch.cuda.stream(third):
            pre = eng.decode(n_days, t1 - t0, t1 - t0, ctx=ctx_i, zb=zm_i, state_in=st_d, state_div=1,
                             want_state=True, want_gates=last)
            st_d = pre["state"]
        ctx_i.record_stream(third)
        zm_i.record_stream(third)
    main.wait_stream(third)
    main.wait_stream(side)
    pre["state"].record_stream(main)
    pre["gates"].record_stream(main)
    return pre["state"], pre["gates"]


def _context_prefix_layers(eng: Engine, sv, se, n_days: int, ctx_len: int, n_seg: int, main):
    """_context_prefix with every stack split into its two layers (see there)."""
    L, dev = eng.L, eng.device
    pool = getattr(eng, "_stage_streams", None)
    if pool is None:
        pool = eng._stage_streams = [torch.cuda.Stream(dev) for _ in range(6)]
    e0, e1, c0, c1, d0, d1 = pool
    for s in pool:
        s.wait_stream(main)
    f32 = dict(dtype=torch.float32, device=dev)
    keep = []                                                  # every buffer stays referenced until the streams have joined
    nse, nsd = eng.enc_state_floats(), eng.state_floats
    st_e = st_c = st_d = None
    gates = torch.empty((n_days, eng.gate_floats), **f32)
    for i in range(n_seg):
        t0, t1 = (i * ctx_len) // n_seg, ((i + 1) * ctx_len) // n_seg
        Ts, last = t1 - t0, i == n_seg - 1
        v, e = sv[t0:], (se[t0:] if se is not None else None)
        ne = None if last else torch.empty((n_days, nse), **f32)      # the encoders' final states are not needed
        nc = None if last else torch.empty((n_days, nse), **f32)
        nd = torch.empty((n_days, nsd), **f32)
        outs = {}
        for which, s0, s1, st_in, st_out in ((0, e0, e1, st_e, ne), (1, c0, c1, st_c, nc)):
            with torch.cuda.stream(s0):
                ws = eng.encode_windows_layer0(which, v, e, n_days, Ts, st_in, st_out)
                ev = s0.record_event()
            s1.wait_event(ev)
            with torch.cuda.stream(s1):
                o = eng.encode_windows_layer1(which, n_days, Ts, st_in, st_out, ws)
                outs[which] = o[..., :L].contiguous() if which == 0 else o
                evo = s1.record_event()
            d0.wait_event(evo)
            keep += [ws, o]
        with torch.cuda.stream(d0):
            wsd = eng.decode_prefix_layer(0, n_days, Ts, outs[1], outs[0], st_d, nd, gates if last else None)
            ev = d0.record_event()
        d1.wait_event(ev)
        with torch.cuda.stream(d1):
            eng.decode_prefix_layer(1, n_days, Ts, None, None, st_d, nd, gates if last else None, ws=wsd)
        keep += [wsd, outs[0], outs[1], st_e, st_c, st_d]
        st_e, st_c, st_d = ne, nc, nd
    for s in pool:
        main.wait_stream(s)
    del keep
    return st_d, gates


def _empty_quantiles(eng: Engine, horizon: int, nq: int, channel: Optional[int]) -> torch.Tensor:
    """The reduced product of an empty day range (a rank of a sharded run that owns no day)."""
    shape = (0, horizon, nq, 5, 5) if channel is not None else (0, horizon, nq, eng.Q, 5, 5)
    return torch.empty(shape, dtype=torch.float32, device=eng.device)


def _empty_samples(eng: Engine, lead: tuple):
    s = torch.empty(lead + (eng.Q, 5, 5), dtype=torch.float32, device=eng.device)
    e = torch.empty(lead + (eng.X,), dtype=torch.float32, device=eng.device) if eng.X > 0 else None
    return s, e


def prior_mc_windows(eng: Engine, vol: torch.Tensor, ex: Optional[torch.Tensor], first_day: int, n_days: int,
                     ctx_len: int, horizon: int, n_samples: int, seed: int = 0,
                     qs: Sequence[float] = (5, 50, 95), channel: Optional[int] = 1,
                     chunk_days: Optional[int] = None, day_index0: Optional[int] = None, decode_events=None):
    """The Monte-Carlo backfill loop (eval_scripts/eval_utils.py:42-70 /
    generate_surfaces.py:60-102) for days [first_day, first_day+n_days): context
    window vol[day-C:day], S prior samples per day, reduced on device to
    np.percentile-style quantiles.  vol (N,5,5) and ex (N,X) live on the GPU and
    the overlapping windows are read in place (row stride = one day).
    Returns (n_days, horizon, nq, 5, 5) for one decoder channel (default p50),
    or (n_days, horizon, nq, Q, 5, 5) when channel is None.
    Philox counters use the GLOBAL day index (day_index0 + i), so any sharding or
    chunking of the day range gives identical numbers.  chunk_days: days per decode
    launch (None: sized to 2 GiB of samples).
    decode_events: optional list; a (start, stop) hipevents.Event pair bracketing the
    decoder-LSTM kernel of every chunk is appended (benchmark instrumentation)."""
    assert first_day >= ctx_len and first_day + n_days <= vol.shape[0]
    vol = vol.to(device=eng.device, dtype=torch.float32).contiguous()
    if eng.X > 0:
        ex = ex.to(device=eng.device, dtype=torch.float32).contiguous()
    if day_index0 is None:
        day_index0 = first_day
    L, Q = eng.L, eng.Q
    if n_days == 0:
        return _empty_quantiles(eng, horizon, len(qs), channel)
    if not chunk_days:
        # as many days per decode launch as fit 2 GiB of samples: a launch ends with a partly filled
        # round of workgroups (512 resident at a time), so fewer, longer launches waste less
        # (4000 days x 1000 samples: one launch 12.5 ms vs eight of 1.65 ms = 13.2 ms)
        chunk_days = max(1, (1 << 31) // (n_samples * horizon * (Q if channel is None else 1) * 25 * 4))
    # context prefix for ALL days in one go (three sequential-in-time kernels whose
    # parallelism is the day count): posterior means, context embedding, decoder state
    w0 = first_day - ctx_len                               # first calendar row of the first window
    sv = vol[w0:]
    se = ex[w0:] if eng.X > 0 else None
    state, gates = _context_prefix(eng, sv, se, n_days, ctx_len)
    outs = []
    for d0 in range(0, n_days, chunk_days):
        nd = min(chunk_days, n_days - d0)
        _arm_decode_events(eng, decode_events, nd * n_samples)
        # one channel wanted: the decoder neither computes nor stores the others
        r = eng.decode(nd * n_samples, horizon, 0, state_in=state[d0:d0 + nd], state_div=n_samples,
                       gate_in=gates[d0:d0 + nd], seed=seed, S=n_samples, day0=day_index0 + d0,
                       out_mode=1 if channel is None else eng.OUT_CHANNEL(channel))
        smp = r["surf"]                                    # (nd, H, Q*25 | 25, S)
        # reduced on the side stream while the next chunk decodes
        more = d0 + nd < n_days                               # (a next chunk to decode beside this reduction)
        if channel is None:
            q = eng.reduce_async(smp, qs, Q * 25, None, 1, overlap=more).view(nd, horizon, len(qs), Q, 5, 5)
        else:
            q = eng.reduce_async(smp, qs, 25, None, 1, overlap=more).view(nd, horizon, len(qs), 5, 5)
        outs.append(q)
    eng.join_side()
    return torch.cat(outs, dim=0)


def prior_windows(eng: Engine, vol: torch.Tensor, ex: Optional[torch.Tensor], first_day: int, n_days: int,
                  ctx_len: int, horizon: int, n_samples: int = 1, seed: int = 0,
                  day_index0: Optional[int] = None, chunk_rows: int = 1 << 19, mu: float = 0.0, std: float = 1.0):
    """The per-day generation loop of generate_surfaces.py:60-102 / eval_utils.py:42-70 /
    generate_vae_tf_sequences.py (sampling_mode "prior") WITHOUT the reduction: for every day in
    [first_day, first_day + n_days), n_samples prior draws from the window vol[day-C:day].
    Returns (n_days, n_samples, horizon, Q, 5, 5)[, (n_days, n_samples, horizon, X)] -- the raw
    samples in the reference's layout, on device.  Philox keyed by the global day index.
    Future latents are N(mu, std); std = 0 is the z = mu "maximum likelihood" generation of
    generate_surfaces_max_likelihood.py:13-58 (z = 0)."""
    assert first_day >= ctx_len and first_day + n_days <= vol.shape[0]
    vol = vol.to(device=eng.device, dtype=torch.float32).contiguous()
    if eng.X > 0:
        ex = ex.to(device=eng.device, dtype=torch.float32).contiguous()
    if day_index0 is None:
        day_index0 = first_day
    L, Q, X, S = eng.L, eng.Q, eng.X, n_samples
    if n_days == 0:
        return _empty_samples(eng, (0, S, horizon))
    w0 = first_day - ctx_len
    enc, ctx = eng.encode_both_windows(vol[w0:], ex[w0:] if X > 0 else None, n_days, ctx_len)
    zm = enc[..., :L].contiguous()
    pre = eng.decode(n_days, ctx_len, ctx_len, ctx=ctx, zb=zm, want_state=True, want_gates=S > 1)
    out_s = torch.empty((n_days, S, horizon, Q, 5, 5), dtype=torch.float32, device=eng.device)
    out_e = torch.empty((n_days, S, horizon, X), dtype=torch.float32, device=eng.device) if X > 0 else None
    chunk_days = max(1, chunk_rows // max(1, S * horizon))
    for d0 in range(0, n_days, chunk_days):
        nd = min(chunk_days, n_days - d0)
        r = eng.decode(nd * S, horizon, 0, state_in=pre["state"][d0:d0 + nd], state_div=S,
                       gate_in=None if pre["gates"] is None else pre["gates"][d0:d0 + nd],
                       seed=seed, S=S, day0=day_index0 + d0, out_mode=0, mu0=float(mu), std0=float(std))
        out_s[d0:d0 + nd] = r["surf"].view(nd, S, horizon, Q, 5, 5)
        if X > 0:
            out_e[d0:d0 + nd] = r["ex"].view(nd, S, horizon, X)
    return out_s, out_e


def legacy_prior_windows(eng: Engine, vol: torch.Tensor, ex: Optional[torch.Tensor], first_day: int, n_days: int,
                         ctx_len: int, horizon: int, n_samples: int = 1, seed: int = 0,
                         day_index0: Optional[int] = None, chunk_rows: int = 1 << 16, mu: float = 0.0, std: float = 1.0):
    """The Monte-Carlo loop of eval_scripts/eval_utils.py:8-70 for the LEGACY model (CVAEMem.get_surface_given_conditions,
    cvae_with_mem.py:520-562): z ~ N(mu, std) at EVERY one of the C + H positions -- the context latents are not the
    posterior mean -- so a day's samples share the context embedding only and each decodes all C + H steps.
    Raw samples (n_days, n_samples, horizon, Q, 5, 5)[, (n_days, n_samples, horizon, X)]; Philox keyed by the
    global day index."""
    T = ctx_len + horizon
    assert first_day >= ctx_len and first_day + n_days <= vol.shape[0]
    vol = vol.to(device=eng.device, dtype=torch.float32).contiguous()
    if eng.X > 0:
        ex = ex.to(device=eng.device, dtype=torch.float32).contiguous()
    if day_index0 is None:
        day_index0 = first_day
    Q, X, S = eng.Q, eng.X, n_samples
    if n_days == 0:
        return _empty_samples(eng, (0, S, horizon))
    w0 = first_day - ctx_len
    ctx = eng.encode_windows(1, vol[w0:], ex[w0:] if X > 0 else None, n_days, ctx_len)
    out_s = torch.empty((n_days, S, horizon, Q, 5, 5), dtype=torch.float32, device=eng.device)
    out_e = torch.empty((n_days, S, horizon, X), dtype=torch.float32, device=eng.device) if X > 0 else None
    chunk_days = max(1, chunk_rows // max(1, S))
    for d0 in range(0, n_days, chunk_days):
        nd = min(chunk_days, n_days - d0)
        r = eng.decode(nd * S, T, ctx_len, ctx=ctx[d0:d0 + nd], ctx_div=S, seed=seed, S=S, day0=day_index0 + d0,
                       out_mode=0, mu0=float(mu), std0=float(std))
        out_s[d0:d0 + nd] = r["surf"].view(nd, S, horizon, Q, 5, 5)
        if X > 0:
            out_e[d0:d0 + nd] = r["ex"].view(nd, S, horizon, X)
    return out_s, out_e


def oracle_windows(eng: Engine, vol: torch.Tensor, ex: Optional[torch.Tensor], first_day: int, n_days: int,
                   ctx_len: int, horizon: int, n_samples: int = 1, seed: int = 0,
                   day_index0: Optional[int] = None, chunk_rows: int = 1 << 16):
    """Teacher-forcing generation with posterior ("oracle") sampling, generate_vae_tf_sequences.py:109-123:
    model.forward on vol[day-C : day+H] for every day, z sampled at all C+H positions.  Raw samples
    (n_days, n_samples, horizon, Q, 5, 5)[, (n_days, n_samples, horizon, X)]."""
    T = ctx_len + horizon
    assert first_day >= ctx_len and first_day + n_days + horizon <= vol.shape[0] + 0
    vol = vol.to(device=eng.device, dtype=torch.float32).contiguous()
    if eng.X > 0:
        ex = ex.to(device=eng.device, dtype=torch.float32).contiguous()
    if day_index0 is None:
        day_index0 = first_day
    L, Q, X, S = eng.L, eng.Q, eng.X, n_samples
    if n_days == 0:
        return _empty_samples(eng, (0, S, horizon))
    w0 = first_day - ctx_len
    enc, ctx = eng.encode_both_windows(vol[w0:], ex[w0:] if X > 0 else None, n_days, T, ctx_len)
    zm, zl = enc[..., :L].contiguous(), enc[..., L:].contiguous()
    out_s = torch.empty((n_days, S, horizon, Q, 5, 5), dtype=torch.float32, device=eng.device)
    out_e = torch.empty((n_days, S, horizon, X), dtype=torch.float32, device=eng.device) if X > 0 else None
    chunk_days = max(1, chunk_rows // max(1, S))
    for d0 in range(0, n_days, chunk_days):
        nd = min(chunk_days, n_days - d0)
        r = eng.decode(nd * S, T, ctx_len, ctx=ctx[d0:d0 + nd], ctx_div=S, zb=zm[d0:d0 + nd], lv=zl[d0:d0 + nd],
                       z_div=S, seed=seed, S=S, day0=day_index0 + d0, out_mode=0)
        out_s[d0:d0 + nd] = r["surf"].view(nd, S, horizon, Q, 5, 5)
        if X > 0:
            out_e[d0:d0 + nd] = r["ex"].view(nd, S, horizon, X)
    return out_s, out_e


def ar_offset_windows(eng: Engine, vol: torch.Tensor, ex: Optional[torch.Tensor], first_day: int, n_days: int,
                      ctx_len: int, ar_steps: int, horizon: int, seed: int = 0,
                      day_index0: Optional[int] = None, chunk_days: int = 4096):
    """generate_vae_ar_sequences.py:47-152: one generate_autoregressive_offset rollout per day from
    the window vol[day-C:day].  Returns (n_days, ar_steps*horizon, Q, 5, 5)[, (n_days, ar_steps*horizon, X)]."""
    assert first_day >= ctx_len and first_day + n_days <= vol.shape[0]
    vol = vol.to(device=eng.device, dtype=torch.float32).contiguous()
    X = eng.X
    if X > 0:
        ex = ex.to(device=eng.device, dtype=torch.float32).contiguous()
    if day_index0 is None:
        day_index0 = first_day
    if n_days == 0:
        return _empty_samples(eng, (0, ar_steps * horizon))
    outs_s, outs_e = [], []
    for d0 in range(0, n_days, chunk_days):
        nd = min(chunk_days, n_days - d0)
        w0 = first_day + d0 - ctx_len
        win = vol[w0:w0 + nd + ctx_len - 1].reshape(-1, 25).unfold(0, ctx_len, 1).permute(0, 2, 1)    # (nd, C, 25)
        wex = ex[w0:w0 + nd + ctx_len - 1].unfold(0, ctx_len, 1).permute(0, 2, 1) if X > 0 else None
        s, e = ar_offset_generate(eng, win.reshape(nd, ctx_len, 5, 5).contiguous(), None if wex is None else wex.contiguous(),
                                  ar_steps, horizon, seed=seed, day0=day_index0 + d0)
        outs_s.append(s)
        outs_e.append(e)
    return torch.cat(outs_s, dim=0), (torch.cat(outs_e, dim=0) if X > 0 else None)


def ar_generate(eng: Engine, surf_ctx: torch.Tensor, ex_ctx: Optional[torch.Tensor], horizon: int,
                n_samples: int = 1, eps: Optional[torch.Tensor] = None, seed: int = 0, day0: int = 0,
                share_prefix: Optional[bool] = None, reduce_qs: Optional[Sequence[float]] = None,
                reduce_channel: Optional[int] = 1, decode_events=None):
    """generate_autoregressive_sequence (cvae_with_mem_randomized.py:717-807) for D
    context windows x S samples: every row rolls its OWN window -- each step is a
    one-day prior generation (:776), whose p50 channel (:786/:794) and predicted
    ex-features (:788) slide into the row's context (_update_context, :925-967).
    The windows live in one (rows, C+horizon, 25) ring so a step reads its context
    in place (time stride = one day) and the kernels write the feedback day behind it.
    share_prefix (default: S > 1): at step k < C the first C-k days of every row's window are still
    the day's ORIGINAL context, identical for its S samples, so both encoders and the decoder run
    them once per day and the rows continue from those states over their own k generated days
    (the same products in the same order: results do not change).
    eps: (D*S, horizon, L) injected future latents, or None -> Philox keyed by
    (seed; sample, day0 + day, step).  Returns (D*S, horizon, Q, 5, 5)[, (D*S, horizon, X)].
    reduce_qs: the Monte-Carlo product instead of the samples -- every step's decode writes its surfaces
    sample-minor (one decoder channel, or all with reduce_channel None), the step's percentiles over the S
    rollouts of a day are taken on the side stream while the next step's encoders run, and the samples are
    never stored: returns (D, horizon, nq, 5, 5) or (D, horizon, nq, Q, 5, 5), and None."""
    D, C = surf_ctx.shape[:2]
    S, R, L, X, Q = n_samples, D * n_samples, eng.L, eng.X, eng.Q
    dev = eng.device
    TT = C + horizon
    share = (S > 1) if share_prefix is None else bool(share_prefix)
    ctx0 = surf_ctx.to(dev, torch.float32).reshape(D, C, 25).contiguous()
    ex0 = ex_ctx.to(dev, torch.float32).reshape(D, C, X).contiguous() if X > 0 else None
    buf = torch.empty((R, TT, 25), dtype=torch.float32, device=dev)
    buf[:, :C] = ctx0.reshape(D, 1, C, 25).expand(D, S, C, 25).reshape(R, C, 25)
    exb = None
    if X > 0:
        exb = torch.empty((R, TT, X), dtype=torch.float32, device=dev)
        exb[:, :C] = ex0.reshape(D, 1, C, X).expand(D, S, C, X).reshape(R, C, X)
    reducing = reduce_qs is not None
    om = 0
    if reducing:
        # the fed-back p50 channel is needed in any case: decoded alone when it is also the reduced one
        om = eng.OUT_CHANNEL(1) if reduce_channel == 1 else 1
        out_q = []
        out_s = out_e = None
    else:
        out_s = torch.empty((R, horizon, Q, 5, 5), dtype=torch.float32, device=dev)
        out_e = torch.empty((R, horizon, X), dtype=torch.float32, device=dev) if X > 0 else None
    if eps is not None:
        eps = eps.to(dev, torch.float32).contiguous()
        assert eps.shape == (R, horizon, L)
    # per generated day and encoder: the position-independent part (embedding + first-layer input product),
    # computed once when the day is generated and read by every later step of the rollout
    ptab = None
    if share and horizon > 1:
        ptab = [torch.empty((horizon - 1, eng.pre_table_floats(R)), dtype=torch.float32, device=dev) for _ in range(2)]
    for k in range(horizon):
        ek = None if eps is None else eps[:, k:k + 1].contiguous()
        if share and k >= C:
            # the window is all generated days k-C .. k-1
            enc = eng.encode_from_pre(0, ptab[0][k - C:], R, C)
            ctx = eng.encode_from_pre(1, ptab[1][k - C:], R, C)
            _arm_decode_events(eng, decode_events, R)
            r = eng.decode(R, C + 1, C, ctx=ctx, zb=enc[..., :L].contiguous(), eps=ek, eps_t0=C,
                           seed=seed, S=S, day0=day0, step0=k, out_mode=om)
        elif share and k < C:
            ns = C - k                                         # days of the original context still in the window
            enc_s, st_e = eng.encode_state(0, ctx0[:, k:], C * 25, 25, ex0[:, k:] if X > 0 else None, C * X, X,
                                           D, ns, want_state=k > 0)
            ctx_s, st_c = eng.encode_state(1, ctx0[:, k:], C * 25, 25, ex0[:, k:] if X > 0 else None, C * X, X,
                                           D, ns, want_state=k > 0)
            pre = eng.decode(D, ns, ns, ctx=ctx_s, zb=enc_s[..., :L].contiguous(), want_state=True, want_gates=k == 0)
            if k == 0:
                _arm_decode_events(eng, decode_events, R)
                r = eng.decode(R, 1, 0, state_in=pre["state"], state_div=S, gate_in=pre["gates"],
                               eps=ek, eps_t0=0, seed=seed, S=S, day0=day0, step0=0, out_mode=om)
            else:
                # the rows' own generated days 0..k-1, from their tables
                enc_r = eng.encode_from_pre(0, ptab[0], R, k, state_in=st_e, state_div=S)
                ctx_r = eng.encode_from_pre(1, ptab[1], R, k, state_in=st_c, state_div=S)
                _arm_decode_events(eng, decode_events, R)
                r = eng.decode(R, k + 1, k, ctx=ctx_r, zb=enc_r[..., :L].contiguous(),
                               state_in=pre["state"], state_div=S,
                               eps=ek, eps_t0=k, seed=seed, S=S, day0=day0, step0=k, out_mode=om)
        else:
            win = buf[:, k:]
            wex = exb[:, k:] if X > 0 else None
            enc, ctx = eng.encode_both_strided(win, TT * 25, 25, wex, TT * X, X, R, C)
            zm = enc[..., :L].contiguous()
            _arm_decode_events(eng, decode_events, R)
            r = eng.decode(R, C + 1, C, ctx=ctx, zb=zm, eps=ek, eps_t0=C,
                           seed=seed, S=S, day0=day0, step0=k, out_mode=om)
        if reducing:
            smp = r["surf"]                                    # (D, 1, Q*25 | 25, S), sample-minor
            p50 = smp[:, 0] if reduce_channel == 1 else smp[:, 0, 25:50]
            buf[:, C + k] = p50.permute(0, 2, 1).reshape(R, 25)
            nq = len(reduce_qs)
            if reduce_channel is None:
                out_q.append(eng.reduce_async(smp, reduce_qs, Q * 25, None, 1).view(D, 1, nq, Q, 5, 5))
            elif reduce_channel == 1:
                out_q.append(eng.reduce_async(smp, reduce_qs, 25, None, 1).view(D, 1, nq, 5, 5))
            else:
                out_q.append(eng.reduce_async(smp, reduce_qs, 25, reduce_channel, Q).view(D, 1, nq, 5, 5))
            if X > 0:
                exb[:, C + k] = r["ex"][:, 0]
        else:
            out_s[:, k] = r["surf"][:, 0]
            buf[:, C + k] = r["surf"][:, 0, 1].reshape(R, 25)
            if X > 0:
                out_e[:, k] = r["ex"][:, 0]
                exb[:, C + k] = r["ex"][:, 0]
        if ptab is not None and k < horizon - 1:
            for which in (0, 1):
                eng.encode_pre(which, buf[:, C + k], TT * 25, exb[:, C + k] if X > 0 else None, TT * X, R, ptab[which][k])
    if reducing:
        eng.join_side()
        return torch.cat(out_q, dim=1), None
    return out_s, out_e


def ar_generate_multi(eng: Engine, surf_ctx: torch.Tensor, ex_ctx: Optional[torch.Tensor], horizon: int,
                      model_horizon: int, eps: Optional[torch.Tensor] = None, seed: int = 0, day0: int = 0):
    """generate_autoregressive_sequence (:717-807) of a model whose own horizon is model_horizon > 1: the inner
    call at :776 passes no horizon, so every one of the `horizon` steps generates model_horizon days, ALL of them
    are appended to the output (:801) and only the first one's p50 channel / ex-features slide into the context
    (:786-788).  No reference config uses it; a plain loop of shared-prefix generations.
    eps: (B, horizon, model_horizon, L) or None -> Philox.  Returns (B, horizon*model_horizon, Q, 5, 5)[, (.., X)]."""
    B, C = surf_ctx.shape[:2]
    X = eng.X
    surf = surf_ctx.to(eng.device, torch.float32).reshape(B, C, 5, 5)
    ex = ex_ctx.to(eng.device, torch.float32) if X > 0 else None
    outs_s, outs_e = [], []
    for k in range(horizon):
        e = None if eps is None else eps[:, k].contiguous()
        s, ee, _ = prior_generate(eng, surf, ex, model_horizon, 1, eps=e, seed=seed, day0=day0, step0=k * model_horizon)
        outs_s.append(s)
        surf = torch.cat([surf[:, 1:], s[:, :1, 1]], dim=1).contiguous()
        if X > 0:
            outs_e.append(ee)
            ex = torch.cat([ex[:, 1:], ee[:, :1]], dim=1).contiguous()
    return torch.cat(outs_s, dim=1), (torch.cat(outs_e, dim=1) if X > 0 else None)


def ar_offset_generate(eng: Engine, surf_ctx: torch.Tensor, ex_ctx: Optional[torch.Tensor], ar_steps: int,
                       horizon: int, eps: Optional[torch.Tensor] = None, seed: int = 0, day0: int = 0):
    """generate_autoregressive_offset (cvae_with_mem_randomized.py:809-923): ar_steps
    prior generations of `horizon` days each; the next context is the last C days of
    the concatenated p50 medians (:905-914).  eps: (rows, ar_steps*horizon, L) or None."""
    B, C = surf_ctx.shape[:2]
    X = eng.X
    surf = surf_ctx.to(eng.device, torch.float32).reshape(B, C, 5, 5)
    ex = ex_ctx.to(eng.device, torch.float32) if X > 0 else None
    all_s, all_e, med_s, med_e = [], [], [], []
    for k in range(ar_steps):
        e = None if eps is None else eps[:, k * horizon:(k + 1) * horizon].contiguous()
        s, ee, _ = prior_generate(eng, surf, ex, horizon, 1, eps=e, seed=seed, day0=day0, step0=k * horizon)
        all_s.append(s)
        med_s.append(s[:, :, 1])
        if X > 0:
            all_e.append(ee)
            med_e.append(ee)
        if k < ar_steps - 1:
            surf = torch.cat(med_s, dim=1)[:, -C:].contiguous()
            if X > 0:
                ex = torch.cat(med_e, dim=1)[:, -C:].contiguous()
    return torch.cat(all_s, dim=1), (torch.cat(all_e, dim=1) if X > 0 else None)


def ar_mc_windows(eng: Engine, vol: torch.Tensor, ex: Optional[torch.Tensor], first_day: int, n_days: int,
                  ctx_len: int, horizon: int, n_samples: int, seed: int = 0,
                  qs: Sequence[float] = (5, 50, 95), channel: Optional[int] = 1, chunk_days: Optional[int] = None,
                  day_index0: Optional[int] = None, decode_events=None):
    """Autoregressive MC backfill with on-device quantile reduction (BASELINE configs[4]):
    for every day, S independent `horizon`-day rollouts from the window vol[day-C:day],
    reduced to np.percentile-style quantiles -> (n_days, horizon, nq, 5, 5) for one decoder
    channel (default p50) or (n_days, horizon, nq, Q, 5, 5) when channel is None."""
    assert first_day >= ctx_len and first_day + n_days <= vol.shape[0]
    vol = vol.to(device=eng.device, dtype=torch.float32).contiguous()
    if eng.X > 0:
        ex = ex.to(device=eng.device, dtype=torch.float32).contiguous()
    if day_index0 is None:
        day_index0 = first_day
    Q, S = eng.Q, n_samples
    if n_days == 0:
        return _empty_quantiles(eng, horizon, len(qs), channel)
    if not chunk_days:
        # days per chunk: the day count whose rollouts come closest to filling a whole number (<= 4) of rounds of
        # resident row tiles (10 000 rollouts a day: 13 days = 3.97 rounds; 8 days = 2.44 rounds ran 23 % slower)
        cands = range(1, max(1, min(n_days, 4 * ROUND_ROWS // max(1, S))) + 1)
        chunk_days = max(cands, key=lambda d: (d * S / ROUND_ROWS) / -(-d * S // ROUND_ROWS) + 1e-6 * d)
    outs = []
    for d0 in range(0, n_days, chunk_days):
        nd = min(chunk_days, n_days - d0)
        w0 = first_day + d0 - ctx_len
        win = vol[w0:w0 + nd + ctx_len - 1].reshape(-1, 25).unfold(0, ctx_len, 1).permute(0, 2, 1)   # (nd, C, 25)
        wex = ex[w0:w0 + nd + ctx_len - 1].unfold(0, ctx_len, 1).permute(0, 2, 1) if eng.X > 0 else None
        q, _ = ar_generate(eng, win.reshape(nd, ctx_len, 5, 5), wex, horizon, S, seed=seed, day0=day_index0 + d0,
                           reduce_qs=qs, reduce_channel=channel, decode_events=decode_events)
        outs.append(q)
    return torch.cat(outs, dim=0)


def oracle_mc_windows(eng: Engine, vol: torch.Tensor, ex: Optional[torch.Tensor], first_day: int, n_days: int,
                      ctx_len: int, horizon: int, n_samples: int, seed: int = 0,
                      qs: Sequence[float] = (5, 50, 95), channel: Optional[int] = 1, chunk_days: Optional[int] = None,
                      day_index0: Optional[int] = None, decode_events=None):
    """Oracle (posterior) MC generation over a date range (the teacher-forcing generators,
    experiments/backfill/context20/generate_vae_tf_sequences.py:109-123): for every day the
    model sees context AND target, vol[day-C : day+H], and z is SAMPLED from the posterior at
    all C+H positions (cvae_with_mem_randomized.py:198-199, :1064), so every sample decodes the
    whole sequence.  The two encoders run once per day; the decoder runs per (day, sample) with
    Philox noise keyed by (seed; sample, global day, position).  Reduced on device like
    prior_mc_windows.  Returns (n_days, horizon, nq, 5, 5) or (n_days, horizon, nq, Q, 5, 5)."""
    T = ctx_len + horizon
    assert first_day >= ctx_len and first_day + n_days + horizon <= vol.shape[0] + 0
    vol = vol.to(device=eng.device, dtype=torch.float32).contiguous()
    if eng.X > 0:
        ex = ex.to(device=eng.device, dtype=torch.float32).contiguous()
    if day_index0 is None:
        day_index0 = first_day
    L, Q, X = eng.L, eng.Q, eng.X
    if n_days == 0:
        return _empty_quantiles(eng, horizon, len(qs), channel)
    w0 = first_day - ctx_len
    sv = vol[w0:]
    se = ex[w0:] if X > 0 else None
    enc, ctx = eng.encode_both_windows(sv, se, n_days, T, ctx_len)   # posterior on context + target
    zm, zl = enc[..., :L].contiguous(), enc[..., L:].contiguous()
    if not chunk_days:
        # as prior_mc_windows: as many days per decode launch as fit 2 GiB of samples (long launches waste the
        # smallest share of their last round of resident row tiles)
        chunk_days = max(1, (1 << 31) // (n_samples * horizon * (Q if channel is None else 1) * 25 * 4))
    outs = []
    for d0 in range(0, n_days, chunk_days):
        nd = min(chunk_days, n_days - d0)
        _arm_decode_events(eng, decode_events, nd * n_samples)
        r = eng.decode(nd * n_samples, T, ctx_len, ctx=ctx[d0:d0 + nd], ctx_div=n_samples,
                       zb=zm[d0:d0 + nd], lv=zl[d0:d0 + nd], z_div=n_samples,
                       seed=seed, S=n_samples, day0=day_index0 + d0,
                       out_mode=1 if channel is None else eng.OUT_CHANNEL(channel))
        smp = r["surf"]                                                       # (nd, H, Q*25 | 25, S)
        more = d0 + nd < n_days                               # (a next chunk to decode beside this reduction)
        if channel is None:
            q = eng.reduce_async(smp, qs, Q * 25, None, 1, overlap=more).view(nd, horizon, len(qs), Q, 5, 5)
        else:
            q = eng.reduce_async(smp, qs, 25, None, 1, overlap=more).view(nd, horizon, len(qs), 5, 5)
        outs.append(q)
    eng.join_side()
    return torch.cat(outs, dim=0)
