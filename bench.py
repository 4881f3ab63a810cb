#!/usr/bin/env python3
"""Benchmark of the CVAEMemRand Monte-Carlo prior generation path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input: the
whole backfill of `--days` context windows x `--samples` prior MC samples x
`--horizon` decoded days (BASELINE.json configs[3]: context 60, latent 12,
3 extra features, 4000 days x 1000 samples), ending in the on-device
p05/p50/p95 reduction.  Inputs (the calendar of surfaces, the packed weights)
are resident in HBM before the timed region.  With N > 1 there is one process
per GPU over RCCL: either the caller launched them (torch.distributed.run sets
RANK / WORLD_SIZE, which must equal --gpus) or, called plainly as
`python bench.py --gpus N`, this script spawns them itself through
torch.distributed.run before it touches a GPU.  Every rank owns its own date
range of `--days` days (weak scaling: `value`; global Philox day counters) and
the reduced quantile tensors are all-gathered at the end of every step; a second
timed leg splits `--days` days in total over the ranks (strong scaling, the
BASELINE configs[3] split) and is reported under "strong".

Prints ONE JSON line on rank 0.  See DESIGN.md "Measurement".
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(REPO, "vol-surface-vae-pub_amd")
for _p in (REPO, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense


def flops_per_unit(L: int, X: int, Hm: int = 100, c_last: int = 5, Q: int = 3):
    """Algorithmic FLOPs (2*MAC, no padding, no recomputation) per time step of one
    row, SURVEY.md 8(d): encoder stack, context stack, decoder LSTM, decoder head.
    Q: decoder channels the output Conv produces (3; 1 when only the reduced channel is decoded)."""
    E = c_last * 25 + X
    Hd = c_last * 25 + X
    conv_enc = 2 * 25 * 9 * (1 * 5 + 5 * 5 + 5 * 5)                 # 3x3 convs on 5x5 incl. padding taps = 24750
    lstm_enc = 2 * 4 * Hm * (E + Hm) + 2 * 4 * Hm * (2 * Hm)
    f_enc = conv_enc + lstm_enc + 2 * Hm * 2 * L
    f_ctx = conv_enc + lstm_enc + 2 * Hm * L
    f_dec_lstm = 2 * 4 * Hd * (2 * L + Hd) + 2 * 4 * Hd * (2 * Hd)
    conv_dec = 2 * 25 * 9 * (5 * 5 * 3 + 5 * Q)
    f_dec_head = 2 * 125 * Hd + conv_dec + (2 * X * Hd + 2 * X * X if X else 0)
    # first decode step of an MC sample: its (h, c) is the day's shared prefix state, so the
    # recurrent products W_hh.h_prev are computed once per day, not per sample; only the
    # z-dependent part of layer 0 (the context half of the input is the zero padding) and the
    # layer-0 -> layer-1 product remain per sample
    f_dec_lstm_first = 2 * 4 * Hd * L + 2 * 4 * Hd * Hd
    return {"enc": f_enc, "ctx": f_ctx, "dec_lstm": f_dec_lstm, "dec_head": f_dec_head,
            "dec_lstm_first": f_dec_lstm_first}


from vsv.synth import init_weights, make_config, synthetic_calendar  # noqa: E402


def cpu_baseline(cfg, W, vol, ex, C, H, S, budget_s: float):
    """The CPU port (oracle/torch_port.py: the torch CPU ops the reference calls, all host threads torch was given),
    SURVEY 8(d)'s loop shapes on a bounded sample of the same workload:
      (ii)  the headline figure -- the reference's strongest usage, one get_surface_given_conditions call per day with
            batch = S samples (generate_surfaces.py:41-49), literal evaluation (all C+H decoder steps for every sample),
            float32: whole days until `budget_s` seconds have been spent (>= 2 days after one untimed warm-up day);
      (i)   `literal` -- the reference's own Monte-Carlo loop, one B = 1 call per (day, sample) with a host copy of the
            result (eval_scripts/eval_utils.py:29-38), float32: 4 days x 48 samples, scaled per call (a call's cost
            does not depend on how many follow it);
      fp64  `fp64` -- shape (ii) in float64, the generation scripts' default dtype (train_quantile_models.py:20)."""
    from oracle.torch_port import TorchPort
    L = cfg["latent_dim"]
    tv = torch.from_numpy(vol)
    te = None if ex is None else torch.from_numpy(ex)
    n_win = max(1, vol.shape[0] - C)

    def batched(dtype, budget):
        port = TorchPort(cfg, W, dtype)
        gen = torch.Generator().manual_seed(99)
        v = tv.to(dtype)
        e = None if te is None else te.to(dtype)

        def one_day(d):
            win = v[d:d + C].unsqueeze(0).expand(S, C, 5, 5).contiguous()
            wex = None if e is None else e[d:d + C].unsqueeze(0).expand(S, C, e.shape[1]).contiguous()
            z = torch.randn((S, C + H, L), generator=gen, dtype=dtype)
            port.get_surface_given_conditions(win, wex, z, H)

        one_day(0)
        t0 = time.perf_counter()
        days = 0
        while True:
            one_day(days % n_win)
            days += 1
            el = time.perf_counter() - t0
            if days >= 2 and el >= budget:
                break
        return days, el

    days, el = batched(torch.float32, budget_s)
    out = {"value": days * S * H / el, "unit": "surfaces/s", "cores": int(torch.get_num_threads()), "kind": "port",
           "sample": f"{days} days x {S} samples x horizon {H}, one batched call per day (literal C+H decode), "
                     f"torch CPU float32, {el:.1f} s"}
    try:
        d64, e64 = batched(torch.float64, budget_s / 3)
        out["fp64"] = {"value": d64 * S * H / e64, "unit": "surfaces/s",
                       "sample": f"{d64} days x {S} samples, the same batched shape in float64 (the scripts' default), {e64:.1f} s"}
        port = TorchPort(cfg, W, torch.float32)
        gen = torch.Generator().manual_seed(98)
        n_days_lit, n_smp_lit = 4, 48

        def literal_day(d, n):
            win = tv[d:d + C].unsqueeze(0).contiguous()
            wex = None if te is None else te[d:d + C].unsqueeze(0).contiguous()
            for _ in range(n):
                z = torch.randn((1, C + H, L), generator=gen)
                s_, _e = port.get_surface_given_conditions(win, wex, z, H)
                s_.detach().cpu().numpy()

        literal_day(0, 4)
        t0 = time.perf_counter()
        for d in range(n_days_lit):
            literal_day(d % n_win, n_smp_lit)
        el_lit = time.perf_counter() - t0
        calls = n_days_lit * n_smp_lit
        out["literal"] = {"value": calls * H / el_lit, "unit": "surfaces/s", "ms_per_call": el_lit / calls * 1e3,
                          "sample": f"{n_days_lit} days x {n_smp_lit} samples, one B = 1 call per (day, sample) with a host "
                                    f"copy (eval_utils.py:29-38), float32, {el_lit:.1f} s; per-call cost scales to "
                                    f"{S} samples/day unchanged"}
    except Exception as exc:                       # the extra shapes must not cost the bench line
        out["extra_error"] = repr(exc)[:200]
    return out


def train_bench(dev, steps: int, cpu_steps: int):
    """The metric's second half, 'train samples/s' (BASELINE configs[1]): CVAEMemRand ex_loss
    (surface + 3 extra features, return-only L2 term), batch 64, sequence length drawn per batch
    from [2, 21] (context 1-20 + horizon 1), AdamW(1e-5), the drop-in class's train_step on the
    HIP kernels.  The CPU figure is the torch port's autograd step on the same batches."""
    from vae.cvae_with_mem_randomized import CVAEMemRand
    B = 64
    cfg = make_config(5, 3, device=str(dev))
    W = init_weights(cfg, 0, 1.0, np.float32)
    torch.manual_seed(0)
    model = CVAEMemRand(dict(cfg))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=False)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-5)
    rng = np.random.Generator(np.random.PCG64(4321))
    lens = rng.integers(2, 22, size=steps + 3)
    batches = [{"surface": torch.from_numpy((0.2 + 0.1 * rng.standard_normal((B, int(T), 5, 5))).astype(np.float32)).to(dev),
                "ex_feats": torch.from_numpy((0.01 * rng.standard_normal((B, int(T), 3))).astype(np.float32)).to(dev)}
               for T in lens]
    # default path: forward + loss + backward + scatter replayed as one HIP graph per sequence length
    # (vsv/graphed.py).  Untimed first: one step per length, so every graph is captured.
    warm = {int(b["surface"].shape[1]): b for b in batches}
    for T in range(2, 22):
        if T not in warm:
            warm[T] = {"surface": torch.full((B, T, 5, 5), 0.2, device=dev), "ex_feats": torch.zeros((B, T, 3), device=dev)}
    for b in list(warm.values()) + batches[:3]:
        model.train_step(b, opt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in batches[3:]:
        model.train_step(b, opt)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"metric": "train samples/s", "value": B * steps / dt, "unit": "samples/s", "ms_per_step": dt / steps * 1e3,
           "config": {"workload": "BASELINE configs[1]: CVAEMemRand ex_loss train_step, batch 64, T ~ U[2,21], "
                                  "horizon 1, AdamW(1e-5), clip 1.0, dropout 0", "steps": steps,
                      "hip_graphs": bool(model._use_graphs())}}
    # roofline of the step: algorithmic FLOPs (3 x forward; context stack over T - 1 steps, posterior
    # encoder and decoder over T) over the step time, and the latency model that bounds it -- the step is a chain of
    # eight recurrent launches (encoder | context encoder layer 0, layer 1, decoder layer 0, layer 1, and their BPTT
    # mirrors), each walking T dependent time steps on the two-to-sixteen CUs a batch of 64 occupies: step time
    # fitted as fixed + per_time_step * T from replays at T = 5 and T = 21
    flt = flops_per_unit(5, 3, Q=3)
    Ts = [int(b["surface"].shape[1]) for b in batches[3:]]
    fl_step = float(np.mean([B * 3 * ((T - 1) * flt["ctx"] + T * (flt["enc"] + flt["dec_lstm"]) + flt["dec_head"]) for T in Ts]))
    out["roofline"] = {"bound": "latency (recurrent chain), not mfma", "flops_per_step": fl_step,
                       "achieved": fl_step / (dt / steps) / 1e12, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                       "frac": fl_step / (dt / steps) / 1e12 / PEAK_F32_MFMA_TFLOPS}
    try:
        fit = {}
        for T in (5, 21):
            b = warm[T]
            for _ in range(3):
                model.train_step(b, opt)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                model.train_step(b, opt)
            torch.cuda.synchronize()
            fit[T] = (time.perf_counter() - t0) / 20
        per_t = (fit[21] - fit[5]) / 16
        out["roofline"]["critical_path"] = {
            "ms_at_T5": fit[5] * 1e3, "ms_at_T21": fit[21] * 1e3, "us_per_time_step": per_t * 1e6,
            "fixed_ms": (fit[5] - 5 * per_t) * 1e3, "mean_T": float(np.mean(Ts)),
            "model_ms_at_mean_T": (fit[5] + (float(np.mean(Ts)) - 5) * per_t) * 1e3,
            "note": "a time step of the chain = 8 dependent recurrent steps (4 forward, 4 BPTT) plus the per-T share of the "
                    "T-parallel launches; the fixed part is the launches whose length does not grow with T"}
    except Exception as exc:
        out["roofline"]["critical_path"] = {"error": repr(exc)[:200]}
    # the same steps launched eagerly (config hip_graphs = False)
    try:
        model.config["hip_graphs"] = False
        for b in batches[:3]:
            model.train_step(b, opt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in batches[3:]:
            model.train_step(b, opt)
        torch.cuda.synchronize()
        out["ms_per_step_eager"] = (time.perf_counter() - t0) / steps * 1e3
    except Exception as exc:                       # the comparison leg must not cost the bench line
        out["ms_per_step_eager"] = None
        out["eager_error"] = repr(exc)[:200]
    model.config["hip_graphs"] = True
    # BASELINE configs[2]: multi-horizon [1, 7, 14, 30] quantile training, context 20 (sequence 50), batch 64:
    # four forward/backward passes and one optimizer step per call (train_step_multihorizon)
    try:
        mh_batch = {"surface": torch.from_numpy((0.2 + 0.1 * rng.standard_normal((B, 50, 5, 5))).astype(np.float32)).to(dev),
                    "ex_feats": torch.from_numpy((0.01 * rng.standard_normal((B, 50, 3))).astype(np.float32)).to(dev)}
        for _ in range(2):
            model.train_step_multihorizon(mh_batch, opt, horizons=[1, 7, 14, 30])
        torch.cuda.synchronize()
        n_mh = 10
        t0 = time.perf_counter()
        for _ in range(n_mh):
            model.train_step_multihorizon(mh_batch, opt, horizons=[1, 7, 14, 30])
        torch.cuda.synchronize()
        mh_dt = (time.perf_counter() - t0) / n_mh
        out["multihorizon"] = {"workload": "BASELINE configs[2]: train_step_multihorizon, horizons [1,7,14,30], batch 64, "
                                           "sequence 50 (context 20 + 30)", "ms_per_step": mh_dt * 1e3,
                               "samples_per_s": B / mh_dt}
    except Exception as exc:
        out["multihorizon"] = {"error": repr(exc)[:200]}
    if cpu_steps > 0:
        from oracle.torch_port import TorchPort
        port = TorchPort(cfg, W, torch.float32)
        params = list(port.parameters())
        copt = torch.optim.AdamW(params, lr=1e-5)
        gen = torch.Generator().manual_seed(5)

        def cpu_step(b):
            copt.zero_grad()
            s, e = b["surface"].cpu(), b["ex_feats"].cpu()
            eps = torch.randn((B, s.shape[1], 5), generator=gen)
            port.losses(s, e, eps, 1)["loss"].backward()
            torch.nn.utils.clip_grad_norm_(params, 1.0)
            copt.step()

        cpu_step(batches[0])
        t0 = time.perf_counter()
        for b in batches[3:3 + cpu_steps]:
            cpu_step(b)
        ct = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": B * cpu_steps / ct, "unit": "samples/s", "cores": int(torch.get_num_threads()),
                               "kind": "port", "sample": f"{cpu_steps} train steps of the same batches, torch CPU "
                                                         f"float32 autograd, {ct:.1f} s"}
    return out


def _leg_roofline(kernel: str, flops: float, wall_s: float, events):
    """Per-leg roofline: algorithmic FLOPs over the leg's wall time (whole job) and, where the decode
    launches were bracketed by HIP events on their stream, over the summed launch time of that kernel."""
    out = {"bound": "mfma", "kernel": kernel, "flops": flops, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
           "whole_job_achieved": flops / wall_s / 1e12, "whole_job_frac": flops / wall_s / 1e12 / PEAK_F32_MFMA_TFLOPS}
    if events:
        ms = float(sum(a.elapsed_ms(b) for a, b, _ in events))
        out.update({"decode_launch_ms_total": ms, "decode_launches": len(events),
                    "avg_launch_ms": ms / len(events)})
    return out


def other_workloads(eng, dev, L, X, C, S, cpu_budget: float):
    """The other BASELINE configs on the same engine, each with its own roofline (one untimed + one timed pass):
    c3 prior at horizon 30 and c3 oracle (posterior) MC at the config's 4000 days, c4-shaped autoregressive 30-step
    rollouts (64 days x 10000), c0 (no_ex, context 20, latent 5, 256 days x 1000: the reference's CPU-runnable case)
    on the GPU and on the CPU port."""
    from vsv import paths
    from vsv.engine import Engine
    res = []

    def timed(fn):
        fn(None)
        torch.cuda.synchronize()
        ev = []
        t0 = time.perf_counter()
        q = fn(ev)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, ev, q

    def cal(n, extra=0, ctx=C, x=X, seed=77):
        v, e = synthetic_calendar(n + extra, ctx, x, seed=seed)
        return torch.from_numpy(v).to(dev), (torch.from_numpy(e).to(dev) if x else None)

    fl = flops_per_unit(L, X, Q=1)
    per_day_prefix = C * (fl["enc"] + fl["ctx"] + fl["dec_lstm"])

    n, H = 4000, 30
    v, e = cal(n)
    dt, ev, q = timed(lambda evs: paths.prior_mc_windows(eng, v, e, C, n, C, H, S, seed=1, decode_events=evs))
    check_quantiles(q, "c3 prior H=30")
    flops = n * (per_day_prefix + S * (fl["dec_lstm_first"] + (H - 1) * fl["dec_lstm"] + H * fl["dec_head"]))
    res.append({"workload": f"c3 prior MC, context {C}, latent {L}, {n} days x {S} samples x horizon {H}",
                "value": n * S * H / dt, "unit": "surfaces/s", "ms": dt * 1e3,
                "roofline": _leg_roofline("dec_seq_kernel<16> (T-step fused MC decode, one wave per row tile)", flops, dt, ev)})
    del v, e, q
    n, H = 4000, 1
    v, e = cal(n, H)
    dt, ev, q = timed(lambda evs: paths.oracle_mc_windows(eng, v, e, C, n, C, H, S, seed=1, decode_events=evs))
    check_quantiles(q, "c3 oracle")
    flops = n * ((C + H) * fl["enc"] + C * fl["ctx"] + S * ((C + H) * fl["dec_lstm"] + H * fl["dec_head"]))
    res.append({"workload": f"c3 oracle (posterior) MC, context {C}, latent {L}, {n} days x {S} samples x horizon {H} "
                            f"(every sample decodes all {C + H} positions)",
                "value": n * S * H / dt, "unit": "surfaces/s", "ms": dt * 1e3,
                "roofline": _leg_roofline("dec_seq_kernel<16> (decoder over all C+H positions per sample)", flops, dt, ev)})
    del v, e, q
    # c4 at 64 days x 10000 rollouts x 30 steps (19.2 M surfaces): one untimed chunk, then the timed 64 days
    n, H, Sa, Ca = 64, 30, 10000, 20
    va, ea = cal(n, 0, Ca, X, 78)
    paths.ar_mc_windows(eng, va, ea, Ca, 13, Ca, H, Sa, seed=1)
    torch.cuda.synchronize()
    ev = []
    t0 = time.perf_counter()
    q = paths.ar_mc_windows(eng, va, ea, Ca, n, Ca, H, Sa, seed=1, decode_events=ev)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    check_quantiles(q, "c4 autoregressive MC")
    # FLOPs actually required after the build's de-duplication (what the roofline fraction is priced on): at rollout
    # step k the first C - k days of a window are still the day's own context (run once per DAY), the other
    # min(k, C) days are the rollout's own (per row; their conv embedding and first-layer input product computed
    # once, when the day was generated), and every row decodes its one new day
    Hm, E = 100, 125 + X
    enc_rec = 2 * 4 * Hm * Hm + 2 * 4 * Hm * 2 * Hm                  # a memory step whose input product comes from the table
    enc_tab = 2 * 25 * 9 * 55 + 2 * 4 * Hm * E                       # conv embedding + first-layer input product of one day
    flops = 0.0
    for k in range(H):
        ns, nr = max(Ca - k, 0), min(k, Ca)
        flops += n * ns * (fl["enc"] + fl["ctx"] + fl["dec_lstm"])
        flops += n * Sa * (nr * (2 * enc_rec + 2 * Hm * 3 * L + fl["dec_lstm"]) + fl["dec_lstm"] + fl["dec_head"])
        if k < H - 1:
            flops += n * Sa * 2 * enc_tab
    # SURVEY 8(d)'s counted figure, for reference: per (rollout, step) a full re-encode of the C-day window by both
    # encoders, C decoder memory steps and one decoded day -- no sharing (the build skips part of that work, so this
    # fraction can exceed 1)
    flops_survey = n * Sa * H * (Ca * (fl["enc"] + fl["ctx"] + fl["dec_lstm"]) + fl["dec_lstm"] + fl["dec_head"])
    res.append({"workload": f"c4-shaped autoregressive MC, context {Ca}, latent {L}, ex_feats {X}, {n} days x {Sa} "
                            f"rollouts x {H} steps in chunks of 13 days (each step re-encodes its {Ca}-day window; "
                            f"the days a day's rollouts still share once per day; every step's p05/p50/p95 over the "
                            f"{Sa} rollouts reduced on the side stream, samples never stored)",
                "value": n * Sa * H / dt, "unit": "surfaces/s", "ms": dt * 1e3,
                "roofline": dict(_leg_roofline("enc_seq_kernel<13> x2 + dec_seq_kernel<16> per rollout step", flops, dt, ev),
                                 survey_count_flops=flops_survey,
                                 survey_count_frac=flops_survey / dt / 1e12 / PEAK_F32_MFMA_TFLOPS)})
    del va, ea, q
    # c0: the reference's own CPU-runnable configuration
    L0, X0, C0, n0 = 5, 0, 20, 256
    cfg0 = make_config(L0, X0)
    W0 = init_weights(cfg0, 0, 1.0, np.float32)
    eng0 = Engine(cfg0, W0, dev)
    v0np, _ = synthetic_calendar(n0, C0, X0, seed=79)
    v0 = torch.from_numpy(v0np).to(dev)
    dt, ev, q = timed(lambda evs: paths.prior_mc_windows(eng0, v0, None, C0, n0, C0, 1, S, seed=1, decode_events=evs))
    check_quantiles(q, "c0 prior MC")
    fl0 = flops_per_unit(L0, X0, Q=1)
    flops = n0 * (C0 * (fl0["enc"] + fl0["ctx"] + fl0["dec_lstm"]) + S * (fl0["dec_lstm_first"] + fl0["dec_head"]))
    leg = {"workload": f"c0 prior MC (no ex-features), context {C0}, latent {L0}, {n0} days x {S} samples x horizon 1",
           "value": n0 * S / dt, "unit": "surfaces/s", "ms": dt * 1e3,
           "roofline": _leg_roofline("mc_decode1_kernel<16> + the 20-step date-range prefix of 256 days (latency-bound: "
                                     "8 row tiles per stack)", flops, dt, ev)}
    if cpu_budget > 0:
        leg["cpu_baseline"] = cpu_baseline(cfg0, W0, v0np, None, C0, 1, S, min(cpu_budget, 6.0))
    res.append(leg)
    return res


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--days", type=int, default=4000, help="days per rank (weak leg) / in total (strong leg)")
    ap.add_argument("--samples", type=int, default=1000)
    ap.add_argument("--context", type=int, default=60)
    ap.add_argument("--horizon", type=int, default=1)
    ap.add_argument("--latent", type=int, default=12)
    ap.add_argument("--ex-feats", type=int, default=3)
    ap.add_argument("--chunk-days", type=int, default=4000, help="days per decode launch")
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of CPU baseline work (0 = skip)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--allow-oversubscribe", action="store_true",
                    help="let several ranks share one GPU (rehearsals on a 1-GPU box; reported in the output)")
    ap.add_argument("--train-steps", type=int, default=50, help="training steps of the secondary metric (0 = skip)")
    ap.add_argument("--others", type=int, default=1, help="1 = also time the secondary workloads (N=1 only)")
    ap.add_argument("--master-port", type=int, default=29611, help="rendezvous port when this script spawns the ranks")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher check only: the ranks rendezvous, count themselves with one all_reduce and rank 0 "
                         "prints the count; no GPU work (runs on a CPU-only host with --backend gloo)")
    return ap.parse_args(argv)


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` with no launcher around it: start N ranks through torch.distributed.run
    as a CHILD process (this parent never initialises a GPU) and return its exit code."""
    import subprocess
    n_dev = torch.cuda.device_count()                     # counting devices does not initialise the GPU
    if n_dev < args.gpus and not args.allow_oversubscribe:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) are visible "
                         "(one rank per GPU; --allow-oversubscribe lets ranks share a GPU for rehearsals)")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(args.master_port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")).returncode


def check_quantiles(q: "torch.Tensor", what: str):
    """Size-independent sanity of a reduced (days, H, 3, 5, 5) product: finite and p05 <= p50 <= p95."""
    if not bool(torch.isfinite(q).all()):
        raise RuntimeError(f"{what}: non-finite quantiles")
    if not bool(((q[:, :, 0] <= q[:, :, 1]) & (q[:, :, 1] <= q[:, :, 2])).all()):
        raise RuntimeError(f"{what}: quantiles out of order")


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.gpus > 1 and not launched:
        raise SystemExit(spawn_ranks(args, argv))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.dry_run:
        import torch.distributed as dist
        if world > 1:
            dist.init_process_group("gloo")           # host-side count only
            t = torch.ones(1)
            dist.all_reduce(t)
            seen = int(t.item())
            dist.destroy_process_group()
        else:
            seen = 1
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "ranks": seen}), flush=True)
        return
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a GPU: the HIP path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    if local_rank >= n_dev:
        if not args.allow_oversubscribe:
            raise SystemExit(f"bench.py: rank {rank} has no GPU of its own ({n_dev} visible); "
                             "pass --allow-oversubscribe to share GPUs (rehearsal only)")
        local_rank %= n_dev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from vsv import paths, sharded  # noqa: F401
    from vsv.engine import Engine

    L, X, C, H, S, N = args.latent, args.ex_feats, args.context, args.horizon, args.samples, args.days
    cfg = make_config(L, X)
    W = init_weights(cfg, 0, 1.0, np.float32)
    eng = Engine(cfg, W, dev)
    vol_np, ex_np = synthetic_calendar(N * world, C, X)
    # every rank holds the whole (small) calendar; vsv.sharded gives it its own N-day range
    vol = torch.from_numpy(vol_np).to(dev)
    ex = torch.from_numpy(ex_np).to(dev) if X else None

    def one_step(n_total, events=None):
        return sharded.prior_mc_sharded(eng, vol, ex, C, n_total, C, H, S, seed=0, qs=(5, 50, 95), channel=1,
                                        chunk_days=args.chunk_days, decode_events=events)

    def fence():
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n_total, events=None):
        """W untimed + K timed steps over n_total days (split over the ranks); max over ranks of the wall time."""
        for _ in range(args.warmup):
            one_step(n_total)
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            q = one_step(n_total, events)
        fence()
        dt = time.perf_counter() - t0
        if dist:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        check_quantiles(q, f"prior MC over {n_total} days")       # the last step's product, outside the timed region
        assert q.shape == (n_total, H, 3, 5, 5), q.shape
        return dt

    events = []
    dt = timed(N * world, events)                                  # weak: N days per GPU
    dt_strong = timed(N) if world > 1 else dt                      # strong: N days in total

    fl = flops_per_unit(L, X, Q=1)                                 # one_step() decodes the p50 channel only
    k_ms = [a.elapsed_ms(b) for a, b, _ in events]
    k_rows = [r for _, _, r in events]
    avg_ms = float(np.mean(k_ms))
    # the timed launch is the fused MC decode kernel: per sample one de-duplicated first step,
    # H-1 full LSTM steps and H decoder heads (DESIGN.md "Measurement")
    per_row = fl["dec_lstm_first"] + (H - 1) * fl["dec_lstm"] + H * fl["dec_head"]
    flops_per_launch = float(np.mean(k_rows)) * per_row
    achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12
    traffic = None
    tpath = os.path.join(REPO, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("workload") == f"c3-L{L}-X{X}-C{C}-H{H}-S{S}-chunk{args.chunk_days}":
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    surfaces = world * N * S * H * args.steps
    gpus_used = len({r % n_dev for r in range(world)}) if world > n_dev else world
    out = {
        "metric": "MC surfaces/sec (5x5 grid, 1000 samples/day)",
        "value": surfaces / dt,
        "unit": "surfaces/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (0.2+0.1*randn surfaces, 0.01*randn ex-features, seeded random weights, Philox noise)",
        "config": {"workload": f"BASELINE configs[3]: CVAEMemRand context {C}, latent {L}, ex_feats {X}, prior MC "
                               f"generation, {N} days/GPU x {S} samples x horizon {H}, p05/p50/p95 reduced on device",
                   "days_per_gpu": N, "samples_per_day": S, "context_len": C, "horizon": H,
                   "chunk_days": args.chunk_days, "parallelism": f"date-range shards x{world}",
                   "backend": (args.backend if world > 1 else None),
                   "ranks": (dist.get_world_size() if dist else 1), "gpus_used": gpus_used,
                   "oversubscribed": world > n_dev},
        "strong": {"workload": f"the same backfill with {N} days IN TOTAL split over the {world} rank(s)",
                   "value": N * S * H * args.steps / dt_strong, "unit": "surfaces/s",
                   "ms_per_step": dt_strong / args.steps * 1e3, "scaling": "strong"},
        "roofline": {"bound": "mfma", "kernel": ("mc_decode1_kernel<16>" if H == 1 else "lstm_kernel<16,2,1,true>") +
                               " (fused MC decode: Philox latents, LSTM step(s), decoder head, p50 channel only)",
                     "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                     "flops_per_launch": flops_per_launch, "avg_launch_ms": avg_ms, "launches": len(k_ms)},
    }
    if rank == 0 and world == 1 and args.cpu_budget > 0:
        out["cpu_baseline"] = cpu_baseline(cfg, W, vol_np, ex_np, C, H, S, args.cpu_budget)
    if rank == 0 and world == 1 and args.train_steps > 0:
        out["train"] = train_bench(dev, args.train_steps, 10 if args.cpu_budget > 0 else 0)
    if world == 1 and args.others:
        out["other_workloads"] = other_workloads(eng, dev, L, X, C, S, args.cpu_budget)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
