"""The benchmarked workload at its own size (BASELINE configs[3]: context 60, latent 12, 3 ex-features,
4000 days x 1000 samples, one 4M-row fused decode launch, sample-minor stores, on-device p05/p50/p95) and
one day of the configs[4] stress shape (10000 autoregressive rollouts x 30 steps).  What the golden-size
tests cannot reach: 31-bit row arithmetic, the 2 GiB chunk rule, the 3-segment prefix at 125 row tiles,
the LDS percentile kernel behind a real rollout.  Checks: (i) bit-equality under re-chunking and under a
2-way day split (what date-range sharding relies on), (ii) sampled days against the CPU oracle fed the
materialised Philox noise, (iii) finite and ordered quantiles everywhere."""
import numpy as np
import pytest
import torch

from conftest import record_margin
from oracle import vsv_oracle as vo

pytestmark = pytest.mark.gpu
TOL = 1e-4          # north-star: max abs err on vols


def _t(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def _n(t):
    return t.detach().cpu().numpy().astype(np.float64)


def _ordered(q):
    return bool(torch.isfinite(q).all()) and bool(((q[:, :, 0] <= q[:, :, 1]) & (q[:, :, 1] <= q[:, :, 2])).all())


def test_c3_full_size_prior_mc_backfill():
    from vsv import paths
    from vsv.engine import Engine
    from vsv.synth import synthetic_calendar
    L, X, C, H, N, S, seed = 12, 3, 60, 1, 4000, 1000, 0
    cfg = vo.make_config(L, X)
    W = vo.init_weights(cfg, 0, 1.0, np.float32)
    eng = Engine(cfg, W, "cuda:0")
    vol_np, ex_np = synthetic_calendar(N, C, X)
    vol, ex = _t(vol_np), _t(ex_np)
    q = paths.prior_mc_windows(eng, vol, ex, C, N, C, H, S, seed=seed, chunk_days=4000)        # the bench's launch shape
    assert q.shape == (N, H, 3, 5, 5)
    assert _ordered(q)                                                                          # (iii)
    # (i) re-chunked, and split in two date ranges with global day counters
    q500 = paths.prior_mc_windows(eng, vol, ex, C, N, C, H, S, seed=seed, chunk_days=500)
    assert torch.equal(q, q500)
    del q500
    half = N // 2
    qa = paths.prior_mc_windows(eng, vol, ex, C, half, C, H, S, seed=seed, chunk_days=4000)
    qb = paths.prior_mc_windows(eng, vol, ex, C + half, N - half, C, H, S, seed=seed, chunk_days=4000)
    assert torch.equal(q, torch.cat([qa, qb]))
    # (ii) sampled days against the oracle + np.percentile, same noise
    days = [0, 1, 31, 32, 1999, 2000, 3998, 3999]
    eps = np.stack([_n(eng.philox_normal(seed, C + d, 1, S, 0, H)) for d in days])              # (8, S, H, L)
    W64 = {k: v.astype(np.float64) for k, v in W.items()}
    win = np.stack([vol_np[d:d + C] for d in days]).astype(np.float64)
    wex = np.stack([ex_np[d:d + C] for d in days]).astype(np.float64)
    smp, _ = vo.prior_mc_shared_prefix(W64, cfg, win, wex, eps)                                  # (8, S, H, 3, 5, 5)
    want = np.percentile(smp[:, :, :, 1], [5, 50, 95], axis=1).transpose(1, 2, 0, 3, 4)          # (8, H, 3, 5, 5)
    e = float(np.max(np.abs(_n(q[days]) - want)))
    record_margin(e)
    assert e <= TOL, e
    # all three decoder channels at once reduce to the same p50-channel numbers
    # (a 64-day call runs its date-range prefix on 4-row tiles, the 4000-day call on 16-row tiles: the same products
    # accumulated in the same k order, so even across tile shapes the numbers are bit-equal)
    qall = paths.prior_mc_windows(eng, vol, ex, C, 64, C, H, S, seed=seed, channel=None)
    assert torch.equal(qall[:, :, :, 1], q[:64])


def test_c4_one_day_autoregressive_rollouts_vs_oracle():
    """One day of the stress shape: 10000 rollouts x 30 steps, context 20, reduced on device by the
    S > 1024 percentile path.  The oracle's literal rollout of all 10000 rows would take ~20 minutes of
    CPU, and rows are independent, so the check is split: (a) 128 rows spread over the 10000 against the
    oracle on the same Philox noise, all 30 steps; (b) the on-device reduction of the backfill path against
    np.percentile of the raw rollouts of the same seed (all 10000 rows)."""
    from vsv import paths
    from vsv.engine import Engine
    from vsv.synth import synthetic_calendar
    L, X, C, H, S, seed = 5, 0, 20, 30, 10000, 3
    cfg = vo.make_config(L, X)
    W = vo.init_weights(cfg, 0, 1.0, np.float32)
    eng = Engine(cfg, W, "cuda:0")
    vol_np, _ = synthetic_calendar(2, C, X, seed=78)
    q = paths.ar_mc_windows(eng, _t(vol_np), None, C, 1, C, H, S, seed=seed, chunk_days=1)
    assert q.shape == (1, H, 3, 5, 5) and _ordered(q)
    raw, _ = paths.ar_generate(eng, _t(vol_np[None, :C]), None, H, S, seed=seed, day0=C)         # (S, H, 3, 5, 5)
    raw = _n(raw)
    # (b) reduction
    wq = np.percentile(raw[:, :, 1], [5, 50, 95], axis=0).transpose(1, 0, 2, 3)[None]             # (1, H, 3, 5, 5)
    e = float(np.max(np.abs(_n(q) - wq)))
    record_margin(e, "reduce")
    assert e <= 1e-6, e
    # (a) sampled rows against the oracle
    rows = np.unique(np.concatenate([np.arange(0, S, 79), [31, 32, 33, S - 1]]))
    noise = _n(eng.philox_normal(seed, C, 1, S, 0, H))[rows]                                      # (R, H, L)
    zs = np.zeros((H, len(rows), C + 1, L))
    zs[:, :, C] = noise.transpose(1, 0, 2)
    W64 = {k: v.astype(np.float64) for k, v in W.items()}
    win = np.broadcast_to(vol_np[:C][None].astype(np.float64), (len(rows), C, 5, 5)).copy()
    want, _ = vo.generate_autoregressive_sequence(W64, cfg, win, None, zs)                        # (R, H, 3, 5, 5)
    e = float(np.max(np.abs(raw[rows] - want)))
    record_margin(e)
    assert e <= TOL, e
