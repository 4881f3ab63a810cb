"""dec_seq_kernel (round 3): the decoder over T steps on the one-wave-per-row-tile schedule, chosen by vsv_decode for
launches of >= 1024 row tiles (horizon > 1 prior draws, oracle decodes, autoregressive rollout steps).  It runs the
same products in the same k order as lstm_kernel, which the golden tests pin to the reference, so the checks here are
(i) BIT-equality with the same work issued in pieces small enough to stay on lstm_kernel (Philox counters are keyed by
global day and sample, so a day range splits exactly), for every input mode the decoder has, and (ii) sampled days
against the numpy oracle fed the materialised Philox noise (<= 1e-4 on vols)."""
import numpy as np
import pytest
import torch

from conftest import record_margin
from oracle import vsv_oracle as vo

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _t(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def _n(t):
    return t.detach().cpu().numpy().astype(np.float64)


def _engine(L, X):
    from vsv.engine import Engine
    cfg = vo.make_config(L, X)
    W = vo.init_weights(cfg, 0, 1.5, np.float32)
    return cfg, W, Engine(cfg, W, "cuda:0")


def _windows(D, C, X, seed):
    rng = np.random.default_rng(seed)
    surf = (0.2 + 0.1 * rng.standard_normal((D, C, 5, 5))).astype(np.float32)
    ex = (0.01 * rng.standard_normal((D, C, X))).astype(np.float32) if X else None
    return surf, ex


@pytest.mark.parametrize("L,X", [(12, 3), (5, 0)])
def test_prior_horizon5_equals_lstm_kernel_and_oracle(L, X):
    """Prior draws at horizon 5 from a shared per-day state: 33 days x 1000 samples = 33000 rows (1032 row tiles, the
    last one ragged) in one launch against the same days in launches of 11 days (344 tiles: lstm_kernel)."""
    from vsv import paths
    cfg, W, eng = _engine(L, X)
    D, S, C, H, seed, day0 = 33, 1000, 6, 5, 11, 100
    surf, ex = _windows(D, C, X, 5)
    ts, te = _t(surf), _t(ex)
    s, e, _ = paths.prior_generate(eng, ts, te, H, S, seed=seed, day0=day0)
    assert s.shape == (D * S, H, 3, 5, 5)
    for d0 in range(0, D, 11):
        s1, e1, _ = paths.prior_generate(eng, ts[d0:d0 + 11], None if te is None else te[d0:d0 + 11], H, S, seed=seed,
                                         day0=day0 + d0)
        assert torch.equal(s[d0 * S:(d0 + 11) * S], s1)
        if X:
            assert torch.equal(e[d0 * S:(d0 + 11) * S], e1)
    # sample-minor stores, all channels and one channel
    sm, _, _ = paths.prior_generate(eng, ts, te, H, S, seed=seed, day0=day0, out_mode=1)           # (D, H, Q*25, S)
    assert torch.equal(sm, s.view(D, S, H, 75).permute(0, 2, 3, 1))
    s1c, _, _ = paths.prior_generate(eng, ts, te, H, S, seed=seed, day0=day0, out_mode=eng.OUT_CHANNEL(1))
    assert torch.equal(s1c, sm[:, :, 25:50])
    # two days against the oracle on the same noise
    days = [0, D - 1]
    eps = np.stack([_n(eng.philox_normal(seed, day0 + d, 1, S, 0, H)) for d in days])              # (2, S, H, L)
    W64 = {k: v.astype(np.float64) for k, v in W.items()}
    want_s, want_e = vo.prior_mc_shared_prefix(W64, cfg, surf[days].astype(np.float64),
                                               None if ex is None else ex[days].astype(np.float64), eps)
    got = _n(s.view(D, S, H, 3, 5, 5)[days])
    err = float(np.max(np.abs(got - want_s)))
    record_margin(err)
    assert err <= TOL, err
    if X:
        err = float(np.max(np.abs(_n(e.view(D, S, H, X)[days]) - want_e)))
        record_margin(err, "ex")
        assert err <= TOL, err


def test_oracle_decode_equals_lstm_kernel():
    """Posterior decode of all C + H positions (z = mean + exp(logvar / 2) * Philox at every step, context embedding
    for the first C): 33 days x 1000 samples in one launch against launches of 16 days."""
    from vsv import paths
    from vsv.synth import synthetic_calendar
    L, X, C, H, N, S = 12, 3, 6, 3, 33, 1000
    cfg, W, eng = _engine(L, X)
    vol_np, ex_np = synthetic_calendar(N + H, C, X, seed=9)
    vol, ex = _t(vol_np), _t(ex_np)
    s, e = paths.oracle_windows(eng, vol, ex, C, N, C, H, S, seed=4, chunk_rows=1 << 16)           # one 33000-row launch
    s1, e1 = paths.oracle_windows(eng, vol, ex, C, N, C, H, S, seed=4, chunk_rows=1 << 14)         # 16 days per launch
    assert torch.equal(s, s1) and torch.equal(e, e1)
    assert bool(torch.isfinite(s).all())
    # one day against the oracle: forward() with the materialised noise
    d = 17
    eps = _n(eng.philox_normal(4, C + d, 1, S, 0, C + H))                                           # (S, T, L)
    W64 = {k: v.astype(np.float64) for k, v in W.items()}
    rows = np.arange(0, S, 37)
    x = np.broadcast_to(vol_np[d:d + C + H][None].astype(np.float64), (len(rows), C + H, 5, 5))
    xe = np.broadcast_to(ex_np[d:d + C + H][None].astype(np.float64), (len(rows), C + H, X))
    want = vo.forward(W64, cfg, x, xe, eps[rows], H)[0]
    err = float(np.max(np.abs(_n(s[d, rows]) - want)))
    record_margin(err)
    assert err <= TOL, err


def test_rollout_step_decode_modes_equal_lstm_kernel():
    """The decode call of an autoregressive rollout step: per-row context embedding and posterior means for the first
    steps, a prior draw with INJECTED noise at the last, continuing from a per-day state without shared gates."""
    L, X = 12, 3
    cfg, W, eng = _engine(L, X)
    D, S, k = 33, 1000, 3
    R = D * S
    g = torch.Generator(device="cuda").manual_seed(3)
    ctx = 0.3 * torch.randn((R, k, eng.Lc), device="cuda", generator=g)
    zb = 0.5 * torch.randn((R, k, L), device="cuda", generator=g)
    eps = torch.randn((R, 1, L), device="cuda", generator=g)
    state = 0.2 * torch.randn((D, eng.state_floats), device="cuda", generator=g)
    r = eng.decode(R, k + 1, k, ctx=ctx, zb=zb, state_in=state, state_div=S, eps=eps, eps_t0=k, S=S, day0=7, step0=k)
    rp = eng.decode(R, k + 1, k, ctx=ctx, zb=zb, state_in=state, state_div=S, eps_t0=k, seed=21, S=S, day0=7, step0=k,
                    out_mode=eng.OUT_CHANNEL(1))                                                    # Philox, sample-minor
    for d0 in range(0, D, 11):
        a, b = d0 * S, (d0 + 11) * S
        r1 = eng.decode(b - a, k + 1, k, ctx=ctx[a:b], zb=zb[a:b], state_in=state[d0:d0 + 11], state_div=S,
                        eps=eps[a:b], eps_t0=k, S=S, day0=7 + d0, step0=k)
        assert torch.equal(r["surf"][a:b], r1["surf"]) and torch.equal(r["ex"][a:b], r1["ex"])
        rp1 = eng.decode(b - a, k + 1, k, ctx=ctx[a:b], zb=zb[a:b], state_in=state[d0:d0 + 11], state_div=S,
                         eps_t0=k, seed=21, S=S, day0=7 + d0, step0=k, out_mode=eng.OUT_CHANNEL(1))
        assert torch.equal(rp["surf"][d0:d0 + 11], rp1["surf"]) and torch.equal(rp["ex"][a:b], rp1["ex"])
    # no state at all (zero initial state), outputs from step 1 on
    r0 = eng.decode(R, k + 1, 1, ctx=ctx, zb=zb, eps=eps, eps_t0=k, S=S)
    r01 = eng.decode(11 * S, k + 1, 1, ctx=ctx[:11 * S], zb=zb[:11 * S], eps=eps[:11 * S], eps_t0=k, S=S)
    assert torch.equal(r0["surf"][:11 * S], r01["surf"]) and torch.equal(r0["ex"][:11 * S], r01["ex"])


def test_ar_rollouts_equal_per_day_rollouts():
    """generate_autoregressive_sequence for 4 days x 10000 rollouts (1250 row tiles) against the same days one at a
    time (313 tiles each: lstm_kernel), configs[4]'s model (context 20 shortened to 5, latent 5, no ex-features)."""
    from vsv import paths
    L, X, C, H, D, S, seed = 5, 0, 5, 7, 4, 10000, 3
    cfg, W, eng = _engine(L, X)
    surf, _ = _windows(D, C, X, 8)
    ts = _t(surf)
    s, _ = paths.ar_generate(eng, ts, None, H, S, seed=seed, day0=50)
    for d in range(D):
        s1, _ = paths.ar_generate(eng, ts[d:d + 1], None, H, S, seed=seed, day0=50 + d)
        assert torch.equal(s[d * S:(d + 1) * S], s1)
    assert bool(torch.isfinite(s).all())
