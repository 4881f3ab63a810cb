"""GPU tests of the generation drivers (vsv/drivers.py: the reference's generate_surfaces.py,
eval_scripts/eval_utils.py and the teacher-forcing / autoregressive sequence generators as batched
device passes) and of the raw window paths under them, against the CPU oracle fed the same Philox
noise, plus the .npz products' key sets and shapes."""
import numpy as np
import pytest
import torch

from oracle import vsv_oracle as vo

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _t(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def _n(t):
    return t.detach().cpu().numpy().astype(np.float64)


def _err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    from conftest import record_margin
    e = float(np.max(np.abs(a - b)))
    record_margin(e)
    return e


def _calendar(n, X, seed=5):
    rng = np.random.default_rng(seed)
    vol = (0.2 + 0.1 * rng.standard_normal((n, 5, 5))).astype(np.float32)
    ex = (0.01 * rng.standard_normal((n, X))).astype(np.float32) if X else None
    return vol, ex


def _model(L, X, seed=3, Q=3):
    from vae.cvae_with_mem_randomized import CVAEMemRand
    cfg = vo.make_config(L, X, device="cuda:0")
    if Q == 1:
        cfg["quantiles"] = [0.5]
    W = vo.init_weights(cfg, seed, 2.0, np.float32)
    model = CVAEMemRand(dict(cfg))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=False)
    model.eval()
    return model, cfg, W


def test_prior_and_oracle_windows_vs_cpu_oracle():
    from vsv import paths
    from vsv.engine import Engine
    L, X, C, H, S, N = 5, 3, 6, 3, 5, 7
    cfg = vo.make_config(L, X)
    W = vo.init_weights(cfg, 17, 2.0, np.float32)
    eng = Engine(cfg, W, "cuda:0")
    W64 = {k: v.astype(np.float64) for k, v in W.items()}
    vol, ex = _calendar(N + C + H, X)
    first = C + 2                                                   # not the first possible day: global day keys
    s, e = paths.prior_windows(eng, _t(vol), _t(ex), first, N - 2, C, H, S, seed=8, chunk_rows=3 * S * H)
    assert s.shape == (N - 2, S, H, 3, 5, 5) and e.shape == (N - 2, S, H, X)
    eps = _n(eng.philox_normal(8, first, N - 2, S, 0, H)).reshape(N - 2, S, H, L)
    win = np.stack([vol[d - C:d] for d in range(first, first + N - 2)]).astype(np.float64)
    wex = np.stack([ex[d - C:d] for d in range(first, first + N - 2)]).astype(np.float64)
    ws, we = vo.prior_mc_shared_prefix(W64, cfg, win, wex, eps)
    assert _err(_n(s), ws) <= TOL and _err(_n(e), we) <= TOL
    # any split of the day range gives the same numbers (what date-range sharding relies on)
    s2, _ = paths.prior_windows(eng, _t(vol), _t(ex), first + 2, N - 4, C, H, S, seed=8)
    assert torch.equal(s2, s[2:])

    so, eo = paths.oracle_windows(eng, _t(vol), _t(ex), first, N - 2, C, H, S, seed=4, chunk_rows=2 * S)
    assert so.shape == (N - 2, S, H, 3, 5, 5)
    for i, d in enumerate(range(first, first + N - 2)):
        z = _n(eng.philox_normal(4, d, 1, S, 0, C + H))             # (S, T, L)
        ws_, we_ = vo.forward(W64, cfg, np.broadcast_to(vol[d - C:d + H][None].astype(np.float64), (S, C + H, 5, 5)),
                              np.broadcast_to(ex[d - C:d + H][None].astype(np.float64), (S, C + H, X)), z, H)[:2]
        assert _err(_n(so[i]), ws_) <= TOL and _err(_n(eo[i]), we_) <= TOL


def test_ar_offset_windows_vs_cpu_oracle():
    from vsv import paths
    from vsv.engine import Engine
    L, X, C, H, K, N = 5, 3, 6, 7, 3, 4
    cfg = vo.make_config(L, X)
    cfg["context_len"] = C
    W = vo.init_weights(cfg, 19, 2.0, np.float32)
    eng = Engine(cfg, W, "cuda:0")
    W64 = {k: v.astype(np.float64) for k, v in W.items()}
    vol, ex = _calendar(N + C, X, seed=6)
    s, e = paths.ar_offset_windows(eng, _t(vol), _t(ex), C, N, C, K, H, seed=2, chunk_days=3)
    assert s.shape == (N, K * H, 3, 5, 5) and e.shape == (N, K * H, X)
    for i in range(N):
        noise = _n(eng.philox_normal(2, C + i, 1, 1, 0, K * H))[0]                  # (K*H, L)
        zs = np.zeros((K, 1, C + H, L))
        for k in range(K):
            zs[k, 0, C:] = noise[k * H:(k + 1) * H]
        ws, we = vo.generate_autoregressive_offset(W64, cfg, vol[i:i + C][None].astype(np.float64),
                                                   ex[i:i + C][None].astype(np.float64), list(zs), H)
        assert _err(_n(s[i]), ws[0]) <= TOL and _err(_n(e[i]), we[0]) <= TOL


def test_generate_surfaces_drivers_shapes_and_consistency():
    import generate_surfaces as gs
    from eval_scripts import eval_utils as eu
    from vae.utils import set_seeds
    from vsv import paths
    model, cfg, W = _model(5, 3)
    vol, ex = _calendar(40, 3)
    C, N = 5, 12
    set_seeds(0)
    (p05, p50, p95), exf = gs.generate_surfaces_multiday(model, ex.astype(np.float64), vol.astype(np.float64), C, N, 1000,
                                                         check_ex_feats=True, ctx_len=C)
    assert p05.shape == p50.shape == p95.shape == (N, 5, 5) and p50.dtype == np.float64 and exf.shape == (N, 3)
    # the same seed through the raw path gives the same day-by-day draws
    set_seeds(0)
    seed = model._draw_seed()
    s, e = paths.prior_windows(model._engine(), _t(vol), _t(ex), C, N, C, 1, 1, seed=seed)
    assert np.array_equal(p50, _n(s[:, 0, 0, 1])) and np.array_equal(exf, _n(e[:, 0, 0]))
    (q05, q50, q95), ex1 = gs.generate_surfaces_random(model, ex, vol, C + 3, C, 1000, True, True)
    assert q50.shape == (5, 5) and ex1.shape == (3,)
    _, none_ex = gs.generate_surfaces_random(model, ex, vol, C + 3, C, 1000, True, False)
    assert none_ex is None
    # Monte-Carlo product of eval_utils: (days, num_vaes, ...) raw samples
    s3, e3 = eu.generate_surfaces_multiday(model, ex, vol, C, 4, 50, check_ex_feats=True, ctx_len=C)
    assert s3.shape == (4, 50, 3, 5, 5) and e3.shape == (4, 50, 3)
    m1, _, _ = _model(5, 0, Q=1)
    s1, e1 = eu.generate_surfaces_multiday(m1, None, vol, C, 4, 50, ctx_len=C)
    assert s1.shape == (4, 50, 5, 5)                                   # the legacy 1-channel layout
    v1, _ = eu.generate_surfaces(m1, None, vol, C + 1, C, 20, False, False)
    assert v1.shape == (20, 5, 5) and np.std(v1, axis=0).max() > 0     # samples differ


def test_sequence_npz_products(tmp_path):
    from vsv import drivers
    model, cfg, W = _model(5, 3)
    vol, ex = _calendar(60, 3)
    C, H = 6, 3
    for mode in ("oracle", "prior"):
        f = drivers.generate_and_save_horizon(model, vol, ex, 2, 40, H, "crisis", mode, context_len=C, output_dir=str(tmp_path / mode))
        d = np.load(f)
        n = (40 - H + 1) - (2 + C) + 1
        assert set(d.files) == {"surfaces", "indices", "quantiles", "horizon", "period_start", "period_end", "method", "sampling_mode"}
        assert d["surfaces"].shape == (n, H, 3, 5, 5) and d["surfaces"].dtype == np.float64
        assert np.array_equal(d["indices"], np.arange(2 + C, 40 - H + 2)) and int(d["horizon"]) == H
        assert str(d["method"]) == "teacher_forcing" and str(d["sampling_mode"]) == mode
        assert np.isfinite(d["surfaces"]).all()
    with pytest.raises(ValueError):
        drivers.generate_and_save_horizon(model, vol, ex, 2, 40, H, "crisis", "posterior", context_len=C, output_dir=str(tmp_path))
    with pytest.raises(ValueError):
        drivers.generate_and_save_ar_sequences(model, vol, ex, 0, 50, 100, "oos", "prior", output_dir=str(tmp_path))
    # the 180-day product on a short context (the reference fixes context_len = 60)
    vol2, ex2 = _calendar(6 + 180 + 3, 3)
    f = drivers.generate_and_save_ar_sequences(model, vol2, ex2, 0, 6 + 180 + 1, 180, "oos", "prior", context_len=6,
                                               output_dir=str(tmp_path / "ar"))
    d = np.load(f)
    assert set(d.files) == {"surfaces", "indices", "quantiles", "total_horizon", "period_start", "period_end", "method",
                            "sampling_mode", "context_len"}
    assert d["surfaces"].shape == (3, 180, 3, 5, 5) and np.array_equal(d["indices"], [6, 7, 8])
    assert str(d["method"]) == "autoregressive_multistep" and int(d["context_len"]) == 6
