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


# ---- round 2: z = 0 driver and fitted-GMM prior against the reference's own outputs (goldens g11 / g12) ----

def _golden_model(g, device="cuda:0"):
    from conftest import cfg_of
    from vae.cvae_with_mem_randomized import CVAEMemRand
    cfg = cfg_of(g)
    cfg["device"] = device
    W = vo.init_weights(cfg, int(g["seed"]), float(g["scale"]), np.float32)
    model = CVAEMemRand(dict(cfg))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=False)
    model.eval()
    return model, cfg


@pytest.mark.parametrize("name", ["g11_mle_ex5_C5_N40", "g11_mle_noex_C20_N33", "g11_mle_ex12_C60_N35"])
def test_mle_driver_vs_reference(name, tmp_path):
    """generate_surfaces_multiday_mle / generate_surfaces_mle (generate_surfaces_max_likelihood.py:13-103) and the
    generate_quantile_surfaces.py products against the reference's run on the same synthetic calendar."""
    from conftest import load_golden
    import generate_surfaces_max_likelihood as mle              # the import-path mirror
    from vsv import drivers
    g = load_golden(name)
    model, cfg = _golden_model(g)
    C, N, start = int(g["ctx_len"]), int(g["n_days"]), int(g["start_day"])
    chk = bool(int(g["check_ex_feats"]))
    (p05, p50, p95), ex = mle.generate_surfaces_multiday_mle(model, g["ex"], g["vol"], start, N, 1, check_ex_feats=chk, ctx_len=C)
    assert p05.dtype == np.float64 and p05.shape == (N, 5, 5) and ex.shape == g["ex_out"].shape
    for got, k in ((p05, "p05"), (p50, "p50"), (p95, "p95")):
        assert _err(got, g[k]) <= TOL
    assert _err(ex, g["ex_out"]) <= TOL                            # zeros unless check_ex_feats, like the reference
    use_ex = cfg["ex_feats_dim"] > 0
    (a, b, c), e1 = mle.generate_surfaces_mle(model, g["ex"], g["vol"], start + 3, C, 1, use_ex, chk)
    assert _err(np.stack([a, b, c]), np.stack([g["p05"][3], g["p50"][3], g["p95"][3]])) <= TOL
    assert (e1 is None) == (not (use_ex and chk))
    # the two .npz products of generate_quantile_surfaces.py
    f_st, f_mle = drivers.generate_quantile_products(model, g["ex"], g["vol"], start, N, C, str(tmp_path), "m", check_ex_feats=chk)
    want_keys = {"surfaces_p05", "surfaces_p50", "surfaces_p95"} | ({"ex_feats"} if chk else set())
    assert f_st.name == f"m_quantile_gen{C}.npz" and f_mle.name == f"m_quantile_mle_gen{C}.npz"
    zm, zs = np.load(f_mle), np.load(f_st)
    assert set(zm.files) == want_keys and set(zs.files) == want_keys
    assert _err(zm["surfaces_p50"], g["p50"]) <= TOL and zs["surfaces_p50"].shape == (N, 5, 5)
    assert np.max(np.abs(zs["surfaces_p50"] - zm["surfaces_p50"])) > 1e-6      # the stochastic product is not the z = 0 one


@pytest.mark.parametrize("name", ["g12_gmm_ex5_C8_H4_K3", "g12_gmm_noex12_C20_H7_K5"])
def test_fitted_gmm_prior_vs_reference(name, tmp_path):
    """vsv_gmm_sample with the reference's multinomial picks and normal draws injected, and the generation from
    those latents, against _sample_from_fitted_prior / get_surface_given_conditions(prior_mode="fitted") of the
    reference (:605-640, :689-691); then the on-device (Philox) draw distributionally."""
    from conftest import load_golden
    from scipy import stats
    g = load_golden(name)
    model, cfg = _golden_model(g)
    eng = model._engine()
    B, C = g["surface"].shape[:2]
    H, L, K = int(g["horizon"]), cfg["latent_dim"], g["means"].shape[0]
    path = str(tmp_path / "gmm.pt")
    torch.save({"means": torch.from_numpy(g["means"]), "covariances": torch.from_numpy(g["covariances"]),
                "weights": torch.from_numpy(g["weights"]), "n_components": K}, path)
    model.load_fitted_prior(path)
    fp = model.fitted_prior
    z = eng.gmm_sample(fp["means"], fp["cholesky"], fp["weights"], B * (C + H), comp=torch.from_numpy(g["comp"]),
                       eps=torch.from_numpy(g["eps"]))
    assert _err(_n(z).reshape(B, C + H, L), g["z"]) <= 1e-5
    c = {"surface": _t(g["surface"])}
    if "ex_feats" in g:
        c["ex_feats"] = _t(g["ex_feats"])
    out = model.get_surface_given_conditions(c, z=z.reshape(B, C + H, L).clone(), horizon=H)
    if "ex_feats" in g:
        assert _err(_n(out[0]), g["surf_out"]) <= TOL and _err(_n(out[1]), g["ex_out"]) <= TOL
    else:
        assert _err(_n(out), g["surf_out"]) <= TOL
    # device draw: component frequencies follow the weights, whitened residuals are standard normal, seeds matter
    n = 200_000
    zz, comp = eng.gmm_sample(fp["means"], fp["cholesky"], fp["weights"], n, seed=11, want_comp=True)
    comp = comp.cpu().numpy()
    freq = np.bincount(comp, minlength=K) / n
    assert np.max(np.abs(freq - g["weights"])) <= 5 * np.sqrt(0.25 / n)
    chol = np.linalg.cholesky(g["covariances"])
    resid = _n(zz) - g["means"][comp]
    white = np.linalg.solve(chol[comp], resid[..., None])[..., 0]
    assert stats.kstest(white.ravel()[:100_000], "norm").pvalue > 1e-3
    assert abs(white.mean()) < 0.01 and abs(white.std() - 1) < 0.01
    assert np.max(np.abs(np.corrcoef(white.T) - np.eye(L))) < 0.02
    z2 = eng.gmm_sample(fp["means"], fp["cholesky"], fp["weights"], 1000, seed=11)
    z3 = eng.gmm_sample(fp["means"], fp["cholesky"], fp["weights"], 1000, seed=12)
    assert torch.equal(z2, zz[:1000]) and not torch.equal(z2, z3)
    # the model API draws through it
    torch.manual_seed(0)
    zs = model._sample_from_fitted_prior(50, C + H)
    assert zs.shape == (50, C + H, L) and zs.dtype == torch.float64
    s = model.get_surface_given_conditions(c, horizon=H, prior_mode="fitted")
    s = s[0] if isinstance(s, tuple) else s
    assert s.shape == (B, H, 3, 5, 5) and bool(torch.isfinite(s).all())


def test_legacy_model_monte_carlo_product_vs_cpu_oracle():
    """eval_utils.generate_surfaces_multiday with model_type = the legacy CVAEMem: the original
    (days, num_vaes, 5, 5) Monte-Carlo product, z drawn at EVERY position (cvae_with_mem.py:547), against the
    CPU oracle fed the same Philox draws."""
    from eval_scripts import eval_utils as eu
    from vae.cvae_with_mem import CVAEMem
    from vae.utils import set_seeds
    from vsv import paths
    L, X, C, N, S = 5, 3, 5, 7, 24
    cfg = vo.make_config(L, X, device="cuda:0", seq_len=C + 1, ctx_len=C)
    model = CVAEMem(dict(cfg))
    W = vo.init_weights(dict(cfg, quantiles=[0.5]), 6, 2.0, np.float32)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=True)
    model.eval()
    vol, ex = _calendar(30, X)
    set_seeds(1)
    s, e = eu.generate_surfaces_multiday(model, ex, vol, C + 2, N, S, model_type=CVAEMem, check_ex_feats=True, ctx_len=C)
    assert s.shape == (N, S, 5, 5) and e.shape == (N, S, X)
    set_seeds(1)
    seed = model._draw_seed()
    eng = model._engine()
    W64 = {k: v.astype(np.float64) for k, v in W.items()}
    ocfg = dict(cfg, quantiles=[0.5])
    for d in (0, N - 1):
        day = C + 2 + d
        z = _n(eng.philox_normal(seed, day, 1, S, 0, C + 1))                       # (S, C+1, L): every position
        win = np.broadcast_to(vol[day - C:day][None].astype(np.float64), (S, C, 5, 5))
        wex = np.broadcast_to(ex[day - C:day][None].astype(np.float64), (S, C, X))
        ctx_pad = np.zeros((S, C + 1, L))
        ctx_pad[:, :C] = vo.ctx_encoder_forward(W64, ocfg, win, wex)
        ws, we, _ = vo.decoder_forward(W64, ocfg, np.concatenate([ctx_pad, z], axis=-1))
        assert _err(s[d], ws[:, C, 0]) <= TOL and _err(e[d], we[:, C]) <= TOL
    # the same call on a CVAEMemRand keeps that class's semantics (posterior mean at the context positions)
    raw, _ = paths.legacy_prior_windows(eng, _t(vol), _t(ex), C + 2, 2, C, 1, 3, seed=seed)
    assert raw.shape == (2, 3, 1, 1, 5, 5)
