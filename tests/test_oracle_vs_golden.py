"""Pins the CPU oracle (oracle/vsv_oracle.py) against outputs of the
reference itself (tests/golden/*.npz, made by oracle/make_goldens.py with the
reference imported in the build container).  CPU only."""
import numpy as np
import pytest

from conftest import cfg_of, golden_names, load_golden
from oracle import vsv_oracle as vo


def tol(dtype):
    # float64 goldens: the oracle evaluates the same equations in another
    # order; float32 goldens carry the reference's own float32 rounding.
    return 1e-9 if dtype == np.float64 else 2e-5


@pytest.mark.parametrize("name", golden_names("g1_"))
def test_forward_and_losses(name):
    g = load_golden(name)
    cfg = cfg_of(g)
    dt = g["surface"].dtype
    W = vo.init_weights(cfg, int(g["seed"]), float(g["scale"]), dt)
    ex = g.get("ex_feats")
    H = int(g["horizon"])
    s, e, zm, zl, z = vo.forward(W, cfg, g["surface"], ex, g["eps"], H)
    assert s.shape == g["recon"].shape
    for got, want in ((s, g["recon"]), (zm, g["z_mean"]), (zl, g["z_log_var"]), (z, g["z"])):
        assert np.max(np.abs(got - want)) <= tol(dt)
    if ex is not None:
        assert np.max(np.abs(e - g["ex_recon"])) <= tol(dt)
    losses = vo.test_step(W, cfg, g["surface"], ex, g["eps"], H)
    for k, v in losses.items():
        want = float(g["loss_" + k])
        assert abs(v - want) <= 1e-3 * abs(want) + 1e-12, (k, v, want)


@pytest.mark.parametrize("name", golden_names("g2_"))
def test_prior_generation(name):
    g = load_golden(name)
    cfg = cfg_of(g)
    dt = g["surface"].dtype
    W = vo.init_weights(cfg, int(g["seed"]), float(g["scale"]), dt)
    ex = g.get("ex_feats")
    H = int(g["horizon"])
    z = g["z"].copy()
    s, e = vo.get_surface_given_conditions(W, cfg, g["surface"], ex, z, H)
    assert np.max(np.abs(s - g["surf_out"])) <= tol(dt)
    assert np.max(np.abs(z - g["z_after"])) <= tol(dt)          # in-place overwrite (:700)
    if ex is not None:
        assert np.max(np.abs(e - g["ex_out"])) <= tol(dt)
    zm, zl = vo.encoder_forward(W, cfg, g["surface"], ex)
    assert np.max(np.abs(zm - g["enc_z_mean"])) <= tol(dt)
    assert np.max(np.abs(zl - g["enc_z_log_var"])) <= tol(dt)
    ctx = vo.ctx_encoder_forward(W, cfg, g["surface"], ex)
    assert np.max(np.abs(ctx - g["ctx_embedding"])) <= tol(dt)
    # the shared-prefix restructuring the HIP path uses gives the same values
    C = g["surface"].shape[1]
    s2, e2 = vo.prior_mc_shared_prefix(W, cfg, g["surface"], ex, g["z"][:, None, C:])
    assert np.max(np.abs(s2[:, 0] - g["surf_out"])) <= tol(dt)
    if ex is not None:
        assert np.max(np.abs(e2[:, 0] - g["ex_out"])) <= tol(dt)


@pytest.mark.parametrize("name", golden_names("g6_") + golden_names("g13_"))
def test_autoregressive(name):
    g = load_golden(name)
    cfg = cfg_of(g)
    W = vo.init_weights(cfg, int(g["seed"]), float(g["scale"]), np.float64)
    s, e = vo.generate_autoregressive_sequence(W, cfg, g["surface"], g.get("ex_feats"), g["z_steps"])
    assert s.shape == g["surf_out"].shape
    assert np.max(np.abs(s - g["surf_out"])) <= 1e-9
    if e is not None:
        assert np.max(np.abs(e - g["ex_out"])) <= 1e-9


@pytest.mark.parametrize("name", golden_names("g7_"))
def test_autoregressive_offset(name):
    g = load_golden(name)
    cfg = cfg_of(g)
    W = vo.init_weights(cfg, int(g["seed"]), float(g["scale"]), np.float64)
    s, e = vo.generate_autoregressive_offset(W, cfg, g["surface"], g.get("ex_feats"),
                                             list(g["z_steps"]), int(g["horizon"]))
    assert s.shape == g["surf_out"].shape
    assert np.max(np.abs(s - g["surf_out"])) <= 1e-9
    if e is not None:
        assert np.max(np.abs(e - g["ex_out"])) <= 1e-9


def test_quantile_loss_kats():
    g = load_golden("g3_quantile_loss")
    q = [0.05, 0.5, 0.95]
    assert abs(vo.quantile_loss(g["preds"], g["target"], q) - float(g["loss_w111"])) < 1e-7
    assert abs(vo.quantile_loss(g["preds"], g["target"], q, [5, 1, 5]) - float(g["loss_w515"])) < 1e-7
    # the reference's own known-answer cases (test_code/test_quantile_decoder.py:58-94)
    assert vo.quantile_loss(np.ones((2, 1, 3, 5, 5)), np.ones((2, 1, 5, 5)), q) == 0.0
    yt = np.ones((1, 1, 2, 2))
    assert abs(vo.quantile_loss(np.zeros((1, 1, 3, 2, 2)), yt, q) - 0.5) < 1e-12
    assert abs(vo.quantile_loss(np.full((1, 1, 3, 2, 2), 2.0), yt, q) - 0.5) < 1e-12
    assert abs(float(g["kat_under"]) - 0.5) < 1e-6 and abs(float(g["kat_over"]) - 0.5) < 1e-6
    assert float(g["kat_equal"]) == 0.0


def test_philox_known_answers():
    # Random123 kat_vectors for philox4x32-10
    out = vo.philox4x32_10(0, 0, 0, 0, 0, 0)
    assert [int(x) for x in out] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    f = 0xFFFFFFFF
    out = vo.philox4x32_10(f, f, f, f, f, f)
    assert [int(x) for x in out] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    out = vo.philox4x32_10(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0)
    assert [int(x) for x in out] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_philox_normal_is_standard_normal():
    from scipy import stats
    z = vo.philox_normal(7, 3, 16, 500, 0, 4, 5)
    assert z.shape == (16, 500, 4, 5) and z.dtype == np.float32
    assert stats.kstest(z.ravel().astype(np.float64), "norm").pvalue > 1e-3
    # sharding invariance: a sub-range of days gives the same numbers
    z2 = vo.philox_normal(7, 5, 4, 500, 0, 4, 5)
    assert np.array_equal(z2, z[2:6])


# ---- the torch-CPU port (oracle/torch_port.py: bench cpu_baseline + autograd oracle) ----

def _port(g, dt):
    import torch
    from oracle.torch_port import TorchPort
    cfg = cfg_of(g)
    W = vo.init_weights(cfg, int(g["seed"]), float(g["scale"]), dt)
    tdt = torch.float64 if dt == np.float64 else torch.float32
    return TorchPort(cfg, W, tdt), cfg, tdt


@pytest.mark.parametrize("name", golden_names("g1_"))
def test_torch_port_forward_and_losses(name):
    import torch
    g = load_golden(name)
    dt = g["surface"].dtype
    port, cfg, tdt = _port(g, dt)
    ex = g.get("ex_feats")
    T = lambda a: None if a is None else torch.tensor(a, dtype=tdt)      # noqa: E731
    with torch.no_grad():
        s, e, zm, zl, z = port.forward(T(g["surface"]), T(ex), T(g["eps"]), int(g["horizon"]))
        losses = port.losses(T(g["surface"]), T(ex), T(g["eps"]), int(g["horizon"]),
                             cfg.get("quantile_loss_weights"))
    for got, want in ((s, g["recon"]), (zm, g["z_mean"]), (zl, g["z_log_var"]), (z, g["z"])):
        assert np.max(np.abs(got.numpy() - want)) <= tol(dt)
    if ex is not None:
        assert np.max(np.abs(e.numpy() - g["ex_recon"])) <= tol(dt)
    for k, v in losses.items():
        want = float(g["loss_" + k])
        assert abs(float(v) - want) <= 1e-3 * abs(want) + 1e-12, (k, float(v), want)


@pytest.mark.parametrize("name", golden_names("g2_"))
def test_torch_port_prior_generation(name):
    import torch
    g = load_golden(name)
    dt = g["surface"].dtype
    port, cfg, tdt = _port(g, dt)
    ex = g.get("ex_feats")
    z = torch.tensor(g["z"], dtype=tdt)
    s, e = port.get_surface_given_conditions(torch.tensor(g["surface"], dtype=tdt),
                                             None if ex is None else torch.tensor(ex, dtype=tdt), z, int(g["horizon"]))
    assert np.max(np.abs(s.numpy() - g["surf_out"])) <= tol(dt)
    assert np.max(np.abs(z.numpy() - g["z_after"])) <= tol(dt)
    if ex is not None:
        assert np.max(np.abs(e.numpy() - g["ex_out"])) <= tol(dt)


@pytest.mark.parametrize("name", golden_names("g4_"))
def test_torch_port_gradients(name):
    """The torch-CPU port as an autograd oracle: its gradients equal the reference's."""
    import torch
    from oracle.torch_port import TorchPort
    g = load_golden(name)
    cfg = cfg_of(g)
    for k in ("ex_loss_on_ret_only", "ex_feats_loss_type", "kl_weight", "re_feat_weight"):
        v = g["cfg_" + k]
        cfg[k] = v.item() if v.dtype.kind != "U" else str(v)
    W = vo.init_weights(cfg, int(g["seed"]), float(g["scale"]), np.float64)
    port = TorchPort(cfg, W, torch.float64)
    T = lambda a: None if a is None else torch.tensor(a, dtype=torch.float64)      # noqa: E731
    losses = port.losses(T(g["surface"]), T(g.get("ex_feats")), T(g["eps"]), int(g["horizon"]),
                         [float(x) for x in g["cfg_quantile_loss_weights"]])
    losses["loss"].backward()
    for k in ("loss", "re_surface", "kl_loss"):
        assert abs(float(losses[k]) - float(g["loss_" + k])) <= 1e-9 * max(1.0, abs(float(g["loss_" + k])))
    total = 0.0
    for n, p in port.named_reference_parameters().items():
        gr = p.grad.numpy()
        total += float((gr ** 2).sum())
        assert abs(np.sqrt((gr ** 2).sum()) - float(g["gnorm/" + n])) <= 1e-8 * max(1.0, float(g["gnorm/" + n])), n
    assert abs(np.sqrt(total) - float(g["grad_total_norm"])) <= 1e-8 * float(g["grad_total_norm"])


def test_dropout_mask_restatement_and_masked_port():
    """The numpy restatement of the HIP path's inter-layer dropout masks: keep rate, scale, seed and
    stream separation; and the torch port's masked two-pass LSTM equals nn.LSTM when the mask is 1."""
    import torch
    from oracle.torch_port import TorchPort
    p = 0.3
    m = vo.dropout_mask(p, 99, 0, 0, 64, 20, 100)
    assert m.shape == (64, 20, 100) and set(np.unique(m)) == {np.float32(0.0), np.float32(1.0) / (np.float32(1.0) - np.float32(p))}
    assert abs(float((m > 0).mean()) - (1 - p)) < 0.01
    assert not np.array_equal(m, vo.dropout_mask(p, 99, 1, 0, 64, 20, 100))
    assert not np.array_equal(m, vo.dropout_mask(p, 100, 0, 0, 64, 20, 100))
    assert np.array_equal(m[:10, :5], vo.dropout_mask(p, 99, 0, 0, 10, 5, 100))       # a pure function of (row, t, unit)
    cfg = vo.make_config(5, 3)
    W = vo.init_weights(cfg, 3, 1.0, np.float32)
    port = TorchPort(cfg, W, torch.float32)
    rng = np.random.default_rng(0)
    B, T, h = 3, 7, 2
    s = torch.from_numpy((0.2 + 0.1 * rng.standard_normal((B, T, 5, 5))).astype(np.float32))
    e = torch.from_numpy((0.01 * rng.standard_normal((B, T, 3))).astype(np.float32))
    eps = torch.from_numpy(rng.standard_normal((B, T, 5)).astype(np.float32))
    a = port.losses(s, e, eps, h)["loss"]
    port.drop_masks = {n: torch.ones(B, T, hd) for n, hd in (("encoder", 100), ("ctx_encoder", 100), ("decoder", 128))}
    b = port.losses(s, e, eps, h)["loss"]
    assert abs(float(a) - float(b)) <= 1e-6 * abs(float(a))


# ---- conditional prior (goldens g8 from the reference's CVAEMemRandConditionalPrior) ----

@pytest.mark.parametrize("name", golden_names("g8_cprior_gen"))
def test_oracle_conditional_prior_generation(name):
    g = load_golden(name)
    cfg = cfg_of(g)
    f64 = name.endswith("f64")
    tol = 1e-9 if f64 else 2e-5
    dt = np.float64 if f64 else np.float32
    W = vo.init_weights(cfg, int(g["seed"]), float(g["scale"]), dt)
    Wp = vo.init_prior_weights(cfg, int(g["seed"]), float(g["scale"]), dt)
    surf, ex = g["surface"], g.get("ex_feats")
    pm, plv = vo.prior_network_forward(Wp, cfg, surf, ex)
    assert np.max(np.abs(pm - g["prior_mean"])) <= tol and np.max(np.abs(plv - g["prior_logvar"])) <= tol
    phi = float(g["ar_phi"])
    fut = g["innov"].astype(dt)
    if phi > 0:
        fut = vo.ar1_noise(fut, phi)
    s, e = vo.cond_prior_generate(W, Wp, cfg, surf, ex, fut, g["eps_ctx"].astype(dt) if int(g["sample_context"]) else None)
    assert np.max(np.abs(s - g["surf_out"])) <= tol
    if ex is not None:
        assert np.max(np.abs(e - g["ex_out"])) <= tol


@pytest.mark.parametrize("name", golden_names("g8_cprior_train"))
def test_oracle_conditional_prior_kl(name):
    g = load_golden(name)
    cfg = cfg_of(g)
    W = vo.init_weights(cfg, int(g["seed"]), float(g["scale"]), np.float64)
    Wp = vo.init_prior_weights(cfg, int(g["seed"]), float(g["scale"]), np.float64)
    surf, ex, h = g["surface"], g.get("ex_feats"), int(g["horizon"])
    C = surf.shape[1] - h
    zm, zl = vo.encoder_forward(W, cfg, surf, ex)
    pm, plv = vo.prior_network_forward(Wp, cfg, surf[:, :C], None if ex is None else ex[:, :C])
    kl = vo.kl_divergence_gaussians(zm[:, :C], zl[:, :C], pm, plv)
    assert abs(kl - float(g["loss_kl_loss"])) <= 1e-9 * max(1.0, abs(kl))


@pytest.mark.parametrize("name", golden_names("g11_"))
def test_mle_generation_driver(name):
    """generate_surfaces_multiday_mle of the reference (z = 0, one call per day) against the oracle."""
    g = load_golden(name)
    cfg = cfg_of(g)
    W = vo.init_weights(cfg, int(g["seed"]), float(g["scale"]), np.float64)
    X = cfg["ex_feats_dim"]
    (p05, p50, p95), e = vo.generate_surfaces_multiday_mle(W, cfg, g["vol"], g["ex"] if X else None, int(g["start_day"]),
                                                         int(g["n_days"]), int(g["ctx_len"]))
    for got, k in ((p05, "p05"), (p50, "p50"), (p95, "p95")):
        assert got.shape == g[k].shape and np.max(np.abs(got - g[k])) <= 1e-9
    if int(g["check_ex_feats"]):
        assert np.max(np.abs(e - g["ex_out"])) <= 1e-9
    else:
        assert not g["ex_out"].any()                    # the reference leaves zeros unless check_ex_feats


@pytest.mark.parametrize("name", golden_names("g12_"))
def test_fitted_gmm_prior(name):
    """_sample_from_fitted_prior and prior_mode="fitted" generation with the reference's own draws."""
    g = load_golden(name)
    cfg = cfg_of(g)
    W = vo.init_weights(cfg, int(g["seed"]), float(g["scale"]), np.float64)
    B, C = g["surface"].shape[:2]
    H = int(g["horizon"])
    z = vo.gmm_sample(g["means"], g["covariances"], g["comp"], g["eps"]).reshape(B, C + H, -1)
    assert np.max(np.abs(z - g["z"])) <= 1e-12
    s, e = vo.get_surface_given_conditions(W, cfg, g["surface"], g.get("ex_feats"), z.copy(), H)
    assert np.max(np.abs(s - g["surf_out"])) <= 1e-9
    if e is not None:
        assert np.max(np.abs(e - g["ex_out"])) <= 1e-9


@pytest.mark.parametrize("name", golden_names("g14_legacy"))
def test_oracle_matches_legacy_one_channel_model(name):
    """The legacy fixed-length CVAEMem (reference vae/cvae_with_mem.py) is the same network with one decoder
    channel: the oracle with quantiles=[0.5] reproduces its forward (uniform noise injected), its MSE / ex / KL
    losses, and its literal generation from a given z (context latents NOT replaced by the posterior mean)."""
    g = load_golden(name)
    X = int(g["cfg_ex_feats_dim"])
    cfg = vo.make_config(int(g["cfg_latent_dim"]), X, quantiles=[0.5])
    W = vo.init_weights(cfg, int(g["seed"]), float(g["scale"]), np.float64)
    T, C = int(g["seq_len"]), int(g["ctx_len"])
    ex = g["ex_feats"] if X else None
    out = vo.forward(W, cfg, g["surface"], ex, g["eps"], T - C)
    assert np.max(np.abs(out[0][:, :, 0] - g["recon"])) <= 1e-9
    assert np.max(np.abs(out[2] - g["z_mean"])) <= 1e-9 and np.max(np.abs(out[4] - g["z"])) <= 1e-9
    mse = float(np.mean((out[0][:, :, 0] - g["surface"][:, C:]) ** 2))
    assert abs(mse - float(g["loss_re_surface"])) <= 1e-12
    kl = float(np.mean(np.sum(-0.5 * (1 + out[3] - np.exp(out[3]) - out[2] ** 2), axis=-1)))
    assert abs(kl - float(g["loss_kl_loss"])) <= 1e-9 * max(1.0, abs(kl))
    if X:
        assert np.max(np.abs(out[1] - g["ex_recon"])) <= 1e-9
    # literal generation: decoder over all T positions on cat([padded context embedding, z])
    B, L = g["surface"].shape[0], cfg["latent_dim"]
    ctx_pad = np.zeros((B, T, L))
    ctx_pad[:, :C] = vo.ctx_encoder_forward(W, cfg, g["surface"][:, :C], None if ex is None else ex[:, :C])
    s, e, _ = vo.decoder_forward(W, cfg, np.concatenate([ctx_pad, g["z_given"]], axis=-1))
    assert np.max(np.abs(s[:, C:, 0] - g["surf_out"])) <= 1e-9
    if X:
        assert np.max(np.abs(e[:, C:] - g["ex_out"])) <= 1e-9
