"""GPU parity of CVAEMemRandConditionalPrior (vae/cvae_conditional_prior.py) against the goldens the
reference's own class produced (tests/golden/g8_*, oracle/make_goldens.py cprior): generation from the
learned prior (iid / AR(1) noise, sampled context latents), the Gaussian-Gaussian KL losses, and the
gradients of every parameter -- prior network included -- for train_step and the multi-horizon step."""
import numpy as np
import pytest
import torch

from conftest import cfg_of, golden_names, load_golden
from oracle import vsv_oracle as vo

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _t(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def _n(t):
    return t.detach().cpu().numpy().astype(np.float64)


def _model(g):
    from vae.cvae_conditional_prior import CVAEMemRandConditionalPrior
    cfg = cfg_of(g)
    for k in ("ex_loss_on_ret_only", "ex_feats_loss_type", "kl_weight", "re_feat_weight"):
        v = g["cfg_" + k]
        cfg[k] = v.item() if v.dtype.kind != "U" else str(v)
    cfg.update(device="cuda:0", use_conditional_prior=True)
    model = CVAEMemRandConditionalPrior(dict(cfg))
    W = dict(vo.init_weights(cfg, int(g["seed"]), float(g["scale"]), np.float32))
    W.update(vo.init_prior_weights(cfg, int(g["seed"]), float(g["scale"]), np.float32))
    missing = model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=False)
    assert all(k.startswith("quantile_loss_fn") for k in missing.missing_keys) and not missing.unexpected_keys
    return model, cfg


def _x(g):
    x = {"surface": _t(g["surface"])}
    if "ex_feats" in g:
        x["ex_feats"] = _t(g["ex_feats"])
    return x


@pytest.mark.parametrize("name", golden_names("g8_cprior_gen"))
def test_conditional_prior_generation_vs_reference(name):
    g = load_golden(name)
    model, cfg = _model(g)
    model.eval()
    c = _x(g)
    pm, plv = model.prior_network(c)
    assert np.max(np.abs(_n(pm) - g["prior_mean"])) <= TOL and np.max(np.abs(_n(plv) - g["prior_logvar"])) <= TOL
    sc = bool(int(g["sample_context"]))
    out = model.get_surface_given_conditions(c, horizon=int(g["horizon"]), sample_context=sc, ar_phi=float(g["ar_phi"]),
                                             _eps_future=_t(g["innov"]), _eps_ctx=_t(g["eps_ctx"]) if sc else None)
    if "ex_feats" in g:
        assert np.max(np.abs(_n(out[0]) - g["surf_out"])) <= TOL and np.max(np.abs(_n(out[1]) - g["ex_out"])) <= TOL
    else:
        assert np.max(np.abs(_n(out) - g["surf_out"])) <= TOL
    # on-device noise: right shapes, finite, and a fresh draw per call
    a = model.get_surface_given_conditions(c, horizon=int(g["horizon"]), sample_context=sc, ar_phi=float(g["ar_phi"]))
    b = model.get_surface_given_conditions(c, horizon=int(g["horizon"]), sample_context=sc, ar_phi=float(g["ar_phi"]))
    a, b = (a[0], b[0]) if "ex_feats" in g else (a, b)
    assert a.shape == g["surf_out"].shape and torch.isfinite(a).all() and not torch.equal(a, b)


def _check(g, prefix, model, rtol=1e-3):
    total = 0.0
    for n, p in model.named_parameters():
        gr = _n(p.grad)
        total += float((gr ** 2).sum())
        want_norm = float(g[f"{prefix}/gnorm/{n}"])
        scale = max(want_norm / np.sqrt(gr.size), 1e-12)
        idx, want = g[f"{prefix}/gidx/{n}"], g[f"{prefix}/gval/{n}"]
        assert np.max(np.abs(gr.ravel()[idx] - want)) <= rtol * max(np.max(np.abs(want)), 10 * scale) + 1e-9, n
        assert abs(np.sqrt((gr ** 2).sum()) - want_norm) <= rtol * want_norm + 1e-9, n
    tn = float(g[f"{prefix}/grad_total_norm"])
    assert abs(np.sqrt(total) - tn) <= rtol * tn


@pytest.mark.parametrize("name", golden_names("g8_cprior_train"))
def test_conditional_prior_losses_and_gradients_vs_reference(name):
    g = load_golden(name)
    model, cfg = _model(g)
    x, h = _x(g), int(g["horizon"])
    model.horizon = h
    model.eval()
    losses, _, _ = model._cond_loss_terms(x, h, False, eps=_t(g["eps"]))
    for k in ("loss", "re_surface", "reconstruction_loss", "kl_loss"):
        want = float(g["loss_" + k])
        assert abs(float(losses[k]) - want) <= 1e-3 * abs(want) + 1e-9, (k, float(losses[k]), want)
    # compute_kl_loss on the forward's own outputs (what analysis code calls)
    out = model.forward(x)
    kl = model.compute_kl_loss(out[-3], out[-2], x)
    assert abs(float(kl) - float(g["loss_kl_loss"])) <= 1e-3 * abs(float(g["loss_kl_loss"]))
    # train_step's gradients (un-clipped): base model through the HIP BPTT with the Gaussian-Gaussian KL
    # gradient injected, prior network through its own encoder pass, folded heads unfolded
    model.train()
    model.zero_grad()
    losses, gp, gpp = model._cond_loss_terms(x, h, True, eps=_t(g["eps"]))
    assert abs(float(losses["loss"]) - float(g["train_loss"])) <= 1e-3 * abs(float(g["train_loss"]))
    model._accumulate_grads(gp)
    model._accumulate_prior_grads(gpp)
    _check(g, "ts", model)
    # multi-horizon: mean over horizons, one injected posterior draw each
    model.zero_grad()
    hs = [int(v) for v in g["horizons"]]
    tot, per = 0.0, []
    for i, hh in enumerate(hs):
        losses, gp, gpp = model._cond_loss_terms(x, hh, True, eps=_t(g["eps_h"][i]))
        model._accumulate_grads(gp, 1.0 / len(hs))
        model._accumulate_prior_grads(gpp, 1.0 / len(hs))
        tot += float(losses["loss"]) / len(hs)
        per.append(float(losses["loss"]))
    assert abs(tot - float(g["mh_loss"])) <= 1e-3 * abs(float(g["mh_loss"]))
    assert np.allclose(per, g["mh_horizon_losses"], rtol=1e-3)
    _check(g, "mh", model)


def test_conditional_prior_class_api(tmp_path):
    """Public steps run end to end, return the reference's keys and types, move the prior network's
    weights, and the subclass checkpoint format round-trips."""
    g = load_golden("g8_cprior_train_ex5_B6_T21_h3")
    model, cfg = _model(g)
    x = _x(g)
    model.horizon = int(g["horizon"])
    model.eval()
    t = model.test_step(x)
    assert set(t) == {"loss", "re_surface", "re_ex_feats", "reconstruction_loss", "kl_loss"}
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    before = model.prior_network.prior_mean_layer.weight.detach().clone()
    out = model.train_step(x, opt)
    assert set(out) == set(t) and torch.is_tensor(out["loss"]) and torch.isfinite(out["loss"])
    assert not torch.equal(before, model.prior_network.prior_mean_layer.weight)
    out = model.train_step_multihorizon(x, opt, horizons=[1, 7, 14])
    assert set(out) == {"loss", "horizon_losses", "kl_loss"} and set(out["horizon_losses"]) == {1, 7, 14}
    path = str(tmp_path / "cp.pt")
    model.save_model(path)
    from vae.cvae_conditional_prior import CVAEMemRandConditionalPrior
    m2 = CVAEMemRandConditionalPrior(dict(cfg))
    m2.load_weights(dict_to_load=torch.load(path, weights_only=False))
    for (n1, p1), (n2, p2) in zip(model.named_parameters(), m2.named_parameters()):
        assert n1 == n2 and torch.equal(p1, p2), n1
    # use_conditional_prior=False behaves like the base class
    cfg0 = dict(cfg, use_conditional_prior=False)
    m0 = CVAEMemRandConditionalPrior(cfg0)
    assert m0.prior_network is None and len(m0.state_dict()) == 58
