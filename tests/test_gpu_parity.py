"""GPU parity tests: the HIP path (through the C ABI of libvsv_hip.so) against
the reference's own outputs (tests/golden) and against the CPU oracle on seeded
inputs.  Tolerances are the north-star's: max abs err <= 1e-4 on vols."""
import os

import numpy as np
import pytest
import torch

from conftest import cfg_of, golden_names, load_golden, record_margin
from oracle import vsv_oracle as vo

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _engine(cfg, seed, scale):
    from vsv.engine import Engine
    W = vo.init_weights(cfg, seed, scale, np.float32)
    return Engine(cfg, W, "cuda:0"), W


def _t(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def _n(t):
    return t.detach().cpu().numpy().astype(np.float64)


def err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    e = float(np.max(np.abs(a - b))) if a.size else 0.0
    record_margin(e)
    return e


@pytest.mark.parametrize("name", golden_names("g2_"))
def test_encoders_and_prior_generation_vs_reference(name):
    from vsv import paths
    g = load_golden(name)
    cfg = cfg_of(g)
    eng, _ = _engine(cfg, int(g["seed"]), float(g["scale"]))
    surf, ex = _t(g["surface"]), _t(g.get("ex_feats"))
    C, H = g["surface"].shape[1], int(g["horizon"])
    zm, zl = paths.encode_both(eng, surf, ex)
    assert err(_n(zm), g["enc_z_mean"]) <= TOL
    assert err(_n(zl), g["enc_z_log_var"]) <= TOL
    ctx = eng.encode(1, surf, ex)
    assert err(_n(ctx), g["ctx_embedding"]) <= TOL
    # shared-prefix prior generation with the reference's injected future latents
    eps = _t(g["z"][:, C:])
    s, e, _ = paths.prior_generate(eng, surf, ex, H, 1, eps=eps)
    assert s.shape == g["surf_out"].shape
    assert err(_n(s), g["surf_out"]) <= TOL
    if ex is not None:
        assert err(_n(e), g["ex_out"]) <= TOL
    # literal evaluation: decoder over all C+H steps on cat([ctx_pad, z_after])
    s2, e2 = paths.decode_given_z(eng, ctx, _t(g["z_after"]), C)
    assert err(_n(s2), g["surf_out"]) <= TOL
    if ex is not None:
        assert err(_n(e2), g["ex_out"]) <= TOL


@pytest.mark.parametrize("name", golden_names("g1_"))
def test_forward_vs_reference(name):
    from vsv import paths
    g = load_golden(name)
    cfg = cfg_of(g)
    eng, _ = _engine(cfg, int(g["seed"]), float(g["scale"]))
    surf, ex = _t(g["surface"]), _t(g.get("ex_feats"))
    s, e, zm, zl, z = paths.forward(eng, surf, ex, int(g["horizon"]), eps=_t(g["eps"]))
    assert s.shape == g["recon"].shape
    assert err(_n(zm), g["z_mean"]) <= TOL
    assert err(_n(zl), g["z_log_var"]) <= TOL
    assert err(_n(z), g["z"]) <= TOL
    assert err(_n(s), g["recon"]) <= TOL
    if ex is not None:
        assert err(_n(e), g["ex_recon"]) <= TOL


@pytest.mark.parametrize("L,X,D,S,C,H,mem", [(5, 0, 3, 40, 20, 1, "lstm"), (12, 3, 5, 24, 60, 7, "lstm"),
                                             (5, 3, 2, 1000, 5, 2, "lstm"), (5, 3, 3, 300, 7, 4, "gru"),
                                             (5, 3, 3, 300, 7, 4, "rnn")])
def test_prior_mc_vs_oracle_ragged_tiles(L, X, D, S, C, H, mem):
    """rows = D*S is not a multiple of the 16-row tile and tiles straddle days."""
    from vsv import paths
    cfg = vo.make_config(L, X, mem_type=mem)
    eng, W = _engine(cfg, 77, 2.0)
    rng = np.random.default_rng(5)
    surf = (0.2 + 0.1 * rng.standard_normal((D, C, 5, 5))).astype(np.float32)
    ex = (0.01 * rng.standard_normal((D, C, X))).astype(np.float32) if X else None
    eps = rng.standard_normal((D, S, H, L)).astype(np.float32)
    W64 = {k: v.astype(np.float64) for k, v in W.items()}
    want_s, want_e = vo.prior_mc_shared_prefix(W64, cfg, surf.astype(np.float64),
                                               None if ex is None else ex.astype(np.float64),
                                               eps.astype(np.float64))
    s, e, _ = paths.prior_generate(eng, _t(surf), _t(ex), H, S, eps=_t(eps.reshape(D * S, H, L)))
    assert err(_n(s).reshape(want_s.shape), want_s) <= TOL
    if X:
        assert err(_n(e).reshape(want_e.shape), want_e) <= TOL
    # sample-minor layout holds the same numbers
    s1, _, _ = paths.prior_generate(eng, _t(surf), _t(ex), H, S, eps=_t(eps.reshape(D * S, H, L)), out_mode=1)
    assert s1.shape == (D, H, 75, S)
    ref = s.view(D, S, H, 75).permute(0, 2, 3, 1)
    assert torch.equal(s1, ref)


def test_fused_mc_decode_many_rows_vs_oracle():
    """> 32768 rows: the one-wave-per-tile schedule with the fused decoder head (the
    MC production path), ragged last tile, tiles straddling days, both output layouts."""
    from vsv import paths
    for L, X, D, S, C, H, mem in ((12, 3, 37, 1000, 4, 2, "lstm"), (5, 0, 67, 500, 3, 1, "lstm"),
                                  (5, 3, 35, 1000, 3, 3, "gru"), (12, 0, 40, 1000, 3, 2, "rnn")):
        cfg = vo.make_config(L, X, mem_type=mem)
        eng, W = _engine(cfg, 21, 2.0)
        rng = np.random.default_rng(17)
        surf = (0.2 + 0.1 * rng.standard_normal((D, C, 5, 5))).astype(np.float32)
        ex = (0.01 * rng.standard_normal((D, C, X))).astype(np.float32) if X else None
        eps = rng.standard_normal((D, S, H, L)).astype(np.float32)
        want_s, want_e = vo.prior_mc_shared_prefix(W, cfg, surf, ex, eps)
        s, e, _ = paths.prior_generate(eng, _t(surf), _t(ex), H, S, eps=_t(eps.reshape(D * S, H, L)))
        assert err(_n(s).reshape(want_s.shape), want_s) <= TOL
        if X:
            assert err(_n(e).reshape(want_e.shape), want_e) <= TOL
        s1, _, _ = paths.prior_generate(eng, _t(surf), _t(ex), H, S, eps=_t(eps.reshape(D * S, H, L)), out_mode=1)
        assert torch.equal(s1, s.view(D, S, H, 75).permute(0, 2, 3, 1))


def test_percentile_matches_numpy():
    cfg = vo.make_config(5, 0)
    eng, _ = _engine(cfg, 0, 1.0)
    rng = np.random.default_rng(1)
    # S <= 1024: wavefront-level register sort (every lane-count class and its edges); above: workgroup-level
    # register sort on 1, 2, 4 or 8 waves (2048 elements each) with its edges and lengths that are not multiples of 4
    for S in (1000, 10000, 37, 1, 2, 3, 63, 64, 65, 128, 200, 256, 300, 512, 513, 1023, 1024, 1025, 2048,
              2049, 3001, 4096, 4097, 7777, 8192, 8193, 12345, 16383, 16384):
        x = rng.standard_normal((4, 25, S)).astype(np.float32)
        got = _n(eng.percentile(_t(x), (5, 50, 95), inner=25))          # (4, 3, 25)
        want = np.percentile(x.astype(np.float64), [5, 50, 95], axis=-1).transpose(1, 0, 2)
        assert got.shape == want.shape
        assert err(got, want) <= 1e-6, S
    # ties and infinities do not break the selection
    x = np.zeros((1, 25, 100), dtype=np.float32)
    x[..., 50:] = 1.0
    got = _n(eng.percentile(_t(x), (5, 50, 95), inner=25))
    want = np.percentile(x, [5, 50, 95], axis=-1).transpose(1, 0, 2)
    assert err(got, want) <= 1e-7
    # one decoder channel of a (groups, Q*25, S) tensor, read in place through the group stride
    for S in (1000, 5000):
        x = rng.standard_normal((6, 3, 25, S)).astype(np.float32)
        for ch in range(3):
            got = _n(eng.percentile(_t(x).view(6, 75, S), (5, 50, 95), inner=25, channel=ch, n_channels=3))
            want = np.percentile(x[:, ch].astype(np.float64), [5, 50, 95], axis=-1).transpose(1, 0, 2)
            assert got.shape == want.shape and err(got, want) <= 1e-6, (S, ch)
    # the extremes and ties on the workgroup-level kernel
    x = rng.standard_normal((2, 25, 9000)).astype(np.float32)
    x[..., 4000:4600] = x[..., :1]
    qs = (0, 0.01, 50, 99.99, 100)
    got = _n(eng.percentile(_t(x), qs, inner=25))
    want = np.percentile(x.astype(np.float64), list(qs), axis=-1).transpose(1, 0, 2)
    assert err(got, want) <= 1e-6
    # other percentile sets, including the extremes
    x = rng.standard_normal((2, 25, 777)).astype(np.float32)
    qs = (0, 1, 25, 33.3, 75, 99, 100)
    got = _n(eng.percentile(_t(x), qs, inner=25))
    want = np.percentile(x.astype(np.float64), list(qs), axis=-1).transpose(1, 0, 2)
    assert err(got, want) <= 1e-6


def test_philox_matches_oracle_and_fused_mode():
    from scipy import stats
    cfg = vo.make_config(5, 0)
    eng, _ = _engine(cfg, 3, 2.0)
    L = 5
    z = _n(eng.philox_normal(1234, 7, 6, 50, 2, 3))                 # (6*50, 3, 5)
    want = vo.philox_normal(1234, 7, 6, 50, 2, 3, L).reshape(300, 3, L)
    assert err(z, want) <= 1e-4                                        # device vs numpy log/sin/cos
    big = _n(eng.philox_normal(99, 0, 64, 1000, 0, 1)).ravel()
    assert stats.kstest(big, "norm").pvalue > 1e-3
    assert abs(big.mean()) < 0.01 and abs(big.std() - 1) < 0.01
    # fused Philox decode == decode with the materialised noise injected (bit-exact)
    from vsv import paths
    rng = np.random.default_rng(2)
    D, S, C, H = 3, 50, 20, 4
    surf = _t(0.2 + 0.1 * rng.standard_normal((D, C, 5, 5)))
    eps = eng.philox_normal(55, 100, D, S, 0, H)
    a, _, _ = paths.prior_generate(eng, surf, None, H, S, seed=55, day0=100)
    b, _, _ = paths.prior_generate(eng, surf, None, H, S, eps=eps)
    assert torch.equal(a, b)
    # sharding invariance: days [1,3) alone give the same samples
    c, _, _ = paths.prior_generate(eng, surf[1:], None, H, S, seed=55, day0=101)
    assert torch.equal(c, a.view(D, S, H, 3, 5, 5)[1:].reshape(c.shape))


def test_prior_mc_windows_reduction_vs_oracle():
    """End-to-end MC backfill on a small calendar: Philox noise materialised, fed
    to the oracle, np.percentile on the oracle's samples."""
    from vsv import paths
    L, X, C, H, S, N = 5, 3, 5, 2, 64, 6
    cfg = vo.make_config(L, X)
    eng, W = _engine(cfg, 11, 2.0)
    rng = np.random.default_rng(9)
    vol = (0.2 + 0.1 * rng.standard_normal((N + C, 5, 5))).astype(np.float32)
    ex = (0.01 * rng.standard_normal((N + C, X))).astype(np.float32)
    q = paths.prior_mc_windows(eng, _t(vol), _t(ex), C, N, C, H, S, seed=5, chunk_days=4, channel=None)
    assert q.shape == (N, H, 3, 3, 5, 5)
    eps = _n(eng.philox_normal(5, C, N, S, 0, H)).reshape(N, S, H, L)
    W64 = {k: v.astype(np.float64) for k, v in W.items()}
    win = np.stack([vol[d:d + C] for d in range(N)]).astype(np.float64)
    wex = np.stack([ex[d:d + C] for d in range(N)]).astype(np.float64)
    smp, _ = vo.prior_mc_shared_prefix(W64, cfg, win, wex, eps)           # (N,S,H,3,5,5)
    want = np.percentile(smp, [5, 50, 95], axis=1).transpose(1, 2, 0, 3, 4, 5)   # (N,H,3q,3ch,5,5)
    assert err(_n(q), want) <= TOL
    q1 = paths.prior_mc_windows(eng, _t(vol), _t(ex), C, N, C, H, S, seed=5, chunk_days=256, channel=1)
    assert err(_n(q1), want[:, :, :, 1]) <= TOL


@pytest.mark.parametrize("name", golden_names("g6_"))
def test_autoregressive_sequence_vs_reference(name):
    """generate_autoregressive_sequence: 30 one-day steps with p50 feedback; the golden is the
    reference's float64 run, the HIP path computes in float32 (errors compound over the steps)."""
    from vsv import paths
    g = load_golden(name)
    cfg = cfg_of(g)
    eng, _ = _engine(cfg, int(g["seed"]), float(g["scale"]))
    surf, ex = _t(g["surface"]), _t(g.get("ex_feats"))
    C = g["surface"].shape[1]
    eps = _t(np.ascontiguousarray(g["z_steps"][:, :, C].transpose(1, 0, 2)))     # (B, H, L)
    s, e = paths.ar_generate(eng, surf, ex, int(g["horizon"]), 1, eps=eps)
    assert s.shape == g["surf_out"].shape
    assert err(_n(s), g["surf_out"]) <= TOL
    if ex is not None:
        assert err(_n(e), g["ex_out"]) <= TOL
    # the shared-prefix schedule (encoders / decoder continue from the state after the days that are
    # still original context) runs the same products in the same order
    # (the first step from a shared state adds bias + W_hh . h_prev before the input product, the literal schedule
    # after it: last-place differences, which an ungated tanh RNN carries through 30 fed-back steps undamped)
    s2, e2 = paths.ar_generate(eng, surf, ex, int(g["horizon"]), 1, eps=eps, share_prefix=True)
    tol2 = 1e-6 if str(cfg.get("mem_type", "lstm")) in ("lstm", "gru") else 5e-6
    assert float((s2 - s).abs().max()) <= tol2
    if ex is not None:
        assert float((e2 - e).abs().max()) <= tol2


def test_autoregressive_shared_prefix_equals_literal_rollout():
    """S > 1 (Philox noise): every row re-encoding its whole window against the per-day shared prefix."""
    from vsv import paths
    # (the last case has > 8192 rows: the four-waves-per-tile encoder schedule reading the per-day tables,
    # and steps k >= C whose window is all generated days)
    for mem, L, X, D, S, C, H in (("lstm", 12, 3, 3, 40, 6, 9), ("gru", 5, 0, 2, 33, 4, 3), ("lstm", 5, 3, 3, 3000, 4, 6)):
        cfg = vo.make_config(L, X, mem_type=mem)
        eng, _ = _engine(cfg, 41, 2.0)
        rng = np.random.default_rng(19)
        surf = _t((0.2 + 0.1 * rng.standard_normal((D, C, 5, 5))).astype(np.float32))
        ex = _t((0.01 * rng.standard_normal((D, C, X))).astype(np.float32)) if X else None
        a_s, a_e = paths.ar_generate(eng, surf, ex, H, S, seed=77, day0=5, share_prefix=False)
        b_s, b_e = paths.ar_generate(eng, surf, ex, H, S, seed=77, day0=5, share_prefix=True)
        assert float((a_s - b_s).abs().max()) <= 1e-6
        if X:
            assert float((a_e - b_e).abs().max()) <= 1e-6


@pytest.mark.parametrize("name", golden_names("g7_"))
def test_autoregressive_offset_vs_reference(name):
    from vsv import paths
    g = load_golden(name)
    cfg = cfg_of(g)
    eng, _ = _engine(cfg, int(g["seed"]), float(g["scale"]))
    surf, ex = _t(g["surface"]), _t(g.get("ex_feats"))
    C, H = g["surface"].shape[1], int(g["horizon"])
    zs = g["z_steps"]                                                            # (steps, B, C+H, L)
    eps = _t(np.concatenate([zs[k][:, C:] for k in range(zs.shape[0])], axis=1))
    s, e = paths.ar_offset_generate(eng, surf, ex, zs.shape[0], H, eps=eps)
    assert s.shape == g["surf_out"].shape
    assert err(_n(s), g["surf_out"]) <= TOL
    if ex is not None:
        assert err(_n(e), g["ex_out"]) <= TOL


def test_ar_mc_windows_vs_oracle():
    """AR MC backfill with on-device reduction: Philox noise materialised and fed to the oracle."""
    from vsv import paths
    L, X, C, H, S, N = 5, 0, 6, 3, 40, 3
    cfg = vo.make_config(L, X)
    eng, W = _engine(cfg, 31, 2.0)
    rng = np.random.default_rng(3)
    vol = (0.2 + 0.1 * rng.standard_normal((N + C, 5, 5))).astype(np.float32)
    q = paths.ar_mc_windows(eng, _t(vol), None, C, N, C, H, S, seed=9, chunk_days=2, channel=None)
    assert q.shape == (N, H, 3, 3, 5, 5)
    W64 = {k: v.astype(np.float64) for k, v in W.items()}
    want = np.zeros((N, S, H, 3, 5, 5))
    for d in range(N):
        noise = _n(eng.philox_normal(9, C + d, 1, S, 0, H))                      # (S, H, L)
        zs = np.zeros((H, S, C + 1, L))
        zs[:, :, C] = noise.transpose(1, 0, 2)
        win = np.broadcast_to(vol[d:d + C][None].astype(np.float64), (S, C, 5, 5)).copy()
        want[d], _ = vo.generate_autoregressive_sequence(W64, cfg, win, None, zs)
    wq = np.percentile(want, [5, 50, 95], axis=1).transpose(1, 2, 0, 3, 4, 5)
    assert err(_n(q), wq) <= TOL


def test_oracle_mc_windows_vs_oracle():
    """Posterior-sampled (oracle) MC: z ~ q(z | context + target) at every position."""
    from vsv import paths
    L, X, C, H, S, N = 5, 3, 5, 3, 48, 5
    cfg = vo.make_config(L, X)
    eng, W = _engine(cfg, 13, 2.0)
    rng = np.random.default_rng(4)
    vol = (0.2 + 0.1 * rng.standard_normal((N + C + H, 5, 5))).astype(np.float32)
    ex = (0.01 * rng.standard_normal((N + C + H, X))).astype(np.float32)
    q = paths.oracle_mc_windows(eng, _t(vol), _t(ex), C, N, C, H, S, seed=3, chunk_days=2, channel=None)
    assert q.shape == (N, H, 3, 3, 5, 5)
    W64 = {k: v.astype(np.float64) for k, v in W.items()}
    want = np.zeros((N, S, H, 3, 5, 5))
    for d in range(N):
        eps = _n(eng.philox_normal(3, C + d, 1, S, 0, C + H))                 # (S, T, L)
        win = np.broadcast_to(vol[d:d + C + H][None].astype(np.float64), (S, C + H, 5, 5))
        wex = np.broadcast_to(ex[d:d + C + H][None].astype(np.float64), (S, C + H, X))
        want[d] = vo.forward(W64, cfg, win, wex, eps, H)[0]
    wq = np.percentile(want, [5, 50, 95], axis=1).transpose(1, 2, 0, 3, 4, 5)
    assert err(_n(q), wq) <= TOL


# ---- training: losses and gradients against the reference's autograd (goldens g4 / g5) ----

def _train_setup(g):
    from vsv.engine import Engine
    from vsv.train import TrainEngine
    cfg = cfg_of(g)
    W = vo.init_weights(cfg, int(g["seed"]), float(g["scale"]), np.float32)
    eng = Engine(cfg, W, "cuda:0")
    te = TrainEngine(eng, vo.state_dict_spec(cfg))
    return cfg, eng, te


def _loss_cfg(cfg, g):
    X = cfg["ex_feats_dim"]
    return dict(quantiles=cfg["quantiles"], quantile_weights=[float(x) for x in g["cfg_quantile_loss_weights"]],
                ex_n=(1 if cfg["ex_loss_on_ret_only"] else max(X, 1)), ex_l2=int(cfg["ex_feats_loss_type"] == "l2"),
                ex_weight=cfg["re_feat_weight"], kl_weight=cfg["kl_weight"])


def _check_grads(g, names, grads, rtol=1e-3):
    total = 0.0
    for n, gr in zip(names, grads):
        gr = _n(gr)
        total += float((gr ** 2).sum())
        want_norm = float(g["gnorm/" + n])
        scale = max(want_norm / np.sqrt(gr.size), 1e-12)            # RMS magnitude of the tensor's gradient
        if "grad/" + n in g:
            want = g["grad/" + n].astype(np.float64)
            assert np.max(np.abs(gr - want)) <= rtol * max(np.max(np.abs(want)), scale) + 1e-9, n
        else:
            idx = g["gidx/" + n]
            want = g["gval/" + n]
            assert np.max(np.abs(gr.ravel()[idx] - want)) <= rtol * max(np.max(np.abs(want)), 10 * scale) + 1e-9, n
        assert abs(np.sqrt((gr ** 2).sum()) - want_norm) <= rtol * want_norm + 1e-9, n
    assert abs(np.sqrt(total) - float(g["grad_total_norm"])) <= rtol * float(g["grad_total_norm"])


@pytest.mark.parametrize("name", golden_names("g4_"))
def test_train_losses_and_gradients_vs_reference(name):
    g = load_golden(name)
    cfg, eng, te = _train_setup(g)
    # goldens that pass extra loss settings store them in the file name's config: rebuild from the golden keys
    for k in ("ex_loss_on_ret_only", "ex_feats_loss_type", "kl_weight", "re_feat_weight"):
        if "cfg_" + k in g:
            v = g["cfg_" + k]
            cfg[k] = v.item() if v.dtype.kind != "U" else str(v)
    losses, gp, _ = te.loss_and_grads(_t(g["surface"]), _t(g.get("ex_feats")), int(g["horizon"]), _loss_cfg(cfg, g),
                                      eps=_t(g["eps"]))
    for k in ("loss", "re_surface", "re_ex_feats", "reconstruction_loss", "kl_loss"):
        want = float(g["loss_" + k])
        assert abs(float(losses[k]) - want) <= 1e-3 * abs(want) + 1e-9, (k, float(losses[k]), want)
    grads = te.param_grads(gp)
    _check_grads(g, [n for n, _ in vo.state_dict_spec(cfg)], grads)


def test_multihorizon_losses_and_gradients_vs_reference():
    """train_step_multihorizon's accumulated loss and gradient (golden g5; weights {1: 1.0, 7: 0.8, 14: 0.6,
    30: 0.4}, each horizon's forward with its own injected noise): (a) one forward/backward per horizon,
    summed; (b) the horizons as row groups of ONE forward/backward (loss_and_grads_multihorizon);
    (c) through the drop-in class, model.train_step_multihorizon with the private noise hook."""
    from vae.cvae_with_mem_randomized import CVAEMemRand
    g = load_golden("g5_multih_ex5_B4_T50")
    cfg, eng, te = _train_setup(g)
    for k in ("ex_loss_on_ret_only", "ex_feats_loss_type", "kl_weight", "re_feat_weight"):
        v = g["cfg_" + k]
        cfg[k] = v.item() if v.dtype.kind != "U" else str(v)
    wts = {1: 1.0, 7: 0.8, 14: 0.6, 30: 0.4}
    hz = g["horizons"].tolist()
    names = [n for n, _ in vo.state_dict_spec(cfg)]
    total, gp_sum, hl = 0.0, None, []
    for i, h in enumerate(hz):
        losses, gp, _ = te.loss_and_grads(_t(g["surface"]), _t(g["ex_feats"]), h, _loss_cfg(cfg, g), eps=_t(g["eps"][i]))
        hl.append(float(losses["loss"]))
        total += wts[h] * float(losses["loss"])
        gp_sum = wts[h] * gp if gp_sum is None else gp_sum + wts[h] * gp
    assert np.max(np.abs(np.asarray(hl) - g["horizon_losses"]) / np.abs(g["horizon_losses"])) <= 1e-3
    assert abs(total - float(g["loss"])) <= 1e-3 * abs(float(g["loss"]))
    _check_grads(g, names, te.param_grads(gp_sum))
    # (b)
    losses, gp = te.loss_and_grads_multihorizon(_t(g["surface"]), _t(g["ex_feats"]), hz, [wts[h] for h in hz],
                                                _loss_cfg(cfg, g), eps=[_t(e) for e in g["eps"]])
    assert np.max(np.abs(_n(losses["horizon_losses"]) - g["horizon_losses"]) / np.abs(g["horizon_losses"])) <= 1e-3
    assert abs(float(losses["loss"]) - float(g["loss"])) <= 1e-3 * abs(float(g["loss"]))
    assert abs(float(losses["loss"]) - total) <= 1e-5 * abs(total)
    _check_grads(g, names, te.param_grads(gp))
    # (c)
    cfg.update(device="cuda:0")
    model = CVAEMemRand(dict(cfg))
    W = vo.init_weights(cfg, int(g["seed"]), float(g["scale"]), np.float32)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=False)
    model.eval()                                                   # (no dropout: the golden's reference ran without)
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    out = model.train_step_multihorizon({"surface": _t(g["surface"]), "ex_feats": _t(g["ex_feats"])}, opt, horizons=hz,
                                        _eps_per_horizon=[_t(e) for e in g["eps"]])
    assert isinstance(out["loss"], float) and abs(out["loss"] - float(g["loss"])) <= 1e-3 * abs(float(g["loss"]))
    assert list(out["horizon_losses"]) == hz
    assert np.max(np.abs(np.asarray([out["horizon_losses"][h] for h in hz]) - g["horizon_losses"])
                  / np.abs(g["horizon_losses"])) <= 1e-3
    for k in ("reconstruction_loss", "kl_loss"):
        if k in g:
            assert abs(out[k] - float(g[k])) <= 1e-3 * abs(float(g[k])) + 1e-9, k
    _check_grads(g, [n for n, _ in model.named_parameters()], [p.grad for _, p in model.named_parameters()])


@pytest.mark.parametrize("L,X,B,T,h,p,mem", [(5, 3, 37, 12, 3, 0.25, "lstm"), (12, 3, 8, 30, 7, 0.3, "lstm"),
                                             (5, 0, 64, 6, 1, 0.2, "lstm"), (5, 3, 20, 14, 4, 0.3, "gru"),
                                             (5, 3, 20, 14, 4, 0.3, "rnn")])
def test_train_with_lstm_dropout_vs_autograd_oracle(L, X, B, T, h, p, mem):
    """nn.LSTM's inter-layer dropout (train mode, mem_dropout 0.2-0.3 in every production config).  The
    reference draws its masks from torch's generator inside nn.LSTM, so they cannot be reproduced: the
    HIP path's Philox masks are (a) checked against their numpy restatement and for their keep rate,
    (b) injected into the torch-port autograd oracle (pinned to the reference by goldens g4), whose
    loss and parameter gradients the HIP forward + BPTT must match."""
    from oracle.torch_port import TorchPort
    from vsv.engine import Engine
    from vsv.train import TrainEngine
    cfg = vo.make_config(L, X, mem_type=mem)
    cfg.update(ex_loss_on_ret_only=True, ex_feats_loss_type="l2", kl_weight=1e-3, re_feat_weight=1.0)
    W = vo.init_weights(cfg, 21, 1.5, np.float32)
    eng = Engine(cfg, W, "cuda:0")
    spec = vo.state_dict_spec(cfg)
    te = TrainEngine(eng, spec)
    rng = np.random.default_rng(8)
    surf = (0.2 + 0.1 * rng.standard_normal((B, T, 5, 5))).astype(np.float32)
    ex = (0.01 * rng.standard_normal((B, T, X))).astype(np.float32) if X else None
    eps = rng.standard_normal((B, T, L)).astype(np.float32)
    seed = 0x1234_5678_9ABC
    Hm, Hd = cfg["mem_hidden"], 125 + X
    masks = {}
    for name, lid, hid, Tw in (("encoder", 0, Hm, T), ("ctx_encoder", 1, Hm, T - h), ("decoder", 2, Hd, T)):
        m = _n(eng.dropout_mask(p, seed, lid, 0, B, Tw, hid))
        assert np.array_equal(m, vo.dropout_mask(p, seed, lid, 0, B, Tw, hid)), name
        keep = float((m > 0).mean())
        assert abs(keep - (1 - p)) <= 4 * np.sqrt(p * (1 - p) / m.size) + 1e-3, (name, keep)
        assert np.allclose(m[m > 0], 1.0 / (1.0 - p))
        masks[name] = torch.from_numpy(m.astype(np.float32))
    assert not np.array_equal(_n(eng.dropout_mask(p, seed, 0, 0, B, T, Hm)), _n(eng.dropout_mask(p, seed + 1, 0, 0, B, T, Hm)))
    lc = dict(quantiles=cfg["quantiles"], quantile_weights=[5.0, 1.0, 5.0], ex_n=1, ex_l2=1, ex_weight=1.0, kl_weight=1e-3)
    losses, gp, _ = te.loss_and_grads(_t(surf), _t(ex), h, lc, eps=_t(eps), dropout_p=p, dropout_seed=seed)
    port = TorchPort(cfg, W, torch.float32)
    port.drop_masks = masks
    want = port.losses(torch.from_numpy(surf), None if ex is None else torch.from_numpy(ex), torch.from_numpy(eps), h,
                       quantile_weights=[5.0, 1.0, 5.0])
    want["loss"].backward()
    for k in ("loss", "re_surface", "reconstruction_loss", "kl_loss"):
        assert abs(float(losses[k]) - float(want[k])) <= 1e-3 * abs(float(want[k])) + 1e-9, k
    ref = port.named_reference_parameters()
    tot_g = tot_w = 0.0
    for (n, _), gr in zip(spec, te.param_grads(gp)):
        if n.startswith("quantile_loss_fn"):
            continue
        wg = ref[n].grad
        wg = np.zeros(ref[n].shape, dtype=np.float32) if wg is None else wg.numpy()
        gr = _n(gr)
        nrm = float(np.sqrt((wg.astype(np.float64) ** 2).sum()))
        assert float(np.sqrt(((gr - wg).astype(np.float64) ** 2).sum())) <= 1e-3 * nrm + 1e-7, n
        tot_g += float((gr.astype(np.float64) ** 2).sum())
        tot_w += nrm ** 2
    assert abs(np.sqrt(tot_g) - np.sqrt(tot_w)) <= 1e-3 * np.sqrt(tot_w)
    # the mask matters (the same step without dropout gives another loss) and p = 0 is the old path
    l0, _, _ = te.loss_and_grads(_t(surf), _t(ex), h, lc, eps=_t(eps), want_grads=False)
    assert abs(float(l0["loss"]) - float(losses["loss"])) > 1e-7


@pytest.mark.parametrize("name", golden_names("g9_artrain"))
def test_autoregressive_training_step_vs_reference(name):
    """vae.utils.train_autoregressive_step (reference :426-623): losses and every parameter's gradient,
    which flow through the fed-back p50 predictions into the earlier steps (goldens g9)."""
    from vae.cvae_with_mem_randomized import CVAEMemRand
    from vae.utils import train_autoregressive_step
    g = load_golden(name)
    cfg = cfg_of(g)
    for k in ("ex_loss_on_ret_only", "ex_feats_loss_type", "kl_weight", "re_feat_weight"):
        v = g["cfg_" + k]
        cfg[k] = v.item() if v.dtype.kind != "U" else str(v)
    cfg.update(device="cuda:0", context_len=int(g["context_len"]))
    model = CVAEMemRand(dict(cfg))
    W = vo.init_weights(cfg, int(g["seed"]), float(g["scale"]), np.float32)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=False)
    model.train()
    batch = {"surface": _t(g["surface"])}
    if "ex_feats" in g:
        batch["ex_feats"] = _t(g["ex_feats"])
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    K = int(g["ar_steps"])
    out = train_autoregressive_step(model, batch, opt, K, int(g["horizon"]), int(g["offset"]),
                                    _eps_steps=[_t(g["eps"][k]) for k in range(K)])
    assert out["ar_steps"] == K and out["offset"] == int(g["offset"])
    for k in ("loss", "reconstruction_loss", "kl_loss"):
        assert isinstance(out[k], float) and abs(out[k] - float(g[k])) <= 1e-3 * abs(float(g[k])) + 1e-9, k
    names = [n for n, _ in model.named_parameters()]
    _check_grads(g, names, [p.grad for _, p in model.named_parameters()])
    # too short a sequence is the reference's ValueError
    with pytest.raises(ValueError):
        train_autoregressive_step(model, {"surface": batch["surface"][:, :5]}, opt, K, int(g["horizon"]), int(g["offset"]))


def test_fitted_gmm_prior_generation(tmp_path):
    """prior_mode="fitted" (reference :583-640, :689-690): future latents from the GMM fitted to the
    aggregated posterior (file format of analysis_code/fit_aggregated_posterior.py:139-155)."""
    from vae.cvae_with_mem_randomized import CVAEMemRand
    L, X, B, C, H = 5, 3, 6, 8, 4
    cfg = vo.make_config(L, X, device="cuda:0")
    W = vo.init_weights(cfg, 9, 2.0, np.float32)
    model = CVAEMemRand(dict(cfg))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=False)
    model.eval()
    rng = np.random.default_rng(2)
    c = {"surface": _t(0.2 + 0.1 * rng.standard_normal((B, C, 5, 5))), "ex_feats": _t(0.01 * rng.standard_normal((B, C, X)))}
    with pytest.raises(ValueError):
        model.get_surface_given_conditions(c, horizon=H, prior_mode="fitted")        # not loaded yet
    # one (near-)degenerate component: every future latent equals its mean
    mean = rng.standard_normal(L)
    path = str(tmp_path / "gmm.pt")
    torch.save({"means": torch.tensor(mean[None]), "covariances": (1e-16 * torch.eye(L, dtype=torch.float64))[None],
                "weights": torch.tensor([1.0], dtype=torch.float64), "n_components": 1}, path)
    model.load_fitted_prior(path)
    s, e = model.get_surface_given_conditions(c, horizon=H, prior_mode="fitted")
    z = torch.zeros((B, C + H, L), dtype=torch.float32)
    z[:, C:] = torch.tensor(mean, dtype=torch.float32)
    s2, e2 = model.get_surface_given_conditions(c, z=z.cuda(), horizon=H)
    assert s.shape == (B, H, 3, 5, 5) and err(_n(s), _n(s2)) <= 1e-5 and err(_n(e), _n(e2)) <= 1e-5
    # two well-separated components: the mixture weights and means show in the samples
    torch.save({"means": torch.tensor([[-3.0] * L, [3.0] * L], dtype=torch.float64),
                "covariances": torch.stack([0.01 * torch.eye(L, dtype=torch.float64)] * 2),
                "weights": torch.tensor([0.25, 0.75], dtype=torch.float64), "n_components": 2}, path)
    model.load_fitted_prior(path)
    zs = model._sample_from_fitted_prior(500, 8)
    assert zs.shape == (500, 8, L) and zs.dtype == torch.float64
    frac_hi = float((zs[..., 0] > 0).double().mean())
    assert abs(frac_hi - 0.75) < 0.03 and abs(float(zs[zs[..., 0] > 0].mean()) - 3.0) < 0.02


def test_model_class_training_api():
    """The drop-in class end to end on the GPU: the shapes the reference's own script test checks
    (test_code/test_quantile_decoder.py:34-50, 105-178), train_step / test_step dict contract,
    clipping, multi-horizon return types, and that optimisation actually reduces the loss."""
    from vae.cvae_with_mem_randomized import CVAEMemRand
    torch.manual_seed(0)
    cfg = vo.make_config(5, 3, device="cuda:0", kl_weight=1e-5)
    model = CVAEMemRand(cfg)
    rng = np.random.default_rng(0)
    x = {"surface": torch.from_numpy((0.2 + 0.1 * rng.standard_normal((16, 8, 5, 5))).astype(np.float32)),
         "ex_feats": torch.from_numpy((0.01 * rng.standard_normal((16, 8, 3))).astype(np.float32))}
    out = model.forward(x)
    assert out[0].shape == (16, 1, 3, 5, 5) and out[1].shape == (16, 1, 3) and out[2].shape == (16, 8, 5)
    s, e = model.get_surface_given_conditions({"surface": x["surface"][0, :7], "ex_feats": x["ex_feats"][0, :7]})
    assert s.shape == (1, 1, 3, 5, 5) and e.shape == (1, 1, 3)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    before = model.test_step(x)
    assert set(before) == {"loss", "re_surface", "re_ex_feats", "reconstruction_loss", "kl_loss"}
    first = None
    for _ in range(30):
        losses = model.train_step(x, opt)
        assert all(torch.is_tensor(v) for v in losses.values()) and torch.isfinite(losses["loss"])
        gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters()))
        assert gn <= 1.0 + 1e-4                                     # clip_grad_norm_(1.0)
        first = first if first is not None else float(losses["loss"])
    assert float(losses["loss"]) < first
    mh = model.train_step_multihorizon(x, opt, horizons=[1, 3])
    assert isinstance(mh["loss"], float) and set(mh["horizon_losses"]) == {1, 3}
    with pytest.raises(ValueError):
        model.train_step_multihorizon(x, opt, horizons=[1, 30])
    ar = model.generate_autoregressive_sequence({"surface": x["surface"][:2, :7], "ex_feats": x["ex_feats"][:2, :7]}, horizon=4)
    assert ar[0].shape == (2, 4, 3, 5, 5) and ar[1].shape == (2, 4, 3)
    # no-ex model: 4-tuple forward, zeros(1) placeholder
    m2 = CVAEMemRand(vo.make_config(5, 0, device="cuda:0"))
    l2 = m2.train_step({"surface": x["surface"]}, torch.optim.AdamW(m2.parameters(), lr=1e-4))
    assert l2["re_ex_feats"].shape == (1,) and len(m2.forward({"surface": x["surface"]})) == 4


@pytest.mark.parametrize("L,X,N,C,mem", [(12, 3, 70, 9, "lstm"), (5, 0, 33, 4, "lstm"), (5, 3, 101, 20, "gru")])
def test_windows_encoder_equals_strided_encoder(L, X, N, C, mem):
    """vsv_encode_windows (embedding and first-layer input product once per calendar day) against
    vsv_encode reading the same overlapping windows in place: the same products in the same order."""
    cfg = vo.make_config(L, X, mem_type=mem)
    eng, _ = _engine(cfg, 31, 2.0)
    rng = np.random.default_rng(12)
    vol = _t((0.2 + 0.1 * rng.standard_normal((N + C - 1, 5, 5))).astype(np.float32))
    ex = _t((0.01 * rng.standard_normal((N + C - 1, X))).astype(np.float32)) if X else None
    for which in (0, 1):
        a = eng.encode_windows(which, vol, ex, N, C)
        b = eng.encode_strided(which, vol, 25, 25, ex, X, X, N, C)
        assert a.shape == b.shape and float((a - b).abs().max()) <= 1e-6
    e2, c2 = eng.encode_both_windows(vol, ex, N - 3, C, C - 2)
    assert e2.shape == (N - 3, C, 2 * L) and c2.shape == (N - 3, C - 2, L)
    assert float((c2 - eng.encode_strided(1, vol, 25, 25, ex, X, X, N - 3, C - 2)).abs().max()) <= 1e-6


def test_windows_encoder_edges():
    """One-step windows, a window count that is not a multiple of the 32-row tile, zero windows, and the
    fall-back to the strided entry point beyond Engine.MAX_WINDOWS."""
    import ctypes as C
    cfg = vo.make_config(5, 3)
    eng, _ = _engine(cfg, 32, 2.0)
    rng = np.random.default_rng(13)
    n_big = eng.MAX_WINDOWS + 5
    vol = _t((0.2 + 0.1 * rng.standard_normal((n_big + 2, 5, 5))).astype(np.float32))
    ex = _t((0.01 * rng.standard_normal((n_big + 2, 3))).astype(np.float32))
    a = eng.encode_windows(0, vol, ex, 45, 1)
    assert float((a - eng.encode_strided(0, vol, 25, 25, ex, 3, 3, 45, 1)).abs().max()) <= 1e-6
    assert eng.encode_windows(1, vol, ex, 0, 4).shape == (0, 4, 5)
    ws = eng._ws(1 << 20)
    rc = eng.lib.vsv_encode_windows(C.byref(eng.dims), eng.packed.data_ptr(), 0, vol.data_ptr(), ex.data_ptr(),
                                    n_big, 2, ws.data_ptr(), ws.data_ptr(), ws.numel(), None)
    assert rc == -5 and b"8192" in eng.lib.vsv_last_error()
    e1, c1 = eng.encode_both_windows(vol, ex, n_big, 3, 2)               # > MAX_WINDOWS: strided path
    e2, c2 = eng.encode_both_windows(vol, ex, 64, 3, 2)
    assert torch.allclose(e1[:64], e2, atol=1e-6) and torch.allclose(c1[:64], c2, atol=1e-6)


def test_fused_two_layer_training_launches_still_match_reference():
    """Two-layer stacks train through one launch per layer by default; the single two-layer launches
    (kept for > 8192 rows and behind VSV_FWD_FUSED / VSV_BWD_FUSED) must give the same gradients.  The
    switches are read once per process, so the golden-gradient tests are re-run in a child process."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, VSV_FWD_FUSED="1", VSV_BWD_FUSED="1")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_parity.py"), "-q", "-x", "-m", "gpu",
                        "-k", "test_train_losses_and_gradients_vs_reference or test_train_with_lstm_dropout",
                        "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
    assert " passed" in r.stdout


def test_segmented_context_prefix_equals_single_launches():
    """The date-range prefix cut into time segments (encoders' segment i+1 beside the decoder's segment i,
    memory states handed on) against one launch per stack: the same steps in the same order."""
    from vsv import paths
    for mem, L, X, N, C in (("lstm", 12, 3, 70, 13), ("gru", 5, 0, 40, 9)):
        cfg = vo.make_config(L, X, mem_type=mem)
        eng, _ = _engine(cfg, 52, 2.0)
        rng = np.random.default_rng(23)
        vol = _t((0.2 + 0.1 * rng.standard_normal((N + C - 1, 5, 5))).astype(np.float32))
        ex = _t((0.01 * rng.standard_normal((N + C - 1, X))).astype(np.float32)) if X else None
        s1, g1 = paths._context_prefix(eng, vol, ex, N, C, n_seg=1)
        for layers in ("0", "py", "1"):
            # "py" / "1": every stack also split into its two layers (vsv_*_layer0 / _layer1 stages driven from
            # Python, resp. all issued by the one native call vsv_context_prefix)
            os.environ["VSV_PREFIX_LAYERS"] = layers
            try:
                for n_seg in (2, 3):
                    s2, g2 = paths._context_prefix(eng, vol, ex, N, C, n_seg=n_seg)
                    torch.cuda.synchronize()
                    assert torch.equal(s1, s2) and torch.equal(g1, g2), (mem, n_seg, layers)
            finally:
                os.environ.pop("VSV_PREFIX_LAYERS", None)


@pytest.mark.parametrize("compress,L,X,C", [(True, 12, 3, 7), (False, 12, 3, 7), (True, 6, 0, 7), (True, 5, 0, 7), (False, 7, 0, 7),
                                              (True, 12, 3, 1), (True, 12, 3, 2), (False, 12, 0, 3)])
def test_prefix_hand_over_in_tile_b_equals_layer_stages(compress, L, X, C):
    """Beyond 2048 days the native prefix lets the decoder read the encoders' outputs where their 16-row recurrent
    launches leave them (the posterior head -- with compress_context the context head too -- as the spare gate tile of
    lstm_rec16_kernel<13>, no head launches, no input assembly): bit for bit the layer stages issued one by one through
    the public entry points, ragged last row tile included, unsegmented and in two time segments, down to one and two
    context steps (the two-phase step's first step and its tail alone)."""
    from vsv import paths
    N = 2085
    cfg = vo.make_config(L, X, compress_context=compress)
    eng, _ = _engine(cfg, 31, 2.0)
    rng = np.random.default_rng(5)
    vol = _t((0.2 + 0.1 * rng.standard_normal((N + C - 1, 5, 5))).astype(np.float32))
    ex = _t((0.01 * rng.standard_normal((N + C - 1, X))).astype(np.float32)) if X else None
    os.environ["VSV_PREFIX_LAYERS"] = "py"
    try:
        s1, g1 = paths._context_prefix(eng, vol, ex, N, C, n_seg=1)
    finally:
        os.environ.pop("VSV_PREFIX_LAYERS", None)
    for n_seg in (1, 2):
        if n_seg > C:
            continue
        s2, g2 = paths._context_prefix(eng, vol, ex, N, C, n_seg=n_seg)
        torch.cuda.synchronize()
        assert torch.isfinite(s2).all() and torch.isfinite(g2).all()
        assert torch.equal(s1, s2) and torch.equal(g1, g2), (compress, L, C, n_seg)


@pytest.mark.parametrize("Hm,sh,X,mem", [(64, (3, 3, 3), 0, "lstm"), (48, (5, 4, 4), 3, "lstm"), (80, (4, 4), 0, "gru"),
                                          (100, (5, 5, 3), 0, "lstm")])
def test_hidden_widths_between_built_tile_counts(Hm, sh, X, mem):
    """the recurrent kernels are built for 13 and 16 gate tiles; any other hidden width runs
    zero-padded on the next built size (vsv_pack.cpp make_layout).  Forward, prior generation and the
    training gradients of such models against the oracles."""
    from oracle.torch_port import TorchPort
    from vsv import paths
    from vsv.engine import Engine
    from vsv.train import TrainEngine
    L, B, T, h = 5, 9, 7, 2
    cfg = vo.make_config(L, X, mem_hidden=Hm, surface_hidden=sh, mem_type=mem)
    cfg.update(ex_loss_on_ret_only=True, ex_feats_loss_type="l2", kl_weight=1e-3, re_feat_weight=1.0)
    W = vo.init_weights(cfg, 5, 2.0, np.float32)
    eng = Engine(cfg, W, "cuda:0")
    rng = np.random.default_rng(6)
    surf = (0.2 + 0.1 * rng.standard_normal((B, T, 5, 5))).astype(np.float32)
    ex = (0.01 * rng.standard_normal((B, T, X))).astype(np.float32) if X else None
    eps = rng.standard_normal((B, T, L)).astype(np.float32)
    W64 = {k: v.astype(np.float64) for k, v in W.items()}
    want = vo.forward(W64, cfg, surf.astype(np.float64), None if ex is None else ex.astype(np.float64),
                      eps.astype(np.float64), h)
    s, e, zm, zl, z = paths.forward(eng, _t(surf), _t(ex), h, eps=_t(eps))
    assert err(_n(s), want[0]) <= TOL
    if X:
        assert err(_n(e), want[1]) <= TOL
    # shared-prefix MC generation (fused decode kernel) against the literal oracle
    C, S = T - h, 40
    e2 = rng.standard_normal((B, S, h, L)).astype(np.float32)
    ws, we = vo.prior_mc_shared_prefix(W64, cfg, surf[:, :C].astype(np.float64),
                                       None if ex is None else ex[:, :C].astype(np.float64), e2.astype(np.float64))
    gs, ge, _ = paths.prior_generate(eng, _t(surf[:, :C]), _t(None if ex is None else ex[:, :C]), h, S,
                                     eps=_t(e2.reshape(B * S, h, L)))
    assert err(_n(gs).reshape(ws.shape), ws) <= TOL
    # training gradients against the autograd port
    spec = vo.state_dict_spec(cfg)
    te = TrainEngine(eng, spec)
    lc = dict(quantiles=cfg["quantiles"], quantile_weights=[1.0, 1.0, 1.0], ex_n=1, ex_l2=1, ex_weight=1.0, kl_weight=1e-3)
    losses, gp, _ = te.loss_and_grads(_t(surf), _t(ex), h, lc, eps=_t(eps))
    port = TorchPort(cfg, W, torch.float32)
    wl = port.losses(torch.from_numpy(surf), None if ex is None else torch.from_numpy(ex), torch.from_numpy(eps), h)
    wl["loss"].backward()
    assert abs(float(losses["loss"]) - float(wl["loss"])) <= 1e-3 * abs(float(wl["loss"])) + 1e-9
    ref = port.named_reference_parameters()
    for (n, _), gr in zip(spec, te.param_grads(gp)):
        wg = ref[n].grad
        wg = np.zeros(ref[n].shape, dtype=np.float32) if wg is None else wg.numpy()
        nrm = float(np.sqrt((wg.astype(np.float64) ** 2).sum()))
        assert float(np.sqrt(((_n(gr) - wg).astype(np.float64) ** 2).sum())) <= 1e-3 * nrm + 1e-7, n


@pytest.mark.parametrize("name", golden_names("g13_"))
def test_autoregressive_sequence_with_model_horizon_above_one(name):
    """the reference's generate_autoregressive_sequence generates self.horizon days per step (:776)
    and returns them all, (B, horizon * self.horizon, ...), feeding back the first day only."""
    from vae.cvae_with_mem_randomized import CVAEMemRand
    from vsv import paths
    g = load_golden(name)
    cfg = cfg_of(g)
    eng, W = _engine(cfg, int(g["seed"]), float(g["scale"]))
    surf, ex = _t(g["surface"]), _t(g.get("ex_feats"))
    C, Hm, H = g["surface"].shape[1], int(g["model_horizon"]), int(g["horizon"])
    eps = _t(np.ascontiguousarray(g["z_steps"][:, :, C:].transpose(1, 0, 2, 3)))      # (B, H, Hm, L)
    s, e = paths.ar_generate_multi(eng, surf, ex, H, Hm, eps=eps)
    assert s.shape == g["surf_out"].shape and err(_n(s), g["surf_out"]) <= TOL
    if ex is not None:
        assert err(_n(e), g["ex_out"]) <= TOL
    model = CVAEMemRand(dict(cfg, device="cuda:0", horizon=Hm))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=False)
    c = {"surface": surf} if ex is None else {"surface": surf, "ex_feats": ex}
    out = model.generate_autoregressive_sequence(c, horizon=H)
    out_s = out[0] if ex is not None else out
    assert out_s.shape == g["surf_out"].shape and bool(torch.isfinite(out_s).all())


def test_in_place_data_edits_need_invalidate():
    """p.data edits bump no version counter; invalidate_packed_weights() makes them visible."""
    from vae.cvae_with_mem_randomized import CVAEMemRand
    cfg = vo.make_config(5, 0, device="cuda:0")
    model = CVAEMemRand(dict(cfg))
    model.eval()
    rng = np.random.default_rng(0)
    c = {"surface": _t(0.2 + 0.1 * rng.standard_normal((3, 6, 5, 5)))}
    z = torch.zeros((3, 7, 5), device="cuda:0")
    a = model.get_surface_given_conditions(c, z=z.clone())
    with torch.no_grad():
        model.decoder.surface_decoder.dec_output.bias.data.add_(1.0)       # no version bump
    model.invalidate_packed_weights()
    b = model.get_surface_given_conditions(c, z=z.clone())
    assert float((b - a - 1.0).abs().max()) <= 1e-5
    with torch.no_grad():
        model.decoder.surface_decoder.dec_output.bias.add_(1.0)            # a tracked in-place op: picked up by itself
    d = model.get_surface_given_conditions(c, z=z.clone())
    assert float((d - b - 1.0).abs().max()) <= 1e-5


@pytest.mark.parametrize("L,X,D,S,C,H,mem", [(12, 3, 9, 1000, 6, 1, "lstm"), (5, 0, 5, 333, 4, 1, "lstm"),
                                             (12, 3, 4, 250, 5, 3, "lstm"), (5, 3, 3, 96, 4, 2, "gru")])
def test_single_channel_output_equals_channel_of_full_output(L, X, D, S, C, H, mem):
    """vsv_decode with out_mode VSV_OUT_CHANNEL(c) (the output Conv restricted to one decoder channel) gives,
    bit for bit, channel c of the full sample-minor output -- on the one-wave Monte-Carlo kernel (H = 1, Philox,
    shared prefix) and on the generic schedule (H > 1, GRU); the reduced products agree likewise."""
    from vsv import paths
    cfg = vo.make_config(L, X, mem_type=mem)
    eng, _ = _engine(cfg, 31, 2.0)
    N = C + D + H
    rng = np.random.default_rng(8)
    vol = _t(0.2 + 0.1 * rng.standard_normal((N, 5, 5)))
    ex = _t(0.01 * rng.standard_normal((N, X))) if X else None
    full = paths.prior_mc_windows(eng, vol, ex, C, D, C, H, S, seed=5, channel=None, chunk_days=4)
    assert full.shape == (D, H, 3, 3, 5, 5)
    for c in range(3):
        one = paths.prior_mc_windows(eng, vol, ex, C, D, C, H, S, seed=5, channel=c, chunk_days=4)
        assert one.shape == (D, H, 3, 5, 5)
        assert torch.equal(one, full[:, :, :, c])
    # the raw samples
    state, gates = paths._context_prefix(eng, vol, ex, D, C)
    kw = dict(state_in=state, state_div=S, gate_in=gates, seed=5, S=S, day0=C)
    s_all = eng.decode(D * S, H, 0, out_mode=1, **kw)["surf"]
    for c in range(3):
        r = eng.decode(D * S, H, 0, out_mode=eng.OUT_CHANNEL(c), **kw)
        assert r["surf"].shape == (D, H, 25, S)
        assert torch.equal(r["surf"], s_all[:, :, 25 * c:25 * c + 25])
    with pytest.raises(ValueError):
        eng.decode(D * S, H, 0, out_mode=eng.OUT_CHANNEL(3), **kw)


@pytest.mark.parametrize("L,X,D,S,C,H,mem", [(5, 3, 3, 200, 4, 6, "lstm"), (12, 0, 2, 1500, 3, 3, "lstm"), (5, 3, 2, 64, 3, 4, "gru")])
def test_reducing_rollout_equals_percentiles_of_stored_rollouts(L, X, D, S, C, H, mem):
    """ar_generate(reduce_qs=...) -- every step decoded sample-minor (p50 channel alone when that is the reduced
    one), reduced on the side stream, samples never stored -- against np.percentile of the rollouts it replaces;
    the rollouts themselves are bit-identical (same feedback values), so the quantiles agree to rounding of the
    interpolation."""
    from vsv import paths
    cfg = vo.make_config(L, X, mem_type=mem)
    eng, _ = _engine(cfg, 13, 2.0)
    rng = np.random.default_rng(3)
    surf = _t(0.2 + 0.1 * rng.standard_normal((D, C, 5, 5)))
    ex = _t(0.01 * rng.standard_normal((D, C, X))) if X else None
    raw, _ = paths.ar_generate(eng, surf, ex, H, S, seed=4, day0=2)              # (D*S, H, 3, 5, 5)
    want = np.percentile(_n(raw).reshape(D, S, H, 3, 5, 5), [5, 50, 95], axis=1).transpose(1, 2, 0, 3, 4, 5)
    for ch in (1, None, 2):
        q, none = paths.ar_generate(eng, surf, ex, H, S, seed=4, day0=2, reduce_qs=(5, 50, 95), reduce_channel=ch)
        assert none is None
        w = want if ch is None else want[:, :, :, ch]
        assert q.shape == w.shape
        assert err(_n(q), w) <= 1e-6


@pytest.mark.parametrize("name", golden_names("g14_legacy"))
def test_legacy_one_channel_mse_model_vs_reference(name):
    """vae.cvae_with_mem.CVAEMem -- the legacy fixed-length model: one decoder channel, MSE surface loss, uniform
    reparameterisation noise, generation from z at every position -- against the reference's own class (goldens
    g14): forward, test_step losses, every parameter's gradient through train_step (SGD lr 0; the legacy step
    does not clip), get_surface_given_conditions."""
    from vae.cvae_with_mem import CVAEMem
    g = load_golden(name)
    X = int(g["cfg_ex_feats_dim"])
    T, C = int(g["seq_len"]), int(g["ctx_len"])
    cfg = vo.make_config(int(g["cfg_latent_dim"]), X, device="cuda:0", seq_len=T, ctx_len=C)
    for k in ("ex_feats_loss_type", "kl_weight", "re_feat_weight"):
        v = g["cfg_" + k]
        cfg[k] = v.item() if v.dtype.kind != "U" else str(v)
    model = CVAEMem(dict(cfg))
    W = vo.init_weights(dict(cfg, quantiles=[0.5]), int(g["seed"]), float(g["scale"]), np.float32)
    assert {k: tuple(v.shape) for k, v in model.state_dict().items()} == {k: v.shape for k, v in W.items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=True)
    x = {"surface": _t(g["surface"])}
    if X:
        x["ex_feats"] = _t(g["ex_feats"])
    eps = _t(g["eps"])
    model.eval()
    with torch.no_grad():
        out = model.forward(x, _eps=eps)
    assert out[0].shape == g["recon"].shape == (g["surface"].shape[0], T - C, 5, 5)
    assert err(_n(out[0]), g["recon"]) <= TOL
    assert err(_n(out[-3]), g["z_mean"]) <= TOL and err(_n(out[-1]), g["z"]) <= TOL
    if X:
        assert err(_n(out[1]), g["ex_recon"]) <= TOL
    losses = model.test_step(x, _eps=eps)
    for k in ("loss", "re_surface", "re_ex_feats", "reconstruction_loss", "kl_loss"):
        want = float(g["loss_" + k])
        assert abs(float(losses[k]) - want) <= 1e-3 * abs(want) + 1e-9, (k, float(losses[k]), want)
    model.train()
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    tl = model.train_step(x, opt, _eps=eps)
    assert abs(float(tl["loss"]) - float(g["loss_loss"])) <= 1e-3 * abs(float(g["loss_loss"]))
    _check_grads(g, [n for n, _ in model.named_parameters()], [p.grad for _, p in model.named_parameters()])
    # generation: the caller's z at every position
    c = {"surface": x["surface"][:, :C]}
    if X:
        c["ex_feats"] = x["ex_feats"][:, :C]
    model.eval()
    gen = model.get_surface_given_conditions(c, z=_t(g["z_given"]))
    s = gen[0] if X else gen
    assert s.shape == g["surf_out"].shape and err(_n(s), g["surf_out"]) <= TOL
    if X:
        assert err(_n(gen[1]), g["ex_out"]) <= TOL
    # and the prior draw runs, with the legacy shapes
    s2 = model.get_surface_given_conditions(c)
    assert (s2[0] if X else s2).shape == g["surf_out"].shape
    with pytest.raises(ValueError):
        model.forward({"surface": x["surface"][:, :T - 1]})


@pytest.mark.parametrize("mem,L,X", [("lstm", 5, 3), ("lstm", 12, 0), ("gru", 5, 3)])
def test_uncompressed_context_mc_product_and_model_class(mem, L, X):
    """compress_context=False (the class default, reference :1014-1015): the decoder reads the context memory's
    hidden state (width mem_hidden) next to z.  The goldens named *_nocomp_* cover forward / generation / gradients
    against the reference; here the Monte-Carlo product (shared prefix, fused decoder with Philox noise, percentile
    reduction, the segmented prefix) against the oracle, and the drop-in class (device-side packing, training)."""
    from vsv import paths
    from vae.cvae_with_mem_randomized import CVAEMemRand
    C, H, S, N = 6, 2, 1000, 40
    cfg = vo.make_config(L, X, mem_type=mem, compress_context=False)
    eng, W = _engine(cfg, 61, 2.0)
    assert eng.Lc == cfg["mem_hidden"]
    rng = np.random.default_rng(29)
    vol = (0.2 + 0.1 * rng.standard_normal((N + C, 5, 5))).astype(np.float32)
    ex = (0.01 * rng.standard_normal((N + C, X))).astype(np.float32) if X else None
    q = paths.prior_mc_windows(eng, _t(vol), _t(ex), C, N, C, H, S, seed=7, chunk_days=16, channel=None)
    eps = _n(eng.philox_normal(7, C, N, S, 0, H)).reshape(N, S, H, L)
    win = np.stack([vol[d:d + C] for d in range(N)])
    wex = np.stack([ex[d:d + C] for d in range(N)]) if X else None
    smp, _ = vo.prior_mc_shared_prefix(W, cfg, win, wex, eps.astype(np.float32))
    want = np.percentile(smp.astype(np.float64), [5, 50, 95], axis=1).transpose(1, 2, 0, 3, 4, 5)
    assert err(_n(q), want) <= TOL
    q1 = paths.prior_mc_windows(eng, _t(vol), _t(ex), C, N, C, H, S, seed=7, chunk_days=256, channel=1)
    assert err(_n(q1), want[:, :, :, 1]) <= TOL
    # the date-range prefix in time segments: same steps in the same order
    s1, g1 = paths._context_prefix(eng, _t(vol), _t(ex), N, C, n_seg=1)
    for layers in ("0", "1"):
        os.environ["VSV_PREFIX_LAYERS"] = layers
        try:
            s2, g2 = paths._context_prefix(eng, _t(vol), _t(ex), N, C, n_seg=2)
            torch.cuda.synchronize()
            assert torch.equal(s1, s2) and torch.equal(g1, g2), layers
        finally:
            os.environ.pop("VSV_PREFIX_LAYERS", None)
    # the class: parameters packed on the device from the nn.Module's tensors, training reduces the loss
    torch.manual_seed(1)
    model = CVAEMemRand(vo.make_config(L, X, mem_type=mem, compress_context=False, device="cuda:0"))
    assert isinstance(model.ctx_encoder.final_compression, torch.nn.Identity)
    assert model.decoder.mem.weight_ih_l0.shape[1] == cfg["mem_hidden"] + L
    x = {"surface": torch.from_numpy(win[:16, :, :, :].copy())}
    if X:
        x["ex_feats"] = torch.from_numpy(wex[:16].copy())
    sd = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in model.state_dict().items()
          if not k.startswith("quantile_loss_fn")}
    e_in = rng.standard_normal((16, C, L))
    want_f = vo.forward(sd, cfg, x["surface"].numpy().astype(np.float64),
                        x["ex_feats"].numpy().astype(np.float64) if X else None, e_in, 1)
    got = paths.forward(model._engine(), x["surface"].cuda(), x["ex_feats"].cuda() if X else None, 1, eps=_t(e_in))
    assert err(_n(got[0]), want_f[0]) <= TOL
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    first = None
    for _ in range(20):
        losses = model.train_step(x, opt)
        assert torch.isfinite(losses["loss"])
        first = first if first is not None else float(losses["loss"])
    assert float(losses["loss"]) < first
    got2 = model.forward(x)
    assert got2[0].shape == (16, 1, 3, 5, 5)


def test_one_launch_adamw_step_equals_torch_adamw():
    """train_step's optimizer.step() runs as vsv_adamw_step when the optimizer is a plain torch.optim.AdamW over the
    model's parameters: the parameters, the optimizer's exp_avg / exp_avg_sq and its step counts after several steps
    against the same steps taken with torch's own step() (VSV_NATIVE_ADAMW=0), from the same seeds."""
    from vae.cvae_with_mem_randomized import CVAEMemRand
    rng = np.random.default_rng(3)
    x = {"surface": torch.from_numpy((0.2 + 0.1 * rng.standard_normal((16, 9, 5, 5))).astype(np.float32)),
         "ex_feats": torch.from_numpy((0.01 * rng.standard_normal((16, 9, 3))).astype(np.float32))}
    runs = []
    for native in ("1", "0"):
        os.environ["VSV_NATIVE_ADAMW"] = native
        try:
            torch.manual_seed(4)
            model = CVAEMemRand(vo.make_config(5, 3, device="cuda:0"))
            opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.02)
            for _ in range(6):
                model.train_step(x, opt)
            torch.cuda.synchronize()
            runs.append(([p.detach().clone() for p in model.parameters()],
                         [opt.state[p]["exp_avg"].clone() for p in model.parameters()],
                         [opt.state[p]["exp_avg_sq"].clone() for p in model.parameters()],
                         [float(opt.state[p]["step"]) for p in model.parameters()]))
        finally:
            os.environ.pop("VSV_NATIVE_ADAMW", None)
    (pa, ma, va, sa), (pb, mb, vb, sb) = runs
    assert sa == sb and set(sa) == {6.0}
    for a, b in zip(pa, pb):
        assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max()))
    for a, b in zip(ma + va, mb + vb):
        assert float((a - b).abs().max()) <= 1e-5 * max(1e-12, float(b.abs().max()))
    # the validated-once fast path notices a replaced optimizer state: state_dict round trip in the middle of a run
    # (new state tensors, step counters that are no longer views of one block) against an uninterrupted run
    finals = []
    for reload in (False, True):
        torch.manual_seed(4)
        model = CVAEMemRand(vo.make_config(5, 3, device="cuda:0"))
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.02)
        eps = torch.from_numpy(np.random.default_rng(9).standard_normal((8, 16, 9, 5)).astype(np.float32))
        for k in range(8):
            if reload and k == 4:
                sd = {kk: (vv.clone() if torch.is_tensor(vv) else vv) for kk, vv in opt.state_dict().items()}
                import copy
                opt.load_state_dict(copy.deepcopy(opt.state_dict()))
                assert float(opt.state[next(model.parameters())]["step"]) == 4.0
            model.train_step(x, opt, _eps=eps[k])
        assert {float(opt.state[q]["step"]) for q in model.parameters()} == {8.0}
        finals.append([q.detach().clone() for q in model.parameters()])
    for a, b in zip(*finals):
        assert torch.equal(a, b)
    # anything but the plain case keeps torch's step(): amsgrad here
    model = CVAEMemRand(vo.make_config(5, 0, device="cuda:0"))
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, amsgrad=True)
    for _ in range(3):
        model.train_step({"surface": x["surface"]}, opt)
    assert "max_exp_avg_sq" in opt.state[next(model.parameters())]


@pytest.mark.parametrize("mem,L,X,sh,csh", [("lstm", 5, 3, (40, 16, 8), (40, 16, 8)), ("lstm", 12, 0, (24, 12), (30,)),
                                            ("gru", 5, 3, (96,), (7, 33))])
def test_dense_surface_mc_product_and_model_class(mem, L, X, sh, csh):
    """use_dense_surface=True (reference :84-93, :237-246, :375-380, :394-407): Linear stacks in place of the conv
    embeddings and the conv-transpose head, decoder memory of width surface_hidden[-1] + ex_feats_dim.  The goldens
    named *_dense_* cover forward / generation / gradients / AR / conditional prior against the reference; here the
    Monte-Carlo product (shared prefix, fused one-wave decode with Philox noise, one-channel output, percentile
    reduction, the segmented prefix) against the oracle, and the drop-in class (state_dict keys, device-side packing,
    training)."""
    from vsv import paths
    from vae.cvae_with_mem_randomized import CVAEMemRand
    C, H, S, N = 6, 2, 1000, 40
    mk = dict(mem_type=mem, use_dense_surface=True, surface_hidden=sh)
    cfg = vo.make_config(L, X, **mk)
    cfg["ctx_surface_hidden"] = list(csh)
    eng, W = _engine(cfg, 63, 2.0)
    rng = np.random.default_rng(31)
    vol = (0.2 + 0.1 * rng.standard_normal((N + C, 5, 5))).astype(np.float32)
    ex = (0.01 * rng.standard_normal((N + C, X))).astype(np.float32) if X else None
    q = paths.prior_mc_windows(eng, _t(vol), _t(ex), C, N, C, H, S, seed=7, chunk_days=16, channel=None)
    eps = _n(eng.philox_normal(7, C, N, S, 0, H)).reshape(N, S, H, L)
    win = np.stack([vol[d:d + C] for d in range(N)])
    wex = np.stack([ex[d:d + C] for d in range(N)]) if X else None
    smp, _ = vo.prior_mc_shared_prefix(W, cfg, win, wex, eps.astype(np.float32))
    want = np.percentile(smp.astype(np.float64), [5, 50, 95], axis=1).transpose(1, 2, 0, 3, 4, 5)
    assert err(_n(q), want) <= TOL
    # horizon 1, one channel: the one-wave decode kernel with the packed one-channel output section
    q1 = paths.prior_mc_windows(eng, _t(vol), _t(ex), C, N, C, 1, S, seed=7, channel=1)
    assert err(_n(q1), want[:, :1, :, 1]) <= TOL
    s1, g1 = paths._context_prefix(eng, _t(vol), _t(ex), N, C, n_seg=1)
    for layers in ("0", "1"):
        os.environ["VSV_PREFIX_LAYERS"] = layers
        try:
            s2, g2 = paths._context_prefix(eng, _t(vol), _t(ex), N, C, n_seg=2)
            torch.cuda.synchronize()
            assert torch.equal(s1, s2) and torch.equal(g1, g2), layers
        finally:
            os.environ.pop("VSV_PREFIX_LAYERS", None)
    torch.manual_seed(1)
    mcfg = vo.make_config(L, X, device="cuda:0", **mk)
    mcfg["ctx_surface_hidden"] = list(csh)
    model = CVAEMemRand(mcfg)
    assert set(k for k in model.state_dict() if not k.startswith("quantile_loss_fn")) == set(k for k, _ in vo.state_dict_spec(cfg))
    assert model.decoder.mem.hidden_size == sh[-1] + X
    x = {"surface": torch.from_numpy(win[:16].copy())}
    if X:
        x["ex_feats"] = torch.from_numpy(wex[:16].copy())
    sd = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in model.state_dict().items()
          if not k.startswith("quantile_loss_fn")}
    e_in = rng.standard_normal((16, C, L))
    want_f = vo.forward(sd, cfg, x["surface"].numpy().astype(np.float64),
                        x["ex_feats"].numpy().astype(np.float64) if X else None, e_in, 1)
    got = paths.forward(model._engine(), x["surface"].cuda(), x["ex_feats"].cuda() if X else None, 1, eps=_t(e_in))
    assert err(_n(got[0]), want_f[0]) <= TOL
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    first = None
    for _ in range(20):
        losses = model.train_step(x, opt)
        assert torch.isfinite(losses["loss"])
        first = first if first is not None else float(losses["loss"])
    assert float(losses["loss"]) < first
    assert model.forward(x)[0].shape == (16, 1, 3, 5, 5)
