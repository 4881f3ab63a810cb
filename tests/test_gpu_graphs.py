"""HIP-graph replay of the training step (vsv/graphed.py) against the eager launches of the same kernels."""
import numpy as np
import pytest
import torch

from oracle import vsv_oracle as vo

pytestmark = pytest.mark.gpu


def _model(p=0.0, **kw):
    from vae.cvae_with_mem_randomized import CVAEMemRand
    torch.manual_seed(0)
    kw.setdefault("hip_graphs", True)
    cfg = vo.make_config(5, 3, device="cuda:0", kl_weight=1e-3, mem_dropout=p, **kw)
    m = CVAEMemRand(cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in vo.init_weights(cfg, 4, 1.5, np.float32).items()}, strict=False)
    m.train()
    return m


def _batch(rng, B, T):
    return {"surface": torch.from_numpy((0.2 + 0.1 * rng.standard_normal((B, T, 5, 5))).astype(np.float32)).cuda(),
            "ex_feats": torch.from_numpy((0.01 * rng.standard_normal((B, T, 3))).astype(np.float32)).cuda()}


@pytest.mark.parametrize("p", [0.0, 0.3])
def test_graphed_step_matches_eager_launches(p):
    """Same torch generator state -> the replay's torch.randn noise (and dropout seed word) can be
    re-drawn eagerly; losses and all parameter gradients must agree with the eager path, on replays
    with new batches and after an in-place parameter update (the packing runs inside the graph)."""
    model = _model(p)
    rng = np.random.default_rng(2)
    B, T, h = 24, 9, 1
    te = model._train_engine()
    model._loss_terms(_batch(rng, B, T), h, True)                    # eager warm-up passes + capture draw noise too
    for it in range(3):
        x = _batch(rng, B, T)
        torch.cuda.manual_seed(100 + it)
        losses, grads = model._loss_terms(x, h, True)
        grads = [g.clone() for g in grads]
        assert isinstance(grads, list) and len(model.__dict__["_vsv_graphs"]._d) == 1      # one capture, then replays
        torch.cuda.manual_seed(100 + it)
        eps = torch.randn(B, T, 5, dtype=torch.float32, device="cuda:0")
        seed = int(torch.empty(1, dtype=torch.int64, device="cuda:0").random_().item()) if p > 0 else 0
        model._engine()                                              # eager packing of the current parameters
        want, gp, _ = te.loss_and_grads(x["surface"], x["ex_feats"], h, model._loss_cfg(), eps=eps,
                                        dropout_p=p, dropout_seed=seed)
        for k in want:
            assert abs(float(losses[k]) - float(want[k])) <= 1e-6 * abs(float(want[k])) + 1e-12, (it, k)
        for (n, _), g, w in zip(model._hip_named_params(), grads, te.param_grads(gp)):
            assert float((g - w).norm()) <= 1e-5 * float(w.norm()) + 1e-9, (it, n)
        with torch.no_grad():                                        # in-place update, as an optimizer does
            for q in model.parameters():
                if q.requires_grad:
                    q.add_(0.01 * torch.randn_like(q))


def test_train_step_over_sequence_lengths_uses_one_graph_per_shape():
    model = _model(0.2)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    rng = np.random.default_rng(3)
    xs = {T: _batch(rng, 16, T) for T in (2, 5, 9)}
    first, last = {}, {}
    for ep in range(12):
        for T, x in xs.items():
            l = model.train_step(x, opt)
            assert torch.isfinite(l["loss"]) and set(l) == {"loss", "re_surface", "re_ex_feats", "reconstruction_loss", "kl_loss"}
            gn = torch.sqrt(sum((q.grad.double() ** 2).sum() for q in model.parameters() if q.grad is not None))
            assert gn <= 1.0 + 1e-4
            first.setdefault(T, float(l["loss"]))
            last[T] = float(l["loss"])
    assert len(model.__dict__["_vsv_graphs"]._d) == 3
    assert all(last[T] < first[T] for T in xs)
    # inference after training sees the trained weights (the blob the last replay packed, or a fresh pack)
    model.eval()
    with torch.no_grad():
        a = model.get_surface_given_conditions({k: v[:, :4] for k, v in xs[5].items()}, z=torch.zeros(16, 5, 5, device="cuda:0"))
        model2 = _model(0.2, hip_graphs=False)
        model2.load_state_dict(model.state_dict())
        model2.eval()
        b = model2.get_surface_given_conditions({k: v[:, :4] for k, v in xs[5].items()}, z=torch.zeros(16, 5, 5, device="cuda:0"))
    assert torch.allclose(a[0], b[0], atol=1e-6) and torch.allclose(a[1], b[1], atol=1e-6)
    # multi-horizon accumulates over replays of different graphs without aliasing their outputs
    model.train()
    mh = model.train_step_multihorizon(xs[9], opt, horizons=[1, 3, 3])
    assert np.isfinite(mh["loss"])


@pytest.mark.parametrize("p", [0.0, 0.25])
def test_graphed_multihorizon_step_matches_eager_step(p):
    """train_step_multihorizon as one replay (the horizons are row groups of one captured forward / backward) against
    the same step launched eagerly: same injected noise per horizon -> same reported floats and, with dropout off, the
    same parameters after every step (with dropout the two draw their mask seeds differently: losses stay finite and
    the graph is captured once)."""
    rng = np.random.default_rng(5)
    B, T, hs = 16, 12, [1, 3, 7]
    xs = [_batch(rng, B, T) for _ in range(3)]
    eps = [[torch.from_numpy(rng.standard_normal((B, T, 5)).astype(np.float32)) for _ in hs] for _ in xs]
    outs = {}
    for graphs in (True, False):
        model = _model(p, hip_graphs=graphs)
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
        res = []
        for x, e in zip(xs, eps):
            r = model.train_step_multihorizon(x, opt, horizons=hs, _eps_per_horizon=e)
            assert set(r) == {"loss", "reconstruction_loss", "kl_loss", "horizon_losses"} and set(r["horizon_losses"]) == set(hs)
            assert all(isinstance(r[k], float) for k in ("loss", "reconstruction_loss", "kl_loss"))
            res.append(r)
        outs[graphs] = (res, [q.detach().clone() for q in model.parameters()])
        if graphs:
            assert len(model.__dict__["_vsv_graphs"]._d) == 1
    if p == 0.0:
        for a, b in zip(outs[True][0], outs[False][0]):
            for k in ("loss", "reconstruction_loss", "kl_loss"):
                assert abs(a[k] - b[k]) <= 1e-9 * abs(b[k]) + 1e-12, k
            for h in hs:
                assert abs(a["horizon_losses"][h] - b["horizon_losses"][h]) <= 1e-9 * abs(b["horizon_losses"][h]) + 1e-12
        for qa, qb in zip(outs[True][1], outs[False][1]):
            assert torch.equal(qa, qb)
    else:
        assert all(np.isfinite(r["loss"]) for r in outs[True][0])
