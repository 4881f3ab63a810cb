"""forward() as a torch.autograd.Function over the HIP kernels (vsv/autograd.py): a caller-written loss
and loss.backward() must give the reference's gradients, exactly as the fused train_step path does."""
import numpy as np
import pytest
import torch

from conftest import cfg_of, golden_names, load_golden
from oracle import vsv_oracle as vo

pytestmark = pytest.mark.gpu

from test_gpu_parity import _check_grads, _n, _t, _train_setup          # noqa: E402


def _leaf_params(cfg, W, dev="cuda:0"):
    spec = vo.state_dict_spec(cfg)
    return spec, [torch.tensor(np.asarray(W[n], dtype=np.float32), device=dev, requires_grad=not n.startswith("quantile_loss_fn"))
                  for n, _ in spec]


def _reference_loss(cfg, qw, s, e, zm, zl, surf, ex, h):
    """train_step's loss (:1113-1125) written with torch ops on the Function's outputs (float64)."""
    C = surf.shape[1] - h
    q = torch.tensor(cfg["quantiles"], dtype=torch.float64, device=s.device)
    w = torch.tensor(qw, dtype=torch.float64, device=s.device)
    tgt = surf[:, C:].double()
    per_q = []
    for i in range(len(q)):
        err = tgt - s[:, :, i].double()
        per_q.append(torch.mean(torch.max((q[i] - 1) * err, q[i] * err)))
    loss = torch.sum(torch.stack(per_q) * w) / torch.sum(w)
    if ex is not None:
        pe, te_ = e.double(), ex[:, C:].double()
        if cfg["ex_loss_on_ret_only"]:
            pe, te_ = pe[:, :, :1], te_[:, :, :1]
        re = torch.mean((pe - te_) ** 2) if cfg["ex_feats_loss_type"] == "l2" else torch.mean(torch.abs(pe - te_))
        loss = loss + cfg["re_feat_weight"] * re
    kl = torch.mean(torch.sum(-0.5 * (1 + zl.double() - torch.exp(zl.double()) - zm.double() ** 2), dim=-1))
    return loss + cfg["kl_weight"] * kl


@pytest.mark.parametrize("name", ["g4_grads_ex5_B8_T21_h1", "g4_grads_noex_B5_T9_h3", "g4_grads_gru_ex5_B6_T21_h3",
                                  "g4_grads_int2_ex12_B4_T20_h3"])
def test_autograd_function_gives_reference_gradients(name):
    from vsv.autograd import CVAEForward
    g = load_golden(name)
    cfg, eng, te = _train_setup(g)
    W = vo.init_weights(cfg, int(g["seed"]), float(g["scale"]), np.float32)
    spec, params = _leaf_params(cfg, W)
    surf, ex, h = _t(g["surface"]), _t(g.get("ex_feats")), int(g["horizon"])
    s, e, zm, zl, z = CVAEForward.apply(te, surf, ex, h, _t(g["eps"]), 0, 0.0, 0, *params)
    assert s.requires_grad and zm.requires_grad and z.requires_grad
    qw = [float(x) for x in g["cfg_quantile_loss_weights"]]
    loss = _reference_loss(cfg, qw, s, e if ex is not None else None, zm, zl, surf, ex, h)
    assert abs(float(loss) - float(g["loss_loss"])) <= 1e-3 * abs(float(g["loss_loss"]))
    loss.backward()
    grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in params]
    _check_grads(g, [n for n, _ in spec], grads)


@pytest.mark.parametrize("mem", ["lstm", "gru"])
def test_autograd_function_custom_loss_and_input_gradient(mem):
    """A loss the fused loss kernel does not know (every output used, z included) and a surface that
    requires grad: parameter gradients and d(loss)/d(surface) against the torch-port autograd oracle."""
    from oracle.torch_port import TorchPort
    from vsv.autograd import CVAEForward
    from vsv.engine import Engine
    from vsv.train import TrainEngine
    L, X, B, T, h = 5, 3, 9, 11, 3
    cfg = vo.make_config(L, X, mem_type=mem)
    W = vo.init_weights(cfg, 5, 1.5, np.float32)
    eng = Engine(cfg, W, "cuda:0")
    te = TrainEngine(eng, vo.state_dict_spec(cfg))
    spec, params = _leaf_params(cfg, W)
    rng = np.random.default_rng(3)
    surf_np = (0.2 + 0.1 * rng.standard_normal((B, T, 5, 5))).astype(np.float32)
    ex_np = (0.01 * rng.standard_normal((B, T, X))).astype(np.float32)
    eps_np = rng.standard_normal((B, T, L)).astype(np.float32)

    def custom(s, e, zm, zl, z, surf):
        return ((s.double() ** 2).mean() + 0.1 * e.double().sum() + (z.double() * zm.double()).mean()
                + zl.double().abs().mean() + (s[:, :, 1].double() * surf[:, -h:].double()).mean())

    surf = _t(surf_np).requires_grad_(True)
    out = CVAEForward.apply(te, surf, _t(ex_np), h, _t(eps_np), 0, 0.0, 0, *params)
    loss = custom(*out, surf)
    loss.backward()
    port = TorchPort(cfg, W, torch.float32)
    ps = torch.from_numpy(surf_np).requires_grad_(True)
    want = custom(*port.forward(ps, torch.from_numpy(ex_np), torch.from_numpy(eps_np), h), ps)
    want.backward()
    assert abs(float(loss) - float(want)) <= 1e-4 * abs(float(want))
    ref = port.named_reference_parameters()
    for (n, _), p in zip(spec, params):
        if n.startswith("quantile_loss_fn"):
            continue
        wg = ref[n].grad.numpy().astype(np.float64)
        nrm = float(np.sqrt((wg ** 2).sum()))
        assert float(np.sqrt(((_n(p.grad) - wg) ** 2).sum())) <= 1e-3 * nrm + 1e-7, n
    ws = ps.grad.numpy().astype(np.float64)
    assert float(np.sqrt(((_n(surf.grad) - ws) ** 2).sum())) <= 1e-3 * float(np.sqrt((ws ** 2).sum()))
    with pytest.raises(RuntimeError):
        loss.backward()                                    # saved activations are gone, like retain_graph=False


def test_model_forward_backward_matches_train_step_gradients():
    """The drop-in class: out = model(x); model's own loss terms on it; loss.backward() -- the user-written
    training loop the reference's forward() allows -- fills p.grad with what train_step's fused path computes."""
    from vae.cvae_with_mem_randomized import CVAEMemRand
    torch.manual_seed(1)
    # hip_graphs False: the eager train path takes its noise from the same Philox key forward() draws
    model = CVAEMemRand(vo.make_config(5, 3, device="cuda:0", kl_weight=1e-3, hip_graphs=False))
    model.train()
    rng = np.random.default_rng(1)
    x = {"surface": torch.from_numpy((0.2 + 0.1 * rng.standard_normal((12, 9, 5, 5))).astype(np.float32)),
         "ex_feats": torch.from_numpy((0.01 * rng.standard_normal((12, 9, 3))).astype(np.float32))}
    torch.manual_seed(7)
    out = model(x)
    assert out[0].grad_fn is not None and out[0].shape == (12, 1, 3, 5, 5)
    losses = model._losses(x, out, model.horizon)
    model.zero_grad()
    losses["loss"].backward()
    got = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    assert len(got) == len([1 for _, p in model.named_parameters() if p.requires_grad])
    model.zero_grad(set_to_none=True)
    torch.manual_seed(7)
    want_losses, gp = model._loss_terms(x, model.horizon, True)
    model._accumulate_grads(gp)
    assert abs(float(losses["loss"]) - float(want_losses["loss"])) <= 1e-5 * abs(float(want_losses["loss"]))
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        w = p.grad.double()
        assert float((got[n].double() - w).norm()) <= 1e-4 * float(w.norm()) + 1e-9, n
    # eval mode under no_grad stays on the inference path (no graph, no saved activations)
    model.eval()
    with torch.no_grad():
        assert model(x)[0].grad_fn is None
