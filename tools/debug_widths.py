"""Per-parameter gradient error of the HIP training path against the autograd port for odd widths."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vol-surface-vae-pub_amd"))
import numpy as np, torch
from oracle import vsv_oracle as vo
from oracle.torch_port import TorchPort
from vsv.engine import Engine
from vsv.train import TrainEngine

def t(a): return None if a is None else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()

for Hm, sh, X, mem in [(64, (5, 5, 5), 0, "lstm"), (100, (3, 3, 3), 0, "lstm"), (100, (5, 5, 3), 0, "lstm"), (100, (3, 5, 5), 0, "lstm"),
                       (64, (3, 3, 3), 0, "lstm"), (48, (5, 4, 4), 3, "lstm"), (80, (4, 4), 0, "gru")]:
    L, B, T, h = 5, 9, 7, 2
    cfg = vo.make_config(L, X, mem_hidden=Hm, surface_hidden=sh, mem_type=mem)
    cfg.update(ex_loss_on_ret_only=True, ex_feats_loss_type="l2", kl_weight=1e-3, re_feat_weight=1.0)
    W = vo.init_weights(cfg, 5, 2.0, np.float32)
    eng = Engine(cfg, W, "cuda:0")
    rng = np.random.default_rng(6)
    surf = (0.2 + 0.1 * rng.standard_normal((B, T, 5, 5))).astype(np.float32)
    ex = (0.01 * rng.standard_normal((B, T, X))).astype(np.float32) if X else None
    eps = rng.standard_normal((B, T, L)).astype(np.float32)
    spec = vo.state_dict_spec(cfg)
    te = TrainEngine(eng, spec)
    lc = dict(quantiles=cfg["quantiles"], quantile_weights=[1.0, 1.0, 1.0], ex_n=1, ex_l2=1, ex_weight=1.0, kl_weight=1e-3)
    losses, gp, _ = te.loss_and_grads(t(surf), t(ex), h, lc, eps=t(eps))
    port = TorchPort(cfg, W, torch.float32)
    wl = port.losses(torch.from_numpy(surf), None if ex is None else torch.from_numpy(ex), torch.from_numpy(eps), h)
    wl["loss"].backward()
    ref = port.named_reference_parameters()
    print("==", Hm, sh, X, mem, "loss", float(losses["loss"]), float(wl["loss"]))
    for (n, _), gr in zip(spec, te.param_grads(gp)):
        wg = ref[n].grad
        wg = np.zeros(ref[n].shape, dtype=np.float32) if wg is None else wg.numpy()
        nrm = float(np.sqrt((wg.astype(np.float64) ** 2).sum()))
        e = float(np.sqrt(((gr.cpu().numpy() - wg).astype(np.float64) ** 2).sum()))
        flag = "" if e <= 1e-3 * nrm + 1e-7 else "   <<<<"
        print(f"  {n:60s} |g| {nrm:10.3e} err {e:10.3e}{flag}")
