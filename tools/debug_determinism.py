import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "vol-surface-vae-pub_amd"))
import numpy as np, torch
from vsv import synth
from vsv.engine import Engine
from vsv.train import TrainEngine
from vsv import paths
for (L, X, B, T, h) in [(5, 3, 14, 6, 1), (5, 3, 16, 9, 1), (5, 3, 64, 11, 1), (5, 3, 8, 21, 1), (5, 0, 14, 6, 1)]:
    cfg = synth.make_config(L, X)
    W = synth.init_weights(cfg, 0, 1.0, np.float32)
    eng = Engine(cfg, W, "cuda:0")
    te = TrainEngine(eng, synth.state_dict_spec(cfg))
    rng = np.random.default_rng(0)
    surf = torch.from_numpy((0.2 + 0.1 * rng.standard_normal((B, T, 5, 5))).astype(np.float32)).cuda()
    ex = torch.from_numpy((0.01 * rng.standard_normal((B, T, X))).astype(np.float32)).cuda() if X else None
    lc = dict(quantiles=cfg["quantiles"], quantile_weights=[1.0, 1.0, 1.0], ex_n=1, ex_l2=1, ex_weight=1.0, kl_weight=1e-3)
    outs = []
    for rep in range(4):
        junk = torch.randn(1 << 22, device="cuda") * (rep + 1)      # dirty the allocator's blocks differently each time
        del junk
        losses, gp, o = te.loss_and_grads(surf, ex, h, lc, seed=5, want_grads=(rep % 2 == 1))
        outs.append({k: v.clone() for k, v in o.items() if v is not None})
        outs[-1]["loss"] = losses["loss"].clone()
    for k in outs[0]:
        d = [float((outs[i][k].double() - outs[0][k].double()).abs().max()) for i in range(1, 4)]
        print((L, X, B, T, h), k, d)
    # inference path
    a = paths.forward(eng, surf, ex, h, seed=5)
    junk = torch.randn(1 << 22, device="cuda"); del junk
    b = paths.forward(eng, surf, ex, h, seed=5)
    print("  inference path", [float((x - y).abs().max()) for x, y in zip(a, b) if x is not None])
