#!/usr/bin/env python3
"""The 16-row recurrent launches (layer-split prefix, VSV_PREFIX_LAYERS=1) against the fused two-layer launches
(VSV_PREFIX_LAYERS=0): max |difference| of the decoder state and gates, by days and segments."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "vol-surface-vae-pub_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from vsv import paths, synth
from vsv.engine import Engine
for L, X, C, N in ((12, 3, 60, 500), (5, 0, 9, 37), (12, 3, 13, 70)):
    cfg = synth.make_config(L, X)
    eng = Engine(cfg, synth.init_weights(cfg, 0, 2.0, np.float32), torch.device("cuda", 0))
    v, e = synth.synthetic_calendar(N, C, X)
    vol, ex = torch.from_numpy(v).cuda(), (torch.from_numpy(e).cuda() if X else None)
    os.environ["VSV_PREFIX_LAYERS"] = "0"
    s0, g0 = paths._context_prefix(eng, vol, ex, N, C, n_seg=1)
    for segs in (1, 2, 3):
        os.environ["VSV_PREFIX_LAYERS"] = "1"
        s1, g1 = paths._context_prefix(eng, vol, ex, N, C, n_seg=segs)
        torch.cuda.synchronize()
        print(L, X, C, N, "segs", segs, "state", float((s0 - s1).abs().max()), "gates", float((g0 - g1).abs().max()),
              "finite", bool(torch.isfinite(s1).all() and torch.isfinite(g1).all()), flush=True)
