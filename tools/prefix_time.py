#!/usr/bin/env python3
"""Latency of the date-range prefix (both encoders + decoder prefix over the context) by variant:
python tools/prefix_time.py [days ...]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "vol-surface-vae-pub_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from vsv import paths, synth
from vsv.engine import Engine
L, X, C = 12, 3, 60
cfg = synth.make_config(L, X)
eng = Engine(cfg, synth.init_weights(cfg, 0, 1.0, np.float32), torch.device("cuda", 0))
for N in [int(a) for a in sys.argv[1:]] or [500, 4000]:
    v, e = synth.synthetic_calendar(N, C, X)
    vol, ex = torch.from_numpy(v).cuda(), torch.from_numpy(e).cuda()
    variants = (("0", 1), ("0", 3), ("0", 6), ("py", 3), ("1", 3), ("1", 4), ("1", 6), ("1", 10), ("1", 15))
    if os.environ.get("PREFIX_VARIANTS"):
        variants = [tuple(x.split(":")) for x in os.environ["PREFIX_VARIANTS"].split(",")]
        variants = [(a, int(b)) for a, b in variants]
    for layers, segs in variants:
        os.environ["VSV_PREFIX_LAYERS"] = layers
        for _ in range(3):
            paths._context_prefix(eng, vol, ex, N, C, n_seg=segs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            paths._context_prefix(eng, vol, ex, N, C, n_seg=segs)
        torch.cuda.synchronize()
        print(f"days {N:5d}  layers {layers}  segments {segs:2d}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms", flush=True)
