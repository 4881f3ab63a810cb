#!/usr/bin/env python3
"""Time of the captured part of a training step alone (pack -> forward -> loss -> backward -> parameter gradients: one
HIP-graph replay, no clip / optimizer), c1 shape, vs the whole train_step.   python tools/train_graph_time.py [T]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "vol-surface-vae-pub_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from oracle import vsv_oracle as vo
from vae.cvae_with_mem_randomized import CVAEMemRand
T = int(sys.argv[1]) if len(sys.argv) > 1 else 11
B = 64
dev = torch.device("cuda", 0)
cfg = vo.make_config(5, 3, device="cuda:0")
model = CVAEMemRand(dict(cfg))
model.train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-5)
rng = np.random.default_rng(0)
b = {"surface": torch.from_numpy((0.2 + 0.1 * rng.standard_normal((B, T, 5, 5))).astype(np.float32)).to(dev),
     "ex_feats": torch.from_numpy((0.01 * rng.standard_normal((B, T, 3))).astype(np.float32)).to(dev)}
for _ in range(5):
    model.train_step(b, opt)
g = next(iter(model.__dict__["_vsv_graphs"]._d.values()))
torch.cuda.synchronize()
for name, fn in (("graph replay only", lambda: g.graph.replay()), ("train_step", lambda: model.train_step(b, opt))):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500):
        fn()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"T={T} {name:18s}: {(time.perf_counter() - t0) / 500 * 1e3:.3f} ms per call (host issue {th / 500 * 1e3:.3f})", flush=True)
