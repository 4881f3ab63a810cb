#!/usr/bin/env python3
"""cProfile of the host side of train_step (eager launches).  python tools/train_hostprof.py [steps]"""
import cProfile, os, pstats, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "vol-surface-vae-pub_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from oracle import vsv_oracle as vo
from vae.cvae_with_mem_randomized import CVAEMemRand
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
cfg = vo.make_config(5, 3, device="cuda:0")
model = CVAEMemRand(dict(cfg))
model.train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-5)
rng = np.random.default_rng(0)
b = {"surface": torch.from_numpy((0.2 + 0.1 * rng.standard_normal((64, 11, 5, 5))).astype(np.float32)).to(dev),
     "ex_feats": torch.from_numpy((0.01 * rng.standard_normal((64, 11, 3))).astype(np.float32)).to(dev)}
for _ in range(10):
    model.train_step(b, opt)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    model.train_step(b, opt)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
