#!/usr/bin/env python3
"""BASELINE configs[2] (train_step_multihorizon, horizons [1,7,14,30], batch 64, T = 50) in a loop, for
rocprofv3 --kernel-trace + tools/train_timeline.py.   python tools/train_mh_profile.py [steps]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "vol-surface-vae-pub_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from oracle import vsv_oracle as vo
from vae.cvae_with_mem_randomized import CVAEMemRand
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, T = 64, 50
dev = torch.device("cuda", 0)
cfg = vo.make_config(5, 3, device="cuda:0")
model = CVAEMemRand(dict(cfg))
model.load_state_dict({k: torch.from_numpy(v) for k, v in vo.init_weights(cfg, 0, 1.0, np.float32).items()}, strict=False)
model.train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-5)
rng = np.random.default_rng(0)
b = {"surface": torch.from_numpy((0.2 + 0.1 * rng.standard_normal((B, T, 5, 5))).astype(np.float32)).to(dev),
     "ex_feats": torch.from_numpy((0.01 * rng.standard_normal((B, T, 3))).astype(np.float32)).to(dev)}
for _ in range(5):
    model.train_step_multihorizon(b, opt, [1, 7, 14, 30])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    model.train_step_multihorizon(b, opt, [1, 7, 14, 30])
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step")
