#!/usr/bin/env python3
"""Training-step timing split: wall time per train_step vs GPU-busy time (torch profiler-free: uses
HIP events around the step and rocprofv3 --kernel-trace outside).  python tools/train_profile.py [steps]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "vol-surface-vae-pub_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from oracle import vsv_oracle as vo
from vae.cvae_with_mem_randomized import CVAEMemRand
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
T = int(os.environ.get("T", "11"))
B = 64
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
    model.train_step(b, opt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    model.train_step(b, opt)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"T={T}: {t_all / steps * 1e3:.3f} ms/step wall, host issue {t_issue / steps * 1e3:.3f} ms/step")
