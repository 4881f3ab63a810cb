#!/usr/bin/env python3
"""The c4-shaped autoregressive leg alone (26 days x 10000 rollouts x 30 steps, 13-day chunks), three passes: run under
rocprofv3 --kernel-trace and feed the CSV to tools/trace_busy.py."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "vol-surface-vae-pub_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from vsv import paths, synth
from vsv.engine import Engine
L, X, C, H, S, N = 12, 3, 20, 30, 10000, 26
cfg = synth.make_config(L, X)
eng = Engine(cfg, synth.init_weights(cfg, 0, 1.0, np.float32), torch.device("cuda", 0))
v, e = synth.synthetic_calendar(N, C, X)
vol, ex = torch.from_numpy(v).cuda(), torch.from_numpy(e).cuda()
for i in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    q = paths.ar_mc_windows(eng, vol, ex, C, N, C, H, S, seed=1)
    torch.cuda.synchronize()
    print(f"pass {i}: {(time.perf_counter() - t0) * 1e3:.1f} ms for {N} days", flush=True)
