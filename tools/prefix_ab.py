#!/usr/bin/env python3
"""Wall time of the date-range prefix (vsv_context_prefix, 60 context days) at the given day counts, HIP events around
30 calls, and a checksum of its outputs: run once per library build (VSV_LIB=...) to compare two builds on one box.
   python tools/prefix_ab.py [days ...]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "vol-surface-vae-pub_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
from vsv import paths, synth
from vsv.engine import Engine

L, X, C = 12, 3, 60
cfg = synth.make_config(L, X)
eng = Engine(cfg, synth.init_weights(cfg, 0, 1.0, np.float32), torch.device("cuda", 0))
for N in [int(a) for a in sys.argv[1:]] or [500, 4000]:
    v, e = synth.synthetic_calendar(N, C, X)
    vol, ex = torch.from_numpy(v).cuda(), torch.from_numpy(e).cuda()
    for _ in range(5):
        out = paths._context_prefix(eng, vol, ex, N, C)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(30):
            out = paths._context_prefix(eng, vol, ex, N, C)
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 30)
    sums = [float(o.double().sum()) for o in out[:2]]
    print(f"{os.environ.get('VSV_LIB', 'default lib'):>40s}  days {N:5d}  prefix {best:7.3f} ms  checksums {sums[0]:.9e} {sums[1]:.9e}", flush=True)
