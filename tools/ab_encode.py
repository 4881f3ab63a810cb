#!/usr/bin/env python3
"""Times the date-range prefix (both encoders + decoder prefix) of the bench workload: VSV_LIB=... python tools/ab_encode.py"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "vol-surface-vae-pub_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from oracle import vsv_oracle as vo
from vsv.engine import Engine
from bench import synthetic_calendar
L, X, C, N = 12, 3, 60, int(os.environ.get("N", "4000"))
cfg = vo.make_config(L, X)
eng = Engine(cfg, vo.init_weights(cfg, 0, 1.0, np.float32), "cuda:0")
v, e = synthetic_calendar(N, C, X)
vol, ex = torch.from_numpy(v).cuda(), torch.from_numpy(e).cuda()
def run():
    enc, ctx = eng.encode_both_strided(vol, 25, 25, ex, X, X, N, C)
    zm = enc[..., :L].contiguous()
    return eng.decode(N, C, C, ctx=ctx, zb=zm, want_state=True, want_gates=True)
for _ in range(3): run()
torch.cuda.synchronize()
for name, fn in (("encoders", lambda: eng.encode_both_strided(vol, 25, 25, ex, X, X, N, C)), ("prefix chain", run)):
    t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) * 100:.3f} ms")
