#!/usr/bin/env python3
"""Date-range prefix (vsv_context_prefix) over day counts x time segments x worker streams: wall time per call
(HIP events around 20 calls) and bit-equality with the unsegmented result.
   python tools/prefix_sweep.py [days ...]"""
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
days = [int(a) for a in sys.argv[1:]] or [250, 500, 1000, 2000]
for N in days:
    v, e = synth.synthetic_calendar(N, C, X)
    vol, ex = torch.from_numpy(v).cuda(), torch.from_numpy(e).cuda()
    ref = None
    for n_seg, n_str in [(1, 2), (1, 4), (2, 4), (2, 6), (3, 6), (4, 6), (6, 6), (10, 6)]:
        os.environ["VSV_PREFIX_STREAMS"] = str(n_str)
        for _ in range(3):
            out = paths._context_prefix(eng, vol, ex, N, C, n_seg)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        import time
        a.record()
        h0 = time.perf_counter()
        for _ in range(20):
            out = paths._context_prefix(eng, vol, ex, N, C, n_seg)
        host = (time.perf_counter() - h0) / 20 * 1e3
        b.record()
        torch.cuda.synchronize()
        if ref is None:
            ref = out
        same = torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
        print(f"days {N:5d}  segments {n_seg:2d}  streams {n_str}  {a.elapsed_time(b) / 20:7.3f} ms  (host {host:.3f} ms to issue)  bit-equal {same}", flush=True)
