#!/usr/bin/env python3
"""Step timeline of the recurrent launches (lstm_rec4_kernel / lstm_rec16_kernel) from a diagnostic build
(make stamps; VSV_LIB=.../libvsv_hip_stamps.so): the LAST recurrent launch of a date-range prefix over N windows
of 60 days (decoder layer 1) writes the stamps of two consecutive steps from the middle of the sequence.
   python tools/stamps_rec.py [days ...]"""
import ctypes, os, sys
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
eng.lib.vsv_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
eng.lib.vsv_debug_stamps.restype = ctypes.c_int
n = 16384 * 4 * 32
for N in [int(a) for a in sys.argv[1:]] or [500, 4000]:
    v, e = synth.synthetic_calendar(N, C, X)
    vol, ex = torch.from_numpy(v).cuda(), torch.from_numpy(e).cuda()
    for which, label in ((-1, "whole prefix: its last recurrent launch, decoder layer 1 (16 gate tiles)"),):
        for _ in range(3):
            if which >= 0:
                eng.encode_windows(which, vol, ex, N, C)
            else:
                paths._context_prefix(eng, vol, ex, N, C)
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * n)()
        assert eng.lib.vsv_debug_stamps(buf, n) == 0
        st = np.frombuffer(buf, dtype=np.uint64).reshape(16384 * 4, 32).astype(np.int64)
        wv = np.arange(16384 * 4) % 8
        ok = (st[:, 0] > 0) & (st[:, 8] > st[:, 0])
        print(f"--- {N} days, {label}: {int(ok.sum())} waves stamped; cycles")
        for grp, sel in (("waves 0-3", ok & (wv < 4)), ("waves 4-7", ok & (wv >= 4))):
            g = st[sel]
            if not len(g):
                continue
            seq = [0, 1, 4, 5, 2, 3] if g[:, 4].max() > 0 else [0, 1, 2, 3]
            names = (["late cells | k < 16", "barrier B", "k >= 16 | early cells", "h to LDS, stores", "barrier A"]
                     if len(seq) == 6 else ["products", "cell + stores", "barrier"])
            print(f" {grp} (second stamped step):")
            for i, nm in enumerate(names):
                d = g[:, 8 + seq[i + 1]] - g[:, 8 + seq[i]]
                print(f"  {nm:24s} median {np.median(d):7.0f}  p10 {np.percentile(d, 10):7.0f}  p90 {np.percentile(d, 90):7.0f}")
            print(f"  step period (top to top) median {np.median(g[:, 8] - g[:, 0]):7.0f}")
