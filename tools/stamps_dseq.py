#!/usr/bin/env python3
"""Phase timeline of dec_seq_kernel (time step 1 of every wave) from a diagnostic build (-DVSV_STAMPS):
   hipcc ... -DVSV_STAMPS -o lib/libvsv_hip_stamps.so ; VSV_LIB=.../libvsv_hip_stamps.so python tools/stamps_dseq.py"""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "vol-surface-vae-pub_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
from vsv import paths, synth
from vsv.engine import Engine
L, X, C, S, N, H = 12, 3, 60, 1000, 262, 4
cfg = synth.make_config(L, X)
W = synth.init_weights(cfg, 0, 1.0, np.float32)
dev = torch.device("cuda", 0)
eng = Engine(cfg, W, dev)
v, e = synth.synthetic_calendar(N, C, X)
vol, ex = torch.from_numpy(v).to(dev), torch.from_numpy(e).to(dev)
for _ in range(2):
    paths.prior_mc_windows(eng, vol, ex, C, N, C, H, S, seed=0, chunk_days=N)
torch.cuda.synchronize()
n = 16384 * 4 * 32
buf = (ctypes.c_ulonglong * n)()
eng.lib.vsv_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
eng.lib.vsv_debug_stamps.restype = ctypes.c_int
assert eng.lib.vsv_debug_stamps(buf, n) == 0
nw = (N * S + 31) // 32
st = np.frombuffer(buf, dtype=np.uint64).reshape(16384 * 4, 32).astype(np.int64)[:nw]
st = st[1024:]                                    # not the first round of waves
names = [f"L{s // 4} p{s % 4} {k}" for s in range(8) for k in ("gemm", "cell+fetch")] + ["columns"] + \
        [f"head{li} {k}" for li in range(4) for k in ("mfma", "epilogue")]
d = np.diff(st[:, :26], axis=1)
print("phase                median    p10     p90   (cycles per wave, one wave per SIMD)")
tot = 0
for i, nm in enumerate(names):
    x = d[:, i]
    tot += np.median(x)
    print(f"{nm:18s} {np.median(x):9.0f} {np.percentile(x, 10):7.0f} {np.percentile(x, 90):7.0f}")
print(f"head layer 1, tile 1: next tile's 16 fragment loads {np.median(st[:, 28] - st[:, 27]):.0f}, chain of {np.median(st[:, 31]):.0f} quads "
      f"{np.median(st[:, 29] - st[:, 28]):.0f} cycles")
life = st[:, 26] - st[:, 0]
print(f"step {np.median(life):9.0f}  (sum of medians {tot:.0f}; matrix work of a full step: 3920 products x 64 = 250880)")
