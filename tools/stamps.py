#!/usr/bin/env python3
"""Phase timeline of the fused MC decode kernel from a diagnostic build (-DVSV_STAMPS):
   hipcc ... -DVSV_STAMPS -o lib/libvsv_hip_stamps.so ; VSV_LIB=.../libvsv_hip_stamps.so python tools/stamps.py
Runs one 500-day x 1000-sample H=1 decode chunk of the bench workload and prints, per phase, the median
cycles per wave, plus how the two workgroups that share a CU overlap in time."""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "vol-surface-vae-pub_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
from oracle import vsv_oracle as vo
from vsv import paths
from vsv.engine import Engine
from bench import synthetic_calendar

H = int(os.environ.get("H", "1"))
L, X, C, S, N = 12, 3, 60, 1000, 500
cfg = vo.make_config(L, X)
W = vo.init_weights(cfg, 0, 1.0, np.float32)
dev = torch.device("cuda", 0)
eng = Engine(cfg, W, dev)
v, e = synthetic_calendar(N, C, X)
vol, ex = torch.from_numpy(v).to(dev), torch.from_numpy(e).to(dev)
if os.environ.get("NOISE") == "0":              # diagnostic: no Philox work in the decode (std0 = 0)
    _dec = eng.decode
    eng.decode = lambda *a, **k: _dec(*a, **{**k, "std0": 0.0})
for _ in range(3):
    paths.prior_mc_windows(eng, vol, ex, C, N, C, H, S, seed=0, chunk_days=500)
torch.cuda.synchronize()
n = 16384 * 4 * 32
buf = (ctypes.c_ulonglong * n)()
eng.lib.vsv_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
eng.lib.vsv_debug_stamps.restype = ctypes.c_int
rc = eng.lib.vsv_debug_stamps(buf, n)
assert rc == 0, rc
st = np.frombuffer(buf, dtype=np.uint64).reshape(16384, 4, 32)[: (N * S + 31) // 32].astype(np.int64)
names = ["prologue", "L0 gemm", "L0 cell", "L0 barriers", "L1 gemm", "L1 cell", "L1 barriers",
         "head0", "head1", "head2", "head3", "head4", "tail"]
t = st[:, :, :14]
d = np.diff(t, axis=2)
print("phase                median    p10     p90   (cycles per wave)")
for i, nm in enumerate(names):
    x = d[:, :, i].ravel()
    print(f"{nm:18s} {np.median(x):9.0f} {np.percentile(x, 10):7.0f} {np.percentile(x, 90):7.0f}")
pro = st[:, :, [0, 16, 17, 18, 1]]
dp = np.diff(pro, axis=2)
for i, nm in enumerate(["  setup (ids, divisions)", "  gate/weight/state loads + h init", "  x assembly (Philox)", "  barrier"]):
    x = dp[:, :, i].ravel()
    print(f"{nm:34s} {np.median(x):9.0f} {np.percentile(x, 10):7.0f} {np.percentile(x, 90):7.0f}")
life = (t[:, :, 13] - t[:, :, 0]).ravel()
print(f"wave lifetime      {np.median(life):9.0f} {np.percentile(life, 10):7.0f} {np.percentile(life, 90):7.0f}")
hw = st[:, 0, 15]
cu = ((hw >> 32) & 0xf) * 4096 + ((hw >> 13) & 7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 0xf)
print("distinct CUs:", len(np.unique(cu)))
# overlap of consecutive workgroups on one CU: start offset between partners relative to lifetime
start = t[:, 0, 0]
end = t[:, 0, 13]
offs = []
for c in np.unique(cu)[:64]:
    idx = np.where(cu == c)[0]
    o = np.argsort(start[idx])
    s, en = start[idx][o], end[idx][o]
    for k in range(1, len(s)):
        offs.append((s[k] - s[k - 1]))
offs = np.array(offs)
print("start-to-start gap of successive WGs on a CU: median %.0f  p10 %.0f  p90 %.0f" %
      (np.median(offs), np.percentile(offs, 10), np.percentile(offs, 90)))
# phase alignment: for a WG, where in its partner's lifetime does it start (fraction)
fr = []
for c in np.unique(cu)[:64]:
    idx = np.where(cu == c)[0]
    o = np.argsort(start[idx])
    s, en = start[idx][o], end[idx][o]
    for k in range(2, len(s)):
        # partner = the other WG alive at s[k]
        alive = [j for j in range(k) if en[j] > s[k]]
        if alive:
            j = alive[-1]
            fr.append((s[k] - s[j]) / max(1, en[j] - s[j]))
fr = np.array(fr)
print("start position inside partner's lifetime (fraction): hist", np.histogram(fr, bins=10, range=(0, 1))[0].tolist())
