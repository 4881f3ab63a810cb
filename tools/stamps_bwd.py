#!/usr/bin/env python3
"""Step timeline of lstm_bwd4_kernel<16> (the decoder's BPTT recurrence at batch 64) from a diagnostic build
(make stamps; VSV_LIB=.../libvsv_hip_stamps.so): one step from the middle of a T = 21 sequence.
The decoder's layer-0 launch is the last <16> launch of a step, so its stamps are what the buffer holds."""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "vol-surface-vae-pub_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from oracle import vsv_oracle as vo
from vae.cvae_with_mem_randomized import CVAEMemRand
T, B = 21, 64
cfg = vo.make_config(5, 3, device="cuda:0", hip_graphs=False)
model = CVAEMemRand(dict(cfg))
model.train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-5)
rng = np.random.default_rng(0)
b = {"surface": torch.from_numpy((0.2 + 0.1 * rng.standard_normal((B, T, 5, 5))).astype(np.float32)).cuda(),
     "ex_feats": torch.from_numpy((0.01 * rng.standard_normal((B, T, 3))).astype(np.float32)).cuda()}
for _ in range(5):
    model.train_step(b, opt)
torch.cuda.synchronize()
lib = model._engine().lib
n = 16384 * 4 * 32
buf = (ctypes.c_ulonglong * n)()
lib.vsv_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
lib.vsv_debug_stamps.restype = ctypes.c_int
assert lib.vsv_debug_stamps(buf, n) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(16384 * 4, 32).astype(np.int64)
st = st[(st[:, 0] > 0) & (st[:, 5] > st[:, 0])][:64]
d = np.diff(st[:, :6], axis=1)
T = 21
print(f"{len(st)} waves stamped (cycles)")
for i, nm in enumerate(["products", "partials to LDS", "barrier 1", "sum + cell backward + stores", "barrier 2"]):
    print(f"  {nm:30s} median {np.median(d[:, i]):7.0f}  p10 {np.percentile(d[:, i], 10):7.0f}  p90 {np.percentile(d[:, i], 90):7.0f}")
print(f"  stamped part of the step: median {np.median(st[:, 5] - st[:, 0]):7.0f}")
print(f"  kernel start -> time loop (W_hh^T gather, first fetch): median {np.median(st[:, 7] - st[:, 6]):7.0f}")
print(f"  time loop, {T} steps: median {np.median(st[:, 8] - st[:, 7]):7.0f} = {np.median(st[:, 8] - st[:, 7]) / T:.0f} per step")
