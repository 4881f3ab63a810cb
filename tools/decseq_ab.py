"""A/B of the T-step decode legs (run once with VSV_DSEQ=0 and once without): c3 prior H=30, c3 oracle, c4-shaped AR."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "vol-surface-vae-pub_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from vsv import paths
from vsv.engine import Engine
from vsv.synth import init_weights, make_config, synthetic_calendar

dev = torch.device("cuda:0")
which = sys.argv[1:] or ["h30", "oracle", "ar"]

def timed(fn, n=2):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

L, X, C, S = 12, 3, 60, 1000
cfg = make_config(L, X); W = init_weights(cfg, 0, 1.0, np.float32); eng = Engine(cfg, W, dev)
def cal(n, extra=0, C=C, X=X, seed=77):
    v, e = synthetic_calendar(n + extra, C, X, seed=seed)
    return torch.from_numpy(v).to(dev), (torch.from_numpy(e).to(dev) if X else None)
tag = "DSEQ=" + os.environ.get("VSV_DSEQ", "1")
if "h30" in which:
    n, H = 1000, 30
    v, e = cal(n)
    for chunk in (100, 250):
        dt = timed(lambda: paths.prior_mc_windows(eng, v, e, C, n, C, H, S, seed=1, chunk_days=chunk))
        print(f"{tag} c3 prior H=30 {n} days chunk {chunk}: {dt*1e3:.1f} ms  {n*S*H/dt:.3e} surf/s", flush=True)
if "oracle" in which:
    n, H = 500, 1
    v, e = cal(n, H)
    for chunk in (50, 125):
        dt = timed(lambda: paths.oracle_mc_windows(eng, v, e, C, n, C, H, S, seed=1, chunk_days=chunk))
        print(f"{tag} c3 oracle {n} days chunk {chunk}: {dt*1e3:.1f} ms  {n*S*H/dt:.3e} surf/s", flush=True)
if "ar" in which:
    H, Sa, Ca = 30, 10000, 20
    for n, chunk in ((16, 8), (26, 13), (32, 16)):
        va, ea = cal(n, 0, Ca, X, 78)
        dt = timed(lambda: paths.ar_mc_windows(eng, va, ea, Ca, n, Ca, H, Sa, seed=1, chunk_days=chunk), n=1)
        print(f"{tag} c4 AR {n} days x {Sa} x {H} chunk {chunk}: {dt*1e3:.1f} ms  {n*Sa*H/dt:.3e} surf/s  ({dt/n*64:.2f} s per 64 days)", flush=True)
