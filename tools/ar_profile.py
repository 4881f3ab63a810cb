import os, sys
REPO = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ.get("GRAFT_REPO_ROOT", ".")
for p in (REPO, os.path.join(REPO, "vol-surface-vae-pub_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from oracle import vsv_oracle as vo
from vsv.engine import Engine
from vsv import paths
cfg = vo.make_config(12, 3)
W = vo.init_weights(cfg, 0, 1.0, np.float32)
eng = Engine(cfg, W, "cuda:0")
rng = np.random.default_rng(0)
N, C, H, S = 8, 20, 30, 10000
vol = torch.from_numpy((0.2 + 0.1 * rng.standard_normal((N + C, 5, 5))).astype(np.float32)).cuda()
ex = torch.from_numpy((0.01 * rng.standard_normal((N + C, 3))).astype(np.float32)).cuda()
for _ in range(2):
    q = paths.ar_mc_windows(eng, vol, ex, C, N, C, H, S, seed=1, chunk_days=8)
torch.cuda.synchronize()
