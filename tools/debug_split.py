#!/usr/bin/env python3
"""Where does the layer-split prefix differ from the fused one?"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "vol-surface-vae-pub_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from vsv import paths, synth
from vsv.engine import Engine
L, X, N, C = 12, 3, 70, 13
cfg = synth.make_config(L, X)
eng = Engine(cfg, synth.init_weights(cfg, 52, 2.0, np.float32), torch.device("cuda", 0))
rng = np.random.default_rng(23)
vol = torch.from_numpy((0.2 + 0.1 * rng.standard_normal((N + C - 1, 5, 5))).astype(np.float32)).cuda()
ex = torch.from_numpy((0.01 * rng.standard_normal((N + C - 1, X))).astype(np.float32)).cuda()
for which in (0, 1):
    ref, st_ref = eng.encode_windows(which, vol, ex, N, C, want_state=True)
    st = torch.zeros((N, eng.enc_state_floats()), device="cuda")
    ws = eng.encode_windows_layer0(which, vol, ex, N, C, None, st)
    out = eng.encode_windows_layer1(which, N, C, None, st, ws)
    torch.cuda.synchronize()
    print("encoder", which, "out diff", float((ref - out).abs().max()), "state diff", float((st_ref - st).abs().max()))
    hp = st.shape[1] // 4
    for l in range(2):
        for k, nm in enumerate(("h", "c")):
            a = st_ref.view(N, 2, 2, hp)[:, l, k]; b = st.view(N, 2, 2, hp)[:, l, k]
            print("   layer", l, nm, float((a - b).abs().max()))
enc, ctx = eng.encode_both_windows(vol, ex, N, C)
zm = enc[..., :L].contiguous()
pre = eng.decode(N, C, C, ctx=ctx, zb=zm, want_state=True, want_gates=True)
nd = torch.zeros((N, eng.state_floats), device="cuda")
g = torch.zeros((N, eng.gate_floats), device="cuda")
wsd = eng.decode_prefix_layer(0, N, C, ctx, zm, None, nd, g)
eng.decode_prefix_layer(1, N, C, None, None, None, nd, g, ws=wsd)
torch.cuda.synchronize()
hp = nd.shape[1] // 4
for l in range(2):
    for k, nm in enumerate(("h", "c")):
        a = pre["state"].view(N, 2, 2, hp)[:, l, k]; b = nd.view(N, 2, 2, hp)[:, l, k]
        print("decoder layer", l, nm, float((a - b).abs().max()))
gp = g.shape[1] // 2
for l in range(2):
    print("decoder gates layer", l, float((pre["gates"].view(N, 2, gp)[:, l] - g.view(N, 2, gp)[:, l]).abs().max()))
