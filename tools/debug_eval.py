import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "vol-surface-vae-pub_amd")); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch, random
from test_gpu_loops import _loaders, _model
from vae.utils import model_eval, set_seeds
tr, va, te = _loaders(3)
m = _model(3)
m.eval()
set_seeds(1)
batches = [b for b in va]
print("n batches", len(batches), [tuple(b["surface"].shape) for b in batches])
for rep in range(3):
    set_seeds(1)
    out = []
    for b in batches:
        r = m.test_step(b)
        out.append((float(r["kl_loss"]), float(r["re_surface"])))
    print(rep, out)
# same batch, explicit engine calls
eng = m._engine()
from vsv import paths
b = batches[0]
for rep in range(3):
    zm, zl = paths.encode_both(eng, b["surface"].cuda().float(), b["ex_feats"].cuda().float())
    print("enc", float(zm.double().sum()), float(zl.double().sum()))
te_ = m._train_engine()
for rep in range(3):
    l, _, o = te_.loss_and_grads(b["surface"], b["ex_feats"], 1, m._loss_cfg(), seed=3, want_grads=False)
    print("te", float(l["kl_loss"]), float(o["z_mean"].double().sum()), b["surface"].dtype, b["surface"].device)
for rep in range(3):
    l, _, o = te_.loss_and_grads(b["surface"].cuda().float(), b["ex_feats"].cuda().float(), 1, m._loss_cfg(), seed=3, want_grads=False)
    print("te f32 cuda", float(l["kl_loss"]), float(o["z_mean"].double().sum()))
