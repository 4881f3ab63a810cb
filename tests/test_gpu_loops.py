"""the training / evaluation loops of vae/utils.py on the HIP model, end to end on a synthetic
VolSurfaceDataSetRand -- vae.utils.train (reference :45-100), train_with_scheduled_sampling (:102-260), model_eval
(:23-42), test (:262-270) and BaseVAE.save_weights / load_weights (base.py:126-149): log-file lines, the best-validation
checkpoint, its round trip and the loss-dict contracts."""
import os
import re

import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader

from oracle import vsv_oracle as vo

pytestmark = pytest.mark.gpu


def _loaders(X, n=96, seed=0, min_len=4, max_len=9, batch=16):
    from vae.datasets_randomized import CustomBatchSampler, VolSurfaceDataSetRand
    rng = np.random.default_rng(seed)
    surf = (0.2 + 0.1 * rng.standard_normal((n, 5, 5))).astype(np.float32)
    ex = (0.01 * rng.standard_normal((n, X))).astype(np.float32)
    out = []
    for lo, hi in ((0, 64), (64, 80), (80, 96)):
        ds = VolSurfaceDataSetRand((surf[lo:hi], ex[lo:hi]) if X else surf[lo:hi], min_seq_len=min_len, max_seq_len=max_len)
        out.append(DataLoader(ds, batch_sampler=CustomBatchSampler(ds, batch, min_len)))
    return out


def _model(X, **kw):
    from vae.cvae_with_mem_randomized import CVAEMemRand
    cfg = vo.make_config(5, X, device="cuda:0", kl_weight=1e-5, **kw)
    torch.manual_seed(0)
    return CVAEMemRand(cfg)


def _epoch_blocks(text):
    return re.findall(r"epoch (\d+): \ntrain loss :: (.*), \ndev loss :: (.*), \ntime elapsed :: ([0-9.e+-]+)", text)


def test_train_eval_test_loops_and_checkpoint_round_trip(tmp_path):
    from vae.cvae_with_mem_randomized import CVAEMemRand
    from vae.utils import model_eval, set_seeds, test, train
    set_seeds(0)
    tr, va, te = _loaders(3)
    model = _model(3)
    d = str(tmp_path / "ckpt")
    train_losses, dev_losses = train(model, tr, va, lr=1e-3, epochs=2, model_dir=d, file_name="unit.pt")
    keys = {"loss", "re_surface", "re_ex_feats", "reconstruction_loss", "kl_loss"}
    assert set(train_losses) == keys and set(dev_losses) == keys
    assert all(isinstance(v, float) and np.isfinite(v) for v in list(train_losses.values()) + list(dev_losses.values()))
    # log file: name, header and one block per epoch in the reference's format (utils.py:59-66, 93-99)
    log = open(os.path.join(d, "unit-2-log.txt"), encoding="utf-8").read()
    assert log.startswith("Model config: \n{") and "LR: 0.001\nEpochs: 2\n" in log
    blocks = _epoch_blocks(log)
    assert [b[0] for b in blocks] == ["0", "1"]
    assert all(re.fullmatch(r"(\w+: -?\d+\.\d{3}(, )?)+", b[1]) and "kl_loss" in b[2] for b in blocks)
    assert "training finished, total time :: " in log
    # best-validation checkpoint written by save_weights: the reference's dictionary (base.py:126-137)
    ck_path = os.path.join(d, "unit.pt")
    ck = torch.load(ck_path, weights_only=False)
    assert set(ck) == {"model", "optimizer", "model_config", "system_rng", "numpy_rng", "torch_rng"}
    assert ck["model_config"]["num_quantiles"] == 3 and ck["model_config"]["padding"] == 1
    assert set(ck["model"]) == set(model.state_dict())
    # load_weights round trip into a fresh model: same evaluation losses as the checkpointed weights give
    m2 = CVAEMemRand(dict(ck["model_config"]))
    m2.load_weights(f=ck_path)
    for k, v in m2.state_dict().items():
        assert torch.equal(v.cpu(), ck["model"][k].cpu()), k
    m3 = _model(3)
    m3.load_weights(dict_to_load=ck)
    # (fresh loaders: the sampler shuffles its groups in place, like the reference's, so a loader's batches
    # depend on how often it has been iterated)
    set_seeds(1)
    a = model_eval(m2, _loaders(3)[1])
    set_seeds(1)
    b = model_eval(m3, _loaders(3)[1])
    assert a.keys() == b.keys() and all(abs(a[k] - b[k]) <= 1e-6 * max(1.0, abs(a[k])) for k in a), (dict(a), dict(b))
    # test(): loads the file, evaluates both loaders
    dev, tst = test(m3, va, te, model_fn=ck_path)
    assert set(dev) == keys and set(tst) == keys and np.isfinite(tst["loss"])
    # training moved the weights and did not blow up
    fresh = _model(3).state_dict()
    assert any(not torch.equal(fresh[k].cpu(), ck["model"][k].cpu()) for k in fresh if k.endswith("weight"))


def test_scheduled_sampling_loop_switches_phases(tmp_path):
    from vae.utils import set_seeds, train_with_scheduled_sampling
    set_seeds(0)
    tr, va, _ = _loaders(3, min_len=6, max_len=9)
    model = _model(3)
    d = str(tmp_path / "ss")
    train_losses, dev_losses = train_with_scheduled_sampling(model, tr, va, lr=1e-3, epochs=2, model_dir=d,
                                                             file_name="ss_model.pt", teacher_forcing_epochs=1,
                                                             horizons=[1, 3])
    # phase 2 (multi-horizon) ran last: its loss dict has the per-horizon entries (utils.py:213-219)
    assert {"loss", "reconstruction_loss", "kl_loss", "horizon_losses_h1", "horizon_losses_h3"} <= set(train_losses)
    assert set(dev_losses) == {"loss", "re_surface", "re_ex_feats", "reconstruction_loss", "kl_loss"}
    log = open(os.path.join(d, "ss_model-2-log.txt"), encoding="utf-8").read()
    for line in ("SCHEDULED SAMPLING TRAINING", "Total Epochs: 2", "Teacher Forcing Epochs: 1 (Phase 1)",
                 "Multi-Horizon Epochs: 1 (Phase 2)", "Multi-Horizon Targets: [1, 3]", "best validation loss :: "):
        assert line in log, line
    assert len(_epoch_blocks(log)) == 2
    assert os.path.exists(os.path.join(d, "ss_model.pt"))
