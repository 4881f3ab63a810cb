"""CPU checks of the drop-in model class: checkpoint contract (state_dict keys and
shapes of SURVEY appendix A), config mutation and error behaviour of the
reference (cvae_with_mem_randomized.py:969-1020, :581, :1043), the C-ABI symbol
table, and the device-side gather packing.  No compute call needs a GPU here."""
import ctypes as C
import re

import numpy as np
import pytest
import torch

from oracle import vsv_oracle as vo


def _model(cfg):
    from vae.cvae_with_mem_randomized import CVAEMemRand
    return CVAEMemRand(cfg)


@pytest.mark.parametrize("L,X,NL", [(5, 0, 2), (5, 3, 2), (12, 3, 2), (5, 0, 1)])
def test_state_dict_contract(L, X, NL):
    cfg = vo.make_config(L, X, mem_layers=NL)
    m = _model(cfg)
    sd = m.state_dict()
    spec = vo.state_dict_spec(cfg)
    want = [n for n, _ in spec] + ["quantile_loss_fn.quantiles", "quantile_loss_fn.weights"]
    assert list(sd.keys()) == want
    for n, shape in spec:
        assert tuple(sd[n].shape) == shape, n
    assert sd["quantile_loss_fn.quantiles"].dtype == torch.float64
    if NL == 2:
        assert len(sd) == (58 if X else 54)
        total = sum(p.numel() for p in m.parameters())
        assert total == {(5, 0): 556813, (5, 3): 569263, (12, 3): 578552}[(L, X)]
    # a reference-shaped checkpoint loads strictly
    W = vo.init_weights(cfg, 3)
    ck = {k: torch.from_numpy(v) for k, v in W.items()}
    ck["quantile_loss_fn.quantiles"] = sd["quantile_loss_fn.quantiles"]
    ck["quantile_loss_fn.weights"] = sd["quantile_loss_fn.weights"]
    m.load_weights(dict_to_load={"model": ck})
    assert torch.equal(m.decoder.mem.weight_hh_l0, ck["decoder.mem.weight_hh_l0"])


def test_config_defaults_and_mutation():
    cfg = {"feat_dim": (5, 5), "latent_dim": 5, "device": "cpu", "kl_weight": 1e-5, "surface_hidden": [5, 5, 5],
           "ex_feats_dim": 0, "mem_hidden": 100, "mem_layers": 2, "ctx_surface_hidden": [5, 5, 5],
           "compress_context": True}
    m = _model(cfg)
    assert m.config is cfg
    assert cfg["padding"] == 1 and cfg["deconv_output_padding"] == 0 and cfg["num_quantiles"] == 3
    assert cfg["mem_type"] == "lstm" and cfg["ex_feats_loss_type"] == "l1" and cfg["re_feat_weight"] == 1.0
    assert m.horizon == 1 and m.kl_weight == 1e-5 and m.device == "cpu"
    assert isinstance(m.ex_feats_loss_fn, torch.nn.L1Loss)
    assert hasattr(m.encoder, "mem") and hasattr(m.ctx_encoder, "final_compression")
    assert m.count_parameters()["num_params"].iloc[-1] == 556813


def test_reference_error_behaviour():
    base = vo.make_config(5, 0)
    with pytest.raises(KeyError):                       # kl_weight is read before defaulting (base.py:38)
        _model({k: v for k, v in base.items() if k != "kl_weight"})
    with pytest.raises(ValueError):
        _model({k: v for k, v in base.items() if k != "mem_hidden"})
    with pytest.raises(ValueError):
        _model(dict(base, horizon=0))
    with pytest.raises(TypeError):
        _model(dict(base, ex_feats_hidden=[4]))
    m = _model(dict(base))
    with pytest.raises(RuntimeError, match="GPU"):      # product path fails loudly without a GPU
        m.get_surface_given_conditions({"surface": torch.zeros(1, 5, 5, 5)})
    with pytest.raises(ValueError):
        m.generate_autoregressive_sequence({"ex_feats": torch.zeros(1, 5, 3)})
    with pytest.raises(ValueError):
        m.generate_autoregressive_offset({"surface": torch.zeros(1, 5, 5, 5)})   # context_len defaults to 60


def test_quantile_loss_module_matches_reference_values():
    from vae.cvae_with_mem_randomized import QuantileLoss
    from conftest import load_golden
    g = load_golden("g3_quantile_loss")
    p, t = torch.from_numpy(g["preds"]), torch.from_numpy(g["target"])
    assert abs(QuantileLoss()(p, t).item() - float(g["loss_w111"])) < 1e-7
    assert abs(QuantileLoss(weights=[5.0, 1.0, 5.0])(p, t).item() - float(g["loss_w515"])) < 1e-7
    assert QuantileLoss()(torch.ones(2, 1, 3, 5, 5), torch.ones(2, 1, 5, 5)).item() == 0.0
    assert QuantileLoss()(p, t).dtype == torch.float64


def test_library_exports_every_declared_symbol():
    import os
    from vsv import lib as vlib
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(here, "include", "vsv_hip.h")).read()
    declared = set(re.findall(r"\b(vsv_[a-z_0-9]+)\s*\(", text))
    assert declared == set(vlib.SIGNATURES), declared ^ set(vlib.SIGNATURES)
    L = vlib.load()
    for name in declared:
        assert hasattr(L, name), name
    assert L.vsv_abi_version() == 23
    # argument validation happens before any GPU work
    d = vlib.make_dims(5, 0, 100, 2, [5, 5, 5], [5, 5, 5], 3)
    assert L.vsv_packed_floats(C.byref(d)) > 0
    bad = vlib.make_dims(40, 0, 100, 2, [5, 5, 5], [5, 5, 5], 3)
    assert L.vsv_packed_floats(C.byref(bad)) == 0
    assert b"latent_dim" in L.vsv_last_error()


@pytest.mark.parametrize("L,X,extra", [(5, 0, {}), (12, 3, {}), (12, 3, {"interaction_layers": 2}),
                                       (5, 3, {"mem_type": "gru", "mem_layers": 1})])
def test_gather_maps_reproduce_host_packing(L, X, extra):
    from vsv.engine import build_gather_maps, dims_from_config, pack_state_dict
    cfg = vo.make_config(L, X, **extra)
    W = vo.init_weights(cfg, 9, 2.0)
    dims = dims_from_config(cfg)
    spec = vo.state_dict_spec(cfg)
    ia, ib = build_gather_maps(dims, spec)
    flat = np.concatenate([[0.0]] + [W[n].ravel() for n, _ in spec]).astype(np.float32)
    got, want = flat[ia] + flat[ib], pack_state_dict(dims, W)
    # everything but the folded Linear . ConvT section of the decoder head is a gather of parameters ...
    from vsv.engine import build_fold_maps
    fm = build_fold_maps(dims)
    mask = np.ones(want.size, bool)
    for lo, hi in (fm["fold_w"], fm["fold_b"]):
        mask[lo:hi] = False
        assert not got[lo:hi].any()
    assert np.array_equal(got[mask], want[mask])
    # ... and that section is the product of two gathered matrices (what set_weights_from_params does on
    # the device), equal to the host packer's float64 fold up to float32 rounding
    A, Bm = got[fm["a_w"]].astype(np.float64), got[fm["b_w"]].astype(np.float64)
    F = A @ Bm
    bF = A @ got[fm["b_b"]].astype(np.float64) + got[fm["a_b"]]
    assert np.allclose(F, want[fm["f_w"]], rtol=1e-5, atol=1e-6) and np.allclose(bF, want[fm["f_b"]], rtol=1e-5, atol=1e-6)
    touched = np.zeros(want.size, bool)
    touched[fm["f_w"].ravel()] = True
    touched[fm["f_b"]] = True
    rest = ~mask & ~touched
    assert not want[rest].any()                       # what the fold maps do not write is zero padding


def test_dataset_and_sampler_match_reference():
    """VolSurfaceDataSetRand / CustomBatchSampler against the reference's own output (golden g10,
    random.seed(0)): same item windows, same batches in the same order."""
    import random
    from conftest import load_golden
    from vae.datasets_randomized import CustomBatchSampler, VolSurfaceDataSetRand
    g = load_golden("g10_sampler")
    ds = VolSurfaceDataSetRand((g["surface"], g["ex_feats"]), min_seq_len=4, max_seq_len=10)
    assert len(ds) == int(g["n_items"])
    item = ds[3 * 64 + 60]
    assert item["surface"].shape[0] == int(g["tail_len"]) and item["ex_feats"].shape == (int(g["tail_len"]), 3)
    assert np.array_equal(item["surface"][0].numpy(), g["tail_first"])
    random.seed(0)
    batches = list(CustomBatchSampler(ds, 16, 4))
    assert [len(b) for b in batches] == g["sizes"].tolist()
    assert [i for b in batches for i in b] == g["flat"].tolist()
    assert [ds[b[0]]["surface"].shape[0] for b in batches] == g["lens"].tolist()
    for b in batches:                                     # a batch never mixes lengths
        assert len({ds[i]["surface"].shape[0] for i in b}) == 1
    # single-array dataset, float32 option
    ds32 = VolSurfaceDataSetRand(g["surface"], 2, 3, dtype=torch.float32)
    assert ds32[0]["surface"].dtype == torch.float32 and "ex_feats" not in ds32[0]


def test_conditional_prior_contract_cpu():
    """CVAEMemRandConditionalPrior: state_dict carries the reference's prior_network.* keys and shapes
    (conditional_prior_network.py:62-82), the base keys are unchanged, and compute without a GPU raises."""
    import torch
    from oracle import vsv_oracle as vo
    from vae.conditional_prior_network import kl_divergence_gaussians
    from vae.cvae_conditional_prior import CVAEMemRandConditionalPrior
    cfg = vo.make_config(12, 3, device="cpu")
    cfg["use_conditional_prior"] = True
    m = CVAEMemRandConditionalPrior(dict(cfg))
    sd = m.state_dict()
    spec = vo.prior_state_dict_spec(cfg)
    assert [k for k in sd if k.startswith("prior_network.")] == [n for n, _ in spec]
    assert all(tuple(sd[n].shape) == s for n, s in spec)
    assert len(sd) == 58 + len(spec)
    assert [n for n, _ in m._hip_named_params()] == [n for n, _ in vo.state_dict_spec(cfg)]
    # the folded prior heads sit in the encoder's slots of the parameter list handed to the packer
    named = dict(m._prior_named_params())
    Wc, bc = m.prior_network.compression.weight, m.prior_network.compression.bias
    Wm, bm = m.prior_network.prior_mean_layer.weight, m.prior_network.prior_mean_layer.bias
    assert torch.allclose(named["encoder.z_mean_layer.weight"], Wm @ Wc)
    assert torch.allclose(named["encoder.z_mean_layer.bias"], Wm @ bc + bm)
    assert named["encoder.mem.weight_hh_l1"] is m.prior_network.mem.weight_hh_l1
    with pytest.raises(RuntimeError):
        m.get_surface_given_conditions({"surface": torch.zeros(1, 5, 5, 5), "ex_feats": torch.zeros(1, 5, 3)})
    # KL of identical Gaussians is zero; a known value otherwise
    z = torch.zeros(2, 3, 4)
    assert float(kl_divergence_gaussians(z, z, z, z)) == 0.0
    assert abs(float(kl_divergence_gaussians(z + 1, z, z, z)) - 2.0) < 1e-6      # 0.5 * sum_4 (1^2)


def test_interaction_layers_state_dict_keys():
    """interaction_layers=2 (v3 conditional-prior trainer): parameter names and shapes of the three
    `interaction` stacks and of the prior network's follow the reference (cvae_with_mem_randomized.py
    :131-142, :283-294, :452-463; conditional_prior_network.py:131-143)."""
    from oracle import vsv_oracle as vo
    from vae.cvae_conditional_prior import CVAEMemRandConditionalPrior
    cfg = vo.make_config(12, 3, device="cpu", interaction_layers=2)
    cfg["use_conditional_prior"] = True
    m = CVAEMemRandConditionalPrior(dict(cfg))
    sd = m.state_dict()
    want = vo.state_dict_spec(cfg) + vo.prior_state_dict_spec(cfg)
    got = [(k, tuple(v.shape)) for k, v in sd.items() if not k.startswith("quantile_loss_fn")]
    assert sorted(got) == sorted(want)
    assert [k for k, _ in got if k.startswith("encoder.")] == [k for k, _ in want if k.startswith("encoder.")]
    assert "decoder.interaction.dec_interact_linear_final.weight" in sd
    assert "prior_network.interaction.prior_interact_linear_1.bias" in sd


def test_mem_types_share_one_blob_shape():
    """mem_type "gru" maps to vsv_dims.mem_type 1 and any value other than "lstm" / "gru" to 2 (the reference builds
    nn.RNN for those, :158-159); all pack into the same blob size as "lstm" (four gate slots per unit: r, z, n_x,
    n_h for the GRU; the pre-activation in slot 0 and zero rows in the others for the RNN)."""
    import ctypes as C
    from oracle import vsv_oracle as vo
    from vsv import lib as L_
    from vsv.engine import dims_from_config, pack_state_dict
    d0 = dims_from_config(vo.make_config(5, 3))
    d1 = dims_from_config(vo.make_config(5, 3, mem_type="gru"))
    d2 = dims_from_config(vo.make_config(5, 3, mem_type="rnn"))
    assert (d0.mem_type, d1.mem_type, d2.mem_type) == (0, 1, 2)
    lib = L_.load()
    assert lib.vsv_packed_floats(C.byref(d0)) == lib.vsv_packed_floats(C.byref(d1)) == lib.vsv_packed_floats(C.byref(d2))
    cfg = vo.make_config(5, 3, mem_type="rnn")
    W = vo.init_weights(cfg, 3, 1.0, np.float32)
    assert W["encoder.mem.weight_hh_l0"].shape == (100, 100) and W["decoder.mem.weight_ih_l0"].shape == (128, 10)
    blob = pack_state_dict(d2, W)
    tot = sum(float(np.abs(v).sum()) for k, v in W.items() if "bias" not in k)
    assert np.isfinite(blob).all() and float(np.abs(blob).sum()) >= tot * (1 - 1e-5)
    cfg = vo.make_config(5, 3, mem_type="gru")
    W = vo.init_weights(cfg, 3, 1.0, np.float32)
    assert W["encoder.mem.weight_hh_l0"].shape == (300, 100)
    blob = pack_state_dict(d1, W)
    # every parameter value lands in the blob exactly once except the summed r/z biases
    tot = sum(float(np.abs(v).sum()) for k, v in W.items() if "bias" not in k and not k.startswith("quantile_loss_fn"))
    assert np.isfinite(blob).all() and float(np.abs(blob).sum()) >= tot * (1 - 1e-5)


def test_hidden_widths_run_on_the_built_tile_counts():
    """The recurrent kernels exist for 13 and 16 gate tiles of 8 hidden units; every other width up to 128 is
    laid out (zero-padded) on the next built size, and a width beyond 128 is refused where the Engine's
    dims are made -- not at the first compute call."""
    from vsv.engine import dims_from_config, pack_state_dict
    from vsv.train import read_layout
    for Hm, sh, X, want in ((100, (5, 5, 5), 3, (13, 16)), (64, (3, 3, 3), 0, (13, 13)), (128, (5, 5, 5), 0, (16, 16)),
                            (8, (4,), 0, (13, 13)), (105, (5, 4, 4), 3, (16, 13))):
        cfg = vo.make_config(5, X, mem_hidden=Hm, surface_hidden=sh)
        d = dims_from_config(cfg)
        lay = read_layout(d)
        assert (lay["NTm"], lay["NTd"]) == want, (Hm, sh, lay["NTm"], lay["NTd"])
        blob = pack_state_dict(d, vo.init_weights(cfg, 0, 1.0, np.float32))
        assert np.isfinite(blob).all()
    with pytest.raises(NotImplementedError, match="mem_hidden"):
        dims_from_config(vo.make_config(5, 0, mem_hidden=200))


def _frozen_configs():
    import json
    import os
    from conftest import GOLDEN
    return json.load(open(os.path.join(GOLDEN, "configs.json")))


@pytest.mark.parametrize("entry", _frozen_configs(), ids=[e["name"] for e in _frozen_configs()])
def test_reference_driver_configs_construct_with_the_reference_contract(entry):
    """The literal config dicts of the reference's training drivers (tests/golden/configs.json, frozen by
    oracle/freeze_configs.py from the drivers' own source): the class built from each -- under the default dtype the
    driver sets -- mutates its config as the reference's check_input / __init__ do and has the reference's state_dict,
    name for name, shape for shape, dtype for dtype."""
    import torch
    from vae.cvae_conditional_prior import CVAEMemRandConditionalPrior
    from vae.cvae_with_mem_randomized import CVAEMemRand
    cls = CVAEMemRandConditionalPrior if entry["model_class"] == "CVAEMemRandConditionalPrior" else CVAEMemRand
    cfg = dict(entry["config"], device="cpu")
    cfg["feat_dim"] = tuple(cfg["feat_dim"])
    old = torch.get_default_dtype()
    torch.set_default_dtype(getattr(torch, entry["default_dtype"]))
    try:
        model = cls(cfg)
        sd = model.state_dict()
        got = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()]
        assert got == entry["state_dict"]
        assert int(sum(p.numel() for p in model.parameters())) == entry["n_params"]
        want_cfg = dict(entry["mutated_config"])
        have = {k: (list(v) if isinstance(v, tuple) else (str(v) if isinstance(v, torch.device) else v))
                for k, v in model.config.items()}
        for k in ("device",):
            want_cfg.pop(k, None)
            have.pop(k, None)
        have = {k: v for k, v in have.items() if k != "hip_graphs"}
        assert have == want_cfg
    finally:
        torch.set_default_dtype(old)
