"""CPU check of vsv_pack.cpp and of every index map the kernels use, through a
numpy model of the kernels' dataflow on the packed blob (tests/pack_emulator.py),
against the oracle.  No GPU, no compute call into the library."""
import numpy as np
import pytest

from oracle import vsv_oracle as vo
import pack_emulator as pe


CASES = [
    dict(L=5, X=0, NL=2, T=4),
    dict(L=5, X=3, NL=2, T=3),
    dict(L=12, X=3, NL=2, T=3),
    dict(L=5, X=0, NL=1, T=3),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"L{c['L']}X{c['X']}NL{c['NL']}")
def test_packed_dataflow_matches_oracle(case):
    cfg = vo.make_config(case["L"], case["X"], mem_layers=case["NL"])
    W = vo.init_weights(cfg, 42, 2.0, np.float64)
    pm = pe.PackedModel(cfg, W)       # blob is float32-rounded weights
    W32 = {k: v.astype(np.float32).astype(np.float64) for k, v in W.items()}
    rng = np.random.default_rng(0)
    T, L, X = case["T"], case["L"], case["X"]
    surf = 0.2 + 0.1 * rng.standard_normal((16, T, 5, 5))
    ex = 0.01 * rng.standard_normal((16, T, X)) if X else None

    zm, zl = vo.encoder_forward(W32, cfg, surf, ex)
    got = pe.emu_encode(pm, 0, surf.reshape(16, T, 25), ex)
    assert np.max(np.abs(got[..., :L] - zm)) < 2e-7
    assert np.max(np.abs(got[..., L:] - zl)) < 2e-7
    ctx = vo.ctx_encoder_forward(W32, cfg, surf, ex)
    got = pe.emu_encode(pm, 1, surf.reshape(16, T, 25), ex)
    assert np.max(np.abs(got - ctx)) < 2e-7

    dec_in = rng.standard_normal((16, T, 2 * L))
    s, e, finals = vo.decoder_forward(W32, cfg, dec_in)
    gs, ge, st = pe.emu_decode(pm, dec_in)
    assert np.max(np.abs(gs.reshape(s.shape) - s)) < 2e-7
    if X:
        assert np.max(np.abs(ge - e)) < 2e-7
    Hd = pm.Hd
    for l, (h, c) in enumerate(finals):
        assert np.max(np.abs(st[:, l, 0, :Hd] - h)) < 2e-7
        assert np.max(np.abs(st[:, l, 1, :Hd] - c)) < 2e-7
    # resume from a saved state
    s2, e2, _ = vo.decoder_forward(W32, cfg, dec_in[:, :2], [(h, c) for h, c in finals])
    gs2, _, _ = pe.emu_decode(pm, dec_in[:, :2], state=st)
    assert np.max(np.abs(gs2.reshape(s2.shape) - s2)) < 2e-7
