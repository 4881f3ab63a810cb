import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "vol-surface-vae-pub_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def golden_names(prefix):
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith(".npz"))


def cfg_of(g):
    """Rebuild the model config a golden was generated with."""
    from oracle import vsv_oracle as vo
    cfg = vo.make_config(int(g["cfg_latent_dim"]), int(g["cfg_ex_feats_dim"]),
                         mem_layers=int(g["cfg_mem_layers"]))
    w = [float(x) for x in g["cfg_quantile_loss_weights"]]
    if w != [1.0, 1.0, 1.0]:
        cfg["quantile_loss_weights"] = w
    if "cfg_interaction_layers" in g and int(g["cfg_interaction_layers"]) > 0:
        cfg["interaction_layers"] = int(g["cfg_interaction_layers"])
    if "cfg_mem_type" in g:
        cfg["mem_type"] = str(g["cfg_mem_type"])
    if "cfg_compress_context" in g:
        cfg["compress_context"] = bool(g["cfg_compress_context"])
    if "cfg_use_dense_surface" in g and bool(g["cfg_use_dense_surface"]):
        cfg["use_dense_surface"] = True
        cfg["surface_hidden"] = [int(x) for x in g["cfg_surface_hidden"]]
        cfg["ctx_surface_hidden"] = [int(x) for x in g["cfg_ctx_surface_hidden"]]
    for k in ("ex_loss_on_ret_only", "ex_feats_loss_type", "kl_weight", "re_feat_weight"):
        if "cfg_" + k in g:
            v = g["cfg_" + k]
            cfg[k] = v.item() if v.dtype.kind != "U" else str(v)
    return cfg


# ---- achieved-error record: the parity tests assert a tolerance; the margin under it goes on record ----
_MARGINS = {}


def record_margin(value: float, what: str = "max|err|"):
    """Called by the tests' err() helpers: keeps the largest error every test observed."""
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0].split("::", 1)[-1]
    key = (test, what)
    if value > _MARGINS.get(key, -1.0):
        _MARGINS[key] = float(value)


def pytest_terminal_summary(terminalreporter):
    if not _MARGINS:
        return
    import json
    tr = terminalreporter
    tr.write_sep("-", "achieved parity errors (largest per test)")
    rows = sorted(_MARGINS.items(), key=lambda kv: -kv[1])
    for (test, what), v in rows[:40]:
        tr.write_line(f"{v:10.3e}  {what:10s} {test}")
    out_dir = os.path.join(REPO, "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "parity_margins.json"), "w") as f:
            json.dump([{"test": t, "what": w, "err": v} for (t, w), v in rows], f, indent=0)
    except OSError:
        pass
