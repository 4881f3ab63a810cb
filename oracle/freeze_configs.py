#!/usr/bin/env python3
"""Freeze the literal model-config dicts the REFERENCE's training drivers build into tests/golden/configs.json.

Runs only in the build container (needs /root/reference).  SURVEY.md 2: "their config dicts are the test inputs".
For every driver below the `model_config = {...}` (or `config = {...}`) dict literal is taken from the driver's
source with `ast` and evaluated against the reference's own config class (imported from /root/reference/config), so
the frozen values are exactly what the driver would pass to CVAEMemRand(...) -- dropout 0.2 - 0.3, [5, 1, 5] quantile
weights, interaction_layers 2 of the v3 trainer, the lot.  The reference's class is then constructed from each dict
(under the default dtype the driver sets) and its MUTATED config (check_input adds padding / deconv_output_padding /
num_quantiles, :545-558, :969-1020) and its state_dict's names, shapes and dtypes are stored next to it: the build's
class must reproduce them (tests/test_model_cpu.py), and run a train_step and a generation from each on the GPU
(tests/test_gpu_parity_gaps.py).

Usage:  python oracle/freeze_configs.py        (writes tests/golden/configs.json)
"""
import ast
import json
import os
import sys
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REF)

import torch  # noqa: E402

from vae.cvae_with_mem_randomized import CVAEMemRand  # noqa: E402  (the reference)
from vae.cvae_conditional_prior import CVAEMemRandConditionalPrior  # noqa: E402
import vae as _ref_vae  # noqa: E402

assert os.path.abspath(_ref_vae.__path__[0]).startswith(REF), "configs must come from the reference"

DRIVERS = [
    # (name, driver file, dict variable, namespace builder)
    ("context20_backfill", "experiments/backfill/context20/train_backfill_model.py", "model_config",
     lambda: {"BackfillConfig": __import__("config.backfill_config", fromlist=["BackfillConfig"]).BackfillConfig}),
    ("context60_latent12_v2", "experiments/backfill/context60/train_backfill_context60_latent12_v2.py", "model_config",
     lambda: {"cfg": __import__("config.backfill_context60_config_latent12_v2",
                                fromlist=["BackfillContext60ConfigLatent12V2"]).BackfillContext60ConfigLatent12V2}),
    ("context60_latent12", "experiments/backfill/context60/train_backfill_context60_latent12.py", "model_config", None),
    ("context60_v3_conditional_prior",
     "experiments/backfill/context60_v3_fixed/train_backfill_context60_latent12_v3_conditional_prior.py", "model_config",
     lambda: {"cfg": __import__("config.backfill_context60_config_latent12_v3_conditional_prior",
                                fromlist=["BackfillContext60ConfigLatent12V3ConditionalPrior"]
                                ).BackfillContext60ConfigLatent12V3ConditionalPrior}),
]


def find_dict(tree, var):
    for node in ast.walk(tree):
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name) \
                and node.targets[0].id == var and isinstance(node.value, ast.Dict):
            return node
    raise KeyError(var)


def default_dtype_of(src):
    return "float64" if "set_default_dtype(torch.float64)" in src else "float32"


def cfg_class_from_imports(tree):
    """namespace {'cfg': <the one config class the driver imports from config.*>} for drivers whose dict reads cfg.X"""
    for node in ast.walk(tree):
        if isinstance(node, ast.ImportFrom) and node.module and node.module.startswith("config."):
            mod = __import__(node.module, fromlist=[a.name for a in node.names])
            return {"cfg": getattr(mod, node.names[0].name), node.names[0].name: getattr(mod, node.names[0].name)}
    raise KeyError("no config import")


def jsonable(v):
    if isinstance(v, tuple):
        return [jsonable(x) for x in v]
    if isinstance(v, list):
        return [jsonable(x) for x in v]
    if isinstance(v, dict):
        return {k: jsonable(x) for k, x in v.items()}
    if isinstance(v, torch.device):
        return str(v)
    return v


def reference_contract(cfg, dtype, cls=CVAEMemRand):
    torch.set_default_dtype(getattr(torch, dtype))
    try:
        model = cls(dict(cfg))
        sd = model.state_dict()
        return {"mutated_config": jsonable({k: v for k, v in model.config.items()}),
                "state_dict": [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()],
                "n_params": int(sum(p.numel() for p in model.parameters()))}
    finally:
        torch.set_default_dtype(torch.float32)


def main():
    out = []
    for name, rel, var, ns_fn in DRIVERS:
        path = os.path.join(REF, rel)
        src = open(path).read()
        tree = ast.parse(src)
        node = find_dict(tree, var)
        ns = {"torch": torch, "device": torch.device("cpu")}          # (the drivers' `device` on a box without a GPU)
        ns.update(ns_fn() if ns_fn else cfg_class_from_imports(tree))
        cfg = eval(compile(ast.Expression(node.value), path, "eval"), ns)
        dtype = default_dtype_of(src)
        cond = "CVAEMemRandConditionalPrior(model_config)" in src
        entry = {"name": name, "driver": f"{rel}:{node.lineno}", "default_dtype": dtype, "config": jsonable(cfg),
                 "model_class": "CVAEMemRandConditionalPrior" if cond else "CVAEMemRand"}
        entry.update(reference_contract(cfg, dtype, CVAEMemRandConditionalPrior if cond else CVAEMemRand))
        out.append(entry)
    # train_quantile_models.py:115-141: one dict per model variant, built from the script's own parameter grid
    rel = "train_quantile_models.py"
    src = open(os.path.join(REF, rel)).read()
    tree = ast.parse(src)
    grid = eval(compile(ast.Expression(find_dict(tree, "conv_param_grid").value), rel, "eval"), {})
    node = find_dict(tree, "config")
    for variant in ("no_ex", "ex_no_loss", "ex_loss"):
        ns = {"conv_param_grid": grid, "param": {"model_name": variant}, "ex_data": types.SimpleNamespace(shape=(0, 3))}
        cfg = eval(compile(ast.Expression(node.value), rel, "eval"), ns)
        dtype = default_dtype_of(src)
        entry = {"name": f"quantile_{variant}", "driver": f"{rel}:{node.lineno}", "default_dtype": dtype, "config": jsonable(cfg),
                 "model_class": "CVAEMemRand"}
        entry.update(reference_contract(dict(cfg, device="cpu"), dtype))
        out.append(entry)
    dst = os.path.join(REPO, "tests", "golden", "configs.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for e in out:
        print(e["name"], e["default_dtype"], "latent", e["config"]["latent_dim"], "dropout", e["config"]["mem_dropout"],
              "interaction", e["config"].get("interaction_layers"), "params", e["n_params"])
    print("wrote", dst, os.path.getsize(dst) // 1024, "KiB")


if __name__ == "__main__":
    main()
