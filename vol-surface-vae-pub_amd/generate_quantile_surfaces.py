"""The reference's generate_quantile_surfaces.py (:1-143) on the HIP path: for the three quantile-regression
checkpoints (no_ex / ex_no_loss / ex_loss) the stochastic product {name}_quantile_gen5.npz and the z = 0
product {name}_quantile_mle_gen5.npz.  Unlike the reference script nothing runs at import: call main()
(python generate_quantile_surfaces.py [data.npz] [model_dir]); the per-model work is
vsv.drivers.generate_quantile_products, one batched device pass per product."""
import sys

import numpy as np
import torch

from vsv.drivers import generate_quantile_products  # noqa: F401

MODELS = (("no_ex", False), ("ex_no_loss", False), ("ex_loss", True))      # (name, check_ex_feats), reference :45-64


def _load_checkpoint(path):
    """The caller's own checkpoint ({"model", "model_config", ...}, vae/base.py save_weights): tensors-only load
    first; checkpoints that pickle other objects (a torch.device in model_config, RNG states) need the full load."""
    try:
        return torch.load(path, weights_only=True)
    except Exception:
        return torch.load(path, weights_only=False)


def main(data_path="data/vol_surface_with_ret.npz", model_dir="test_spx/quantile_regression", ctx_len=5, start_day=5000):
    from vae.cvae_with_mem_randomized import CVAEMemRand
    from vae.utils import set_seeds
    set_seeds(0)
    torch.set_default_dtype(torch.float64)
    data = np.load(data_path)
    vol = data["surface"]
    ex = np.concatenate([data["ret"][..., np.newaxis], data["skews"][..., np.newaxis], data["slopes"][..., np.newaxis]], axis=-1)
    num_days = vol.shape[0] - start_day
    written = []
    for name, check_ex in MODELS:
        model_data = _load_checkpoint(f"{model_dir}/{name}.pt")
        written += generate_quantile_products(model_data, ex, vol, start_day, num_days, ctx_len, model_dir, name,
                                              check_ex_feats=check_ex, model_type=CVAEMemRand)
    return written


if __name__ == "__main__":
    for f in main(*sys.argv[1:3]):
        print(f)
