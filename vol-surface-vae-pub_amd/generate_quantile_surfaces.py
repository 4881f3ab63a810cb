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


def _safe_globals():
    """What a checkpoint written by vae/base.py save_weights holds besides tensors: the numpy RNG state (an ndarray) and
    possibly a torch.device in model_config.  Allow-listed for the weights-only unpickler; nothing else is."""
    out = [torch.device, np.ndarray, np.dtype]
    try:
        from numpy.core.multiarray import _reconstruct          # numpy < 2 pickles
        out.append(_reconstruct)
    except Exception:
        pass
    try:
        from numpy._core.multiarray import _reconstruct as _reconstruct2
        out.append(_reconstruct2)
    except Exception:
        pass
    out += [type(np.dtype(t)) for t in ("uint32", "int64", "float64")]
    return out


def _load_checkpoint(path, allow_full_pickle=None):
    """The caller's own checkpoint ({"model", "model_config", ...}, vae/base.py save_weights), loaded with the
    weights-only unpickler (which executes nothing from the file) and an allow-list for the few non-tensor types such
    a checkpoint holds.  The full pickle loader -- the reference's own `weights_only=False` -- runs code from the file,
    so it is used only when the caller asks for it (allow_full_pickle=True or VSV_ALLOW_FULL_PICKLE=1), only after the
    safe loader has refused to unpickle the file (I/O errors are not retried), and with a warning."""
    import os
    import pickle
    import warnings
    if allow_full_pickle is None:
        allow_full_pickle = os.environ.get("VSV_ALLOW_FULL_PICKLE", "") == "1"
    try:
        with torch.serialization.safe_globals(_safe_globals()):
            return torch.load(path, weights_only=True)
    except pickle.UnpicklingError as exc:
        if not allow_full_pickle:
            raise RuntimeError(f"{path}: the weights-only loader refused this checkpoint ({exc}); if the file is your own "
                               "and trusted, pass allow_full_pickle=True or set VSV_ALLOW_FULL_PICKLE=1") from exc
        warnings.warn(f"{path}: loading with the full pickle loader (executes code from the file) at the caller's request")
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
