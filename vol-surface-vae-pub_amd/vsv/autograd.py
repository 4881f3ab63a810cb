"""torch.autograd.Function over the HIP forward / backward kernels.

The reference's forward() (vae/cvae_with_mem_randomized.py:1022-1074) returns tensors that carry an
autograd graph, so callers may write their own loss and call .backward() instead of train_step().
CVAEForward gives the HIP path the same property: forward = TrainEngine.forward_saved (the three
vsv_train_* launches, activations kept on the device), backward = TrainEngine.backward_saved (decoder
head, BPTT, conv backward kernels) from whatever gradients autograd hands back for the five outputs,
scattered onto the nn.Parameters through the packing index maps.
"""
from __future__ import annotations

import torch


class CVAEForward(torch.autograd.Function):
    """apply(te, surf, ex | None, horizon, eps | None, seed, dropout_p, dropout_seed, *params)
    -> (recon (B,H,Q,5,5), ex_recon (B,H,X) or an empty tensor, z_mean, z_log_var, z), float32.
    `params` are the model's parameters in the order the TrainEngine was built with; the engine's
    packed blob must have been refreshed from them (CVAEMemRand._engine does)."""

    @staticmethod
    def forward(ctx, te, surf, ex, horizon, eps, seed, dropout_p, dropout_seed, *params):
        c = te.forward_saved(surf, ex, int(horizon), eps, int(seed), float(dropout_p), int(dropout_seed))
        ctx.te, ctx.c = te, c
        ctx.want_dsurf = bool(ctx.needs_input_grad[1])
        ctx.surf_meta = (surf.shape, surf.dtype, surf.device)
        ctx.param_meta = [(p.dtype, p.device) for p in params]
        o = c.outputs
        e = o["ex"] if o["ex"] is not None else torch.empty(0, device=te.dev)
        if o["ex"] is None:
            ctx.mark_non_differentiable(e)
        return o["surf"], e, o["z_mean"], o["z_log_var"], o["z"]

    @staticmethod
    def backward(ctx, d_s, d_e, d_zm, d_zl, d_z):
        te, c = ctx.te, ctx.c
        if c is None:
            raise RuntimeError("CVAEForward: backward through the HIP path a second time (the saved "
                               "activations are freed after the first backward, as with retain_graph=False)")
        ctx.c = None
        if d_s is None:
            d_s = torch.zeros_like(c.outputs["surf"])
        if c.outputs["ex"] is None:
            d_e = None
        gp = te.backward_saved(c, d_s, d_e, d_zm, d_zl, d_z, ctx.want_dsurf)
        grads = []
        for g, (dt, dev), need in zip(te.param_grads(gp), ctx.param_meta, ctx.needs_input_grad[8:]):
            grads.append(g.to(device=dev, dtype=dt) if need else None)
        d_surf = None
        if ctx.want_dsurf:
            shape, dt, dev = ctx.surf_meta
            d_surf = c.outputs["d_surface"].reshape(shape).to(device=dev, dtype=dt)
        return (None, d_surf, None, None, None, None, None, None, *grads)
