"""HIP-graph replay of the training step's device work.

One train_step issues ~80 launches (the vsv_* kernels plus the small torch ops around them; ~250 before the
weight-gradient kernel); at batch 64 they are short, so issued one by one the step can be bound by the host.
GraphedLossGrads captures, once per input shape, everything between "parameters + batch" and
"loss terms + parameter gradients" into one HIP graph (torch.cuda.CUDAGraph: the vsv_* launches go to
torch's capturing stream like any other kernel, the side stream forks and joins inside the capture)
and replays it per step.  What varies between replays lives in device memory:
  * the batch                      -> copied into static input buffers before the replay,
  * the parameters                 -> packed into the kernels' blob INSIDE the graph,
  * the latent noise               -> torch.randn inside the graph (torch advances its Philox offset
                                      per replay; the reference draws torch.randn_like here, :199),
  * the nn.LSTM dropout mask seed  -> a device word filled by Tensor.random_() inside the graph and read
                                      by the kernels through vsv_train_save.drop_seed_dev,
  * kl_weight / re_feat_weight     -> a device pair the loss kernel reads (vsv_loss.weights_dev) and the 5 x 3
                                      matrix that mixes the loss terms: a KL-annealing schedule rewrites 17
                                      numbers per change instead of capturing a graph per value.
Clipping and the optimizer stay outside (they belong to the caller's optimizer object).
The multi-horizon step (train_step_multihorizon: the horizons as row groups of one forward / backward) is captured the
same way; its loss weights stay kernel arguments (part of the cache key).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Optional

import torch


class GraphedLossGrads:
    """Captured (pack weights -> forward -> loss -> backward -> parameter gradients) for one shape."""

    def __init__(self, te, named_params, B: int, T: int, has_ex: bool, horizon: int, loss_cfg: dict,
                 dropout_p: float, inject_eps: bool = False, horizons=None, weights=None):
        """horizons / weights: the multi-horizon step (TrainEngine.loss_and_grads_multihorizon) instead of the
        single-horizon one; `horizon` is then unused."""
        self.te, self.eng = te, te.eng
        self.params = [p for _, p in named_params]
        self.named = list(named_params)
        self.horizon, self.loss_cfg, self.p = int(horizon), dict(loss_cfg), float(dropout_p)
        self.horizons = None if horizons is None else [int(h) for h in horizons]
        self.weights = None if weights is None else [float(w) for w in weights]
        dev = te.dev
        self.surf = torch.zeros(B, T, 5, 5, dtype=torch.float32, device=dev)
        self.ex = torch.zeros(B, T, te.X, dtype=torch.float32, device=dev) if has_ex else None
        self.shape = (B, T)
        # parity tests: the posterior noise comes from a static buffer the caller fills, not from torch.randn
        G = 1 if self.horizons is None else len(self.horizons)
        eshape = (B, T, te.L) if self.horizons is None else (G, B, T, te.L)
        self.eps_in = torch.zeros(eshape, dtype=torch.float32, device=dev) if inject_eps else None
        self.wts = torch.zeros(2, dtype=torch.float32, device=dev)
        self.mix = torch.zeros(5, 3, dtype=torch.float64, device=dev)
        self._wkey = None
        self.set_loss_weights(loss_cfg)
        if self.horizons is None:                # (the multi-horizon step reads the two weights from its arguments)
            self.loss_cfg["weights_dev"], self.loss_cfg["mix_dev"] = self.wts, self.mix
        cur = torch.cuda.current_stream(dev)
        warm = torch.cuda.Stream(dev)
        warm.wait_stream(cur)
        with torch.cuda.stream(warm):           # eager passes: index maps, hipBLASLt workspaces, allocator pools
            for _ in range(2):
                self._body()
        cur.wait_stream(warm)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.loss_vec, self.grads, self.packed = self._body()
        self.flat = te.last_flat                 # the gradients are views of this static buffer
        self.partials = te.last_partials         # (its sums of squares per 256 elements, for the clip)

    def set_loss_weights(self, loss_cfg: dict) -> None:
        """The two loss weights as the replays will read them (no-op while they are unchanged)."""
        exw = float(loss_cfg["ex_weight"]) if self.ex is not None else 0.0
        klw = float(loss_cfg["kl_weight"])
        if self._wkey == (exw, klw):
            return
        self._wkey = (exw, klw)
        self.wts.copy_(torch.tensor([float(loss_cfg["ex_weight"]), klw], dtype=torch.float32))
        # rows: loss, re_surface, re_ex_feats, reconstruction_loss, kl_loss from (re_surface, re_ex_feats, kl)
        self.mix.copy_(torch.tensor([[1.0, exw, klw], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [1.0, exw, 0.0],
                                     [0.0, 0.0, 1.0]], dtype=torch.float64))

    def _body(self):
        te, eng = self.te, self.eng
        B, T = self.shape
        with torch.no_grad():
            eng.set_weights_from_params(self.named)
            eps = self.eps_in
            if eps is None and self.horizons is None:
                eps = torch.randn(B, T, te.L, dtype=torch.float32, device=te.dev)
            seed = 0
            if self.p > 0:
                seed = torch.empty(1, dtype=torch.int64, device=te.dev).random_()
            if self.horizons is not None:
                G = len(self.horizons)
                if eps is None:                  # fresh noise per horizon (the reference's randn_like per forward, :1218)
                    eps = torch.randn(G, B, T, te.L, dtype=torch.float32, device=te.dev)
                losses, gp = te.loss_and_grads_multihorizon(self.surf, self.ex, self.horizons, self.weights, self.loss_cfg,
                                                            eps=[eps[g] for g in range(G)], dropout_p=self.p,
                                                            dropout_seed=seed)
                grads = te.param_grads(gp)
                vec = torch.cat([torch.stack([losses["loss"], losses["reconstruction_loss"], losses["kl_loss"]]),
                                 losses["horizon_losses"]]).to(torch.float64)
                return vec, grads, eng.packed
            losses, gp, outs = te.loss_and_grads(self.surf, self.ex, self.horizon, self.loss_cfg, eps=eps,
                                                 dropout_p=self.p, dropout_seed=seed)
            grads = te.param_grads(gp)
            vec = outs["loss_vec"]               # (loss, re_surface, re_ex_feats, reconstruction_loss, kl_loss: no gather)
        return vec, grads, eng.packed

    def run(self, surf: torch.Tensor, ex: Optional[torch.Tensor], eps: Optional[torch.Tensor] = None):
        """-> (losses dict of 0-d float64 tensors (own memory), parameter gradients in parameter order:
        views of the graph's static output, valid until this object's next run)."""
        if self.eps_in is not None:
            self.eps_in.copy_(eps.to(self.eps_in.device, torch.float32), non_blocking=True)
        self.surf.copy_(surf, non_blocking=True)
        if self.ex is not None:
            self.ex.copy_(ex, non_blocking=True)
        self.graph.replay()
        self.eng.packed = self.packed            # what the blob now holds: the parameters as of this replay
        self.eng._fold_stale = True              # (generation's folded head section is rebuilt on demand)
        v = self.loss_vec.clone()
        if self.horizons is not None:
            losses = {"loss": v[0], "reconstruction_loss": v[1], "kl_loss": v[2], "horizon_losses": v[3:]}
        else:
            losses = {"loss": v[0], "re_surface": v[1], "re_ex_feats": v[2], "reconstruction_loss": v[3], "kl_loss": v[4]}
        self.te.last_flat, self.te.last_partials = self.flat, self.partials
        return losses, self.grads


class GraphCache:
    """Shape-keyed cache of GraphedLossGrads (least recently used out beyond `capacity`)."""

    def __init__(self, capacity: int = 64):
        self.capacity = capacity
        self._d: "OrderedDict[tuple, GraphedLossGrads]" = OrderedDict()

    def get(self, te, named_params, surf, ex, horizon, loss_cfg: Dict, dropout_p: float,
            inject_eps: bool = False, horizons=None, weights=None) -> GraphedLossGrads:
        B, T = surf.shape[:2]
        mh = horizons is not None
        key = (B, T, ex is not None, int(horizon), float(dropout_p), bool(inject_eps),
               tuple(sorted((k, tuple(v) if isinstance(v, (list, tuple)) else v) for k, v in loss_cfg.items()
                            if mh or k not in ("ex_weight", "kl_weight"))),
               None if not mh else (tuple(int(h) for h in horizons), tuple(float(w) for w in weights)),
               tuple(p.data_ptr() for _, p in named_params), id(te))
        g = self._d.get(key)
        if g is None:
            g = GraphedLossGrads(te, named_params, B, T, ex is not None, horizon, loss_cfg, dropout_p, inject_eps,
                                 horizons, weights)
            self._d[key] = g
            while len(self._d) > self.capacity:
                self._d.popitem(last=False)
        else:
            self._d.move_to_end(key)
            g.set_loss_weights(loss_cfg)
        return g
