"""Date-range sharding of the Monte-Carlo generation paths over the GPUs of a node:
one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI), every rank
owns a contiguous range of days, and the only exchange is one all_gather of the REDUCED
quantile tensor (days, H, nq, 5, 5) -- raw samples never leave their GPU.  Days are
independent and the Philox counters are keyed by the global day index, so the result
is bit-identical for any world size.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_days: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous day range [lo, hi) of `rank`: the first n_days % world ranks get one extra day."""
    q, r = divmod(n_days, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_days(local: torch.Tensor, n_days: int, group=None) -> torch.Tensor:
    """all_gather of per-rank day blocks (ragged in dim 0) into the full (n_days, ...) tensor,
    in day order, on every rank.  Blocks are padded to the largest shard for the collective."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        assert local.shape[0] == n_days
        return local
    world = dist.get_world_size(group)
    sizes = [shard_range(n_days, r, world) for r in range(world)]
    big = max(hi - lo for lo, hi in sizes)
    dev = local.device
    if local.is_cuda and dist.get_backend(group) != "nccl":
        local = local.cpu()                    # gloo rehearsals: the collective runs on host buffers
    pad = local.new_zeros((big,) + tuple(local.shape[1:]))
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad.contiguous(), group=group)
    return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], dim=0).to(dev)


def sharded_days(compute: Callable[[int, int], torch.Tensor], n_days: int, group=None) -> torch.Tensor:
    """Runs compute(lo, hi) -> (hi-lo, ...) on this rank's day range and gathers all ranges."""
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    lo, hi = shard_range(n_days, rank, world)
    return gather_days(compute(lo, hi), n_days, group)


def prior_mc_sharded(eng, vol, ex, first_day: int, n_days: int, ctx_len: int, horizon: int, n_samples: int,
                     seed: int = 0, group=None, **kw) -> torch.Tensor:
    """paths.prior_mc_windows over days [first_day, first_day + n_days) split across the ranks.
    Every rank holds the whole (small) calendar; returns the full (n_days, horizon, nq, ...) tensor."""
    from . import paths

    def compute(lo, hi):
        return paths.prior_mc_windows(eng, vol, ex, first_day + lo, hi - lo, ctx_len, horizon, n_samples,
                                      seed=seed, day_index0=first_day + lo, **kw)
    return sharded_days(compute, n_days, group)


def ar_mc_sharded(eng, vol, ex, first_day: int, n_days: int, ctx_len: int, horizon: int, n_samples: int,
                  seed: int = 0, group=None, **kw) -> torch.Tensor:
    from . import paths

    def compute(lo, hi):
        return paths.ar_mc_windows(eng, vol, ex, first_day + lo, hi - lo, ctx_len, horizon, n_samples,
                                   seed=seed, day_index0=first_day + lo, **kw)
    return sharded_days(compute, n_days, group)


def oracle_mc_sharded(eng, vol, ex, first_day: int, n_days: int, ctx_len: int, horizon: int, n_samples: int,
                      seed: int = 0, group=None, **kw) -> torch.Tensor:
    from . import paths

    def compute(lo, hi):
        return paths.oracle_mc_windows(eng, vol, ex, first_day + lo, hi - lo, ctx_len, horizon, n_samples,
                                       seed=seed, day_index0=first_day + lo, **kw)
    return sharded_days(compute, n_days, group)
