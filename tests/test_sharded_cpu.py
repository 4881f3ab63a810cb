"""The date-range sharding layer (vsv/sharded.py) with world_size 2 over gloo on CPU: shard
arithmetic, ragged all_gather in day order, and that a computation keyed by the GLOBAL day
index (as the Philox counters are) gives the same tensor for 1 and 2 ranks.  The compute
callback is a stand-in (no kernels run here); the GPU tests check the same invariance on
the real kernels within one process."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import vsv_oracle as vo


def _fake_compute(first_day, lo, hi, H=2):
    """(hi-lo, H, 3, 5, 5) 'quantiles' that depend only on the global day index -- drawn from the
    oracle's Philox stream keyed like the kernels' (seed; sample, day, step)."""
    z = vo.philox_normal(7, first_day + lo, hi - lo, 3, 0, H, 25)            # (days, 3, H, 25)
    return torch.from_numpy(np.ascontiguousarray(z.transpose(0, 2, 1, 3))).reshape(hi - lo, H, 3, 5, 5)


def _worker(rank, world, port, n_days, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vsv import sharded
    full = sharded.sharded_days(lambda lo, hi: _fake_compute(20, lo, hi), n_days)
    out_q.put((rank, full.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_range_partitions_days():
    from vsv.sharded import shard_range
    for n, w in ((4000, 8), (7, 2), (5, 8), (1, 1), (0, 4)):
        r = [shard_range(n, k, w) for k in range(w)]
        assert r[0][0] == 0 and r[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
        assert max(hi - lo for lo, hi in r) - min(hi - lo for lo, hi in r) <= 1


@pytest.mark.parametrize("n_days", [7, 8, 1])      # 1 day over 2 ranks: rank 1 owns an empty range
def test_two_ranks_gather_equals_single_process(n_days):
    from vsv import sharded
    want = sharded.sharded_days(lambda lo, hi: _fake_compute(20, lo, hi), n_days).numpy()     # no process group
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_days, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        assert got[r].shape == want.shape and np.array_equal(got[r], want)


class _FakeModel:
    @staticmethod
    def _draw_seed():
        return int(torch.randint(0, 2 ** 62, (1,)).item())


def _seed_worker(rank, world, port, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vsv import drivers
    torch.manual_seed(1000 + rank)                       # the ranks' generators disagree on purpose
    own = int(torch.randint(0, 2 ** 62, (1,)).item())
    torch.manual_seed(1000 + rank)
    out_q.put((rank, own, drivers._shared_seed(_FakeModel), drivers._shared_seed(_FakeModel)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_drivers_use_rank0_seed():
    """every rank must key Philox with the same seed, or the gathered product depends on the
    world size.  Ranks seeded differently still agree on rank 0's draw, call after call."""
    from vsv import drivers
    torch.manual_seed(1000)
    want = [drivers._shared_seed(_FakeModel), drivers._shared_seed(_FakeModel)]        # single process = rank 0's stream
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_seed_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {r: (own, a, b) for r, own, a, b in (q.get(timeout=120) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][0] != got[1][0]
    for r in range(2):
        assert [got[r][1], got[r][2]] == want
