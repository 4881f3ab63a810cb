"""bench.py's own launcher (`python bench.py --gpus N` with no torchrun around it) on a CPU-only host:
the parent must spawn N ranks before touching a GPU, refuse more ranks than GPUs unless told otherwise,
and refuse a launcher whose WORLD_SIZE disagrees with --gpus."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")


def _run(args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=300, env=e)


def test_more_ranks_than_gpus_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("host has >= 2 GPUs")
    r = _run(["--gpus", "2", "--dry-run"])
    assert r.returncode != 0
    assert "--gpus 2 but only" in r.stderr and "--allow-oversubscribe" in r.stderr


def test_parent_spawns_two_ranks():
    r = _run(["--gpus", "2", "--allow-oversubscribe", "--dry-run", "--master-port", "29641"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out == {"dry_run": True, "n_gpus": 2, "ranks": 2}


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "2", "--dry-run"], env={"RANK": "0", "WORLD_SIZE": "4", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "WORLD_SIZE=4" in r.stderr
