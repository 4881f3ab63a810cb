#!/usr/bin/env python3
"""A/B of the headline decode launch across library builds: runs bench.py's headline leg only, once per library
named on the command line (each in its own process: VSV_LIB is read at import), and prints the decode launch time
(HIP events) and the whole step.   python tools/mc1_ab.py base=lib/libvsv_hip_base.so new=lib/libvsv_hip.so"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for arg in sys.argv[1:]:
    name, path = arg.split("=")
    env = dict(os.environ, VSV_LIB=os.path.join(REPO, "vol-surface-vae-pub_amd", path))
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "10", "--warmup", "3", "--cpu-budget", "0",
                        "--train-steps", "0", "--others", "0"], env=env, capture_output=True, text=True, timeout=300)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(name, "FAILED", r.stderr[-2000:])
        sys.exit(1)
    j = json.loads(line[-1])
    rf = j["roofline"]
    print(f"{name:8s} step {j['ms_per_step']:.3f} ms  decode launch {rf.get('avg_launch_ms', rf.get('launch_ms'))} ms  frac {rf['frac']:.4f}", flush=True)
