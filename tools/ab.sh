#!/bin/bash
# A/B of library builds on the default bench: tools/ab.sh v00 v01 ...  (lib/libvsv_hip_<tag>.so)
cd "$GRAFT_REPO_ROOT" || exit 1
for round in 1 2; do
for v in "$@"; do
  VSV_LIB=$GRAFT_REPO_ROOT/vol-surface-vae-pub_amd/lib/libvsv_hip_$v.so timeout -k 10 200 python -u bench.py --steps 5 --warmup 2 --cpu-budget 0 --train-steps 0 --others 0 > gpurun_out/ab_$v.log 2>&1 || { tail -5 gpurun_out/ab_$v.log; exit 1; }
  python -c "import json;d=json.loads(open('gpurun_out/ab_$v.log').read().strip().splitlines()[-1]);print('$v','step_ms=%.2f'%d['ms_per_step'],'decode_ms=%.3f'%d['roofline']['avg_launch_ms'])"
done
done
