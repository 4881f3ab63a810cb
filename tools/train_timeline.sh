#!/bin/bash
# Kernel timeline of one steady-state train_step (c1, T = 11) on the GPU box: tools/train_timeline.sh <tag>
tag=${1:-tt}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $out/$tag
VSV_GRAPHS=0 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/$tag -o p -- python3 tools/train_profile.py > $out/$tag.log 2>&1 || exit 1
python3 tools/train_timeline.py $(ls $out/$tag/*kernel_trace.csv | head -1) > $out/${tag}_timeline.txt
