cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out
for mode in 0 1; do
  rm -rf $out/ptl_$mode
  PREFIX_VARIANTS=$mode:$([ $mode = 0 ] && echo 3 || echo 1) timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $out/ptl_$mode -o p -- python3 tools/prefix_time.py ${DAYS:-4000} > $out/ptl_$mode.log 2>&1 || exit 1
  python3 tools/trace_tail.py $(ls $out/ptl_$mode/*kernel_trace.csv | head -1) 40 > $out/ptl_$mode.txt
done
cat $out/ptl_0.log $out/ptl_1.log | grep days
