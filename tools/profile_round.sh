#!/bin/bash
# Evidence for one round, run on the GPU box: tools/profile_round.sh <tag>
#   gpurun_out/<tag>_bench.json          default bench line (with cpu_baseline + training leg)
#   gpurun_out/<tag>_kernel_stats.txt    rocprofv3 --kernel-trace --stats of the same command (no CPU leg)
#   gpurun_out/<tag>_pmc_fetch|write.csv FETCH_SIZE / WRITE_SIZE of the fused decode kernel, separate passes
# Copy what is to be judged into profiles/.
tag=${1:-rX}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out
set -e
timeout -k 10 300 python3 -u bench.py > $out/${tag}_bench.log 2>&1
tail -1 $out/${tag}_bench.log > $out/${tag}_bench.json
rm -rf $out/prof_$tag $out/pmcf_$tag $out/pmcw_$tag
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/prof_$tag -o bench -- python3 bench.py --steps 5 --warmup 2 --cpu-budget 0 --train-steps 5 --others 0 > $out/prof_$tag.log 2>&1
python3 tools/rocpd_stats.py $(ls $out/prof_$tag/*.db | head -1) > $out/${tag}_kernel_stats.txt
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmcf_$tag -o p -- python3 bench.py --steps 1 --warmup 1 --cpu-budget 0 --train-steps 0 --others 0 > $out/pmcf_$tag.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmcw_$tag -o p -- python3 bench.py --steps 1 --warmup 1 --cpu-budget 0 --train-steps 0 --others 0 > $out/pmcw_$tag.log 2>&1
python3 - "$out" "$tag" <<'EOF'
import csv, sys, collections
out, tag = sys.argv[1:3]
for kind in ("pmcf", "pmcw"):
    acc = collections.defaultdict(lambda: [0, 0.0])
    with open(f"{out}/{kind}_{tag}/p_counter_collection.csv") as f:
        for r in csv.DictReader(f):
            k = (r["Kernel_Name"][:100], r["Counter_Name"])
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
    with open(f"{out}/{tag}_{kind}.txt", "w") as g:
        g.write("# per-dispatch average of the counter (rocprofv3 units: KB), by kernel\n")
        for (k, c), (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            g.write(f"{n:6d} {v/n:16.1f} {c:12s} {k}\n")
EOF
echo done
