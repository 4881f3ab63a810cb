#!/bin/bash
# SQ counters of the fused decode kernel, one rocprofv3 --pmc pass per counter group (run on the GPU box):
#   tools/pmc_decode.sh <tag>  ->  gpurun_out/<tag>_pmc_sq.txt  (per-dispatch averages by kernel)
tag=${1:-rX}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out
args="bench.py --steps 1 --warmup 1 --cpu-budget 0 --train-steps 0 --others 0"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rm -rf $out/pmcsq_${tag}_$i
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pmcsq_${tag}_$i -o p -- python3 $args > $out/pmcsq_${tag}_$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/pmcsq_${tag}_$i.log; exit 1; }
done
python3 - "$out" "$tag" $i <<'PY'
import csv, sys, collections
out, tag, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
acc = collections.defaultdict(lambda: [0, 0.0])
for i in range(1, n + 1):
    with open(f"{out}/pmcsq_{tag}_{i}/p_counter_collection.csv") as f:
        for r in csv.DictReader(f):
            k = (r["Kernel_Name"][:90], r["Counter_Name"])
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
with open(f"{out}/{tag}_pmc_sq.txt", "w") as g:
    g.write("# per-dispatch average of each counter, by kernel (rocprofv3 --pmc, one pass per group)\n")
    for (k, c), (cnt, v) in sorted(acc.items()):
        g.write(f"{cnt:5d} {v/cnt:18.1f} {c:30s} {k}\n")
PY
echo done
