#!/usr/bin/env python3
"""One steady-state train_step as a kernel timeline, from a rocprofv3 --kernel-trace --output-format csv run of
tools/train_profile.py:  python tools/train_timeline.py <p_kernel_trace.csv> [steps=30] [warmup=5]
Prints launches per step, GPU-busy time per step and the ordered kernels of the median step."""
import csv, sys, collections
path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Stream_Id", 0) or 0)))
rows.sort()
# a step ends with the optimizer's last multi_tensor_apply launch: split on the loss kernel instead (one per step)
marks = [i for i, r in enumerate(rows) if "loss_grad_kernel" in r[2]]
import os
per_step = int(os.environ.get("MARKS_PER_STEP", "1"))          # (the multi-horizon step launches the loss kernel once per horizon)
marks = marks[::per_step][-steps:]
per = []
for a, b in zip(marks[:-1], marks[1:]):
    seg = rows[a:b]
    busy = sum(e - s for s, e, _, _ in seg)
    per.append((len(seg), busy / 1e3, (seg[-1][1] - seg[0][0]) / 1e3, a, b))
per.sort(key=lambda x: x[2])
n, busy, span, a, b = per[len(per) // 2]
print(f"steps analysed {len(per)}: launches/step {n}, sum of kernel durations {busy:.1f} us, span {span:.1f} us (median step)")
t0 = rows[a][0]
short = lambda k: (k.split("(")[0].replace("void ", "").replace("at::native::", "")[:70])
for s, e, k, st in rows[a:b]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  s{st:<3d} {short(k)}")
cnt = collections.Counter(short(k) for _, _, k, _ in rows[a:b])
print("--- launches by kernel")
for k, v in cnt.most_common():
    print(f"{v:4d}  {k}")
