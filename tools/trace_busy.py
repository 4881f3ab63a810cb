#!/usr/bin/env python3
"""Busy / idle accounting of a rocprofv3 --kernel-trace CSV: wall span, union of kernel intervals (GPU busy), idle
gaps by size, kernel time by name -- over the LAST fraction of the trace (the warm passes).
   python tools/trace_busy.py kernel_trace.csv [tail_fraction]"""
import collections
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
t_lo = rows[0][0] + (rows[-1][1] - rows[0][0]) * (1.0 - frac)
rows = [r for r in rows if r[0] >= t_lo]
span = rows[-1][1] - rows[0][0]
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
gaps = []
for s, e, _ in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"kernels {len(rows)}  span {span / 1e6:.2f} ms  GPU busy (union) {busy / 1e6:.2f} ms = {100.0 * busy / span:.1f} %  idle {(span - busy) / 1e6:.2f} ms in {len(gaps)} gaps")
for lo, hi in ((0, 5), (5, 20), (20, 100), (100, 1000), (1000, 1e9)):
    g = [x for x in gaps if lo * 1e3 <= x < hi * 1e3]
    print(f"  gaps {lo:>5} - {hi:<6g} us: {len(g):5d}  total {sum(g) / 1e6:8.2f} ms")
by = collections.Counter()
cnt = collections.Counter()
for s, e, k in rows:
    by[k[:60]] += e - s
    cnt[k[:60]] += 1
for k, v in by.most_common(12):
    print(f"  {v / 1e6:9.2f} ms  {cnt[k]:5d} x  {k}")
