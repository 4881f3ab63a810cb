#!/usr/bin/env python3
"""Last N kernels of a rocprofv3 kernel trace CSV as a timeline: python tools/trace_tail.py <csv> [N]"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", "0")))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = rows[-n:]
t0 = rows[0][0]
for s, e, k, st in rows:
    k = k.split("(")[0].replace("void ", "").replace("vsv::", "")[:60]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  s{st:<3s} {k}")
