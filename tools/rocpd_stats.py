#!/usr/bin/env python3
"""Per-kernel summary (calls, total/avg/min/max duration, share) of a rocprofv3
rocpd SQLite database (`rocprofv3 --kernel-trace --stats` default output).
Usage: python tools/rocpd_stats.py <results.db> [> profiles/<name>.txt]"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    rows = c.execute(f"""select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start),
                         min(d.end-d.start), max(d.end-d.start)
                         from {kd} d join {ks} s on d.kernel_id = s.id
                         group by s.kernel_name order by 3 desc""").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# {path}")
    print(f"{'calls':>7} {'total_ms':>11} {'avg_us':>11} {'min_us':>10} {'max_us':>10} {'pct':>6}  kernel")
    for name, n, tot, avg, mn, mx in rows:
        print(f"{n:7d} {tot/1e6:11.3f} {avg/1e3:11.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*tot/total:6.2f}  {name[:150]}")
    print(f"# total kernel time {total/1e6:.3f} ms")


if __name__ == "__main__":
    main(sys.argv[1])
