"""Per basic block instruction classes of a kernel body written by tools/isa_count.py:  python tools/isa_blocks.py body.s"""
import collections
import sys


def cls_of(k):
    return ("mfma" if "mfma" in k else "trans" if k.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq"))
            else "valu" if k.startswith("v_") else "salu" if k.startswith("s_") else "lds" if k.startswith("ds_")
            else "vmem" if k.startswith(("buffer_", "global_", "flat_", "scratch_")) else "other")


name, cnt = "entry", collections.Counter()
for line in open(sys.argv[1]).read().split("\n"):
    s = line.strip()
    if s.startswith(".LBB") and s.split()[0].endswith(":"):
        if sum(cnt.values()):
            print(f"{name:14s} {sum(cnt.values()):5d} {dict(cnt)}")
        name, cnt = s.split(":")[0], collections.Counter()
        continue
    if not s or s.startswith((".", ";")) or s.endswith(":"):
        continue
    cnt[cls_of(s.split()[0])] += 1
print(f"{name:14s} {sum(cnt.values()):5d} {dict(cnt)}")
