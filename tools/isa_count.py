"""Static instruction mix of one kernel in a hipcc -S dump:  python tools/isa_count.py dump.s kernel_substring [out.s]"""
import collections
import sys

lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sys.argv[2] in l and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end + 1]
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write("\n".join(body))
cnt = collections.Counter()
for line in body:
    line = line.strip()
    if not line or line.startswith((".", ";")) or line.endswith(":") or ":" in line.split()[0]:
        continue
    cnt[line.split()[0]] += 1
tot = sum(cnt.values())
cls = collections.Counter()
for k, v in cnt.items():
    c = ("mfma" if "mfma" in k else "trans" if k.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_sin", "v_cos", "v_rsq"))
         else "valu" if k.startswith("v_") else "salu" if k.startswith("s_") else "lds" if k.startswith("ds_")
         else "vmem" if k.startswith(("buffer_", "global_", "flat_", "scratch_")) else "other")
    cls[c] += v
print(tot, dict(cls))
for k, v in cnt.most_common(int(sys.argv[4]) if len(sys.argv) > 4 else 45):
    print(f"{v:6d} {k}")
