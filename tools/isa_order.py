"""Instruction order of basic blocks as one letter per instruction (M product, T transcendental, v other vector, r / W LDS
read / write, L / S buffer or global load / store, w s_waitcnt, . other scalar):  python tools/isa_order.py body.s LABEL [LABEL ...]"""
import sys

lines = open(sys.argv[1]).read().split("\n")
for lab in sys.argv[2:]:
    i = next(k for k, l in enumerate(lines) if l.startswith(lab + ":"))
    out = []
    for l in lines[i + 1:]:
        s = l.strip()
        if s.startswith(".LBB"):
            break
        if not s or s.startswith((";", ".")):
            continue
        n = s.split()[0]
        out.append("M" if "mfma" in n else "T" if n.startswith(("v_exp", "v_rcp")) else "v" if n.startswith("v_")
                   else "r" if n.startswith("ds_read") else "W" if n.startswith("ds_write")
                   else "L" if n.startswith(("buffer_load", "global_load")) else "S" if n.startswith(("buffer_store", "global_store"))
                   else " |BAR| " if n == "s_barrier" else "w" if n == "s_waitcnt" else ".")
    print(lab, "".join(out))
