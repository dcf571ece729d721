#!/usr/bin/env python3
"""(CPU) Run-length summary of a range of lines of a kernel listing: M mfma, R/W LDS read/write, v VALU,
a AGPR copy, LD/ST buffer load/store, WAIT(...), [label], BR branch.
    python tools/isa_stream.py <kernel.s> <first line> <last line>"""
import sys
lines = open(sys.argv[1]).read().split("\n")
a, b = int(sys.argv[2]), int(sys.argv[3])
out, run, cnt = [], None, 0
for l in lines[a:b]:
    t = l.strip()
    if not t or t.startswith(";"):
        continue
    op = t.split()[0]
    if op.endswith(":"): key = "[" + op + "]"
    elif op.startswith("v_mfma"): key = "M"
    elif op.startswith("ds_read"): key = "R"
    elif op.startswith("ds_write"): key = "W"
    elif op == "s_waitcnt": key = "WAIT(" + t.split(None, 1)[1] + ")"
    elif "accvgpr" in op: key = "a"
    elif op.startswith("v_"): key = "v"
    elif op == "s_nop": key = "nop"
    elif op.startswith("s_cbranch") or op.startswith("s_branch"): key = "BR"
    elif op.startswith("s_"): key = "s"
    elif op.startswith("buffer_load"): key = "LD"
    elif op.startswith("buffer_store"): key = "ST"
    else: key = op
    if key == run: cnt += 1
    else:
        if run: out.append(f"{run}{cnt}" if cnt > 1 else run)
        run, cnt = key, 1
out.append(f"{run}{cnt}")
print(" ".join(out))
