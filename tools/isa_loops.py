#!/usr/bin/env python3
"""(CPU) Per-loop instruction mix of one kernel in a gfx950 assembly listing (--save-temps):
MFMAs, AGPR copies, LDS reads/writes, stores, plain VALU, waits -- to see what hipcc made of a
hand-pipelined tile loop before spending GPU time on it.
    python tools/isa_loops.py <listing.s> <substring of the mangled kernel name>"""
import re, sys

s = open(sys.argv[1]).read()
name = sys.argv[2]
m0 = re.search(r"^(\S*" + re.escape(name) + r"\S*):", s, re.M)
i = m0.start()
k = s[i:s.index(".Lfunc_end", i)]
lines = k.split("\n")
valu = re.compile(r"^\s+v_(?!mfma|accvgpr)", re.M)
def mix(b):
    return (f"mfma {b.count('v_mfma')}, agpr_rd {b.count('v_accvgpr_read')}, agpr_wr {b.count('v_accvgpr_write')}, "
            f"ds_read {b.count('ds_read')}, ds_write {b.count('ds_write')}, buf_load {b.count('buffer_load')}, "
            f"store {b.count('buffer_store') + b.count('global_store')}, valu {len(valu.findall(b))}, "
            f"waitcnt {b.count('s_waitcnt')}, nop {b.count('s_nop')}, scratch {b.count('scratch_')}")
print(m0.group(1), len(lines), "lines;", mix(k))
labels = {}
for n, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = n
for n, l in enumerate(lines):
    m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < n:
        a = labels[m.group(1)]
        print(f"loop {m.group(1)} lines {a}-{n} ({n - a}):", mix("\n".join(lines[a:n])))
