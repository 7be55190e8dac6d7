#!/usr/bin/env python
"""Instruction histogram / wait pattern of one kernel in a hipcc -save-temps .s file.
usage: isa_stats.py file.s kernel_substring"""
import re
import sys
from collections import Counter

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if key in l and l.rstrip().endswith(":") is False and re.match(r"^_Z.*:", l) and key in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = [l.strip() for l in lines[start + 1:end]]
ins = [l for l in body if l and not l.startswith((";", ".")) and not l.endswith(":")]
c = Counter(l.split()[0] for l in ins)
print(f"{len(ins)} instructions")
for k, v in c.most_common(28):
    print(f"  {k:32s} {v}")
seq = []
for l in ins:
    op = l.split()[0]
    if op.startswith("v_mfma"):
        t = "M"
    elif op.startswith(("global_load", "buffer_load")):
        t = "L"
    elif op.startswith("ds_read") or op.startswith("ds_load"):
        t = "r"
    elif op.startswith("ds_write") or op.startswith("ds_store"):
        t = "w"
    elif op == "s_waitcnt":
        m = re.search(r"vmcnt\((\d+)\)", l)
        t = f"[v{m.group(1)}]" if m else ("[lg]" if "lgkmcnt" in l else "[w]")
    elif op == "s_barrier":
        t = "|B|"
    elif op.startswith(("s_cbranch", "s_branch")):
        t = "^"
    elif op.startswith("global_store"):
        t = "S"
    else:
        continue
    seq.append(t)
# run-length compress
out, prev, n = [], None, 0
for t in seq + [None]:
    if t == prev:
        n += 1
    else:
        if prev is not None:
            out.append(prev if n == 1 else f"{prev}x{n}")
        prev, n = t, 1
print(" ".join(out))
for l in lines[end:end + 400]:
    if any(k in l for k in ("vgpr_count", "agpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "vgpr_spill")) and key in "".join(lines[end:end+5]) or False:
        print(l.strip())
