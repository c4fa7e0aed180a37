#!/usr/bin/env python3
"""Static instruction mix per basic block of one kernel in a hipcc --save-temps .s file.
usage: isa_blocks.py file.s kernel_symbol_substring"""
import re, sys
from collections import Counter
src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and key in l)
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("\t.section") or lines[i].startswith(".Lfunc_end"))
blk, blocks = "entry", [("entry", Counter(), [])]
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("v_exp", "v_rcp", "v_rsq", "v_sqrt", "v_log")): return "trans"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")): return "vmem"
    return "other"
for l in lines[start + 1:end]:
    m = re.match(r"^(\.LBB\w+):", l)
    if m:
        blocks.append((m.group(1), Counter(), []))
        continue
    t = l.strip()
    if not t or t.startswith((";", ".")): continue
    op = t.split()[0]
    blocks[-1][1][cls(op)] += 1
    if cls(op) == "branch": blocks[-1][2].append(t)
tot = Counter()
for name, c, br in blocks:
    tot.update(c)
    n = sum(c.values())
    if n < 8: continue
    print(f"{name:12s} n={n:5d} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())) + "  | " + "; ".join(b.replace("\t", " ") for b in br))
print("TOTAL", dict(tot))
