#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in a hipcc -S dump (tuning aid).
   usage: asm_mix.py file.s <substring of the mangled kernel name> [min block size]"""
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
tag = sys.argv[2]
minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 20
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and tag in l and ": " in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
blocks, cur = [], ("entry", [])
for l in lines[start + 1:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur); cur = (m.group(1), [])
    elif re.match(r"\s+[a-z_0-9]+", l) and not l.strip().startswith((".", ";")):
        cur[1].append(l.strip().split()[0])
blocks.append(cur)
tot = collections.Counter()
for name, ins in blocks:
    c = collections.Counter()
    for i in ins:
        k = ("valu" if i.startswith("v_") else "salu" if i.startswith("s_") else "lds" if i.startswith("ds_")
             else "vmem" if i.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
        c[k] += 1; tot[k] += 1
    if len(ins) >= minsz:
        print(name, len(ins), dict(c), collections.Counter(ins).most_common(7))
print("total", sum(tot.values()), dict(tot))
