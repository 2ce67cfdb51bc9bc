#!/usr/bin/env python3
"""Instruction mix per kernel of a gfx950 .s file (development aid)."""
import collections
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
labels = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\w+):", s, re.M)]
for idx, (pos, name) in enumerate(labels):
    if pat not in name:
        continue
    end = s.find(".Lfunc_end", pos)
    body = s[pos:end]
    ops = collections.Counter()
    for line in body.split("\n"):
        line = line.strip()
        if not line or line.startswith((".", ";", "//")) or line.endswith(":"):
            continue
        ops[line.split()[0]] += 1
    groups = collections.Counter()
    for op, c in ops.items():
        if op.startswith("v_pk"):
            groups["v_pk"] += c
        elif op.startswith("v_"):
            groups["valu"] += c
        elif op.startswith(("ds_", "global_", "scratch_", "buffer_")):
            groups[op] += c
        elif op.startswith("s_waitcnt"):
            groups["s_waitcnt"] += c
        elif op.startswith("s_"):
            groups["salu"] += c
        else:
            groups[op] += c
    print(name, "total", sum(ops.values()))
    for g, c in sorted(groups.items(), key=lambda x: -x[1]):
        print("    %-28s %d" % (g, c))
    print("    top valu:", [(o, c) for o, c in ops.most_common(40) if o.startswith("v_")][:14])
    m = re.search(r"\.vgpr_count:\s*(\d+)", s[s.find(name, end):])
