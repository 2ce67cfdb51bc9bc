#!/usr/bin/env python3
"""How far behind its producer does a vector instruction sit?  For one kernel of a device assembly file (hipcc -S --cuda-device-only), over its
innermost loop: for every v_* instruction the distance (in vector instructions of the stream) to the nearest earlier vector instruction that
writes one of its source registers.  A wave of gfx950 issues a dependent vector instruction 8 clocks behind its producer and independent ones
every 4.3 - 4.7 (tools/valu_cadence.hip): distances of 1 cost 8 clocks, of 2 about 6, of 4 about 5.
usage: tools/isa_dep_distance.py file.s kernel-name-substring"""
import collections
import re
import sys

path, want = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and want in l and l.rstrip().endswith(":") or (l.startswith("_Z") and want in l.split(":")[0] and ":" in l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
heads = [i for i, l in enumerate(body) if "Loop Header" in l]
lo = heads[0] if heads else 0
backs = [i for i, l in enumerate(body) if re.match(r"\s+s_cbranch", l)]
hi = max(backs) if backs else len(body)
# the loop may be rotated: take from the first label that the last back edge targets
tgt = re.search(r"(\.LBB\d+_\d+)", body[hi]).group(1) if backs else None
if tgt:
    for i, l in enumerate(body):
        if l.startswith(tgt + ":"):
            lo = min(lo, i)


def regs(tok):
    out = []
    for m in re.finditer(r"v\[(\d+):(\d+)\]|v(\d+)", tok):
        if m.group(3) is not None:
            out.append(int(m.group(3)))
        else:
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


last_write = {}
n = 0
hist = collections.Counter()
cost = 0.0
for l in body[lo:hi]:
    m = re.match(r"\s+(v_\w+)\s+(.*)", l)
    if not m:
        continue
    ops = [o.strip() for o in m.group(2).split(",")]
    dst, srcs = regs(ops[0]), [r for o in ops[1:] for r in regs(o)]
    if m.group(1).startswith(("v_fmac", "v_pk_fma")) or "fmac" in m.group(1):
        srcs += dst
    d = min((n - last_write[r] for r in srcs if r in last_write), default=99)
    hist[min(d, 9)] += 1
    cost += {1: 8.06, 2: 6.06, 3: 5.5, 4: 5.1}.get(d, 4.7)
    for r in dst:
        last_write[r] = n
    n += 1
print(f"{n} vector instructions in the loop; distance to the producer: " + "  ".join(f"{k}{'+' if k == 9 else ''}: {hist[k]}" for k in sorted(hist)))
print(f"one wave alone would need about {cost:.0f} clocks for them ({cost / n:.2f} per instruction)")
