#!/usr/bin/env python3
"""Per kernel: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE and friends from a rocprofv3 --pmc output directory."""
import collections
import csv
import glob
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        acc[(r["Kernel_Name"], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (name, grid), ctr in sorted(acc.items(), key=lambda kv: -kv[0][1])[:12]:
    m = {k: sum(v) / len(v) for k, v in ctr.items()}
    line = f"{name[:70]:70s} grid {grid:9d} launches {len(next(iter(ctr.values()))):3d}"
    if m.get("SQ_LDS_IDX_ACTIVE"):
        line += f"  conflict/active {m.get('SQ_LDS_BANK_CONFLICT', 0) / m['SQ_LDS_IDX_ACTIVE']:.4f}"
    for k in sorted(m):
        line += f"  {k}={m[k]:.4g}"
    print(line)
