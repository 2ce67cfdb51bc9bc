#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection csv files: mean counter value per kernel name (largest-grid dispatches only)."""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(path)))
        acc = collections.defaultdict(list)
        for r in rows:
            acc[(r["Kernel_Name"][:60], r["Counter_Name"], r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
        for (k, c, g), v in sorted(acc.items()):
            if "stft" in k or "render" in k or "magnitude" in k or "deinterleave" in k:
                print(f"{k:60s} grid={g:>8s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
