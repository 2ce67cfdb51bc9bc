#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace csv into the per-kernel summary committed under profiles/.

Dispatches are grouped by (kernel, grid size) so that the headline launches (1e6 frames per launch)
are not averaged together with the small launches of the pixel-path leg.
usage: tools/summarize_prof.py <dir with *_kernel_trace.csv> > profiles/<name>.txt
"""
import collections
import csv
import glob
import sys

rows = []
for path in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(path)))
groups = collections.defaultdict(list)
meta = {}
for r in rows:
    key = (r["Kernel_Name"], int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]))
    groups[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    meta[key] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"], r["Workgroup_Size_X"])
total = sum(sum(v) for v in groups.values())
print("# the last column as rocprofv3's kernel trace reports it: VGPR_Count is HALF the registers the wave is allocated on gfx950 (a K1 launch of 116")
print("# registers, allocated as 120, reads 60); tools/kernel_resources.py lists the ISA's own counts, scratch and spills per kernel")
print("%-84s %10s %6s %12s %12s %12s %7s  %s" % ("kernel", "grid", "calls", "avg_us", "min_us", "max_us", "pct", "vgpr(alloc/2)/agpr/sgpr/lds/scratch/wg"))
for key, v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
    name, grid = key
    print("%-84s %10d %6d %12.2f %12.2f %12.2f %6.2f%%  %s" % (name[:84], grid, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3,
                                                              max(v) / 1e3, 100.0 * sum(v) / total, "/".join(meta[key])))
