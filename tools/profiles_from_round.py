#!/usr/bin/env python3
"""After tools/profile_round.sh has run on a GPU box and its gpurun_out/pr_* directories are back: write the round's files under profiles/
(the stamped summaries bench.py quotes, the condensed FETCH_SIZE / WRITE_SIZE tables, the SQ counter summary, the kernel trace).
usage: tools/profiles_from_round.py [r03]"""
import collections
import csv
import glob
import os
import shutil
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
g = os.path.join(root, "gpurun_out")
# the counters were measured on the kernel sources whose hash profile_round.sh wrote beside them: the stamp pmc_round.py puts on the
# summaries (the tree's hash NOW) must be that one, or the summaries would claim a build they were not measured on
sys.path.insert(0, root)
from bench import csrc_sha16  # noqa: E402
try:
    measured = open(os.path.join(g, "pr_csrc_sha16.txt")).read().strip()
except OSError:
    sys.exit("gpurun_out/pr_csrc_sha16.txt is missing: run tools/profile_round.sh on a GPU box first")
if measured != csrc_sha16():
    sys.exit(f"the profile round in gpurun_out/ was measured on kernel sources {measured}, the tree is at {csrc_sha16()}: run tools/profile_round.sh again")
# gpurun merges a call's gpurun_out/ into the local one: counter files of EARLIER profile rounds (other builds, other devices) are still
# there and would be averaged in.  Keep this run's only (pr_files.txt, written on the box, where nothing older exists).
try:
    keep = {os.path.normpath(l.strip()) for l in open(os.path.join(g, "pr_files.txt")) if l.strip()}
except OSError:
    sys.exit("gpurun_out/pr_files.txt is missing: run tools/profile_round.sh on a GPU box first")
stale = [p for p in glob.glob(os.path.join(g, "pr_*", "**", "*.csv"), recursive=True) if os.path.normpath(os.path.relpath(p, g)) not in keep]
for p in stale:
    os.remove(p)
if stale:
    print(f"removed {len(stale)} counter / trace files of earlier profile rounds from gpurun_out/")
dirs = [os.path.join(g, d) for d in ("pr_fetch_bench", "pr_fetch_micro", "pr_write_bench", "pr_write_micro", "pr_sq1_bench", "pr_sq2_bench")]
subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_round.py"), rnd, *dirs], check=True, stdout=open(os.path.join(g, "pr_round.txt"), "w"))
for d, n in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    acc = collections.defaultdict(list)
    for path in glob.glob(os.path.join(g, f"pr_{d}_bench", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            acc[(r["Kernel_Name"], int(r["Grid_Size"]), int(r["Workgroup_Size"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    rows = sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:14]
    with open(os.path.join(root, "profiles", f"{rnd}_pmc_{n}_bench.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Grid_Size", "Workgroup_Size", "Counter_Name", "dispatches", "mean_Counter_Value"])
        for (k, grid, wg, c), v in rows:
            w.writerow([k, grid, wg, c, len(v), sum(v) / len(v)])
shutil.copy(os.path.join(g, "pr_kernel_trace.txt"), os.path.join(root, "profiles", f"{rnd}_kernel_trace_bench.txt"))
shutil.copy(os.path.join(g, "pr_sq_counters.txt"), os.path.join(root, "profiles", f"{rnd}_pmc_sq_counters.txt"))
print("profiles written; now run bench.py on a GPU box (it quotes the stamped summaries) and copy its line to profiles/%s_bench_n1.json" % rnd)
