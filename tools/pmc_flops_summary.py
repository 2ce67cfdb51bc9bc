#!/usr/bin/env python3
"""rocprofv3 --pmc csv of tools/fp32_legs.py + the plan it wrote -> the JSON bench.py reads as profiles/<round>_fp32_flops.json:
flop_per_unit[leg] = 64 lanes x (ADD + MUL + 2 FMA + TRANS wave-instructions) per launch / units per launch.
usage: pmc_flops_summary.py <counter dir> <plan json>"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_sha16  # noqa: E402

rows = []
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(path)))
plan = json.load(open(sys.argv[2]))
# dispatches of the transform kernels, in dispatch order
disp = collections.OrderedDict()
for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
    k = r["Kernel_Name"]
    if "stft" not in k or "duplicate" in k or "deinterleave" in k:
        continue
    d = disp.setdefault(int(r["Dispatch_Id"]), {"kernel": k})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
seq = list(disp.values())
out = {"csrc_sha16": csrc_sha16(),
       "how": "rocprofv3 --pmc SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32 over tools/fp32_legs.py (one pass, counters only); wave-instructions x 64 lanes, FMA twice; "
              "per launch / units per launch.  A packed instruction (v_pk_*_f32: the cubic interpolator of config 3) is one wave-instruction to the counter: "
              "those legs are a lower bound.  Peak: 157.3 TFLOP/s = 256 CUs x 4 SIMDs x 64 flop / clock x 2.4 GHz (v_fma_f32 every 2 clocks per SIMD)",
       "flop_per_unit": {}, "legs": {}}
i = 0
for p in plan:
    mine = seq[i:i + p["launches"]]
    i += p["launches"]
    assert len(mine) == p["launches"] and len({m["kernel"] for m in mine}) == 1, (p, [m["kernel"] for m in mine])
    last = mine[-1]
    ops = {c: last.get("SQ_INSTS_VALU_" + c + "_F32", 0.0) for c in ("ADD", "MUL", "FMA", "TRANS")}
    flop = 64.0 * (ops["ADD"] + ops["MUL"] + 2.0 * ops["FMA"] + ops["TRANS"])
    out["flop_per_unit"][p["leg"]] = flop / p["units_per_launch"]
    out["legs"][p["leg"]] = {"kernel": last["kernel"][:90], "unit": p["unit"], "units_per_launch": p["units_per_launch"],
                             "wave_instructions": ops, "valu_wave_instructions_all": last.get("SQ_INSTS_VALU"), "waves": last.get("SQ_WAVES")}
assert i == len(seq), (i, len(seq))
print(json.dumps(out, indent=1))
