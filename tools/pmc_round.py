#!/usr/bin/env python3
"""Turn one round's rocprofv3 --pmc passes over bench.py into the two summaries bench.py quotes:
  profiles/<round>_hbm_traffic.json  -- HBM bytes per unit for the kernels of configs 2, 3 and 4 (FETCH_SIZE x 2 + WRITE_SIZE,
                                        the gfx950 corrections of MI355X_MICROARCH.md section HBM, calibrated on known byte counts)
  profiles/<round>_pixel_pipes.json  -- which pipe the fused pixel kernel keeps busy (SQ counters per launch)
usage: tools/pmc_round.py <round> <fetch_bench> <fetch_micro> <write_bench> <write_micro> <sq1_bench> <sq2_bench>
(directories under gpurun_out/ written by tools/pmc_bench.sh)"""
import collections
import csv
import glob
import json
import os
import sys

rnd, fb_d, fm_d, wb_d, wm_d, s1_d, s2_d = sys.argv[1:8]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from bench import csrc_sha16  # noqa: E402  (the stamp bench.py checks before it quotes these summaries)


def load(d):
    acc = collections.defaultdict(list)
    paths = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    for path in paths[-1:]:   # the newest run only (gpurun_out/ keeps earlier rounds' files beside it)
        for r in csv.DictReader(open(path)):
            acc[(r["Kernel_Name"], int(r["Grid_Size"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    return acc


def pick(acc, name_part, counter):
    """mean counter value over the dispatches of the LARGEST grid whose kernel name contains name_part -- the full-size ones among them:
    a persistent kernel launches the same grid for a 512-hop check as for the leg itself, so dispatches under half the largest value
    (bench.py's same-bytes checks) are left out"""
    c = [(k, v) for k, v in acc.items() if name_part in k[0] and k[2] == counter]
    if not c:
        return None
    k, v = max(c, key=lambda kv: kv[0][1])
    v = [x for x in v if x >= 0.5 * max(v)]
    return sum(v) / len(v)


fb, fm, wb, wm, s1, s2 = [load(d) for d in (fb_d, fm_d, wb_d, wm_d, s1_d, s2_d)]
KIB = 1024.0
out = {"how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes over `bench.py --steps 3 --warmup 1 --sustain-s 0 "
              "--cpu-frames 0` (tools/pmc_bench.sh); counters are KiB; FETCH_SIZE x 2 (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE x 1",
       "calibration": {}, "csrc_sha16": csrc_sha16()}
f = pick(fm, "copy_kernel", "FETCH_SIZE")
if f:
    out["calibration"]["fetch_float4_copy_ratio_raw"] = f * KIB / (262144 * 4096 * 4.0)
vals = [sum(v) / len(v) * KIB for kk, v in wm.items() if "store_kernel<2>" in kk[0] and kk[2] == "WRITE_SIZE"]
if vals:
    out["calibration"]["write_float2_rows_ratio"] = (sum(vals) / len(vals)) / (262144 * 4095 * 4.0)


def traffic(name, units, algorithmic):
    w, f = pick(wb, name, "WRITE_SIZE"), pick(fb, name, "FETCH_SIZE")
    if w is None or f is None:
        return None
    total = w * KIB + 2.0 * f * KIB
    return {"kernel": name, "units_per_launch": units, "WRITE_SIZE_bytes": w * KIB, "FETCH_SIZE_bytes_raw": f * KIB,
            "bytes_per_launch": total, "bytes_per_unit": total / units, "algorithmic_bytes_per_unit": algorithmic,
            "traffic_over_algorithmic": total / units / algorithmic}


F = 1_000_000
# config 2 / 3: the default mono mode is the real-input kernel (every frame its own transform); rows = <0, 0>, cosine pixels = <2, 2>
t2 = traffic("stft4096_real_kernel<0, 0, true>", F, 17400)
t3 = traffic("stft4096_real_kernel<2, 2, true>", F, 5120)
t2p = traffic("stft4096_wg_kernel<true, 0, false, 0>", F, 17400)     # the paired leg (SGX_FLAG_PAIRED_FRAMES), if the pass ran it
hops = 20_000      # (tools/pmc_bench.sh runs the config-4 leg at 20 000 hop positions: a counter pass serialises every dispatch)
t4a = traffic("stft16384_w_kernel<false, true, true>", hops, 278496)      # the pairs read where they lie, the window sliding in registers (hop 512), ONE kernel
out["config2_stft"] = t2
out["config2_stft_paired_frames"] = t2p
out["config3_fused_pixel"] = t3
out["config4_transform"] = t4a
if t2:
    out["stft_bytes_per_frame"] = t2["bytes_per_unit"]
if t3:
    out["pixel_bytes_per_frame"] = t3["bytes_per_unit"]
if t4a:
    out["config4_bytes_per_hop"] = t4a["bytes_per_unit"]
    out["config4_traffic_over_algorithmic"] = out["config4_bytes_per_hop"] / 278496
if t4a:
    fetch_kb = 2.0 * t4a["FETCH_SIZE_bytes_raw"] / hops / 1e3
    write_kb = t4a["WRITE_SIZE_bytes"] / hops / 1e3
    out["note_config4"] = (
        "config 4: FETCH_SIZE / WRITE_SIZE count requests between L2 and the fabric, Infinity-Cache hits included. Per hop position the "
        f"transform kernel (csrc/stft16384_w.hip) writes {write_kb:.0f} KB (262 KB of algorithmic output: 8-byte row stores at an 8-byte row phase, a wave's 512-byte run "
        f"shares its first and last line with its neighbours') and fetches {fetch_kb:.0f} KB: the interleaved stream itself (the pairs are read where they lie; "
        "at hop 512 the window slides in registers, a thread requests ONE new sample per transform: every line is wanted by the four pairs of a run of hop positions, "
        "which work on CUs of one XCD at about the same time; 16 KB algorithmic).")
json.dump(out, open(os.path.join(root, "profiles", f"{rnd}_hbm_traffic.json"), "w"), indent=1)

# ---- pipes of the fused pixel kernel -------------------------------------------------------------------------------
name = "stft4096_real_kernel<2, 2, true>"
g = lambda acc, c: pick(acc, name, c)  # noqa: E731
pipes = {"csrc_sha16": csrc_sha16(), "how": "rocprofv3 --pmc SQ counters (two passes) over the same bench.py command; per launch of 1e6 frames = 5e5 transforms; "
                "SQ_*_CYCLES / ACTIVE counters are in units of 4 clocks summed over waves; GRBM_GUI_ACTIVE summed over the 8 XCDs",
         "kernel": name}
for c in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE"):
    pipes[c] = g(s1, c)
for c in ("SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SALU", "SQ_WAIT_INST_LDS"):
    pipes[c] = g(s2, c)
if pipes.get("GRBM_GUI_ACTIVE") and pipes.get("SQ_INSTS_VALU"):
    cyc_per_cu = pipes["GRBM_GUI_ACTIVE"] / 8.0                 # clocks the launch lasted
    n_simd = 256 * 4
    pipes["launch_clocks"] = cyc_per_cu
    # a wave64 VALU instruction occupies its SIMD for 2 clocks at full rate (SIMD-32)
    pipes["valu_issue_fraction"] = pipes["SQ_INSTS_VALU"] * 2.0 / n_simd / cyc_per_cu
    if pipes.get("SQ_LDS_IDX_ACTIVE"):
        pipes["lds_array_fraction"] = pipes["SQ_LDS_IDX_ACTIVE"] / 256.0 / cyc_per_cu
    if pipes.get("SQ_WAVE_CYCLES"):
        pipes["wave_time_waiting_fraction"] = (pipes["SQ_WAIT_ANY"] or 0) / pipes["SQ_WAVE_CYCLES"]
        pipes["wave_time_issue_stalled_fraction"] = (pipes["SQ_WAIT_INST_ANY"] or 0) / pipes["SQ_WAVE_CYCLES"]
json.dump(pipes, open(os.path.join(root, "profiles", f"{rnd}_pixel_pipes.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
print(json.dumps(pipes, indent=1))
