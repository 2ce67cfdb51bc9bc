#!/usr/bin/env python3
"""profiles/<round>_summary.md's table and CPU line from profiles/<round>_bench_n1.json + <round>_hbm_traffic.json (the prose under
them is kept).  usage: tools/round_summary.py r05"""
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1]
j = json.load(open(os.path.join(root, "profiles", rnd + "_bench_n1.json")))
t = json.load(open(os.path.join(root, "profiles", rnd + "_hbm_traffic.json")))
c = j["cpu_baseline"]
c3, c4, st = j["config3"], j["config4"], j["stereo4096"]
def vec(r):
    """FP32 fraction (executed f32 operations / 157.3 TFLOP/s) and share of the vector issue slots, from the round's counter pass"""
    return "%.3f / %.3f" % (r["fp32_frac"], r["valu_issue_frac"]) if r.get("fp32_frac") and r.get("valu_issue_frac") else "-"


rows = [
    ("2: 1e6 frames, 4096-pt, hop 256, mono", "frames/s", "%.1f M" % (j["value"] / 1e6), "%.3f" % j["roofline"]["frac"], vec(j["roofline"]),
     "%.4f" % t["config2_stft"]["traffic_over_algorithmic"], j["roofline"]["kernel"]),
    ("3: + log rows + Viridis -> RGBA (cosine)", "frames/s", "%.1f M" % (c3["frames_per_s"] / 1e6), "%.3f" % c3["roofline"]["frac"], vec(c3["roofline"]),
     "%.4f" % t["config3_fused_pixel"]["traffic_over_algorithmic"], c3["roofline"]["kernel"]),
    ("3 with the cubic interpolator (what the reference runs)", "frames/s", "%.1f M" % (c3["cubic"]["frames_per_s"] / 1e6),
     "%.3f" % (c3["cubic"]["frames_per_s"] * 5120 / 8e12), vec(c3["cubic"]), "-", "sgx::wgr::stft4096_real_kernel<2, 1, true>"),
    ("4: 16384-pt, hop 512, 8 ch, 1e5 hops", "hop positions/s", "%.2f M" % (c4["hop_positions_per_s"] / 1e6), "%.3f" % c4["roofline"]["frac"], vec(c4["roofline"]),
     "%.4f" % t["config4_traffic_over_algorithmic"], c4["roofline"]["kernel"]),
    ("(l, r) stream, 4096-pt (what the reference feeds)", "frames/s", "%.1f M" % (st["frames_per_s"] / 1e6), "%.3f" % st["roofline"]["frac"], vec(st["roofline"]), "-",
     st["roofline"]["kernel"]),
]
path = os.path.join(root, "profiles", rnd + "_summary.md")
old = open(path).read().split("\n")
i = [k for k, ln in enumerate(old) if ln.startswith("CPU, same run")][0]
out = old[:4] + ["| " + " | ".join(r) + " |" for r in rows] + ["",
    "CPU, same run, %d host threads (%s) -- a STAND-IN for the reference's FFTW path, which cannot be built here: whole frame loop via numpy + pocketfft %.2f M frames/s (`kind: port`, the `value`), the oracle's C port %.2f M, "
    "one thread %.0f k; FFTW: %s." % (c["cores"], c["cpu_model"], c["library"]["value"] / 1e6, c["port"]["value"] / 1e6,
                                     c["single_thread"]["value"] / 1e3, c["fftw"])] + old[i + 1:]
open(path, "w").write("\n".join(out))
print("\n".join(out[:12]))
