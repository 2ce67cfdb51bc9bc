#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel in libsgx.so, from the device assembly (hipcc -S --cuda-device-only).

  python tools/kernel_resources.py [file.hip ...] [-D...]     (default: every .hip of the Makefile's SRCS)

Prints one line per kernel: VGPRs (+AGPRs), SGPRs, scratch bytes per lane (private_segment_fixed_size), spilled VGPRs,
static LDS, and the occupancy the register count allows (512 VGPRs per SIMD lane, granule 8).  A non-zero scratch size on
a hot kernel is a finding: a spill reload is a vector-memory load and queues behind the store stream.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "spectrogram_rs_amd", "csrc")
FLAGS = "-O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -munsafe-fp-atomics".split()


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def main(argv):
    defs = [a for a in argv if a.startswith("-D")]
    files = [a for a in argv if not a.startswith("-")]
    if not files:
        mk = open(os.path.join(CSRC, "Makefile")).read()
        files = [f for f in re.search(r"^SRCS\s*:=\s*(.*)$", mk, re.M).group(1).split() if f.endswith(".hip")]
    rows = []
    for f in files:
        src = f if os.path.isabs(f) else os.path.join(CSRC, f)
        asm = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *defs, "-S", "--cuda-device-only", "-o", "-", src], capture_output=True, text=True, cwd=CSRC)
        if asm.returncode != 0:
            print(asm.stderr, file=sys.stderr)
            return 1
        for blk in re.findall(r"- \.agpr_count:.*?\.wavefront_size:\s*\d+", asm.stdout, re.S):
            g = lambda k: int(re.search(r"\.%s:\s*(\d+)" % k, blk).group(1))
            rows.append((os.path.basename(f), re.search(r"\.name:\s*(\S+)", blk).group(1), g("vgpr_count"), g("agpr_count"), g("sgpr_count"),
                         g("private_segment_fixed_size"), g("vgpr_spill_count"), g("group_segment_fixed_size")))
    names = demangle([r[1] for r in rows])
    print("%-22s %5s %5s %5s %8s %6s %7s %5s  %s" % ("file", "vgpr", "agpr", "sgpr", "scratch", "spill", "lds", "w/simd", "kernel"))
    for f, n, v, a, s, sc, sp, lds in rows:
        tot = max(v + a, 1)
        occ = min(8, 512 // (((tot + 7) // 8) * 8))
        print("%-22s %5d %5d %5d %8d %6d %7d %5d  %s" % (f, v, a, s, sc, sp, lds, occ, names[n][:150]))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
