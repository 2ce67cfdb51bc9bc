#!/usr/bin/env python3
"""Ablation builds of the real-input kernel (csrc/stft4096_real.hip): patched COPIES built into spectrogram_rs_amd/ab/ (the tree's
kernel stays as it is); run with `BENCH=tools/stereo_bench.py tools/ab.sh` (the "mono, default" line is this kernel).

  a_base     the tree's kernel
  b_order    every wave stores its row in ascending address order (the eight k runs, then the eight 2048 - k runs from q3 = 7 down)
             instead of alternating between the two halves of the row
  c_nostore  one store in sixteen (compute + loads only)
(measured and dropped from this script: the descending half stored at ascending lane addresses (wrong bins) -- the lane order inside a
512-byte run makes no difference; no wave priorities -- 1-3 % slower)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "spectrogram_rs_amd", "csrc")
FLAGS = "-O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-result --offload-arch=gfx950 -munsafe-fp-atomics".split()


def sub(text, old, new):
    assert text.count(old) == 1, (text.count(old), old)
    return text.replace(old, new)


def build(name, text):
    src = os.path.join(CSRC, f"ab_{name}.hip")
    open(src, "w").write(text)
    os.makedirs(os.path.join(CSRC, "build", "ab"), exist_ok=True)
    os.makedirs(os.path.join(ROOT, "spectrogram_rs_amd", "ab"), exist_ok=True)
    obj = os.path.join(CSRC, "build", "ab", name + ".o")
    subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", src, "-o", obj], check=True, cwd=CSRC)
    objs = [os.path.join(CSRC, "build", f) for f in sorted(os.listdir(os.path.join(CSRC, "build"))) if f.endswith(".o") and f != "stft4096_real.hip.o"]
    out = os.path.join(ROOT, "spectrogram_rs_amd", "ab", name + ".so")
    subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-o", out] + objs + [obj], check=True)
    print("built", os.path.relpath(out, ROOT), flush=True)


def main():
    subprocess.run(["make", "-s", "-C", CSRC, "-j8"], check=True)
    base = open(os.path.join(CSRC, "stft4096_real.hip")).read()
    build("a_base", base)
    # v1: L in flight at the loop entry (the compiler then waits with vmcnt(1) at the exchange write: the previous iteration's stores)
    t = sub(base, '        L = column(r0, 9216);         // c[128 fa + 1152 + tid]\n', "")
    t = sub(t, '        asm volatile("" ::"v"(carry.x), "v"(carry.y), "v"(L.x), "v"(L.y));\n', '        asm volatile("" ::"v"(carry.x), "v"(carry.y));\n')
    t = sub(t, "        // (L as well: in flight at the loop entry", "        L = column(r0, 9216);\n        // (L as well: in flight at the loop entry")
    build("b_v1_publish_wait", t)
    # v0: no waits forced in front of the loop at all (loop-header vmcnt(2): every iteration waits for the row stores just issued)
    t = re.sub(r'\n#pragma unroll\n        for \(int j = 0; j < 8; \+\+j\) asm volatile.*?"v"\(win\[q\]\)\);\n', "\n", base, flags=re.S)
    assert t != base
    t = sub(t, '        asm volatile("" : "+v"(Ln.x), "+v"(Ln.y));\n', "")
    build("c_v0_header_wait", t)
    return 0


if __name__ == "__main__":
    sys.exit(main())
