#!/usr/bin/env python3
"""Ablation builds of the real-input kernel (csrc/stft4096_real.hip): patched COPIES built into spectrogram_rs_amd/ab/ (the tree's
kernel stays as it is); run with `BENCH=tools/stereo_bench.py tools/ab.sh` (the "mono, default" line is this kernel).

  a_base     the tree's kernel
  b_ln_after_pass1 / c_ln_after_pass2   the next iteration's column requested earlier than the end of pass 3
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
    # where the next column is requested: after pass 1 / after pass 2 instead of after pass 3 (more time to return, two more live registers)
    decl = "        float2 Ln = make_float2(0.0f, 0.0f);\n        if (MODE != kPixels) Ln = column(columns_from(128 * (fa + 2) + 1152), 0);\n"
    t = sub(base, decl, "")
    build("b_ln_after_pass1", sub(t, "        __builtin_amdgcn_s_setprio(0);  // (wave priorities: stft4096_wg.hip)\n",
                                  "        __builtin_amdgcn_s_setprio(0);  // (wave priorities: stft4096_wg.hip)\n" + decl))
    build("c_ln_after_pass2", sub(t, "        // ---- pass 3: thread (F, u): 16-point FFT over t0 -> Z[u + 128 q3]\n",
                                  decl + "        // ---- pass 3: thread (F, u): 16-point FFT over t0 -> Z[u + 128 q3]\n"))
    return 0


if __name__ == "__main__":
    sys.exit(main())
