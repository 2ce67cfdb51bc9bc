#!/usr/bin/env python3
"""What would a FIFTH resident workgroup per CU buy the 4096-point kernel?  K1 is pinned at four
workgroups per CU twice over: 36 864 B of LDS per workgroup (padded strides 272 / 257 + 2 KB of pass-2 twiddles) and 104-128
VGPRs.  Five need <= 32 768 B -- exactly the 4096-point image, no padding: XOR-swizzled rows instead -- and <= 96 VGPRs.

This script writes patched COPIES of csrc/stft4096_wg.{hip,hpp} (the tree's kernel stays as it is) and builds one library per
variant into spectrogram_rs_amd/ab/ (run the workloads with `BENCH=tools/stereo_bench.py tools/ab.sh`):

  a_base    the tree's kernel
  b_swz4    the image at strides 256 / 256 with XOR-swizzled columns (conflict-free like the padded strides under the bank rules
            measured on the device, see `variant`), pass-2 twiddles still in LDS, four workgroups per CU.
            CORRECT results (the tests' checksums must equal a_base's): the price of the swizzle's address arithmetic alone
  c_wg5     b_swz4 + the pass-2 twiddles replaced by a stand-in from registers (WRONG results, same instruction count, no LDS,
            no loads) + __launch_bounds__(256, 5): the bound of the fifth workgroup, whatever the compiler has to spill for it
  d_wg5lean c_wg5 with the register need cut artificially where c_wg5 spills (one Hann factor and one pass-1 twiddle stand in
            for all: WRONG results, same instruction count): the bound with no scratch traffic in the way
  e_wg5lean_noprio   d_wg5lean without the wave priorities (they were tuned for four waves per SIMD)
  f_wg4lean          d_wg5lean held at four workgroups per CU (8 KB more LDS, a grid of 4 per CU): what the stand-ins alone change

usage: tools/k1_wg5_bound.py            (then: BENCH=tools/stereo_bench.py REPS=2 tools/ab.sh  on the GPU box)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "spectrogram_rs_amd", "csrc")
FLAGS = "-O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-result --offload-arch=gfx950 -munsafe-fp-atomics".split()


def sub(text, old, new, count=1):
    assert text.count(old) == count, (text.count(old), old)
    return text.replace(old, new)


def variant(name, swizzle, wg5, lean, noprio=False, four=False):
    hip = open(os.path.join(CSRC, "stft4096_wg.hip")).read()
    hpp = open(os.path.join(CSRC, "stft4096_wg.hpp")).read()
    hip = sub(hip, '#include "stft4096_wg.hpp"', f'#include "ab_{name}.hpp"')
    if swizzle:
        hpp = sub(hpp, "constexpr int kS1 = 272;", "constexpr int kS1 = 256;")
        hpp = sub(hpp, "constexpr int kS2 = 257;", "constexpr int kS2 = 256;")
        # The swizzled addresses are derived, inside every iteration, from an OPAQUE copy of the thread index: visible, the compiler
        # hoists all 64 of them out of the transform loop (loop invariants) and spills 24 registers at the 128 budget.
        hip = sub(hip, "        float er[8], ei[8];\n",
                  "        float er[8], ei[8];\n        int tx = tid;\n        asm volatile(\"\" : \"+v\"(tx));\n        const int q1x = tx >> 4, t0x = tx & 15;\n")
        # Bank rules measured on this device: b64 WRITES go 16 lanes at a time over 32 banks (float2 index mod 16 must differ
        # within a 16-lane group), b64 READS 32 lanes at a time over 64 banks (float2 index mod 32 within a 32-lane group).  (A first
        # attempt XORed bits 4..7 of the column: no effect on the bank at all, 4-way conflicts, the (l, r) launch 5.3 -> 8.8 ms.)
        #   image 1 [q1][t]:            column ^ ((q1 & 1) << 4)   -- the two q1 of a 32-lane read group land in different halves
        #   image 2 [t0][q1 + 16 q2]:   column ^ t0               -- the sixteen t0 of a 16-lane write group land in different banks
        # pass-1 write: row q, column tid
        hip = sub(hip, "buf[(2 * j) * kS1 + tid] =", "buf[(2 * j) * kS1 + tx] =")
        hip = sub(hip, "buf[(2 * j + 1) * kS1 + tid] =", "buf[(2 * j + 1) * kS1 + (tx ^ 16)] =")
        # pass-2 read: row q1, column t0 + 16 t1 -> t0 + 16 (t1 ^ (q1 & 1))
        hip = sub(hip, "buf[q1_2 * kS1 + t0_2 + 16 * t1];", "buf[(q1x * kS1 + t0x) + 16 * (t1 ^ (q1x & 1))];")
        # pass-2 write: row t0, column q1 + 16 q2 -> (q1 ^ t0) + 16 q2
        hip = sub(hip, "buf[t0_2 * kS2 + q1_2 + 16 * q2] =", "buf[(t0x * kS2 + (q1x ^ t0x)) + 16 * q2] =")
        # pass-3 read: row t0, column col -> col ^ t0
        hip = sub(hip, "buf[t0 * kS2 + col];", "buf[t0 * kS2 + (tx ^ t0)];")
    if wg5:
        hpp = sub(hpp, "constexpr size_t kLdsBytes = (size_t)(kBufComplex + 256) * sizeof(float2);",
                  "constexpr size_t kLdsBytes = (size_t)kBufComplex * sizeof(float2);   // 32 768 B: five workgroups per CU")
        hip = sub(hip, "__launch_bounds__(256, 4) stft4096_wg_kernel", "__launch_bounds__(256, 5) stft4096_wg_kernel")
        hip = sub(hip, "    tw2[tid] = p.tw2[tid];\n", "")
        hip = sub(hip, "cmulf(v, tw2[q2 * 16 + t0_2]);", "cmulf(v, tw1[q2]);   // stand-in: no table in LDS")
        hip = sub(hip, "    float2 *tw2 = buf + kBufComplex;\n", "")
        hip = sub(hip, "reinterpret_cast<uint2 *>(tw2 + 256);", "reinterpret_cast<uint2 *>(buf + kBufComplex);")
        hip = sub(hip, "unsigned long long blocks = (unsigned long long)n_cu * 4;", "unsigned long long blocks = (unsigned long long)n_cu * 5;")
        # say once how many workgroups the runtime will really keep resident per CU
        hip = sub(hip, "        const dim3 grid((unsigned)blocks), block(256);\n",
                  "        const dim3 grid((unsigned)blocks), block(256);\n"
                  "        { static bool said = false; if (!said) { said = true; int nb = -1, nb2 = -1;\n"
                  "            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, stft4096_wg_kernel<false, kPairAdjacent, true, RENDER>, 256, RENDER ? kLdsBytesRender : kLdsBytes);\n"
                  "            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb2, stft4096_wg_kernel<false, kPairAdjacent, true, RENDER>, 256, (RENDER ? kLdsBytesRender : kLdsBytes) - 1024);\n"
                  "            fprintf(stderr, \"occupancy: %d workgroups per CU at %zu B of LDS, %d at 1 KB less\\n\", nb, (size_t)(RENDER ? kLdsBytesRender : kLdsBytes), nb2); } }\n")
    if lean:
        hip = sub(hip, "for (int a = 0; a < 8; ++a) win[a] = p.window[tid + 256 * a] * inv_w;",
                  "for (int a = 0; a < 8; ++a) win[a] = p.window[tid] * inv_w;   // stand-in: one factor for all eight rows")
        hip = sub(hip, "for (int q = 1; q < 16; ++q) tw1[q] = p.tw1[q * 256 + tid];",
                  "for (int q = 1; q < 16; ++q) tw1[q] = p.tw1[256 + tid];   // stand-in: one twiddle for all fifteen")
    if noprio:
        import re
        hip, n = re.subn(r"__builtin_amdgcn_s_setprio\(\d\);", ";", hip)
        assert n >= 5, n
    if four:   # the lean build held at FOUR workgroups per CU (one KB more LDS than five allow, grid of 4 per CU): the stand-ins' own effect
        hip = sub(hip, "const size_t lds = RENDER ? kLdsBytesRender : kLdsBytes;", "const size_t lds = (RENDER ? kLdsBytesRender : kLdsBytes) + 8192;")
        hip = sub(hip, "unsigned long long blocks = (unsigned long long)n_cu * 5;", "unsigned long long blocks = (unsigned long long)n_cu * 4;")
    open(os.path.join(CSRC, f"ab_{name}.hip"), "w").write(hip)
    open(os.path.join(CSRC, f"ab_{name}.hpp"), "w").write(hpp)


def build(name, src):
    os.makedirs(os.path.join(CSRC, "build", "ab"), exist_ok=True)
    os.makedirs(os.path.join(ROOT, "spectrogram_rs_amd", "ab"), exist_ok=True)
    obj = os.path.join(CSRC, "build", "ab", name + ".o")
    subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", src, "-o", obj], check=True, cwd=CSRC)
    objs = [os.path.join(CSRC, "build", f) for f in sorted(os.listdir(os.path.join(CSRC, "build")))
            if f.endswith(".o") and f != "stft4096_wg.hip.o"]
    out = os.path.join(ROOT, "spectrogram_rs_amd", "ab", name + ".so")
    subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-o", out] + objs + [obj], check=True)
    print("built", os.path.relpath(out, ROOT), flush=True)


def main():
    subprocess.run(["make", "-s", "-C", CSRC, "-j8"], check=True)
    build("a_base", "stft4096_wg.hip")
    for name, kw in (("b_swz4", dict(swizzle=True, wg5=False, lean=False)), ("c_wg5", dict(swizzle=True, wg5=True, lean=False)),
                     ("d_wg5lean", dict(swizzle=True, wg5=True, lean=True)),
                     ("e_wg5lean_noprio", dict(swizzle=True, wg5=True, lean=True, noprio=True)),
                     ("f_wg4lean", dict(swizzle=True, wg5=True, lean=True, four=True))):
        variant(name, **kw)
        build(name, f"ab_{name}.hip")
    return 0


if __name__ == "__main__":
    sys.exit(main())
