#!/usr/bin/env python3
"""Generate straight-line, in-register FFT codelets for the tuned 4096-point STFT kernel.

Each codelet is a forward DFT (kernel e^{-2 pi i n k / N}) on N complex points held one per
register pair (arrays `r[]`, `i[]` with literal indices, so they live in VGPRs), decimation in
frequency with radix-4 / radix-2 stages.  Twiddles are emitted as float literals rounded from
float64; multiplications by 1, -1, -i, i cost nothing (the ISA has operand negation).  A complex
twiddle multiply is 2 v_mul + 2 v_fma.

The result is left in place in permuted order; `<name>_OUT[k]` names the element that holds bin k.

Writes spectrogram_rs_amd/csrc/fft_codelets.inc (committed; regenerate with this script).
"""
import math
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lit(x: float) -> str:
    v = np.float32(x)
    if v == 0:
        return "0.0f"
    s = repr(float(v))
    # shortest repr of the float32 value
    s = np.format_float_scientific(v, unique=True, trim="0") if ("e" in s) else np.format_float_positional(v, unique=True, trim="0")
    return s + "f"


class Emitter:
    def __init__(self):
        self.lines = []
        self.n = 0
        self.ops = 0

    def tmp(self):
        self.n += 1
        return "t%d" % self.n

    def emit(self, expr):
        v = self.tmp()
        self.lines.append("    const T %s = %s;" % (v, expr))
        self.ops += 1
        return v

    def add(self, a, b):
        return self.emit("%s + %s" % (a, b))

    def sub(self, a, b):
        return self.emit("%s - %s" % (a, b))


def neg(x):
    return x[1:] if x.startswith("-") else "-" + x


def twiddle(e, z, idx, N):
    """z * e^{-2 pi i idx / N}; z = (re, im) expression names"""
    idx %= N
    a, b = z
    if idx == 0:
        return (a, b)
    if 4 * idx == N:  # -i
        return (b, "(-%s)" % a)
    if 2 * idx == N:  # -1
        return ("(-%s)" % a, "(-%s)" % b)
    if 4 * idx == 3 * N:  # +i
        return ("(-%s)" % b, a)
    ang = -2.0 * math.pi * idx / N
    c, d = math.cos(ang), math.sin(ang)
    u = e.emit("%s * %s" % (b, lit(-d)))
    re = e.emit("cl_fma(%s, %s, %s)" % (a, lit(c), u))
    v = e.emit("%s * %s" % (b, lit(c)))
    im = e.emit("cl_fma(%s, %s, %s)" % (a, lit(d), v))
    return (re, im)


def dif(e, vals, N, radices, out_pos, base_k=0, k_stride=1, positions=None):
    """vals: dict position -> (re, im) names for the N positions in `positions` (ordered).
    After processing, out_pos[k] = position holding bin base_k + k_stride * k."""
    if positions is None:
        positions = list(range(N))
    if N == 1:
        out_pos[base_k] = positions[0]
        if getattr(e, "hooks", None) is not None:      # bin base_k is final here: hand it to the caller at once
            re, im = vals[positions[0]]
            e.lines.append("    done(std::integral_constant<int, %d>{}, %s, %s);" % (base_k, re, im))
        return
    R = radices[0]
    L = N // R
    for j in range(L):
        p = [positions[j + L * q] for q in range(R)]
        x = [vals[q] for q in p]
        if R == 2:
            y0 = (e.add(x[0][0], x[1][0]), e.add(x[0][1], x[1][1]))
            d = (e.sub(x[0][0], x[1][0]), e.sub(x[0][1], x[1][1]))
            ys = [y0, twiddle(e, d, j, N)]
        elif R == 4:
            s02 = (e.add(x[0][0], x[2][0]), e.add(x[0][1], x[2][1]))
            d02 = (e.sub(x[0][0], x[2][0]), e.sub(x[0][1], x[2][1]))
            s13 = (e.add(x[1][0], x[3][0]), e.add(x[1][1], x[3][1]))
            d13 = (e.sub(x[1][0], x[3][0]), e.sub(x[1][1], x[3][1]))
            y0 = (e.add(s02[0], s13[0]), e.add(s02[1], s13[1]))
            y2 = (e.sub(s02[0], s13[0]), e.sub(s02[1], s13[1]))
            y1 = (e.add(d02[0], d13[1]), e.sub(d02[1], d13[0]))  # d02 - i d13
            y3 = (e.sub(d02[0], d13[1]), e.add(d02[1], d13[0]))  # d02 + i d13
            ys = [y0, twiddle(e, y1, j, N), twiddle(e, y2, 2 * j, N), twiddle(e, y3, 3 * j, N)]
        else:
            raise ValueError(R)
        for q in range(R):
            vals[p[q]] = ys[q]
        if getattr(e, "hooks", None) is not None:      # a call-back point behind every butterfly (gen_fft(..., hooked=True))
            e.lines.append("    hook(std::integral_constant<int, %d>{});" % e.hooks)
            e.hooks += 1
    for q in range(R):
        dif(e, vals, L, radices[1:], out_pos, base_k + k_stride * q, k_stride * R, positions[q * L:(q + 1) * L])


def gen_fft(name, N, radices, hooked=False):
    e = Emitter()
    if hooked:
        e.hooks = 0
    vals = {p: ("r[%d]" % p, "i[%d]" % p) for p in range(N)}
    out_pos = {}
    dif(e, vals, N, radices, out_pos)
    body = list(e.lines)
    for p in range(N):
        re, im = vals[p]
        body.append("    r[%d] = %s; i[%d] = %s;" % (p, re, p, im))
    src = []
    src.append("// %d-point forward DFT, DIF radices %s: %d VALU operations" % (N, radices, e.ops))
    if hooked:
        src.append("// (the same with a call-back point behind each of its %d butterflies: hook(std::integral_constant<int, k>{}), k in order --\n"
                   "// where the caller puts vector-memory instructions it wants spread thinly through the arithmetic; bins as %s_OUT)" % (e.hooks, name[:-1].upper()))
        src.append("// and done(std::integral_constant<int, bin>{}, re, im) the moment a bin is final (the last stage's outputs, as they come)")
        src.append("template <typename T, typename H, typename D>\n__device__ __forceinline__ void %s(T (&r)[%d], T (&i)[%d], H &&hook, D &&done)\n{" % (name, N, N))
    else:
        src.append("template <typename T>\n__device__ __forceinline__ void %s(T (&r)[%d], T (&i)[%d])\n{" % (name, N, N))
    src += body
    src.append("}")
    if not hooked:
        src.append("// bin k of %s is left in element %s_OUT[k]" % (name, name.upper()))
        src.append("__device__ constexpr int %s_OUT[%d] = {%s};" % (name.upper(), N, ", ".join(str(out_pos[k]) for k in range(N))))
    return "\n".join(src), e.ops, out_pos


def gen_pretwiddle(name, n, N):
    """element a *= e^{-2 pi i a / N} for a < n"""
    e = Emitter()
    out = []
    for a in range(n):
        z = twiddle(e, ("r[%d]" % a, "i[%d]" % a), a, N)
        out.append((a, z))
    body = list(e.lines)
    for a, (re, im) in out:
        if (re, im) != ("r[%d]" % a, "i[%d]" % a):
            # swaps read both old values first: the temporaries above already captured products,
            # the trivial cases are pure renames of the *old* values
            body.append("    { const T nr = %s, ni = %s; r[%d] = nr; i[%d] = ni; }" % (re, im, a, a))
    src = ["// element a *= w_%d^a, a < %d: %d VALU operations" % (N, n, e.ops),
           "template <typename T>\n__device__ __forceinline__ void %s(T (&r)[%d], T (&i)[%d])\n{" % (name, n, n)]
    src += body
    src.append("}")
    return "\n".join(src), e.ops


def reference_check():
    """numerically execute the generated algorithm in float64 to validate structure + permutation"""
    rng = np.random.default_rng(0)
    for N, radices in ((32, [4, 4, 2]), (64, [4, 4, 4]), (16, [4, 4]), (8, [4, 2])):
        x = rng.normal(size=N) + 1j * rng.normal(size=N)
        vals = {p: x[p] for p in range(N)}
        out_pos = {}

        def run(N_, rad, base_k, ks, positions):
            if N_ == 1:
                out_pos[base_k] = positions[0]
                return
            R = rad[0]
            L = N_ // R
            for j in range(L):
                p = [positions[j + L * q] for q in range(R)]
                xs = [vals[q] for q in p]
                for q in range(R):
                    acc = sum(xs[pp] * np.exp(-2j * np.pi * pp * q / R) for pp in range(R))
                    vals[p[q]] = acc * np.exp(-2j * np.pi * j * q / N_)
            for q in range(R):
                run(L, rad[1:], base_k + ks * q, ks * R, positions[q * L:(q + 1) * L])

        run(N, radices, 0, 1, list(range(N)))
        X = np.fft.fft(x)
        got = np.array([vals[out_pos[k]] for k in range(N)])
        assert np.abs(got - X).max() < 1e-10, (N, np.abs(got - X).max())


def main():
    reference_check()
    parts = ["// GENERATED by tools/gen_fft_codelets.py -- do not edit.\n"
             "// In-register forward-DFT codelets for stft4096.hip.  T is float (one transform) or a\n"
             "// 2-wide float vector (two independent transforms in the halves of packed registers).\n"
             "// cl_fma(a, c, u) = a * c + u with a scalar constant c.\n"]
    total = {}
    for name, N, radices in (("fft8", 8, [4, 2]), ("fft16", 16, [4, 4]), ("fft32", 32, [4, 4, 2]), ("fft64", 64, [4, 4, 4])):
        src, ops, _ = gen_fft(name, N, radices)
        parts.append(src + "\n")
        total[name] = ops
    for name, n, N in (("pretwiddle32_w64", 32, 64), ("pretwiddle8_w16", 8, 16), ("pretwiddle16_w32", 16, 32)):
        src, ops = gen_pretwiddle(name, n, N)
        parts.append(src + "\n")
        total[name] = ops
    for name, N, radices in (("fft16h", 16, [4, 4]), ("fft32h", 32, [4, 4, 2])):
        src, ops, _ = gen_fft(name, N, radices, hooked=True)
        parts.append(src + "\n")
        total[name] = ops
    path = os.path.join(ROOT, "spectrogram_rs_amd", "csrc", "fft_codelets.inc")
    with open(path, "w") as f:
        f.write("\n".join(parts))
    print("wrote", path, total)


if __name__ == "__main__":
    main()
