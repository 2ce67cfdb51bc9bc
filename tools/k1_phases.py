#!/usr/bin/env python3
"""Where a wave of K1 (csrc/stft4096_wg.hip) spends its cycles: a diagnostic build (-DSGX_STAMPS=1, tools/build_variant.sh) stamps
s_memtime at every phase boundary of the transform loop and sums the differences over all waves and iterations.
usage: SGX_LIB=spectrogram_rs_amd/ab/<stamps>.so tools/k1_phases.py [frames]     (every stamp drains lgkmcnt: phases that end in LDS
traffic include its completion; the build runs ~10 % slower than the product)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spectrogram_rs_amd import SpectrogramEngine, _lib

F = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
PHASES = ["prefetch wait + Hann + pass-1 FFT8 x2", "barrier 0 (partner reads of the previous transform done)", "pass-1 twiddles + image-1 writes (to completion)",
          "barrier 1", "image-1 reads + FFT16", "barrier 2", "pass-2 twiddles + image-2 writes (to completion)", "barrier 3",
          "image-2 reads + FFT16 + next samples requested", "barrier 4", "partner writes (to completion)", "barrier 5",
          "partner reads + split + sqrt", "row stores issued", "-", "loop control"]
lib = _lib.load()
fn = lib.sgx_debug_phase_cycles
fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 20)()
out = torch.empty((F, 1, 2047, 2), dtype=torch.float32, device="cuda")
for name, kw in (("(l, r) stream", dict(channels=2)), ("mono pairs", dict(channels=1, paired_frames=True))):
    eng = SpectrogramEngine(48000.0, window_samples=2048, hop_samples=256, **kw)
    pcm = eng.white_noise((F - 1) * 256 + 2048)
    for _ in range(3):
        eng.stft_batch(pcm, out=out)
    torch.cuda.synchronize()
    fn(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        eng.stft_batch(pcm, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fn(buf, 1)
    iters = buf[16]
    total = sum(buf[i] for i in range(16))
    print(f"== {name}: {ms:.3f} ms per {F} frames (stamped build); {iters} wave-iterations, {total / iters:.0f} cycles per wave-iteration")
    groups = {"arithmetic + LDS phases": 0, "barriers": 0, "stores": 0}
    for i, ph in enumerate(PHASES):
        if ph == "-":
            continue
        c = buf[i] / iters
        print(f"  {c:8.0f} cycles  {100.0 * buf[i] / total:5.1f} %   {ph}")
        groups["barriers" if ph.startswith("barrier") else "stores" if ph.startswith("row stores") else "arithmetic + LDS phases"] += buf[i]
    print("  " + "   ".join(f"{k}: {100.0 * v / total:.1f} %" for k, v in groups.items()))
    eng.close()
