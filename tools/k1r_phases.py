#!/usr/bin/env python3
"""Where a wave of K1R (csrc/stft4096_real.hip: the headline's mono rows and BASELINE config 3's fused pixels) spends its cycles:
diagnostic build, SRC=stft4096_real.hip tools/build_variant.sh <name> -DSGX_STAMPS=1; s_memtime at every phase boundary.
usage: SGX_LIB=spectrogram_rs_amd/ab/<name>.so tools/k1r_phases.py [frames]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spectrogram_rs_amd import SpectrogramEngine, _lib

F = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
PH = {0: "Hann + pass 1 (two frames: four half-zero FFT8)", 1: "barrier 0", 2: "pass-1 twiddles + image-1 writes", 3: "barrier 1", 4: "image-1 reads + FFT16", 5: "barrier 2",
      6: "pass-2 twiddles + image-2 writes", 7: "barrier 3", 8: "image-2 reads + FFT16 + next column requested", 9: "barrier 4", 10: "partner + window-exchange writes",
      11: "barrier 5", 12: "window slide + partner reads + untangle + sqrt", 13: "(pixels) barrier 6", 14: "(pixels) column writes", 15: "(pixels) barrier 7",
      16: "(pixels) sample pass", 17: "(pixels) barrier 8", 18: "rows: 16 stores issued + next column waited for / pixels: row pass", 19: "loop control"}
lib = _lib.load()
fn = lib.sgx_debug_phase_cycles_r
fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 24)()
eng = SpectrogramEngine(48000.0, window_samples=2048, hop_samples=256, channels=1, interp=1, gradient="viridis")
pcm = eng.white_noise((F - 1) * 256 + 2048)
rows = torch.empty((F, 1, 2047, 2), dtype=torch.float32, device="cuda")
pix = torch.empty((F, 1, 1024, 4), dtype=torch.uint8, device="cuda")
for name, fnc in (("mono rows (the headline)", lambda: eng.stft_batch(pcm, out=rows)), ("config 3: fused pixels, cosine", lambda: eng.render_batch(pcm, out=pix))):
    for _ in range(3):
        fnc()
    torch.cuda.synchronize()
    fn(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fnc()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fn(buf, 1)
    iters = buf[20]
    total = sum(buf[i] for i in range(20))
    print(f"== {name}: {ms:.3f} ms per {F} frames (stamped build); {iters} wave-iterations (two frames each), {total / iters:.0f} cycles per wave-iteration")
    bar = 0
    for i, ph in sorted(PH.items()):
        if buf[i]:
            print(f"  {buf[i] / iters:8.0f} cycles  {100.0 * buf[i] / total:5.1f} %   {ph}")
        if "barrier" in ph:
            bar += buf[i]
    print(f"  barriers: {100.0 * bar / total:.1f} %")
