#!/usr/bin/env python3
"""Where a wave of K16 (csrc/stft16384_w.hip, BASELINE config 4) spends its cycles: diagnostic build -DSGX_STAMPS=1
(SRC=stft16384_w.hip tools/build_variant.sh <name> -DSGX_STAMPS=1), s_memtime at every phase boundary, summed over all waves.
usage: SGX_LIB=spectrogram_rs_amd/ab/<name>.so tools/k16_phases.py [hop positions]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spectrogram_rs_amd import SpectrogramEngine, _lib

HOPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
PH = {1: "job bookkeeping + barrier B0 (previous pass-3 reads done)", 2: "pass 1: pretwiddle, FFT16 x2, 16 sample requests, twiddles, 32 image writes",
      3: "barrier B1 (image complete)", 5: "pass 2: 32 image reads, FFT32, 16 row stores, 31 twiddle reads, 32 writes in place", 6: "barrier B2",
      8: "pass 3: 2 x 16 reads + FFT16 x2", 9: "split", 10: "wait for next samples + Hann", 19: "loop control"}
lib = _lib.load()
fn = lib.sgx_debug_phase_cycles16w
fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 24)()
eng = SpectrogramEngine(48000.0, window_samples=8192, hop_samples=512, channels=8)
pcm = eng.white_noise((HOPS - 1) * 512 + 8192)
out = torch.empty((HOPS, 4, 8191, 2), dtype=torch.float32, device="cuda")
for _ in range(3):
    eng.stft_batch(pcm, out=out)
torch.cuda.synchronize()
fn(buf, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    eng.stft_batch(pcm, out=out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
fn(buf, 1)
iters = buf[20]
total = sum(buf[i] for i in range(20))
print(f"== config 4: {ms:.3f} ms per {HOPS} hop positions (stamped build); {iters} wave-iterations, {total / iters:.0f} cycles per wave-iteration")
bar = 0
for i, ph in sorted(PH.items()):
    print(f"  {buf[i] / iters:8.0f} cycles  {100.0 * buf[i] / total:5.1f} %   {ph}")
    if ph.startswith("barrier"):
        bar += buf[i]
print(f"  barriers: {100.0 * bar / total:.1f} %")
