#!/usr/bin/env python3
"""K1 / stereo / fused-pixel rates for one run length (SGX_K1_RUN, read by the library once per process) at several places
inside one 80 GiB allocation (profiles/r03_k1_slow_box.txt, part 6).
The run-length loop this measured was an experiment and is NOT in the tree (part 6a describes it): against the current library every
SGX_K1_RUN value reads like the first line of that table."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spectrogram_rs_amd import SpectrogramEngine

F = 1_000_000
pool = torch.empty(80 << 28, dtype=torch.float32, device="cuda")


def timed(fn, n=4):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


out = [f"run={os.environ.get('SGX_K1_RUN', 'per')}"]
eng = SpectrogramEngine(48000.0, window_samples=2048, hop_samples=256, channels=1)
pcm = eng.white_noise((F - 1) * 256 + 2048)
need = F * 2047 * 2
for gib in (0, 30, 56):
    o = pool[gib << 28:(gib << 28) + need].view(F, 1, 2047, 2)
    ms = timed(lambda: eng.stft_batch(pcm, out=o))
    out.append(f"K1@{gib}: {ms:.3f} ms {F * 17400 / ms / 1e6 / 8000:.3f}")
eng.close()
for interp, nm in ((1, "cos"), (0, "cub")):
    eng = SpectrogramEngine(48000.0, window_samples=2048, hop_samples=256, channels=1, interp=interp, gradient="viridis")
    rg = torch.empty((F, 1, 1024, 4), dtype=torch.uint8, device="cuda")
    ms = timed(lambda: eng.render_batch(pcm, out=rg))
    out.append(f"fused {nm}: {ms:.3f} ms {F / ms / 1e3:.1f} M/s")
    eng.close()
Fs = 500_000
eng2 = SpectrogramEngine(48000.0, window_samples=2048, hop_samples=256, channels=2)
pcm2 = eng2.white_noise((Fs - 1) * 256 + 2048)
for gib in (0, 56):
    o = pool[gib << 28:(gib << 28) + Fs * 2047 * 2].view(Fs, 1, 2047, 2)
    ms = timed(lambda: eng2.stft_batch(pcm2, out=o))
    out.append(f"stereo@{gib}: {ms:.3f} ms {Fs / ms / 1e3:.1f} M/s")
print("  ".join(out), flush=True)
