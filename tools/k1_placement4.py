#!/usr/bin/env python3
"""K1's rate as a function of WHERE inside one big allocation its 16.4 GB output lies (profiles/r03_k1_slow_box.txt, part 5)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spectrogram_rs_amd import SpectrogramEngine

F = 1_000_000
eng = SpectrogramEngine(48000.0, window_samples=2048, hop_samples=256, channels=1)
pcm = eng.white_noise((F - 1) * 256 + 2048)
pool_gib = int(sys.argv[1]) if len(sys.argv) > 1 else 96
pool = torch.empty(pool_gib << 28, dtype=torch.float32, device="cuda")      # pool_gib GiB
print(f"pool {pool_gib} GiB at {pool.data_ptr():#x}", flush=True)
need = F * 2047 * 2


def k1(out):
    for _ in range(2):
        eng.stft_batch(pcm, out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(4):
        eng.stft_batch(pcm, out=out)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 4


step = 1 << 28     # floats: 1 GiB
offs = list(range(0, pool.numel() - need + 1, step))
for name, order in (("forward", offs), ("reverse", offs[::-1])):
    print(name, flush=True)
    for off in order:
        ms = k1(pool[off:off + need].view(F, 1, 2047, 2))
        print(f"offset {off * 4 / 2**30:8.3f} GiB: {ms:.3f} ms = {F * 17400 / ms / 1e6 / 8000:.3f}", flush=True)

# a local speed map: the same kernel over a 1 GiB window (65 536 frames) at every GiB of the pool
Fs = 65536
pcm_s = pcm[: (Fs - 1) * 256 + 2048]
F, pcm = Fs, pcm_s
need = Fs * 2047 * 2
print("1 GiB windows", flush=True)
for off in range(0, pool.numel() - need + 1, step):
    ms = k1(pool[off:off + need].view(Fs, 1, 2047, 2))
    print(f"offset {off * 4 / 2**30:8.3f} GiB: {ms * 1e3:.1f} us = {Fs * 17400 / ms / 1e6 / 8000:.3f}", flush=True)
