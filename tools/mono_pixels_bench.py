#!/usr/bin/env python3
"""A mono stream in the default mode at other windows: rows and PCM -> pixels (Viridis, cubic), median of 7 launches.
usage: mono_pixels_bench.py [W,H ...]   (default: the windows with compile-time W-point plans)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spectrogram_rs_amd import SpectrogramEngine


def timeit(fn, reps=7):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(512, 64), (1024, 128), (4096, 256), (4800, 187), (4410, 172), (1600, 62), (800, 31)]
for W, H in cases:
    eng = SpectrogramEngine(48000.0, window_samples=W, hop_samples=H, channels=1, interp=0, gradient="viridis")
    F = max(20000, min(400000, (1 << 31) // (W * 8)))
    pcm = eng.white_noise((F - 1) * H + W)
    out = eng.stft_batch(pcm)
    ms = timeit(lambda: eng.stft_batch(pcm, out=out))
    del out
    rgba = eng.render_batch(pcm)
    msp = timeit(lambda: eng.render_batch(pcm, out=rgba))
    print(f"W={W} H={H} F={F} kernel {eng.info.stft_kernel} path {eng.info.render_path}: rows {ms:.3f} ms = {F / ms / 1e3:.1f} M frames/s   "
          f"pixels {msp:.3f} ms = {F / msp / 1e3:.1f} M frames/s   checksum {eng.checksum(rgba[:512]):016x}", flush=True)
    del rgba, pcm
    eng.close()
