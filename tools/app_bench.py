#!/usr/bin/env python3
"""The application's operating point (W 2400, 4800-point mixed radix, hop 93): rows and PCM -> RGBA, stereo and mono.
usage: app_bench.py [frames] [reps]  -- the workload of the A/B and counter passes on the mixed-radix kernel"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spectrogram_rs_amd import SpectrogramEngine

F = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
RATE = float(os.environ.get("APP_RATE", "48000"))   # 44100 -> W 2205, the 4410-point mixed-radix kernel
W, H = int(round(RATE * 0.05)), int(os.environ.get("APP_HOP", "93"))


def timeit(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


for ch in (2, 1):
    eng = SpectrogramEngine(RATE, period=0.05, hop_samples=H, channels=ch, interp=0, gradient="viridis",
                            mixed_generic=bool(os.environ.get("APP_GENERIC")))
    assert eng.W == W
    pcm = eng.white_noise((F - 1) * H + W)
    out = torch.empty((F, 1, W - 1, 2), dtype=torch.float32, device="cuda")
    ms = timeit(lambda: eng.stft_batch(pcm, out=out))
    cs = eng.checksum(out[:1024])
    del out
    rgba = torch.empty((F, 1, 1024, 4), dtype=torch.uint8, device="cuda")
    msp = timeit(lambda: eng.render_batch(pcm, out=rgba))
    print(f"ch={ch}: rows {ms:.3f} ms -> {F / ms / 1e3:.1f} M frames/s ({F * (H * ch * 4 + (W - 1) * 8) / ms / 1e6 / 8000:.3f} of HBM)   "
          f"pixels {msp:.3f} ms -> {F / msp / 1e3:.1f} M frames/s   kernel {eng.info.stft_kernel} checksum {cs:016x} / {eng.checksum(rgba[:1024]):016x}", flush=True)
    del rgba, pcm
    eng.close()
