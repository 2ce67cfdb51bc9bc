#!/usr/bin/env python3
"""K1 (mono, 1e6 frames) under the library named by SGX_LIB, with the device's own fill rate beside it -- for the slow-box A/B of
profiles/r03_k1_slow_box.txt.  SGX_AB_PITCH16K=1: the output buffer holds 16 384 bytes per row (the pitch variant's layout)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spectrogram_rs_amd import SpectrogramEngine

F = 1_000_000
eng = SpectrogramEngine(48000.0, window_samples=2048, hop_samples=256, channels=1)
pcm = eng.white_noise((F - 1) * 256 + 2048)
pitch = os.environ.get("SGX_AB_PITCH16K") == "1"
big = torch.empty(F * (2048 if pitch else 2047) * 2, dtype=torch.float32, device="cuda")
out = big[:F * 2047 * 2].view(F, 1, 2047, 2)


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


fill = timeit(lambda: big.zero_(), 5)
ms = timeit(lambda: eng.stft_batch(pcm, out=out), 40)
print(f"{os.path.basename(os.environ.get('SGX_LIB', 'libsgx.so'))}: {ms:.3f} ms per 1e6 frames -> {F / ms / 1e3:.1f} M frames/s = "
      f"{F * 17400 / ms / 1e6 / 8000:.3f} of 8 TB/s; device fill {big.numel() * 4 / fill / 1e6:.0f} GB/s", flush=True)
