#!/usr/bin/env python3
"""Config-3 timing of the fused PCM -> RGBA kernel (development aid; bench.py is the contract): mono 1e6 frames, cosine and
cubic, and an (l, r) stream; SGX_LIB=<other build> swaps the library for a same-device A/B (BENCH=tools/pixel_bench.py tools/ab.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spectrogram_rs_amd import SpectrogramEngine

F = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 7


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


for ch, interp, name in ((1, 1, "mono cosine"), (1, 0, "mono cubic"), (2, 0, "stereo cubic")):
    eng = SpectrogramEngine(48000.0, window_samples=2048, hop_samples=256, channels=ch, interp=interp, gradient="viridis")
    pcm = eng.white_noise((F - 1) * 256 + 2048)
    out = torch.empty((F, 1, 1024, 4), dtype=torch.uint8, device="cuda")
    ms = timeit(lambda: eng.render_batch(pcm, out=out))
    print(f"{name}: {ms:.3f} ms per {F} frames -> {F / ms / 1e3:.1f} M frames/s (render_path {eng.info.render_path}, checksum {eng.checksum(out[:4096]):016x})", flush=True)
    del out, pcm
    eng.close()
