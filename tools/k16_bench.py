#!/usr/bin/env python3
"""BASELINE config 4 alone (16384-point, hop 512, 8 interleaved channels): ms per launch, median of 9, for A/B builds under tools/ab.sh.
usage: [hop positions] [planes]   (planes: SGX_FLAG_CHANNEL_PLANES for the 8-channel stream)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spectrogram_rs_amd import SpectrogramEngine

HOPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
PLANES = len(sys.argv) > 2 and sys.argv[2] == "planes"
for ch in ((8,) if PLANES else (8, 2, 1)):
    eng = SpectrogramEngine(48000.0, window_samples=8192, hop_samples=512, channels=ch, channel_planes=PLANES)
    pcm = eng.white_noise((HOPS - 1) * 512 + 8192)
    out = torch.empty((HOPS, eng.pairs, 8191, 2), dtype=torch.float32, device="cuda")
    for _ in range(3):
        eng.stft_batch(pcm, out=out)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(9)]
    for a, b in evs:
        a.record(); eng.stft_batch(pcm, out=out); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    byts = HOPS * (512 * ch * 4 + eng.pairs * 8191 * 8)
    print(f"W 8192 / H 512, {ch} channel(s), {HOPS} hop positions: {ts[4]:.3f} ms (min {ts[0]:.3f}) = {HOPS * eng.pairs / ts[4] / 1e3:.2f} M transforms/s, "
          f"{byts / ts[4] / 1e6 / 8000:.3f} of the HBM peak   checksum {eng.checksum(out[:256]):016x}", flush=True)
    eng.close()
