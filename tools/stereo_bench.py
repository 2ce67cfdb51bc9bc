#!/usr/bin/env python3
"""4096-point rows at hop 256 for the stream shapes that do not slide a register window: (l, r) stream, independent mono frames, mono
pairs at hop 255, half rows of an (l, r) stream, fused (l, r) pixels; plus the headline mono path as the control.  usage: [frames]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spectrogram_rs_amd import SpectrogramEngine

F = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000


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


out = torch.empty((F, 1, 2047, 2), dtype=torch.float32, device="cuda")
for name, kw, H in (("mono, default (real-input kernel)", dict(channels=1), 256), ("mono pairs, hop 256", dict(channels=1, paired_frames=True), 256), ("(l, r) stream", dict(channels=2), 256),
                    ("independent mono frames", dict(channels=1), 256), ("mono pairs, hop 255", dict(channels=1), 255)):
    eng = SpectrogramEngine(48000.0, window_samples=2048, hop_samples=H, gradient="viridis", **kw)
    pcm = eng.white_noise((F - 1) * H + 2048)
    ms = timeit(lambda: eng.stft_batch(pcm, out=out))
    line = f"{name:32s} rows {ms:.3f} ms = {F / ms / 1e3:.1f} M frames/s  checksum {eng.checksum(out[:2048]):016x}"
    if kw.get("channels") == 2:
        h = torch.empty((F, 1, 2047, 2), dtype=torch.float16, device="cuda")
        ms16 = timeit(lambda: eng.stft_batch_f16(pcm, out=h))
        rg = torch.empty((F, 1, 1024, 4), dtype=torch.uint8, device="cuda")
        msp = timeit(lambda: eng.render_batch(pcm, out=rg))
        line += f"   half rows {ms16:.3f} ms   pixels {msp:.3f} ms = {F / msp / 1e3:.1f} M frames/s"
        del h, rg
    print(line, flush=True)
    del pcm
    eng.close()
