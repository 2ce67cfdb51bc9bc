#!/usr/bin/env python3
"""Does K1's rate depend on WHERE its 16.4 GB output buffer lies?  Several buffers in one process, one device: fill rate (zero_()) and
the K1 launch time on each, with the buffer's address (profiles/r03_k1_slow_box.txt)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spectrogram_rs_amd import SpectrogramEngine

F = 1_000_000
eng = SpectrogramEngine(48000.0, window_samples=2048, hop_samples=256, channels=1)
pcm = eng.white_noise((F - 1) * 256 + 2048)


def timeit(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


def probe(tag, nfloats, offset_floats=0):
    big = torch.empty(nfloats + offset_floats, dtype=torch.float32, device="cuda")
    flat = big[offset_floats:offset_floats + F * 2047 * 2]
    out = flat.view(F, 1, 2047, 2)
    fill = timeit(lambda: flat.zero_(), 5)
    ms = timeit(lambda: eng.stft_batch(pcm, out=out), 20)
    p = out.data_ptr()
    print(f"{tag}: ptr {p:#x} (mod 2 MiB {p % (1 << 21):#x}, mod 1 GiB {p % (1 << 30):#x})  fill {flat.numel() * 4 / fill / 1e6:.0f} GB/s  "
          f"K1 {ms:.3f} ms = {F * 17400 / ms / 1e6 / 8000:.3f} of 8 TB/s", flush=True)
    return big


keep = []
for i in range(6):
    keep.append(probe(f"alloc {i} (exact size, kept)", F * 2047 * 2))
del keep
torch.cuda.empty_cache()
for i in range(3):
    b = probe(f"alloc after free {i}", F * 2047 * 2)
    del b
    torch.cuda.empty_cache()
b = probe("size rounded up to 16 384 B per row", F * 2048 * 2)
del b
torch.cuda.empty_cache()
for off in (2, 16, 32, 128, 1024, 1 << 19):
    b = probe(f"start offset {off * 4} B", F * 2047 * 2, off)
    del b
    torch.cuda.empty_cache()
