#!/usr/bin/env python3
"""K1's three store variants (spectrogram_rs_amd/ab/{a_base,b_pitch16k,c_aligned_stage}.so) on the SAME buffers, one process, one
device: is the slow class of placements (profiles/r03_k1_slow_box.txt) a matter of store alignment?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spectrogram_rs_amd import SpectrogramEngine, _lib

F = 1_000_000
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
engines = {}
for name in ("a_base", "b_pitch16k", "c_aligned_stage"):
    _lib.LIB_PATH = os.path.join(root, "spectrogram_rs_amd", "ab", name + ".so")
    _lib._lib = None
    engines[name] = SpectrogramEngine(48000.0, window_samples=2048, hop_samples=256, channels=1)
pcm = engines["a_base"].white_noise((F - 1) * 256 + 2048)


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


keep = []
for i in range(10):
    big = torch.empty(F * 2048 * 2, dtype=torch.float32, device="cuda")     # 16 384 B per row: room for either layout
    keep.append(big)
    out = big[:F * 2047 * 2].view(F, 1, 2047, 2)
    row = [timeit(lambda e=e: e.stft_batch(pcm, out=out)) for e in engines.values()]
    fill = timeit(lambda: big.zero_(), 3)
    print(f"buffer {i} ptr {big.data_ptr():#x}: " + "  ".join(f"{n} {ms:.3f} ms" for n, ms in zip(engines, row)) +
          f"  fill {big.numel() * 4 / fill / 1e6:.0f} GB/s", flush=True)
