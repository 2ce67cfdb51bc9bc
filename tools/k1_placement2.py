#!/usr/bin/env python3
"""Placement statistics for K1's output buffer: N fresh allocations in one process (optionally keeping the earlier ones alive)."""
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


def k1(out):
    return timeit(lambda: eng.stft_batch(pcm, out=out), 10)


mode = sys.argv[1] if len(sys.argv) > 1 else "keep"
row_floats = 2 * (2048 if os.environ.get("SGX_AB_PITCH16K") == "1" else 2047)    # the pitch variant of the A/B writes 16 384 B per row
keep, res = [], []
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    big = torch.empty(F * row_floats, dtype=torch.float32, device="cuda")[:F * 2047 * 2]
    ms = k1(big.view(F, 1, 2047, 2))
    res.append(ms)
    print(f"{os.path.basename(os.environ.get('SGX_LIB', 'libsgx.so'))} {mode} {i}: ptr {big.data_ptr():#x} K1 {ms:.3f} ms = {F * 17400 / ms / 1e6 / 8000:.3f}", flush=True)
    if mode == "keep":
        keep.append(big)
    else:
        del big
        torch.cuda.empty_cache()
# re-measure the kept buffers in reverse order: is the rate a property of the buffer or of the moment?
for i in reversed(range(len(keep))):
    ms = k1(keep[i].view(F, 1, 2047, 2))
    print(f"again {i}: K1 {ms:.3f} ms (was {res[i]:.3f})", flush=True)
