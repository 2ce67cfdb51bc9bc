#!/usr/bin/env python3
"""Worst error of the GPU magnitudes against the float64 truth (numpy FFT of the reference's f32-windowed frame), per
10 dB band below the frame peak: (a) worst PURE relative error |x - t| / |t|, (b) worst absolute error / frame peak.
Development aid behind tests/test_gpu_parity.py::test_error_by_level_against_float64_truth (run on the GPU box)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import oracle
from spectrogram_rs_amd import SpectrogramEngine


def signals(W, H, frames, ch):
    n = (frames - 1) * H + W
    t = np.arange(n, dtype=np.float64) / 48000.0
    out = {}
    noise = oracle.white_noise(n * ch, seed=77)
    out["white noise"] = noise
    tone = (0.5 * np.sin(2 * np.pi * 997.0 * t)).astype(np.float32)
    quiet = (1e-4 * oracle.white_noise(n, seed=3)).astype(np.float32)
    s = (tone + quiet).astype(np.float32)
    out["tone + noise at -74 dB"] = np.repeat(s, ch) if ch > 1 else s
    sw = oracle.sine_sweep(n)
    out["sweep"] = np.repeat(sw, ch) if ch > 1 else sw
    return out


def bands(got, truth):
    peak = np.abs(truth).max(axis=(1, 2), keepdims=True)
    level = 20 * np.log10(np.maximum(np.abs(truth), 1e-300) / peak)
    rel = np.abs(got - truth) / np.maximum(np.abs(truth), 1e-300)
    ab = np.abs(got - truth) / peak
    rows = []
    for lo in range(0, 160, 10):
        m = (level <= -lo) & (level > -(lo + 10))
        if m.any():
            rows.append((lo, int(m.sum()), float(rel[m].max()), float(ab[m].max())))
    return rows


def run(name, W, H, ch, frames=24, **kw):
    eng = SpectrogramEngine(48000.0, window_samples=W, hop_samples=H, channels=ch, **kw)
    print(f"== {name}: W={W} H={H} ch={ch} kernel={eng.info.stft_kernel}")
    for sname, pcm in signals(W, H, frames, ch).items():
        got = eng.stft_batch(torch.from_numpy(pcm).cuda()).cpu().numpy().astype(np.float64)[:, 0]
        lr = pcm.reshape(-1, ch) if ch > 1 else np.stack([pcm, pcm], 1)
        truth = np.stack([oracle.np_truth_frame(lr[t * H:t * H + W, :2], W) for t in range(got.shape[0])])
        print(f"  -- {sname}")
        for lo, cnt, r, a in bands(got, truth):
            print(f"     {-lo:5d}..{-lo - 10:5d} dB  bins {cnt:8d}  worst rel {r:9.2e}  worst abs/peak {a:9.2e}")


if __name__ == "__main__" and "--chirp" in sys.argv:
    for Wt in (1102, 1852, 2731, 2732, 3001, 4519, 5003, 5461):
        run(f"KB {2 * Wt} (chirp-z)", Wt, max(Wt // 10, 1), 2, frames=6)
    sys.exit(0)

if __name__ == "__main__" and "--real" in sys.argv:   # the real-input modes of round 4 beside the (s, s) transform they replace
    for Wt, Ht in ((2400, 93), (2205, 86), (4096, 256), (1102, 43), (2048, 255)):
        run(f"W {Wt} mono default (real-input mode)", Wt, Ht, 1)
        run(f"W {Wt} mono, the (s, s) transform per frame", Wt, Ht, 1, complex_mono=True)
    sys.exit(0)

if __name__ == "__main__":
    run("K1R mono (default: every frame its own real-input transform)", 2048, 256, 1)
    run("K1 mono (frame pairs)", 2048, 256, 1, paired_frames=True)
    run("K1 mono ((s, s) transform per frame)", 2048, 256, 1, complex_mono=True)
    run("K1 stereo", 2048, 256, 2)
    run("K16 stereo (32 x 32 x 16, round 6)", 8192, 512, 2, frames=8)
    run("K16 mono (duplicated plane)", 8192, 512, 1, frames=8)
    run("KM 4800", 2400, 93, 2)
    run("KB 2204 (chirp-z)", 1102, 100, 2)
    run("KB 3704 (chirp-z)", 1852, 100, 2)
    run("K0 2048", 1024, 128, 2)
