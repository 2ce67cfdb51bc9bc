#!/usr/bin/env python3
"""Regenerate the committed golden fixtures (run from the repo root: python tests/golden/make_fixtures.py).

The reference holds no golden vectors for this path (no tests at all), so these are the build's
own pins.  What each file pins and where its numbers come from:

  gradients.npz          the four 256-entry ramps               tools/gen_gradients.py (matplotlib)
  config1_sweep.npz      BASELINE config 1: two frames of the 20 Hz -> 20 kHz sweep (mono, W 2048)
                         input f32; expected magnitudes in f64 from numpy.fft on the f32-windowed
                         input (independent of oracle/spectro_oracle.c)
  noise_frames.npz       8 frames of the counter-based white noise (frame t = i*977), same truth
  bin_edges_1024.npy     1025 row edges, f64 math in pure Python (math.log / math.exp), cast f32
  row_counts_1024.npy    per-row magnitude_in sample counts at config A (M 2047, 48 kHz)
  rgba_columns.npz       RGBA columns of the above frames from the C oracle (regression pins for
                         cubic + cosine, Viridis) -- these pin the restatement against drift, not
                         against the reference
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def truth_frame(lr, W, win):
    z = (lr[:, 0] * win).astype(np.float64) + 1j * (lr[:, 1] * win).astype(np.float64)
    P = 2 * W
    F = np.fft.fft(np.concatenate([z, np.zeros(W, np.complex128)]))
    k = np.arange(1, W)
    a, b = F[k], F[P - k]
    return np.stack([np.abs(a + np.conj(b)) / 2.0, np.abs(a - np.conj(b)) / 2.0], axis=1) * (2.0 / W)


def main():
    import oracle

    W, H, SR, R = 2048, 256, 48000, 1024
    win = oracle.hann_window(W)  # libm cosf, f32 op order of fft.rs:61

    # config 1: sweep, frames at n = 0 and n = 24000
    sweep_in = np.stack([oracle.sine_sweep(W, 0), oracle.sine_sweep(W, 24000)])
    sweep_truth = np.stack([truth_frame(np.stack([s, s], 1), W, win) for s in sweep_in])
    np.savez_compressed(os.path.join(GOLD, "config1_sweep.npz"), input=sweep_in, first=np.array([0, 24000]),
                        expected_f64=sweep_truth, window=win)

    # noise frames t = i * 977
    ts = np.array([(i * 977) % 1_000_000 for i in range(8)], dtype=np.int64)
    noise_in = np.stack([oracle.white_noise(W, int(t) * H) for t in ts])
    noise_truth = np.stack([truth_frame(np.stack([s, s], 1), W, win) for s in noise_in])
    np.savez_compressed(os.path.join(GOLD, "noise_frames.npz"), input=noise_in, frame_index=ts, expected_f64=noise_truth)

    # bin edges: pure-Python f64
    lo, hi = math.log(32.0), math.log(22030.0)
    edges = np.array([np.float32(math.exp((hi - lo) * (p / R) + lo)) for p in range(R + 1)], np.float32)
    np.save(os.path.join(GOLD, "bin_edges_1024.npy"), edges)
    counts = np.array([oracle.num_samples_in(W - 1, SR, float(edges[i]), float(edges[i + 1])) for i in range(R)], np.uint32)
    np.save(os.path.join(GOLD, "row_counts_1024.npy"), counts)

    # RGBA regression columns from the oracle
    g = np.load(os.path.join(GOLD, "gradients.npz"))
    frames = np.concatenate([sweep_in, noise_in[:2]])
    mags = np.stack([oracle.fft_process(np.stack([s, s], 1), W) for s in frames])
    out = {"mags": mags}
    for name, interp in (("cubic", oracle.INTERP_CUBIC), ("cosine", oracle.INTERP_COSINE)):
        out["viridis_" + name] = oracle.render_columns(mags, SR, g["viridis"], interp=interp)
    np.savez_compressed(os.path.join(GOLD, "rgba_columns.npz"), **out)
    print("fixtures written to", GOLD)


if __name__ == "__main__":
    main()
