#!/usr/bin/env python3
"""Device-side timing of sgx_render_mags (the pixel stage alone) across window sizes and colour rules (development aid)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectrogram_rs_amd import SpectrogramEngine  # noqa: E402
from tools.quick_bench import timeit  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "gradients.npz")
CASES = ((2048, 256, 1, 200_000, "viridis"), (2048, 256, 2, 200_000, "viridis"), (2400, 93, 2, 100_000, "viridis"),
         (2205, 86, 2, 100_000, "viridis"), (8192, 512, 2, 20_000, "viridis"), (2400, 93, 2, 100_000, "spectral"),
         (2400, 93, 2, 100_000, "plasma_stereo"))
for W, H, ch, F, grad in CASES:
    eng = SpectrogramEngine(48000.0, window_samples=W, hop_samples=H, channels=ch, gradient="viridis",
                            interp=1 if "--cosine" in sys.argv else 0)
    if grad == "plasma_stereo":
        eng.set_gradient(np.load(GOLDEN)["plasma"], stereo=True)
    elif grad != "viridis":
        eng.set_builtin_scheme(grad, stereo=True)
    pcm = eng.white_noise((F - 1) * eng.H + eng.W)
    out = eng.stft_batch(pcm)
    rg = torch.empty((F, eng.R, 4), dtype=torch.uint8, device="cuda")
    med, _ = timeit(lambda: eng.render_mags(out[:, 0], out=rg), iters=5)
    print(f"W={W} ch={ch} {grad}: render_mags {med:.3f} ms per {F} columns -> {F / med / 1e3:.1f} M columns/s, "
          f"reads {F * eng.M * 8 / med / 1e6:.0f} GB/s", flush=True)
