#!/usr/bin/env python3
"""The mono modes of the 4096-point path on ONE output buffer, hot and interleaved: default (real-input kernel, every frame its own
transform), paired (two frames per transform), complex ((s, s) transform per frame).  Each mode runs back to back for `--seconds`, the
modes take turns for `--rounds` rounds; the figure of a round is the mean launch time of the last two thirds of its window.
usage: [--frames N] [--seconds S] [--rounds R] [--modes default,paired,complex]     (SGX_LIB=<other build> for a same-device A/B)"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spectrogram_rs_amd import SpectrogramEngine

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=1_000_000)
ap.add_argument("--seconds", type=float, default=0.7)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--modes", default="default,paired")
ap.add_argument("--hop", type=int, default=256)
ap.add_argument("--pixels", action="store_true", help="the fused PCM -> RGBA path (cosine, Viridis) instead of float rows")
args = ap.parse_args()
F = args.frames
FLAGS = {"default": {}, "paired": dict(paired_frames=True), "complex": dict(complex_mono=True)}
engs = {m: SpectrogramEngine(48000.0, window_samples=2048, hop_samples=args.hop, channels=1, interp=1, gradient="viridis", **FLAGS[m]) for m in args.modes.split(",")}
first = next(iter(engs.values()))
pcm = first.white_noise((F - 1) * args.hop + 2048)
out = torch.empty((F, 1, 1024, 4), dtype=torch.uint8, device="cuda") if args.pixels else torch.empty((F, 1, 2047, 2), dtype=torch.float32, device="cuda")
out.zero_()


def window(eng, seconds):
    run = (lambda: eng.render_batch(pcm, out=out)) if args.pixels else (lambda: eng.stft_batch(pcm, out=out))
    ts, t0 = [], time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(16)]
        for a, b in evs:
            a.record(); run(); b.record()
        torch.cuda.synchronize()
        ts += [a.elapsed_time(b) for a, b in evs]
    steady = ts[len(ts) // 3:]
    return sum(steady) / len(steady)


window(first, 1.0)     # heat
res = {m: [] for m in engs}
for _ in range(args.rounds):
    for m, e in engs.items():
        res[m].append(window(e, args.seconds))
lib = os.path.basename(os.environ.get("SGX_LIB", "libsgx.so"))
for m, v in res.items():
    print(f"{lib:28s} hop {args.hop:4d} {'pixels' if args.pixels else 'rows':6s} {m:8s} " + "  ".join(f"{x:.3f}" for x in v) + f"  ms per {F} frames   best {F / min(v) / 1e3:.1f} M frames/s", flush=True)
