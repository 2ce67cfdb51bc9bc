#!/usr/bin/env python3
"""Quick device-side timing of the STFT / render kernels (development aid; bench.py is the contract)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spectrogram_rs_amd import SpectrogramEngine


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=200_000)
    ap.add_argument("--generic", action="store_true")
    ap.add_argument("--render", action="store_true")
    ap.add_argument("--walk", action="store_true", help="fused pixel kernel: walk the LUT thresholds (the first version) instead of seed + one compare pair")
    ap.add_argument("--interp", type=int, default=0)
    args = ap.parse_args()
    F = args.frames
    for ch in (1, 2):
        eng = SpectrogramEngine(48000.0, window_samples=2048, hop_samples=256, channels=ch, force_generic=args.generic,
                                gradient="viridis", lut_walk=args.walk, interp=args.interp)
        n = (F - 1) * eng.H + eng.W
        pcm = eng.white_noise(n)
        out = torch.empty((F, 1, eng.M, 2), dtype=torch.float32, device="cuda")
        med, best = timeit(lambda: eng.stft_batch(pcm, out=out))
        byts = F * (eng.H * ch * 4 + eng.M * 8)
        print(f"stft ch={ch} kernel={eng.info.stft_kernel} F={F}: median {med:.3f} ms best {best:.3f} ms -> "
              f"{F / med / 1e3:.1f} M frames/s, {byts / med / 1e6:.1f} GB/s algorithmic", flush=True)
        if args.render:
            rg = torch.empty((F, 1, eng.R, 4), dtype=torch.uint8, device="cuda")
            med, best = timeit(lambda: eng.render_batch(pcm, out=rg), iters=5)
            print(f"render ch={ch}: median {med:.3f} ms -> {F / med / 1e3:.1f} M frames/s", flush=True)
            med, best = timeit(lambda: eng.render_mags(out[:, 0], out=rg[:, 0]), iters=5)
            print(f"render_mags only: median {med:.3f} ms -> {F / med / 1e3:.1f} M columns/s", flush=True)


if __name__ == "__main__" and not any(f in sys.argv for f in ("--extra", "--live", "--generic-sizes", "--config4", "--others")):
    main()


def config4(frames=20000, channels=8):
    """BASELINE config 4: 16384-point STFT, hop 512, 8 interleaved channels (4 pairs)"""
    eng = SpectrogramEngine(48000.0, window_samples=8192, hop_samples=512, channels=channels)
    n = (frames - 1) * eng.H + eng.W
    pcm = eng.white_noise(n)
    out = torch.empty((frames, eng.pairs, eng.M, 2), dtype=torch.float32, device="cuda")
    med, best = timeit(lambda: eng.stft_batch(pcm, out=out), iters=5)
    byts = frames * (512 * channels * 4 + eng.pairs * eng.M * 8)
    tr = eng.pairs * frames * (1 if channels > 1 else 0.5)
    print(f"config4 ch={channels} kernel={eng.info.stft_kernel} hops={frames}: median {med:.3f} ms best {best:.3f} -> {frames / med / 1e3:.3f} M hop positions/s "
          f"({tr / med / 1e3:.3f} M transforms/s), {byts / med / 1e6:.1f} GB/s algorithmic = {byts / med / 1e6 / 8000:.3f} of 8 TB/s", flush=True)
    return out


def app_default(frames=20000):
    """the app's own operating point: 0.05 s window at 48 kHz (W 2400: 4800-point mixed radix), hop 93, stereo"""
    eng = SpectrogramEngine(48000.0, period=0.05, stride=2.0 / 1024, channels=2)
    n = (frames - 1) * eng.H + eng.W
    pcm = eng.white_noise(n)
    out = torch.empty((frames, 1, eng.M, 2), dtype=torch.float32, device="cuda")
    med, best = timeit(lambda: eng.stft_batch(pcm, out=out), iters=5)
    print(f"app default W={eng.W} H={eng.H} kernel={eng.info.stft_kernel}: median {med:.3f} ms -> {frames / med / 1e3:.3f} M frames/s "
          f"(real time is {48000 / eng.H:.0f} frames/s)", flush=True)


if __name__ == "__main__" and "--config4" in sys.argv:
    for ch in ((2,) if "--stereo-only" in sys.argv else (8,) if "--ch8-only" in sys.argv else (8, 2, 1)):
            del a, b
            continue
        config4(channels=ch)

if __name__ == "__main__" and "--extra" in sys.argv:
    config4()
    app_default()


def f16(frames=1_000_000):
    eng = SpectrogramEngine(48000.0, window_samples=2048, hop_samples=256, channels=1)
    n = (frames - 1) * eng.H + eng.W
    pcm = eng.white_noise(n)
    out = torch.empty((frames, 1, eng.M, 2), dtype=torch.float16, device="cuda")
    med, best = timeit(lambda: eng.stft_batch_f16(pcm, out=out))
    byts = frames * (256 * 4 + eng.M * 4)
    print(f"stft f16 ring rows F={frames}: median {med:.3f} ms -> {frames / med / 1e3:.1f} M frames/s, {byts / med / 1e6:.1f} GB/s algorithmic", flush=True)


if __name__ == "__main__" and "--extra" in sys.argv:
    f16()


def live_ticks(ticks=600):
    """the reference's real-time operating point: 60 GUI ticks/s, 800 new stereo samples per tick, 0.05 s window,
    hop 93 -- host buffer in, host buffer out, per-tick wall time (PCIe both ways included)"""
    import time

    import numpy as np

    eng = SpectrogramEngine(48000.0, period=0.05, stride=2.0 / 1024, channels=2, gradient="magma")
    rng = np.random.default_rng(0)
    chunk = rng.uniform(-1, 1, (800, 2)).astype(np.float32)
    for what in ("mags", "mags_f16", "rgba"):
        ring = eng.live(4096)
        ring.push(rng.uniform(-1, 1, (2400, 2)).astype(np.float32), 2)
        ring.tick(what)
        times, frames = [], 0
        for _ in range(ticks):
            ring.push(chunk, 2)
            t0 = time.perf_counter()
            frames += len(ring.tick(what, max_frames=32))
            times.append(time.perf_counter() - t0)
        t = np.array(times[20:]) * 1e6
        print(f"live tick [{what}] W={eng.W} H={eng.H}: {frames / ticks:.1f} frames/tick, median {np.median(t):.0f} us, "
              f"p99 {np.percentile(t, 99):.0f} us per tick (the tick period at 60 Hz is 16667 us)", flush=True)


if __name__ == "__main__" and "--live" in sys.argv:
    live_ticks()


def generic_sizes(frames=200_000):
    """the generic power-of-two kernel at other window sizes (mono and stereo)"""
    for W in (512, 1024, 4096):
        for ch in (1, 2):
            eng = SpectrogramEngine(48000.0, window_samples=W, hop_samples=W // 8, channels=ch)
            F = frames * 2048 // W
            pcm = eng.white_noise((F - 1) * eng.H + eng.W)
            out = torch.empty((F, 1, eng.M, 2), dtype=torch.float32, device="cuda")
            med, best = timeit(lambda: eng.stft_batch(pcm, out=out), iters=5)
            print(f"generic W={W} ch={ch} kernel={eng.info.stft_kernel} F={F}: median {med:.3f} ms -> {F / med / 1e3:.1f} M frames/s, "
                  f"{F * (eng.H * ch * 4 + eng.M * 8) / med / 1e6:.0f} GB/s algorithmic", flush=True)


if __name__ == "__main__" and "--generic-sizes" in sys.argv:
    generic_sizes()


def others(frames=100_000):
    """the kernels beside the tuned ones: mixed radix (the app's window), chirp-z, generic power of two"""
    sizes = ((2400, 93, 2), (2400, 93, 1), (2205, 86, 2), (1102, 100, 2), (1024, 128, 2), (1024, 128, 1), (512, 64, 2))
    if "--rates" in sys.argv:   # 0.05 s at 8 / 16 / 32 / 88.2 / 96 / 192 kHz
        sizes = ((400, 16, 2), (800, 31, 2), (1600, 62, 2), (4410, 172, 2), (4800, 187, 2), (9600, 375, 2))
    for W, H, ch in sizes:
        eng = SpectrogramEngine(48000.0, window_samples=W, hop_samples=H, channels=ch)
        pcm = eng.white_noise((frames - 1) * eng.H + eng.W)
        out = torch.empty((frames, 1, eng.M, 2), dtype=torch.float32, device="cuda")
        med, best = timeit(lambda: eng.stft_batch(pcm, out=out), iters=5)
        print(f"W={W} H={H} ch={ch} kernel={eng.info.stft_kernel} F={frames}: median {med:.3f} ms best {best:.3f} -> {frames / med / 1e3:.1f} M frames/s", flush=True)


if __name__ == "__main__" and "--others" in sys.argv:
    others()
