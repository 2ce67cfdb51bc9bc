#!/usr/bin/env python3
"""The output-buffer placement study of profiles/r03_k1_slow_box.txt as ONE tool: does K1's rate (config 2: 1e6 mono frames, 16.4 GB
of output) depend on WHERE that output lies?  One process, one device, the library named by SGX_LIB.

  k1_placement.py buffers            six exact-size allocations kept alive, three allocations after a free, a row-rounded size,
                                     start offsets of 8 B ... 2 MiB: address, the runtime's fill rate and K1's time on each
  k1_placement.py fresh [keep|free] [n]   n fresh allocations (kept alive, or freed one by one), then the kept ones again in
                                     reverse order: is the rate a property of the buffer or of the moment?
  k1_placement.py offsets [pool GiB] K1 at every GiB of one big allocation, forward and reverse, then a 1 GiB window (65 536
                                     frames) at every GiB: a local speed map

bench.py's place_output is the production form of `fresh`: candidates timed hot and interleaved, two passes in opposite order."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spectrogram_rs_amd import SpectrogramEngine

F = 1_000_000
eng = SpectrogramEngine(48000.0, window_samples=2048, hop_samples=256, channels=1)
pcm = eng.white_noise((F - 1) * 256 + 2048)
frac = lambda ms, frames=F: frames * 17400 / ms / 1e6 / 8000


def median_ms(fn, reps, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


def k1(out, stream=pcm, reps=10):
    return median_ms(lambda: eng.stft_batch(stream, out=out), reps)


def buffers():
    def probe(tag, nfloats, offset_floats=0):
        big = torch.empty(nfloats + offset_floats, dtype=torch.float32, device="cuda")
        flat = big[offset_floats:offset_floats + F * 2047 * 2]
        fill = median_ms(lambda: flat.zero_(), 5)
        ms = k1(flat.view(F, 1, 2047, 2), reps=20)
        p = flat.data_ptr()
        print(f"{tag}: ptr {p:#x} (mod 2 MiB {p % (1 << 21):#x}, mod 1 GiB {p % (1 << 30):#x})  fill {flat.numel() * 4 / fill / 1e6:.0f} GB/s  "
              f"K1 {ms:.3f} ms = {frac(ms):.3f} of 8 TB/s", flush=True)
        return big

    keep = [probe(f"alloc {i} (exact size, kept)", F * 2047 * 2) for i in range(6)]
    del keep
    torch.cuda.empty_cache()
    for tag, n, off in [(f"alloc after free {i}", F * 2047 * 2, 0) for i in range(3)] + [("size rounded up to 16 384 B per row", F * 2048 * 2, 0)] \
            + [(f"start offset {o * 4} B", F * 2047 * 2, o) for o in (2, 16, 32, 128, 1024, 1 << 19)]:
        b = probe(tag, n, off)
        del b
        torch.cuda.empty_cache()


def fresh(mode="keep", n=12):
    keep, res = [], []
    for i in range(n):
        big = torch.empty(F * 2047 * 2, dtype=torch.float32, device="cuda")
        res.append(k1(big.view(F, 1, 2047, 2)))
        print(f"{os.path.basename(os.environ.get('SGX_LIB', 'libsgx.so'))} {mode} {i}: ptr {big.data_ptr():#x} K1 {res[-1]:.3f} ms = {frac(res[-1]):.3f}", flush=True)
        if mode == "keep":
            keep.append(big)
        else:
            del big
            torch.cuda.empty_cache()
    for i in reversed(range(len(keep))):
        print(f"again {i}: K1 {k1(keep[i].view(F, 1, 2047, 2)):.3f} ms (was {res[i]:.3f})", flush=True)


def offsets(pool_gib=96):
    pool = torch.empty(pool_gib << 28, dtype=torch.float32, device="cuda")
    print(f"pool {pool_gib} GiB at {pool.data_ptr():#x}", flush=True)
    need, step = F * 2047 * 2, 1 << 28      # floats; 1 GiB
    offs = list(range(0, pool.numel() - need + 1, step))
    for name, order in (("forward", offs), ("reverse", offs[::-1])):
        print(name, flush=True)
        for off in order:
            ms = k1(pool[off:off + need].view(F, 1, 2047, 2), reps=4)
            print(f"offset {off * 4 / 2**30:8.3f} GiB: {ms:.3f} ms = {frac(ms):.3f}", flush=True)
    Fs = 65536
    pcm_s, need = pcm[: (Fs - 1) * 256 + 2048], Fs * 2047 * 2
    print("1 GiB windows", flush=True)
    for off in range(0, pool.numel() - need + 1, step):
        ms = k1(pool[off:off + need].view(Fs, 1, 2047, 2), stream=pcm_s, reps=4)
        print(f"offset {off * 4 / 2**30:8.3f} GiB: {ms * 1e3:.1f} us = {frac(ms, Fs):.3f}", flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "buffers"
    if what == "buffers":
        buffers()
    elif what == "fresh":
        fresh(sys.argv[2] if len(sys.argv) > 2 else "keep", int(sys.argv[3]) if len(sys.argv) > 3 else 12)
    elif what == "offsets":
        offsets(int(sys.argv[2]) if len(sys.argv) > 2 else 96)
    else:
        sys.exit(__doc__)
