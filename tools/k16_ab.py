#!/usr/bin/env python3
"""BASELINE config 4 (16384-point, hop 512, 8 interleaved channels), the two 16384-point kernels in ONE process, interleaved rounds
(stft16384_w.hip): ms per launch per round, median and min; the same bytes within the
tolerance (different decompositions round differently), each against the float64 truth on sampled rows.
usage: tools/k16_ab.py [hop positions] [rounds] [channels]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from spectrogram_rs_amd import SpectrogramEngine

HOPS = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 7
CH = int(sys.argv[3]) if len(sys.argv) > 3 else 8
W, H = 8192, 512
# (round 6 compared the lane-quad kernel of rounds 3-5 with the 32 x 32 x 16 one here, in one process: profiles/r06_k16.txt; the former is gone --
# what is left is the one kernel against another BUILD of it: K16_AB_LIB=spectrogram_rs_amd/ab/<variant>.so, tools/build_variant.sh)
engs = {"wide (512 thr x 32 pts)": SpectrogramEngine(48000.0, window_samples=W, hop_samples=H, channels=CH)}
print({k: e.info.stft_kernel for k, e in engs.items()})
e0 = next(iter(engs.values()))
pcm = e0.white_noise((HOPS - 1) * H + W)
out = torch.empty((HOPS, e0.pairs, W - 1, 2), dtype=torch.float32, device="cuda")
times = {k: [] for k in engs}
sums = {}
for k, e in engs.items():
    for _ in range(2):
        e.stft_batch(pcm, out=out)
torch.cuda.synchronize()
for r in range(ROUNDS):
    for k, e in (list(engs.items()) if r % 2 == 0 else list(engs.items())[::-1]):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            e.stft_batch(pcm, out=out)
        b.record()
        torch.cuda.synchronize()
        times[k].append(a.elapsed_time(b) / 3)
byts = HOPS * (H * CH * 4 + e0.pairs * (W - 1) * 8)
for k, v in times.items():
    s = sorted(v)
    print(f"{k:28s} {HOPS} hop positions x {CH} ch: median {s[len(s) // 2]:.3f} ms  min {s[0]:.3f}  max {s[-1]:.3f}   "
          f"{HOPS / s[len(s) // 2] / 1e3:.3f} M hop positions/s = {byts / s[len(s) // 2] / 1e6 / 8000:.4f} of the HBM peak", flush=True)
# agreement on sampled rows (and against the float64 truth of the reference's f32-windowed frame)
import oracle
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import mags_error
rows = {}
for k, e in engs.items():
    e.stft_batch(pcm, out=out)
    rows[k] = np.stack([out[h, p].cpu().numpy() for h, p in [(0, 0), (HOPS // 2, e0.pairs - 1), (HOPS - 1, 0)]])
p2 = pcm.view(-1, CH) if CH > 1 else pcm.view(-1, 1)
truth = []
for h, p in [(0, 0), (HOPS // 2, e0.pairs - 1), (HOPS - 1, 0)]:
    x = p2[h * H:h * H + W].cpu().numpy()
    lr = np.stack([x[:, 0], x[:, 0]], 1) if CH == 1 else x[:, 2 * p:2 * p + 2]
    truth.append(oracle.np_truth_frame(lr, W))
truth = np.stack(truth)
for k in rows:
    print(f"{k:28s} worst error / allowance against the float64 truth: floor 0.012: {mags_error(rows[k], truth, 0.012):.3f}   floor 0.02: {mags_error(rows[k], truth, 0.02):.3f}")
