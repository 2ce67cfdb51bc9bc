#!/usr/bin/env python3
"""The checksum BASELINE config 5's root must arrive at, computed by ONE process on one GPU: the same stream, the same fused kernel,
chunk by chunk, sgx_checksum_add with global word indices.  usage: tools/config5_single.py <total frames> [chunk]
(tests/test_gpu_config5.py imports single_process_checksum)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

W, H, R = 2048, 256, 1024


def single_process_checksum(total, chunk=65536):
    import torch

    from spectrogram_rs_amd import SpectrogramEngine

    eng = SpectrogramEngine(48000.0, window_samples=W, hop_samples=H, channels=1, interp=1, gradient="viridis")
    acc = torch.zeros(1, dtype=torch.int64, device="cuda")
    buf = torch.empty((chunk, 1, R, 4), dtype=torch.uint8, device="cuda")
    pcm = torch.empty((chunk - 1) * H + W, dtype=torch.float32, device="cuda")
    for f0 in range(0, total, chunk):
        n = min(chunk, total - f0)
        ns = (n - 1) * H + W
        eng.white_noise(ns, first=f0 * H, out=pcm)
        eng.render_batch(pcm[:ns], max_frames=n, out=buf[:n])
        eng.checksum_add(buf[:n].view(-1, R, 4), acc, base_word=f0 * R)
    torch.cuda.synchronize()
    value = int(acc[0]) & (2**64 - 1)
    eng.close()
    return value


if __name__ == "__main__":
    print(single_process_checksum(int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 65536))
