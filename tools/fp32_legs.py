#!/usr/bin/env python3
"""The legs whose executed f32 operations bench.py quotes as `fp32_frac` ("report both" -- the HBM fraction and the FP32
vector fraction), run under `rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32`
(tools/pmc_flops.sh).  Every leg launches its kernel a DIFFERENT number of times, so that tools/pmc_flops_summary.py can tell the legs
apart in the dispatch list (config 3's cosine and cubic legs are the same kernel instantiation).  Writes the plan it ran to
gpurun_out/fp32_legs_plan.json."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spectrogram_rs_amd import SpectrogramEngine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, R = 2048, 256, 1024
F = 200_000
plan = []


def leg(name, unit, units, launches, eng, call):
    for _ in range(launches):
        call()
    torch.cuda.synchronize()
    plan.append({"leg": name, "unit": unit, "units_per_launch": units, "launches": launches, "stft_kernel": eng.info.stft_kernel})
    eng.close()


mono = SpectrogramEngine(48000.0, window_samples=W, hop_samples=H, channels=1)
pcm = mono.white_noise((F - 1) * H + W)
out = torch.empty((F, 1, W - 1, 2), dtype=torch.float32, device="cuda")
leg("config2_stft", "frame", F, 2, mono, lambda: mono.stft_batch(pcm, out=out))
rgba = torch.empty((F, 1, R, 4), dtype=torch.uint8, device="cuda")
cos = SpectrogramEngine(48000.0, window_samples=W, hop_samples=H, channels=1, interp=1, gradient="viridis")
leg("config3_cosine", "frame", F, 3, cos, lambda: cos.render_batch(pcm, out=rgba))
cub = SpectrogramEngine(48000.0, window_samples=W, hop_samples=H, channels=1, interp=0, gradient="viridis")
leg("config3_cubic", "frame", F, 4, cub, lambda: cub.render_batch(pcm, out=rgba))
del pcm
st = SpectrogramEngine(48000.0, window_samples=W, hop_samples=H, channels=2)
pcm2 = st.white_noise((F - 1) * H + W)
leg("stereo4096", "frame", F, 5, st, lambda: st.stft_batch(pcm2, out=out))
del pcm2, out, rgba
HOPS = 20_000
c4 = SpectrogramEngine(48000.0, window_samples=8192, hop_samples=512, channels=8)
pcm8 = c4.white_noise((HOPS - 1) * 512 + 8192)
out4 = torch.empty((HOPS, 4, 8191, 2), dtype=torch.float32, device="cuda")
leg("config4", "hop position", HOPS, 6, c4, lambda: c4.stft_batch(pcm8, out=out4))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "fp32_legs_plan.json"), "w") as f:
    json.dump(plan, f)
print("fp32 legs done:", [(p["leg"], p["launches"]) for p in plan])
