"""Seeded sweep over the configuration space: random window / hop / channel count / stream length / axis / dB range
/ interpolation / LUT rule, every draw checked against the CPU oracle (magnitudes within the stated tolerance,
pixel bytes bit-exact on identical magnitudes) and against the library's own invariants (one-kernel == two-kernel
pixel path, any sub-range == the same frames of the full run).  Run with -m gpu on an MI355X."""
import os

import numpy as np
import pytest

import oracle
from conftest import FLOOR_WIDE          # random windows and channel counts: the floor of every kernel that is not BASELINE's

pytestmark = pytest.mark.gpu

# SGX_FUZZ_OFFSET=k: the same sweeps over OTHER draws (seed + k) -- for a longer hunt than the suite's own draws, by hand
OFFSET = int(os.environ.get("SGX_FUZZ_OFFSET", "0"))


def draw(rng):
    kind = rng.choice(["pow2", "pow2", "odd", "smooth"])
    if kind == "pow2":
        W = int(2 ** rng.integers(2, 13))               # 4 .. 4096 (2W = the power-of-two kernels)
    elif kind == "smooth":                              # 2W = 2^a 3^b 5^c 7^d up to 20480: the mixed-radix kernel
        W = 1
        while W < 4 or W > 10240 or W & (W - 1) == 0:
            W = int(2 ** rng.integers(0, 8) * 3 ** rng.integers(0, 5) * 5 ** rng.integers(0, 4) * 7 ** rng.integers(0, 3))
    else:
        W = int(rng.integers(5, 5400))                  # chirp-z kernel: any length with 3W - 1 <= 16384
        if W & (W - 1) == 0:
            W += 1
    H = int(rng.integers(1, max(2, W + W // 2)))        # hops above W (gaps between frames) included
    channels = int(rng.choice([1, 2, 2, 4]))
    frames = int(rng.integers(0, 40))
    n = 0 if frames == 0 else (frames - 1) * H + W + int(rng.integers(0, H))   # ragged tail
    if rng.random() < 0.1:
        n = int(rng.integers(0, W))                     # shorter than one window
    sr = int(rng.choice([8000, 22050, 44100, 48000, 96000]))
    f_lo = float(rng.uniform(1.0, 200.0))
    f_hi = float(rng.uniform(f_lo * 2, sr * rng.uniform(0.3, 0.7)))
    lo_db = float(rng.uniform(-120.0, -40.0))
    return dict(W=W, H=H, channels=channels, n=n, sr=sr, rows=int(rng.integers(1, 1400)), f_min=f_lo, f_max=f_hi,
                min_db=lo_db, max_db=float(lo_db + rng.uniform(10.0, 90.0)), interp=int(rng.integers(0, 2)),
                lut=int(rng.integers(0, 2)), amp=float(10.0 ** rng.uniform(-4, 0.3)), grad=str(rng.choice(["viridis", "magma", "inferno", "plasma"])),
                diverging=bool(rng.random() < 0.3), paired=bool(rng.random() < 0.5))


@pytest.mark.parametrize("seed", range(160))
def test_random_configuration(seed, mags_err, gradients):
    import torch
    from spectrogram_rs_amd import SpectrogramEngine
    seed += OFFSET
    rng = np.random.default_rng(1000 + seed)
    c = draw(rng)
    W, H, ch, n = c["W"], c["H"], c["channels"], c["n"]
    kw = dict(window_samples=W, hop_samples=H, channels=ch, rows=c["rows"], f_min=c["f_min"], f_max=c["f_max"],
              min_db=c["min_db"], max_db=c["max_db"], interp=c["interp"], lut_index_mode=c["lut"], gradient=c["grad"],
              paired_frames=c["paired"])        # (a mono stream: two frames per transform, or -- the default -- every frame its own)
    eng = SpectrogramEngine(float(c["sr"]), **kw)
    if c["diverging"]:   # colour from the left / right balance, alpha from the level (colorscheme.rs:63-66)
        eng.set_gradient(gradients[c["grad"]], stereo=True)
    pcm = (oracle.white_noise(n * ch, seed=seed) * np.float32(c["amp"])).astype(np.float32)
    dev = torch.from_numpy(pcm).cuda()
    total = oracle.num_frames(n, W, H)
    assert eng.num_frames(n) == total
    got = eng.stft_batch(dev).cpu().numpy()
    ref = oracle.stream_process(pcm, ch, W, H, threads=8)
    assert got.shape == ref.shape == (total, max(ch // 2, 1), W - 1, 2)
    if total == 0:
        assert eng.render_batch(dev).shape[0] == 0
        return
    assert eng.info.stft_kernel in (0, 2, 4, 5, 6, 8, 9)
    if eng.info.stft_kernel == 4:
        # lengths with a large prime factor: the float32 oracle evaluates that factor as a plain O(p^2) sum in float32
        # (FFTW would not), so it is itself off by several times the tolerance there -- the reference for these sizes
        # is the oracle's float64 mode, and the chirp-z kernel is held to 1x the tolerance against it like every other
        # (conftest.chirpz_bound: 1 x, and 1.005 x at the largest convolution length only)
        from conftest import chirpz_bound
        truth = oracle.stream_process(pcm, ch, W, H, threads=8, precision=oracle.F64)
        assert mags_err(got, truth, FLOOR_WIDE) <= chirpz_bound(W), c
    else:
        # float32 against float32: each within the tolerance of the exact transform
        assert mags_err(got, ref, FLOOR_WIDE) <= 2.0, c
    # any sub-range writes the bytes of the full run
    first = int(rng.integers(0, total))
    count = int(rng.integers(1, total - first + 1))
    assert np.array_equal(eng.stft_batch(dev, first_frame=first, max_frames=count).cpu().numpy(), got[first:first + count]), c
    # pixels: bit-exact against the oracle on the engine's own magnitudes; PCM -> pixels in one call gives the same bytes
    if W >= 8:
        cols = eng.render_mags(torch.from_numpy(got).cuda().reshape(-1)).cpu().numpy()
        pick = np.unique(rng.integers(0, cols.shape[0], 4))
        want = oracle.render_columns(got.reshape(-1, W - 1, 2)[pick], c["sr"], gradients[c["grad"]], R=c["rows"], f_min=c["f_min"],
                                     f_max=c["f_max"], interp=c["interp"], min_db=c["min_db"], max_db=c["max_db"], mode=c["lut"],
                                     stereo=c["diverging"])
        assert np.array_equal(cols[pick], want), c
        assert np.array_equal(eng.render_batch(dev).cpu().numpy().reshape(cols.shape), cols), c


@pytest.mark.parametrize("seed", range(48))
def test_random_hops_and_channels_at_the_compiled_plans(seed, mags_err):
    # the lengths with compile-time plans (0.05 s at the usual rates, powers of two, two chirp-z lengths and two smooth
    # lengths on the run-time geometry for contrast): random hops (also beyond the window), 1 - 6 channels, ragged tails,
    # row counts and both interpolators; transform against the oracle, sub-range and half-row bytes, and the one-kernel
    # pixel path against the pixel stage alone on the stored magnitudes
    import torch
    from spectrogram_rs_amd import SpectrogramEngine
    seed += OFFSET
    rng = np.random.default_rng(7000 + seed)
    W = int(rng.choice([400, 800, 1600, 2205, 2400, 4410, 4800, 8820, 9600, 512, 1024, 4096, 2048, 8192, 1102, 551, 406, 1218]))
    H = int(rng.integers(1, W + W // 3))
    ch = int(rng.choice([1, 2, 2, 4, 6]))
    frames = int(rng.integers(1, 12))
    n = (frames - 1) * H + W + int(rng.integers(0, H))
    eng = SpectrogramEngine(48000.0, window_samples=W, hop_samples=H, channels=ch, rows=int(rng.integers(100, 1300)), gradient="viridis",
                            interp=int(rng.integers(0, 2)), paired_frames=bool(seed & 1))   # (mono streams: frame pairs on odd seeds)
    pcm = (oracle.white_noise(n * ch, seed=seed) * np.float32(10.0 ** rng.uniform(-3, 0))).astype(np.float32)
    dev = torch.from_numpy(pcm).cuda()
    got = eng.stft_batch(dev).cpu().numpy()
    ref = oracle.stream_process(pcm, ch, W, H, threads=8)
    assert got.shape == ref.shape and mags_err(got, ref, FLOOR_WIDE) <= 2.0
    first = int(rng.integers(0, frames))
    cnt = int(rng.integers(1, frames - first + 1))
    assert np.array_equal(eng.stft_batch(dev, first_frame=first, max_frames=cnt).cpu().numpy(), got[first:first + cnt])
    assert torch.equal(eng.stft_batch_f16(dev), torch.from_numpy(got).cuda().to(torch.float16))
    px = eng.render_batch(dev).cpu().numpy()
    own = eng.render_mags(torch.from_numpy(got).cuda().reshape(-1, eng.M, 2)).cpu().numpy()
    assert np.array_equal(px.reshape(own.shape), own)


@pytest.mark.parametrize("seed", range(24))
def test_random_mono_streams_through_the_real_input_kernel(seed, mags_err, gradients):
    # The default mono path at W 2048 / H 256 (csrc/stft4096_real.hip): random lengths (workgroups of one job, odd frame counts, a
    # partner frame past the end of the stream), random level envelopes spanning 100 dB inside one stream (every frame is held to the
    # tolerance against ITS OWN peak: frames share a workgroup, never a transform), sub-ranges from random first frames, an 8-byte and
    # a 4-byte aligned start (the same bytes), half rows and fused pixels against the same rows.
    import torch
    from spectrogram_rs_amd import SpectrogramEngine
    seed += OFFSET
    rng = np.random.default_rng(7000 + seed)
    W, H = 2048, 256
    frames = int(rng.choice([1, 2, 3, 5, int(rng.integers(6, 300)), int(rng.integers(300, 5000))]))
    n = (frames - 1) * H + W + int(rng.integers(0, H))
    pcm = oracle.white_noise(n, seed=seed)
    knots = np.sort(rng.integers(0, n, int(rng.integers(1, 12))))
    gain = np.ones(n, np.float32)
    for k in knots:                                        # level steps of up to +-50 dB each, anywhere (inside hops too)
        gain[k:] *= np.float32(10.0 ** rng.uniform(-2.5, 2.5))
    gain = np.clip(gain, 1e-5, 1.0).astype(np.float32)
    if rng.random() < 0.3:
        gain[: int(rng.integers(0, n))] = 0.0              # digital silence in front
    pcm = (pcm * gain).astype(np.float32)
    eng = SpectrogramEngine(48000.0, window_samples=W, hop_samples=H, channels=1, interp=int(rng.integers(0, 2)), gradient="magma")
    assert eng.info.render_path & 8
    buf = torch.zeros(n + 2, dtype=torch.float32, device="cuda")
    buf[2:] = torch.from_numpy(pcm).cuda()
    dev = buf[2:]
    assert dev.data_ptr() % 8 == 0
    got = eng.stft_batch(dev).cpu().numpy()
    assert got.shape == (frames, 1, W - 1, 2)
    pick = np.unique(np.concatenate([rng.integers(0, frames, 12), [0, frames - 1]]))
    truth = np.stack([oracle.np_truth_frame(np.stack([pcm[t * H:t * H + W]] * 2, 1), W) for t in pick])
    assert mags_err(got[pick, 0], truth) <= 1.0, (seed, frames)
    first = int(rng.integers(0, frames))
    count = int(rng.integers(1, frames - first + 1))
    assert np.array_equal(eng.stft_batch(dev, first_frame=first, max_frames=count).cpu().numpy(), got[first:first + count])
    assert torch.equal(eng.stft_batch_f16(dev, first_frame=first, max_frames=count), torch.from_numpy(got[first:first + count]).cuda().to(torch.float16))
    px = eng.render_batch(dev, first_frame=first, max_frames=count)
    assert torch.equal(px[:, 0], eng.render_mags(torch.from_numpy(got[first:first + count, 0]).cuda().contiguous()))
    # the same samples from a 4-byte aligned address (the 8-byte sample pairs are then dword-aligned loads): the same kernel, the same bytes
    shifted = torch.zeros(n + 1, dtype=torch.float32, device="cuda")
    shifted[1:] = dev
    assert shifted[1:].data_ptr() % 8 == 4
    assert np.array_equal(eng.stft_batch(shifted[1:]).cpu().numpy(), got), (seed, frames)
