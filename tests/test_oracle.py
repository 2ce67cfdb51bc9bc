"""CPU tests of the oracle (oracle/spectro_oracle.c) against analytic known answers, an
independent float64 numpy FFT, and the committed golden fixtures.  No GPU."""
import math

import numpy as np
import pytest

import oracle

W, H, SR, R = 2048, 256, 48000, 1024
M = W - 1


def test_sizes_follow_f32_truncation():
    # fft.rs:19: (period * sample_rate) as usize in f32: 0.05 * 48000 -> 2400, 0.05 * 44100 -> 2205
    assert oracle.window_samples(48000.0, 0.05) == 2400
    assert oracle.window_samples(44100.0, 0.05) == 2205
    # gpu_spectrogram.rs:21-23 hop: (1/828?) -- the documented hops: 58 @ FRAMES_PER_SECOND, 93 for 2/1024
    assert oracle.hop_samples(48000.0, 2.0 / 1024.0) == 93
    assert oracle.num_frames(2047, W, H) == 0
    assert oracle.num_frames(2048, W, H) == 1
    assert oracle.num_frames(2048 + 255, W, H) == 1
    assert oracle.num_frames(2048 + 256, W, H) == 2
    assert oracle.num_frames(256_001_792, W, H) == 1_000_000


def test_hann_is_periodic_f32():
    w = oracle.hann_window(W)
    assert w.dtype == np.float32 and w[0] == 0.0
    ref = 0.5 * (1 - np.cos(2 * np.pi * np.arange(W) / W))  # periodic: denominator W (fft.rs:61)
    assert np.abs(w - ref).max() < 2e-7
    assert abs(w[W // 2] - 1.0) < 1e-7


def test_short_input_is_none():
    x = np.zeros((W - 1, 2), np.float32)
    assert oracle.fft_process(x, W) is None
    assert oracle.fft_process(np.zeros((W, 2), np.float32), W) is not None


def test_bin_centred_sine_known_answer():
    # periodic Hann + 2x padding: amplitude A at even P-bin k0 -> A/2 at k0, A/4 at k0+-2, 0 at even offsets >= 4
    A, k0 = 0.8, 200
    n = np.arange(W)
    s = (A * np.sin(2 * np.pi * k0 * n / (2 * W) + 0.3)).astype(np.float32)
    m = oracle.fft_process(np.stack([s, np.zeros_like(s)], 1), W, oracle.F64)
    j = k0 - 1  # bin k is stored at index k-1 (DC dropped, fft.rs:81)
    assert abs(m[j, 0] - A / 2) < 1e-6
    assert abs(m[j - 2, 0] - A / 4) < 1e-6 and abs(m[j + 2, 0] - A / 4) < 1e-6
    assert m[j + 4, 0] < 1e-6 and m[j + 6, 0] < 1e-6 and m[j - 4, 0] < 1e-6
    assert m[:, 1].max() < 1e-12  # right channel silent
    m32 = oracle.fft_process(np.stack([s, np.zeros_like(s)], 1), W, oracle.F32)
    assert abs(m32[j, 0] - A / 2) < 1e-6


def test_left_right_separation_and_linearity():
    rng = np.random.default_rng(1)
    l = rng.uniform(-1, 1, W).astype(np.float32)
    r = rng.uniform(-1, 1, W).astype(np.float32)
    z = np.zeros(W, np.float32)
    both = oracle.fft_process(np.stack([l, r], 1), W, oracle.F64)
    only_l = oracle.fft_process(np.stack([l, z], 1), W, oracle.F64)
    only_r = oracle.fft_process(np.stack([z, r], 1), W, oracle.F64)
    assert np.abs(both[:, 0] - only_l[:, 0]).max() < 1e-12
    assert np.abs(both[:, 1] - only_r[:, 1]).max() < 1e-12
    assert only_l[:, 1].max() < 1e-12 and only_r[:, 0].max() < 1e-12
    # magnitude is homogeneous: scaling the input by 0.5 (exact in f32) halves every bin
    half = oracle.fft_process(np.stack([l * 0.5, r * 0.5], 1), W, oracle.F32)
    full = oracle.fft_process(np.stack([l, r], 1), W, oracle.F32)
    assert np.array_equal(half * 2, full)


def test_impulse_gives_window_sample():
    # delta at n0 -> every bin magnitude = hann[n0] * 2/W
    n0 = 700
    x = np.zeros((W, 2), np.float32)
    x[n0, 0] = 1.0
    m = oracle.fft_process(x, W, oracle.F64)
    expect = float(oracle.hann_window(W)[n0]) * 2.0 / W
    assert np.abs(m[:, 0] - expect).max() < 1e-12


def test_parseval_with_hann_energy():
    x = oracle.white_noise(W, 12345)
    m = oracle.fft_process(np.stack([x, np.zeros_like(x)], 1), W, oracle.F64)
    w = oracle.hann_window(W)
    # the reference multiplies in f32 (fft.rs:62); bins 1..W-1 hold half of the energy minus DC and Nyquist
    z = (x * w).astype(np.float64)
    F = np.fft.fft(np.concatenate([z, np.zeros(W)]))
    lhs = (m[:, 0] * W / 2.0) ** 2
    assert abs(lhs.sum() - (np.abs(F[1:W]) ** 2).sum()) / lhs.sum() < 1e-12


@pytest.mark.parametrize("Wt", [8, 64, 1024, 2048, 8192, 2400, 2205])
def test_f64_oracle_matches_numpy(Wt):
    x = oracle.white_noise(2 * Wt, 99).reshape(-1, 2)
    a = oracle.fft_process(x, Wt, oracle.F64)
    b = oracle.np_truth_frame(x, Wt)
    assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max()


@pytest.mark.parametrize("Wt", [2048, 8192, 2400])
def test_f32_oracle_within_tolerance_of_truth(Wt, mags_err):
    x = oracle.white_noise(2 * Wt, 7).reshape(-1, 2)
    a = oracle.fft_process(x, Wt, oracle.F32)
    b = oracle.np_truth_frame(x, Wt)
    assert mags_err(a, b) <= 1.0


def test_golden_sweep_and_noise(gold, mags_err):
    g = gold("config1_sweep.npz")
    for s, exp in zip(g["input"], g["expected_f64"]):
        lr = np.stack([s, s], 1)
        assert np.abs(oracle.fft_process(lr, W, oracle.F64) - exp).max() <= 1e-12 * exp.max()
        assert mags_err(oracle.fft_process(lr, W, oracle.F32), exp) <= 1.0
    assert np.array_equal(g["input"][0], oracle.sine_sweep(W, 0))
    assert np.array_equal(g["window"], oracle.hann_window(W))
    g = gold("noise_frames.npz")
    for t, s, exp in zip(g["frame_index"], g["input"], g["expected_f64"]):
        assert np.array_equal(s, oracle.white_noise(W, int(t) * H))
        assert mags_err(oracle.fft_process(np.stack([s, s], 1), W, oracle.F32), exp) <= 1.0


def test_stream_matches_per_frame_and_channel_layouts():
    n = W + H * 9 + 17
    mono = oracle.white_noise(n)
    out = oracle.stream_process(mono, 1, W, H)
    assert out.shape == (10, 1, M, 2)
    for t in (0, 3, 9):
        s = mono[t * H:t * H + W]
        assert np.array_equal(out[t, 0], oracle.fft_process(np.stack([s, s], 1), W))
    assert np.array_equal(out[..., 0], out[..., 1])  # mono -> (s, s): L == R
    # stereo and 4-channel (two pairs) interleaved
    st = oracle.white_noise(2 * n, seed=3).reshape(n, 2)
    out2 = oracle.stream_process(st, 2, W, H, threads=3)
    assert np.array_equal(out2[4, 0], oracle.fft_process(st[4 * H:4 * H + W], W))
    q = np.concatenate([st, st[:, ::-1]], 1)  # pairs (l, r) and (r, l)
    out4 = oracle.stream_process(q, 4, W, H)
    assert np.array_equal(out4[:, 0], out2[:, 0])
    assert np.array_equal(out4[:, 1, :, 0], out2[:, 0, :, 1])
    # sub-ranges
    part = oracle.stream_process(mono, 1, W, H, first=4, count=3)
    assert np.array_equal(part, out[4:7])
    assert oracle.stream_process(mono[:W - 1], 1, W, H).shape[0] == 0


def test_white_noise_definition():
    def lowbias32(x):
        x &= 0xFFFFFFFF
        x ^= x >> 16; x = (x * 0x7feb352d) & 0xFFFFFFFF
        x ^= x >> 15; x = (x * 0x846ca68b) & 0xFFFFFFFF
        x ^= x >> 16
        return x
    got = oracle.white_noise(5, first=1000)
    for i in range(5):
        h = lowbias32(0x5EED0001 ^ (1000 + i))
        assert got[i] == np.float32((h >> 8) * 2.0**-23 - 1.0)
    big = oracle.white_noise(1 << 16)
    assert -1.0 <= big.min() and big.max() < 1.0 and abs(big.mean()) < 0.01 and abs(big.var() - 1 / 3) < 0.01
    # past 2^32 samples the high word is mixed in (config 5): not a repeat of the first 2^32
    assert not np.array_equal(oracle.white_noise(8, first=1 << 32), oracle.white_noise(8, first=0))


# ---- resampling ------------------------------------------------------------------------------

def test_bin_edges_fixture_and_counts(gold):
    e = oracle.bin_edges(R)
    assert np.array_equal(e, gold("bin_edges_1024.npy"))
    assert e[0] == np.float32(32.0) and abs(float(e[-1]) - 22030.0) < 2e-3
    counts = np.array([oracle.num_samples_in(M, SR, float(e[i]), float(e[i + 1])) for i in range(R)])
    assert np.array_equal(counts, gold("row_counts_1024.npy"))
    assert counts.max() == 11 and counts.sum() == 2173 and (counts > 1).sum() == 281


def test_index_of_and_period_quirk_q2():
    assert oracle.period(M, SR) == np.float32(np.float32(2.0 * M) / np.float32(SR))
    assert oracle.index_of(-5.0, M, SR) == 0.0
    assert oracle.index_of(1e9, M, SR) == float(M - 1)
    # magnitudes[0] is treated as 0 Hz (one-bin downward shift, quirk Q2)
    assert abs(oracle.index_of(1000.0, M, SR) - 1000.0 * 2 * M / SR) < 1e-3


def test_interpolators_on_hand_computed_values():
    data = np.zeros((8, 2), np.float32)
    data[:, 0] = [0, 1, 4, 9, 16, 25, 36, 49]
    data[:, 1] = 1.0
    # integer index: both return the sample itself
    assert oracle.cubic_interpolate(data, 3.0)[0] == 9.0 and oracle.cosine_interpolate(data, 3.0)[0] == 9.0
    # Paul Bourke cubic at x=2.5 with y = 1, 4, 9, 16
    y0, y1, y2, y3, mu = 1.0, 4.0, 9.0, 16.0, 0.5
    a0 = y3 - y2 - y0 + y1; a1 = y0 - y1 - a0; a2 = y2 - y0
    assert oracle.cubic_interpolate(data, 2.5)[0] == np.float32(a0 * mu**3 + a1 * mu**2 + a2 * mu + y1)
    assert oracle.cubic_interpolate(data, 2.5)[1] == 1.0
    # cosine midpoint is the plain average
    assert abs(oracle.cosine_interpolate(data, 2.5)[0] - 6.5) < 1e-6
    # saturation at both ends (quirk Q4 and the clamp at len-1)
    assert oracle.cubic_interpolate(data, 0.25)[0] == oracle.cubic_interpolate(data, 0.25)[0]
    assert oracle.cubic_interpolate(data, 7.0)[0] == 49.0
    assert oracle.cosine_interpolate(data, 7.0)[0] == 49.0


def test_magnitude_in_is_mean_over_half_open_linspace():
    rng = np.random.default_rng(5)
    data = rng.uniform(0, 1, (M, 2)).astype(np.float32)
    f0, f1 = 15000.0, 15130.0
    n = oracle.num_samples_in(M, SR, f0, f1)
    assert n == int(math.floor(oracle.index_of(f1, M, SR) - oracle.index_of(f0, M, SR)))
    step = np.float32(np.float32(f1 - f0) / np.float32(n))
    acc = np.zeros(2, np.float32)
    for i in range(n):
        f = np.float32(np.float32(f0) + np.float32(np.float32(i) * step))
        acc = (acc + oracle.cubic_interpolate(data, oracle.index_of(float(f), M, SR))).astype(np.float32)
    assert np.array_equal(oracle.magnitude_in(data, SR, f0, f1), (acc / np.float32(n)).astype(np.float32))
    # a range narrower than one bin still takes one sample, at its start
    one = oracle.magnitude_in(data, SR, 100.0, 100.5)
    assert np.array_equal(one, oracle.cubic_interpolate(data, oracle.index_of(100.0, M, SR)))


# ---- colour ------------------------------------------------------------------------------------

def test_lut_index_rules_and_saturation():
    assert oracle.lut_index(-0.1) == 0 and oracle.lut_index(float("nan")) == 0
    assert oracle.lut_index(0.0) == 0 and oracle.lut_index(0.999) == 255 and oracle.lut_index(1.0) == 255
    assert oracle.lut_index(7.0) == 255 and oracle.lut_index(0.5) == 128
    assert oracle.lut_index(0.5, 256, oracle.LUT_ROUND_NM1) == 128 and oracle.lut_index(1.0, 256, oracle.LUT_ROUND_NM1) == 255
    assert oracle.alpha_u8(-3.0) == 0 and oracle.alpha_u8(float("nan")) == 0 and oracle.alpha_u8(2.0) == 255
    assert oracle.alpha_u8(0.5) == 127


def test_color_for_mono_and_stereo(gradients):
    v = gradients["viridis"]
    rgb, a = oracle.color_for(v, 0.0, 0.0)  # silence: 10 log10(1e-7) = -70.00001 -> t slightly < 0 (Q11)
    assert tuple(rgb) == tuple(v[0]) and a == 1.0
    rgb, a = oracle.color_for(v, 1.0, 1.0)  # far above -10 dB
    assert tuple(rgb) == tuple(v[255])
    # -40 dB power -> t = 0.5 -> index 128 (mono input: L = R, power = 2 m^2, quirk Q5)
    m = math.sqrt(10 ** (-40 / 10) / 2)
    rgb, _ = oracle.color_for(v, m, m)
    assert tuple(rgb) in (tuple(v[127]), tuple(v[128]))
    # stereo: colour from l / (|l| + |r|), alpha = bounded dB
    rgb, a = oracle.color_for(v, 0.01, 0.0, stereo=True)
    assert tuple(rgb) == tuple(v[255]) and abs(a - (10 * math.log10(1e-4 + 1e-7) + 70) / 60) < 1e-5
    rgb, a = oracle.color_for(v, 0.0, 0.0, stereo=True)  # 0/0 = NaN -> index 0
    assert tuple(rgb) == tuple(v[0])


def test_lookup_table_normalises_by_256(gradients):
    t = oracle.lookup_table(gradients["magma"], 32)
    assert t.shape == (32, 32, 4) and t[..., 3].min() == 1.0
    assert t[31, 0, 0] == np.float32(gradients["magma"][255][0] / 256.0)  # quirk Q10
    ts = oracle.lookup_table(gradients["magma"], 32, stereo=True)
    assert ts[5, 0, 3] == np.float32(5 / 31) and np.array_equal(ts[0, :, :3], ts[31, :, :3])


def test_render_column_layout_and_golden(gold, gradients):
    g = gold("rgba_columns.npz")
    v = gradients["viridis"]
    for i, mags in enumerate(g["mags"]):
        for name, interp in (("cubic", oracle.INTERP_CUBIC), ("cosine", oracle.INTERP_COSINE)):
            col = oracle.render_column(mags, SR, v, interp=interp)
            assert col.shape == (R, 4) and np.array_equal(col, g["viridis_" + name][i])
            assert (col[:, 3] == 255).all()
    # low frequencies at the bottom (simple_spectrogram.rs:150): a 100 Hz tone lights the lower rows
    mags = np.zeros((M, 2), np.float32)
    k = int(round(100.0 * 2 * M / SR))
    mags[k] = 0.3
    col = oracle.render_column(mags, SR, v)
    lit = np.nonzero((col[:, :3] != v[0]).any(axis=1))[0]
    assert len(lit) and lit.min() > R // 2


def test_spectrum_analyzer_bands_and_levels():
    # log_space(32, end, 129, 10) (spectrum_analyzer.rs:20-36): geometric, first edge 32 Hz, edge 129 = end
    edges = np.array([oracle.log_space(32.0, 24000.0, 129, 10.0, i) for i in range(130)])
    assert edges[0] == pytest.approx(32.0, rel=1e-6) and edges[129] == pytest.approx(24000.0, rel=1e-5)
    ratio = edges[1:] / edges[:-1]
    assert np.allclose(ratio, (24000.0 / 32.0) ** (1 / 129), rtol=1e-5)
    # a flat spectrum (l = r = c): magnitude_in of every band is c, so every bar shows
    # (10 log10(sqrt(2) c + 1e-7) + 70) / 60 (:60-62); bars start at 0.3 (:92)
    c = 0.01
    mags = np.full((M, 2), c, np.float32)
    levels = np.full(128, 0.3)
    oracle.spectrum_levels(mags, SR, levels)
    want = (10 * math.log10(math.sqrt(2) * c + 1e-7) + 70) / 60
    assert np.allclose(levels, want, atol=2e-6) and want > 0.3
    # a quieter frame: the bars decay by 1 % per push instead of dropping (:64)
    oracle.spectrum_levels(mags * np.float32(1e-3), SR, levels)
    assert np.allclose(levels, want * 0.99, atol=2e-6)
    # below 44.1 kHz the last edge stays at 22050 Hz (.max(22050.0), :54)
    assert oracle.log_space(32.0, max(8000 / 2, 22050.0), 129, 10.0, 129) == pytest.approx(22050.0, rel=1e-5)


@pytest.mark.parametrize("seed", range(24))
def test_oracle_fft_on_random_lengths_against_numpy(seed, mags_err):
    # the oracle's own mixed-radix FFT (it has to serve the application's 4800-point transform as well as the
    # power-of-two configs) against numpy's float64 FFT of the same windowed, padded frame, for lengths with small and
    # large prime factors alike
    rng = np.random.default_rng(seed)
    Wt = int(rng.choice([int(rng.integers(2, 300)), int(rng.integers(300, 6000)), int(2 ** rng.integers(1, 13)),
                         int(2 ** rng.integers(0, 6) * 3 ** rng.integers(0, 4) * 5 ** rng.integers(0, 3))]))
    Wt = max(Wt, 2)
    x = (oracle.white_noise(2 * Wt, seed) * np.float32(10.0 ** rng.uniform(-3, 0))).reshape(-1, 2)
    truth = oracle.np_truth_frame(x, Wt)
    assert truth.shape == (Wt - 1, 2)
    if Wt > 2:
        assert np.abs(oracle.fft_process(x, Wt, oracle.F64) - truth).max() <= 1e-11 * np.abs(truth).max()
        # the float32 mode evaluates a prime factor p of 2W as a plain p-term float32 sum (FFTW would use Rader /
        # Bluestein there): its error grows with p, and for the large primes it is the float64 mode that is the
        # reference (the GPU's chirp-z kernel is held to 1x against it: tests/test_gpu_fuzz.py)
        p, big, m = 2, 1, 2 * Wt
        while p * p <= m:
            while m % p == 0:
                big, m = p, m // p
            p += 1
        big = max(big, m) if m > 1 else big
        assert mags_err(oracle.fft_process(x, Wt, oracle.F32), truth) <= (1.0 if big <= 64 else 4.0)
    assert oracle.fft_process(x[:Wt - 1], Wt) is None                     # fft.rs:72
