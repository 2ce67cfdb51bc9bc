"""Live capture: the ring between the audio callback and the GUI tick, with its consumed side
resident on the device, and the SpectrumAnalyzer consumer -- against a pure-Python model of the
reference's HeapRb + hop loop and the CPU oracle.  Run with -m gpu on an MI355X."""
import threading

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

SR = 48000


class RingModel:
    """ringbuf::HeapRb<(f32, f32)> as the reference uses it (audio_input_list_model.rs:30,63-72) and the hop
    loop of AudioStreamTransform::process over it (audio_transform.rs:34-42), in plain numpy."""

    def __init__(self, capacity, W, H, reference_skip):
        self.capacity, self.W, self.H, self.reference_skip = capacity, W, H, reference_skip
        self.data = np.zeros((0, 2), np.float32)

    def push(self, values, channels):
        v = np.asarray(values, np.float32).reshape(-1)
        pairs = np.stack([v, v], 1) if channels == 1 else v[:len(v) // 2 * 2].reshape(-1, 2)
        take = min(len(pairs), self.capacity - len(self.data))   # push_iter drops what does not fit
        self.data = np.concatenate([self.data, pairs[:take]])
        return take

    def tick(self, max_frames=None):
        n = len(self.data)
        frames = oracle.num_frames(n, self.W, self.H)
        truncated = max_frames is not None and frames > max_frames
        if truncated:
            frames = max_frames
        lr = self.data[:(frames - 1) * self.H + self.W].copy() if frames else None
        skip = frames * self.H + (self.H if self.reference_skip and not truncated else 0)
        self.data = self.data[min(skip, n):]
        return frames, lr


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    return torch


def engine(**kw):
    from spectrogram_rs_amd import SpectrogramEngine
    return SpectrogramEngine(float(SR), channels=2, **kw)


@pytest.mark.parametrize("reference_skip", [False, True])
@pytest.mark.parametrize("cfg", [dict(period=0.05, hop_samples=93, capacity=4096),      # the application's own sizes
                                 dict(window_samples=2048, hop_samples=256, capacity=5000),
                                 dict(window_samples=64, hop_samples=16, capacity=200)])
def test_ticks_reproduce_the_hop_loop_over_the_ring(torch_cuda, mags_err, cfg, reference_skip):
    cfg = dict(cfg)
    capacity = cfg.pop("capacity")
    eng = engine(**cfg)
    ring = eng.live(capacity, reference_skip=reference_skip)
    model = RingModel(capacity, eng.W, eng.H, reference_skip)
    rng = np.random.default_rng(5)
    total_frames = 0
    for step in range(60):
        # capture callbacks of uneven size between two GUI ticks; every so often a burst that overflows the ring
        for _ in range(int(rng.integers(0, 4))):
            channels = int(rng.integers(1, 3))
            n = int(rng.integers(0, capacity // 3)) if step % 11 else capacity
            values = rng.uniform(-1, 1, n * channels + (channels == 2 and step % 3 == 0)).astype(np.float32)
            assert ring.push(values, channels) == model.push(values, channels)
            assert ring.occupied_len() == len(model.data)
        max_frames = 2 if step % 7 == 3 else None
        got = ring.tick("mags", max_frames)
        frames, lr = model.tick(max_frames)
        assert got.shape == (frames, eng.M, 2)
        assert ring.occupied_len() == len(model.data)
        if frames:
            ref = oracle.stream_process(lr, 2, eng.W, eng.H)[:, 0]
            assert mags_err(got, ref) <= (3.0 if eng.info.stft_kernel == 4 else 2.0)
        total_frames += frames
    assert total_frames > 20


def test_three_channel_input_is_refused_like_the_reference(torch_cuda):
    from spectrogram_rs_amd import SgxError
    eng = engine(window_samples=64, hop_samples=16)
    ring = eng.live(256)
    with pytest.raises(SgxError, match="3-channel input not supported"):
        ring.push(np.zeros(30, np.float32), 3)
    assert ring.occupied_len() == 0 and ring.tick().shape == (0, 63, 2)
    # a ring that cannot hold one window, and a mono context, are configuration errors
    with pytest.raises(SgxError):
        eng.live(63)
    from spectrogram_rs_amd import SpectrogramEngine
    with pytest.raises(SgxError):
        SpectrogramEngine(float(SR), channels=1, window_samples=64, hop_samples=16).live(256)


def test_tick_formats_agree_with_the_batch_entry_points(torch_cuda, gradients):
    torch = torch_cuda
    eng = engine(window_samples=2048, hop_samples=256, gradient="viridis")
    lr = oracle.white_noise(2 * (2048 + 256 * 9 + 100), seed=77).reshape(-1, 2) * np.float32(0.05)
    pcm = torch.from_numpy(lr).cuda().reshape(-1)
    outs = {}
    for what in ("mags", "mags_f16", "rgba"):
        ring = eng.live(8192)
        assert ring.push(lr, 2) == len(lr)
        outs[what] = ring.tick(what)
        assert ring.occupied_len() == len(lr) - 10 * 256
    assert np.array_equal(outs["mags"], eng.stft_batch(pcm)[:, 0].cpu().numpy())
    assert np.array_equal(outs["mags_f16"], eng.stft_batch_f16(pcm)[:, 0].cpu().numpy())
    assert np.array_equal(outs["rgba"], eng.render_batch(pcm)[:, 0].cpu().numpy())
    # and the pixel columns against the oracle's own magnitudes: isolated one-step differences only
    ref = oracle.render_columns(oracle.stream_process(lr, 2, 2048, 256)[:, 0], SR, gradients["viridis"])
    assert (outs["rgba"] != ref).any(axis=2).mean() < 5e-3


def test_producer_and_consumer_threads(torch_cuda, mags_err):
    # one capture thread pushing 10 ms callbacks, the GUI thread ticking: with a ring that never fills, the
    # frames of all ticks together are exactly the frames of the whole stream (t * H framing, nothing dropped)
    eng = engine(period=0.05, hop_samples=93)
    ring = eng.live(1 << 16)
    n = 48000
    lr = oracle.white_noise(2 * n, seed=3).reshape(-1, 2)
    done = threading.Event()

    def capture():
        for i in range(0, n, 480):
            assert ring.push(lr[i:i + 480], 2) == len(lr[i:i + 480])
        done.set()

    th = threading.Thread(target=capture)
    th.start()
    parts = []
    while True:
        finished = done.is_set()
        parts.append(ring.tick("mags"))
        if finished:
            break
    th.join()
    got = np.concatenate(parts)
    ref = oracle.stream_process(lr, 2, eng.W, eng.H, threads=8)[:, 0]
    assert got.shape == ref.shape and len(got) == oracle.num_frames(n, eng.W, eng.H)
    assert mags_err(got, ref) <= 3.0
    assert ring.occupied_len() == n - len(got) * eng.H


def test_widget_in_live_mode(torch_cuda, gradients):
    from spectrogram_rs_amd import ColorScheme, RingBuffer, SimpleSpectrogram
    tone = (0.25 * np.sin(2 * np.pi * 440.0 * np.arange(6000) / SR)).astype(np.float32)
    live = SimpleSpectrogram(None, sample_rate=SR)
    host = SimpleSpectrogram(RingBuffer(1 << 16), sample_rate=SR)
    for w in (live, host):
        w.set_palette(ColorScheme.new_mono("viridis", "Viridis"))
    for i in range(0, len(tone), 1500):       # four callbacks, a tick after each
        assert live.push(tone[i:i + 1500], 1) == 1500
        host.input_stream.push_mono(tone[i:i + 1500])
        assert live.snapshot() == host.snapshot()
        assert live.offset == host.offset
    assert live.offset > 0 and torch_cuda.equal(live.buffer, host.buffer)


def test_widget_survives_input_device_changes(torch_cuda):
    # the reference calls set_sample_rate on every input-device change (simple_spectrogram.rs:214-219, main.rs:73): the
    # transform is replaced wholesale.  The device-resident ring points into the context, so it has to go first --
    # twice in a row used to free the context under a live ring (use after free in sgx_live_destroy).
    from spectrogram_rs_amd import SimpleSpectrogram
    w = SimpleSpectrogram(None, sample_rate=SR)
    tone = (0.25 * np.sin(2 * np.pi * 440.0 * np.arange(6000) / SR)).astype(np.float32)
    for sr in (44100, 48000, 96000, 48000):
        old_ring, old_engine = w.live, w.engine
        w.set_sample_rate(sr)
        assert not old_ring._h.value and not old_engine._ctx.value       # both destroyed, ring first
        assert w.engine.W == oracle.window_samples(float(sr), 0.05) and w.live is not old_ring
        burst = tone[:w.live.capacity - 100]                               # (a fuller ring drops the overflow, as HeapRb does)
        assert w.push(burst, 1) == len(burst)
        assert w.snapshot() == w.engine.num_frames(len(burst))
    # an engine closes the rings it handed out before its context
    eng = engine(window_samples=2048, hop_samples=256)
    ring = eng.live(4096)
    eng.close()
    assert not ring._h.value
    ring.close()   # idempotent


@pytest.mark.parametrize("sr,interp", [(48000, 0), (44100, 0), (96000, 1)])
def test_spectrum_analyzer_levels(torch_cuda, sr, interp):
    # SpectrumAnalyzer::push_frequencies (spectrum_analyzer.rs:46-68): identical bars, push after push
    from spectrogram_rs_amd import SpectrogramEngine, SpectrumAnalyzer
    torch = torch_cuda
    eng = SpectrogramEngine(float(sr), channels=2, window_samples=2048, hop_samples=256, interp=interp)
    mags = oracle.stream_process(oracle.white_noise(2 * (2048 + 7 * 256), seed=21) * np.float32(0.02), 2, 2048, 256)[:, 0]
    mags[3] *= np.float32(1e-3)      # a quiet frame: the bars decay by 1 %
    mags[5] = 0.0                    # silence: 10 log10(1e-7) = -70 -> level ~ 0, bars keep decaying
    sa = SpectrumAnalyzer(eng)
    ref = np.full(128, 0.3, np.float64)
    assert np.array_equal(sa.level_bars, ref)
    for m in mags:
        sa.push_frequencies(torch.from_numpy(m).cuda())
        oracle.spectrum_levels(m, sr, ref, interp)
        assert np.array_equal(sa.level_bars, ref)
    assert ref.min() > 0.0 and ref.max() < 1.5
    # the band edges: log_space(32, max(sr / 2, 22050), 129, 10)
    assert oracle.log_space(32.0, max(sr / 2, 22050.0), 129, 10.0, 0) == pytest.approx(32.0, rel=1e-6)
