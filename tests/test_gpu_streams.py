"""The stream contract of the C ABI (threading row; include/sgx.h:14-19): work is enqueued on the context's
hipStream_t (sgx_set_stream) and is asynchronous, sgx_sync waits for it, one context per stream.  The reference's counterpart is a
single-threaded RefCell (src/widgets/gpu_spectrogram.rs:58): nothing there ever overlaps, so the contract here is the library's own
and is tested as such -- two contexts on two NON-default streams, their launches interleaved from one host thread with no host
synchronisation in between, every stream held back by a long sleep kernel first: a kernel, copy or table upload that the library
put on the NULL stream (or on the other context's stream) instead of its own would run before its input exists and the bytes would
differ from the NULL-stream results.  Calls go straight through ctypes (no torch wrapper re-binding the stream per call)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H, R = 2048, 256, 1024
M = W - 1
SLEEP_CYCLES = 400_000_000          # ~0.2 s at 2 GHz: everything enqueued behind it is still pending when the host moves on


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    return torch


def engine(**kw):
    from spectrogram_rs_amd import SpectrogramEngine
    return SpectrogramEngine(48000.0, **kw)


def ptr(t):
    return C.c_void_p(t.data_ptr())


class Chain:
    """one context's work per round, through the C ABI only: noise -> float rows, noise -> pixel columns -> image ring"""

    def __init__(self, torch, eng, frames, rounds, width):
        self.torch, self.eng, self.lib, self.frames, self.rounds = torch, eng, eng._lib, frames, rounds
        self.n = (frames - 1) * H + W
        dev = eng.device
        # garbage everywhere: work that ran too early (before the noise existed) would transform THIS
        self.pcm = [torch.full((self.n * eng.channels,), float("nan"), dtype=torch.float32, device=dev) for _ in range(rounds)]
        self.mags = torch.full((rounds, frames, eng.pairs, M, 2), -1.0, dtype=torch.float32, device=dev)
        self.rgba = torch.full((rounds, frames, eng.pairs, R, 4), 7, dtype=torch.uint8, device=dev)
        self.picture = torch.zeros((R, width, 4), dtype=torch.uint8, device=dev)
        self.img = eng.image(width)
        self.acc = torch.zeros(1, dtype=torch.int64, device=dev)

    def enqueue(self, i):
        e, lib, got, off = self.eng, self.lib, C.c_size_t(0), C.c_uint32(0)
        ck = e._check
        ck(lib.sgx_synth_white_noise(e._ctx, ptr(self.pcm[i]), i * 1000, self.n, e.channels, 0x5EED0001 + i))
        ck(lib.sgx_stft_batch(e._ctx, ptr(self.pcm[i]), self.n, 0, self.frames, ptr(self.mags[i]), C.byref(got)))
        assert got.value == self.frames
        ck(lib.sgx_render_batch(e._ctx, ptr(self.pcm[i]), self.n, 0, self.frames, ptr(self.rgba[i]), C.byref(got)))
        ck(lib.sgx_image_write_columns(self.img._h, ptr(self.rgba[i]), self.frames * e.pairs, C.byref(off)))
        ck(lib.sgx_checksum_add(e._ctx, ptr(self.mags[i]), self.mags[i].numel() * 4, 0, ptr(self.acc)))

    def finish(self):
        self.eng._check(self.lib.sgx_image_read(self.img._h, 1, ptr(self.picture)))      # stream-ordered device-to-device copy

    def results(self):
        t = self.torch
        return (self.mags.clone(), self.rgba.clone(), self.picture.clone(), int(self.acc[0]), self.img.offset)


def test_two_contexts_on_two_streams_interleaved_without_host_syncs(torch_cuda):
    torch = torch_cuda
    frames, rounds = 3000, 4
    cfgs = [dict(window_samples=W, hop_samples=H, channels=1, interp=1, gradient="viridis"),       # K1R + fused mono pixels
            dict(window_samples=W, hop_samples=H, channels=2, interp=0, gradient="magma")]         # K1 (l, r) + fused pixels, cubic
    widths = [1000, 777]

    def run(streams):
        engs = [engine(**c) for c in cfgs]
        chains = [Chain(torch, e, frames, rounds, w) for e, w in zip(engs, widths)]
        torch.cuda.synchronize()                                  # allocations and fills are done before anything is enqueued
        for e, s in zip(engs, streams):
            e.set_stream(0 if s is None else s.cuda_stream)       # sgx_set_stream, once per context
            if s is not None:
                with torch.cuda.stream(s):
                    torch.cuda._sleep(SLEEP_CYCLES)               # the stream is busy: the library's work queues up behind it
        for i in range(rounds):                                   # A B A B ..., one host thread, no host sync anywhere
            for ch in chains:
                ch.enqueue(i)
        for ch in chains:
            ch.finish()
        pending = [s is not None and not s.query() for s in streams]
        for e in engs:
            e.sync()                                              # ONE sgx_sync per context
        if streams[0] is not None:
            assert all(s.query() for s in streams)                # sgx_sync really waited for the context's own stream
        out = [ch.results() for ch in chains]
        for ch in chains:
            ch.img.close()
        for e in engs:
            e.close()
        return out, pending

    want, _ = run([None, None])                                   # the NULL stream: the reference result
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    got, pending = run([s1, s2])
    assert all(pending), "the host must have run ahead of both streams (nothing in the enqueue path may synchronise)"
    for (gm, gp, gi, gc, go), (wm, wp, wi, wc, wo) in zip(got, want):
        assert go == wo and gc == wc
        assert torch.equal(gm, wm) and torch.equal(gp, wp) and torch.equal(gi, wi)
        assert bool(torch.isfinite(gm).all()) and float(gm.min()) >= 0.0      # no garbage was transformed
    # and the other order of streams (B's stream first): a dependence on creation / launch order would show here
    got2, _ = run([s2, s1])
    for g, w in zip(got2, want):
        assert g[3] == w[3] and torch.equal(g[1], w[1]) and torch.equal(g[2], w[2])


def test_a_live_tick_while_the_other_contexts_full_size_launch_is_in_flight(torch_cuda, mags_err):
    """context A: BASELINE config 2's own launch (1e6 frames, 16.4 GB of rows), several times over, on its stream; context B: a GUI
    tick of the live ring (host -> device copy, one launch, device -> host copy, host-synchronous ON ITS OWN STREAM ONLY) meanwhile.
    The tick returns while A is still running, its frames are the oracle's, and A's bytes are those of an undisturbed launch."""
    import oracle
    torch = torch_cuda
    F = 1_000_000
    a = engine(window_samples=W, hop_samples=H, channels=1)
    b = engine(window_samples=W, hop_samples=H, channels=2)
    pcm = a.white_noise((F - 1) * H + W)
    out = torch.empty((F, 1, M, 2), dtype=torch.float32, device=a.device)
    a.stft_batch(pcm, out=out)
    want = a.checksum(out)                                        # (synchronises: the undisturbed launch)
    out.zero_()
    ring = b.live(4096)
    lr = oracle.white_noise(2 * 6000, seed=5).reshape(-1, 2)
    assert ring.push(lr[:4000].reshape(-1), 2) == 4000
    warm = ring.tick("mags")                                      # a first tick on the default stream: the ring's lazy allocations
    assert warm.shape[0] == 8                                     # 8 frames consumed = 2048 pairs skipped, 1952 left
    assert ring.push(lr[4000:].reshape(-1), 2) == 2000
    held = lr[2048:]                                              # what the ring now holds: 3952 pairs
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    a.set_stream(sa.cuda_stream)
    b.set_stream(sb.cuda_stream)
    got = C.c_size_t(0)
    for _ in range(12):                                           # ~40 ms of launches on A's stream
        a._check(a._lib.sgx_stft_batch(a._ctx, ptr(pcm), pcm.numel(), 0, F, ptr(out), C.byref(got)))
    frames = ring.tick("mags")                                    # B's tick: returns host data, so it synchronises -- stream B only
    assert not sa.query(), "the tick must not have waited for the other context's stream"
    n = oracle.num_frames(len(held), W, H)
    assert n == 8 and frames.shape == (n, M, 2)
    ref = np.stack([oracle.np_truth_frame(held[t * H:t * H + W], W) for t in range(n)])
    assert mags_err(frames, ref) <= 1.0
    a.sync()
    assert sa.query()
    a.set_stream(0)
    assert a.checksum(out) == want
    ring.close()
    a.close()
    b.close()
