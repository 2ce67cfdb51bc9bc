"""The default widget's variant of the pixel path: GPUSpectrogram's F16F16 ring texture and
fragment program (src/widgets/gpu_spectrogram.rs:150-186,218-226,255-275) as sgx_view.  Checked against the oracle's
restatement of the same program text and sampler state; float tolerance, not bits: OpenGL leaves the filtering
arithmetic to the implementation, so this variant is not a parity target."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

SR = 48000


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    return torch


def engine(**kw):
    from spectrogram_rs_amd import SpectrogramEngine
    return SpectrogramEngine(48000.0, **kw)


def test_ring_texture_upload_wraps_like_the_reference(torch_cuda):
    # the upload loop (:255-275): rows land at `offset`, a block that reaches the top of the texture is split,
    # offset = (offset + block) % height
    torch = torch_cuda
    eng = engine(window_samples=64, hop_samples=16, channels=2, gradient="magma")
    view = eng.view(8)
    ring = np.zeros((8, eng.M, 2), np.float16)
    rng = np.random.default_rng(1)
    off = 0
    for n in (3, 4, 6, 8, 1, 11):       # 11 > height: the texture is overwritten more than once in one call
        rows = rng.uniform(0.0, 0.3, (n, eng.M, 2)).astype(np.float16)
        new_off = view.write_rows(torch.from_numpy(rows).cuda())
        for r in rows:
            ring[off] = r
            off = (off + 1) % 8
        assert new_off == off == view.offset
        got = view.draw(5, 3).cpu().numpy()
        pal = eng.lookup_table(32)
        assert np.allclose(got, oracle.glsl_fragments(ring, off, pal, 5, 3), atol=2e-4)


@pytest.mark.parametrize("scheme,stereo", [("magma", False), ("red_yellow_blue", True)])
def test_fragment_program_matches_its_restatement(torch_cuda, scheme, stereo):
    # the application's own sizes: W 2400 (M 2399 texels wide), a 2048-row ring, real frames (half precision rows from
    # the transform itself), a viewport whose pixel grid does not line up with the texels; mono and diverging palettes
    torch = torch_cuda
    eng = engine(period=0.05, hop_samples=58, channels=2)
    eng.set_builtin_scheme(scheme, stereo=stereo)
    n = eng.W + 58 * 299
    t = np.arange(n) / SR
    lr = np.stack([0.3 * np.sin(2 * np.pi * 440.0 * t) + 0.01 * oracle.white_noise(n, seed=1),
                   0.1 * np.sin(2 * np.pi * 3000.0 * t) + 0.01 * oracle.white_noise(n, seed=2)], 1).astype(np.float32)
    rows = eng.stft_batch_f16(torch.from_numpy(lr).cuda().reshape(-1))[:, 0].contiguous()     # [300][M][2] half
    view = eng.view(2048)
    assert view.write_rows(rows) == 300
    ring = np.zeros((2048, eng.M, 2), np.float16)
    ring[:300] = rows.cpu().numpy()
    got = view.draw(317, 211).cpu().numpy()
    ref = oracle.glsl_fragments(ring, 300, eng.lookup_table(32), 317, 211)
    assert got.shape == ref.shape == (211, 317, 4)
    # device logf / expf and the host's differ in the last bit or two: a palette coordinate moves by ~1e-6, a colour
    # by ~1e-5; one texel boundary crossed by such a move shows up as a (rare) larger step along an edge
    err = np.abs(got - ref)
    assert np.quantile(err, 0.999) < 2e-4 and err.max() < 2e-2
    # the picture is not trivial: the 440 Hz line of the left channel is visible where the ring holds frames
    assert got[..., :3].std() > 0.01
    # a palette change rebuilds the palette texture (set_palette, :329-333)
    eng.set_builtin_scheme("viridis")
    again = view.draw(317, 211).cpu().numpy()
    assert not np.allclose(again, got, atol=1e-3)
    assert np.quantile(np.abs(again - oracle.glsl_fragments(ring, 300, eng.lookup_table(32), 317, 211)), 0.999) < 2e-4


def test_default_widget_mirror(torch_cuda):
    # GPUSpectrogram: 1 / FRAMES_PER_SECOND stride (58 samples at 48 kHz), frames uploaded as half rows every render,
    # transform + texture rebuilt on a device change (:320-327)
    from spectrogram_rs_amd import ColorScheme, GPUSpectrogram
    w = GPUSpectrogram(sample_rate=SR)
    assert w.engine.H == 58 and w.engine.W == 2400 and w.texture.rows == 2048
    tone = (0.25 * np.sin(2 * np.pi * 1000.0 * np.arange(4000) / SR)).astype(np.float32)
    assert w.push(tone, 1) == 4000
    frames = w.engine.num_frames(4000)
    img = w.render(64, 48)
    assert img.shape == (48, 64, 4) and w.texture.offset == frames and bool(torch_cuda.isfinite(img).all())
    w.set_palette(ColorScheme.new_stereo("spectral", (0, 0, 0), "Spectral (Stereo)"))
    img2 = w.render(64, 48)
    assert not torch_cuda.allclose(img, img2)
    w.set_sample_rate(44100)
    assert w.engine.W == 2205 and w.texture.offset == 0
    assert w.render(16, 16).shape == (16, 16, 4)


def test_a_view_that_outlives_its_context_fails_cleanly(torch_cuda):
    # C and Rust callers have no weakref list doing the ordering for them: sgx_destroy(ctx) detaches the views still alive,
    # whose calls then answer SGX_ERR_INVALID_ARG instead of reading freed memory; destroying them afterwards is fine
    import ctypes as C

    from spectrogram_rs_amd import _lib
    torch = torch_cuda
    lib = _lib.load()
    cfg = _lib.sgx_config()
    lib.sgx_config_init(C.byref(cfg))
    cfg.sample_rate, cfg.window_samples, cfg.hop_samples, cfg.channels = 48000.0, 64, 16, 2
    ctx, view = C.c_void_p(), C.c_void_p()
    assert lib.sgx_create(C.byref(cfg), C.byref(ctx)) == 0
    assert lib.sgx_view_create(ctx, 8, C.byref(view)) == 0
    rows = torch.zeros((2, 63, 2), dtype=torch.float16, device="cuda")
    off = C.c_uint32(0)
    assert lib.sgx_view_write_rows(view, C.c_void_p(rows.data_ptr()), 2, C.byref(off)) == 0 and off.value == 2
    lib.sgx_destroy(ctx)                 # the view is still alive
    out = torch.zeros((4, 4, 4), dtype=torch.float32, device="cuda")
    assert lib.sgx_view_write_rows(view, C.c_void_p(rows.data_ptr()), 2, C.byref(off)) == _lib.SGX_ERR_INVALID_ARG
    assert lib.sgx_view_draw(view, 4, 4, C.c_void_p(out.data_ptr())) == _lib.SGX_ERR_INVALID_ARG
    lib.sgx_view_destroy(view)


def test_tick_into_the_ring_texture_equals_the_host_round_trip(torch_cuda):
    # the widget's tick (gpu_spectrogram.rs:255-275) device to device (sgx_live_tick_view) against the same tick through a host
    # array and sgx_view_write_rows: same rows, same offset, same picture
    torch = torch_cuda
    rng = np.random.default_rng(5)
    pics = []
    for direct in (False, True):
        eng = engine(period=0.05, hop_samples=58, channels=2, gradient="magma")
        live, view = eng.live(8192, reference_skip=True), eng.view(64)
        rng = np.random.default_rng(5)
        total = 0
        for n in (2400, 700, 58, 4000, 1):
            live.push(rng.uniform(-0.5, 0.5, (n, 2)).astype(np.float32), 2)
            if direct:
                total += live.tick_into(view)
            else:
                rows = live.tick("mags_f16")
                total += rows.shape[0]
                if rows.shape[0]:
                    view.write_rows(torch.from_numpy(np.ascontiguousarray(rows)).cuda())
        pics.append((total, view.offset, view.draw(96, 40).cpu().numpy()))
        view.close(); live.close(); eng.close()
    assert pics[0][0] == pics[1][0] > 64 and pics[0][1] == pics[1][1]       # more rows than the texture holds: it wrapped
    assert np.array_equal(pics[0][2], pics[1][2])


def test_tick_into_a_view_of_another_context_is_refused(torch_cuda):
    # sgx_live_tick_view's precondition (include/sgx.h): the view belongs to the ring's context.  A view of a context with another
    # window has another row length, another stream and possibly another device; it is refused before anything is uploaded,
    # transformed or skipped -- the ring keeps its samples, and the same tick into the ring's own view then yields every frame.
    from spectrogram_rs_amd import _lib
    from spectrogram_rs_amd.engine import SgxError
    a = engine(window_samples=256, hop_samples=64, channels=2)
    b = engine(window_samples=1024, hop_samples=64, channels=2)      # M = 1023: four times a's row length
    live, own, foreign = a.live(8192), a.view(16), b.view(16)
    rng = np.random.default_rng(11)
    live.push(rng.uniform(-0.5, 0.5, (1000, 2)).astype(np.float32), 2)
    before = len(live)
    with pytest.raises(SgxError) as err:
        live.tick_into(foreign)
    assert err.value.code == _lib.SGX_ERR_INVALID_ARG and "another" in str(err.value)
    assert len(live) == before and foreign.offset == 0 and own.offset == 0
    assert live.tick_into(own) == a.num_frames(1000) == own.offset
    b.close()                                                        # a destroyed context: the same answer
    with pytest.raises(SgxError):
        live.tick_into(foreign)
    foreign.close(); own.close(); live.close(); a.close()
