"""The CPU pixel path's image ring behind the C ABI: SimpleSpectrogram's row-major RGBA Pixbuf
(src/widgets/simple_spectrogram.rs:89-94), one pixel column per frame at `offset` (:140-164), the scrolling picture of two
sub-images (:181-209) -- as sgx_image_*.  Integer work: every comparison is bit for bit, against a plain torch scatter of the
same columns in the reference's order (column by column, later columns over earlier ones)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    return torch


def engine(**kw):
    from spectrogram_rs_amd import SpectrogramEngine
    return SpectrogramEngine(48000.0, **kw)


def reference_scatter(torch, buffer, offset, cols):
    """the pixel loop as written: for every column, put_pixel(px, row, ...) for all rows, then offset = (px + 1) % width"""
    width = buffer.shape[1]
    for col in cols:
        buffer[:, offset, :] = col
        offset = (offset + 1) % width
    return offset


@pytest.mark.parametrize("width,rows", [(40, 64), (33, 100), (1024, 1024)])
def test_columns_land_like_put_pixel_one_by_one(torch_cuda, width, rows):
    torch = torch_cuda
    eng = engine(window_samples=256, hop_samples=64, channels=2, rows=rows, gradient="magma")
    img = eng.image(width)
    assert (img.width, img.height, img.offset) == (width, rows, 0)
    want = torch.zeros((rows, width, 4), dtype=torch.uint8, device="cuda")      # a fresh image: zeros
    assert torch.equal(img.read(), want)
    gen = torch.Generator(device="cuda").manual_seed(3)
    off = 0
    # 0 columns, a few, exactly up to the edge, across the wrap, exactly the width, more than the width (the ring laps itself
    # and the later columns win), far more than the width
    for n in (0, 3, width - 3, 5, width, width + 7, 3 * width + 1, 1):
        cols = torch.randint(0, 256, (n, rows, 4), dtype=torch.uint8, device="cuda", generator=gen)
        new_off = img.write_columns(cols)
        off = reference_scatter(torch, want, off, cols)
        assert new_off == off == img.offset
        assert torch.equal(img.read(), want), n
        assert torch.equal(img.read(scrolled=True), torch.cat([want[:, off:], want[:, :off]], dim=1)), n
    img.close()
    eng.close()


def test_live_tick_into_the_image_equals_the_host_round_trip(torch_cuda):
    # SimpleSpectrogram's tick (:136-165) device to device (sgx_live_tick_image) against the same tick through a host array and
    # sgx_image_write_columns: same columns, same offset, same picture -- and more columns than the image is wide in one tick
    torch = torch_cuda
    pics = []
    for direct in (False, True):
        eng = engine(period=0.05, hop_samples=93, channels=2, gradient="viridis")
        live, img = eng.live(65536, reference_skip=True), eng.image(48)
        rng = np.random.default_rng(5)
        total = 0
        for n in (2400, 700, 93, 9000, 1):
            live.push(rng.uniform(-0.5, 0.5, (n, 2)).astype(np.float32), 2)
            if direct:
                total += live.tick_image(img)
            else:
                cols = live.tick("rgba")
                total += cols.shape[0]
                if cols.shape[0]:
                    img.write_columns(torch.from_numpy(np.ascontiguousarray(cols)).cuda())
        pics.append((total, img.offset, img.read().cpu().numpy(), img.read(scrolled=True).cpu().numpy()))
        img.close(); live.close(); eng.close()
    assert pics[0][0] == pics[1][0] > 48 and pics[0][1] == pics[1][1]
    assert np.array_equal(pics[0][2], pics[1][2]) and np.array_equal(pics[0][3], pics[1][3])
    assert pics[0][2].any()


def test_an_image_of_another_or_a_destroyed_context_is_refused(torch_cuda):
    import ctypes as C

    from spectrogram_rs_amd import _lib
    from spectrogram_rs_amd.engine import SgxError
    torch = torch_cuda
    a = engine(window_samples=256, hop_samples=64, channels=2)
    b = engine(window_samples=256, hop_samples=64, channels=2, rows=512)
    live, own, foreign = a.live(8192), a.image(16), b.image(16)
    live.push(np.random.default_rng(1).uniform(-0.5, 0.5, (1000, 2)).astype(np.float32), 2)
    before = len(live)
    with pytest.raises(SgxError) as err:
        live.tick_image(foreign)
    assert err.value.code == _lib.SGX_ERR_INVALID_ARG and "another" in str(err.value)
    assert len(live) == before and foreign.offset == 0 and own.offset == 0
    assert live.tick_image(own) == a.num_frames(1000) and own.offset == a.num_frames(1000) % 16
    # C callers: an image that outlives its context answers SGX_ERR_INVALID_ARG instead of touching freed memory
    lib = _lib.load()
    cols = torch.zeros((2, 512, 4), dtype=torch.uint8, device="cuda")
    h = foreign._h
    b.close()
    off = C.c_uint32(0)
    assert lib.sgx_image_write_columns(h, C.c_void_p(cols.data_ptr()), 2, C.byref(off)) == _lib.SGX_ERR_INVALID_ARG
    assert lib.sgx_image_read(h, 0, C.c_void_p(cols.data_ptr())) == _lib.SGX_ERR_INVALID_ARG
    foreign.close(); own.close(); live.close(); a.close()
