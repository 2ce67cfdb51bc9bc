"""SpectrogramEngine -- thin object wrapper over one sgx_ctx (include/sgx.h).

Device buffers are torch tensors (plumbing: allocation, streams); the arithmetic is entirely in
libsgx.so's HIP kernels.  All batch calls take and return tensors on the engine's GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib
from ._lib import SgxError, sgx_config, sgx_info


class SpectrogramEngine:
    def __init__(self, sample_rate: float = 48000.0, *, period: float = 0.0, stride: float = 0.0,
                 window_samples: int = 0, hop_samples: int = 0, channels: int = 1, rows: int = 1024,
                 f_min: float = 32.0, f_max: float = 22030.0, min_db: float = -70.0, max_db: float = -10.0,
                 interp: int = _lib.INTERP_CUBIC, lut_index_mode: int = _lib.LUT_FLOOR_N,
                 device: Optional[int] = None, force_generic: bool = False, gradient: Optional[str] = None,
                 fused_render: bool = True, lut_walk: bool = False,
                 mixed_generic: bool = False, complex_mono: bool = False, paired_frames: bool = False):
        import torch

        self._lib = _lib.load()
        self._ctx = C.c_void_p()
        cfg = sgx_config()
        self._lib.sgx_config_init(C.byref(cfg))
        cfg.sample_rate = sample_rate
        cfg.period, cfg.stride = period, stride
        if period == 0.0 and window_samples == 0:
            window_samples = 2048
        if stride == 0.0 and hop_samples == 0:
            hop_samples = 256
        cfg.window_samples, cfg.hop_samples = window_samples, hop_samples
        cfg.channels, cfg.rows = channels, rows
        cfg.f_min, cfg.f_max = f_min, f_max
        cfg.min_db, cfg.max_db = min_db, max_db
        cfg.interp, cfg.lut_index_mode = interp, lut_index_mode
        cfg.device = -1 if device is None else int(device)
        cfg.flags = (_lib.FLAG_FORCE_GENERIC if force_generic else 0) | (0 if fused_render else _lib.FLAG_NO_FUSED_RENDER) \
            | (_lib.FLAG_LUT_WALK if lut_walk else 0) | (_lib.FLAG_MIXED_GENERIC if mixed_generic else 0) \
            | (_lib.FLAG_COMPLEX_MONO if complex_mono else 0) | (_lib.FLAG_PAIRED_FRAMES if paired_frames else 0)
        if device is not None and torch.cuda.is_available():
            torch.cuda.set_device(int(device))
        rc = self._lib.sgx_create(C.byref(cfg), C.byref(self._ctx))
        if rc != 0:
            msg = self._lib.sgx_last_error(None).decode()
            self._ctx = C.c_void_p()
            raise SgxError(rc, msg)
        info = self._query()
        self.W, self.P, self.M, self.H = info.window_samples, info.fft_length, info.num_frequencies, info.hop_samples
        self.channels, self.pairs, self.R = info.channels, info.pairs, info.rows
        self.sample_rate = float(sample_rate)
        self.sample_rate_u32 = info.sample_rate_u32
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
        self.use_current_stream()
        if gradient is not None:
            self.set_builtin_gradient(gradient)

    # ---- plumbing --------------------------------------------------------------------------
    def _query(self):
        """sgx_query into self.info (again after a palette change: render_path depends on the colour scheme)"""
        info = sgx_info()
        self._check(self._lib.sgx_query(self._ctx, C.byref(info)))
        self.info = info
        return info

    def _check(self, rc: int) -> int:
        if rc < 0:
            raise SgxError(rc, self._lib.sgx_last_error(self._ctx).decode())
        return rc

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            # rings handed out by live() point into the context: they are destroyed first (include/sgx.h)
            for ref in list(getattr(self, "_rings", ())):
                ring = ref()
                if ring is not None:
                    ring.close()
            self._rings = []
            self._lib.sgx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def use_current_stream(self):
        """Enqueue on torch's current stream for this device."""
        import torch

        s = torch.cuda.current_stream(self.device)
        self._check(self._lib.sgx_set_stream(self._ctx, C.c_void_p(s.cuda_stream)))

    def set_stream(self, cuda_stream: int):
        self._check(self._lib.sgx_set_stream(self._ctx, C.c_void_p(cuda_stream)))

    def sync(self):
        self._check(self._lib.sgx_sync(self._ctx))

    def _dev_f32(self, t):
        import torch

        assert isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), \
            "expected a contiguous float32 CUDA tensor"
        assert t.device == self.device, f"tensor on {t.device}, engine on {self.device}"
        return C.c_void_p(t.data_ptr())

    def _out(self, out, shape, dtype):
        """The caller's output buffer, checked (the kernels write shape-many elements through a raw pointer), or a
        fresh one.  Also re-binds the context to torch's current stream: a batch call is ordered like a torch op."""
        import torch

        self.use_current_stream()
        if out is None:
            return torch.empty(shape, dtype=dtype, device=self.device)
        need = 1
        for d in shape:
            need *= int(d)
        assert isinstance(out, torch.Tensor) and out.is_cuda and out.device == self.device, \
            f"out must be a CUDA tensor on {self.device}"
        assert out.dtype == dtype, f"out must be {dtype}, got {out.dtype}"
        assert out.is_contiguous(), "out must be contiguous"
        assert out.numel() >= need, f"out holds {out.numel()} elements, the call writes {need}"
        return out

    # ---- sizes -----------------------------------------------------------------------------
    def num_frames(self, n_samples: int) -> int:
        return int(self._lib.sgx_num_frames(self._ctx, n_samples))

    # ---- transform -------------------------------------------------------------------------
    def stft_batch(self, pcm, first_frame: int = 0, max_frames: Optional[int] = None, out=None):
        """pcm: CUDA float32 tensor of n_samples*channels interleaved samples.
        Returns [frames][pairs][M][2] float32 (left, right) magnitudes."""
        import torch

        n_samples = pcm.numel() // self.channels
        total = self.num_frames(n_samples)
        n = max(total - first_frame, 0)
        if max_frames is not None:
            n = min(n, max_frames)
        out = self._out(out, (n, self.pairs, self.M, 2), torch.float32)
        got = C.c_size_t(0)
        if n:
            self._check(self._lib.sgx_stft_batch(self._ctx, self._dev_f32(pcm), n_samples, first_frame, n,
                                                 C.c_void_p(out.data_ptr()), C.byref(got)))
            assert got.value == n
        return out

    def stft_batch_f16(self, pcm, first_frame: int = 0, max_frames: Optional[int] = None, out=None):
        """As stft_batch, magnitudes as float16 (l, r) pairs: [frames][pairs][M][2] torch.float16
        (the texel format of the widget's F16F16 ring texture, gpu_spectrogram.rs:218-226)."""
        import torch

        n_samples = pcm.numel() // self.channels
        total = self.num_frames(n_samples)
        n = max(total - first_frame, 0)
        if max_frames is not None:
            n = min(n, max_frames)
        out = self._out(out, (n, self.pairs, self.M, 2), torch.float16)
        got = C.c_size_t(0)
        if n:
            self._check(self._lib.sgx_stft_batch_f16(self._ctx, self._dev_f32(pcm), n_samples, first_frame, n,
                                                     C.c_void_p(out.data_ptr()), C.byref(got)))
            assert got.value == n
        return out

    def process_one(self, lr: np.ndarray) -> Optional[np.ndarray]:
        """AudioTransform::process on host (l, r) pairs: [n][2] -> [M][2] or None."""
        lr = np.ascontiguousarray(lr, np.float32).reshape(-1, 2)
        out = np.empty((self.M, 2), np.float32)
        rc = self._check(self._lib.sgx_process_one(self._ctx, lr.ctypes.data_as(C.c_void_p), lr.shape[0],
                                                   out.ctypes.data_as(C.c_void_p)))
        return out if rc == 1 else None

    # ---- pixel path ------------------------------------------------------------------------
    def render_batch(self, pcm, first_frame: int = 0, max_frames: Optional[int] = None, out=None):
        """PCM -> [frames][pairs][R][4] uint8 RGBA columns (image-row order: row 0 = top)."""
        import torch

        n_samples = pcm.numel() // self.channels
        total = self.num_frames(n_samples)
        n = max(total - first_frame, 0)
        if max_frames is not None:
            n = min(n, max_frames)
        out = self._out(out, (n, self.pairs, self.R, 4), torch.uint8)
        got = C.c_size_t(0)
        if n:
            self._check(self._lib.sgx_render_batch(self._ctx, self._dev_f32(pcm), n_samples, first_frame, n,
                                                   C.c_void_p(out.data_ptr()), C.byref(got)))
            assert got.value == n
        return out

    def render_mags(self, mags, out=None):
        """[columns][M][2] float32 magnitudes -> [columns][R][4] uint8."""
        import torch

        n = mags.numel() // (self.M * 2)
        out = self._out(out, (n, self.R, 4), torch.uint8)
        if n:
            self._check(self._lib.sgx_render_mags(self._ctx, self._dev_f32(mags), n, C.c_void_p(out.data_ptr())))
        return out

    def magnitude_in(self, mags, ranges: np.ndarray, out=None):
        """FrequencySample::magnitude_in for every column and every (f0, f1) range:
        mags [columns][M][2] -> [columns][n_ranges][2] float32."""
        import torch

        ranges = np.ascontiguousarray(ranges, np.float32).reshape(-1, 2)
        n = mags.numel() // (self.M * 2)
        out = self._out(out, (n, ranges.shape[0], 2), torch.float32)
        if n and ranges.shape[0]:
            self._check(self._lib.sgx_magnitude_in(self._ctx, self._dev_f32(mags), n, ranges.ctypes.data_as(C.c_void_p),
                                                   ranges.shape[0], C.c_void_p(out.data_ptr())))
        return out

    def spectrum_levels(self, column, levels: np.ndarray) -> np.ndarray:
        """SpectrumAnalyzer::push_frequencies (spectrum_analyzer.rs:46-68) for one column [M][2] on the device;
        `levels` (float64, one per bar) is updated in place and returned."""
        assert levels.dtype == np.float64 and levels.flags.c_contiguous
        assert column.numel() == self.M * 2
        self._check(self._lib.sgx_spectrum_levels(self._ctx, self._dev_f32(column), levels.size,
                                                  levels.ctypes.data_as(C.c_void_p)))
        return levels

    # ---- live capture ----------------------------------------------------------------------
    def live(self, capacity: int = 4096, reference_skip: bool = False) -> "LiveRing":
        """The ring between the audio callback and the GUI tick, consumed side resident on the device."""
        import weakref

        ring = LiveRing(self, capacity, reference_skip)
        if not hasattr(self, "_rings"):
            self._rings = []
        self._rings = [r for r in self._rings if r() is not None] + [weakref.ref(ring)]
        return ring

    def image(self, width: int = 1024) -> "ImageRing":
        """the row-major width x rows RGBA image ring of SimpleSpectrogram (simple_spectrogram.rs:89-94) on the device"""
        return ImageRing(self, width)

    def view(self, viewport_frames: int = 2048) -> "ViewRing":
        """GPUSpectrogram's F16F16 ring texture + fragment program on the device (include/sgx.h: sgx_view)."""
        import weakref

        v = ViewRing(self, viewport_frames)
        if not hasattr(self, "_rings"):
            self._rings = []
        self._rings = [r for r in self._rings if r() is not None] + [weakref.ref(v)]
        return v

    # ---- colour scheme ---------------------------------------------------------------------
    def set_gradient(self, rgb: np.ndarray, stereo: bool = False):
        rgb = np.ascontiguousarray(rgb, np.uint8).reshape(-1, 3)
        self._check(self._lib.sgx_set_gradient(self._ctx, rgb.ctypes.data_as(C.c_void_p), rgb.shape[0], int(stereo)))
        self._query()

    def set_gradient_fn(self, fn, stereo: bool = False):
        """Continuous gradient: `fn(t) -> (r, g, b)` stands in for colorous' eval_continuous(t).  It is
        called on the host while the thresholds are built (and by lookup_table), never during a launch."""
        def thunk(t, out, _user):
            r, g, b = fn(t)
            out[0], out[1], out[2] = int(r), int(g), int(b)
        self._gradient_cb = _lib.GRADIENT_FN(thunk)  # keep alive: lookup_table calls it again
        self._check(self._lib.sgx_set_gradient_fn(self._ctx, C.cast(self._gradient_cb, C.c_void_p), None, int(stereo)))
        self._query()

    def set_builtin_scheme(self, name: str, stereo: bool = False):
        """ColorScheme::new_mono / new_stereo with a gradient the library evaluates itself (include/sgx.h)"""
        self._check(self._lib.sgx_set_builtin_scheme(self._ctx, name.encode(), int(stereo)))
        self._query()

    def set_builtin_gradient(self, name: str):
        self._check(self._lib.sgx_set_builtin_gradient(self._ctx, name.encode()))
        self._query()

    def lookup_table(self, resolution: int = 32) -> np.ndarray:
        out = np.empty((resolution, resolution, 4), np.float32)
        self._check(self._lib.sgx_lookup_table(self._ctx, resolution, out.ctypes.data_as(C.c_void_p)))
        return out

    # ---- introspection ---------------------------------------------------------------------
    def bin_edges(self) -> np.ndarray:
        out = np.empty(self.R + 1, np.float32)
        self._check(self._lib.sgx_bin_edges(self._ctx, out.ctypes.data_as(C.c_void_p)))
        return out

    def row_sample_counts(self) -> np.ndarray:
        out = np.empty(self.R, np.uint32)
        self._check(self._lib.sgx_row_sample_counts(self._ctx, out.ctypes.data_as(C.c_void_p)))
        return out

    def window(self) -> np.ndarray:
        out = np.empty(self.W, np.float32)
        self._check(self._lib.sgx_window(self._ctx, out.ctypes.data_as(C.c_void_p)))
        return out

    # ---- harness helpers -------------------------------------------------------------------
    def white_noise(self, n_samples: int, first: int = 0, seed: int = 0x5EED0001, channels: Optional[int] = None, out=None):
        import torch

        ch = self.channels if channels is None else channels
        out = self._out(out, (n_samples * ch,), torch.float32)
        self._check(self._lib.sgx_synth_white_noise(self._ctx, C.c_void_p(out.data_ptr()), first, n_samples, ch, seed))
        return out

    def checksum_add(self, t, acc, base_word: int = 0) -> None:
        """Add the checksum of `t` (word indices from base_word) into `acc`, a one-element int64 tensor on this
        device; asynchronous on the current stream (no host round trip)."""
        import torch

        assert acc.is_cuda and acc.device == self.device and acc.dtype == torch.int64 and acc.numel() == 1
        assert t.is_cuda and t.device == self.device and t.is_contiguous()
        self.use_current_stream()
        self._check(self._lib.sgx_checksum_add(self._ctx, C.c_void_p(t.data_ptr()), t.numel() * t.element_size(),
                                               base_word, C.c_void_p(acc.data_ptr())))

    def checksum(self, t, base_word: int = 0) -> int:
        nbytes = t.numel() * t.element_size()
        v = C.c_uint64(0)
        self._check(self._lib.sgx_checksum(self._ctx, C.c_void_p(t.data_ptr()), nbytes, base_word, C.byref(v)))
        return int(v.value)


class LiveRing:
    """sgx_live: ringbuf::HeapRb<(f32, f32)> (audio_input_list_model.rs:30) whose consumer is the hop loop of
    AudioStreamTransform::process (audio_transform.rs:34-42), run on the GPU once per tick."""

    _FORMATS = {"mags": (_lib.LIVE_MAGS, np.float32), "mags_f16": (_lib.LIVE_MAGS_F16, np.float16),
                "rgba": (_lib.LIVE_RGBA, np.uint8)}

    def __init__(self, engine: SpectrogramEngine, capacity: int = 4096, reference_skip: bool = False):
        self.engine = engine
        self.capacity = int(capacity)
        self._lib = engine._lib
        self._h = C.c_void_p()
        engine._check(self._lib.sgx_live_create(engine._ctx, self.capacity,
                                                _lib.LIVE_REFERENCE_SKIP if reference_skip else 0, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.sgx_live_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def push(self, data, channels: int) -> int:
        """The input callback (audio_input_list_model.rs:63-75): interleaved float32 `data`; returns the pairs
        accepted (overflow is dropped).  Raises SgxError(SGX_ERR_UNSUPPORTED) for more than two channels."""
        data = np.ascontiguousarray(data, np.float32).reshape(-1)
        n = int(self._lib.sgx_live_push(self._h, data.ctypes.data_as(C.c_void_p), data.size, channels))
        if n < 0:
            raise SgxError(n, f"{channels}-channel input not supported!" if n == _lib.SGX_ERR_UNSUPPORTED else "sgx_live_push")
        return n

    def occupied_len(self) -> int:
        return int(self._lib.sgx_live_occupied(self._h))

    def __len__(self) -> int:
        return self.occupied_len()

    def tick(self, what: str = "mags", max_frames: Optional[int] = None) -> np.ndarray:
        """One GUI tick: every complete frame of the ring as a host array
        ("mags": [frames][M][2] f32, "mags_f16": the same in half, "rgba": [frames][R][4] u8)."""
        code, dtype = self._FORMATS[what]
        e = self.engine
        if max_frames is None:
            max_frames = e.num_frames(self.capacity)
        shape = (max_frames, e.R, 4) if what == "rgba" else (max_frames, e.M, 2)
        out = np.empty(shape, dtype)
        got = C.c_size_t(0)
        e._check(self._lib.sgx_live_tick(self._h, code, out.ctypes.data_as(C.c_void_p), max_frames, C.byref(got)))
        return out[:got.value]

    def tick_image(self, image: "ImageRing", max_frames: Optional[int] = None) -> int:
        """one GUI tick of SimpleSpectrogram (simple_spectrogram.rs:136-165): every complete frame becomes a pixel column of the image,
        device to device; returns the number of columns written"""
        e = self.engine
        if max_frames is None:
            max_frames = e.num_frames(self.capacity)
        e.use_current_stream()
        got = C.c_size_t(0)
        e._check(self._lib.sgx_live_tick_image(self._h, image._h, max_frames, C.byref(got)))
        return int(got.value)

    def tick_into(self, view: "ViewRing", max_frames: Optional[int] = None) -> int:
        """One GUI tick of the default widget (gpu_spectrogram.rs:255-275): every complete frame of the ring goes, as a half-pair
        row, straight into `view`'s ring texture -- device to device, no host copy.  Returns the number of rows appended."""
        e = self.engine
        if max_frames is None:
            max_frames = e.num_frames(self.capacity)
        got = C.c_size_t(0)
        e._check(self._lib.sgx_live_tick_view(self._h, view._h, max_frames, C.byref(got)))
        return int(got.value)


class ImageRing:
    """sgx_image: the width x rows RGBA Pixbuf of simple_spectrogram.rs:89-94 on the device, written one pixel column per frame at
    `offset` (:140-164) and read back as it lies or as the scrolling picture of :181-209."""

    def __init__(self, engine: SpectrogramEngine, width: int = 1024):
        self.engine = engine
        self.width, self.height = int(width), engine.R
        self._lib = engine._lib
        self._h = C.c_void_p()
        engine._check(self._lib.sgx_image_create(engine._ctx, self.width, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.sgx_image_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def offset(self) -> int:
        return int(self._lib.sgx_image_offset(self._h))

    def write_columns(self, rgba) -> int:
        """append [n][rows][4] uint8 pixel columns (a device tensor, e.g. render_batch(...)[:, 0]); returns the new offset"""
        import torch

        e = self.engine
        assert rgba.is_cuda and rgba.device == e.device and rgba.dtype == torch.uint8 and rgba.is_contiguous()
        n = rgba.numel() // (e.R * 4)
        assert rgba.numel() == n * e.R * 4
        e.use_current_stream()
        off = C.c_uint32(0)
        e._check(self._lib.sgx_image_write_columns(self._h, C.c_void_p(rgba.data_ptr()), n, C.byref(off)))
        return int(off.value)

    def read(self, scrolled: bool = False, out=None):
        """[rows][width][4] uint8: the buffer as it lies, or (scrolled) columns [offset, width) followed by [0, offset)"""
        import torch

        e = self.engine
        out = e._out(out, (self.height, self.width, 4), torch.uint8)
        e._check(self._lib.sgx_image_read(self._h, 1 if scrolled else 0, C.c_void_p(out.data_ptr())))
        return out


class ViewRing:
    """sgx_view: the VIEWPORT_FRAMES x (W - 1) F16F16 texture of gpu_spectrogram.rs:218-226 used as a ring (:255-275)
    and the fragment program of :150-186 as a kernel."""

    def __init__(self, engine: SpectrogramEngine, viewport_frames: int = 2048):
        self.engine = engine
        self.rows = int(viewport_frames)
        self._lib = engine._lib
        self._h = C.c_void_p()
        engine._check(self._lib.sgx_view_create(engine._ctx, self.rows, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.sgx_view_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def offset(self) -> int:
        return int(self._lib.sgx_view_offset(self._h))

    def write_rows(self, rows_f16) -> int:
        """append [n][M][2] float16 rows (a device tensor, e.g. stft_batch_f16(...)[:, 0]); returns the new offset"""
        import torch

        e = self.engine
        assert rows_f16.is_cuda and rows_f16.device == e.device and rows_f16.dtype == torch.float16 and rows_f16.is_contiguous()
        n = rows_f16.numel() // (e.M * 2)
        assert rows_f16.numel() == n * e.M * 2
        e.use_current_stream()
        off = C.c_uint32(0)
        e._check(self._lib.sgx_view_write_rows(self._h, C.c_void_p(rows_f16.data_ptr()), n, C.byref(off)))
        return int(off.value)

    def draw(self, width: int, height: int, out=None):
        """the fragment program over a width x height viewport -> [height][width][4] float32, row 0 at the bottom"""
        import torch

        e = self.engine
        out = e._out(out, (height, width, 4), torch.float32)
        e._check(self._lib.sgx_view_draw(self._h, width, height, C.c_void_p(out.data_ptr())))
        return out


def builtin_gradient_eval(name: str, t: float):
    """eval_continuous(t) of a built-in gradient -> (r, g, b)"""
    lib = _lib.load()
    out = (C.c_uint8 * 3)()
    if lib.sgx_builtin_gradient_eval(name.encode(), float(t), out) != 0:
        raise KeyError(name)
    return tuple(int(x) for x in out)


def builtin_gradient(name: str) -> np.ndarray:
    lib = _lib.load()
    out = np.empty((256, 3), np.uint8)
    if lib.sgx_builtin_gradient(name.encode(), out.ctypes.data_as(C.c_void_p)) != 0:
        raise KeyError(name)
    return out
