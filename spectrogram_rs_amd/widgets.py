"""Host-side mirror of the pixel loop of src/widgets/simple_spectrogram.rs.

SimpleSpectrogram keeps a TEXTURE_WIDTH x TEXTURE_HEIGHT RGBA image that is used as a ring of
pixel columns (simple_spectrogram.rs:34-35,89-94,164).  `snapshot()` here does what the first half
of the reference's `snapshot` does (:120-165): drain the ring through the transform, turn every
frame into one pixel column and advance `offset`.  Drawing the two sub-pixbufs (:181-209) is GTK
and out of scope; `scrolled()` returns the same image they would compose.

SpectrumAnalyzer mirrors src/widgets/spectrum_analyzer.rs: 128 level bars over log-spaced bands.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from .colorscheme import ColorScheme
from .engine import SpectrogramEngine
from .fourier import RingBuffer

TEXTURE_WIDTH = 1024   # simple_spectrogram.rs:34
TEXTURE_HEIGHT = 1024  # simple_spectrogram.rs:35


class SimpleSpectrogram:
    """`sample_stream` is the host RingBuffer mirror, or None: the widget then owns a device-resident LiveRing
    (engine.live(), filled through `push`), so that a tick moves only the new samples to the GPU."""

    def __init__(self, sample_stream: Optional[RingBuffer], *, sample_rate: int = 48000, period: float = 0.05,
                 window_samples: int = 0, device: Optional[int] = None, interp: int = 0,
                 width: int = TEXTURE_WIDTH, height: int = TEXTURE_HEIGHT):
        import torch

        self.input_stream = sample_stream
        self.width, self.height = width, height
        self.palette = ColorScheme.new_mono("magma", "magma")  # :95
        self.image = None   # the Pixbuf of :89-94 as a device-resident ring (engine.image: sgx_image_*)
        self._period, self._window_samples, self._device, self._interp = period, window_samples, device, interp
        self.stride = 2.0 / width  # :102
        self.engine: Optional[SpectrogramEngine] = None
        self.live = None
        self.set_sample_rate(sample_rate)

    def set_sample_rate(self, sample_rate: int) -> None:
        """:214-219 -- replaces the transform wholesale"""
        sr = np.float32(sample_rate)
        hop = int(np.float32(self.stride) * sr)
        # the ring holds a pointer to the engine's context: it goes first (include/sgx.h: destroy every ring
        # before sgx_destroy).  The reference calls this on every input-device change.
        if self.live is not None:
            self.live.close()
            self.live = None
        kept = None   # the reference's Pixbuf outlives the transform (:214-219 replace `fft` only): carry the picture over
        if self.image is not None:
            kept = (self.image.read().permute(1, 0, 2).contiguous(), self.image.offset)
            self.image.close()
            self.image = None
        if self.engine is not None:
            self.engine.close()
        kw = dict(window_samples=self._window_samples) if self._window_samples else dict(period=self._period)
        self.engine = SpectrogramEngine(float(sr), hop_samples=max(hop, 1), channels=2, rows=self.height,
                                        interp=self._interp, device=self._device, **kw)
        self.palette.apply(self.engine)
        self.image = self.engine.image(self.width)
        if kept is not None:
            import torch

            cols, k = kept                      # columns in x order; k dummy columns first, so that the ring ends at offset k again
            cols = cols.to(self.engine.device)
            self.image.write_columns(cols[:k].contiguous())
            self.image.write_columns(torch.cat([cols[k:], cols[:k]]).contiguous())
        # the reference's ring outlives the transform; the device-resident one is rebuilt with it (samples in
        # flight at a sample-rate change are dropped)
        self.live = self.engine.live(max(4096, 2 * self.engine.W), reference_skip=True) if self.input_stream is None else None

    def push(self, data, channels: int) -> int:
        """the input callback (audio_input_list_model.rs:63-75), LiveRing mode only"""
        return self.live.push(data, channels)

    def set_palette(self, palette: ColorScheme) -> None:
        self.palette = palette
        palette.apply(self.engine)

    def snapshot(self) -> int:
        """Drain the ring; returns the number of pixel columns written."""
        import torch

        eng = self.engine
        if self.live is not None:
            return self.live.tick_image(self.image)     # device to device: only the new samples cross the bus
        n = len(self.input_stream)
        frames = eng.num_frames(n)
        if frames:
            lr = self.input_stream.iter()[:(frames - 1) * eng.H + eng.W]
            pcm = torch.from_numpy(np.ascontiguousarray(lr)).to(eng.device).reshape(-1)
            cols = eng.render_batch(pcm)[:, 0].contiguous()        # [frames][R][4], image-row order
            self.image.write_columns(cols)                          # put_pixel column by column + offset = (px + 1) % width (:150-164)
        self.input_stream.skip((frames + 1) * eng.H)                # audio_transform.rs:37-41
        return frames

    def scrolled(self):
        """The image the two append_scaled_texture calls compose (:181-209): columns
        [offset, width) followed by [0, offset)."""
        return self.image.read(scrolled=True)

    @property
    def offset(self) -> int:
        return self.image.offset

    @property
    def buffer(self):
        """[height][width][4] uint8: the Pixbuf as it lies"""
        return self.image.read()


VIEWPORT_FRAMES = 2048                                                        # gpu_spectrogram.rs:20
VIEWPORT_SECONDS = np.float32(2.5)                                            # :21
FRAMES_PER_SECOND = np.float32(np.float32(VIEWPORT_FRAMES) / VIEWPORT_SECONDS)  # :22 (f32: 819.2)


class GPUSpectrogram:
    """Mirror of the reference's DEFAULT visualiser (src/widgets/gpu_spectrogram.rs): frames go, as F16F16 rows, into a
    VIEWPORT_FRAMES-row ring texture (:255-275); the picture is made per fragment from that texture and the 32 x 32
    palette texture (:150-186).  Texture and fragment program live in the engine (sgx_view); the capture ring is the
    engine's LiveRing (fed through `push`), ticked for half-precision rows."""

    def __init__(self, *, sample_rate: int = 48000, period: float = 0.05, device: Optional[int] = None,
                 viewport_frames: int = VIEWPORT_FRAMES):
        self._period, self._device, self._viewport = period, device, viewport_frames
        self.palette = ColorScheme.new_mono("magma", "magma")          # :40 (ColorScheme::new_mono(MAGMA, "magma"))
        self.stride = np.float32(np.float32(1.0) / FRAMES_PER_SECOND)      # :81: 1f32 / FRAMES_PER_SECOND (58 samples at 48 kHz)
        self.engine: Optional[SpectrogramEngine] = None
        self.live = None
        self.texture = None
        self.set_sample_rate(sample_rate)

    def set_sample_rate(self, sample_rate: int) -> None:
        """:320-327 -- the transform is replaced and the texture is rebuilt (fft_texture.set(None))"""
        sr = np.float32(sample_rate)
        for h in (self.texture, self.live):
            if h is not None:
                h.close()
        if self.engine is not None:
            self.engine.close()
        hop = max(int(np.float32(self.stride) * sr), 1)
        self.engine = SpectrogramEngine(float(sr), period=self._period, hop_samples=hop, channels=2, device=self._device)
        self.palette.apply(self.engine)
        self.live = self.engine.live(max(4096, 2 * self.engine.W), reference_skip=True)
        self.texture = self.engine.view(self._viewport)

    def set_palette(self, palette: ColorScheme) -> None:
        """:329-333"""
        self.palette = palette
        palette.apply(self.engine)

    def push(self, data, channels: int) -> int:
        return self.live.push(data, channels)

    def render(self, width: int, height: int):
        """:208-316 without the GL frame: upload this tick's frames, run the fragment program; [height][width][4] float32"""
        self.live.tick_into(self.texture)      # device to device: the rows never visit the host
        return self.texture.draw(width, height)


class SpectrumAnalyzer:
    """spectrum_analyzer.rs:38-68: `level_bars` LevelBar values (a fresh bar holds 0.3, :92); push_frequencies
    takes one frame of magnitudes ([M][2] on the engine's device) and raises / decays every bar."""

    def __init__(self, engine: SpectrogramEngine, n_bars: int = 128):
        self.engine = engine
        self.level_bars = np.full(n_bars, 0.3, np.float64)

    def push_frequencies(self, frequency_sample) -> np.ndarray:
        return self.engine.spectrum_levels(frequency_sample, self.level_bars)
