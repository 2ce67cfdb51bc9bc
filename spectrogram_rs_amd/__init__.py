"""spectrogram_rs_amd -- MI355X (gfx950) streaming-STFT spectrogram engine.

A from-scratch HIP implementation of the one data-parallel hot path of
JacksonCampolattaro/spectrogram-rs, behind a C ABI (include/sgx.h, libsgx.so):

    PCM -> Hann + 2x zero-pad -> c2c FFT -> stereo magnitudes -> log-frequency resample
        -> dB -> colour ramp -> RGBA pixel columns

Python here is the host-side mirror of the reference's interface for that path
(fourier::{AudioTransform, FastFourierTransform, AudioStreamTransform},
colorscheme::ColorScheme, log_scaling::LogCoordf64, widgets::SimpleSpectrogram's pixel loop) plus
device-memory plumbing via torch.  There is no CPU fallback.
"""
from ._lib import (INTERP_COSINE, INTERP_CUBIC, LIB_PATH, LUT_FLOOR_N, LUT_ROUND_NM1, SgxError)  # noqa: F401
from .engine import LiveRing, SpectrogramEngine, ViewRing, builtin_gradient  # noqa: F401
from .fourier import AudioStreamTransform, AudioTransform, FastFourierTransform, RingBuffer  # noqa: F401
from .colorscheme import ColorScheme, default_color_schemes  # noqa: F401
from .log_scaling import LogCoordf64  # noqa: F401
from .widgets import GPUSpectrogram, SimpleSpectrogram, SpectrumAnalyzer  # noqa: F401

__all__ = [
    "SpectrogramEngine", "builtin_gradient", "AudioTransform", "FastFourierTransform", "AudioStreamTransform",
    "RingBuffer", "ColorScheme", "default_color_schemes", "LogCoordf64", "SimpleSpectrogram", "GPUSpectrogram", "SpectrumAnalyzer", "LiveRing", "ViewRing", "SgxError",
    "INTERP_CUBIC", "INTERP_COSINE", "LUT_FLOOR_N", "LUT_ROUND_NM1", "LIB_PATH",
]
