"""Host-side mirror of the reference's `fourier` module, backed by the HIP engine.

  trait AudioTransform                      src/fourier/audio_transform.rs:4-11
  struct AudioStreamTransform<T>            src/fourier/audio_transform.rs:14-43
  struct FastFourierTransform               src/fourier/fft.rs:11-99
  type StereoMagnitude = (f32, f32)         src/fourier/mod.rs:13

Names, argument meaning and the None-on-short-input rule follow the reference; the arithmetic
runs in libsgx.so's kernels (no CPU fallback).
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Iterable, Iterator, Optional

import numpy as np

from .engine import SpectrogramEngine


def _f32(x: float) -> np.float32:
    return np.float32(x)


def _as_usize(v: np.float32) -> int:
    """Rust `f32 as usize`: truncate toward zero, saturate, NaN -> 0."""
    v = float(v)
    if not (v > 0.0):
        return 0
    return int(min(v, 2.0**63))


class RingBuffer:
    """The SPSC ring the capture thread fills (ringbuf::HeapRb<(f32, f32)>,
    src/devices/audio_input_list_model.rs:30,63-72): push_iter drops what does not fit, iter()
    peeks without consuming, skip(n) consumes at most what is there."""

    def __init__(self, capacity: int):
        self.capacity = int(capacity)
        self._data = np.zeros((0, 2), np.float32)

    def push_iter(self, samples) -> int:
        s = np.asarray(list(samples) if not isinstance(samples, np.ndarray) else samples, np.float32).reshape(-1, 2)
        room = self.capacity - len(self._data)
        s = s[:max(room, 0)]
        self._data = np.concatenate([self._data, s])
        return len(s)

    def push_mono(self, samples) -> int:
        """mono -> (s, s), audio_input_list_model.rs:67-69"""
        s = np.asarray(samples, np.float32).reshape(-1)
        return self.push_iter(np.stack([s, s], axis=1))

    def iter(self) -> np.ndarray:
        return self._data

    def skip(self, n: int) -> int:
        n = min(int(n), len(self._data))
        self._data = self._data[n:]
        return n

    def __len__(self) -> int:
        return len(self._data)


class AudioTransform(ABC):
    """audio_transform.rs:4-11"""

    @abstractmethod
    def sample_rate(self) -> float: ...

    @abstractmethod
    def num_input_samples(self) -> int: ...

    @abstractmethod
    def process(self, samples: Iterable) -> Optional[np.ndarray]: ...


class FastFourierTransform(AudioTransform):
    """fft.rs:11-99.  `process` returns Vec<StereoMagnitude> as an [M][2] float32 array."""

    def __init__(self, sample_rate: float, period: float, *, device: Optional[int] = None,
                 force_generic: bool = False):
        self._sample_rate = _f32(sample_rate)
        self._period = _f32(period)
        self._device = device
        self._force_generic = force_generic
        self._engines = {}
        self._per_frame = self._engine(1)  # "planning" happens at construction, as FFTW's does (fft.rs:20-24)

    # engines are cached per hop: the wrapper's `stride` is a public, mutable field
    def _engine(self, hop: int) -> SpectrogramEngine:
        hop = max(int(hop), 1)
        if hop not in self._engines:
            self._engines[hop] = SpectrogramEngine(float(self._sample_rate), window_samples=self.num_input_samples(),
                                                   hop_samples=hop, channels=2, device=self._device,
                                                   force_generic=self._force_generic)
        return self._engines[hop]

    def sample_rate(self) -> float:
        return float(self._sample_rate)

    def num_input_samples(self) -> int:
        # fft.rs:41 -- (self.period * self.sample_rate) as usize, f32 product
        return _as_usize(self._period * self._sample_rate)

    def num_output_frequencies(self) -> int:
        return self.num_input_samples() - 1  # fft.rs:33

    def process(self, samples: Iterable) -> Optional[np.ndarray]:
        W = self.num_input_samples()
        if isinstance(samples, np.ndarray):
            lr = samples.reshape(-1, 2)[:W]  # .take(W), fft.rs:50
        else:
            taken = []
            for s in samples:
                if len(taken) >= W:
                    break
                taken.append(s)
            lr = np.asarray(taken, np.float32).reshape(-1, 2)
        if lr.shape[0] < W:
            return None  # fft.rs:72
        return self._per_frame.process_one(lr)

    def process_stream(self, lr: np.ndarray, hop: int) -> np.ndarray:
        """All complete frames of an (l, r) buffer in one launch: [frames][M][2]."""
        import torch

        eng = self._engine(hop)
        lr = np.ascontiguousarray(lr, np.float32).reshape(-1, 2)
        if eng.num_frames(lr.shape[0]) == 0:
            return np.zeros((0, eng.M, 2), np.float32)
        pcm = torch.from_numpy(lr).to(eng.device)
        out = eng.stft_batch(pcm.reshape(-1))
        return out[:, 0].cpu().numpy()


class AudioStreamTransform:
    """audio_transform.rs:14-43.  `input_stream`, `transform` and `stride` are public fields that
    callers assign directly (gpu_spectrogram.rs:34,321; simple_spectrogram.rs:46,215)."""

    def __init__(self, input_stream: RingBuffer, transform: AudioTransform, stride: float):
        self.input_stream = input_stream
        self.transform = transform
        self.stride = stride

    def stride_samples(self) -> int:
        # audio_transform.rs:35 -- (self.stride * self.transform.sample_rate()) as usize
        return _as_usize(_f32(self.stride) * _f32(self.transform.sample_rate()))

    def process(self) -> Iterator[np.ndarray]:
        """Yields one [M][2] array per frame, exactly the frames the reference's
        repeat_with/take_while loop yields.  All frames available in the ring are transformed in
        ONE batched launch; the ring is then advanced as the reference advances it, including
        the skip that its terminating short read performs (audio_transform.rs:37-41)."""
        H = self.stride_samples()
        W = self.transform.num_input_samples()
        n = len(self.input_stream)
        if H == 0:
            # skip(0) never advances: the reference would loop forever on a full ring; refuse.
            raise ValueError("stride * sample_rate truncates to 0 samples")
        frames = 0 if n < W else (n - W) // H + 1
        if frames and hasattr(self.transform, "process_stream"):
            out = self.transform.process_stream(self.input_stream.iter()[:(frames - 1) * H + W], H)
        else:
            out = []
            for t in range(frames):
                out.append(self.transform.process(self.input_stream.iter()[t * H:]))
        self.input_stream.skip((frames + 1) * H)
        for t in range(frames):
            yield out[t]
