"""ctypes front-end of the CPU oracle (oracle/spectro_oracle.c).

TEST INFRASTRUCTURE -- PARITY UNPINNED (see spectro_oracle.h).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; nothing under
spectrogram_rs_amd/ does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("ORACLE_SO") or os.path.join(_HERE, "liboracle.so")  # ORACLE_SO: sanitizer build (oracle/Makefile)

F32, F64 = 0, 1
INTERP_CUBIC, INTERP_COSINE = 0, 1
LUT_FLOOR_N, LUT_ROUND_NM1 = 0, 1


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("spectro_oracle.c", "spectro_oracle.h", "Makefile")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-s", "-B" if force else "-s"], check=True)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        fp = C.POINTER(C.c_float)
        dp = C.POINTER(C.c_double)
        u8p = C.POINTER(C.c_uint8)
        L.orc_window_samples.restype = C.c_size_t
        L.orc_window_samples.argtypes = [C.c_float, C.c_float]
        L.orc_hop_samples.restype = C.c_size_t
        L.orc_hop_samples.argtypes = [C.c_float, C.c_float]
        L.orc_num_frames.restype = C.c_size_t
        L.orc_num_frames.argtypes = [C.c_size_t] * 3
        L.orc_hann_window.restype = None
        L.orc_hann_window.argtypes = [C.c_size_t, fp]
        L.orc_fft_process.restype = C.c_int
        L.orc_fft_process.argtypes = [fp, C.c_size_t, C.c_size_t, C.c_int, fp, dp]
        L.orc_stream_process.restype = C.c_size_t
        L.orc_stream_process.argtypes = [fp, C.c_size_t, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                                         C.c_int, C.c_int, fp]
        L.orc_fftw_available.restype = C.c_int
        L.orc_fftw_available.argtypes = []
        L.orc_fftw_stream_process.restype = C.c_size_t
        L.orc_fftw_stream_process.argtypes = [fp, C.c_size_t, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                                              C.c_int, C.c_int, fp]
        L.orc_period.restype = C.c_float
        L.orc_period.argtypes = [C.c_size_t, C.c_uint32]
        L.orc_index_of.restype = C.c_float
        L.orc_index_of.argtypes = [C.c_float, C.c_size_t, C.c_uint32]
        L.orc_cubic_interpolate.restype = None
        L.orc_cubic_interpolate.argtypes = [fp, C.c_size_t, C.c_float, fp]
        L.orc_cosine_interpolate.restype = None
        L.orc_cosine_interpolate.argtypes = [fp, C.c_size_t, C.c_float, fp]
        L.orc_magnitude_in.restype = None
        L.orc_magnitude_in.argtypes = [fp, C.c_size_t, C.c_uint32, C.c_float, C.c_float, C.c_int, fp]
        L.orc_num_samples_in.restype = C.c_size_t
        L.orc_num_samples_in.argtypes = [C.c_size_t, C.c_uint32, C.c_float, C.c_float]
        L.orc_log_unmap.restype = C.c_double
        L.orc_log_unmap.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int]
        L.orc_color_for.restype = None
        L.orc_color_for.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, u8p, fp]
        L.orc_lut_index.restype = C.c_int
        L.orc_lut_index.argtypes = [C.c_double, C.c_int, C.c_int]
        L.orc_alpha_u8.restype = C.c_uint8
        L.orc_alpha_u8.argtypes = [C.c_float]
        L.orc_render_column.restype = None
        L.orc_render_column.argtypes = [fp, C.c_size_t, C.c_uint32, C.c_int, C.c_double, C.c_double, C.c_int, u8p,
                                        C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, u8p]
        L.orc_glsl_fragments.restype = None
        L.orc_glsl_fragments.argtypes = [C.POINTER(C.c_uint16), C.c_size_t, C.c_size_t, C.c_size_t, fp, C.c_float, C.c_float, C.c_size_t, C.c_size_t, fp]
        L.orc_lookup_table.restype = None
        L.orc_lookup_table.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, fp]
        L.orc_set_gradient_fn.restype = None
        L.orc_set_gradient_fn.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_log_space.restype = C.c_float
        L.orc_log_space.argtypes = [C.c_float, C.c_float, C.c_size_t, C.c_float, C.c_size_t]
        L.orc_spectrum_levels.restype = None
        L.orc_spectrum_levels.argtypes = [fp, C.c_size_t, C.c_uint32, C.c_int, C.c_size_t, dp]
        L.orc_lowbias32.restype = C.c_uint32
        L.orc_lowbias32.argtypes = [C.c_uint32]
        L.orc_white_noise.restype = None
        L.orc_white_noise.argtypes = [C.c_uint32, C.c_uint64, C.c_size_t, fp]
        L.orc_sine_sweep.restype = None
        L.orc_sine_sweep.argtypes = [C.c_size_t, C.c_size_t, fp]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u8p(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_uint8))


GRADIENT_FN = C.CFUNCTYPE(None, C.c_double, C.POINTER(C.c_uint8), C.c_void_p)
_gradient_cb = None


def set_gradient_fn(fn):
    """Continuous gradient fn(t) -> (r, g, b), used wherever a gradient TABLE argument is None."""
    global _gradient_cb
    if fn is None:
        _gradient_cb = None
        lib().orc_set_gradient_fn(None, None)
        return

    def thunk(t, out, _user):
        r, g, b = fn(t)
        out[0], out[1], out[2] = int(r), int(g), int(b)
    _gradient_cb = GRADIENT_FN(thunk)
    lib().orc_set_gradient_fn(C.cast(_gradient_cb, C.c_void_p), None)


def _grad(g):
    """(table or None) -> (contiguous [n][3] u8 or None, n)"""
    if g is None:
        return None, 0
    g = np.ascontiguousarray(g, np.uint8).reshape(-1, 3)
    return g, g.shape[0]


def _f32c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---- sizes --------------------------------------------------------------------------------
def window_samples(sample_rate: float, period: float) -> int:
    return int(lib().orc_window_samples(sample_rate, period))


def hop_samples(sample_rate: float, stride: float) -> int:
    return int(lib().orc_hop_samples(sample_rate, stride))


def num_frames(n: int, W: int, H: int) -> int:
    return int(lib().orc_num_frames(n, W, H))


def hann_window(W: int) -> np.ndarray:
    out = np.empty(W, np.float32)
    lib().orc_hann_window(W, _fp(out))
    return out


# ---- FastFourierTransform::process ---------------------------------------------------------
def fft_process(lr: np.ndarray, W: int, precision: int = F32):
    """lr: [n][2] float32.  Returns None (fewer than W samples) or [W-1][2] (float32, or float64
    for precision == F64)."""
    lr = _f32c(lr).reshape(-1, 2)
    out = np.empty((max(W - 1, 0), 2), np.float32)
    out64 = np.empty((max(W - 1, 0), 2), np.float64)
    ok = lib().orc_fft_process(_fp(lr), lr.shape[0], W, precision, _fp(out),
                               out64.ctypes.data_as(C.POINTER(C.c_double)))
    if not ok:
        return None
    return out64 if precision == F64 else out


def stream_process(pcm: np.ndarray, channels: int, W: int, H: int, first: int = 0, count: int | None = None,
                   precision: int = F32, threads: int = 1) -> np.ndarray:
    """pcm: [n][channels] (or flat).  Returns [frames][pairs][W-1][2] float32."""
    pcm = _f32c(pcm).reshape(-1)
    n = pcm.shape[0] // channels
    total = num_frames(n, W, H)
    if count is None:
        count = max(total - first, 0)
    count = max(min(count, total - first), 0)
    pairs = 1 if channels == 1 else channels // 2
    out = np.empty((count, pairs, W - 1, 2), np.float32)
    if count:
        got = lib().orc_stream_process(_fp(pcm), n, channels, W, H, first, count, precision, threads, _fp(out))
        assert got == count
    return out


def fftw_available() -> bool:
    """does this host have libfftw3f.so.3 (dlopen)?  The image this was built in does not."""
    return bool(lib().orc_fftw_available())


def fftw_stream_process(pcm: np.ndarray, channels: int, W: int, H: int, threads: int = 1, as_written: bool = True):
    """stream_process through FFTW itself (fft.rs:20-24,68,76-77); None when the host has no libfftw3f."""
    pcm = _f32c(pcm).reshape(-1)
    n = pcm.shape[0] // channels
    count = num_frames(n, W, H)
    pairs = 1 if channels == 1 else channels // 2
    out = np.empty((count, pairs, W - 1, 2), np.float32)
    got = lib().orc_fftw_stream_process(_fp(pcm), n, channels, W, H, 0, count, threads, int(as_written), _fp(out))
    if got == C.c_size_t(-1).value:
        return None
    assert got == count
    return out


# ---- InterpolatedFrequencySample -----------------------------------------------------------
def period(M: int, sample_rate: int) -> float:
    return float(lib().orc_period(M, sample_rate))


def index_of(f: float, M: int, sample_rate: int) -> float:
    return float(lib().orc_index_of(f, M, sample_rate))


def cubic_interpolate(data: np.ndarray, index: float) -> np.ndarray:
    data = _f32c(data).reshape(-1, 2)
    out = np.empty(2, np.float32)
    lib().orc_cubic_interpolate(_fp(data), data.shape[0], index, _fp(out))
    return out


def cosine_interpolate(data: np.ndarray, index: float) -> np.ndarray:
    data = _f32c(data).reshape(-1, 2)
    out = np.empty(2, np.float32)
    lib().orc_cosine_interpolate(_fp(data), data.shape[0], index, _fp(out))
    return out


def magnitude_in(data: np.ndarray, sample_rate: int, f0: float, f1: float, interp: int = INTERP_CUBIC) -> np.ndarray:
    data = _f32c(data).reshape(-1, 2)
    out = np.empty(2, np.float32)
    lib().orc_magnitude_in(_fp(data), data.shape[0], sample_rate, f0, f1, interp, _fp(out))
    return out


def num_samples_in(M: int, sample_rate: int, f0: float, f1: float) -> int:
    return int(lib().orc_num_samples_in(M, sample_rate, f0, f1))


def log_space(start: float, end: float, n: int, base: float, i: int) -> float:
    return float(lib().orc_log_space(start, end, n, base, i))


def spectrum_levels(mags: np.ndarray, sample_rate: int, levels: np.ndarray, interp: int = INTERP_CUBIC) -> np.ndarray:
    """SpectrumAnalyzer::push_frequencies: `levels` (float64) is updated in place and returned."""
    data = _f32c(mags).reshape(-1, 2)
    assert levels.dtype == np.float64 and levels.flags.c_contiguous
    lib().orc_spectrum_levels(_fp(data), data.shape[0], sample_rate, interp, levels.size,
                              levels.ctypes.data_as(C.POINTER(C.c_double)))
    return levels


# ---- LogCoordf64 ---------------------------------------------------------------------------
def log_unmap(f_min: float, f_max: float, p: int, pmin: int, pmax: int, zero_point: float = 0.0) -> float:
    return float(lib().orc_log_unmap(f_min, f_max, zero_point, p, pmin, pmax))


def bin_edges(R: int, f_min: float = 32.0, f_max: float = 22030.0) -> np.ndarray:
    """R+1 row edges, f64 -> f32 exactly as simple_spectrogram.rs:142-145 does."""
    return np.array([np.float32(log_unmap(f_min, f_max, p, 0, R)) for p in range(R + 1)], np.float32)


# ---- ColorScheme ---------------------------------------------------------------------------
def lut_index(t: float, n_lut: int = 256, mode: int = LUT_FLOOR_N) -> int:
    return int(lib().orc_lut_index(t, n_lut, mode))


def alpha_u8(alpha: float) -> int:
    return int(lib().orc_alpha_u8(alpha))


def color_for(gradient: np.ndarray, l: float, r: float, stereo: bool = False, min_db: float = -70.0,
              max_db: float = -10.0, mode: int = LUT_FLOOR_N):
    g, n = _grad(gradient)
    rgb = np.empty(3, np.uint8)
    alpha = C.c_float(0)
    lib().orc_color_for(_u8p(g), n, mode, int(stereo), min_db, max_db, l, r, _u8p(rgb), C.byref(alpha))
    return rgb, float(alpha.value)


def render_column(mags: np.ndarray, sample_rate: int, gradient: np.ndarray, R: int = 1024, f_min: float = 32.0,
                  f_max: float = 22030.0, interp: int = INTERP_CUBIC, stereo: bool = False, min_db: float = -70.0,
                  max_db: float = -10.0, mode: int = LUT_FLOOR_N) -> np.ndarray:
    """mags [M][2] -> rgba [R][4] u8 indexed by image row (row 0 = highest frequency)."""
    mags = _f32c(mags).reshape(-1, 2)
    g, n = _grad(gradient)
    out = np.empty((R, 4), np.uint8)
    lib().orc_render_column(_fp(mags), mags.shape[0], sample_rate, R, f_min, f_max, interp, _u8p(g), n,
                            mode, int(stereo), min_db, max_db, _u8p(out))
    return out


def render_columns(mags: np.ndarray, sample_rate: int, gradient: np.ndarray, **kw) -> np.ndarray:
    """mags [F][M][2] -> [F][R][4]."""
    mags = _f32c(mags)
    return np.stack([render_column(m, sample_rate, gradient, **kw) for m in mags])


def lookup_table(gradient: np.ndarray, resolution: int = 32, stereo: bool = False, mode: int = LUT_FLOOR_N) -> np.ndarray:
    g, n = _grad(gradient)
    out = np.empty((resolution, resolution, 4), np.float32)
    lib().orc_lookup_table(_u8p(g), n, mode, int(stereo), resolution, _fp(out))
    return out


# ---- synthetic inputs ----------------------------------------------------------------------
def glsl_fragments(ring_f16: np.ndarray, offset: int, palette32: np.ndarray, width: int, height: int,
                   min_db: float = -70.0, max_db: float = -10.0) -> np.ndarray:
    """The fragment program of GPUSpectrogram (gpu_spectrogram.rs:150-186): ring_f16 [rows][M][2] float16, palette32
    [32][32][4] float32 = lookup_table(32); returns [height][width][4] float32, row 0 at the bottom."""
    ring = np.ascontiguousarray(ring_f16, np.float16)
    rows, M = ring.shape[0], ring.shape[1]
    pal = np.ascontiguousarray(palette32, np.float32).reshape(32, 32, 4)
    out = np.empty((height, width, 4), np.float32)
    lib().orc_glsl_fragments(ring.view(np.uint16).ctypes.data_as(C.POINTER(C.c_uint16)), M, rows, offset, _fp(pal),
                             min_db, max_db, width, height, _fp(out))
    return out


def white_noise(n: int, first: int = 0, seed: int = 0x5EED0001) -> np.ndarray:
    out = np.empty(n, np.float32)
    lib().orc_white_noise(seed, first, n, _fp(out))
    return out


def sine_sweep(n: int, first: int = 0) -> np.ndarray:
    out = np.empty(n, np.float32)
    lib().orc_sine_sweep(first, n, _fp(out))
    return out


# ---- independent float64 numpy restatement of the FFT stage (cross-implementation check) ----
def np_truth_frame(lr: np.ndarray, W: int) -> np.ndarray:
    """fft.rs:43-99 with numpy's float64 FFT on the reference's exact f32-windowed input."""
    lr = _f32c(lr).reshape(-1, 2)[:W]
    win = hann_window(W)
    z = (lr[:, 0] * win).astype(np.float64) + 1j * (lr[:, 1] * win).astype(np.float64)
    P = 2 * W
    F = np.fft.fft(np.concatenate([z, np.zeros(W, np.complex128)]))
    k = np.arange(1, W)
    a, b = F[k], F[P - k]
    left = np.abs(a + np.conj(b)) / 2.0
    right = np.abs(a - np.conj(b)) / 2.0
    return np.stack([left, right], axis=1) * (2.0 / W)
