"""ctypes binding of libsgx.so (the C ABI in include/sgx.h).

There is no CPU fallback: if the shared library is missing, or no HIP device is present when a
context is created, the error is raised -- loudly -- to the caller.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SGX_LIB", os.path.join(_HERE, "libsgx.so"))  # SGX_LIB: A/B builds during kernel development

SGX_OK = 0
SGX_ERR_INVALID_ARG = -1
SGX_ERR_UNSUPPORTED = -2
SGX_ERR_HIP = -3
SGX_ERR_NOMEM = -4
SGX_ERR_NO_DEVICE = -5

INTERP_CUBIC, INTERP_COSINE = 0, 1
LUT_FLOOR_N, LUT_ROUND_NM1 = 0, 1
FLAG_FORCE_GENERIC = 1
FLAG_NO_FUSED_RENDER = 4
FLAG_LUT_WALK = 64
FLAG_MIXED_GENERIC = 256
FLAG_COMPLEX_MONO = 512
FLAG_PAIRED_FRAMES = 1024
LIVE_MAGS, LIVE_MAGS_F16, LIVE_RGBA = 0, 1, 2
LIVE_REFERENCE_SKIP = 1


class SgxError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"sgx error {code}: {message}")
        self.code = code


class sgx_config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("sample_rate", C.c_float),
        ("period", C.c_float),
        ("stride", C.c_float),
        ("window_samples", C.c_uint32),
        ("hop_samples", C.c_uint32),
        ("channels", C.c_uint32),
        ("rows", C.c_uint32),
        ("f_min", C.c_double),
        ("f_max", C.c_double),
        ("min_db", C.c_float),
        ("max_db", C.c_float),
        ("interp", C.c_uint32),
        ("lut_index_mode", C.c_uint32),
        ("device", C.c_int32),
        ("flags", C.c_uint32),
    ]


class sgx_info(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("window_samples", C.c_uint32),
        ("fft_length", C.c_uint32),
        ("num_frequencies", C.c_uint32),
        ("hop_samples", C.c_uint32),
        ("channels", C.c_uint32),
        ("pairs", C.c_uint32),
        ("rows", C.c_uint32),
        ("sample_rate_u32", C.c_uint32),
        ("total_samples_per_column", C.c_uint32),
        ("stft_kernel", C.c_uint32),
        ("render_path", C.c_uint32),
        ("mags_bytes_per_frame", C.c_uint64),
        ("rgba_bytes_per_frame", C.c_uint64),
    ]


# every symbol include/sgx.h declares: (name, restype, argtypes)
_vp, _sz = C.c_void_p, C.c_size_t
_ctx = C.c_void_p
SIGNATURES = [
    ("sgx_version", C.c_char_p, []),
    ("sgx_config_init", C.c_int, [C.POINTER(sgx_config)]),
    ("sgx_create", C.c_int, [C.POINTER(sgx_config), C.POINTER(_ctx)]),
    ("sgx_destroy", None, [_ctx]),
    ("sgx_last_error", C.c_char_p, [_ctx]),
    ("sgx_query", C.c_int, [_ctx, C.POINTER(sgx_info)]),
    ("sgx_num_frames", _sz, [_ctx, _sz]),
    ("sgx_set_stream", C.c_int, [_ctx, _vp]),
    ("sgx_sync", C.c_int, [_ctx]),
    ("sgx_stft_batch", C.c_int, [_ctx, _vp, _sz, _sz, _sz, _vp, C.POINTER(_sz)]),
    ("sgx_stft_batch_f16", C.c_int, [_ctx, _vp, _sz, _sz, _sz, _vp, C.POINTER(_sz)]),
    ("sgx_process_one", C.c_int, [_ctx, _vp, _sz, _vp]),
    ("sgx_render_batch", C.c_int, [_ctx, _vp, _sz, _sz, _sz, _vp, C.POINTER(_sz)]),
    ("sgx_render_mags", C.c_int, [_ctx, _vp, _sz, _vp]),
    ("sgx_magnitude_in", C.c_int, [_ctx, _vp, _sz, _vp, C.c_uint32, _vp]),
    ("sgx_spectrum_levels", C.c_int, [_ctx, _vp, C.c_uint32, _vp]),
    ("sgx_live_create", C.c_int, [_ctx, _sz, C.c_uint32, C.POINTER(_vp)]),
    ("sgx_live_destroy", None, [_vp]),
    ("sgx_live_push", C.c_longlong, [_vp, _vp, _sz, C.c_uint32]),
    ("sgx_live_occupied", _sz, [_vp]),
    ("sgx_live_tick", C.c_int, [_vp, C.c_int, _vp, _sz, C.POINTER(_sz)]),
    ("sgx_set_gradient", C.c_int, [_ctx, _vp, C.c_uint32, C.c_int]),
    ("sgx_set_gradient_fn", C.c_int, [_ctx, _vp, _vp, C.c_int]),
    ("sgx_set_builtin_gradient", C.c_int, [_ctx, C.c_char_p]),
    ("sgx_builtin_gradient", C.c_int, [C.c_char_p, _vp]),
    ("sgx_lookup_table", C.c_int, [_ctx, C.c_uint32, _vp]),
    ("sgx_bin_edges", C.c_int, [_ctx, _vp]),
    ("sgx_row_sample_counts", C.c_int, [_ctx, _vp]),
    ("sgx_window", C.c_int, [_ctx, _vp]),
    ("sgx_synth_white_noise", C.c_int, [_ctx, _vp, C.c_uint64, _sz, C.c_uint32, C.c_uint32]),
    ("sgx_set_builtin_scheme", C.c_int, [_ctx, C.c_char_p, C.c_int]),
    ("sgx_builtin_gradient_eval", C.c_int, [C.c_char_p, C.c_double, C.POINTER(C.c_uint8)]),
    ("sgx_view_create", C.c_int, [_ctx, C.c_uint32, C.POINTER(C.c_void_p)]),
    ("sgx_view_destroy", None, [C.c_void_p]),
    ("sgx_view_write_rows", C.c_int, [C.c_void_p, _vp, _sz, C.POINTER(C.c_uint32)]),
    ("sgx_view_offset", C.c_uint32, [C.c_void_p]),
    ("sgx_live_tick_view", C.c_int, [C.c_void_p, C.c_void_p, _sz, C.POINTER(_sz)]),
    ("sgx_view_draw", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, _vp]),
    ("sgx_image_create", C.c_int, [_ctx, C.c_uint32, C.POINTER(C.c_void_p)]),
    ("sgx_image_destroy", None, [C.c_void_p]),
    ("sgx_image_write_columns", C.c_int, [C.c_void_p, _vp, _sz, C.POINTER(C.c_uint32)]),
    ("sgx_image_offset", C.c_uint32, [C.c_void_p]),
    ("sgx_image_width", C.c_uint32, [C.c_void_p]),
    ("sgx_image_height", C.c_uint32, [C.c_void_p]),
    ("sgx_live_tick_image", C.c_int, [C.c_void_p, C.c_void_p, _sz, C.POINTER(_sz)]),
    ("sgx_image_read", C.c_int, [C.c_void_p, C.c_int, _vp]),
    ("sgx_image_pixels", C.c_void_p, [C.c_void_p]),
    ("sgx_checksum", C.c_int, [_ctx, _vp, _sz, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("sgx_checksum_add", C.c_int, [_ctx, _vp, _sz, C.c_uint64, _vp]),
]

GRADIENT_FN = C.CFUNCTYPE(None, C.c_double, C.POINTER(C.c_uint8), C.c_void_p)

_lib = None


def load():
    """Load libsgx.so.  torch is imported first so that both share one HIP runtime
    (libamdhip64.so.7 is resolved by soname to the copy torch already mapped)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C spectrogram_rs_amd/csrc). "
            "There is no CPU fallback."
        )
    import torch  # noqa: F401  (one HIP runtime per process)

    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, restype, argtypes in SIGNATURES:
        fn = getattr(lib, name)  # AttributeError here = ABI/header mismatch
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib
