"""The C-ABI library loads here (no GPU) and exports every symbol include/sgx.h declares; the
product path fails loudly -- never falls back to a CPU implementation -- when no device exists."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "sgx.h")).read()
    return sorted(set(re.findall(r"SGX_API\s+[\w\s\*]+?\b(sgx_\w+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = declared_symbols()
    for must in ("sgx_create", "sgx_destroy", "sgx_stft_batch", "sgx_render_batch", "sgx_render_mags",
                 "sgx_process_one", "sgx_set_gradient", "sgx_last_error", "sgx_query", "sgx_lookup_table"):
        assert must in names
    assert len(names) >= 20


def test_library_exports_every_declared_symbol():
    from spectrogram_rs_amd import _lib
    lib = _lib.load()
    bound = {name for name, _, _ in _lib.SIGNATURES}
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in sgx.h but not exported by libsgx.so"
        assert name in bound, f"{name} has no ctypes signature in _lib.SIGNATURES"
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (sgx_\w+)", out))
    assert exported == set(declared_symbols()), exported ^ set(declared_symbols())


def test_struct_layouts_match_the_header():
    from spectrogram_rs_amd._lib import load, sgx_config, sgx_info
    lib = load()
    cfg = sgx_config()
    assert lib.sgx_config_init(C.byref(cfg)) == 0
    assert cfg.struct_size == C.sizeof(sgx_config)  # the C side's sizeof agrees with ctypes
    assert (cfg.window_samples, cfg.hop_samples, cfg.channels, cfg.rows) == (2048, 256, 1, 1024)
    assert (cfg.f_min, cfg.f_max, cfg.min_db, cfg.max_db) == (32.0, 22030.0, -70.0, -10.0)
    assert C.sizeof(sgx_info) == 64
    assert lib.sgx_version().startswith(b"sgx")


def test_python_mirror_constants_match_the_header():
    # every SGX_FLAG_* / SGX_INTERP_* / SGX_LUT_* / SGX_LIVE_* of include/sgx.h has the same value in the ctypes mirror, flags are
    # distinct single bits
    from spectrogram_rs_amd import _lib
    text = open(os.path.join(ROOT, "include", "sgx.h")).read()
    defs = {name: int(val) for name, val in re.findall(r"#define\s+SGX_((?:FLAG|INTERP|LUT|LIVE)_\w+)\s+(\d+)u?\b", text)}
    flags = {k: v for k, v in defs.items() if k.startswith("FLAG_")}
    assert len(flags) >= 6 and len(set(flags.values())) == len(flags)
    for name, v in flags.items():
        assert v & (v - 1) == 0, f"SGX_{name} = {v} is not a single bit"
    for name, v in defs.items():
        if hasattr(_lib, name):
            assert getattr(_lib, name) == v, f"_lib.{name} = {getattr(_lib, name)}, sgx.h says {v}"
    for must in ("FLAG_FORCE_GENERIC", "FLAG_NO_FUSED_RENDER", "FLAG_PAIRED_FRAMES", "FLAG_COMPLEX_MONO", "FLAG_MIXED_GENERIC",
                 "INTERP_CUBIC", "INTERP_COSINE", "LUT_FLOOR_N"):
        assert hasattr(_lib, must) and must in defs, must


def test_builtin_gradients_match_fixture(gradients):
    import numpy as np
    from spectrogram_rs_amd import builtin_gradient
    for name, table in gradients.items():
        assert np.array_equal(builtin_gradient(name), table)
    with pytest.raises(KeyError):
        builtin_gradient("nope")


def test_no_device_is_a_loud_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from spectrogram_rs_amd import SgxError, SpectrogramEngine
    with pytest.raises(SgxError) as ei:
        SpectrogramEngine()
    assert ei.value.code == -5 and "no CPU fallback" in str(ei.value)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "spectrogram_rs_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h", ".inc")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "spectro_oracle" not in text and "liboracle" not in text, f
                assert not re.search(r"^\s*(import|from)\s+oracle\b", text, re.M), f
