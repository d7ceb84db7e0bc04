"""C-ABI library: loads without a GPU, exports every symbol include/rmx.h declares, and reports
errors through return codes (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def _build():
    import __graft_entry__ as g
    g.build()


@pytest.fixture(scope="module")
def lib():
    _build()
    from radio_mapper_amd import xcorr
    return xcorr.load_library()


def _declared():
    text = open(os.path.join(ROOT, "include", "rmx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rmx_[a-z_]+)\s*\(", text)))


def test_header_symbols_all_exported(lib):
    from radio_mapper_amd import xcorr
    names = _declared()
    assert sorted(names) == sorted(xcorr.EXPORTS)
    for n in names:
        assert getattr(lib, n) is not None


def test_version_and_device_count(lib):
    assert lib.rmx_version() == 1
    assert lib.rmx_device_count() >= 0


def test_create_fails_cleanly_on_bad_arguments(lib):
    ctx = C.c_void_p()
    assert lib.rmx_create(C.byref(ctx), 0, 1, 4096, 4, 0) == -1       # n_buoys < 2
    assert b"n_buoys" in lib.rmx_last_error(None)
    assert lib.rmx_create(C.byref(ctx), 0, 8, 4095, 4, 0) == -1       # not a power of two
    assert lib.rmx_create(C.byref(ctx), 0, 8, 4096, 0, 0) == -1       # max_windows < 1
    assert lib.rmx_create(None, 0, 8, 4096, 4, 0) == -1
    assert not ctx.value


def test_no_gpu_is_an_error_not_a_fallback(lib):
    """The product path must fail loudly without a device (no CPU fallback)."""
    from radio_mapper_amd import xcorr
    if lib.rmx_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(xcorr.RmxError) as e:
        xcorr.XcorrEngine(8, 4096, 4)
    assert e.value.code == -2


def test_missing_library_raises(monkeypatch):
    from radio_mapper_amd import xcorr
    monkeypatch.setattr(xcorr, "_lib", None)
    monkeypatch.setenv("RMX_LIBRARY", "/nonexistent/librmx_hip.so")
    with pytest.raises(ImportError):
        xcorr.load_library()
    monkeypatch.delenv("RMX_LIBRARY")
    monkeypatch.setattr(xcorr, "_lib", None)
    xcorr.load_library()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "radio-mapper_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_default_options_are_validated_and_never_come_from_the_environment(lib, monkeypatch):
    """rmx_set_default_option: unknown keys and out-of-range values are refused with text; the library reads no
    RMX_* environment variable (the strings do not even occur in it)."""
    from radio_mapper_amd import xcorr
    assert lib.rmx_set_default_option(b"wscr", 2) == 0
    assert lib.rmx_set_default_option(b"wscr", 9) == -1 and b"not in 0..2" in lib.rmx_last_error(None)
    assert lib.rmx_set_default_option(b"nope", 1) == -1 and b"unknown option" in lib.rmx_last_error(None)
    lib.rmx_clear_default_options()
    with pytest.raises(xcorr.RmxError):
        xcorr.set_default_option("stag", 17)
    xcorr.set_default_option("stag", 0)
    env = {"RMX_WSCR": "2", "RMX_LIBRARY": "x", "RMX_NOPE": "1", "RMX_FUSED": "zz", "HOME": "/"}
    assert xcorr.apply_env_options(env, strict=False) == {"wscr": 2}
    with pytest.raises(ValueError, match="RMX_NOPE"):            # a refused knob does not pass silently
        xcorr.apply_env_options(env)
    import io
    buf = io.StringIO()
    assert xcorr.apply_env_options({"RMX_WSCR": "1", "RMX_BENCH_SAME_DEVICE": "1"}, report=buf) == {"wscr": 1}
    assert "applied {'wscr': 1}" in buf.getvalue() and "REFUSED" not in buf.getvalue()
    xcorr.clear_default_options()
    blob = open(xcorr.library_path(), "rb").read()
    assert b"getenv" not in blob or all(k not in blob for k in (b"RMX_WSCR", b"RMX_FUSED", b"RMX_STAG", b"RMX_NCUS", b"RMX_COL_LOGT"))
    assert all(k not in blob for k in (b"RMX_WSCR", b"RMX_FUSED", b"RMX_STAG", b"RMX_NCUS", b"RMX_COL_LOGT", b"RMX_ROWS_TPR"))
