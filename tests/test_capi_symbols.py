"""CPU-side checks of the drop-in boundary: the shared library loads and exports exactly
the entry points include/smhip.h declares (no compute calls -- there is no GPU here)."""
import os
import re

import pytest

from staticmapping_amd import _capi, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "smhip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(smhip_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_ctypes_table_agree():
    assert _declared() == sorted(_capi.SIGNATURES)


def test_library_exports_every_declared_symbol():
    build.build()
    lib = _capi.load_library()
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.smhip_version() >= 100
    assert lib.smhip_status_string(0) == b"ok"


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from staticmapping_amd import IcpFastHip, SmhipError
    with pytest.raises(SmhipError):
        IcpFastHip()


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    monkeypatch.setattr(_capi, "_LIB", None)
    monkeypatch.setattr(_capi, "library_path", lambda: str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _capi.load_library()
