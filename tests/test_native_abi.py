"""CPU-side checks of the C-ABI library: it loads, and exports every symbol that
include/smt_hip.h declares (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    with open(os.path.join(REPO, "include", "smt_hip.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(smt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from smt_amd import native
    if not os.path.exists(native.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    handle = ctypes.CDLL(native.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 7
    for name in names:
        assert hasattr(handle, name), f"{name} declared in include/smt_hip.h but not exported"
    assert sorted(native.exported_symbols()) == names  # the binding covers exactly the header
    assert handle.smt_abi_version() == native.ABI_VERSION


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from smt_amd import native
    monkeypatch.setattr(native, "_lib", None)
    monkeypatch.setattr(native, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(native.NativeLibraryError):
        native.lib()
