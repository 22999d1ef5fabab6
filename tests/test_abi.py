"""CPU-side checks of the C-ABI boundary: the library loads without a GPU and exports exactly the
entry points include/vittrack.h declares.  No compute call is made here."""
import ctypes
import os
import re

import pytest

from conftest import REPO


def _declared_symbols():
    src = open(os.path.join(REPO, "include", "vittrack.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vt_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree():
    from vittracker_amd import native
    assert _declared_symbols() == sorted(native.SYMBOLS)


def test_library_exports_every_declared_symbol():
    from vittracker_amd import native
    if not os.path.exists(native.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    if not os.path.exists(native.LIB_PATH_F16):
        import __graft_entry__ as ge
        ge.build()
    for path, tag in ((native.LIB_PATH, b"f32"), (native.LIB_PATH_F16, b"f16")):
        L = ctypes.CDLL(path)
        for s in _declared_symbols():
            assert hasattr(L, s), f"{s} declared in include/vittrack.h but not exported by {path}"
        L.vt_version.restype = ctypes.c_char_p
        assert b"gfx950" in L.vt_version() and tag in L.vt_version()


def test_bad_config_is_rejected_without_gpu():
    """Argument validation happens before any HIP call, so it is testable on CPU; the message
    names the supported configuration (the reference would fail later with a shape error)."""
    from vittracker_amd import native
    L = native.lib()
    cfg = native.VtConfig(128, 256, 48, 32, 3, 32, 16, 1)   # HEADS=32: impossible at dim 48
    h = ctypes.c_void_p()
    assert L.vt_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert b"divisible by HEADS" in L.vt_last_error() and b"heads=32" in L.vt_last_error()
    cfg = native.VtConfig(100, 200, 48, 1, 3, 32, 16, 1)
    assert L.vt_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert b"unsupported geometry" in L.vt_last_error()
    cfg = native.VtConfig(128, 256, 768, 8, 12, 256, 16, 1)     # ViT-Base with the wrong head count
    assert L.vt_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert b"unsupported ViT-Base" in L.vt_last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from vittracker_amd import native
    monkeypatch.setattr(native, "_lib", None)
    monkeypatch.setattr(native, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(native.VtError, match="no CPU fallback"):
        native.lib()
