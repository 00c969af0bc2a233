"""CPU: the C-ABI library loads and exports every symbol include/*.h declares (no compute calls)."""
import ctypes
import glob
import os
import re

from conftest import ROOT


def declared_symbols():
    syms = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        syms |= set(re.findall(r"\b(st_[a-z0-9_]+)\s*\(", text))
    return syms


def test_library_exports_every_declared_symbol():
    from style_transfer_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build first: python -m style_transfer_amd.build"
    L = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 10
    for s in syms:
        assert hasattr(L, s), "symbol %s declared in include/ but not exported" % s
    assert syms == set(_lib.EXPORTS), "ctypes binding table and header disagree: %s" % (syms ^ set(_lib.EXPORTS))


def test_loader_binds_and_reports_version():
    from style_transfer_amd import _lib
    assert _lib.lib().st_abi_version() == _lib.ABI_VERSION


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from style_transfer_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        _lib.lib()
        assert False, "expected StError"
    except _lib.StError as e:
        assert "no CPU fallback" in str(e)


def test_argument_validation_without_gpu():
    """pure host-side argument checks return ST_EINVAL before any HIP call"""
    from style_transfer_amd import _lib
    L = _lib.lib()
    w = _lib.StDecoderWeights()
    w.d_model = 512
    rc = L.st_decode_greedy(ctypes.byref(w), 0, 4, 60, 2, 3, 0, 0, 1, 0, 0, 0, 0, None, None)
    assert rc == -1 and b"d_model" in L.st_last_error()
    assert L.st_decode_workspace_bytes(None, 4, 60) == 0
