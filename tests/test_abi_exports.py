"""CPU-side checks of the drop-in boundary: the built library exports every symbol that
include/phe_hip.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re
import sys

import pytest

from conftest import PKG, ROOT

if PKG not in sys.path:
    sys.path.insert(0, PKG)


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "phe_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(phe_hip_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    import __graft_entry__ as g
    lib = os.path.join(PKG, "lib", "libphe_hip.so")
    if not os.path.exists(lib):
        g.build_hip()
    L = ctypes.CDLL(lib)
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), "missing export: " + s


def test_binding_declares_every_symbol():
    from phe import _native
    assert sorted(_native.EXPORTED_SYMBOLS) == declared_symbols()


def test_no_cpu_fallback_without_library(monkeypatch, tmp_path):
    from phe import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        _native.lib()


def test_errors_without_gpu_are_loud():
    # on the CPU-only build box creating a context must fail with a HIP error, never silently compute
    from phe import _native
    import ctypes as C
    n = C.c_int(0)
    rc = _native.lib().phe_hip_device_count(C.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises((RuntimeError, ValueError)):
        _native.Context(126869)


def test_abi_version_is_checked_before_the_first_call(monkeypatch):
    """argument lists changed under unchanged symbol names; a mirror built against another revision of the header
    would link and pass shifted arguments.  The header's PHE_HIP_ABI_VERSION, the library's phe_hip_abi_version() and the
    binding's ABI_VERSION must be one number, and a binding that mirrors another version must refuse to load the library."""
    from phe import _native
    header = open(os.path.join(ROOT, "include", "phe_hip.h")).read()
    declared = int(re.search(r"#define\s+PHE_HIP_ABI_VERSION\s+(\d+)", header).group(1))
    assert declared == _native.ABI_VERSION == _native.lib().phe_hip_abi_version()
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "ABI_VERSION", declared + 1)
    with pytest.raises(ImportError, match="ABI version"):
        _native.lib()
    monkeypatch.setattr(_native, "ABI_VERSION", declared)
    assert _native.lib().phe_hip_abi_version() == declared
