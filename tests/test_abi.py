"""CPU-side checks of the drop-in boundary: the C-ABI library loads (no GPU needed for dlopen), exports every symbol
that include/vtp_hip.h declares, and the ctypes signatures in vtp_amd/_lib.py agree with the header's prototypes."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_protos():
    src = open(os.path.join(ROOT, "include", "vtp_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|const char\*)\s+(vtp_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = [a.strip() for a in m.group(3).replace("\n", " ").split(",")]
        if args == ["void"]:
            args = []
        protos[m.group(2)] = args
    return protos


def _ctype_of(arg: str):
    if "*" in arg:
        return ctypes.c_void_p
    t = arg.split()
    if t[0] == "long":
        return ctypes.c_long
    if t[0] == "float":
        return ctypes.c_float
    if t[0] == "int":
        return ctypes.c_int
    raise AssertionError(f"unhandled C type in header: {arg}")


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from vtp_amd import _lib
    return _lib.load()


def test_header_and_ctypes_signatures_agree(lib):
    from vtp_amd import _lib
    protos = _header_protos()
    assert set(protos) - {"vtp_abi_version", "vtp_last_error"} == set(_lib.SIGNATURES), \
        "include/vtp_hip.h and vtp_amd/_lib.py declare different entry points"
    for name, args in protos.items():
        assert hasattr(lib, name), f"libvtp_hip.so does not export {name}"
        if name in _lib.SIGNATURES:
            want = [_ctype_of(a) for a in args]
            assert want == _lib.SIGNATURES[name], f"{name}: header {args} vs ctypes {_lib.SIGNATURES[name]}"


def test_version_and_error_string(lib):
    assert lib.vtp_abi_version() == 1
    assert isinstance(lib.vtp_last_error(), bytes)


def test_argument_validation_needs_no_gpu(lib):
    # rejected before any HIP call is made -> safe on a CPU-only box
    rc = lib.vtp_gemm_nt(None, 8, None, 8, None, 8, None, 0, None, None, None, 4, 4, 8, 0, 0, 0, 0, 0, 1, 1.0, None)
    assert rc == -1 and b"null" in lib.vtp_last_error()
    rc = lib.vtp_rope_qk(ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), 1, 4, 1, 9, 0, None)
    assert rc == -1 and b"prefix" in lib.vtp_last_error()
    rc = lib.vtp_adamw(ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), None, 6, 1e-3, 0.9,
                       0.99, 1e-8, 0.0, 1, 1.0, None)
    assert rc == -1


def test_missing_library_is_a_loud_error(monkeypatch, tmp_path):
    from vtp_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no fallback"):
        _lib.load()


def test_diagnostics_hooks_are_refused_without_vtp_diag(monkeypatch):
    """vtp_gemm_debug feeds production kernels process-wide -- anything but 'all off' needs VTP_DIAG=1 (host-only check)"""
    from vtp_amd import _lib
    lib = _lib.load()
    monkeypatch.delenv("VTP_DIAG", raising=False)
    assert lib.vtp_gemm_debug(None, 0, 0) == 0
    assert lib.vtp_gemm_debug(None, 0, 100) != 0 and b"VTP_DIAG" in lib.vtp_last_error()
    assert lib.vtp_gemm_debug(None, 64, 0) != 0
    monkeypatch.setenv("VTP_DIAG", "1")
    assert lib.vtp_gemm_debug(None, 0, -5) != 0  # the 'no stores' mode is gone
    assert lib.vtp_gemm_debug(None, 64, 0) == 0 and lib.vtp_gemm_debug(None, 0, 0) == 0
