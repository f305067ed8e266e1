"""CPU-side checks of the C-ABI boundary: the library builds, loads, and exports every symbol that
include/avsiam_hip.h declares (no compute calls - there is no GPU here)."""
import ctypes

import pytest

from avsiam_amd import _lib


@pytest.fixture(scope="module")
def lib_path():
    from avsiam_amd.build import build
    return build(verbose=False)


def test_header_symbols_exported(lib_path):
    protos = _lib.parse_header()
    assert len(protos) >= 29
    lib = ctypes.CDLL(lib_path)
    for name in protos:
        assert hasattr(lib, name), f"{name} declared in include/avsiam_hip.h but not exported"


def test_no_undeclared_exports(lib_path):
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib_path], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if " T avs_" in l}
    declared = set(_lib.parse_header()) | {"avs_set_error"}
    assert exported <= declared, exported - declared


def test_abi_version_and_error_string(lib_path):
    lib = _lib.load()
    assert lib.avs_abi_version() == 1
    assert isinstance(lib.avs_last_error(), bytes)
    # argument validation happens before any launch, so it is testable without a GPU
    rc = lib.avs_gemm_nt_bf16(None, 0, None, 0, 10, 100, 64, None, None, 0, None, None, 0, None, 0, 0, None, 0, 1.0, 0, 0, 1.0, None, None)
    assert rc == -2 and b"gemm_nt" in lib.avs_last_error()
    rc = lib.avs_layernorm_fwd(None, None, None, None, None, None, None, None, 0, None, None, 4, 100, 1e-5, None)
    assert rc == -2


def test_missing_library_raises(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libavsiam_hip.so")
    with pytest.raises(_lib.AvsiamHipError):
        _lib.load()
