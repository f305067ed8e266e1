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


def test_attention_refuses_rows_that_are_not_16_byte_aligned(lib_path):
    """the attention epilogues store 16 bytes per lane - an output stride that is a multiple of 4 but not of 8 elements must be
    an argument error (-2), not a misaligned store.  Argument validation precedes every launch, so this runs without a GPU."""
    lib = _lib.load()
    one = ctypes.c_void_p(16)                          # any non-NULL value: the call must fail before touching it
    D, H = 768, 12
    for ldo, ok_shape in ((772, False), (776, True)):
        rc = lib.avs_attn_fwd(one, 3 * D, D, H, one, one, one, 1, 128, one, ldo, one, 128, None) if not ok_shape else 0
        assert (rc == -2 and b"ldo" in lib.avs_last_error()) or ok_shape
    rc = lib.avs_attn_bwd(one, 3 * D, D, H, one, one, one, 1, 128, one, one, 772, one, one, 128, one, None)
    assert rc == -2 and b"ldo" in lib.avs_last_error()
    rc = lib.avs_attn_bwd_fused(one, 3 * D, D, H, one, one, 1, 128, one, one, 772, one, 128, one, None)
    assert rc == -2


def test_tuning_knobs_are_set_through_the_abi_only(lib_path):
    """The library never reads the environment (include/avsiam_hip.h): the knobs are plain integers behind avs_tuning_set / _get, range
    checked; cu_reserve shrinks what a persistent grid may fill."""
    lib = _lib.load()
    full = lib.avs_persistent_cu_slots()
    try:
        _lib.tuning_set("cu_reserve", 8)
        assert _lib.tuning_get("cu_reserve") == 8 and lib.avs_persistent_cu_slots() == full - 8
        with pytest.raises(_lib.AvsiamHipError):
            _lib.tuning_set("nt_tile_h", 100)
        with pytest.raises(_lib.AvsiamHipError):
            _lib.tuning_set("no_such_knob", 1)
        for k in ("gemm_tile", "gemm_persistent", "gemm_nt8", "nt_tile_h", "nt_grid", "ln_dma", "ln_rpw", "attn_ring", "nt_big_min"):
            _lib.tuning_get(k)
    finally:
        _lib.tuning_set("cu_reserve", 0)
    import subprocess
    undefined = subprocess.check_output(["nm", "-D", "--undefined-only", lib_path], text=True)
    assert "getenv" not in undefined, "the library must not read the environment"


def test_missing_library_raises(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libavsiam_hip.so")
    with pytest.raises(_lib.AvsiamHipError):
        _lib.load()
