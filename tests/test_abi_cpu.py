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
    assert lib.avs_abi_version() == 2          # round 6: stricter attention strides, gelu' codes, ln_dma 0 | 1 (api.cpp)
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
        with pytest.raises(_lib.AvsiamHipError):
            _lib.tuning_set("ln_dma", 2)               # (the value 2 never had a meaning)
        for k in ("gemm_tile", "gemm_persistent", "gemm_nt8", "nt_tile_h", "nt_grid", "ln_dma", "ln_rpw", "attn_ring", "nt_big_min"):
            _lib.tuning_get(k)
    finally:
        _lib.tuning_set("cu_reserve", 0)
    import subprocess
    undefined = subprocess.check_output(["nm", "-D", "--undefined-only", lib_path], text=True)
    assert "getenv" not in undefined, "the library must not read the environment"


def test_a_rejected_environment_knob_fails_every_load_the_same_way(monkeypatch):
    """load() used to publish the library BEFORE applying the AVSIAM_* knobs, so a rejected value raised once and every later load()
    returned a half-configured library silently.  Now the library is published only after every knob is applied; an empty value means unset."""
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setenv("AVSIAM_GEMM_TILE", "64")
    for _ in range(2):
        with pytest.raises(_lib.AvsiamHipError, match="AVSIAM_GEMM_TILE"):
            _lib.load()
        assert _lib._lib is None
    monkeypatch.setenv("AVSIAM_GEMM_TILE", "abc")
    with pytest.raises(_lib.AvsiamHipError, match="not an integer"):
        _lib.load()
    monkeypatch.setenv("AVSIAM_GEMM_TILE", "")
    monkeypatch.setenv("AVSIAM_CU_RESERVE", "")
    assert _lib.load() is not None and _lib.env_value("AVSIAM_CU_RESERVE") is None and _lib.tuning_get("gemm_tile") == 0


def test_missing_library_raises(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libavsiam_hip.so")
    with pytest.raises(_lib.AvsiamHipError):
        _lib.load()


# Weight-gradient GEMM time (us, MI355X) against the split count of the token rows: profiles/r05/tn_splits.log (tools/bench_tn_splits.py)
TN_SPLIT_TIMES = {
    (2832, 3072, 768): {1: 37.7, 2: 35.6, 3: 35.2, 4: 45.1, 5: 47.6, 7: 57.0},
    (2832, 768, 768): {1: 31.4, 2: 18.4, 3: 16.1, 4: 15.0, 5: 16.5, 7: 17.3, 9: 21.7, 14: 25.2, 28: 44.5},
    (11328, 3072, 768): {1: 110.1, 2: 90.9, 3: 69.5, 4: 92.3, 5: 83.1, 7: 86.1},
    (11328, 768, 768): {1: 110.9, 2: 55.6, 3: 41.3, 4: 34.7, 5: 31.6, 7: 29.5, 9: 32.1, 14: 37.2, 28: 55.9},
    (27419, 3072, 768): {1: 560.0, 2: 295.7, 3: 210.4, 4: 172.3, 5: 152.8, 7: 143.9},
    (27419, 2304, 768): {1: 558.0, 2: 290.5, 3: 203.2, 4: 161.6, 5: 140.9, 7: 123.5, 9: 122.3},
    (27419, 768, 768): {1: 262.8, 2: 126.7, 3: 89.4, 4: 70.4, 5: 61.7, 7: 50.7, 9: 55.0, 14: 53.7, 28: 72.2},
    (95630, 3072, 768): {1: 1934.7, 2: 986.1, 3: 664.5, 4: 528.7, 5: 460.6, 7: 404.6},
    (95630, 2304, 768): {1: 1931.7, 2: 980.2, 3: 657.9, 4: 502.5, 5: 419.3, 7: 344.9, 9: 321.4},
    (95630, 768, 768): {1: 1158.9, 2: 564.3, 3: 378.5, 4: 284.8, 5: 235.6, 7: 175.8, 9: 159.0, 14: 130.6, 28: 151.2},
}


def test_weight_gradient_split_plan_stays_near_the_measured_optimum(lib_path):
    """avs_gemm_tn_plan (host arithmetic, no GPU): the split count the library picks for a weight-gradient launch - a stage / atomics cost model
    fitted in round 5, rounds 1 - 4 always filled one resident round - against the measured time-vs-splits table of the same shapes: within 20 %
    of the best measured split everywhere (the old rule was 45 - 200 % off below ~30 000 rows), never more than one resident round, and fewer
    splits for fewer rows."""
    lib = _lib.load()

    def plan(M, N1, N2):
        tile, splits = ctypes.c_int(0), ctypes.c_int(0)
        assert lib.avs_gemm_tn_plan(M, N1, N2, ctypes.byref(tile), ctypes.byref(splits)) == 0
        return tile.value, splits.value

    slots = lib.avs_persistent_cu_slots()
    for (M, N1, N2), times in TN_SPLIT_TIMES.items():
        tile, s = plan(M, N1, N2)
        assert tile in (128, 256) and s >= 1
        assert (N1 // tile) * (N2 // tile) * s <= slots * (1 if tile == 256 else 2), (M, N1, N2, tile, s)
        # the measured table is indexed by the split count of a few sampled values: take the nearest sampled neighbours of the plan
        lo = max(k for k in times if k <= s)
        hi = min((k for k in times if k >= s), default=lo)
        t = max(times[lo], times[hi])
        assert t <= 1.2 * min(times.values()), ((M, N1, N2), s, t, min(times.values()))
    for N1, N2 in ((3072, 768), (768, 768), (2048, 512)):
        seq = [plan(M, N1, N2) for M in (708, 2832, 11328, 27419, 95630)]
        wgs = [(N1 // t) * (N2 // t) * s * (4 if t == 256 else 1) for t, s in seq]            # output area x splits, in 128^2 units
        assert all(a <= b for a, b in zip(wgs, wgs[1:])), seq                                  # more rows never mean fewer partial tiles
    assert plan(708, 768, 768)[1] <= 2 and plan(64, 128, 128) == (128, 1)
    assert lib.avs_gemm_tn_plan(100, 100, 128, None, None) == -2


def test_forward_gemm_dispatch_plan(lib_path):
    """avs_gemm_nt_plan (host arithmetic, no GPU): which kernel family a forward / input-gradient GEMM goes to.  The thresholds follow the
    measurements of round 5 (profiles/r05/nt_midsize.log, small_gemm_ab.log): the persistent 256 x 256 kernel from half the CU slots' worth
    of tiles (135 tiles: 76.7 -> 62.6 us; 96 tiles stay on 128 x 128: 44.6 against 57.2 us), the LDS-DMA ring kernel when the 128 x 128 tiling
    gives at most one workgroup per CU, half-height tiles when even that fills less than half the CUs; cu_reserve shifts every threshold;
    the knobs gemm_tile / nt_big_min / gemm_ring override."""
    lib = _lib.load()
    TWOBUF, RING, RING_HALF, TWOBUF_HALF, PERSISTENT = range(5)

    def plan(M, N, K):
        fam, wgs = ctypes.c_int(-1), ctypes.c_int(-1)
        assert lib.avs_gemm_nt_plan(M, N, K, ctypes.byref(fam), ctypes.byref(wgs)) == 0
        return fam.value, wgs.value

    slots = lib.avs_persistent_cu_slots()
    assert slots == 256 or slots > 8                                   # 256 without a device (the default) and on MI355X
    if slots == 256:
        assert plan(95630, 768, 768) == (PERSISTENT, 256)              # the headline shapes: 1122 tiles walked by one workgroup per CU
        assert plan(11328, 768, 3072) == (PERSISTENT, 135)             # one frame at batch 64: 135 tiles >= 128 (round 5; 224 before)
        assert plan(8192, 768, 3072) == (TWOBUF, 384)                  # 96 tiles: 128 x 128 tiles, one round of two per CU
        assert plan(2832, 768, 768) == (RING, 138)                     # batch 4, decoder rows: at most one 128 x 128 workgroup per CU
        assert plan(708, 768, 768) == (RING_HALF, 72)                  # batch 4, encoder rows: 36 full tiles -> 72 half-height ones
        assert plan(708, 768, 128) == (TWOBUF_HALF, 72)                # two K-steps only: nothing for a ring to keep in flight
        assert plan(1979, 2304, 768) == (TWOBUF, 288)                  # more than one workgroup per CU: the ring kernel does not apply
    try:
        _lib.tuning_set("cu_reserve", 64)                              # 192 slots: the persistent threshold moves to 96 tiles
        assert plan(8192, 768, 3072) == (PERSISTENT, 96)
        _lib.tuning_set("cu_reserve", 0)
        _lib.tuning_set("nt_big_min", 224)                             # the threshold of rounds 1 - 4
        assert plan(11328, 768, 3072)[0] != PERSISTENT
        _lib.tuning_set("nt_big_min", 0)
        _lib.tuning_set("gemm_ring", 0)
        assert plan(708, 768, 768) == (TWOBUF, 36) and plan(2832, 768, 768) == (TWOBUF, 138)
        _lib.tuning_set("gemm_ring", 1)
        assert plan(708, 768, 768) == (RING, 36)
        _lib.tuning_set("gemm_ring", 2)
        _lib.tuning_set("gemm_tile", 128)
        assert plan(95630, 768, 768)[0] == TWOBUF
        _lib.tuning_set("gemm_tile", 256)
        assert plan(708, 768, 768)[0] == PERSISTENT and plan(708, 384, 768)[0] != PERSISTENT      # N must be a multiple of 256 for that tile
    finally:
        for k, v in (("cu_reserve", 0), ("nt_big_min", 0), ("gemm_ring", 2), ("gemm_tile", 0)):
            _lib.tuning_set(k, v)
    assert lib.avs_gemm_nt_plan(100, 100, 64, None, None) == -2
