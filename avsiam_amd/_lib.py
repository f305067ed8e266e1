"""ctypes loader of libavsiam_hip.so - the only door from Python into the HIP kernels.

Signatures are generated from include/avsiam_hip.h so the header is the single source of truth.  There is no
fallback: if the library is missing (or a call fails) this raises - the product path never runs on anything
but the hand-written gfx950 kernels.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "avsiam_hip.h")
# AVSIAM_HIP_LIB: load another build of the same ABI (A/B kernel measurements on one box; see tools/ab_lib.sh)
LIB_PATH = os.environ.get("AVSIAM_HIP_LIB") or os.path.join(_HERE, "csrc", "libavsiam_hip.so")

_lib = None
_protos = None


class AvsiamHipError(RuntimeError):
    pass


def parse_header(path=HEADER):
    """-> {name: (restype, [argtype, ...])} as strings 'ptr' | 'int' | 'll' | 'float' | 'str'."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|long\s+long|int)\s+(avs_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        kinds = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a or a.startswith("avs_stream_t"):
                    kinds.append("ptr")
                elif a.startswith("unsigned long long"):
                    kinds.append("ull")
                elif a.startswith("long long"):
                    kinds.append("ll")
                elif a.startswith("float"):
                    kinds.append("float")
                elif a.startswith("int"):
                    kinds.append("int")
                else:
                    raise ValueError(f"unparsed argument '{a}' of {name}")
        protos[name] = ("str" if "char" in ret else "ll" if "long" in ret else "int", kinds)
    return protos


_CT = {"ptr": ctypes.c_void_p, "int": ctypes.c_int, "ll": ctypes.c_longlong, "ull": ctypes.c_ulonglong, "float": ctypes.c_float}


def load():
    global _lib, _protos
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AvsiamHipError(f"{LIB_PATH} not found - build it with `python -m avsiam_amd.build` "
                             "(there is no CPU or PyTorch fallback for the hot path)")
    lib = ctypes.CDLL(LIB_PATH)
    _protos = parse_header()
    for name, (ret, kinds) in _protos.items():
        fn = getattr(lib, name)
        fn.restype = ctypes.c_char_p if ret == "str" else ctypes.c_longlong if ret == "ll" else ctypes.c_int
        fn.argtypes = [_CT[k] for k in kinds]
    # the library never reads the environment (include/avsiam_hip.h): the AVSIAM_* tuning variables are applied HERE, once - and BEFORE the
    # library is published, so that a rejected value fails every load() the same way instead of leaving a half-configured library behind
    for env, knob in _ENV_KNOBS.items():
        v = env_value(env)
        if v is not None:
            try:
                value = int(v)
            except ValueError:
                raise AvsiamHipError(f"{env}={v!r} is not an integer") from None
            rc = lib.avs_tuning_set(knob.encode(), value)
            if rc != 0:
                raise AvsiamHipError(f"{env}={v}: avs_tuning_set({knob}, {value}) failed ({rc}): {lib.avs_last_error().decode()}")
    _lib = lib
    return lib


def env_value(name):
    """an AVSIAM_* variable, or None when it is unset OR empty (one emptiness rule for the loader and for set_distributed)"""
    v = os.environ.get(name)
    return v if v not in (None, "") else None


_ENV_KNOBS = {"AVSIAM_GEMM_TILE": "gemm_tile", "AVSIAM_GEMM_NT8": "gemm_nt8", "AVSIAM_NT_TILE_H": "nt_tile_h", "AVSIAM_NT_GRID": "nt_grid",
              "AVSIAM_CU_RESERVE": "cu_reserve", "AVSIAM_LN_DMA": "ln_dma", "AVSIAM_LN_RPW": "ln_rpw", "AVSIAM_ATTN_RING": "attn_ring", "AVSIAM_GEMM_RING": "gemm_ring",
              "AVSIAM_NT_BIG_MIN": "nt_big_min", "AVSIAM_DET": "det"}


def tuning_set(name, value):
    lib = load()
    rc = lib.avs_tuning_set(name.encode(), int(value))
    if rc != 0:
        raise AvsiamHipError(f"avs_tuning_set({name}, {value}) failed ({rc}): {lib.avs_last_error().decode()}")


def tuning_get(name):
    lib = load()
    out = ctypes.c_int(0)
    rc = lib.avs_tuning_get(name.encode(), ctypes.byref(out))
    if rc != 0:
        raise AvsiamHipError(f"avs_tuning_get({name}) failed ({rc}): {lib.avs_last_error().decode()}")
    return out.value


calls = 0          # entry-point calls so far (graph_step counts the launches a captured step holds)


def call(name, *args):
    """Invoke an entry point; torch tensors are passed as their data_ptr(), None as NULL."""
    global calls
    calls += 1
    lib = load()
    fn = getattr(lib, name)
    conv = []
    for a in args:
        if a is None:
            conv.append(None)
        elif hasattr(a, "data_ptr"):
            conv.append(a.data_ptr())
        else:
            conv.append(a)
    rc = fn(*conv)
    if rc != 0:
        raise AvsiamHipError(f"{name} failed ({rc}): {lib.avs_last_error().decode()}")
    return rc


_raw_stream = None


def current_stream():
    """The HIP stream torch would launch on now (stream contexts and graph capture included), as an integer handle.  Through
    torch._C._cuda_getCurrentRawStream where it exists: torch.cuda.current_stream() builds a Stream object and resolves the device three
    times - 8.6 us per launch, a quarter of the host time of the reference's batch-4 step (round 5)."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        fast = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        probe = None
        if fast is not None and torch.cuda.is_available():
            try:
                probe = fast(torch.cuda.current_device()) == torch.cuda.current_stream().cuda_stream
            except Exception:
                probe = False
        _raw_stream = fast if probe else False
    if _raw_stream:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream
