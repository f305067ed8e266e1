"""Build libavsiam_hip.so (hand-written gfx950 kernels + C ABI) in-tree with hipcc.

    python -m avsiam_amd.build            # incremental
    python -m avsiam_amd.build --force
    AVSIAM_HIPCC_EXTRA=-DNT8_ABLATE=2 python -m avsiam_amd.build --out avsiam_amd/csrc/ab_x.so     # diagnostic build (tools/ab_lib.sh)

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the resulting .so travels
to the GPU box with the source snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(CSRC, "libavsiam_hip.so")
SOURCES = ["api.cpp", "comm.cpp", "layernorm.hip", "elementwise.hip", "losses.hip", "gemm.hip", "attention.hip", "maskplan.hip", "preprocess.hip"]
# -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs (gfx950's register file is unified), so the softmax VALU work of
# the attention kernels reads them directly instead of through v_accvgpr_read/write copies.
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function",
         "-mllvm", "-amdgpu-mfma-vgpr-form=1"]


# per-source flags.  attention.hip: no SLP vectorisation - packed fp32 arithmetic (v_pk_add_f32 ...) is an anti-lever beside MFMAs (see the file)
FLAGS_FOR = {"attention.hip": ["-fno-slp-vectorize"]}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, out=None):
    """The product library (`out` None): incremental, never with extra flags.  A diagnostic build (`out` given; extra hipcc flags from
    AVSIAM_HIPCC_EXTRA) compiles everything afresh into its own object directory and writes `out` - it never touches
    libavsiam_hip.so, so a timing build cannot be left behind as the product."""
    extra = os.environ.get("AVSIAM_HIPCC_EXTRA", "").split()
    if extra and out is None:
        raise SystemExit("AVSIAM_HIPCC_EXTRA is for diagnostic builds: pass --out <file.so> (tools/ab_lib.sh does)")
    obj_dir, lib = (OBJ, LIB) if out is None else (OBJ + "_diag", os.path.abspath(out))
    if out is not None:
        force = True
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    objs, todo = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(obj_dir, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            todo.append([hipcc] + FLAGS + FLAGS_FOR.get(src, []) + extra + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", s, "-o", o])
    if todo:
        # the translation units are independent: compile them side by side (a fresh build is bounded by gemm.hip alone instead of the sum)
        from concurrent.futures import ThreadPoolExecutor
        for cmd in todo:
            if verbose:
                print(" ".join(cmd), flush=True)
        with ThreadPoolExecutor(max_workers=max(1, min(len(todo), (os.cpu_count() or 2), 8))) as pool:
            list(pool.map(subprocess.check_call, todo))
    if force or _stale(lib, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv, out=sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None)
