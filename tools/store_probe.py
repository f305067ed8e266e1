#!/usr/bin/env python3
"""How fast persistent workgroups write (and read-modify-write) 256 x 256 output tiles, by store shape and by how many CUs store at once
(tools/csrc/store_probe.hip).   python tools/store_probe.py"""
import ctypes
import os
import subprocess

import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    src, so = os.path.join(HERE, "csrc", "store_probe.hip"), os.path.join(HERE, "csrc", "store_probe.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", src, "-o", so])
    lib = ctypes.CDLL(so)
    lib.store_probe.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_void_p]
    M, N = 95744, 768
    outf = torch.zeros(M, N, device="cuda")
    outb = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    res = torch.randn(M, N, device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    sink = torch.zeros(256 * 512, device="cuda")

    def run(mode, f32, with_res, active, depth, loads_only=False):
        out = outf if f32 else outb
        def go():
            rc = lib.store_probe(mode, 1 if f32 else 0, out.data_ptr(), res.data_ptr() if with_res else None, M, N, 256, active, depth, sink.data_ptr() if loads_only else None, st)
            assert rc == 0, rc
        go(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            go()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 5 * 1e-3
        nbytes = M * N * ((0 if loads_only else 4 if f32 else 2) + (4 if with_res else 0))
        return t, nbytes / t / 1e9

    for depth in (1, 2, 4):
        msg = f"{'fp32 residual reads only':34s} {'64-column slices (GEMM epilogue)':33s} depth {depth}:"
        for active in (256, 128, 64, 32):
            t, gbs = run(0, True, True, active, depth, loads_only=True)
            msg += f"  {active:3d} CUs {gbs:6.0f} GB/s ({gbs / active:5.1f}/CU)"
        print(msg, flush=True)
    for f32, with_res, label in ((False, False, "bf16 stores"), (True, False, "fp32 stores"), (True, True, "fp32 stores + fp32 residual reads")):
        for mode, mname in ((0, "64-column slices (GEMM epilogue)"), (1, "full tile rows")):
            for depth in ((1, 2, 4) if with_res else (1,)):
                msg = f"{label:34s} {mname:33s} depth {depth}:"
                for active in (256, 128, 64, 32):
                    t, gbs = run(mode, f32, with_res, active, depth)
                    msg += f"  {active:3d} CUs {gbs:6.0f} GB/s ({gbs / active:5.1f}/CU)"
                print(msg, flush=True)


if __name__ == "__main__":
    main()
