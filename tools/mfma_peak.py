#!/usr/bin/env python3
"""Sustained bf16 MFMA rate of the whole chip with register operands only (no LDS, no memory), for random and for all-zero
operands, with the shader clock (probe kernel of tools/clock_probe.py) and rocm-smi's socket power beside it:
the ceiling the power cap leaves to ANY bf16 GEMM on this board.     python tools/mfma_peak.py"""
import ctypes
import os
import subprocess
import sys
import threading
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from clock_probe import build as build_probe, smi_sampler  # noqa: E402


def build():
    src, so = os.path.join(HERE, "csrc", "mfma_peak.hip"), os.path.join(HERE, "csrc", "mfma_peak.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", src, "-o", so])
    lib = ctypes.CDLL(so)
    lib.mfma_peak.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    return lib


def main():
    lib, probe = build(), build_probe()
    dev = "cuda"
    blocks, iters = 256, 20000                                  # one 8-wave workgroup per CU (2 waves per SIMD)
    flops = blocks * 8 * iters * 16 * 2.0 * 16 * 16 * 32
    out = torch.zeros(blocks * 512, device=dev)
    clk = torch.zeros(2, dtype=torch.int64, device=dev)
    side = torch.cuda.Stream()
    rnd, zero = torch.randn(1 << 23, device=dev).to(torch.bfloat16), torch.zeros(1 << 23, device=dev, dtype=torch.bfloat16)
    for name, ops, mode in (("registers, random N(0,1)", rnd, 0), ("registers, all-zero", zero, 0), ("+6 ds_read_b128 /16 MFMA", rnd, 1),
                            ("+6 ds_read +2 LDS-DMA", rnd, 2), ("+12 ds_read_b128 /16 MFMA", rnd, 3)):
        st = torch.cuda.current_stream().cuda_stream
        run = lambda: lib.mfma_peak(ops.data_ptr(), out.data_ptr(), blocks, iters, mode, st)  # noqa: E731
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        stop, samples = threading.Event(), []
        th = threading.Thread(target=smi_sampler, args=(stop, samples))
        th.start()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0, n, probed = time.time(), 0, False
        e0.record()
        while time.time() - t0 < 4.0:
            for _ in range(10):
                run()
                n += 1
            if not probed and time.time() - t0 > 1.5:
                with torch.cuda.stream(side):
                    probe.clock_probe(clk.data_ptr(), int(40 * 1e5), 1, side.cuda_stream)
                probed = True
            torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()
        stop.set(); th.join()
        dt = e0.elapsed_time(e1) * 1e-3 / n
        cyc, ref = clk.tolist()
        mhz = cyc / ref * 100
        mid = samples[1:-1] or samples
        print(f"{name:26s} {flops / dt / 1e12:7.1f} TFLOP/s   shader clock {mhz:6.0f} MHz ({flops / dt / 256 / (mhz * 1e6):5.0f} flop/clk/CU of 4096)"
              f"   rocm-smi: " + " ".join(f"{w:.0f}W/{c}MHz" for w, c in mid[:5]), flush=True)


if __name__ == "__main__":
    main()
